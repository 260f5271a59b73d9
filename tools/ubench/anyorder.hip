// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx boards".)
// A spins ~200 us; B, launched behind it on the same stream with the flag, stamps its start.  Also: the cost of a stream event hop vs
// hipStreamWriteValue32 / hipStreamWaitValue32.   hipcc --offload-arch=gfx950 -O2 anyorder.hip -o anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long* t, unsigned long long ticks) {
	const unsigned long long t0 = wall_clock64();
	if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
	if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
__global__ void stamp(unsigned long long* t) { if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = wall_clock64(); t[1] = t[0]; } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
	unsigned long long* d; CK(hipMalloc(&d, 64 * 8)); CK(hipMemset(d, 0, 64 * 8));
	hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	unsigned long long h[64];
	for (int flag = 0; flag < 2; flag++) {
		for (int rep = 0; rep < 3; rep++) {
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 20000ull);                       // 200 us at 100 MHz
			hipExtLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, d + 2);
			CK(hipStreamSynchronize(s));
			CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
			printf("flag %d: A [0, %.1f us], B starts at %.1f us\n", flag, (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0);
		}
	}
	// event hop: A on s, B on s2 behind an event; gap between A's end and B's start
	hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
	for (int mode = 0; mode < 3; mode++) {
		unsigned int* flagw; CK(hipMalloc(&flagw, 64)); CK(hipMemset(flagw, 0, 64));
		double sum = 0; int n = 0;
		for (int rep = 0; rep < 20; rep++) {
			if (mode == 0) { hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, ev, 0, d, 2000ull); CK(hipStreamWaitEvent(s2, ev, 0)); }
			else if (mode == 1) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000ull); CK(hipEventRecord(ev, s)); CK(hipStreamWaitEvent(s2, ev, 0)); }
			else { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000ull); CK(hipStreamWriteValue32(s, flagw, rep + 1, 0)); CK(hipStreamWaitValue32(s2, flagw, rep + 1, hipStreamWaitValueGte, 0xffffffffu)); }
			hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s2, d + 2);
			CK(hipStreamSynchronize(s2)); CK(hipStreamSynchronize(s));
			CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
			if (rep >= 4) { sum += ((long long)h[2] - (long long)h[1]) / 100.0; n++; }
		}
		printf("%s: B starts %.2f us after A's last stamp (mean of %d)\n", mode == 0 ? "stop event + wait" : mode == 1 ? "record + wait" : "write value + wait value", sum / n, n);
	}
	// same stream, plain order
	{
		double sum = 0; int n = 0;
		for (int rep = 0; rep < 20; rep++) {
			hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000ull);
			hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, s, d + 2);
			CK(hipStreamSynchronize(s));
			CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
			if (rep >= 4) { sum += ((long long)h[2] - (long long)h[1]) / 100.0; n++; }
		}
		printf("same stream: B starts %.2f us after A's last stamp\n", sum / n);
	}
	return 0;
}
