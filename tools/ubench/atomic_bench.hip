// Micro-benchmark: device-scope atomics on MI355X, the access shapes of k_sample / r_draw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE, int SERIAL>
__global__ void k(uint32_t* buf, uint32_t words, uint32_t n, uint32_t* sink, uint32_t salt) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	uint32_t acc = 0;
#pragma unroll 1
	for (int s = 0; s < SERIAL; s++) {
		uint32_t w = rng(i * 16 + s + salt + acc * 0) % words;
		if (MODE == 0) acc += atomicOr(&buf[w], 1u << (i & 31));                  // returning, result used
		else if (MODE == 1) atomicOr(&buf[w], 1u << (i & 31));                     // fire and forget
		else if (MODE == 2) acc += buf[w];                                         // plain load
		else if (MODE == 3) acc += __hip_atomic_fetch_or(&buf[w], 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (MODE == 4) { if (SERIAL > 1) w = rng(i * 16 + s + salt + acc) % words; acc += atomicOr(&buf[w], 1u << (i & 31)); }  // dependent chain
		else if (MODE == 6) { uint32_t v = buf[w]; if ((v >> (i & 31) & 1u) == 0u || salt == 0xffffffffu) acc += atomicOr(&buf[w], 1u << (i & 31)); acc += v; }   // test-then-set
		else if (MODE == 7) { uint32_t v = __hip_atomic_load(&buf[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if ((v >> (i & 31) & 1u) == 0u) acc += atomicOr(&buf[w], 1u << (i & 31)); acc += v; }
		else if (MODE == 8) { uint32_t v = __builtin_nontemporal_load(&buf[w]); if ((v >> (i & 31) & 1u) == 0u) acc += atomicOr(&buf[w], 1u << (i & 31)); acc += v; }
		else if (MODE == 5) atomicMin((unsigned long long*)&buf[(w & ~1u)], ((unsigned long long)i << 32) | w);   // 64-bit min, no return
	}
	if (acc == 0xdeadbeef) sink[0] = acc;
}

template <int MODE, int SERIAL>
float run(uint32_t* buf, uint32_t words, uint32_t n, uint32_t* sink) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL((k<MODE, SERIAL>), dim3((n + 255) / 256), dim3(256), 0, 0, buf, words, n, sink, 1u);
	hipDeviceSynchronize();
	hipEventRecord(a);
	for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<MODE, SERIAL>), dim3((n + 255) / 256), dim3(256), 0, 0, buf, words, n, sink, 7u + r);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	return ms / 5;
}

int main() {
	uint32_t* buf; uint32_t* sink;
	const uint32_t maxWords = 64u << 20;   // 256 MB
	CK(hipMalloc(&buf, (size_t)maxWords * 4)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 0, (size_t)maxWords * 4));
	const uint32_t n = 1u << 20;
	for (uint32_t words : {64u, 1024u, 6400u, 1u << 16, 1u << 20, 36u << 20}) {
		printf("region %6.1f MB, %u threads: ", words * 4.0 / 1e6, n);
		float t;
		t = run<0, 1>(buf, words, n, sink); printf("ret-atomicOr x1 %.1f us (%.2f G/s) | ", t * 1e3, n / t / 1e6);
		t = run<1, 1>(buf, words, n, sink); printf("noret x1 %.1f us (%.2f G/s) | ", t * 1e3, n / t / 1e6);
		t = run<2, 1>(buf, words, n, sink); printf("load x1 %.1f us | ", t * 1e3);
		t = run<3, 1>(buf, words, n, sink); printf("wg-scope ret x1 %.1f us | ", t * 1e3);
		t = run<0, 4>(buf, words, n, sink); printf("ret x4 indep %.1f us | ", t * 1e3);
		t = run<4, 4>(buf, words, n, sink); printf("ret x4 dependent %.1f us | ", t * 1e3);
		t = run<1, 4>(buf, words, n, sink); printf("noret x4 %.1f us | ", t * 1e3);
		t = run<5, 1>(buf, words, n, sink); printf("min64 noret x1 %.1f us (%.2f G/s) | ", t * 1e3, n / t / 1e6);
		t = run<5, 4>(buf, words, n, sink); printf("min64 noret x4 %.1f us\n", t * 1e3);
		hipMemset(buf, 0, (size_t)words * 4);
		t = run<6, 1>(buf, words, n, sink); printf("     test-then-set plain x1 %.1f us | ", t * 1e3);
		hipMemset(buf, 0, (size_t)words * 4);
		t = run<7, 1>(buf, words, n, sink); printf("test(sc1 load)-then-set x1 %.1f us | ", t * 1e3);
		hipMemset(buf, 0, (size_t)words * 4);
		t = run<8, 1>(buf, words, n, sink); printf("test(nt load)-then-set x1 %.1f us | ", t * 1e3);
		hipMemset(buf, 0, (size_t)words * 4);
		t = run<6, 4>(buf, words, n, sink); printf("test-then-set plain x4 %.1f us\n", t * 1e3);
	}
	// fewer threads: latency view
	for (uint32_t nn : {1u << 14, 1u << 17}) {
		float t = run<0, 1>(buf, 36u << 20, nn, sink); printf("%u threads ret x1: %.1f us\n", nn, t * 1e3);
	}
	return 0;
}
