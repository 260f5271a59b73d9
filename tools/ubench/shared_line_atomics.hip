// Micro-benchmark: do device-scope atomics get slow when the SAME lines are concurrently read by plain loads from every CU?
// (the access shape of k_sample: ~1 M probes of occupancy words, ~100 k atomicOr on a subset of them)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ inline uint32_t rng(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// every thread probes one random word of `words`; a fraction 1/16 of the threads then atomicOr's its word.
// LOADMODE: 0 plain, 1 agent-scope relaxed atomic load (sc1), 2 nontemporal, 3 no probe at all (atomics only, same addresses)
template <int LOADMODE>
__global__ void k(uint32_t* buf, uint32_t words, uint32_t n, uint32_t* sink, uint32_t salt) {
	uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t w = rng(i + salt) % words;
	uint32_t v = 0;
	if (LOADMODE == 0) v = buf[w];
	else if (LOADMODE == 1) v = __hip_atomic_load(&buf[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	else if (LOADMODE == 2) v = __builtin_nontemporal_load(&buf[w]);
	uint32_t acc = v;
	if ((rng(i * 7 + salt) & 15u) == 0u) acc += atomicOr(&buf[w], 1u << (i & 31));
	if (acc == 0xdeadbeef) sink[0] = acc;
}
template <int LOADMODE> float run(uint32_t* buf, uint32_t words, uint32_t n, uint32_t* sink) {
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipLaunchKernelGGL((k<LOADMODE>), dim3((n + 255) / 256), dim3(256), 0, 0, buf, words, n, sink, 1u);
	hipDeviceSynchronize(); hipEventRecord(a);
	for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<LOADMODE>), dim3((n + 255) / 256), dim3(256), 0, 0, buf, words, n, sink, 11u + r);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b); return ms / 5 * 1e3f;
}
int main() {
	uint32_t *buf, *sink; hipMalloc(&buf, 256u << 20); hipMalloc(&sink, 64); hipMemset(buf, 0, 256u << 20);
	const uint32_t n = 1u << 20;
	for (uint32_t words : {2048u, 16384u, 1u << 18, 1u << 21, 36u << 20}) {
		printf("region %8.2f MB (%7u lines), 1M probes + 65k atomics: plain %.1f us | sc1 load %.1f us | nt load %.1f us | atomics only %.1f us\n",
		       words * 4.0 / 1e6, words / 32, run<0>(buf, words, n, sink), run<1>(buf, words, n, sink), run<2>(buf, words, n, sink), run<3>(buf, words, n, sink));
	}
	return 0;
}
