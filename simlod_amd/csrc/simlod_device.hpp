// simlod_device.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Hardware model these helpers are written for (MI355X / CDNA4): 64-lane wavefronts, 256 CUs in 8 XCDs with
// private, mutually non-coherent L2s, device-scope atomics resolved at the memory side.  Hence:
//   * BlockTable: an LDS open-addressing table (key -> count) per workgroup, the replacement for the per-warp
//     cg::labeled_partition aggregation of progressive_octree_voxels.cu:203-218 — a spatially compact batch sends most of
//     its points to a few dozen counters, and atomics on one word retire at ~88 M/s here: one global atomic per
//     (workgroup, counter) instead of one per wave;
//   * grid_barrier(): monotonic-counter barrier with agent-scope release/acquire (only the rarely executed
//     expand loop uses it; every other dependency is a kernel boundary, which is cheaper on this chip);
//   * all arithmetic that feeds a comparison with the CPU oracle is written operation by operation and the
//     translation unit is compiled with -ffp-contract=off (no FMA contraction, IEEE divide/sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simlod_abi.h"

#define SIMLOD_WAVE 64

// Measurement aids inside the kernels — the phase clocks of one workgroup per builder kernel (Ctl.phaseNs, Ctl.expandNs[0..6], Ctl.voxT: tools/probe.py),
// the per-item and per-bin clocks of the rasteriser (DrawItem::took, the bins' stat words: tools/raster_items.py, raster_bins.py) — exist only in
// builds made with -DSIMLOD_MEASURE=1 (`make -C simlod_amd/csrc variant NAME=measure DEFS=-DSIMLOD_MEASURE=1`, loaded through SIMLOD_HIP_LIB).  The
// product library carries none of them.
#ifndef SIMLOD_MEASURE
#define SIMLOD_MEASURE 0
#endif

namespace simlod {

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// exact 2^level as fp32 (the reference's pow(2.0f, float(level)))
__device__ __forceinline__ float exp2_int(uint32_t level) { return __uint_as_float((127u + level) << 23); }

// fp32 -> u32 quantisation of one coordinate, progressive_octree_voxels.cu:148-155: scale * (p - min) / size,
// evaluated left to right in fp32, truncated (v_cvt_u32_f32 saturates: NaN/negative -> 0).
__device__ __forceinline__ uint32_t quantize(float scale, float p, float mn, float size) {
	float v = scale * (p - mn);
	v = v / size;
	return (uint32_t)v;
}

__device__ __forceinline__ int child_index(uint32_t X, uint32_t Y, uint32_t Z, int level) {
	const int s = SIMLOD_MAX_DEPTH - 1 - level;   // progressive_octree_voxels.cu:171-179
	return (int)((((X >> s) & 1u) << 2) | (((Y >> s) & 1u) << 1) | ((Z >> s) & 1u));
}

__device__ __forceinline__ bool node_is_leaf(const SimlodNode* n) {
	const ulonglong2* c = reinterpret_cast<const ulonglong2*>(n->children);
	ulonglong2 a = c[0], b = c[1], d = c[2], e = c[3];
	return (a.x | a.y | b.x | b.y | d.x | d.y | e.x | e.y) == 0ull;
}

// P descents to the leaves that own the samples (progressive_octree_voxels.cu:169-187) in lockstep, one level per step: the P child-pointer
// loads of a step are independent, so a thread keeps P L2 round trips in flight instead of walking its samples one after the other (a
// descent is a chain of 5-8 dependent loads).
template <int P>
__device__ __forceinline__ void descend_lockstep(SimlodNode* nodes, uint32_t (&cur)[P], uint32_t (&level)[P], const uint32_t (&X)[P], const uint32_t (&Y)[P],
                                                 const uint32_t (&Z)[P], bool (&walking)[P]) {
	bool any = true;
#pragma unroll 1
	for (int step = 0; step < SIMLOD_MAX_DEPTH && any; ++step) {
		SimlodNode* ch[P];
#pragma unroll
		for (int j = 0; j < P; j++)
			ch[j] = walking[j] && level[j] < (uint32_t)SIMLOD_MAX_DEPTH ? nodes[cur[j]].children[child_index(X[j], Y[j], Z[j], (int)level[j])] : nullptr;
		any = false;
#pragma unroll
		for (int j = 0; j < P; j++) {
			if (ch[j] != nullptr) { cur[j] = (uint32_t)(ch[j] - nodes); level[j] += 1u; any = true; }
			else walking[j] = false;
		}
	}
}

__device__ __forceinline__ uint64_t wall_ns() { return (uint64_t)wall_clock64() * 10ull; }   // 100 MHz constant clock

// AllocatorGlobal::alloc for `count` objects of `size` bytes each, as ONE atomic (utils.h.cu:185-197 advances the
// offset by 16*((size+16)/16) per object, so `count` objects advance it by count times that).  Returns the first.
__device__ __forceinline__ uint8_t* persistent_alloc(uint8_t* pers, uint64_t size, uint32_t count) {
	SimlodAllocatorGlobal* a = reinterpret_cast<SimlodAllocatorGlobal*>(pers);
	unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(&a->offset),
	                                   (unsigned long long)(SIMLOD_ALLOC_ROUND(size) * count));
	return pers + old;
}

// ---- grid barrier (expand loop only) --------------------------------------------------------------------------
// One monotonic device-scope counter, zeroed by the host-enqueued prologue of every launch.  Producer side: every
// wave drains its stores, the block syncs, lane 0 issues the agent-scope release (L2 write-back), arrives, polls
// relaxed, then ONE agent-scope acquire (L1 invalidate) and a block sync.  The kernel that uses it (k_expand) is an ORDINARY launch of at
// most one 1024-thread workgroup per CU: a workgroup that waits here waits only for workgroups of its own launch, and nothing those need
// is held by a waiting workgroup — kernels of other streams (the voxel half of the previous batch fills whole CUs) can delay their
// becoming resident, not prevent it, because they end on their own.  hipLaunchCooperativeKernel would add the residency check at launch
// time and ~20 us per launch (measured); the 2 s bound on the poll is the guard against a broken device instead.
__device__ __forceinline__ bool grid_barrier(uint32_t* counter, uint32_t& generation, uint32_t numBlocks, bool forceTimeout = false) {
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	__shared__ int ok;
	generation += 1;
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t target = generation * numBlocks;
		int good = forceTimeout ? 0 : 1;
		uint64_t t0 = wall_clock64();
		while (good && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
			__builtin_amdgcn_s_sleep(8);
			if (wall_clock64() - t0 > 200000000ull) { good = 0; break; }   // 2 s at 100 MHz
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		ok = good;
	}
	__syncthreads();
	return ok != 0;
}

}  // namespace simlod
