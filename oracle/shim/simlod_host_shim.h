/* simlod_host_shim.h — TEST INFRASTRUCTURE ONLY (oracle/_ref).
 *
 * A single-thread host stand-in for the handful of CUDA device-side facilities the reference's
 * NVRTC `-default-device` sources use, so that progressive_octree_voxels.cu, render.cu, reset.cu and
 * utils.cu compile UNMODIFIED, in place under /root/reference, as plain host C++ (recipe: SURVEY.md §8c).
 * The resulting shared objects are the ground truth the CPU restatement (oracle/simlod_oracle.c) and
 * the HIP kernels are checked against.  Nothing in the product path includes this file.
 *
 * Model: gridDim = blockDim = 1, every cooperative-groups group has one thread, sync() is a no-op,
 * atomics are plain read-modify-writes, %globaltimer reads 0 (so the 10 ms ingest budget of
 * progressive_octree_voxels.cu:939 never triggers).
 * grid.num_blocks() returns 0x7fffffff so that render.cu:1273 computes tilesPerBlock = 0 and the
 * launch-geometry-dependent EDL pass is skipped: the framebuffer that comes back is the
 * pre-EDL uint64 image.
 *
 * -DSIMLOD_SHIM_EDL (oracle/_ref/libref_render_edl.so): the EDL pass runs, one in-tile thread per kernel call.
 * grid.num_blocks() is only used by the EDL block (render.cu:1273, 1279); it returns 1 — ONE block takes every full 16x16 tile —
 * and from that call on block.thread_rank() returns simlod_shim_edl_rank instead of 0, so a call shades pixel #rank of every tile
 * (render.cu:1284).  256 calls with rank = 0..255, each re-rendering the same frame, shade every pixel of every full tile with the
 * reference's own EDL arithmetic; the caller collects pixel #rank of each tile after call #rank.
 */
#pragma once

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <initializer_list>
#include <type_traits>

#ifndef __CUDACC__
#define __CUDACC__ 1   /* keeps helper_math.h from redefining libc's fminf/fmaxf/min/max */
#endif
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline
#define __constant__

/* ---- vector types (builtin_types.h / vector_types.h) ---------------------------------------- */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct dim3 { unsigned x = 1, y = 1, z = 1; };

inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int2 make_int2(int x, int y) { return int2{x, y}; }
inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { return uint3{x, y, z}; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }

/* ---- launch geometry: one thread --------------------------------------------------------------- */
static const dim3 blockDim{1, 1, 1};
static const dim3 gridDim{1, 1, 1};
static const uint3 blockIdx{0, 0, 0};
static const uint3 threadIdx{0, 0, 0};

/* ---- scalar min/max with CUDA's overload behaviour -------------------------------------------- */
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
template <class A, class B, class C = typename std::common_type<A, B>::type,
          class = typename std::enable_if<!(std::is_same<A, B>::value && std::is_floating_point<A>::value)>::type>
inline C min(A a, B b) { C x = (C)a, y = (C)b; return x < y ? x : y; }
template <class A, class B, class C = typename std::common_type<A, B>::type,
          class = typename std::enable_if<!(std::is_same<A, B>::value && std::is_floating_point<A>::value)>::type>
inline C max(A a, B b) { C x = (C)a, y = (C)b; return x > y ? x : y; }

inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __powf(float x, float y) { return powf(x, y); }

/* ---- serial atomics ------------------------------------------------------------------------------ */
template <class T, class U> inline T atomicAdd(T* p, U v) { T old = *p; *p = (T)(old + (T)v); return old; }
template <class T, class U> inline T atomicOr(T* p, U v)  { T old = *p; *p = (T)(old | (T)v); return old; }
template <class T, class U> inline T atomicMin(T* p, U v) { T old = *p; if ((T)v < old) *p = (T)v; return old; }
template <class T, class U> inline T atomicMax(T* p, U v) { T old = *p; if ((T)v > old) *p = (T)v; return old; }

/* ---- cooperative groups ---------------------------------------------------------------------------- */
#ifdef SIMLOD_SHIM_EDL
extern "C" { inline int simlod_shim_edl_rank = 0; inline int simlod_shim_in_edl = 0; }
#endif
namespace cooperative_groups {
struct grid_group {
	void sync() const {}
	unsigned long long thread_rank() const { return 0; }
	unsigned long long size() const { return 1; }
	unsigned long long num_threads() const { return 1; }
	unsigned block_rank() const { return 0; }                /* colorfilter.cu:192 */
#ifdef SIMLOD_SHIM_EDL
	unsigned num_blocks() const { simlod_shim_in_edl = 1; return 1u; }
#else
	unsigned num_blocks() const { return 0x7fffffffu; }   /* disables EDL, see header comment */
#endif
};
struct thread_block {
	void sync() const {}
#ifdef SIMLOD_SHIM_EDL
	unsigned thread_rank() const { return simlod_shim_in_edl ? (unsigned)simlod_shim_edl_rank : 0u; }
#else
	unsigned thread_rank() const { return 0; }
#endif
	unsigned size() const { return 1; }
	unsigned num_threads() const { return 1; }
	dim3 group_index() const { return dim3{0, 0, 0}; }
	dim3 thread_index() const { return dim3{0, 0, 0}; }
};
struct coalesced_group {
	void sync() const {}
	unsigned thread_rank() const { return 0; }
	unsigned size() const { return 1; }
	unsigned num_threads() const { return 1; }
};
inline grid_group this_grid() { return grid_group{}; }
inline thread_block this_thread_block() { return thread_block{}; }
inline coalesced_group coalesced_threads() { return coalesced_group{}; }
template <class L> inline coalesced_group labeled_partition(const coalesced_group&, L) { return coalesced_group{}; }
}  // namespace cooperative_groups

/* ---- surface write: the "GL colour buffer" is a linear host uint32 image ------------------- */
typedef unsigned long long cudaSurfaceObject_t;
extern "C" { inline int simlod_shim_surface_width = 0; }   /* set by the test driver through dlsym */
template <class T> inline void surf2Dwrite(T value, cudaSurfaceObject_t surf, int xBytes, int y) {
	if (surf == 0) return;
	T* img = reinterpret_cast<T*>(surf);
	img[(xBytes / (int)sizeof(T)) + y * simlod_shim_surface_width] = value;
}

/* ---- the reference re-typedefs the <stdint.h> names and re-defines strlen (utils.h.cu:11-25) - */
#define int8_t   ref_int8_t
#define uint8_t  ref_uint8_t
#define int16_t  ref_int16_t
#define uint16_t ref_uint16_t
#define int32_t  ref_int32_t
#define uint32_t ref_uint32_t
#define int64_t  ref_int64_t
#define uint64_t ref_uint64_t
#define strlen   ref_strlen

/* ---- swallow the PTX %globaltimer reads (utils.h.cu:312,320): `asm volatile("…" : "=l"(nanotime));`
 *      becomes `nanotime = 0;` */
#define asm
#define volatile(...) nanotime = 0
