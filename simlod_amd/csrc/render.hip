// render.hip — software point/voxel rasteriser for MI355X (gfx950): the `kernel_render` entry point, and the
// one-thread `kernel` of reset.cu.
//
// Replaces modules/progressive_octree/render.cu:1084-1355 (one persistent cooperative CUDA kernel, ~25 grid.sync())
// behind the same argument list, reading the same Node/Chunk image and leaving the same uint64 framebuffer
// (depth bits << 32 | colour) at the same offset of the momentary buffer.  A frame is a chain of ordinary launches
// (clear -> visibility 1 -> visibility 2 -> draw [depth, colour, resolve] -> output); the only cross-workgroup
// traffic inside a launch is device-scope atomics (visible-node list, work queue, framebuffer).
//
// Arithmetic contract: projection = four fp32 dot products evaluated left to right, IEEE divide,
// pixel coordinate in fp64 exactly as `int x = (ndc.x * 0.5 + 0.5) * width` does (render.cu:66-67), no FMA
// contraction anywhere (this file is compiled with -ffp-contract=off), so the pre-EDL framebuffer is bit-identical
// to the CPU oracle's for the same octree image.
#include "simlod_device.hpp"
#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {

static constexpr uint32_t TPB = 256;

// momentary layout of render.cu:1108-1123 (Allocator::alloc rounds every block up to 16 bytes)
static constexpr uint64_t R_OFF_VISIBLE = 0;
static constexpr uint64_t R_OFF_COUNTERS = (uint64_t)SIMLOD_MAX_VISIBLE_NODES * sizeof(SimlodNode);   // 7 x 16 B
static constexpr uint64_t R_OFF_LINES = R_OFF_COUNTERS + 7 * 16;                                       // 32 B header
static constexpr uint64_t R_OFF_VERTICES = R_OFF_LINES + 32;                                           // 1 M x 16 B
static constexpr uint64_t R_OFF_FB = R_OFF_VERTICES + 16000000ull;

struct RenderArgs {
	uint8_t*     mom;
	SimlodNode*  nodes;
	SimlodStats* stats;
	uint32_t*    colorbuffer;
	uint64_t*    frameStart;
	SimlodMat4   transform, transformUpdate;
	float        width, height, cubeSize, minx, miny, minz, minNodeSize;
	int32_t      W, H, pointSize;
	uint32_t     numPixels, nodeCapacity, frameCounter;
	uint8_t      showPoints, colorByNode, colorByLOD, hqs;
	uint64_t     offWork, offDepth, offColor, offOverflow;
};

// work area: one draw cursor per draw mode
static constexpr int WORK_WORDS = 4;

__device__ __forceinline__ uint32_t* counter_at(const RenderArgs& a, int k) { return reinterpret_cast<uint32_t*>(a.mom + R_OFF_COUNTERS + 16 * k); }
enum { C_VISIBLE = 0, C_POINTS = 1, C_VOXELS = 2, C_INNER = 3, C_LEAVES = 4 };

// ---- clear (render.cu:1126-1131, 233-241) ---------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void r_clear(RenderArgs a) {
	uint64_t* fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	const uint32_t stride = gridDim.x * TPB;
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < a.numPixels; i += stride) fb[i] = SIMLOD_CLEAR_PIXEL;
	if (a.hqs) {
		uint32_t* depth = reinterpret_cast<uint32_t*>(a.mom + a.offDepth);
		unsigned long long* packed = reinterpret_cast<unsigned long long*>(a.mom + a.offColor);
		uint4* overflow = reinterpret_cast<uint4*>(a.mom + a.offOverflow);
		for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < a.numPixels; i += stride) { depth[i] = 0x7f800000u; packed[i] = 0ull; overflow[i] = make_uint4(0, 0, 0, 0); }
	}
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		*a.frameStart = wall_ns();                                    // render.cu:1100-1102
		for (int k = 0; k < 7; k++) *counter_at(a, k) = 0;
		uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
		for (int k = 0; k < WORK_WORDS; k++) work[k] = 0;
		uint32_t* lines = reinterpret_cast<uint32_t*>(a.mom + R_OFF_LINES);
		lines[0] = 0;                                                  // lines->count = 0, render.cu:1118
	}
}

// ---- visibility pass 1: screen-space extent + frustum test per node (render.cu:762-901, math.cuh:154-201) --------
__device__ __forceinline__ float dot_row(const simlod_float4& r, float x, float y, float z) {
	float s = r.x * x;
	s = s + r.y * y;
	s = s + r.z * z;
	s = s + r.w * 1.0f;
	return s;
}

__device__ bool intersects_frustum(const SimlodMat4& m, const float mn[3], const float mx[3]) {
	const simlod_float4* R = m.rows;
	const float m0 = R[0].x, m1 = R[1].x, m2 = R[2].x, m3 = R[3].x;
	const float m4 = R[0].y, m5 = R[1].y, m6 = R[2].y, m7 = R[3].y;
	const float m8 = R[0].z, m9 = R[1].z, m10 = R[2].z, m11 = R[3].z;
	const float m12 = R[0].w, m13 = R[1].w, m14 = R[2].w, m15 = R[3].w;
	const float P[6][4] = {
		{m3 - m0, m7 - m4, m11 - m8, m15 - m12}, {m3 + m0, m7 + m4, m11 + m8, m15 + m12},
		{m3 + m1, m7 + m5, m11 + m9, m15 + m13}, {m3 - m1, m7 - m5, m11 - m9, m15 - m13},
		{m3 - m2, m7 - m6, m11 - m10, m15 - m14}, {m3 + m2, m7 + m6, m11 + m10, m15 + m14}};
	bool inside = true;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		const float x = P[i][0], y = P[i][1], z = P[i][2], w = P[i][3];
		float d2 = x * x; d2 = d2 + y * y; d2 = d2 + z * z;
		const float len = __fsqrt_rn(d2);
		const float nx = x / len, ny = y / len, nz = z / len, c = w / len;
		const float vx = nx > 0.0f ? mx[0] : mn[0];
		const float vy = ny > 0.0f ? mx[1] : mn[1];
		const float vz = nz > 0.0f ? mx[2] : mn[2];
		float d = nx * vx; d = d + ny * vy; d = d + nz * vz; d = d + c;
		if (d < 0.0f) inside = false;
	}
	return inside;
}

__global__ __launch_bounds__(TPB) void r_vis1(RenderArgs a) {
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= numNodes) return;
	SimlodNode* n = a.nodes + i;
	const float nodeSize = a.cubeSize / exp2_int(n->level);
	const float cmin[3] = {a.minx, a.miny, a.minz};
	const uint32_t XYZ[3] = {n->X, n->Y, n->Z};
	float mn[3], mx[3];
#pragma unroll
	for (int k = 0; k < 3; k++) {
		mn[k] = cmin[k] + ((float)XYZ[k] + 0.0f) * nodeSize;
		mx[k] = cmin[k] + ((float)XYZ[k] + 1.0f) * nodeSize;
	}
	float sx[8], sy[8];
#pragma unroll
	for (int k = 0; k < 8; k++) {   // p000, p001, p010, p011, p100, p101, p110, p111 (render.cu:783-790)
		const float x = (k & 4) ? mx[0] : mn[0], y = (k & 2) ? mx[1] : mn[1], z = (k & 1) ? mx[2] : mn[2];
		const float cx = dot_row(a.transformUpdate.rows[0], x, y, z);
		const float cy = dot_row(a.transformUpdate.rows[1], x, y, z);
		const float cw = dot_row(a.transformUpdate.rows[3], x, y, z);
		sx[k] = ((cx / cw) * 0.5f + 0.5f) * a.width;
		sy[k] = ((cy / cw) * 0.5f + 0.5f) * a.height;
	}
	const float minx = fminf(fminf(fminf(sx[0], sx[1]), fminf(sx[2], sx[3])), fminf(fminf(sx[4], sx[5]), fminf(sx[6], sx[7])));
	const float maxx = fmaxf(fmaxf(fmaxf(sx[0], sx[1]), fmaxf(sx[2], sx[3])), fmaxf(fmaxf(sx[4], sx[5]), fmaxf(sx[6], sx[7])));
	const float miny = fminf(fminf(fminf(sy[0], sy[1]), fminf(sy[2], sy[3])), fminf(fminf(sy[4], sy[5]), fminf(sy[6], sy[7])));
	const float maxy = fmaxf(fmaxf(fmaxf(sy[0], sy[1]), fmaxf(sy[2], sy[3])), fmaxf(fmaxf(sy[4], sy[5]), fmaxf(sy[6], sy[7])));
	const float dx = maxx - minx, dy = maxy - miny;
	const bool visible = intersects_frustum(a.transformUpdate, mn, mx) && (n->numPoints > 0 || n->numVoxels > 0);
	const double lim = 2.0 * (double)a.minNodeSize;
	n->visible = visible ? 1 : 0;
	n->isLarge = ((double)dx > lim || (double)dy > lim) ? 1 : 0;                      // render.cu:860-861
}

// ---- visibility pass 2: emit the disjunct set of nodes to draw (render.cu:746-756, 906-933) ----------------------
__device__ void make_visible(const RenderArgs& a, const SimlodNode* node) {
	const uint32_t idx = atomicAdd(counter_at(a, C_VISIBLE), 1u);
	if (idx >= SIMLOD_MAX_VISIBLE_NODES) { atomicOr(&a.stats->dbg, SIMLOD_ERR_VISIBLE_OVERFLOW); return; }
	const ulonglong1* src = reinterpret_cast<const ulonglong1*>(node);
	ulonglong1* dst = reinterpret_cast<ulonglong1*>(a.mom + R_OFF_VISIBLE + (uint64_t)idx * sizeof(SimlodNode));
#pragma unroll
	for (int k = 0; k < (int)(sizeof(SimlodNode) / 8); k++) dst[k] = src[k];
	if (node->numPoints > 0) { atomicAdd(counter_at(a, C_LEAVES), 1u); atomicAdd(counter_at(a, C_POINTS), node->numPoints); }
	else if (node->numVoxels > 0) { atomicAdd(counter_at(a, C_INNER), 1u); atomicAdd(counter_at(a, C_VOXELS), node->numVoxels); }
}

__global__ __launch_bounds__(TPB) void r_vis2(RenderArgs a) {
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= numNodes) return;
	const SimlodNode* n = a.nodes + i;
	if (!n->isLarge) return;
	if (!node_is_leaf(n)) {
		for (int k = 0; k < 8; k++) {
			const SimlodNode* ch = n->children[k];
			if (ch == nullptr || ch->isLarge || !ch->visible) continue;
			make_visible(a, ch);
		}
	} else if (n->visible) {
		make_visible(a, n);
	}
}

// ---- draw ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lod_color(int level) {   // render.cu:38-59
	const uint32_t SPECTRAL[8] = {0x4f3ed5, 0x436df4, 0x61aefd, 0x8be0fe, 0x98f5e6, 0xa4ddab, 0xa5c266, 0xbd8832};
	int index = (int)((float)(8 - level) * 1.8f);
	index = index < 0 ? 0 : (index > 7 ? 7 : index);
	return SPECTRAL[index];
}

__device__ uint32_t node_color(const SimlodNode* n) {   // (getID() % 127) * 123456789, structures.cuh:118-141, render.cu:75
	uint64_t id = (uint64_t)(int64_t)(n->name[0] == 'r' ? 1 : 0);
	for (int i = 1; i <= 9; i++) id |= (uint64_t)(int64_t)(int32_t)(((uint32_t)((int)n->name[i] - '0')) << (3 * i));
	for (int i = 10; i <= 17; i++) id |= ((uint64_t)(int64_t)((int)n->name[i] - '0')) << (3 * i);
	id |= ((uint64_t)(int64_t)((int)n->name[18] - '0')) << 53;
	return (uint32_t)((id % 127ull) * 123456789ull);
}

enum { MODE_MIN64 = 0, MODE_DEPTH = 1, MODE_COLOR = 2 };

struct DrawCtx {
	simlod_float4 r0, r1, r3;
	float  width, height;
	double wlim, hlim;
	int    W, H, pointSize;
	uint32_t numPixels;
	uint64_t* fb;
	uint32_t* depth;
	unsigned long long* color;   // HQS colour sums, packed: B (14 bits) | G << 14 | R << 28 | count << 42
	unsigned long long* overflow;// 2 x u64 per pixel {R | G << 32, B | count << 32}: samples beyond the 64th of a pixel
};

template <int MODE>
__device__ __forceinline__ void draw_sample(const DrawCtx& c, const float4 p, const uint32_t overrideColor, const bool useOverride) {
	// render.cu:62-70 — transform, perspective divide, pixel in fp64
	const float cx = dot_row(c.r0, p.x, p.y, p.z);
	const float cy = dot_row(c.r1, p.x, p.y, p.z);
	const float depth = dot_row(c.r3, p.x, p.y, p.z);
	const float nx = cx / depth, ny = cy / depth;
	const double fx = ((double)nx * 0.5 + 0.5) * (double)c.width;
	const double fy = ((double)ny * 0.5 + 0.5) * (double)c.height;
	const int x = (int)fx, y = (int)fy;                 // v_cvt_i32_f64 saturates; NaN -> 0: rejected below either way
	if (!(x > 1 && (double)x < c.wlim)) return;
	if (!(y > 1 && (double)y < c.hlim)) return;
	if (MODE != MODE_MIN64 && !(depth > 0.0f)) return;   // render.cu:295, 371, 456, 558
	const uint32_t dbits = __float_as_uint(depth);
	const uint32_t color = useOverride ? overrideColor : __float_as_uint(p.w);
	for (int ox = 0; ox < c.pointSize; ox++)
	for (int oy = 0; oy < c.pointSize; oy++) {
		const int px = min(max(x + ox, 0), c.W), py = min(max(y + oy, 0), c.H);   // render.cu:91-92 clamps to W, not W-1
		const uint32_t pixel = (uint32_t)px + (uint32_t)c.W * (uint32_t)py;
		if (pixel >= c.numPixels) continue;                 // only reachable for pointSize >= 4 (out of bounds in the reference)
		if (MODE == MODE_MIN64) {
			const unsigned long long enc = ((unsigned long long)dbits << 32) | color;
			if (enc < c.fb[pixel]) atomicMin(reinterpret_cast<unsigned long long*>(&c.fb[pixel]), enc);   // render.cu:95-100
		} else if (MODE == MODE_DEPTH) {
			if (dbits < c.depth[pixel]) atomicMin(&c.depth[pixel], dbits);                                // render.cu:304-308
		} else {
			const float fbDepth = __uint_as_float(c.depth[pixel]);
			if (depth < fbDepth * 1.01f) {                                                                 // render.cu:485-493
				// ONE 64-bit atomic per accepted sample: the sums of R, G, B and the count share a word (14 + 14 + 14 + 22 bits).
				// The first 64 samples of a pixel fit without carry (64 * 255 < 2^14); a sample that finds count >= 64 takes its
				// addend back and goes to the 32-bit-per-channel overflow plane.  All arithmetic is modular, so transient carries
				// of samples that are about to retract do not disturb the final sums (at most 64 samples ever stay).
				const unsigned long long r = color & 0xffu, g = (color >> 8) & 0xffu, b = (color >> 16) & 0xffu;
				const unsigned long long pk = b | (g << 14) | (r << 28) | (1ull << 42);
				const unsigned long long old = atomicAdd(&c.color[pixel], pk);
				if ((old >> 42) >= 64ull) {
					atomicAdd(&c.color[pixel], 0ull - pk);
					atomicAdd(&c.overflow[2 * pixel + 0], r | (g << 32));
					atomicAdd(&c.overflow[2 * pixel + 1], b | (1ull << 32));
				}
			}
		}
	}
}

template <int MODE>
__device__ __forceinline__ void draw_list(const DrawCtx& c, const SimlodChunk* chunk, uint32_t count, uint32_t overrideColor, bool useOverride) {
	uint32_t done = 0;
	while (done < count && chunk != nullptr) {        // render.cu:106-159: chunk i holds samples [1000 i, 1000 i + 1000)
		const SimlodChunk* next = chunk->next;         // start the pointer chase before streaming the chunk
		const uint32_t inChunk = min(count - done, SIMLOD_POINTS_PER_CHUNK);
		const float4* src = reinterpret_cast<const float4*>(chunk->points);
		for (uint32_t j = threadIdx.x; j < inChunk; j += TPB) draw_sample<MODE>(c, src[j], overrideColor, useOverride);
		done += inChunk;
		chunk = next;
	}
}

template <int MODE>
__global__ __launch_bounds__(TPB) void r_draw(RenderArgs a) {
	if (!a.showPoints) return;
	__shared__ uint32_t sh_idx;
	DrawCtx c;
	c.r0 = a.transform.rows[0]; c.r1 = a.transform.rows[1]; c.r3 = a.transform.rows[3];
	c.width = a.width; c.height = a.height;
	c.wlim = (double)a.width - 2.0; c.hlim = (double)a.height - 2.0;
	c.W = a.W; c.H = a.H; c.pointSize = a.pointSize; c.numPixels = a.numPixels;
	c.fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	c.depth = reinterpret_cast<uint32_t*>(a.mom + a.offDepth);
	c.color = reinterpret_cast<unsigned long long*>(a.mom + a.offColor);
	c.overflow = reinterpret_cast<unsigned long long*>(a.mom + a.offOverflow);
	uint32_t* cursor = reinterpret_cast<uint32_t*>(a.mom + a.offWork) + MODE;
	const uint32_t numVisible = min(*counter_at(a, C_VISIBLE), SIMLOD_MAX_VISIBLE_NODES);
	const SimlodNode* visible = reinterpret_cast<const SimlodNode*>(a.mom + R_OFF_VISIBLE);
	// Workgroup-level node queue (render.cu:179-207).  The first node of a workgroup is its own index — 2048 workgroups
	// fetching their first item from ONE counter would serialise at ~11 ns per atomic (22 us of start-up) — the following
	// ones come from a shared cursor that starts behind the statically assigned range.
	uint32_t idx = blockIdx.x;
	while (idx < numVisible) {
		const SimlodNode* node = visible + idx;
		uint32_t overrideColor = 0; bool useOverride = false;
		if (MODE != MODE_DEPTH) {
			if (a.colorByNode) { overrideColor = node_color(node); useOverride = true; }
			else if (a.colorByLOD) { overrideColor = lod_color((int)node->level); useOverride = true; }
		}
		draw_list<MODE>(c, node->points, node->numPoints, overrideColor, useOverride);
		draw_list<MODE>(c, node->voxelChunks, node->numVoxels, overrideColor, useOverride);
		__syncthreads();
		if (threadIdx.x == 0) sh_idx = gridDim.x + atomicAdd(cursor, 1u);
		__syncthreads();
		idx = sh_idx;
	}
}

// ---- HQS resolve (render.cu:607-632) -------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void r_resolve(RenderArgs a) {
	uint64_t* fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	const uint32_t* depth = reinterpret_cast<const uint32_t*>(a.mom + a.offDepth);
	const unsigned long long* packed = reinterpret_cast<const unsigned long long*>(a.mom + a.offColor);
	const uint4* overflow = reinterpret_cast<const uint4*>(a.mom + a.offOverflow);
	const uint32_t stride = gridDim.x * TPB;
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < a.numPixels; i += stride) {
		const unsigned long long pk = packed[i];
		uint4 s = overflow[i];                           // {R, G, B, count} of the samples beyond the 64th
		s.x += (uint32_t)((pk >> 28) & 0x3fffu); s.y += (uint32_t)((pk >> 14) & 0x3fffu); s.z += (uint32_t)(pk & 0x3fffu); s.w += (uint32_t)(pk >> 42);
		if (s.w == 0u) continue;
		const uint32_t rgba = ((s.x / s.w) & 0xffu) | (((s.y / s.w) & 0xffu) << 8) | (((s.z / s.w) & 0xffu) << 16) | (255u << 24);
		fb[i] = ((uint64_t)depth[i] << 32) | rgba;
	}
}

// ---- output: Stats (render.cu:1244-1252), EDL (:1255-1325, every full 16x16 tile), surface write (:1334-1343) ---------
__global__ __launch_bounds__(TPB) void r_output(RenderArgs a) {
	const uint64_t* fb = reinterpret_cast<const uint64_t*>(a.mom + R_OFF_FB);
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		SimlodStats* s = a.stats;
		s->numVisibleNodes = min(*counter_at(a, C_VISIBLE), SIMLOD_MAX_VISIBLE_NODES);
		s->numVisibleInner = *counter_at(a, C_INNER);
		s->numVisibleLeaves = *counter_at(a, C_LEAVES);
		s->numVisiblePoints = *counter_at(a, C_POINTS);
		s->numVisibleVoxels = *counter_at(a, C_VOXELS);
		s->frameID = a.frameCounter;
	}
	if (a.colorbuffer == nullptr) return;
	const int edlW = (a.W / 16) * 16, edlH = (a.H / 16) * 16;
	const int last = (int)a.numPixels - 1;
	const uint32_t stride = gridDim.x * TPB;
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < a.numPixels; i += stride) {
		const uint64_t enc = fb[i];
		uint32_t color = (uint32_t)enc;
		const int x = (int)(i % (uint32_t)a.W), y = (int)(i / (uint32_t)a.W);
		if (x < edlW && y < edlH) {
			const float lp = __log2f(__uint_as_float((uint32_t)(enc >> 32)));
			// the four neighbours int(1.5 * sin/cos(k * 3.1415 / 2)) of render.cu:1296-1300: (0,+1), (+1,0), (0,-1), (-1,0)
			const int offs[4] = {a.W, 1, -a.W, -1};
			float sum = 0.0f;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				int idx = (int)i + offs[k];
				idx = idx < 0 ? 0 : (idx > last ? last : idx);
				const float ln = __log2f(__uint_as_float((uint32_t)(fb[idx] >> 32)));
				const float d = lp - ln;
				sum = sum + (d > 0.0f ? d : 0.0f);                 // max(NaN, 0) = 0
			}
			const float response = sum / 50.0f;
			const float shade = __expf((float)((double)(-response) * 300.0 * (double)0.4f));
			const uint32_t R = (uint32_t)(shade * (float)(color & 0xffu));
			const uint32_t G = (uint32_t)(shade * (float)((color >> 8) & 0xffu));
			const uint32_t B = (uint32_t)(shade * (float)((color >> 16) & 0xffu));
			color = R | (G << 8) | (B << 16) | (255u << 24);
		}
		a.colorbuffer[i] = color;
	}
}

// ---- reset.cu:20-86 -------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_reset(uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                                               uint32_t* batchSizes, uint32_t frameCounter) {
	// The allocator starts at offset 16 and the root's occupancy grid is its first allocation (reset.cu:42-67), so the
	// grid sits at pers + 16: every workgroup can clear its share without waiting for thread 0.
	uint4* grid = reinterpret_cast<uint4*>(pers + 16);
	for (uint32_t w = blockIdx.x * TPB + threadIdx.x; w < SIMLOD_GRID_NUM_WORDS / 4; w += gridDim.x * TPB) grid[w] = make_uint4(0, 0, 0, 0);
	if (blockIdx.x != 0 || threadIdx.x != 0) return;
	SimlodAllocatorGlobal* alloc = reinterpret_cast<SimlodAllocatorGlobal*>(pers);
	alloc->buffer = pers;
	alloc->offset = 16 + SIMLOD_ALLOC_ROUND(sizeof(SimlodOccupancyGrid));
	SimlodStats s{};
	s.numNodes = 1;
	s.frameID = frameCounter;
	*stats = s;
	SimlodNode* root = nodes;
	for (int k = 0; k < 8; k++) root->children[k] = nullptr;
	root->isFiltered = 0;
	root->counter = 0; root->numPoints = 0; root->level = 0;
	root->X = 0; root->Y = 0; root->Z = 0;
	root->countIteration = 0;
	for (int k = 0; k < 20; k++) root->name[k] = 0;
	root->name[0] = 'r';
	root->numVoxels = 0; root->numVoxelsStored = 0;
	root->voxelChunks = nullptr;
	root->points = nullptr;     // not in reset.cu: a list surviving the allocator restart would alias new allocations
	root->grid = reinterpret_cast<SimlodOccupancyGrid*>(pers + 16);
	*numBatchesUploaded = 0;
	for (uint32_t k = 0; k < SIMLOD_BATCH_STREAM_SIZE; k++) batchSizes[k] = 0;
}

// ---- host side --------------------------------------------------------------------------------------------------------------
static inline uint64_t align16(uint64_t v) { return (v + 15) / 16 * 16; }

uint64_t render_buffer_bytes(uint32_t width, uint32_t height) {
	const uint64_t px = (uint64_t)width * height;
	return R_OFF_FB + align16(px * 8) + 256 + align16(px * 4) + align16(px * 8) + px * 16 + 256;
}

int launch_reset(const SimlodUniforms* u, uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                 uint32_t* batchSizes, hipStream_t stream) {
	SIMLOD_LAUNCH(k_reset, dim3(64), dim3(TPB), stream, pers, nodes, stats, numBatchesUploaded, batchSizes, (uint32_t)u->frameCounter);
	return (int)hipGetLastError();
}

int launch_render(uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                  uint64_t* frameStart, hipStream_t stream) {
	RenderArgs a{};
	a.mom = reinterpret_cast<uint8_t*>(buffer); a.nodes = nodes; a.stats = stats; a.colorbuffer = colorbuffer; a.frameStart = frameStart;
	a.transform = u->transform; a.transformUpdate = u->transform_updateBound;
	a.width = u->width; a.height = u->height;
	a.W = (int)u->width; a.H = (int)u->height;
	if (a.W <= 0 || a.H <= 0) return (int)hipErrorInvalidValue;
	a.numPixels = (uint32_t)a.W * (uint32_t)a.H;
	const float bx = u->boxMax.x - u->boxMin.x, by = u->boxMax.y - u->boxMin.y, bz = u->boxMax.z - u->boxMin.z;
	a.cubeSize = fmaxf(fmaxf(bx, by), bz);                               // render.cu:1135-1137
	a.minx = u->boxMin.x; a.miny = u->boxMin.y; a.minz = u->boxMin.z;
	a.minNodeSize = u->minNodeSize;
	a.pointSize = u->pointSize;
	a.nodeCapacity = node_capacity();
	a.frameCounter = (uint32_t)u->frameCounter;
	a.showPoints = u->showPoints; a.colorByNode = u->colorByNode; a.colorByLOD = u->colorByLOD; a.hqs = u->useHighQualityShading;
	a.offWork = R_OFF_FB + align16((uint64_t)a.numPixels * 8);
	a.offDepth = a.offWork + 256;
	a.offColor = a.offDepth + align16((uint64_t)a.numPixels * 4);
	a.offOverflow = a.offColor + align16((uint64_t)a.numPixels * 8);

	const DeviceInfo& dev = device_info();
	const uint32_t gridPixels = dev.numCUs * 8;
	const uint32_t gridNodes = (a.nodeCapacity + TPB - 1) / TPB;
	const uint32_t gridDraw = dev.numCUs * 8;
	SIMLOD_LAUNCH(r_clear, dim3(gridPixels), dim3(TPB), stream, a);
	SIMLOD_LAUNCH(r_vis1, dim3(gridNodes), dim3(TPB), stream, a);
	SIMLOD_LAUNCH(r_vis2, dim3(gridNodes), dim3(TPB), stream, a);
	if (a.hqs) {
		SIMLOD_LAUNCH(r_draw<MODE_DEPTH>, dim3(gridDraw), dim3(TPB), stream, a);
		SIMLOD_LAUNCH(r_draw<MODE_COLOR>, dim3(gridDraw), dim3(TPB), stream, a);
		SIMLOD_LAUNCH(r_resolve, dim3(gridPixels), dim3(TPB), stream, a);
	} else {
		SIMLOD_LAUNCH(r_draw<MODE_MIN64>, dim3(gridDraw), dim3(TPB), stream, a);
	}
	SIMLOD_LAUNCH(r_output, dim3(gridPixels), dim3(TPB), stream, a);
	if (profile_enabled()) profile_close(stream);
	return (int)hipGetLastError();
}

}  // namespace simlod
