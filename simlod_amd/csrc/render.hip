// render.hip — software point/voxel rasteriser for MI355X (gfx950): the `kernel_render` entry point, and the
// one-thread `kernel` of reset.cu.
//
// Replaces modules/progressive_octree/render.cu:1084-1355 (one persistent cooperative CUDA kernel, ~25 grid.sync())
// behind the same argument list, reading the same Node/Chunk image and leaving the same uint64 framebuffer
// (depth bits << 32 | colour) at the same offset of the momentary buffer.  A frame is a chain of ordinary launches
// (clear -> visibility + draw items -> draw [depth, colour, resolve] -> [debug lines] -> output; simlod_launch_render_part runs it
// in four parts for multi-GPU frames); the only cross-workgroup
// traffic inside a launch is device-scope atomics (visible-node list, work queue, framebuffer).
//
// Arithmetic contract: projection = four fp32 dot products evaluated left to right, IEEE divide,
// pixel coordinate in fp64 exactly as `int x = (ndc.x * 0.5 + 0.5) * width` does (render.cu:66-67), no FMA
// contraction anywhere (this file is compiled with -ffp-contract=off), so the pre-EDL framebuffer is bit-identical
// to the CPU oracle's for the same octree image.
#include "simlod_device.hpp"
#include "simlod_hip.h"
#include "simlod_internal.hpp"
#include <atomic>
#include <chrono>

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
	uint64_t     offWork, offItems, offDepth, offColor, offOverflow, offDir, offBinPool, offBinSegs, offBinSegCount;
	uint32_t     itemCap, useTiles, launchSeq;
	uint32_t     useBins, binTilesX, binTiles, binPoolCap, binMinArea, binsPossible;
	uint32_t*    binFeedback;                       // page-locked: the frame's first draw pass stores here how many nodes sort, or would (launch_render)      // screen bins of the samples that leave their item's tile (r_overflow): tiles per row, tiles in all, entries in the pool
	// the builder's leaf chunk table (simlod_internal.hpp LeafTableRef), or table == nullptr: r_visible walks every list
	const uint8_t* leafTable;                          // packed rows (simlod_internal.hpp leaf_row_get), offsets into leafTablePers
	const uint8_t* leafTablePers;
	const uint32_t* leafTableMagic;
	const uint32_t* leafTableBatch;
	const uint64_t* leafTableNodes;
	const uint64_t* leafTableSig;
	uint32_t     leafTableMagicValue, leafTableSlots, leafTableRows;
};

// work area: [0..2] draw cursors of the three draw modes, [3] unused, [4] chunk directory entries in use, [8..11] draw items per size class
static constexpr int WORK_WORDS = 15;                // ... [12] entries taken from the bin pool, [13] samples that were binned (first draw pass), [14] nodes that sort or would
// Draw items are queued by size, biggest first (longest-processing-time order): the draw workgroups take items from one shared
// cursor, and a 64 000-sample item taken last would keep one CU busy long after the others ran dry (measured on the bench frame:
// average workgroup 55 us, slowest 92 us with the items in emission order).  Class of an item = its chunk count: > 16, > 8, > 4, rest;
// class c has its own array (itemCap entries) and counter, position q of the cursor maps to the classes in order.
static constexpr int ITEM_CLASSES = 4;
__device__ __forceinline__ uint32_t item_class(uint32_t chunks) { return chunks > 16u ? 0u : (chunks > 8u ? 1u : (chunks > 4u ? 2u : 3u)); }
// A draw item = up to 32 consecutive chunks (32 000 samples) of one visible node's list (a full leaf is two items; 64 per item: 5 % slower
// on the bench frame, the biggest item is a fifth of a workgroup's whole share; 16: 10 % slower, twice the tile clears and flushes).
// ONE workgroup draws an item, accumulating in a 128 x 128-pixel LDS tile laid over the node's screen box: the LOD rule draws a node
// while its box spans 64..128 pixels (render.cu:893-901), so nearly every sample of a node lands in the tile, pixels that several
// samples of the node hit (five per pixel on average for a full leaf) cost LDS atomics, and the framebuffer sees one global atomic per
// TOUCHED pixel and item instead of one per sample.  Samples outside the tile take the global path.
static constexpr uint32_t ITEM_CHUNKS = 32;
static constexpr uint32_t DTPB = 1024;              // draw workgroup: 16 waves share one tile (128 KB of LDS: one workgroup per CU)

struct DrawItem {
	const SimlodChunk* const* chunks;               // the item's chunk addresses: in the frame's chunk directory, or straight in a row of the builder's chunk table
	uint32_t samples, visibleIdx;
	int32_t  tileX, tileY;                          // origin of the LDS tile, or tileX < 0: no tile
	uint32_t tileWH;                                // its extent, width | height << 16: the node's screen box, at most TILE x TILE
	uint32_t took;                                  // measurement aid (tools/raster_items.py): how long the item's workgroup took over it in the frame's last draw pass, in 10 ns
};
static_assert(sizeof(DrawItem) == 32, "tools/raster_items.py reads draw items as 32-byte records");
static constexpr int TILE = 128;
// Screen bins.  A node close to the camera is larger on screen than any LDS tile and its samples are thinly spread (fewer than one per
// pixel): each of them used to be one device-scope atomic on the framebuffer, and ~25 G scattered 64-bit atomics per second is all the
// memory system does (measured: 2 M such samples = 80 us whatever the number of CUs that issue them — the whole close-up frame took twice
// the time of the bird's-eye frame with fewer samples).  The draw items of such a node do not rasterise: they SORT — every sample becomes a
// 16-byte entry in the queue of the 64 x 64-pixel screen bin it falls into (two passes over the item's samples: count per tile in LDS,
// ONE reservation per item and tile, then store) — and r_overflow gives every screen tile one workgroup that rasterises the tile's queue in
// LDS and merges it into the plane with plain loads and stores (the tile's pixels are nobody else's in that kernel).
static constexpr int TILE_BINNED = -2;                     // DrawItem::tileX of such an item
static constexpr uint32_t BIN_SHIFT = 5, BIN = 1u << BIN_SHIFT;   // a bin = 32 x 32 pixels: 2074 of them at 1920 x 1080 — the terrain towards the horizon of a close-up is a strip of three hundred
static constexpr uint32_t BIN_ITEM_CHUNKS = 8;            // a sorting item: 8000 samples, 8 per thread — kept in registers between the count and the store (16: r_draw<MODE_MIN64> spills)
static constexpr uint32_t OVERFLOW_STRIDE = 10007;         // prime, larger than any bin count
static constexpr uint32_t BIN_POOL_MIN = 65536;            // a buffer that has room for fewer pool entries than this behind its planes draws without bins
static constexpr uint32_t BIN_POOL_ENTRIES = 3000000;      // 48 MB of entries per frame and pass: the buffer stays inside the host's 200 MB at 1920 x 1080 (main_progressive_octree.cpp:555) (what does not fit: device-scope atomics, as before)
static constexpr uint32_t BIN_SEG_CAP = 256;               // segments (item x bin) a bin can list
static constexpr uint32_t BIN_MAX_TILES = 8704;            // (3840 x 2160 pixels: 8228) the per-bin counters of a sorting workgroup live in its LDS; larger frames do not sort
struct BinSeg { uint32_t base, count; };
static constexpr uint32_t OTPB = 1024;                     // r_overflow's workgroup (512: 31 us for the close-up's bins, 256: 55; 1024: 25)
static constexpr int TILE_EXACT_AREA = TILE * TILE / 2;   // HQS colour: tiles up to this area keep two 64-bit words per pixel (exact 32-bit sums)
static constexpr uint32_t MAX_DIR_CHUNKS = 2000000; // chunk directory of a frame: 2 G visible samples

__device__ __forceinline__ uint32_t* counter_at(const RenderArgs& a, int k) { return reinterpret_cast<uint32_t*>(a.mom + R_OFF_COUNTERS + 16 * k); }
enum { C_VISIBLE = 0, C_POINTS = 1, C_VOXELS = 2, C_INNER = 3, C_LEAVES = 4, C_TABLE_LISTS = 5, C_OUTSIDE_TILES = 6 };   // [5]: lists r_visible read through the builder's chunk table; [6]: samples the first draw pass sent down the global-atomic path (outside their item's LDS tile, or no tile)

#ifdef VAR_PROBE
#define R_PROBE_MAX(k) do { if (lane_id() == 0) reinterpret_cast<unsigned long long*>(a.mom + R_OFF_VERTICES + 8000000ull)[(k) * 8192u + blockIdx.x * (TPB / 64u) + threadIdx.x / 64u] = (unsigned long long)wall_clock64(); } while (0)
#define R_PROBE_MIN(k) R_PROBE_MAX(k)
#else
#define R_PROBE_MAX(k) do {} while (0)
#define R_PROBE_MIN(k) do {} while (0)
#endif
// ---- clear (render.cu:1126-1131, 233-241) ---------------------------------------------------------------------
// Part of r_visible's launch: the planes are cleared by ALL its workgroups (a thousand, of which the octree's nodes keep a few dozen busy
// for three dependent memory round trips), the frame's counters by thread 0 of workgroup 0, which then publishes the launch's
// sequence number; a wave reads that word before its first reservation (by then it has long been there).
__device__ __forceinline__ uint32_t* frame_ready_word(const RenderArgs& a) { return reinterpret_cast<uint32_t*>(a.mom + a.offWork) + 15; }
__device__ __forceinline__ void wait_frame_ready(const RenderArgs& a) {
	while (__hip_atomic_load(frame_ready_word(a), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != a.launchSeq) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ void clear_frame(const RenderArgs& a) {
	// 16-byte stores (every plane starts 16-byte aligned): the planes are 8, 4, 8 and 16 bytes per pixel
	const uint32_t stride = gridDim.x * TPB, first = blockIdx.x * TPB + threadIdx.x;
	auto fill = [&](uint64_t offset, uint64_t bytes, uint4 value, uint64_t tailWord, uint32_t tailBytes) {
		uint4* q = reinterpret_cast<uint4*>(a.mom + offset);
		const uint32_t n16 = (uint32_t)(bytes / 16);
		for (uint32_t i = first; i < n16; i += stride) q[i] = value;
		if (first == 0 && bytes % 16 != 0) {                           // an odd pixel count leaves one 4- or 8-byte element
			if (tailBytes == 8) *reinterpret_cast<uint64_t*>(a.mom + offset + (uint64_t)n16 * 16) = tailWord;
			else for (uint64_t b = (uint64_t)n16 * 16; b < bytes; b += 4) *reinterpret_cast<uint32_t*>(a.mom + offset + b) = (uint32_t)tailWord;
		}
	};
	const uint32_t lo = (uint32_t)SIMLOD_CLEAR_PIXEL, hi = (uint32_t)(SIMLOD_CLEAR_PIXEL >> 32);
	fill(R_OFF_FB, (uint64_t)a.numPixels * 8, make_uint4(lo, hi, lo, hi), SIMLOD_CLEAR_PIXEL, 8);
	if (a.useBins) { uint32_t* segCount = reinterpret_cast<uint32_t*>(a.mom + a.offBinSegCount); for (uint32_t i = first; i < a.binTiles; i += stride) segCount[i] = 0u; }
	if (a.hqs) fill(a.offDepth, (uint64_t)a.numPixels * 4, make_uint4(0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u), 0x7f800000u, 4);
}
// The planes of the HQS colour pass — 24 bytes per pixel, two thirds of what a frame clears — are cleared by the DEPTH pass's draw workgroups
// before they take their first item: stores nobody waits for, in a kernel that is bound by LDS atomics.  In r_visible they queued in
// front of the node loads on its critical path: 6 us of that kernel.
__device__ __forceinline__ void clear_colour_planes(const RenderArgs& a) {
	const uint32_t stride = gridDim.x * blockDim.x, first = blockIdx.x * blockDim.x + threadIdx.x;
	uint4* q = reinterpret_cast<uint4*>(a.mom + a.offColor);                     // the packed plane and the {R, G, B, count} plane are neighbours
	const uint64_t bytes = (a.offOverflow - a.offColor) + (uint64_t)a.numPixels * 16;
	for (uint64_t i = first; i < bytes / 16; i += stride) q[i] = make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void clear_counters(const RenderArgs& a) {     // one thread
	*a.frameStart = wall_ns();                                        // render.cu:1100-1102
	for (int k = 0; k < 7; k++) *counter_at(a, k) = 0;
	uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
	for (int k = 0; k < WORK_WORDS; k++) work[k] = 0;
	uint32_t* lines = reinterpret_cast<uint32_t*>(a.mom + R_OFF_LINES);
	lines[0] = 0;                                                      // lines->count = 0, render.cu:1118
	__hip_atomic_store(frame_ready_word(a), a.launchSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- visibility pass 1: screen-space extent + frustum test per node (render.cu:762-901, math.cuh:154-201) --------
__device__ __forceinline__ float dot_row(const simlod_float4& r, float x, float y, float z) {
	float s = r.x * x;
	s = s + r.y * y;
	s = s + r.z * z;
	s = s + r.w * 1.0f;
	return s;
}

// math.cuh:154-201: six planes from the rows of the matrix, normalised; a box is outside when its corner farthest along a plane's
// normal is behind the plane.  The planes are the same for every node: plane i is normalised ONCE per workgroup (24 correctly rounded
// divisions and 6 square roots that every lane used to repeat), into LDS.
__device__ __forceinline__ void frustum_plane(const SimlodMat4& m, int i, float out[4]) {
	const simlod_float4* R = m.rows;
	const float m0 = R[0].x, m1 = R[1].x, m2 = R[2].x, m3 = R[3].x;
	const float m4 = R[0].y, m5 = R[1].y, m6 = R[2].y, m7 = R[3].y;
	const float m8 = R[0].z, m9 = R[1].z, m10 = R[2].z, m11 = R[3].z;
	const float m12 = R[0].w, m13 = R[1].w, m14 = R[2].w, m15 = R[3].w;
	const float P[6][4] = {
		{m3 - m0, m7 - m4, m11 - m8, m15 - m12}, {m3 + m0, m7 + m4, m11 + m8, m15 + m12},
		{m3 + m1, m7 + m5, m11 + m9, m15 + m13}, {m3 - m1, m7 - m5, m11 - m9, m15 - m13},
		{m3 - m2, m7 - m6, m11 - m10, m15 - m14}, {m3 + m2, m7 + m6, m11 + m10, m15 + m14}};
	const float x = P[i][0], y = P[i][1], z = P[i][2], w = P[i][3];
	float d2 = x * x; d2 = d2 + y * y; d2 = d2 + z * z;
	const float len = sqrtf(d2);
	out[0] = x / len; out[1] = y / len; out[2] = z / len; out[3] = w / len;
}

__device__ __forceinline__ bool intersects_frustum(const float (*planes)[4], const float mn[3], const float mx[3]) {
	bool inside = true;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		const float nx = planes[i][0], ny = planes[i][1], nz = planes[i][2], c = planes[i][3];
		const float vx = nx > 0.0f ? mx[0] : mn[0];
		const float vy = ny > 0.0f ? mx[1] : mn[1];
		const float vz = nz > 0.0f ? mx[2] : mn[2];
		float d = nx * vx; d = d + ny * vy; d = d + nz * vz; d = d + c;
		if (d < 0.0f) inside = false;
	}
	return inside;
}

// render.cu:760-861: a node's box is inside when it meets the frustum, large when its screen box spans more than 2 x minNodeSize pixels.
// Pure geometry of (level, X, Y, Z): a node can evaluate its PARENT's `large` — (level - 1, X/2, Y/2, Z/2) — without reading it.
template <bool FRUSTUM>
__device__ __forceinline__ void node_geometry(const RenderArgs& a, const float (*planes)[4], uint32_t level, uint32_t X, uint32_t Y, uint32_t Z, bool& inside, bool& large) {
	const float nodeSize = a.cubeSize / exp2_int(level);
	const float cmin[3] = {a.minx, a.miny, a.minz};
	const uint32_t XYZ[3] = {X, Y, Z};
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
	inside = FRUSTUM ? intersects_frustum(planes, mn, mx) : false;
	const double lim = 2.0 * (double)a.minNodeSize;
	large = (double)dx > lim || (double)dy > lim;                                           // render.cu:860-861
}

// Wave-wide sums by DPP (row shifts inside the 16-lane rows, then the rows' totals broadcast from lanes 15 and 31): six VALU operations.
// Through ds_bpermute (__shfl_up / __shfl_xor) every step is an LDS round trip; r_visible's dozen scans in a row were 2.5 us of its 19.
__device__ __forceinline__ uint32_t wave_inclusive_u32(uint32_t v) {
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);      // row_shr:1
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);      // row_shr:2
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);      // row_shr:4
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);      // row_shr:8
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
	v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
	return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_u32(v), 63); }
__device__ __forceinline__ uint32_t wave_prefix_u32(uint32_t v) { return wave_inclusive_u32(v) - v; }      // exclusive prefix sum over the wave
__device__ __forceinline__ uint32_t wave_prefix_u32(uint32_t v, uint32_t& total) {                          // ... and the wave's total
	const uint32_t incl = wave_inclusive_u32(v);
	total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
	return incl - v;
}

// ---- visibility, draw items and the frame's chunk directory: ONE launch, one lane per node -------------------------------------------------
// The reference flags every node (render.cu:760-861), then lets every LARGE node emit its small visible children, and itself when it is
// a visible leaf (render.cu:746-756, 906-933), then gives one workgroup a whole node and lets it chase the chunk list while it draws
// (render.cu:106-159, 179-207).  Here a node decides about ITSELF: drawn when visible and either small under a large parent — the
// parent's `large` is geometry of (level - 1, X/2, Y/2, Z/2), computed right here with the parent's own arithmetic — or a large leaf.
// No lane waits for another's flags, so flags, emission and draw items are one kernel whose critical path is four memory round trips
// (node fields; one reservation per wave; the chunk-table row; stores) instead of three kernels with twelve.
// Draw items: a lane writes its node's chunk addresses into the frame's directory — copied from the builder's chunk table when that is
// valid, else by walking the list (the only serial pointer chase left in a frame) — and cuts the list into items of <= 64 chunks.
__device__ __forceinline__ void visible_nodes(const RenderArgs& a, const float (&planes)[6][4], const uint32_t numNodes, SimlodNode* staged, const uint32_t readyEarly) {
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	const bool active = i < numNodes;
	// the workgroup's 256 nodes were staged in LDS with coalesced loads (r_visible): a lane reading ITS 152-byte node from memory touched a
	// cache line per lane and field — a thousand line requests per wave, 5 us of the kernel's 19
	SimlodNode* n = staged + (active ? threadIdx.x : 0u);
	SimlodNode* nGlobal = a.nodes + (active ? i : 0u);
	// The builder keeps, per node, the addresses of the first chunks of its list (construct_*.hip, leaf chunk table: a leaf's row lists
	// its point chunks, an inner node's its voxel chunks) and stamps the table with the octree it describes (k_finish).  When that stamp
	// matches THIS octree as it is now, and the node's row starts at the list's head, the row IS the list.
	// (all words of the stamp in flight together: tested one after the other, each waited for the one before — five round trips)
	bool tableValid = false;
	const SimlodChunk* rowHead = nullptr;                    // first entry of this node's row of the table: in flight with the node's fields
	if (a.leafTable != nullptr) {
		const uint32_t magic = *a.leafTableMagic, batch = *a.leafTableBatch, batchNow = a.stats->batchletIndex;
		const uint64_t tableNodes = *a.leafTableNodes, sig = *a.leafTableSig, sigNow = table_signature(a.stats);
		if (a.leafTableSlots <= 64u && a.leafTableRows != 0u) rowHead = leaf_row_get(a.leafTable, a.leafTablePers, active && i < a.leafTableRows ? i : 0u, 0u);
		tableValid = (magic == a.leafTableMagicValue) & (batch == batchNow) & (tableNodes == (uint64_t)a.nodes) & (sig == sigNow);
	}
	const uint32_t level = n->level, X = n->X, Y = n->Y, Z = n->Z;
	const uint32_t counts[2] = {n->numPoints, n->numVoxels};
	const SimlodChunk* heads[2] = {n->points, n->voxelChunks};
	bool leaf = true;
#pragma unroll
	for (int k = 0; k < 8; k++) leaf = leaf && n->children[k] == nullptr;
	bool inside, large, unused, parentLarge = false;
	node_geometry<true>(a, planes, level, X, Y, Z, inside, large);
	const bool visible = inside && (counts[0] > 0u || counts[1] > 0u);
	if (active) { n->visible = visible ? 1 : 0; n->isLarge = large ? 1 : 0; nGlobal->visible = visible ? 1 : 0; nGlobal->isLarge = large ? 1 : 0; }      // (the staged copy goes to the visible list)
	if (active && visible && !large && level > 0u) node_geometry<false>(a, planes, level - 1u, X >> 1, Y >> 1, Z >> 1, unused, parentLarge);   // only who needs it
	const bool emit = active && visible && (large ? leaf : parentLarge);
	R_PROBE_MAX(2);
	if (__ballot(emit) == 0ull) return;

	// one reservation per wave and counter (returning device-scope atomics on one word retire at ~11 ns each, and a lane waits ~2.5 us
	// for each one it depends on): visible-list slots, directory entries, draw items
	const bool draws = emit && a.showPoints;
	// the LDS tile of the node's draw items: its screen box when that fits a tile; else a tile in the MIDDLE of the box (the corners of a
	// cube's screen box are empty, the terrain runs through its middle) — samples that fall outside take the global path.  A node that
	// reaches behind the camera has no box and no tile.
	int tileX = -1, tileY = -1;
	uint32_t tileW = TILE, tileH = TILE;
	bool noTile = draws, sorts = false;
	if (draws && a.useTiles) {
		const float nodeSize = a.cubeSize / exp2_int(level);
		float mnx = 3.0e38f, mny = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f;
		bool front = true;
		for (int k = 0; k < 8; k++) {
			const float x = a.minx + ((float)X + ((k & 4) ? 1.0f : 0.0f)) * nodeSize, y = a.miny + ((float)Y + ((k & 2) ? 1.0f : 0.0f)) * nodeSize;
			const float z = a.minz + ((float)Z + ((k & 1) ? 1.0f : 0.0f)) * nodeSize;
			const float cw = dot_row(a.transform.rows[3], x, y, z);
			if (!(cw > 0.0f)) { front = false; break; }
			// (the hardware's approximate reciprocal: where the tile lies decides how fast a frame is drawn, not what it shows; the correctly
			// rounded divisions of eight corners were a microsecond of this kernel)
			const float rw = __builtin_amdgcn_rcpf(cw);
			const float sx = ((dot_row(a.transform.rows[0], x, y, z) * rw) * 0.5f + 0.5f) * a.width, sy = ((dot_row(a.transform.rows[1], x, y, z) * rw) * 0.5f + 0.5f) * a.height;
			mnx = fminf(mnx, sx); mny = fminf(mny, sy); mxx = fmaxf(mxx, sx); mxy = fmaxf(mxy, sy);
		}
		if (front && mnx > -1.0e6f && mny > -1.0e6f && mnx < 1.0e6f && mny < 1.0e6f) {
			// the part of the box that is on the screen
			const int x0 = max((int)mnx - 1, 0), y0 = max((int)mny - 1, 0);
			const int x1 = min((int)fminf(mxx, 1.0e6f) + a.pointSize + 2, a.W + 1), y1 = min((int)fminf(mxy, 1.0e6f) + a.pointSize + 2, a.H + 1);
			const int bw = max(x1 - x0, 1), bh = max(y1 - y0, 1);
			// the tile takes the box's shape: TILE x TILE words, as wide or as high as the box asks for (a node seen at a grazing angle — the
			// terrain towards the horizon of a close-up — is a strip of 1000 x 40 pixels: under a square tile most of its samples went outside)
			if (bh <= bw) { tileH = (uint32_t)min(bh, TILE); tileW = (uint32_t)min(bw, TILE * TILE / (int)tileH); }
			else { tileW = (uint32_t)min(bw, TILE); tileH = (uint32_t)min(bh, TILE * TILE / (int)tileW); }
			tileX = x0 + (bw - (int)tileW) / 2; tileY = y0 + (bh - (int)tileH) / 2;
			noTile = false;
			sorts = a.binsPossible && (uint32_t)bw * (uint32_t)bh > a.binMinArea;                 // much larger than a tile: its samples are sorted into the screen bins
		} else sorts = a.binsPossible != 0u;                                                      // reaches behind the camera: no box, no tile — sorted
		if (sorts && a.useBins) { tileX = TILE_BINNED; noTile = false; }
	}
	// (launch_render leaves the bins out of a frame — two kernels — when the buffer's previous frame had nothing to sort: this frame tells the next)
	const uint32_t waveSorts = (uint32_t)__popcll(__ballot(sorts));
	R_PROBE_MAX(9);
	// ... and their size: up to ITEM_CHUNKS chunks; an eighth of that for a node without a tile: every sample of such an item is a scattered
	// global atomic, 64 memory transactions per wave instruction — a 32 000-sample item of that kind took ~100 us, the frame's makespan in the
	// close-up preset; short ones spread over the CUs (and have no tile to clear or flush)
	const uint32_t perItem = noTile ? ITEM_CHUNKS / 8u : tileX == TILE_BINNED ? BIN_ITEM_CHUNKS : ITEM_CHUNKS;
	const uint32_t weight = tileX == TILE_BINNED ? 2u : 1u;                                 // a sorting item takes what a tile item of twice its samples takes: it queues with those
	uint32_t numChunks[2], pieces[2];
#pragma unroll
	for (int l = 0; l < 2; l++) {
		const bool have = draws && counts[l] != 0u && heads[l] != nullptr;
		numChunks[l] = have ? (counts[l] + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK : 0u;
		pieces[l] = (numChunks[l] + perItem - 1) / perItem;
	}
	const uint32_t myChunks = numChunks[0] + numChunks[1];
	// A node has one list worth drawing (a leaf its points, an inner node its voxels): that one may come from the builder's chunk table
	const int rowList = numChunks[0] != 0u ? 0 : 1;
	// (a draw item that reads its chunks straight from the row names the row — 256-byte aligned — with bit 0 set and its first slot in bits 1..7: item_chunk)
	const uint8_t* const slots = tableValid && draws && a.leafTableSlots <= 64u && i < a.leafTableRows ? a.leafTable + (uint64_t)i * LEAF_ROW_BYTES : nullptr;
	const uint32_t fromTable = slots != nullptr ? min(numChunks[rowList], a.leafTableSlots) : 0u;
	uint32_t myClass[ITEM_CLASSES] = {0u, 0u, 0u, 0u};                                   // a list's pieces: full ones (class 0), then the rest
#pragma unroll
	for (int l = 0; l < 2; l++) {
		if (pieces[l] == 0u) continue;
		const uint32_t fullClass = item_class(perItem * weight), lastClass = item_class((numChunks[l] - (pieces[l] - 1u) * perItem) * weight);
#pragma unroll
		for (int cl = 0; cl < ITEM_CLASSES; cl++) myClass[cl] += (fullClass == (uint32_t)cl ? pieces[l] - 1u : 0u) + (lastClass == (uint32_t)cl ? 1u : 0u);
	}
	const unsigned long long emitters = __ballot(emit);
	const uint32_t slotsBefore = __builtin_amdgcn_mbcnt_hi((uint32_t)(emitters >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)emitters, 0u)), waveSlots = (uint32_t)__popcll(emitters);
	uint32_t waveChunks;
	const uint32_t chunksBefore = wave_prefix_u32(myChunks, waveChunks);
	uint32_t classBase[ITEM_CLASSES], waveClass[ITEM_CLASSES];
#pragma unroll
	for (int cl = 0; cl < ITEM_CLASSES; cl++) classBase[cl] = wave_prefix_u32(myClass[cl], waveClass[cl]);
	const bool isLeafDraw = emit && counts[0] > 0u, isInnerDraw = emit && counts[0] == 0u && counts[1] > 0u;   // render.cu:748-754
	const uint32_t wLeaves = (uint32_t)__popcll(__ballot(isLeafDraw)), wInner = (uint32_t)__popcll(__ballot(isInnerDraw));
	const uint32_t wPts = wave_sum_u32(isLeafDraw ? counts[0] : 0u), wVox = wave_sum_u32(isInnerDraw ? counts[1] : 0u);
	uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
	uint32_t slot = 0, dirBase = 0, waveBase[ITEM_CLASSES] = {0u, 0u, 0u, 0u};
	R_PROBE_MAX(10);
	if (lane_id() == 0) {
		if (readyEarly != a.launchSeq) { wait_frame_ready(a); R_PROBE_MAX(11); }            // (read while the nodes were on their way: by then thread 0 had long published)
		R_PROBE_MAX(8);
		slot = atomicAdd(counter_at(a, C_VISIBLE), waveSlots);
		if (waveSorts != 0u) atomicAdd(work + 14, waveSorts);
		if (waveChunks != 0u) {
			dirBase = atomicAdd(work + 4, waveChunks);
#pragma unroll
			for (int cl = 0; cl < ITEM_CLASSES; cl++) if (waveClass[cl] != 0u) waveBase[cl] = atomicAdd(work + 8 + cl, waveClass[cl]);
		}
		if (wLeaves) { atomicAdd(counter_at(a, C_LEAVES), wLeaves); atomicAdd(counter_at(a, C_POINTS), wPts); }
		if (wInner) { atomicAdd(counter_at(a, C_INNER), wInner); atomicAdd(counter_at(a, C_VOXELS), wVox); }
	}
	slot = __shfl(slot, 0) + slotsBefore; dirBase = __shfl(dirBase, 0) + chunksBefore;
	if (slot != 0xffffffffu) R_PROBE_MAX(3);
#pragma unroll
	for (int cl = 0; cl < ITEM_CLASSES; cl++) classBase[cl] += __shfl(waveBase[cl], 0);      // this lane's next free slot in class cl
	DrawItem* items = reinterpret_cast<DrawItem*>(a.mom + a.offItems);
	const SimlodChunk** dir = reinterpret_cast<const SimlodChunk**>(a.mom + a.offDir);
	// A list that fits a row of the builder's chunk table (<= 50 chunks: every leaf below its limit, most inner nodes) is not copied at
	// all: its draw item points INTO the row, once the row is seen to start with the list's head (r_draw ends the item at a gap, should
	// a row ever have one).  Measured: copying the rows into the frame's directory — per lane, or by whole waves — was 10 us of this
	// kernel's 28 (the visible nodes are neighbours in the node array: a few waves had all the copying to do).
	const bool rowDirect = fromTable != 0u && numChunks[rowList] <= a.leafTableSlots && rowHead == heads[rowList];
	uint32_t throughTable = 0;
	if (emit) {
		const bool listed = slot < SIMLOD_MAX_VISIBLE_NODES;
		if (!listed) atomicOr(&a.stats->dbg, SIMLOD_ERR_VISIBLE_OVERFLOW);
		else {
			const ulonglong1* src = reinterpret_cast<const ulonglong1*>(n);
			ulonglong1* dst = reinterpret_cast<ulonglong1*>(reinterpret_cast<SimlodNode*>(a.mom + R_OFF_VISIBLE) + slot);
#pragma unroll
			for (int w = 0; w < (int)(sizeof(SimlodNode) / 8); w++) dst[w] = src[w];
		}
		for (int l = 0; l < 2; l++, dirBase += numChunks[l - 1]) {
			if (numChunks[l] == 0u) continue;
			const bool fits = listed && dirBase + numChunks[l] <= MAX_DIR_CHUNKS;
			if (!fits) atomicOr(&a.stats->dbg, SIMLOD_ERR_VISIBLE_OVERFLOW);                // its items are reserved: they stay, empty
			uint32_t k = 0;
			if (fits) {
				const SimlodChunk* chunk = heads[l];
				if (l == rowList && rowDirect) { k = numChunks[l]; chunk = nullptr; throughTable++; }   // nothing to copy
				for (; k < numChunks[l] && chunk != nullptr; k++) { dir[dirBase + k] = chunk; chunk = chunk->next; }
			}
			const uint32_t have = min(counts[l], k * SIMLOD_POINTS_PER_CHUNK);      // a list shorter than its counter says: draw what is there
			for (uint32_t p = 0; p < pieces[l]; p++) {
				const uint32_t firstSample = p * perItem * SIMLOD_POINTS_PER_CHUNK;
				const uint32_t cl = p + 1u < pieces[l] ? item_class(perItem * weight) : item_class((numChunks[l] - p * perItem) * weight);
				uint32_t at = 0;
#pragma unroll
				for (int q = 0; q < ITEM_CLASSES; q++) if (cl == (uint32_t)q) at = classBase[q]++;
				if (at >= a.itemCap) { atomicOr(&a.stats->dbg, SIMLOD_ERR_VISIBLE_OVERFLOW); continue; }
				DrawItem it;
				it.chunks = l == rowList && rowDirect ? reinterpret_cast<const SimlodChunk* const*>((uint64_t)slots | 1ull | (uint64_t)(p * perItem) << 1) : dir + dirBase + p * perItem;
				it.samples = have > firstSample ? min(have - firstSample, perItem * SIMLOD_POINTS_PER_CHUNK) : 0u;
				it.visibleIdx = slot; it.tileX = tileX; it.tileY = tileY; it.tileWH = tileW | (tileH << 16); it.took = 0u;
				items[(uint64_t)cl * a.itemCap + at] = it;
			}
		}
	}
	R_PROBE_MAX(4);
	const uint32_t waveTable = wave_sum_u32(throughTable);
	if (lane_id() == 0 && waveTable != 0u) atomicAdd(counter_at(a, C_TABLE_LISTS), waveTable);
}

__global__ __launch_bounds__(TPB) void r_visible(RenderArgs a) {
	R_PROBE_MIN(0);
	if (blockIdx.x == 0 && threadIdx.x == 0) clear_counters(a);
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	if (numNodes != 0xffffffffu) R_PROBE_MAX(1);
	if (blockIdx.x * TPB < numNodes) {                                                 // whole workgroups: the lanes of a wave reserve together
		__shared__ float planes[6][4];
		__shared__ unsigned long long staged[TPB * sizeof(SimlodNode) / 8];
		static_assert(sizeof(SimlodNode) % 8 == 0, "nodes are staged as 8-byte words");
		const uint32_t words = min((uint32_t)TPB, numNodes - blockIdx.x * TPB) * (uint32_t)(sizeof(SimlodNode) / 8);
		const unsigned long long* src = reinterpret_cast<const unsigned long long*>(a.nodes + (uint64_t)blockIdx.x * TPB);
		constexpr uint32_t PER_THREAD = sizeof(SimlodNode) / 8;                     // all of a thread's loads in flight, then the stores
		unsigned long long held[PER_THREAD];
#pragma unroll
		for (uint32_t q = 0; q < PER_THREAD; q++) { const uint32_t w = q * TPB + threadIdx.x; held[q] = w < words ? src[w] : 0ull; }
#pragma unroll
		for (uint32_t q = 0; q < PER_THREAD; q++) staged[q * TPB + threadIdx.x] = held[q];
		const uint32_t readyEarly = __hip_atomic_load(frame_ready_word(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (threadIdx.x < 6) frustum_plane(a.transformUpdate, (int)threadIdx.x, planes[threadIdx.x]);
		__syncthreads();
		visible_nodes(a, planes, numNodes, reinterpret_cast<SimlodNode*>(staged), readyEarly);
		R_PROBE_MAX(5);
	}
	clear_frame(a);
	R_PROBE_MAX(6); R_PROBE_MIN(7);
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
	unsigned long long* tile;    // LDS, TILE*TILE entries: MIN64 the 64-bit minimum, COLOR the packed sums (DEPTH uses tile32)
	uint32_t* tile32;            // LDS, TILE*TILE entries: DEPTH the minimum of the depth bits
	int tileX, tileY;            // tile origin; tileX < 0: no tile
	int tileW, tileH;            // tile extent
	bool tileExact;              // COLOR: two words per pixel {R | G << 32, B | count << 32} instead of the packed word
	struct HotTable* hot;        // COLOR, packed tile: the pixels of the item that took more than 64 samples (LDS; nullptr in the other passes)
};

// The packed tile word of the colour pass holds 64 samples of a pixel (64 x 255 < 2^14); what comes beyond used to go to the pixel's words in the
// global {R, G, B, count} plane, two device-scope atomics per sample — on ONE address when the pixel is hot, and the memory system retires ~70 M
// same-address atomics a second: on the 500 M-point octree of BASELINE config 4 a handful of items with a few hot pixels each (ridges seen edge-on:
// thousands of samples on a pixel) took 100-300 us where their neighbours took 20, and set the colour pass's length (bird: 220 us against 64 for the
// depth pass; tools/raster_big.py).  Now the 65th sample onwards of a pixel goes into a small LDS table of the item's hot pixels — tile index ->
// exact 32-bit sums — which the flush adds to the global plane with two atomics per hot PIXEL.  No room in the table: the global words, as before.
static constexpr uint32_t HOT_CAP = 512, HOT_EMPTY = 0xffffffffu;
struct HotTable { uint32_t key[HOT_CAP]; unsigned long long rg[HOT_CAP], bc[HOT_CAP]; };
template <int MODE> struct HotStore { __device__ __forceinline__ HotTable* table() { return nullptr; } };
template <> struct HotStore<2> { HotTable t; __device__ __forceinline__ HotTable* table() { return &t; } };      // (MODE_COLOR)
__device__ __forceinline__ void beyond_64(const DrawCtx& c, uint32_t t, uint32_t pixel, unsigned long long r, unsigned long long g, unsigned long long b) {
	if (c.hot != nullptr) {
		uint32_t h = (t * 2654435761u) >> (32 - 9);
#pragma unroll 1
		for (int probe = 0; probe < 8; probe++) {
			uint32_t k = c.hot->key[h];
			if (k == HOT_EMPTY) { k = atomicCAS(&c.hot->key[h], HOT_EMPTY, t); if (k == HOT_EMPTY) k = t; }
			if (k == t) { atomicAdd(&c.hot->rg[h], r | (g << 32)); atomicAdd(&c.hot->bc[h], b | (1ull << 32)); return; }
			h = (h + 1u) & (HOT_CAP - 1u);
		}
	}
	atomicAdd(&c.overflow[2 * pixel + 0], r | (g << 32));
	atomicAdd(&c.overflow[2 * pixel + 1], b | (1ull << 32));
}

template <int MODE>
__device__ __forceinline__ void draw_sample(const DrawCtx& c, const float4 p, const uint32_t overrideColor, const bool useOverride, uint32_t& outside) {
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
		if (c.tileX >= 0) {                                 // LDS-staged accumulation for nodes that are small on screen
			const unsigned tx = (unsigned)(px - c.tileX), ty = (unsigned)(py - c.tileY);
			if (tx < (unsigned)c.tileW && ty < (unsigned)c.tileH) {
				const unsigned t = tx + ty * (unsigned)c.tileW;
				if (MODE == MODE_MIN64) {
					const unsigned long long enc = ((unsigned long long)dbits << 32) | color;
					if (enc < c.tile[t]) atomicMin(&c.tile[t], enc);
				} else if (MODE == MODE_DEPTH) {
					if (dbits < c.tile32[t]) atomicMin(&c.tile32[t], dbits);
				} else if (depth < __uint_as_float(c.depth[pixel]) * 1.01f && c.tileExact) {
					atomicAdd(&c.tile[2 * t + 0], (unsigned long long)(color & 0xffu) | ((unsigned long long)((color >> 8) & 0xffu) << 32));
					atomicAdd(&c.tile[2 * t + 1], (unsigned long long)((color >> 16) & 0xffu) | (1ull << 32));
				} else if (depth < __uint_as_float(c.depth[pixel]) * 1.01f) {
					// the packed sums of the global plane, in LDS: B | G << 14 | R << 28 | count << 42; the 65th sample of a pixel
					// inside one item takes its addend back and goes to the global overflow plane (exact for any count)
					const unsigned long long r = color & 0xffu, g = (color >> 8) & 0xffu, b = (color >> 16) & 0xffu;
					const unsigned long long pk = b | (g << 14) | (r << 28) | (1ull << 42);
					const unsigned long long old = atomicAdd(&c.tile[t], pk);
					if ((old >> 42) >= 64ull) { atomicAdd(&c.tile[t], 0ull - pk); beyond_64(c, t, pixel, r, g, b); }
				}
				continue;
			}
		}
		outside += 1u;
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

// One sample per lane, the whole wave in step (point size 1, tile in use): when every lane that hits the tile hits the SAME pixel —
// the rule in BASELINE config 5, where thousands of samples of a node fall on one pixel — the wave reduces its values with
// cross-lane shuffles and ONE lane issues the LDS atomic (64 same-address LDS atomics serialise).  Otherwise every lane issues its
// own, as draw_sample does.  Same test-before-atomic rules, same values: the tile ends up identical.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
	for (int o = 32; o > 0; o >>= 1) {
		const unsigned long long w = ((unsigned long long)__shfl_xor((uint32_t)(v >> 32), o, 64) << 32) | __shfl_xor((uint32_t)v, o, 64);
		v = w < v ? w : v;
	}
	return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
	for (int o = 32; o > 0; o >>= 1) { const uint32_t w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
	return v;
}

template <int MODE>
__device__ __forceinline__ void draw_wave(const DrawCtx& c, const float4 p, const bool have, const uint32_t overrideColor, const bool useOverride, uint32_t& outside) {
	const float cx = dot_row(c.r0, p.x, p.y, p.z);
	const float cy = dot_row(c.r1, p.x, p.y, p.z);
	const float depth = dot_row(c.r3, p.x, p.y, p.z);
	const float nx = cx / depth, ny = cy / depth;
	const double fx = ((double)nx * 0.5 + 0.5) * (double)c.width;
	const double fy = ((double)ny * 0.5 + 0.5) * (double)c.height;
	const int x = (int)fx, y = (int)fy;
	bool valid = have && (x > 1 && (double)x < c.wlim) && (y > 1 && (double)y < c.hlim);
	if (MODE != MODE_MIN64) valid = valid && depth > 0.0f;
	const uint32_t dbits = __float_as_uint(depth);
	const uint32_t color = useOverride ? overrideColor : __float_as_uint(p.w);
	const int px = min(max(x, 0), c.W), py = min(max(y, 0), c.H);
	const uint32_t pixel = (uint32_t)px + (uint32_t)c.W * (uint32_t)py;
	valid = valid && pixel < c.numPixels;
	const unsigned tx = (unsigned)(px - c.tileX), ty = (unsigned)(py - c.tileY);
	const bool inTile = valid && tx < (unsigned)c.tileW && ty < (unsigned)c.tileH;
	const unsigned t = tx + ty * (unsigned)c.tileW;
	bool accept = true;
	if (MODE == MODE_COLOR) accept = valid && depth < __uint_as_float(c.depth[valid ? pixel : 0u]) * 1.01f;
	const bool mine = inTile && accept;
	const unsigned long long mask = __ballot(mine);
	bool uniform = false;
	unsigned t0 = 0;
	if (mask != 0ull && (MODE != MODE_COLOR || c.tileExact)) {
		t0 = (unsigned)__shfl((int)t, (int)(__ffsll((long long)mask) - 1), 64);
		uniform = __popcll(mask) >= 8 && __ballot(mine && t == t0) == mask;
	}
	if (uniform) {                                   // wave-uniform branch: every lane takes part in the shuffles
		const bool leader = (unsigned)lane_id() == (unsigned)(__ffsll((long long)mask) - 1);
		if (MODE == MODE_MIN64) {
			const unsigned long long v = wave_min_u64(mine ? (((unsigned long long)dbits << 32) | color) : ~0ull);
			if (leader && v < c.tile[t0]) atomicMin(&c.tile[t0], v);
		} else if (MODE == MODE_DEPTH) {
			const uint32_t v = wave_min_u32(mine ? dbits : 0xffffffffu);
			if (leader && v < c.tile32[t0]) atomicMin(&c.tile32[t0], v);
		} else {
			uint32_t rg = mine ? ((color & 0xffu) | (((color >> 8) & 0xffu) << 16)) : 0u, bc = mine ? (((color >> 16) & 0xffu) | (1u << 16)) : 0u;
			for (int o = 32; o > 0; o >>= 1) { rg += __shfl_xor(rg, o, 64); bc += __shfl_xor(bc, o, 64); }   // 64 x 255 < 2^16: no carry between the halves
			if (leader) {
				atomicAdd(&c.tile[2 * t0 + 0], (unsigned long long)(rg & 0xffffu) | ((unsigned long long)(rg >> 16) << 32));
				atomicAdd(&c.tile[2 * t0 + 1], (unsigned long long)(bc & 0xffffu) | ((unsigned long long)(bc >> 16) << 32));
			}
		}
	} else if (mine) {
		if (MODE == MODE_MIN64) {
			const unsigned long long enc = ((unsigned long long)dbits << 32) | color;
			if (enc < c.tile[t]) atomicMin(&c.tile[t], enc);
		} else if (MODE == MODE_DEPTH) {
			if (dbits < c.tile32[t]) atomicMin(&c.tile32[t], dbits);
		} else if (c.tileExact) {
			atomicAdd(&c.tile[2 * t + 0], (unsigned long long)(color & 0xffu) | ((unsigned long long)((color >> 8) & 0xffu) << 32));
			atomicAdd(&c.tile[2 * t + 1], (unsigned long long)((color >> 16) & 0xffu) | (1ull << 32));
		} else {
			const unsigned long long r = color & 0xffu, g = (color >> 8) & 0xffu, b = (color >> 16) & 0xffu;
			const unsigned long long pk = b | (g << 14) | (r << 28) | (1ull << 42);
			const unsigned long long old = atomicAdd(&c.tile[t], pk);
			if ((old >> 42) >= 64ull) { atomicAdd(&c.tile[t], 0ull - pk); beyond_64(c, t, pixel, r, g, b); }
		}
	}
	if (valid && !inTile) {                          // outside the tile: the global path of draw_sample
		outside += 1u;
		if (MODE == MODE_MIN64) {
			const unsigned long long enc = ((unsigned long long)dbits << 32) | color;
			if (enc < c.fb[pixel]) atomicMin(reinterpret_cast<unsigned long long*>(&c.fb[pixel]), enc);
		} else if (MODE == MODE_DEPTH) {
			if (dbits < c.depth[pixel]) atomicMin(&c.depth[pixel], dbits);
		} else if (accept) {
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

// Point size 1, tile in use — the common case — DU samples per lane in three stages, so that nothing in the loop waits for anything:
//   1. project all DU samples (pure arithmetic; HQS colour: the DU depth-buffer loads go out together),
//   2. tiles of a few pixels only (BASELINE config 5: thousands of samples of a node on one pixel): draw_wave, which merges a wave's
//      samples with shuffles when they all hit the same pixel,
//   3. otherwise every lane issues its LDS atomics straight away — no read-compare first: an LDS atomic that does not change the word
//      costs what the read would, and returns nothing to wait for.  min and add commute: the tile ends up identical.
template <int MODE, uint32_t DU>
__device__ __forceinline__ void draw_staged(const DrawCtx& c, const float4 (&p)[DU], const bool (&have)[DU], const uint32_t overrideColor, const bool useOverride, uint32_t& outside) {
	uint32_t pixel[DU], t[DU], dbits[DU], color[DU];
	float depth[DU];
	bool valid[DU], inTile[DU], accept[DU];
#pragma unroll
	for (uint32_t u = 0; u < DU; u++) {
		const float cx = dot_row(c.r0, p[u].x, p[u].y, p[u].z);
		const float cy = dot_row(c.r1, p[u].x, p[u].y, p[u].z);
		depth[u] = dot_row(c.r3, p[u].x, p[u].y, p[u].z);
		const float nx = cx / depth[u], ny = cy / depth[u];
		const double fx = ((double)nx * 0.5 + 0.5) * (double)c.width;
		const double fy = ((double)ny * 0.5 + 0.5) * (double)c.height;
		const int x = (int)fx, y = (int)fy;
		valid[u] = have[u] && (x > 1 && (double)x < c.wlim) && (y > 1 && (double)y < c.hlim);
		if (MODE != MODE_MIN64) valid[u] = valid[u] && depth[u] > 0.0f;
		dbits[u] = __float_as_uint(depth[u]);
		color[u] = useOverride ? overrideColor : __float_as_uint(p[u].w);
		const int px = min(max(x, 0), c.W), py = min(max(y, 0), c.H);
		pixel[u] = (uint32_t)px + (uint32_t)c.W * (uint32_t)py;
		valid[u] = valid[u] && pixel[u] < c.numPixels;
		const unsigned tx = (unsigned)(px - c.tileX), ty = (unsigned)(py - c.tileY);
		inTile[u] = valid[u] && tx < (unsigned)c.tileW && ty < (unsigned)c.tileH;
		t[u] = tx + ty * (unsigned)c.tileW;
		accept[u] = true;
	}
	uint32_t ref[DU];
	if (MODE == MODE_COLOR) {
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) ref[u] = c.depth[valid[u] ? pixel[u] : 0u];
	}
	if (MODE == MODE_COLOR) {
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) accept[u] = valid[u] && depth[u] < __uint_as_float(ref[u]) * 1.01f;          // render.cu:485-493
	}
#pragma unroll
	for (uint32_t u = 0; u < DU; u++) {
		if (inTile[u] && accept[u]) {
			if (MODE == MODE_MIN64) atomicMin(&c.tile[t[u]], ((unsigned long long)dbits[u] << 32) | color[u]);
			else if (MODE == MODE_DEPTH) atomicMin(&c.tile32[t[u]], dbits[u]);
			else if (c.tileExact) {
				atomicAdd(&c.tile[2 * t[u] + 0], (unsigned long long)(color[u] & 0xffu) | ((unsigned long long)((color[u] >> 8) & 0xffu) << 32));
				atomicAdd(&c.tile[2 * t[u] + 1], (unsigned long long)((color[u] >> 16) & 0xffu) | (1ull << 32));
			} else {
				const unsigned long long r = color[u] & 0xffu, g = (color[u] >> 8) & 0xffu, b = (color[u] >> 16) & 0xffu;
				const unsigned long long pk = b | (g << 14) | (r << 28) | (1ull << 42);
				const unsigned long long old = atomicAdd(&c.tile[t[u]], pk);
				if ((old >> 42) >= 64ull) { atomicAdd(&c.tile[t[u]], 0ull - pk); beyond_64(c, t[u], pixel[u], r, g, b); }
			}
		}
	}
#pragma unroll
	for (uint32_t u = 0; u < DU; u++) {
		if (valid[u] && !inTile[u]) {                  // outside the tile: the global path of draw_sample
			outside += 1u;
			// No read-compare first (render.cu:95-100, 304-308 test before they exchange): a node close to the camera is larger than any tile, a
			// sixth of the close-up frame's samples come this way, and a wave that waits for a framebuffer read per sample draws at half the speed
			// (measured: r_draw 157 us with the reads — in flight together or not —, against 69 us for a frame whose samples stay in their
			// tiles).  The atomic returns nothing to wait for; min is idempotent: the framebuffer ends up identical.
			if (MODE == MODE_MIN64) atomicMin(reinterpret_cast<unsigned long long*>(&c.fb[pixel[u]]), ((unsigned long long)dbits[u] << 32) | color[u]);
			else if (MODE == MODE_DEPTH) atomicMin(&c.depth[pixel[u]], dbits[u]);
			else if (accept[u]) {
				const unsigned long long r = color[u] & 0xffu, g = (color[u] >> 8) & 0xffu, b = (color[u] >> 16) & 0xffu;
				const unsigned long long pk = b | (g << 14) | (r << 28) | (1ull << 42);
				const unsigned long long old = atomicAdd(&c.color[pixel[u]], pk);
				if ((old >> 42) >= 64ull) {
					atomicAdd(&c.color[pixel[u]], 0ull - pk);
					atomicAdd(&c.overflow[2 * pixel[u] + 0], r | (g << 32));
					atomicAdd(&c.overflow[2 * pixel[u] + 1], b | (1ull << 32));
				}
			}
		}
	}
}

template <int MODE>
__device__ __forceinline__ void draw_item(const DrawCtx& c, const SimlodChunk* const* dir, uint32_t count, uint32_t overrideColor, bool useOverride, uint32_t& outside) {
	// render.cu:106-159: chunk i holds samples [1000 i, 1000 i + 1000); the chunk addresses come from the frame's directory (staged in LDS).
	// Four samples per thread are loaded before the first is drawn: the loads overlap instead of queueing behind the atomics.
	constexpr uint32_t DU = 4;
	// (an item without a tile — a node that reaches behind the camera — is staged like the others: all its samples take the global path,
	// DU of a lane in flight together; sample by sample such an item took 60-150 us and was the frame's makespan in the close-up preset)
	const bool wave = c.pointSize == 1;
	const bool merge = c.tileX >= 0 && c.tileW * c.tileH <= 64;     // a node a few pixels across: most lanes of a wave hit the same pixel
	for (uint32_t base = 0; base < count; base += DTPB * DU) {          // uniform trip count: the whole wave stays in step
		float4 p[DU];
		bool have[DU];
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			const uint32_t s = base + u * DTPB + threadIdx.x;
			have[u] = s < count;
			p[u] = have[u] ? reinterpret_cast<const float4*>(dir[s / SIMLOD_POINTS_PER_CHUNK]->points)[s % SIMLOD_POINTS_PER_CHUNK] : make_float4(0, 0, 0, 0);
		}
		if (wave && merge) {
#pragma unroll
			for (uint32_t u = 0; u < DU; u++) draw_wave<MODE>(c, p[u], have[u], overrideColor, useOverride, outside);
		} else if (wave) {
			draw_staged<MODE, DU>(c, p, have, overrideColor, useOverride, outside);
		} else {
#pragma unroll
			for (uint32_t u = 0; u < DU; u++) if (have[u]) draw_sample<MODE>(c, p[u], overrideColor, useOverride, outside);
		}
	}
}

// One item of a node that is much larger than a tile (DrawItem::tileX == TILE_BINNED): its samples are sorted into the screen bins.  Every
// thread projects its <= 8 samples ONCE and keeps {bin, pixel inside it, value} in registers, counting per bin in LDS; then the workgroup
// scans the counters, takes ALL its entries from the pool with ONE atomic (every item of the frame reserves on that word: one atomic per
// item and bin on it took 11 ns each, 200 us a frame) and lists one segment per bin with samples; then every sample writes its entry.
// Value: depth | colour (plain frames), depth (HQS depth pass), colour of an ACCEPTED sample (HQS colour pass, render.cu:485-493).  Pool
// exhausted: the item's samples take the device-scope atomics, as before; a bin's list full: that bin's.
template <int MODE>
__device__ __forceinline__ void bin_item(const DrawCtx& c, const RenderArgs& a, uint32_t* lds, const SimlodChunk* const* dir, uint32_t count, uint32_t overrideColor, bool useOverride,
                                         uint32_t& outside) {
	// (opaque to the optimiser: what depends on the thread only — chunk and offset of each of its 16 samples, the addresses of its bins' words —
	// was hoisted out of r_draw's item loop and occupied 40 registers for the whole kernel: every draw path spilled)
	uint32_t tid = threadIdx.x;
	asm volatile("" : "+v"(tid));
	uint32_t* cnt = lds;                           // [binTiles]: samples of this item per bin, then the cursor inside the segment
	uint32_t* base = lds + BIN_MAX_TILES;          // [binTiles]: first pool entry of the item's segment in that bin, or NONE
	uint32_t* scratch = lds + 2 * BIN_MAX_TILES;   // [16] sums of the waves, [16] the item's first pool entry
	constexpr uint32_t NONE = 0xffffffffu, DU = 4, SLOTS = (BIN_ITEM_CHUNKS * SIMLOD_POINTS_PER_CHUNK + DTPB - 1) / DTPB;
	static_assert(SLOTS % DU == 0, "a thread's samples come in batches of DU loads");
	uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
	uint32_t* segCount = reinterpret_cast<uint32_t*>(a.mom + a.offBinSegCount);
	BinSeg* segs = reinterpret_cast<BinSeg*>(a.mom + a.offBinSegs);
	uint4* pool = reinterpret_cast<uint4*>(a.mom + a.offBinPool);
	for (uint32_t t = tid; t < a.binTiles; t += DTPB) cnt[t] = 0u;
	__syncthreads();
	uint32_t key[SLOTS], lo[SLOTS], hi[SLOTS];     // bin << 10 | pixel inside the bin, or NONE; the value's halves
#pragma unroll
	for (uint32_t r = 0; r < SLOTS / DU; r++) {
		float4 p[DU];
		bool have[DU];
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			const uint32_t sIdx = (r * DU + u) * DTPB + tid;
			have[u] = sIdx < count;
			p[u] = have[u] ? reinterpret_cast<const float4*>(dir[sIdx / SIMLOD_POINTS_PER_CHUNK]->points)[sIdx % SIMLOD_POINTS_PER_CHUNK] : make_float4(0, 0, 0, 0);
		}
		uint32_t pixel[DU];
		float depth[DU];
		bool valid[DU];
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			const uint32_t k = r * DU + u;
			const float cx = dot_row(c.r0, p[u].x, p[u].y, p[u].z);
			const float cy = dot_row(c.r1, p[u].x, p[u].y, p[u].z);
			depth[u] = dot_row(c.r3, p[u].x, p[u].y, p[u].z);
			const float nx = cx / depth[u], ny = cy / depth[u];
			const double fx = ((double)nx * 0.5 + 0.5) * (double)c.width;
			const double fy = ((double)ny * 0.5 + 0.5) * (double)c.height;
			const int x = (int)fx, y = (int)fy;
			valid[u] = have[u] && (x > 1 && (double)x < c.wlim) && (y > 1 && (double)y < c.hlim);
			if (MODE != MODE_MIN64) valid[u] = valid[u] && depth[u] > 0.0f;
			const int px = min(max(x, 0), c.W), py = min(max(y, 0), c.H);
			pixel[u] = (uint32_t)px + (uint32_t)c.W * (uint32_t)py;
			valid[u] = valid[u] && pixel[u] < c.numPixels;
			const uint32_t bin = ((uint32_t)py >> BIN_SHIFT) * a.binTilesX + ((uint32_t)px >> BIN_SHIFT);
			key[k] = (bin << (2 * BIN_SHIFT)) | (((uint32_t)py & (BIN - 1u)) << BIN_SHIFT) | ((uint32_t)px & (BIN - 1u));
			const uint32_t color = useOverride ? overrideColor : __float_as_uint(p[u].w);
			lo[k] = MODE == MODE_DEPTH ? __float_as_uint(depth[u]) : color;
			hi[k] = MODE == MODE_MIN64 ? __float_as_uint(depth[u]) : 0u;
		}
		if (MODE == MODE_COLOR) {                                           // render.cu:485-493
			uint32_t ref[DU];
#pragma unroll
			for (uint32_t u = 0; u < DU; u++) ref[u] = c.depth[valid[u] ? pixel[u] : 0u];
#pragma unroll
			for (uint32_t u = 0; u < DU; u++) valid[u] = valid[u] && depth[u] < __uint_as_float(ref[u]) * 1.01f;
		}
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			const uint32_t k = r * DU + u;
			if (valid[u]) atomicAdd(&cnt[key[k] >> (2 * BIN_SHIFT)], 1u); else key[k] = NONE;
		}

	}
	__syncthreads();
	// exclusive scan of the counters over the workgroup -> every bin's offset inside the item's reservation
	uint32_t carry = 0;
	for (uint32_t t0 = 0; t0 < a.binTiles; t0 += DTPB) {               // uniform
		const uint32_t t = t0 + tid;
		const uint32_t n = t < a.binTiles ? cnt[t] : 0u;
		uint32_t waveTotal;
		const uint32_t before = wave_prefix_u32(n, waveTotal);
		if (lane_id() == 0) scratch[tid / 64u] = waveTotal;
		__syncthreads();
		uint32_t waveBase = 0, roundTotal = 0;
		for (uint32_t w = 0; w < DTPB / 64u; w++) { const uint32_t v = scratch[w]; waveBase += w < tid / 64u ? v : 0u; roundTotal += v; }
		if (t < a.binTiles) base[t] = carry + waveBase + before;
		carry += roundTotal;
		__syncthreads();
	}
	// the item's entries and, per bin, a place in the bin's list: all reservations in flight together (a list place taken in vain — pool
	// exhausted — holds an empty segment)
	constexpr uint32_t ROUNDS = (BIN_MAX_TILES + DTPB - 1) / DTPB;
	uint32_t seg[ROUNDS];
#pragma unroll
	for (uint32_t q = 0; q < ROUNDS; q++) {
		const uint32_t t = q * DTPB + tid;
		seg[q] = t < a.binTiles && cnt[t] != 0u ? atomicAdd(&segCount[t], 1u) : NONE;
	}
	if (tid == 0) {
		uint32_t at = NONE;
		if (carry != 0u) { at = atomicAdd(work + 12, carry); if ((unsigned long long)at + carry > a.binPoolCap) at = NONE; }
		scratch[16] = at;
	}
	__syncthreads();
	const uint32_t itemBase = scratch[16];
	uint32_t binned = 0;
#pragma unroll
	for (uint32_t q = 0; q < ROUNDS; q++) {
		const uint32_t t = q * DTPB + tid;
		if (t >= a.binTiles) continue;
		const uint32_t n = cnt[t];
		uint32_t b = NONE;
		if (seg[q] < BIN_SEG_CAP) {
			if (itemBase != NONE) { b = itemBase + base[t]; binned += n; }
			segs[(uint64_t)t * BIN_SEG_CAP + seg[q]] = BinSeg{b != NONE ? b : 0u, b != NONE ? n : 0u};
		}
		base[t] = b; cnt[t] = 0u;
	}
	if (MODE != MODE_COLOR) { binned = wave_sum_u32(binned); if (lane_id() == 0 && binned != 0u) atomicAdd(work + 13, binned); }
	__syncthreads();
#pragma unroll
	for (uint32_t k = 0; k < SLOTS; k++) {
		if (key[k] == NONE) continue;
		const uint32_t bin = key[k] >> (2 * BIN_SHIFT), local = key[k] & (BIN * BIN - 1u);
		const uint32_t b = base[bin];
		if (b != NONE) { pool[b + atomicAdd(&cnt[bin], 1u)] = make_uint4(lo[k], hi[k], local, 0u); continue; }
		outside += 1u;                                                     // no room in the bins: the global path of draw_sample
		const uint32_t px = ((bin % a.binTilesX) << BIN_SHIFT) | (local & (BIN - 1u)), py = ((bin / a.binTilesX) << BIN_SHIFT) | (local >> BIN_SHIFT);
		const uint32_t pixel = px + (uint32_t)c.W * py;
		if (MODE == MODE_MIN64) atomicMin(reinterpret_cast<unsigned long long*>(&c.fb[pixel]), ((unsigned long long)hi[k] << 32) | lo[k]);
		else if (MODE == MODE_DEPTH) atomicMin(&c.depth[pixel], lo[k]);
		else {
			const unsigned long long r = lo[k] & 0xffu, g = (lo[k] >> 8) & 0xffu, bl = (lo[k] >> 16) & 0xffu;
			const unsigned long long pk = bl | (g << 14) | (r << 28) | (1ull << 42);
			const unsigned long long old = atomicAdd(&c.color[pixel], pk);
			if ((old >> 42) >= 64ull) {
				atomicAdd(&c.color[pixel], 0ull - pk);
				atomicAdd(&c.overflow[2 * pixel + 0], r | (g << 32));
				atomicAdd(&c.overflow[2 * pixel + 1], bl | (1ull << 32));
			}
		}
	}
}

template <int MODE>
__device__ __forceinline__ void tile_clear(const DrawCtx& c) {
	const int words = c.tileW * c.tileH * (MODE == MODE_COLOR && c.tileExact ? 2 : 1);
	for (int t = threadIdx.x; t < words; t += DTPB) {
		if (MODE == MODE_DEPTH) c.tile32[t] = 0xffffffffu; else c.tile[t] = MODE == MODE_COLOR ? 0ull : ~0ull;
	}
	if (MODE == MODE_COLOR && c.hot != nullptr && !c.tileExact)
		for (uint32_t h = threadIdx.x; h < HOT_CAP; h += DTPB) { c.hot->key[h] = HOT_EMPTY; c.hot->rg[h] = 0ull; c.hot->bc[h] = 0ull; }
}

// One global atomic per TOUCHED pixel of the tile.
template <int MODE>
__device__ __forceinline__ void tile_flush(const DrawCtx& c) {
	for (int t = threadIdx.x; t < c.tileW * c.tileH; t += DTPB) {
		const int px = c.tileX + (t % c.tileW), py = c.tileY + (t / c.tileW);
		if (px > c.W || py > c.H) continue;
		const uint32_t pixel = (uint32_t)px + (uint32_t)c.W * (uint32_t)py;
		if (pixel >= c.numPixels) continue;
		if (MODE == MODE_MIN64) {
			// no read-compare first: a thread flushes up to 16 pixels, and 16 dependent framebuffer reads were most of an item's time;
			// the atomic returns nothing to wait for, and a node's pixels are mostly its own, so few of them would have been spared
			const unsigned long long v = c.tile[t];
			if (v != ~0ull) atomicMin(reinterpret_cast<unsigned long long*>(&c.fb[pixel]), v);
		} else if (MODE == MODE_DEPTH) {
			const uint32_t v = c.tile32[t];
			if (v != 0xffffffffu) atomicMin(&c.depth[pixel], v);
		} else if (c.tileExact) {
			const unsigned long long rg = c.tile[2 * t], bc = c.tile[2 * t + 1];
			if ((bc >> 32) != 0ull) { atomicAdd(&c.overflow[2 * pixel + 0], rg); atomicAdd(&c.overflow[2 * pixel + 1], bc); }
		} else {
			const unsigned long long pk = c.tile[t];
			if (pk != 0ull) {                                 // exact: resolve adds the packed plane and the {R, G, B, count} plane
				atomicAdd(&c.overflow[2 * pixel + 0], ((pk >> 28) & 0x3fffull) | (((pk >> 14) & 0x3fffull) << 32));
				atomicAdd(&c.overflow[2 * pixel + 1], (pk & 0x3fffull) | ((pk >> 42) << 32));
			}
		}
	}
	if (MODE == MODE_COLOR && c.hot != nullptr && !c.tileExact) {      // the item's hot pixels: what they took beyond their 64th sample
		for (uint32_t h = threadIdx.x; h < HOT_CAP; h += DTPB) {
			const uint32_t t = c.hot->key[h];
			if (t == HOT_EMPTY) continue;
			const uint32_t pixel = (uint32_t)(c.tileX + (int)(t % (uint32_t)c.tileW)) + (uint32_t)c.W * (uint32_t)(c.tileY + (int)(t / (uint32_t)c.tileW));
			atomicAdd(&c.overflow[2 * pixel + 0], c.hot->rg[h]);
			atomicAdd(&c.overflow[2 * pixel + 1], c.hot->bc[h]);
		}
	}
}

template <int MODE>
__global__ __launch_bounds__(DTPB) void r_draw(RenderArgs a) {
	if (MODE == MODE_DEPTH) clear_colour_planes(a);
	if (!a.showPoints) return;
	__shared__ uint32_t sh_idx;
	__shared__ const SimlodChunk* sh_dir[ITEM_CHUNKS];
	constexpr uint32_t BIN_WORDS = 2 * BIN_MAX_TILES + 32;                                  // bin_item's counters, in the tile's place
	__shared__ unsigned long long sh_tile[MODE == MODE_DEPTH ? (TILE * TILE > BIN_WORDS ? TILE * TILE : BIN_WORDS) / 2 : TILE * TILE];
	__shared__ HotStore<MODE> sh_hot;
	DrawCtx c;
	c.hot = sh_hot.table();
	c.tile = sh_tile; c.tile32 = reinterpret_cast<uint32_t*>(sh_tile); c.tileX = -1; c.tileY = -1; c.tileW = TILE; c.tileH = TILE; c.tileExact = false;
	c.r0 = a.transform.rows[0]; c.r1 = a.transform.rows[1]; c.r3 = a.transform.rows[3];
	c.width = a.width; c.height = a.height;
	c.wlim = (double)a.width - 2.0; c.hlim = (double)a.height - 2.0;
	c.W = a.W; c.H = a.H; c.pointSize = a.pointSize; c.numPixels = a.numPixels;
	c.fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	c.depth = reinterpret_cast<uint32_t*>(a.mom + a.offDepth);
	c.color = reinterpret_cast<unsigned long long*>(a.mom + a.offColor);
	c.overflow = reinterpret_cast<unsigned long long*>(a.mom + a.offOverflow);
	uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
	uint32_t* cursor = work + MODE;
	if (MODE != MODE_COLOR && blockIdx.x == 0 && threadIdx.x == 0 && a.binFeedback != nullptr) *a.binFeedback = work[14];
	uint32_t classEnd[ITEM_CLASSES];                                                       // position q of the cursor: class c while q < classEnd[c]
	for (int cl = 0; cl < ITEM_CLASSES; cl++) classEnd[cl] = (cl > 0 ? classEnd[cl - 1] : 0u) + min(work[8 + cl], a.itemCap);
	const uint32_t numItems = classEnd[ITEM_CLASSES - 1];
	const DrawItem* items = reinterpret_cast<const DrawItem*>(a.mom + a.offItems);
	const SimlodNode* visible = reinterpret_cast<const SimlodNode*>(a.mom + R_OFF_VISIBLE);
	// Workgroup-level queue of draw items.  The first item of a workgroup is its own index, the following ones come from a shared
	// cursor that starts behind the statically assigned range.
	uint32_t idx = blockIdx.x;
	uint32_t outside = 0;                                                                   // samples of this thread that went down the global-atomic path
	while (idx < numItems) {
		uint32_t cl = 0;
		while (idx >= classEnd[cl]) cl++;
		const uint64_t itemAt = (uint64_t)cl * a.itemCap + (idx - (cl > 0u ? classEnd[cl - 1u] : 0u));
		const DrawItem it = items[itemAt];
		const uint64_t itemStart = SIMLOD_MEASURE != 0 && threadIdx.x == 0 ? wall_clock64() : 0ull;
		uint32_t overrideColor = 0; bool useOverride = false;
		if (MODE != MODE_DEPTH && (a.colorByNode || a.colorByLOD)) {
			const SimlodNode* node = visible + it.visibleIdx;
			overrideColor = a.colorByNode ? node_color(node) : lod_color((int)node->level);
			useOverride = true;
		}
		const bool binned = it.tileX == TILE_BINNED;
		c.tileX = binned ? -1 : it.tileX; c.tileY = it.tileY; c.tileW = (int)(it.tileWH & 0xffffu); c.tileH = (int)(it.tileWH >> 16);
		if (it.tileX < 0) { c.tileW = 0; c.tileH = 0; }                                       // no tile: nothing is inside it
		c.tileExact = c.tileW * c.tileH <= TILE_EXACT_AREA;
		bool gap = false;
		if (threadIdx.x < ITEM_CHUNKS && threadIdx.x * SIMLOD_POINTS_PER_CHUNK < it.samples) {
			const uint64_t where = (uint64_t)it.chunks;           // the frame's chunk directory, or (bit 0) a row of the builder's packed chunk table from slot (bits 1..7) on
			const SimlodChunk* ch = (where & 1ull) != 0ull ? leaf_row_get(reinterpret_cast<const uint8_t*>(where & ~255ull), a.leafTablePers, 0, (uint32_t)((where >> 1) & 127ull) + threadIdx.x)
			                                               : it.chunks[threadIdx.x];
			sh_dir[threadIdx.x] = ch;
			gap = ch == nullptr;
		}
		if (it.tileX >= 0) tile_clear<MODE>(c);
		uint32_t samples = it.samples;
		if (__syncthreads_or(gap ? 1 : 0)) {             // a table row with a gap (never seen; rows are complete while their stamp is valid): draw what precedes it
			uint32_t whole = 0;
			while (whole < ITEM_CHUNKS && whole * SIMLOD_POINTS_PER_CHUNK < it.samples && sh_dir[whole] != nullptr) whole++;
			samples = min(samples, whole * SIMLOD_POINTS_PER_CHUNK);
		}
		if (binned) bin_item<MODE>(c, a, reinterpret_cast<uint32_t*>(sh_tile), sh_dir, samples, overrideColor, useOverride, outside);
		else draw_item<MODE>(c, sh_dir, samples, overrideColor, useOverride, outside);
		if (it.tileX >= 0) { __syncthreads(); tile_flush<MODE>(c); }
		__syncthreads();
		if (SIMLOD_MEASURE != 0 && threadIdx.x == 0) const_cast<DrawItem*>(items)[itemAt].took = (uint32_t)(wall_clock64() - itemStart);
		if (threadIdx.x == 0) sh_idx = gridDim.x + atomicAdd(cursor, 1u);
		__syncthreads();
		idx = sh_idx;
	}
	if (MODE != MODE_COLOR) {                                                               // (the colour pass draws the same samples again)
		outside = wave_sum_u32(outside);
		if (lane_id() == 0 && outside != 0u) atomicAdd(counter_at(a, C_OUTSIDE_TILES), outside);
	}
}

// ---- overflow: the screen bins (what the sorting items of r_draw queued) into the planes -----------------------------------------------
// One workgroup per 32 x 32-pixel bin: the bin's pixels in LDS, every wave takes segments of the bin's list — consecutive 16-byte entries, four
// of a lane in flight —, LDS atomics; then the touched pixels are merged into the plane with PLAIN loads and stores: r_draw has ended, and
// in this kernel a pixel belongs to one workgroup.  The colour pass keeps exact 32-bit sums (two 64-bit words per pixel).  Leaves the bins
// empty for the next pass.
template <int MODE>
__global__ __launch_bounds__(OTPB) void r_overflow(RenderArgs a) {
	uint32_t* work = reinterpret_cast<uint32_t*>(a.mom + a.offWork);
	uint32_t* segCount = reinterpret_cast<uint32_t*>(a.mom + a.offBinSegCount);
	// (workgroup -> bin by a stride that is coprime to every bin count: the full bins of a frame are neighbours — a band of the screen — and
	// in launch order they would all start late, behind a thousand empty ones)
	const uint32_t T = (uint32_t)(((uint64_t)blockIdx.x * OVERFLOW_STRIDE) % a.binTiles);
	const uint32_t numSegs = min(segCount[T], BIN_SEG_CAP);
	const uint64_t started = SIMLOD_MEASURE != 0 && threadIdx.x == 0u ? wall_clock64() : 0ull;
	if (T == 0u && threadIdx.x == 0u) work[12] = 0u;                 // (nobody appends in this kernel; this pass's entries stay where they are until the next pass overwrites them)
	if (numSegs == 0u) return;
	// (one workgroup per bin of the SCREEN, most of which find nothing.  Round 5 measured the alternative VERDICT r4 asked for — r_draw lists the bins
	// it opens, a fixed grid of 1 024 workgroups visits the listed ones —: 25.4 us against 20.8 per pass on the close-up, same box, same run: the
	// list costs two more dependent loads in front of every bin's segments, the empty workgroups cost less than that.  Persistent workgroups that
	// loop over bins: 30 us against 24, round 4.)
	constexpr uint32_t PIXELS = BIN * BIN, DU = 4;
	__shared__ unsigned long long sh_tile[MODE == MODE_DEPTH ? PIXELS / 2 : MODE == MODE_COLOR ? 2 * PIXELS : PIXELS];
	__shared__ BinSeg sh_segs[BIN_SEG_CAP];
	uint32_t* tile32 = reinterpret_cast<uint32_t*>(sh_tile);
	const uint4* pool = reinterpret_cast<const uint4*>(a.mom + a.offBinPool);
	__shared__ uint32_t sh_first[BIN_SEG_CAP + 1], sh_waves[OTPB / 64];      // first chunk of every segment; [numSegs] = chunks in all
	static_assert(BIN_SEG_CAP <= OTPB, "one thread per segment");
	uint64_t* fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	uint32_t* depth = reinterpret_cast<uint32_t*>(a.mom + a.offDepth);
	unsigned long long* overflow = reinterpret_cast<unsigned long long*>(a.mom + a.offOverflow);
	const BinSeg* segs = reinterpret_cast<const BinSeg*>(a.mom + a.offBinSegs) + (uint64_t)T * BIN_SEG_CAP;
	{
		BinSeg seg = BinSeg{0u, 0u};
		if (threadIdx.x < numSegs) { seg = segs[threadIdx.x]; sh_segs[threadIdx.x] = seg; }
		const uint32_t n = (seg.count + 64u * DU - 1u) / (64u * DU);
		uint32_t waveTotal;
		const uint32_t before = wave_prefix_u32(n, waveTotal);
		if (lane_id() == 0) sh_waves[threadIdx.x / 64u] = waveTotal;
		__syncthreads();
		uint32_t waveBase = 0;
		for (uint32_t w = 0; w < threadIdx.x / 64u; w++) waveBase += sh_waves[w];
		if (threadIdx.x < numSegs) sh_first[threadIdx.x] = waveBase + before;
		if (threadIdx.x + 1u == numSegs) sh_first[numSegs] = waveBase + before + n;
	}
	for (uint32_t t = threadIdx.x; t < PIXELS * (MODE == MODE_COLOR ? 2u : 1u); t += OTPB) {
		if (MODE == MODE_DEPTH) tile32[t] = 0xffffffffu; else sh_tile[t] = MODE == MODE_COLOR ? 0ull : ~0ull;
	}
	__syncthreads();
	const uint32_t numChunks = sh_first[numSegs];
	// the list as chunks of 64 x DU entries: chunk c belongs to the segment whose first chunk is the last one <= c; waves take chunks in turn
	// (a segment per wave left most waves idle: a bin lists 10-20 segments of 50 to 5000 entries)
	// Two chunks of a wave in flight: the entries of the next one are requested before the current one's go into the tile.  Measured on the
	// close-up's full bins (28 000 entries): the loads alone 7.5 us, the LDS work alone 7 us, one after the other 18 — a wave does not
	// overlap them by itself; with two chunks in flight 14 (with 8 entries per lane and chunk the second set of registers halved the
	// workgroups per CU and gained nothing: 4 entries).
	auto fetch = [&](const uint32_t c, uint4 (&e)[DU], bool (&have)[DU]) {
		uint32_t sgLo = 0, sgHi = numSegs;                       // sh_first[sgLo] <= c < sh_first[sgHi]
		while (sgHi - sgLo > 1u) { const uint32_t mid = (sgLo + sgHi) / 2u; if (sh_first[mid] <= c) sgLo = mid; else sgHi = mid; }
		const BinSeg seg = sh_segs[sgLo];
		const uint32_t first = (c - sh_first[sgLo]) * 64u * DU;
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			const uint32_t i = first + u * 64u + lane_id();
			have[u] = c < numChunks && i < seg.count;
			e[u] = have[u] ? pool[seg.base + i] : make_uint4(0, 0, 0, 0);
		}
	};
	auto apply = [&](const uint4 (&e)[DU], const bool (&have)[DU]) {
#pragma unroll
		for (uint32_t u = 0; u < DU; u++) {
			if (!have[u]) continue;
			const uint32_t local = e[u].z & (PIXELS - 1u);
			// (a full bin has 20-30 entries per pixel: most entries are not their pixel's minimum, and a read that says so is cheaper than
			// the atomic it saves: 16.5 -> 13.5 us for 28 000 entries)
			if (MODE == MODE_MIN64) { const unsigned long long v = ((unsigned long long)e[u].y << 32) | e[u].x; if (v < sh_tile[local]) atomicMin(&sh_tile[local], v); }
			else if (MODE == MODE_DEPTH) { if (e[u].x < tile32[local]) atomicMin(&tile32[local], e[u].x); }
			else {
				atomicAdd(&sh_tile[2 * local + 0], (unsigned long long)(e[u].x & 0xffu) | ((unsigned long long)((e[u].x >> 8) & 0xffu) << 32));
				atomicAdd(&sh_tile[2 * local + 1], (unsigned long long)((e[u].x >> 16) & 0xffu) | (1ull << 32));
			}
		}
	};
	{
		constexpr uint32_t WAVES = OTPB / 64u;
		uint4 eA[DU], eB[DU];
		bool haveA[DU], haveB[DU];
		uint32_t c = threadIdx.x / 64u;
		if (c < numChunks) fetch(c, eA, haveA);
		while (c < numChunks) {                                  // wave-uniform
			fetch(c + WAVES, eB, haveB);                        // (past the end: nothing is loaded)
			apply(eA, haveA);
			c += WAVES;
			if (c >= numChunks) break;
			fetch(c + WAVES, eA, haveA);
			apply(eB, haveB);
			c += WAVES;
		}
	}
	__syncthreads();
	const int x0 = (int)(T % a.binTilesX) * (int)BIN, y0 = (int)(T / a.binTilesX) * (int)BIN;
	for (uint32_t t = threadIdx.x; t < PIXELS; t += OTPB) {
		const int px = x0 + (int)(t & (BIN - 1u)), py = y0 + (int)(t >> BIN_SHIFT);
		if (px >= a.W || py >= a.H) continue;                 // (a valid sample's pixel is inside (1, W - 2) x (1, H - 2): never a pixel of another bin's row)
		const uint32_t pixel = (uint32_t)px + (uint32_t)a.W * (uint32_t)py;
		if (MODE == MODE_MIN64) {
			const unsigned long long v = sh_tile[t];
			if (v != ~0ull && v < fb[pixel]) fb[pixel] = v;
		} else if (MODE == MODE_DEPTH) {
			const uint32_t v = tile32[t];
			if (v != 0xffffffffu && v < depth[pixel]) depth[pixel] = v;
		} else {
			const unsigned long long rg = sh_tile[2 * t], bc = sh_tile[2 * t + 1];
			if ((bc >> 32) != 0ull) { overflow[2 * pixel + 0] += rg; overflow[2 * pixel + 1] += bc; }      // exact: resolve adds the packed plane and the {R, G, B, count} plane
		}
	}
	if (threadIdx.x == 0u) {
		segCount[T] = 0u;
		if (SIMLOD_MEASURE != 0) {                                     // tools/raster_bins.py: {entries, segments << 20 | 10 ns}
			uint32_t entries = 0;
			for (uint32_t k = 0; k < numSegs; k++) entries += sh_segs[k].count;
			BinSeg* stat = reinterpret_cast<BinSeg*>(segCount + (a.binTiles + 3u) / 4u * 4u);
			stat[T] = BinSeg{entries, (numSegs << 20) | min((uint32_t)(wall_clock64() - started), 0xfffffu)};
		}
	}
}

// ---- debug lines (Uniforms.showBoundingBox): node boxes + view frustum -----------------------------------------------
// render.cu:637-688 (four coincident boxes per visible node), :1197-1223 (frustum), rasterization.cuh:5-47 (drawLine,
// drawBoundingBox), :90-183 (rasterizeLines), math.cuh:22-152.  Off by default in the reference (main.cpp:125).
struct LinePlane { float nx, ny, nz, c; };

__device__ __forceinline__ LinePlane make_plane(float x, float y, float z, float w) {
	float d2 = x * x; d2 = d2 + y * y; d2 = d2 + z * z;
	const float len = sqrtf(d2);
	LinePlane p = {x / len, y / len, z / len, w / len};
	return p;
}

__device__ void frustum_planes(const SimlodMat4& m, LinePlane P[6]) {
	const simlod_float4* R = m.rows;
	const float m0 = R[0].x, m1 = R[1].x, m2 = R[2].x, m3 = R[3].x, m4 = R[0].y, m5 = R[1].y, m6 = R[2].y, m7 = R[3].y;
	const float m8 = R[0].z, m9 = R[1].z, m10 = R[2].z, m11 = R[3].z, m12 = R[0].w, m13 = R[1].w, m14 = R[2].w, m15 = R[3].w;
	P[0] = make_plane(m3 - m0, m7 - m4, m11 - m8, m15 - m12);
	P[1] = make_plane(m3 + m0, m7 + m4, m11 + m8, m15 + m12);
	P[2] = make_plane(m3 + m1, m7 + m5, m11 + m9, m15 + m13);
	P[3] = make_plane(m3 - m1, m7 - m5, m11 - m9, m15 - m13);
	P[4] = make_plane(m3 - m2, m7 - m6, m11 - m10, m15 - m14);
	P[5] = make_plane(m3 + m2, m7 + m6, m11 + m10, m15 + m14);
}

__device__ __forceinline__ float plane_dist(const LinePlane& p, float x, float y, float z) {
	float d = p.nx * x; d = d + p.ny * y; d = d + p.nz * z; d = d + p.c;
	return d;
}

__device__ bool frustum_contains(const LinePlane P[6], float x, float y, float z) {
	bool in = true;
	for (int i = 0; i < 6; i++) if (plane_dist(P[i], x, y, z) < 0.0f) in = false;
	return in;
}

__device__ float dist_to_plane(float ox, float oy, float oz, float dx, float dy, float dz, const LinePlane& p) {
	const float INF = __uint_as_float(0x7f800000u);
	float denom = p.nx * dx; denom = denom + p.ny * dy; denom = denom + p.nz * dz;
	if (denom < 0.0f) return INF;
	if (denom == 0.0f) return plane_dist(p, ox, oy, oz) == 0.0f ? 0.0f : INF;
	float num = ox * p.nx; num = num + oy * p.ny; num = num + oz * p.nz; num = num + p.c;
	const float t = -num / denom;
	return t >= 0.0f ? t : INF;
}

__device__ void frustum_intersect_ray(const LinePlane P[6], float ox, float oy, float oz, float dx, float dy, float dz, float out[3]) {
	const float INF = __uint_as_float(0x7f800000u);
	float farthest = -INF;
	for (int i = 0; i < 6; i++) {
		const float d = dist_to_plane(ox, oy, oz, dx, dy, dz, P[i]);
		if (d > 0.0f && d != INF) farthest = fmaxf(farthest, d);
	}
	out[0] = ox + dx * farthest; out[1] = oy + dy * farthest; out[2] = oz + dz * farthest;
}

__device__ __forceinline__ void put_line(float4* v, uint32_t at, float ax, float ay, float az, float bx, float by, float bz, uint32_t color) {
	v[at] = make_float4(ax, ay, az, __uint_as_float(color));
	v[at + 1] = make_float4(bx, by, bz, __uint_as_float(color));
}

static constexpr uint32_t LINE_VERTEX_CAP = 1000000u;     // render.cu:1119

__global__ __launch_bounds__(TPB) void r_lines_emit(RenderArgs a, SimlodMat4 inv) {
	const uint32_t numVisible = min(*counter_at(a, C_VISIBLE), SIMLOD_MAX_VISIBLE_NODES);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	uint32_t* count = reinterpret_cast<uint32_t*>(a.mom + R_OFF_LINES);
	float4* v = reinterpret_cast<float4*>(a.mom + R_OFF_VERTICES);
	if (i == 0) {      // the view frustum as seen by the frozen visibility transform, render.cu:1197-1223
		const float fend = 0.99995f;
		const float C[8][2][3] = {{{1, 1, -1}, {1, 1, fend}}, {{1, -1, -1}, {1, -1, fend}}, {{-1, 1, -1}, {-1, 1, fend}}, {{-1, -1, -1}, {-1, -1, fend}},
		                          {{-1, -1, fend}, {1, -1, fend}}, {{-1, 1, fend}, {1, 1, fend}}, {{-1, -1, fend}, {-1, 1, fend}}, {{1, -1, fend}, {1, 1, fend}}};
		const uint32_t at = atomicAdd(count, 16u);
		for (int l = 0; l < 8; l++) {
			float p[2][3];
			for (int k = 0; k < 2; k++) {
				const float qx = dot_row(inv.rows[0], C[l][k][0], C[l][k][1], C[l][k][2]), qy = dot_row(inv.rows[1], C[l][k][0], C[l][k][1], C[l][k][2]);
				const float qz = dot_row(inv.rows[2], C[l][k][0], C[l][k][1], C[l][k][2]), qw = dot_row(inv.rows[3], C[l][k][0], C[l][k][1], C[l][k][2]);
				p[k][0] = qx / qw; p[k][1] = qy / qw; p[k][2] = qz / qw;
			}
			if (at + 2 * l + 2 <= LINE_VERTEX_CAP) put_line(v, at + 2 * l, p[0][0], p[0][1], p[0][2], p[1][0], p[1][1], p[1][2], 0x000000ffu);
		}
	}
	if (i >= numVisible) return;
	const SimlodNode* n = reinterpret_cast<const SimlodNode*>(a.mom + R_OFF_VISIBLE) + i;
	if (n->numPoints == 0 && n->numVoxels == 0) return;
	const float scale = a.cubeSize / exp2_int(n->level);
	const float pos[3] = {a.minx + ((float)n->X + 0.5f) * scale, a.miny + ((float)n->Y + 0.5f) * scale, a.minz + ((float)n->Z + 0.5f) * scale};
	float mn[3], mx[3];
	for (int k = 0; k < 3; k++) { mn[k] = pos[k] - scale / 2.0f; mx[k] = pos[k] + scale / 2.0f; }
	const uint32_t at0 = atomicAdd(count, 96u);                  // 4 boxes x 12 edges x 2 vertices
	if (at0 + 96u > LINE_VERTEX_CAP) { atomicOr(&a.stats->dbg, SIMLOD_ERR_VISIBLE_OVERFLOW); return; }
	const int E[12][6] = {{0,0,0, 1,0,0}, {1,0,0, 1,1,0}, {1,1,0, 0,1,0}, {0,1,0, 0,0,0}, {0,0,1, 1,0,1}, {1,0,1, 1,1,1}, {1,1,1, 0,1,1}, {0,1,1, 0,0,1},
	                      {1,0,0, 1,0,1}, {1,1,0, 1,1,1}, {0,1,0, 0,1,1}, {0,0,0, 0,0,1}};
	for (int r = 0; r < 4; r++)
		for (int e = 0; e < 12; e++)
			put_line(v, at0 + (uint32_t)(r * 12 + e) * 2u, E[e][0] ? mx[0] : mn[0], E[e][1] ? mx[1] : mn[1], E[e][2] ? mx[2] : mn[2],
			         E[e][3] ? mx[0] : mn[0], E[e][4] ? mx[1] : mn[1], E[e][5] ? mx[2] : mn[2], 0x0000ff00u);
}

__device__ __forceinline__ int to_int_like_the_host(double v) {   // the oracle's cvttsd2si behaviour: out of range -> INT_MIN
	return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
}

__global__ __launch_bounds__(TPB) void r_lines_raster(RenderArgs a) {
	const uint32_t count = min(*reinterpret_cast<const uint32_t*>(a.mom + R_OFF_LINES), LINE_VERTEX_CAP);
	const uint32_t l = blockIdx.x * TPB + threadIdx.x;
	if (2 * l + 1 >= count) return;
	const float4* v = reinterpret_cast<const float4*>(a.mom + R_OFF_VERTICES);
	uint64_t* fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	float4 s = v[2 * l], e = v[2 * l + 1];
	LinePlane P[6];
	frustum_planes(a.transform, P);
	float dx = e.x - s.x, dy = e.y - s.y, dz = e.z - s.z;
	float d2 = dx * dx; d2 = d2 + dy * dy; d2 = d2 + dz * dz;
	const float inv = 1.0f / sqrtf(d2);                    // normalize(): v * rsqrtf(dot(v, v))
	dx = dx * inv; dy = dy * inv; dz = dz * inv;
	if (!frustum_contains(P, s.x, s.y, s.z)) { float I[3]; frustum_intersect_ray(P, s.x, s.y, s.z, dx, dy, dz, I); s.x = I[0]; s.y = I[1]; s.z = I[2]; }
	if (!frustum_contains(P, e.x, e.y, e.z)) { float I[3]; frustum_intersect_ray(P, e.x, e.y, e.z, dx * -1.0f, dy * -1.0f, dz * -1.0f, I); e.x = I[0]; e.y = I[1]; e.z = I[2]; }
	float ax = dot_row(a.transform.rows[0], s.x, s.y, s.z), ay = dot_row(a.transform.rows[1], s.x, s.y, s.z);
	const float aw = dot_row(a.transform.rows[3], s.x, s.y, s.z);
	float bx = dot_row(a.transform.rows[0], e.x, e.y, e.z), by = dot_row(a.transform.rows[1], e.x, e.y, e.z);
	const float bw = dot_row(a.transform.rows[3], e.x, e.y, e.z);
	ax = ax / aw; ay = ay / aw; bx = bx / bw; by = by / bw;
	const float sx0 = (ax * 0.5f + 0.5f) * (float)a.W, sy0 = (ay * 0.5f + 0.5f) * (float)a.H;
	const float sx1 = (bx * 0.5f + 0.5f) * (float)a.W, sy1 = (by * 0.5f + 0.5f) * (float)a.H;
	const float ddx = sx1 - sx0, ddy = sy1 - sy0;
	float st2 = ddx * ddx; st2 = st2 + ddy * ddy; st2 = st2 + 0.0f;
	float steps = sqrtf(st2);
	steps = fmaxf(0.0f, fminf(steps, 400.0f));
	const float stepSize = (float)(1.0 / (double)steps);
	const uint32_t color = __float_as_uint(s.w);
#pragma unroll 1
	for (float t = 0.0f; (double)t <= 1.0; t += stepSize) {
		const float tbx = t * bx, tby = t * by, tbw = t * bw;
		const float nx = (float)((1.0 - (double)t) * (double)ax + (double)tbx);
		const float ny = (float)((1.0 - (double)t) * (double)ay + (double)tby);
		const float depth = (float)((1.0 - (double)t) * (double)aw + (double)tbw);
		if ((double)nx < -1.0 || (double)nx > 1.0) continue;
		if ((double)ny < -1.0 || (double)ny > 1.0) continue;
		int x = to_int_like_the_host(((double)nx * 0.5 + 0.5) * (double)a.W);
		int y = to_int_like_the_host(((double)ny * 0.5 + 0.5) * (double)a.H);
		x = min(max(x, 0), a.W - 1); y = min(max(y, 0), a.H - 1);
		const unsigned long long enc = ((unsigned long long)__float_as_uint(depth) << 32) | color;
		atomicMin(reinterpret_cast<unsigned long long*>(&fb[x + a.W * y]), enc);   // rasterization.cuh:175-178
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

// ---- multi-GPU HQS: fold the packed per-pixel sums into the {R, G, B, count} plane, so that ranks can all-reduce(SUM) it ---------
__global__ __launch_bounds__(TPB) void r_unpack(RenderArgs a) {
	unsigned long long* packed = reinterpret_cast<unsigned long long*>(a.mom + a.offColor);
	uint4* sums = reinterpret_cast<uint4*>(a.mom + a.offOverflow);
	const uint32_t stride = gridDim.x * TPB;
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < a.numPixels; i += stride) {
		const unsigned long long pk = packed[i];
		if (pk == 0ull) continue;
		uint4 s = sums[i];
		s.x += (uint32_t)((pk >> 28) & 0x3fffu); s.y += (uint32_t)((pk >> 14) & 0x3fffu); s.z += (uint32_t)(pk & 0x3fffu); s.w += (uint32_t)(pk >> 42);
		sums[i] = s;
		packed[i] = 0ull;
	}
}

// ---- output: Stats (render.cu:1244-1252), EDL (:1255-1325, every full 16x16 tile), surface write (:1334-1343) ---------
// RESOLVE: the HQS resolve of r_resolve in the same pass (whole frames without debug lines: nothing but the resolve writes the
// framebuffer between the clear and this kernel).  A pixel resolves itself (and stores the word: the pre-EDL framebuffer stays what the
// reference's is); of its four neighbours EDL wants the depth only, and that is the depth plane's word whenever it is a normal number
// (its nearest sample passes its own 1 % test, so the pixel has a colour) or +inf (nothing landed: the cleared framebuffer word has the
// same high half); a denormal depth — whose own sample fails d < d * 1.01f — takes the long way.
template <bool RESOLVE>
__device__ __forceinline__ uint32_t resolved_depth_bits(const RenderArgs& a, const uint64_t* fb, int idx) {
	if (!RESOLVE) return (uint32_t)(fb[idx] >> 32);
	const uint32_t d = reinterpret_cast<const uint32_t*>(a.mom + a.offDepth)[idx];
	if (d >= 0x00800000u) return d == 0x7f800000u ? (uint32_t)(fb[idx] >> 32) : d;
	const unsigned long long pk = reinterpret_cast<const unsigned long long*>(a.mom + a.offColor)[idx];
	const uint32_t count = reinterpret_cast<const uint4*>(a.mom + a.offOverflow)[idx].w + (uint32_t)(pk >> 42);
	return count != 0u ? d : (uint32_t)(fb[idx] >> 32);
}

// One workgroup per 64 x 16-pixel tile, four pixels per thread (rows ty, ty + 4, ty + 8, ty + 12: everything they read is requested
// before the first value is used).  EDL wants log2 of the depth of a pixel and of its four neighbours: every pixel's logarithm is taken
// ONCE, by its own thread, and passed on through LDS (plus a rim of 160 pixels around the tile).  The reference's neighbours are INDEX
// neighbours (i +- 1, i +- W, clamped to the frame: render.cu:1296-1300): the left neighbour of a row's first pixel is the last pixel of
// the row before; the rim is addressed the same way, and a pixel in the frame's last column or row that is not in its tile's last
// column or row reads that one neighbour directly.
static constexpr int OUT_TW = 64, OUT_TH = 16, OUT_PX = 4;
static_assert(OUT_TW * OUT_TH == (int)TPB * OUT_PX && 2 * OUT_TW + 2 * OUT_TH <= (int)TPB, "four pixels per thread; one rim pixel per thread");
template <bool RESOLVE>
__global__ __launch_bounds__(TPB) void r_output(RenderArgs a) {
	uint64_t* fb = reinterpret_cast<uint64_t*>(a.mom + R_OFF_FB);
	if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
		SimlodStats* s = a.stats;
		s->numVisibleNodes = min(*counter_at(a, C_VISIBLE), SIMLOD_MAX_VISIBLE_NODES);
		s->numVisibleInner = *counter_at(a, C_INNER);
		s->numVisibleLeaves = *counter_at(a, C_LEAVES);
		s->numVisiblePoints = *counter_at(a, C_POINTS);
		s->numVisibleVoxels = *counter_at(a, C_VOXELS);
		s->frameID = a.frameCounter;
	}
	if (a.colorbuffer == nullptr) return;
	constexpr int PITCH = OUT_TW + 2, ROWS_PER_STEP = OUT_TH / OUT_PX;
	__shared__ float sh_log[(OUT_TH + 2) * PITCH];
	const int edlW = (a.W / 16) * 16, edlH = (a.H / 16) * 16;
	const int last = (int)a.numPixels - 1;
	const int tx = (int)threadIdx.x % OUT_TW, ty0 = (int)threadIdx.x / OUT_TW;
	const int x0 = (int)blockIdx.x * OUT_TW, y0 = (int)blockIdx.y * OUT_TH;
	const int x = x0 + tx;
	auto log_at = [&](int idx) -> float {                    // log2 of the depth EDL sees at pixel idx (clamped like the reference's index)
		idx = idx < 0 ? 0 : (idx > last ? last : idx);
		return __log2f(__uint_as_float(resolved_depth_bits<RESOLVE>(a, fb, idx)));
	};
	bool inside[OUT_PX];
	uint64_t enc[OUT_PX];
	unsigned long long pk[OUT_PX];
	uint4 sums[OUT_PX];
	uint32_t own[OUT_PX];
#pragma unroll
	for (int q = 0; q < OUT_PX; q++) {
		const int y = y0 + ty0 + q * ROWS_PER_STEP;
		inside[q] = x < a.W && y < a.H;
		const int i = inside[q] ? y * a.W + x : 0;
		enc[q] = fb[i];
		if (RESOLVE) { pk[q] = reinterpret_cast<const unsigned long long*>(a.mom + a.offColor)[i]; sums[q] = reinterpret_cast<const uint4*>(a.mom + a.offOverflow)[i]; own[q] = reinterpret_cast<const uint32_t*>(a.mom + a.offDepth)[i]; }
	}
	float rim = 0.0f;
	int rimSlot = -1;
	if ((int)threadIdx.x < 2 * OUT_TW + 2 * OUT_TH) {       // the rim: the index neighbours of the tile's border pixels
		const int h = (int)threadIdx.x;
		int cx, cy, off, slot;                                // the border pixel (tile coordinates), its neighbour's index offset, the rim's LDS slot
		if (h < OUT_TW) { cx = h; cy = 0; off = -a.W; slot = cx + 1; }
		else if (h < 2 * OUT_TW) { cx = h - OUT_TW; cy = OUT_TH - 1; off = a.W; slot = (OUT_TH + 1) * PITCH + cx + 1; }
		else if (h < 2 * OUT_TW + OUT_TH) { cx = 0; cy = h - 2 * OUT_TW; off = -1; slot = (cy + 1) * PITCH; }
		else { cx = OUT_TW - 1; cy = h - 2 * OUT_TW - OUT_TH; off = 1; slot = (cy + 1) * PITCH + OUT_TW + 1; }
		if (x0 + cx < edlW && y0 + cy < edlH) { rim = log_at((y0 + cy) * a.W + x0 + cx + off); rimSlot = slot; }      // (only pixels EDL shades ask)
	}
#pragma unroll
	for (int q = 0; q < OUT_PX; q++) {
		if (!inside[q]) continue;
		if (RESOLVE) {                                                                          // as r_resolve
			uint4 s = sums[q];
			s.x += (uint32_t)((pk[q] >> 28) & 0x3fffu); s.y += (uint32_t)((pk[q] >> 14) & 0x3fffu); s.z += (uint32_t)(pk[q] & 0x3fffu); s.w += (uint32_t)(pk[q] >> 42);
			if (s.w != 0u) {
				const uint32_t rgba = ((s.x / s.w) & 0xffu) | (((s.y / s.w) & 0xffu) << 8) | (((s.z / s.w) & 0xffu) << 16) | (255u << 24);
				enc[q] = ((uint64_t)own[q] << 32) | rgba;
				fb[(y0 + ty0 + q * ROWS_PER_STEP) * a.W + x] = enc[q];
			}
		}
		sh_log[(ty0 + q * ROWS_PER_STEP + 1) * PITCH + tx + 1] = __log2f(__uint_as_float((uint32_t)(enc[q] >> 32)));
	}
	if (rimSlot >= 0) sh_log[rimSlot] = rim;
	__syncthreads();
#pragma unroll
	for (int q = 0; q < OUT_PX; q++) {
		if (!inside[q]) continue;
		const int ty = ty0 + q * ROWS_PER_STEP, y = y0 + ty, i = y * a.W + x;
		uint32_t color = (uint32_t)enc[q];
		if (x < edlW && y < edlH) {
			const float lp = sh_log[(ty + 1) * PITCH + tx + 1];
			// the four neighbours int(1.5 * sin/cos(k * 3.1415 / 2)) of render.cu:1296-1300: (0,+1), (+1,0), (0,-1), (-1,0)
			float ln[4];
			ln[0] = (y == a.H - 1 && ty != OUT_TH - 1) ? log_at(i + a.W) : sh_log[(ty + 2) * PITCH + tx + 1];
			ln[1] = (x == a.W - 1 && tx != OUT_TW - 1) ? log_at(i + 1) : sh_log[(ty + 1) * PITCH + tx + 2];
			ln[2] = sh_log[ty * PITCH + tx + 1];
			ln[3] = sh_log[(ty + 1) * PITCH + tx];
			float sum = 0.0f;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const float d = lp - ln[k];
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
                                               uint32_t* batchSizes, uint32_t frameCounter, uint32_t* feedback, uint32_t resetSeq) {
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
	// what the next kernel_construct launches size themselves by (simlod_hip.cpp launch_plan; page-locked): nothing ingested, nothing uploaded, as of this reset
	if (feedback != nullptr) { feedback[0] = 0u; feedback[1] = 0u; feedback[2] = 1u; __hip_atomic_store(feedback + 3, resetSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
static constexpr uint32_t MAX_DRAW_ITEMS = 150000;   // per size class: 100 000 visible nodes, one list each, + slices of long lists
static inline uint64_t align16(uint64_t v) { return (v + 15) / 16 * 16; }
static inline uint32_t bin_tiles_x(uint32_t width) { return (width >> BIN_SHIFT) + 1u; }      // pixel columns 0..W (render.cu:91-92 clamps to W, not W - 1)
static inline uint32_t bin_tiles(uint32_t width, uint32_t height) {
	const uint64_t n = (uint64_t)bin_tiles_x(width) * ((height >> BIN_SHIFT) + 1u);
	return n <= BIN_MAX_TILES ? (uint32_t)n : 0u;
}
// per bin: its segment list, its segment counter, {entries, time} of its latest r_overflow (tools/raster_bins.py)
static inline uint64_t bin_tables_bytes(uint64_t tiles) { return tiles * BIN_SEG_CAP * sizeof(BinSeg) + align16(tiles * 4) + tiles * sizeof(BinSeg); }
static inline uint64_t bin_bytes(uint32_t width, uint32_t height) {                   // the tables, then the pool
	const uint64_t tiles = bin_tiles(width, height);
	return tiles == 0 ? 0 : (uint64_t)BIN_POOL_ENTRIES * 16 + bin_tables_bytes(tiles);
}

uint64_t render_buffer_bytes(uint32_t width, uint32_t height) {
	const uint64_t px = (uint64_t)width * height;
	return R_OFF_FB + align16(px * 8) + 256 + (uint64_t)MAX_DRAW_ITEMS * ITEM_CLASSES * sizeof(DrawItem) + align16(px * 4) + align16(px * 8) + px * 16 + (uint64_t)MAX_DIR_CHUNKS * 8 +
	       bin_bytes(width, height) + 256;
}

int launch_reset(Context& ctx, const SimlodUniforms* u, uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                 uint32_t* batchSizes, hipStream_t stream) {
	forget_leaf_table(ctx, nodes);
	ctx.sideTablesStale.store(true);
	uint32_t* words = nullptr; uint32_t seq = 0;
	forget_launch_history(ctx, stats, numBatchesUploaded, &words, &seq);
	SIMLOD_LAUNCH(k_reset, dim3(64), dim3(TPB), stream, pers, nodes, stats, numBatchesUploaded, batchSizes, (uint32_t)u->frameCounter, words, seq);
	return (int)hipGetLastError();
}

static void render_plane_offsets(uint64_t numPixels, uint64_t& offWork, uint64_t& offItems, uint64_t& offDepth, uint64_t& offColor, uint64_t& offOverflow, uint64_t* offDir = nullptr) {
	offWork = R_OFF_FB + align16(numPixels * 8);
	offItems = offWork + 256;
	offDepth = offItems + (uint64_t)MAX_DRAW_ITEMS * ITEM_CLASSES * sizeof(DrawItem);
	offColor = offDepth + align16(numPixels * 4);
	offOverflow = offColor + align16(numPixels * 8);
	if (offDir != nullptr) *offDir = offOverflow + numPixels * 16;
}

uint64_t render_depth_plane_offset(uint32_t width, uint32_t height) {
	uint64_t w, i, d, c, o; render_plane_offsets((uint64_t)width * height, w, i, d, c, o); return d;
}
uint64_t render_sum_planes_offset(uint32_t width, uint32_t height) {
	uint64_t w, i, d, c, o; render_plane_offsets((uint64_t)width * height, w, i, d, c, o); return o;
}

// parts: bit 0 = clear, visibility, draw items and the first pass (plain: the only pass, and the debug lines; HQS: depth)
//        bit 1 = HQS colour pass, sums unpacked        bit 2 = HQS resolve, then the debug lines        bit 3 = Stats, EDL, RGBA8 output
int launch_render(Context& ctx, uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                  uint64_t* frameStart, hipStream_t stream, uint32_t parts) {
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
	a.nodeCapacity = ctx.nodeCapacity.load();
	a.frameCounter = (uint32_t)u->frameCounter;
	a.showPoints = u->showPoints; a.colorByNode = u->colorByNode; a.colorByLOD = u->colorByLOD; a.hqs = u->useHighQualityShading;
	render_plane_offsets(a.numPixels, a.offWork, a.offItems, a.offDepth, a.offColor, a.offOverflow, &a.offDir);
	LeafTableRef lt;
	if (ctx.tune(KNOB_RASTER_LEAF_TABLE, 1) && find_leaf_table(ctx, nodes, lt)) {
		a.leafTable = lt.table; a.leafTablePers = lt.pers; a.leafTableMagic = lt.magic; a.leafTableBatch = lt.batch; a.leafTableNodes = lt.tableNodes; a.leafTableSig = lt.sig;
		a.leafTableMagicValue = lt.magicValue; a.leafTableSlots = lt.slots; a.leafTableRows = lt.rows;
	}
	a.itemCap = MAX_DRAW_ITEMS;
	a.useTiles = (uint32_t)ctx.tune(KNOB_RASTER_LDS_TILES, 1);
	a.binTiles = bin_tiles(u->width, u->height); a.binTilesX = bin_tiles_x(u->width); a.binPoolCap = (uint32_t)min(max(ctx.tune(KNOB_DEBUG_BIN_POOL, (int)BIN_POOL_ENTRIES), 0), (int)BIN_POOL_ENTRIES);   // (tests: a pool that runs out)
	// SIMLOD_RASTER_SCREEN_BINS: 0 = off, else the screen-box area from which a node sorts, in units of 1024 pixels (default: two tiles)
	const int binKnob = ctx.tune(KNOB_RASTER_SCREEN_BINS, 32);
	a.binsPossible = a.useTiles != 0u && a.binTiles != 0u && binKnob > 0 && a.pointSize == 1 ? 1u : 0u;
	a.binMinArea = (uint32_t)max(binKnob, 0) * 1024u;
	// ... and a frame sorts when the buffer's previous frame had nodes to sort (a frame that has none pays 4-5 us for two idle kernels; one that
	// has some and does not sort them draws them the slow way, with the same result)
	bool possible = a.binsPossible != 0u, bins = false;
	uint64_t bufferBytes = 0;
	a.binFeedback = a.binTiles != 0u ? frame_feedback(ctx, buffer, parts, possible, bins, bufferBytes) : nullptr;      // (parts after the first: what the first part decided)
	a.binsPossible = possible ? 1u : 0u;
	a.useBins = bins ? 1u : 0u;
	a.offBinSegs = a.offDir + (uint64_t)MAX_DIR_CHUNKS * 8;
	a.offBinSegCount = a.offBinSegs + (uint64_t)a.binTiles * BIN_SEG_CAP * sizeof(BinSeg);
	a.offBinPool = a.offBinSegs + bin_tables_bytes(a.binTiles);               // the pool comes last: it takes what the buffer has left
	if (a.binsPossible) {
		// The reference host allocates 200 000 000 bytes for this buffer whatever the window's size (main_progressive_octree.cpp:555), and the planes in
		// front of the bins grow with the frame: 1920 x 1080 leaves room for the whole pool, 2560 x 1440 for a quarter of it, 4K for none.  The pool is
		// what the ALLOCATION behind `buffer` has left (asked once per buffer); samples that find it full take the atomics, as before the bins.
		const uint64_t room = bufferBytes > a.offBinPool + 256 ? (bufferBytes - a.offBinPool - 256) / 16 : 0;
		a.binPoolCap = (uint32_t)std::min<uint64_t>(a.binPoolCap, room);
		if (room < BIN_POOL_MIN) { a.binsPossible = 0; a.useBins = 0; frame_feedback_no_bins(ctx, buffer); }
	}
	// (what thread 0 of r_visible publishes once the frame's counters are zero: never the value a stale or poisoned buffer holds)
	static std::atomic<uint32_t> launchSeq{(uint32_t)std::chrono::steady_clock::now().time_since_epoch().count() | 1u};
	a.launchSeq = launchSeq.fetch_add(2u);

	const DeviceInfo& dev = device_info();
	const uint32_t gridPixels = dev.numCUs * 8;
	const uint32_t gridNodes = (a.nodeCapacity + TPB - 1) / TPB;
	// one draw workgroup per CU: its 128 x 128-pixel tile takes 64-128 KB of the CU's 160 KB of LDS.  The depth pass too, though two of
	// its 64 KB tiles would fit: the draw loop is ALU-bound, two workgroups per CU each run at half speed, and the last big items then
	// finish later (HQS frame 0.227 ms against 0.231 ms).
	const uint32_t gridDraw = dev.numCUs * (uint32_t)ctx.tune(KNOB_DRAW_MULT, 1);
	const bool whole = parts == RENDER_ALL;
	auto lines = [&]() {
		if (!u->showBoundingBox) return;
		SIMLOD_LAUNCH(r_lines_emit, dim3((SIMLOD_MAX_VISIBLE_NODES + TPB - 1) / TPB), dim3(TPB), stream, a, u->transformInv_updateBound);
		SIMLOD_LAUNCH(r_lines_raster, dim3((LINE_VERTEX_CAP / 2 + TPB - 1) / TPB), dim3(TPB), stream, a);
	};
	if (parts & RENDER_FIRST) {
		SIMLOD_LAUNCH(r_visible, dim3(gridNodes), dim3(TPB), stream, a);
		if (a.hqs) {
			SIMLOD_LAUNCH(r_draw<MODE_DEPTH>, dim3(gridDraw), dim3(DTPB), stream, a);
			if (a.useBins) SIMLOD_LAUNCH(r_overflow<MODE_DEPTH>, dim3(a.binTiles), dim3(OTPB), stream, a);
		} else {
			SIMLOD_LAUNCH(r_draw<MODE_MIN64>, dim3(gridDraw), dim3(DTPB), stream, a);
			if (a.useBins) SIMLOD_LAUNCH(r_overflow<MODE_MIN64>, dim3(a.binTiles), dim3(OTPB), stream, a);
			lines();
		}
	}
	if (a.hqs && (parts & RENDER_COLOR)) {
		SIMLOD_LAUNCH(r_draw<MODE_COLOR>, dim3(gridDraw), dim3(DTPB), stream, a);
		if (a.useBins) SIMLOD_LAUNCH(r_overflow<MODE_COLOR>, dim3(a.binTiles), dim3(OTPB), stream, a);
		if (!whole) SIMLOD_LAUNCH(r_unpack, dim3(gridPixels), dim3(TPB), stream, a);     // ranks all-reduce(SUM) the {R,G,B,count} plane
	}
	// whole HQS frames without debug lines resolve inside r_output
	const bool fused = a.hqs && whole && !u->showBoundingBox && colorbuffer != nullptr && ctx.tune(KNOB_RASTER_FUSED_RESOLVE, 1) != 0;
	if (a.hqs && (parts & RENDER_RESOLVE) && !fused) {
		SIMLOD_LAUNCH(r_resolve, dim3(gridPixels), dim3(TPB), stream, a);
		lines();
	}
	if (parts & RENDER_OUTPUT) {
		const dim3 gridOutput((uint32_t)(a.W + OUT_TW - 1) / OUT_TW, (uint32_t)(a.H + OUT_TH - 1) / OUT_TH);
		if (fused) SIMLOD_LAUNCH(r_output<true>, gridOutput, dim3(TPB), stream, a);
		else SIMLOD_LAUNCH(r_output<false>, gridOutput, dim3(TPB), stream, a);
	}
	if (profile_enabled()) profile_close(stream);
	return (int)hipGetLastError();
}

}  // namespace simlod
