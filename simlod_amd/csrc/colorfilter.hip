// colorfilter.hip — voxel colour filtering for MI355X (gfx950): the `kernel` of modules/progressive_octree/colorfilter.cu:163-414
// (dead in the reference — its host call is commented out, main_progressive_octree.cpp:430-462 — kept behind the
// same argument list).
//
// What it computes: bottom-up over the inner nodes, every voxel of a node is rewritten as the AVERAGE colour of the samples of the
// node's children (their points, or their already filtered voxels) that fall into the voxel's cell, instead of the colour of
// whichever point happened to claim the cell first — which also removes the scheduling dependence of voxel colours.
// Per node (colorfilter.cu:258-395): for each child octant, the child's samples are accumulated in a 64^3 grid of packed 64-bit sums
// (R << 46 | G << 28 | B << 10 | count, :147), first hits are listed (:155-159), and every listed cell becomes one voxel of the node,
// cell centre + average colour (:312-357), written over the node's voxel list octant by octant.
//
// Here: nodes are handled in waves by their height above the leaves — height 1 first, ten waves as the reference's `abc` loop
// (:247) — one workgroup per node, any number of nodes of a wave in flight; the node's voxel chunks are looked up through a
// per-workgroup directory in LDS instead of a pointer chase per voxel.  The arithmetic is the reference's, operation by operation;
// where the reference is undefined (a cell with more than 1023 child samples overflows its 10-bit count, :143-151; levels above 17
// shift by a negative amount, :120) the surplus voxels are dropped / the node is skipped instead of writing out of bounds.
#include "simlod_device.hpp"
#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {

static constexpr uint32_t CF_TPB = 256;
static constexpr uint32_t CF_SIDE = 64;
static constexpr uint32_t CF_CELLS = CF_SIDE * CF_SIDE * CF_SIDE;        // colorfilter.cu:17-18
static constexpr uint32_t CF_ACCEPTED = 300000;                           // colorfilter.cu:20
static constexpr uint32_t CF_DIR = 2304;                                   // voxel chunks of a node kept in LDS (2^21 cells / 1000)
static constexpr uint32_t CF_WAVES = 10;                                   // colorfilter.cu:247

struct FilterArgs {
	uint8_t* mom;
	SimlodNode* nodes;
	SimlodStats* stats;
	const uint32_t* numNodesPtr;
	uint32_t nodeCapacity;
	float cubeSize, minx, miny, minz;
	float octreeSizeX, octreeSizeY, octreeSizeZ;          // (boxMin + cubeSize) - boxMin per axis, colorfilter.cu:75-79
	uint64_t offHeights, offGrids, offAccepted;
};

__device__ __forceinline__ uint32_t cf_num_nodes(const FilterArgs& a) { return min(a.numNodesPtr != nullptr ? *a.numNodesPtr : a.stats->numNodes, a.nodeCapacity); }

// leaves count as filtered (colorfilter.cu:231-239); every inner node waits for its wave
__global__ __launch_bounds__(CF_TPB) void cf_init(FilterArgs a) {
	const uint32_t i = blockIdx.x * CF_TPB + threadIdx.x;
	if (i >= cf_num_nodes(a)) return;
	const bool leaf = node_is_leaf(a.nodes + i);
	a.nodes[i].isFiltered = leaf ? 1 : 0;
	(a.mom + a.offHeights)[i] = leaf ? 0 : 255;
}

// height of the inner nodes of one level: 1 + the largest height among the children (leaves: 0).  Levels are visited deepest first.
__global__ __launch_bounds__(CF_TPB) void cf_height(FilterArgs a, uint32_t level) {
	const uint32_t i = blockIdx.x * CF_TPB + threadIdx.x;
	if (i >= cf_num_nodes(a)) return;
	const SimlodNode* n = a.nodes + i;
	uint8_t* heights = a.mom + a.offHeights;
	if (n->level != level || heights[i] == 0) return;
	uint32_t h = 0;
	for (int k = 0; k < 8; k++) if (n->children[k] != nullptr) h = max(h, (uint32_t)heights[(uint32_t)(n->children[k] - a.nodes)]);
	heights[i] = (uint8_t)min(h + 1u, 254u);
}

struct FilterShared {
	SimlodChunk* dir[CF_DIR];
	uint32_t acceptedChild, acceptedTotal, dirCount;
};

// colorfilter.cu:58-161: the samples of one child list into the workgroup's sample grid
__device__ void cf_sample(const FilterArgs& a, FilterShared& sh, const SimlodNode* node, uint32_t childIndex, const SimlodChunk* chunk, uint32_t numSamples,
                          unsigned long long* grid, uint32_t* accepted) {
	if (numSamples == 0 || chunk == nullptr) return;
	const uint32_t level = node->level;
	uint32_t chunkIndex = 0;
	for (uint32_t pointIndex = threadIdx.x; pointIndex < numSamples; pointIndex += CF_TPB) {
		const uint32_t target = pointIndex / SIMLOD_POINTS_PER_CHUNK;
		while (chunkIndex < target && chunk != nullptr) { chunk = chunk->next; chunkIndex++; }
		if (chunk == nullptr) break;
		const SimlodPoint p = chunk->points[pointIndex % SIMLOD_POINTS_PER_CHUNK];
		// :116-128 — integer coordinate relative to the root at 2^24, then the node's 128-cell grid, then the child's 64-cell octant
		// (divided per axis by octreeSize = (boxMin + cubeSize) - boxMin, colorfilter.cu:75-79, 117-119: not cubeSize itself when boxMin != 0 in fp32)
		const uint32_t pXf = (uint32_t)((16777216.0f * (p.x - a.minx)) / a.octreeSizeX);
		const uint32_t pYf = (uint32_t)((16777216.0f * (p.y - a.miny)) / a.octreeSizeY);
		const uint32_t pZf = (uint32_t)((16777216.0f * (p.z - a.minz)) / a.octreeSizeZ);
		const uint32_t sh17 = 17u - level;
		const float pX = (float)((pXf >> sh17) % CF_SIDE), pY = (float)((pYf >> sh17) % CF_SIDE), pZ = (float)((pZf >> sh17) % CF_SIDE);
		uint32_t voxelIndex = (uint32_t)(pX + pY * (float)CF_SIDE + pZ * (float)(CF_SIDE * CF_SIDE));
		voxelIndex = min(voxelIndex, CF_CELLS - 1u);
		const unsigned long long R = p.color & 0xffu, G = (p.color >> 8) & 0xffu, B = (p.color >> 16) & 0xffu;
		const unsigned long long c64 = (R << 46) | (G << 28) | (B << 10) | 1ull;
		const unsigned long long old = atomicAdd(&grid[voxelIndex], c64);
		if ((old & 0x3ffull) == 0ull) {                                   // first hit of the cell (:155)
			atomicAdd(&sh.acceptedTotal, 1u);
			const uint32_t ai = atomicAdd(&sh.acceptedChild, 1u);
			if (ai < CF_ACCEPTED) accepted[ai] = (childIndex << 24) | voxelIndex;
		}
	}
}

__global__ __launch_bounds__(CF_TPB) void cf_wave(FilterArgs a, uint32_t height) {
	__shared__ FilterShared sh;
	const uint32_t numNodes = cf_num_nodes(a);
	const uint8_t* heights = a.mom + a.offHeights;
	unsigned long long* grid = reinterpret_cast<unsigned long long*>(a.mom + a.offGrids) + (uint64_t)blockIdx.x * CF_CELLS;
	uint32_t* accepted = reinterpret_cast<uint32_t*>(a.mom + a.offAccepted) + (uint64_t)blockIdx.x * CF_ACCEPTED;
	for (uint32_t i = blockIdx.x; i < numNodes; i += gridDim.x) {
		if (heights[i] != height) continue;
		SimlodNode* node = a.nodes + i;
		if (node->level > 17u) continue;                                  // :120 shifts by (17 - level)
		__syncthreads();
		if (threadIdx.x == 0) {                                           // the node's voxel chunks, once
			uint32_t k = 0;
			for (SimlodChunk* c = node->voxelChunks; c != nullptr && k < CF_DIR; c = c->next) sh.dir[k++] = c;
			sh.dirCount = k; sh.acceptedTotal = 0;
		}
		__syncthreads();
		const float nodeSize = a.cubeSize / exp2_int(node->level);
		const float nodeMin_x = ((float)node->X + 0.0f) * nodeSize + a.minx;
		const float nodeMin_y = ((float)node->Y + 0.0f) * nodeSize + a.miny;
		const float nodeMin_z = ((float)node->Z + 0.0f) * nodeSize + a.minz;
		const uint32_t numVoxels = node->numVoxels;
		uint32_t voxelIndexOffset = 0;
		for (uint32_t childIndex = 0; childIndex < 8; childIndex++) {
			const SimlodNode* child = node->children[childIndex];
			if (child == nullptr) continue;
			__syncthreads();
			if (threadIdx.x == 0) sh.acceptedChild = 0;
			__syncthreads();
			cf_sample(a, sh, node, childIndex, child->points, child->numPoints, grid, accepted);
			cf_sample(a, sh, node, childIndex, child->voxelChunks, child->numVoxels, grid, accepted);
			__threadfence_block();
			__syncthreads();
			const uint32_t count = min(sh.acceptedChild, CF_ACCEPTED);
			for (uint32_t ai = threadIdx.x; ai < count; ai += CF_TPB) {       // :312-357
				const uint32_t enc = accepted[ai], ci = (enc >> 24) & 0xffu, voxelIndex = enc & 0x00ffffffu;
				const unsigned long long c64 = grid[voxelIndex];
				unsigned long long R = (c64 >> 46) & 0x3ffffull, G = (c64 >> 28) & 0x3ffffull, B = (c64 >> 10) & 0x3ffffull;
				const unsigned long long C = c64 & 0x3ffull;
				if (C != 0ull) { R = (R / C) & 0xffull; G = (G / C) & 0xffull; B = (B / C) & 0xffull; } else { R = G = B = 0; }
				grid[voxelIndex] = 0;                                          // :356
				const uint32_t slot = voxelIndexOffset + ai;
				if (slot >= numVoxels || slot / SIMLOD_POINTS_PER_CHUNK >= sh.dirCount) continue;   // more first hits than voxels: see the header
				const int pX = (int)(((ci >> 2) & 1u) * CF_SIDE + voxelIndex % CF_SIDE);
				const int pY = (int)(((ci >> 1) & 1u) * CF_SIDE + (voxelIndex % (CF_SIDE * CF_SIDE)) / CF_SIDE);
				const int pZ = (int)((ci & 1u) * CF_SIDE + voxelIndex / (CF_SIDE * CF_SIDE));
				SimlodPoint v;
				v.x = nodeMin_x + (nodeSize * ((float)pX + 0.5f)) / 128.0f;
				v.y = nodeMin_y + (nodeSize * ((float)pY + 0.5f)) / 128.0f;
				v.z = nodeMin_z + (nodeSize * ((float)pZ + 0.5f)) / 128.0f;
				v.color = (uint32_t)(R | (G << 8) | (B << 16));
				sh.dir[slot / SIMLOD_POINTS_PER_CHUNK]->points[slot % SIMLOD_POINTS_PER_CHUNK] = v;
			}
			voxelIndexOffset += sh.acceptedChild;
		}
		__syncthreads();
		if (threadIdx.x == 0) node->isFiltered = 1;
	}
}

uint64_t colorfilter_min_bytes(uint32_t nodeCapacity) { return 4096 + (uint64_t)nodeCapacity + 256 + (uint64_t)CF_CELLS * 8 + (uint64_t)CF_ACCEPTED * 4; }

int launch_colorfilter(Context& ctx, const SimlodUniforms* u, uint32_t* buffer, SimlodNode* nodes, const uint32_t* numNodes, SimlodStats* stats, hipStream_t stream) {
	FilterArgs a{};
	a.mom = reinterpret_cast<uint8_t*>(buffer); a.nodes = nodes; a.stats = stats; a.numNodesPtr = numNodes;
	a.nodeCapacity = ctx.nodeCapacity.load();
	const float bx = u->boxMax.x - u->boxMin.x, by = u->boxMax.y - u->boxMin.y, bz = u->boxMax.z - u->boxMin.z;
	a.cubeSize = fmaxf(fmaxf(bx, by), bz);
	a.minx = u->boxMin.x; a.miny = u->boxMin.y; a.minz = u->boxMin.z;
	a.octreeSizeX = (a.minx + a.cubeSize) - a.minx; a.octreeSizeY = (a.miny + a.cubeSize) - a.miny; a.octreeSizeZ = (a.minz + a.cubeSize) - a.minz;
	a.offHeights = 4096;
	a.offGrids = (a.offHeights + a.nodeCapacity + 255) / 256 * 256;
	const uint64_t perWg = (uint64_t)CF_CELLS * 8 + (uint64_t)CF_ACCEPTED * 4;
	if (u->momentaryBufferCapacity < a.offGrids + perWg) return (int)hipErrorInvalidValue;
	const uint32_t wgs = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(1u, device_info().numCUs / 4), (u->momentaryBufferCapacity - a.offGrids) / perWg);
	a.offAccepted = a.offGrids + (uint64_t)wgs * CF_CELLS * 8;
	hipError_t e = hipMemsetAsync(a.mom + a.offGrids, 0, (size_t)wgs * CF_CELLS * 8, stream);      // :222-226
	if (e != hipSuccess) return (int)e;
	const uint32_t gridNodes = (a.nodeCapacity + CF_TPB - 1) / CF_TPB;
	SIMLOD_LAUNCH(cf_init, dim3(gridNodes), dim3(CF_TPB), stream, a);
	for (int level = SIMLOD_MAX_DEPTH; level >= 0; level--) SIMLOD_LAUNCH(cf_height, dim3(gridNodes), dim3(CF_TPB), stream, a, (uint32_t)level);
	for (uint32_t h = 1; h <= CF_WAVES; h++) SIMLOD_LAUNCH(cf_wave, dim3(wgs), dim3(CF_TPB), stream, a, h);
	if (profile_enabled()) profile_close(stream);
	return (int)hipGetLastError();
}

}  // namespace simlod
