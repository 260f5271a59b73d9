/* oracle_support.c — TEST INFRASTRUCTURE ONLY.
 *
 *  (1) by-value trampolines for the three kernels of oracle/_ref (the reference sources compiled as host
 *      code): Python's ctypes cannot pass the 16-byte-aligned 480-byte Uniforms struct by value reliably,
 *      so the tests hand over a pointer and these functions do the by-value call with the exact
 *      parameter lists of progressive_octree_voxels.cu:804-816, render.cu:1084-1093 and reset.cu:20-29;
 *  (2) octree-image helpers working on the layout of include/simlod_abi.h: pointer rebasing between
 *      address spaces (device image -> host image) and an order-independent canonical dump used to
 *      compare octrees built by different implementations (node indices, chunk addresses and
 *      the order of samples inside a node are scheduling dependent; topology, per-node sample multisets,
 *      occupancy bitsets and counters are not).
 */
#include <stdint.h>
#include <string.h>

#include "simlod_abi.h"

typedef void (*ref_construct_fn)(SimlodUniforms, SimlodPoint*, uint32_t*, uint8_t*, SimlodNode*, SimlodStats*,
                                 unsigned long long*, void*, uint32_t*, uint32_t*);
typedef void (*ref_render_fn)(uint32_t*, SimlodUniforms, SimlodNode*, unsigned long long, SimlodStats*,
                              unsigned long long*, void*);
typedef void (*ref_reset_fn)(SimlodUniforms, uint8_t*, SimlodNode*, SimlodStats*, void*, uint32_t*, uint32_t*);

void ref_call_construct(void* fn, const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* persistent,
                        SimlodNode* nodes, SimlodStats* stats, unsigned long long* frameStart, void* cudaprint,
                        uint32_t* numBatchesUploaded, uint32_t* batchSizes) {
	((ref_construct_fn)fn)(*u, points, buffer, persistent, nodes, stats, frameStart, cudaprint, numBatchesUploaded, batchSizes);
}

void ref_call_render(void* fn, uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, void* surface,
                     SimlodStats* stats, unsigned long long* frameStart, void* cudaprint) {
	((ref_render_fn)fn)(buffer, *u, nodes, (unsigned long long)(uintptr_t)surface, stats, frameStart, cudaprint);
}

/* colorfilter.cu:163-169 (oracle/_ref/libref_filter.so) */
typedef void (*ref_filter_fn)(const SimlodUniforms, uint32_t*, SimlodNode*, uint32_t*, SimlodStats*);
void ref_call_filter(void* fn, const SimlodUniforms* u, uint32_t* buffer, SimlodNode* nodes, uint32_t* numNodes, SimlodStats* stats) {
	((ref_filter_fn)fn)(*u, buffer, nodes, numNodes, stats);
}

void ref_call_reset(void* fn, const SimlodUniforms* u, uint8_t* persistent, SimlodNode* nodes, SimlodStats* stats,
                    void* cudaprint, uint32_t* numBatchesUploaded, uint32_t* batchSizes) {
	((ref_reset_fn)fn)(*u, persistent, nodes, stats, cudaprint, numBatchesUploaded, batchSizes);
}

/* ---- pointer rebasing ------------------------------------------------------------------------------------
 * An octree image is (nodes array, persistent buffer).  Node::children point into the node array;
 * Node::grid/points/voxelChunks and Chunk::next point into the persistent buffer.  The image was valid at
 * (oldNodes, oldPers) and is now resident at (nodes, pers); rewrite every pointer in place.  Only chunks
 * reachable from a node are visited.  Returns the number of chunks visited, or -1 on a pointer that falls
 * outside the persistent buffer. */
static int in_range(uint64_t p, uint64_t base, uint64_t size) { return p >= base && p < base + size; }

/* General form: the image is resident (and walked) at (nodes, pers), was valid at (oldNodes, oldPers) and is rewritten to be
 * valid at (newNodes, newPers) — e.g. device addresses before an upload.  After a rewrite to foreign addresses the host copy
 * can no longer be walked. */
int64_t oracle_rebase_to(SimlodNode* nodes, uint32_t numNodes, uint8_t* pers, uint64_t persSize,
                         uint64_t oldNodes, uint64_t oldPers, uint64_t newNodes, uint64_t newPers);

int64_t oracle_rebase(SimlodNode* nodes, uint32_t numNodes, uint8_t* pers, uint64_t persSize,
                      uint64_t oldNodes, uint64_t oldPers) {
	return oracle_rebase_to(nodes, numNodes, pers, persSize, oldNodes, oldPers, (uint64_t)(uintptr_t)nodes, (uint64_t)(uintptr_t)pers);
}

int64_t oracle_rebase_to(SimlodNode* nodes, uint32_t numNodes, uint8_t* pers, uint64_t persSize,
                         uint64_t oldNodes, uint64_t oldPers, uint64_t newNodes, uint64_t newPers) {
	int64_t chunks = 0;
	const uint64_t hereP = (uint64_t)(uintptr_t)pers;
	SimlodAllocatorGlobal* a = (SimlodAllocatorGlobal*)pers;
	a->buffer = (uint8_t*)(uintptr_t)newPers;
	for (uint32_t i = 0; i < numNodes; i++) {
		SimlodNode* n = &nodes[i];
		for (int k = 0; k < 8; k++) {
			uint64_t c = (uint64_t)(uintptr_t)n->children[k];
			if (c) n->children[k] = (SimlodNode*)(uintptr_t)(c - oldNodes + newNodes);
		}
		uint64_t g = (uint64_t)(uintptr_t)n->grid;
		if (g) { if (!in_range(g, oldPers, persSize)) return -1; n->grid = (SimlodOccupancyGrid*)(uintptr_t)(g - oldPers + newPers); }
		SimlodChunk** heads[2] = {&n->points, &n->voxelChunks};
		for (int h = 0; h < 2; h++) {
			uint64_t p = (uint64_t)(uintptr_t)*heads[h];
			if (!p) continue;
			if (!in_range(p, oldPers, persSize)) return -1;
			*heads[h] = (SimlodChunk*)(uintptr_t)(p - oldPers + newPers);
			SimlodChunk* c = (SimlodChunk*)(uintptr_t)(p - oldPers + hereP);       /* walk through the resident copy */
			while (c) {
				chunks++;
				uint64_t nx = (uint64_t)(uintptr_t)c->next;
				if (!nx) break;
				if (!in_range(nx, oldPers, persSize)) return -1;
				c->next = (SimlodChunk*)(uintptr_t)(nx - oldPers + newPers);
				c = (SimlodChunk*)(uintptr_t)(nx - oldPers + hereP);
			}
		}
	}
	return chunks;
}

/* ---- canonical dump ---------------------------------------------------------------------------------------- */
typedef struct OracleNodeDump {
	uint64_t key;            /* level<<60 | X<<40 | Y<<20 | Z                                   */
	uint32_t level, X, Y, Z;
	uint32_t isLeaf, counter, numPoints, numVoxels, numVoxelsStored, countIteration;
	uint32_t hasGrid, gridPopcount;
	uint32_t pointChunks, voxelChunks;      /* linked chunks (list length)                        */
	uint64_t gridHash;                      /* FNV-1a over the 65536 words (0 when no grid)       */
	uint64_t pointsSum, pointsXor;          /* order-independent multiset hash of the stored points */
	uint64_t voxelPosSum, voxelPosXor;      /* same over voxel POSITIONS only (colour is H6-dependent) */
	uint64_t childMask;                     /* bit i: children[i] != null                         */
	uint8_t  name[24];
} OracleNodeDump;

static uint64_t mix64(uint64_t x) { /* splitmix64 finaliser */
	x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x;
}

static void hash_list(const SimlodChunk* c, uint32_t count, int withColor, uint64_t* sum, uint64_t* xr, uint32_t* linked) {
	uint64_t s = 0, x = 0; uint32_t l = 0;
	const SimlodChunk* cur = c;
	for (uint32_t i = 0; i < count && cur; i++) {
		if (i > 0 && i % SIMLOD_POINTS_PER_CHUNK == 0) { cur = cur->next; if (!cur) break; }
		const SimlodPoint* p = &cur->points[i % SIMLOD_POINTS_PER_CHUNK];
		uint32_t w[4]; memcpy(w, p, 16);
		uint64_t a = ((uint64_t)w[0] << 32) | w[1], b = ((uint64_t)w[2] << 32) | (withColor ? w[3] : 0u);
		uint64_t h = mix64(a ^ mix64(b + 0x9e3779b97f4a7c15ull));
		s += h; x ^= mix64(h + 1);
	}
	for (cur = c; cur; cur = cur->next) l++;
	*sum = s; *xr = x; *linked = l;
}

void oracle_dump(const SimlodNode* nodes, uint32_t numNodes, OracleNodeDump* out) {
	for (uint32_t i = 0; i < numNodes; i++) {
		const SimlodNode* n = &nodes[i];
		OracleNodeDump* d = &out[i];
		memset(d, 0, sizeof(*d));
		d->key = ((uint64_t)n->level << 60) | ((uint64_t)n->X << 40) | ((uint64_t)n->Y << 20) | (uint64_t)n->Z;
		d->level = n->level; d->X = n->X; d->Y = n->Y; d->Z = n->Z;
		uint64_t cm = 0;
		for (int k = 0; k < 8; k++) if (n->children[k]) cm |= 1ull << k;
		d->childMask = cm; d->isLeaf = cm == 0;
		d->counter = n->counter; d->numPoints = n->numPoints; d->numVoxels = n->numVoxels;
		d->numVoxelsStored = n->numVoxelsStored; d->countIteration = n->countIteration;
		memcpy(d->name, n->name, 20);
		if (n->grid) {
			d->hasGrid = 1;
			uint64_t h = 0xcbf29ce484222325ull; uint32_t pc = 0;
			for (uint32_t w = 0; w < SIMLOD_GRID_NUM_WORDS; w++) {
				uint32_t v = n->grid->values[w];
				pc += (uint32_t)__builtin_popcount(v);
				h = (h ^ v) * 0x100000001b3ull;
			}
			d->gridHash = h; d->gridPopcount = pc;
		}
		hash_list(n->points, n->numPoints, 1, &d->pointsSum, &d->pointsXor, &d->pointChunks);
		hash_list(n->voxelChunks, n->numVoxelsStored, 0, &d->voxelPosSum, &d->voxelPosXor, &d->voxelChunks);
	}
}

/* Copy the samples of one node's list into a flat array (for per-cell colour membership tests). */
uint32_t oracle_gather(const SimlodChunk* head, uint32_t count, SimlodPoint* out) {
	const SimlodChunk* cur = head; uint32_t i = 0;
	for (; i < count && cur; i++) {
		if (i > 0 && i % SIMLOD_POINTS_PER_CHUNK == 0) { cur = cur->next; if (!cur) break; }
		out[i] = cur->points[i % SIMLOD_POINTS_PER_CHUNK];
	}
	return i;
}

/* Structural invariants every octree image must satisfy after a completed kernel_construct, whatever the input was
 * (size-independent properties for the full-size tests).  Returns 0 or the number of the first violated rule. */
/* allowOverfull: leaves above the 50 000-point limit are legitimate where splits were deferred for lack of scratch space
 * (the HIP builder's answer to a regime in which the reference drops points). */
int oracle_check_invariants_ex(const SimlodNode* nodes, uint32_t numNodes, uint64_t* totalPoints, uint64_t* totalVoxels,
                               uint64_t* pointChunks, uint64_t* voxelChunks, uint64_t* grids, int allowOverfull) {
	uint64_t tp = 0, tv = 0, pc = 0, vc = 0, g = 0;
	if (numNodes == 0 || (numNodes - 1) % 8 != 0) return 1;                       /* nodes are created 8 at a time */
	for (uint32_t i = 0; i < numNodes; i++) {
		const SimlodNode* n = &nodes[i];
		int kids = 0;
		for (int k = 0; k < 8; k++) if (n->children[k]) kids++;
		if (kids != 0 && kids != 8) return 2;
		if (n->grid) g++;
		if (kids == 8) {
			if (n->numPoints != 0 || n->points != NULL) return 3;                   /* inner nodes keep no points */
			if (!n->grid) return 4;
			for (int k = 0; k < 8; k++) {
				const SimlodNode* c = n->children[k];
				if (c->level != n->level + 1) return 5;
				if (c->X != 2 * n->X + (uint32_t)((k >> 2) & 1) || c->Y != 2 * n->Y + (uint32_t)((k >> 1) & 1) || c->Z != 2 * n->Z + (uint32_t)(k & 1)) return 6;
				if (c->level < 20 && c->name[c->level] != (uint8_t)('0' + k)) return 7;
			}
		} else {
			if (n->counter != n->numPoints) return 8;                                /* everything counted was stored */
			if (!allowOverfull && n->numPoints > SIMLOD_MAX_POINTS_PER_NODE && n->level < SIMLOD_MAX_DEPTH) return 9;
			if (i != 0 && n->grid) return 10;                                        /* only inner nodes and the root own a grid */
		}
		if (n->numVoxels != n->numVoxelsStored) return 11;
		if (n->grid) {
			uint32_t pcnt = 0;
			for (uint32_t w = 0; w < SIMLOD_GRID_NUM_WORDS; w++) pcnt += (uint32_t)__builtin_popcount(n->grid->values[w]);
			if (i != 0 && pcnt != n->numVoxels) return 12;                          /* the root may hold duplicates (grid cleared at its split) */
			if (i == 0 && pcnt > n->numVoxels) return 12;
		} else if (n->numVoxels != 0) return 13;
		uint32_t l = 0;
		for (const SimlodChunk* c = n->points; c; c = c->next) l++;
		if (l != (n->numPoints + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK) return 14;
		pc += l; l = 0;
		for (const SimlodChunk* c = n->voxelChunks; c; c = c->next) l++;
		if (l != (n->numVoxels + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK) return 15;
		vc += l;
		tp += n->numPoints; tv += n->numVoxels;
	}
	*totalPoints = tp; *totalVoxels = tv; *pointChunks = pc; *voxelChunks = vc; *grids = g;
	return 0;
}

int oracle_check_invariants(const SimlodNode* nodes, uint32_t numNodes, uint64_t* totalPoints, uint64_t* totalVoxels,
                            uint64_t* pointChunks, uint64_t* voxelChunks, uint64_t* grids) {
	return oracle_check_invariants_ex(nodes, numNodes, totalPoints, totalVoxels, pointChunks, voxelChunks, grids, 0);
}

/* Analysis helper (tools/analyze_candidates.py): for every point of a batch, the deepest grid-owning node of its path, the
 * byte address of the occupancy word it would probe, and whether that bit is already set. */
void oracle_probe_batch(const SimlodNode* nodes, const SimlodPoint* pts, uint32_t n, float minx, float miny, float minz, float size,
                        uint32_t* outNode, uint64_t* outWordAddr, uint8_t* outSet) {
	for (uint32_t i = 0; i < n; i++) {
		const SimlodPoint* p = &pts[i];
		uint32_t X = (uint32_t)(1048576.0f * (p->x - minx) / size), Y = (uint32_t)(1048576.0f * (p->y - miny) / size), Z = (uint32_t)(1048576.0f * (p->z - minz) / size);
		uint32_t pX = (uint32_t)(268435456.0f * (p->x - minx) / size), pY = (uint32_t)(268435456.0f * (p->y - miny) / size), pZ = (uint32_t)(268435456.0f * (p->z - minz) / size);
		const SimlodNode* cur = &nodes[0]; const SimlodNode* deepest = cur->grid ? cur : NULL;
		for (int level = 0; level < SIMLOD_MAX_DEPTH; level++) {
			int s = SIMLOD_MAX_DEPTH - level - 1;
			const SimlodNode* ch = cur->children[(((X >> s) & 1u) << 2) | (((Y >> s) & 1u) << 1) | ((Z >> s) & 1u)];
			if (!ch) break;
			if (cur->grid) deepest = cur;
			cur = ch;
		}
		outNode[i] = 0xffffffffu; outWordAddr[i] = 0; outSet[i] = 1;
		if (!deepest) continue;
		uint32_t sh = (uint32_t)(SIMLOD_MAX_DEPTH + 1) - deepest->level;
		uint32_t cell = ((pX >> sh) & 127u) + ((pY >> sh) & 127u) * 128u + ((pZ >> sh) & 127u) * 16384u;
		outNode[i] = (uint32_t)(deepest - nodes);
		outWordAddr[i] = (uint64_t)(uintptr_t)&deepest->grid->values[cell >> 5];
		outSet[i] = (uint8_t)((deepest->grid->values[cell >> 5] >> (cell & 31u)) & 1u);
	}
}
