/* simlod_oracle.c — TEST INFRASTRUCTURE ONLY: single-thread CPU restatement of SimLOD's two hot paths.
 *
 * What it is: an independent, readable C restatement of the reference's octree builder
 * (modules/progressive_octree/progressive_octree_voxels.cu), reset kernel (reset.cu) and software
 * rasteriser (render.cu), operating on the SAME memory layout (include/simlod_abi.h) so that its octree
 * image can be compared, node for node and byte for byte, with one built by the reference's own sources
 * (oracle/_ref, see oracle/Makefile) and with one built by the HIP kernels.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the CHECKER / CPU
 * baseline only.  The product path (simlod_amd/) never links, loads or calls it.
 *
 * Pinning: the reference ships no tests or golden vectors.  This restatement is pinned
 * against the reference ITSELF: tests/test_oracle_pin.py runs the same seeded inputs through
 * oracle/_ref/libref_*.so (the reference sources compiled in place) and through this file and requires
 * identical node arrays, chunk contents, occupancy grids, Stats and pre-EDL framebuffers; the committed
 * fixtures under tests/golden/ were generated from oracle/_ref by tests/golden/make_golden.py.
 *
 * Differences from the reference, all deliberate:
 *   - chunk lists are appended through a per-node tail table kept OUTSIDE the octree image (the
 *     reference walks the list from its head for every insert, voxels.cu:606-610, 688-692: quadratic);
 *   - float->int conversions that would be undefined in C for non-finite input are made explicit and
 *     yield "reject";
 *   - the EDL pass (render.cu:1255-1325) covers every full 16x16 tile instead of a launch-geometry
 *     dependent subset and clamps its neighbour index to W*H-1;
 *   - capacity cliffs of the reference are reported through oracle_last_error() instead
 *     of silently corrupting memory;
 *   - ONE EXTENSION the reference has no counterpart for (it is single-GPU): oracle_set_trunk_mask() names
 *     nodes of levels 0-2 that split whatever their count — the multi-GPU layer's rule "a shared upper
 *     node is inner on every rank iff the GLOBAL count under it exceeds 50 000" (simlod_amd/distributed.py
 *     trunk_mask, include/simlod_hip.h simlod_context_set_trunk_mask).  With the mask at zero — its state
 *     after oracle_create, and the only state the pin tests and golden fixtures use — nothing differs.
 * Everything else — operation order of every fp32/fp64 expression, traversal order, allocation order —
 * follows the cited lines, so a serial run reproduces the reference's serial run exactly.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "simlod_abi.h"

#define VOXEL_BACKLOG_CAPACITY 10000000u /* voxels.cu:21  */
#define SPILLING_CAPACITY      100000u   /* voxels.cu:847 */
#define SPILLED_CAPACITY       10000000u /* voxels.cu:853 */
#define SPILLED_INSERT_LIMIT   3000000u  /* voxels.cu:628 */
#define CHUNK_QUEUE_CAPACITY   1000000u  /* voxels.cu:856 */

enum {
	ORACLE_OK = 0,
	ORACLE_ERR_BACKLOG = 1,      /* > 10 M new voxels in one batch            */
	ORACLE_ERR_SPILLING = 2,     /* > 100 k spilling nodes in one round        */
	ORACLE_ERR_SPILLED = 3,      /* > 3 M spilled points in one batch          */
	ORACLE_ERR_NODES = 4,        /* node array exhausted                       */
	ORACLE_ERR_NULL_CHUNK = 5,   /* voxels.cu:599-604 "chunk is NULL"          */
	ORACLE_ERR_VISIBLE = 6,      /* > 100 k visible nodes                      */
	ORACLE_ERR_DEPTH = 7         /* a node at MAX_DEPTH would have to split    */
};

typedef struct OracleCtx {
	SimlodChunk** chunkQueue;
	SimlodNode**  spilling;
	SimlodPoint*  spilled;
	SimlodPoint*  backlogVoxels;
	SimlodNode**  backlogTargets;
	SimlodChunk** pointTail;   /* per node index: last chunk of node->points                      */
	SimlodChunk** voxelTail;   /* per node index: last chunk of node->voxelChunks                 */
	SimlodChunk** pointCur;    /* per node index: chunk that holds slot numPoints-1 (insert cursor) */
	SimlodChunk** voxelCur;    /* per node index: chunk that holds slot numVoxelsStored-1          */
	uint32_t      numSpilling, numSpilled, numBacklog;
	uint32_t      maxNodes;
	int           lastError;
	uint64_t      trunkMask[2]; /* EXTENSION: nodes of levels 0-2 that must be inner (bit: trunk_index) */
} OracleCtx;

OracleCtx* oracle_create(uint32_t maxNodes) {
	OracleCtx* c = (OracleCtx*)calloc(1, sizeof(OracleCtx));
	c->maxNodes = maxNodes;
	c->chunkQueue = (SimlodChunk**)calloc(CHUNK_QUEUE_CAPACITY, sizeof(void*));
	c->spilling = (SimlodNode**)calloc(SPILLING_CAPACITY, sizeof(void*));
	c->spilled = (SimlodPoint*)malloc((size_t)SPILLED_CAPACITY * sizeof(SimlodPoint));
	c->backlogVoxels = (SimlodPoint*)malloc((size_t)VOXEL_BACKLOG_CAPACITY * sizeof(SimlodPoint));
	c->backlogTargets = (SimlodNode**)malloc((size_t)VOXEL_BACKLOG_CAPACITY * sizeof(void*));
	c->pointTail = (SimlodChunk**)calloc(maxNodes, sizeof(void*));
	c->voxelTail = (SimlodChunk**)calloc(maxNodes, sizeof(void*));
	c->pointCur = (SimlodChunk**)calloc(maxNodes, sizeof(void*));
	c->voxelCur = (SimlodChunk**)calloc(maxNodes, sizeof(void*));
	return c;
}

void oracle_destroy(OracleCtx* c) {
	if (!c) return;
	free(c->chunkQueue); free(c->spilling); free(c->spilled); free(c->backlogVoxels);
	free(c->backlogTargets); free(c->pointTail); free(c->voxelTail); free(c->pointCur); free(c->voxelCur); free(c);
}

int oracle_last_error(const OracleCtx* c) { return c->lastError; }

/* EXTENSION (multi-GPU layer): bit trunk_index(level, X, Y, Z) set = that node of level 0, 1 or 2 splits as soon as
 * it exists, whatever it holds.  Bit 0: the root; 1 + c: level 1, c = x << 2 | y << 1 | z; 9 + c: level 2, c = the
 * level-1 octant << 3 | the octant below it (the cell codes of simlod_amd/distributed.py cell_codes). */
void oracle_set_trunk_mask(OracleCtx* c, uint64_t lo, uint64_t hi) { c->trunkMask[0] = lo; c->trunkMask[1] = hi; }
static uint32_t trunk_index(uint32_t level, uint32_t X, uint32_t Y, uint32_t Z) {
	if (level == 0) return 0;
	if (level == 1) return 1u + ((X & 1u) << 2 | (Y & 1u) << 1 | (Z & 1u));
	return 9u + ((((X >> 1) & 1u) << 2 | ((Y >> 1) & 1u) << 1 | ((Z >> 1) & 1u)) << 3 | ((X & 1u) << 2 | (Y & 1u) << 1 | (Z & 1u)));
}
static int trunk_forced(const OracleCtx* c, const SimlodNode* n) {
	if (n->level >= 3) return 0;
	uint32_t i = trunk_index(n->level, n->X, n->Y, n->Z);
	return (int)((c->trunkMask[i >> 6] >> (i & 63u)) & 1u);
}

/* ---- AllocatorGlobal::alloc, utils.h.cu:185-197 -------------------------------------------------- */
static uint8_t* persistent_alloc(SimlodAllocatorGlobal* a, uint64_t size) {
	uint64_t old = a->offset;
	a->offset = old + SIMLOD_ALLOC_ROUND(size);
	return a->buffer + old;
}

static int node_is_leaf(const SimlodNode* n) { /* structures.cuh:104-116 */
	for (int i = 0; i < 8; i++) if (n->children[i]) return 0;
	return 1;
}

/* octree cube, voxels.cu:860-863 / render.cu:1135-1137 */
static float octree_size(const SimlodUniforms* u) {
	float bx = u->boxMax.x - u->boxMin.x, by = u->boxMax.y - u->boxMin.y, bz = u->boxMax.z - u->boxMin.z;
	return fmaxf(fmaxf(bx, by), bz);
}

/* fp32 -> u32 as the device does it for in-range values (truncation); voxels.cu:148-155 */
static uint32_t quantize(float scale, float p, float min, float size) {
	float v = scale * (p - min) / size;
	if (!(v >= 0.0f)) return 0u;               /* NaN / negative: undefined in the reference */
	if (v >= 4294967296.0f) return 0xffffffffu;
	return (uint32_t)v;
}

static int child_index(uint32_t X, uint32_t Y, uint32_t Z, int level) { /* voxels.cu:171-179 */
	int s = SIMLOD_MAX_DEPTH - level - 1;
	return (int)((((X >> s) & 1u) << 2) | (((Y >> s) & 1u) << 1) | ((Z >> s) & 1u));
}

/* ---- reset.cu:20-86 ------------------------------------------------------------------------------------ */
void oracle_reset(OracleCtx* c, const SimlodUniforms* u, uint8_t* persistent, SimlodNode* nodes,
                  SimlodStats* stats, uint32_t* numBatchesUploaded, uint32_t* batchSizes) {
	SimlodAllocatorGlobal* a = (SimlodAllocatorGlobal*)persistent;
	a->buffer = persistent;
	a->offset = 16;
	memset(stats, 0, sizeof(*stats));
	stats->numNodes = 1;
	stats->frameID = (uint32_t)u->frameCounter;
	SimlodNode* root = &nodes[0];
	memset(root->children, 0, sizeof(root->children));
	root->isFiltered = 0;
	root->counter = 0; root->numPoints = 0; root->level = 0;
	root->X = root->Y = root->Z = 0;
	root->countIteration = 0;
	memset(root->name, 0, 20);
	root->name[0] = 'r';
	root->numVoxels = 0; root->numVoxelsStored = 0;
	root->voxelChunks = NULL;
	root->grid = (SimlodOccupancyGrid*)persistent_alloc(a, sizeof(SimlodOccupancyGrid));
	/* reset.cu leaves root->points untouched: a list from before the reset would stay linked while the
	 * allocator that owns its memory restarts at offset 16.  That is only sound when the pointer is
	 * already null (fresh buffers, or a root that had been split), so this restatement — like the HIP
	 * kernel — nulls it explicitly. */
	root->points = NULL;
	if (c) {
		memset(c->pointTail, 0, c->maxNodes * sizeof(void*));
		memset(c->voxelTail, 0, c->maxNodes * sizeof(void*));
		memset(c->pointCur, 0, c->maxNodes * sizeof(void*));
		memset(c->voxelCur, 0, c->maxNodes * sizeof(void*));
		c->lastError = ORACLE_OK;
	}
	*numBatchesUploaded = 0;
	for (uint32_t i = 0; i < SIMLOD_BATCH_STREAM_SIZE; i++) batchSizes[i] = 0;
	memset(root->grid->values, 0, sizeof(root->grid->values));
}

/* ---- octree construction ---------------------------------------------------------------------------- */
typedef struct BuildEnv {
	OracleCtx* c;
	SimlodNode* nodes;
	SimlodStats* stats;
	SimlodAllocatorGlobal* alloc;
	float minx, miny, minz, size;
} BuildEnv;

/* root-to-leaf descent, voxels.cu:157-189 (same loop in :438-469 and :559-591) */
static SimlodNode* descend(const BuildEnv* e, const SimlodPoint* p) {
	const float fGrid = 1048576.0f; /* pow(2, MAX_DEPTH) */
	uint32_t X = quantize(fGrid, p->x, e->minx, e->size);
	uint32_t Y = quantize(fGrid, p->y, e->miny, e->size);
	uint32_t Z = quantize(fGrid, p->z, e->minz, e->size);
	SimlodNode* cur = &e->nodes[0];
	for (int level = 0; level < SIMLOD_MAX_DEPTH; level++) {
		SimlodNode* ch = cur->children[child_index(X, Y, Z, level)];
		if (!ch) break;
		cur = ch;
	}
	return cur;
}

/* countPoint, voxels.cu:145-220 */
static void count_point(BuildEnv* e, const SimlodPoint* p, uint32_t countIteration) {
	SimlodNode* leaf = descend(e, p);
	if (leaf->countIteration < countIteration) {
		uint32_t old = leaf->counter;
		leaf->counter = old + 1;
		if (old <= SIMLOD_MAX_POINTS_PER_NODE && old + 1 > SIMLOD_MAX_POINTS_PER_NODE) {
			if (e->c->numSpilling >= SPILLING_CAPACITY) { e->c->lastError = ORACLE_ERR_SPILLING; return; }
			e->c->spilling[e->c->numSpilling++] = leaf;
		}
	}
}

/* doCounting, voxels.cu:124-306; returns 1 when no leaf spilled */
static int do_counting(BuildEnv* e, const SimlodPoint* pts, uint32_t n, uint32_t countIteration) {
	OracleCtx* c = e->c;
	c->numSpilling = 0;
	for (uint32_t i = 0; i < n; i++) count_point(e, &pts[i], countIteration);
	uint32_t numSpilledBefore = c->numSpilled; /* processRange(*numSpilledPoints) captures the size first */
	for (uint32_t i = 0; i < numSpilledBefore; i++) count_point(e, &c->spilled[i], countIteration);
	/* EXTENSION (trunk mask, see oracle_set_trunk_mask): an upper node named by the mask that is still a leaf splits in this
	 * round too (one above the limit has been listed by count_point already) */
	if (c->trunkMask[0] | c->trunkMask[1]) {
		for (uint32_t i = 0; i < e->stats->numNodes; i++) {
			SimlodNode* nd = &e->nodes[i];
			if (nd->level < 3 && nd->counter <= SIMLOD_MAX_POINTS_PER_NODE && node_is_leaf(nd) && trunk_forced(c, nd)) {
				if (c->numSpilling >= SPILLING_CAPACITY) { c->lastError = ORACLE_ERR_SPILLING; break; }
				c->spilling[c->numSpilling++] = nd;
			}
		}
	}
	/* move the stored points of every spilling node to the spill buffer, voxels.cu:253-289 */
	for (uint32_t s = 0; s < c->numSpilling; s++) {
		SimlodNode* node = c->spilling[s];
		SimlodChunk* chunk = node->points;
		for (uint32_t i = 0; i < node->numPoints; i++) {
			if (i > 0 && i % SIMLOD_POINTS_PER_CHUNK == 0) chunk = chunk->next;
			if (c->numSpilled >= SPILLED_CAPACITY) { c->lastError = ORACLE_ERR_SPILLED; break; }
			c->spilled[c->numSpilled++] = chunk->points[i % SIMLOD_POINTS_PER_CHUNK];
		}
	}
	for (uint32_t i = 0; i < e->stats->numNodes; i++) e->nodes[i].countIteration = countIteration; /* :298-300 */
	return c->numSpilling == 0;
}

/* doSplitting, voxels.cu:308-383 */
static void do_splitting(BuildEnv* e) {
	OracleCtx* c = e->c;
	for (uint32_t s = 0; s < c->numSpilling; s++) {
		SimlodNode* sp = c->spilling[s];
		if (e->stats->numNodes + 8 > c->maxNodes) { c->lastError = ORACLE_ERR_NODES; return; }
		if (sp->level >= SIMLOD_MAX_DEPTH) c->lastError = ORACLE_ERR_DEPTH; /* name[] overflow, SURVEY H2 */
		uint32_t childOffset = e->stats->numNodes;
		e->stats->numNodes += 8;
		for (int i = 0; i < 8; i++) {
			SimlodNode child;
			memset(&child, 0, sizeof(child));   /* Node's default member initialisers, structures.cuh:74-99 */
			child.name[0] = 'r';
			child.isLeaf = 1;
			child.level = sp->level + 1;
			child.X = 2 * sp->X + (uint32_t)((i >> 2) & 1);
			child.Y = 2 * sp->Y + (uint32_t)((i >> 1) & 1);
			child.Z = 2 * sp->Z + (uint32_t)(i & 1);
			memcpy(child.name, sp->name, 20);
			if (child.level < 20) child.name[child.level] = (uint8_t)(i + '0');
			e->nodes[childOffset + i] = child;
			sp->children[i] = &e->nodes[childOffset + i];
			c->pointTail[childOffset + i] = c->pointCur[childOffset + i] = NULL;
			c->voxelTail[childOffset + i] = c->voxelCur[childOffset + i] = NULL;
		}
		/* return the chunks to the pool: the pool is a stack whose top is stats->numAllocatedChunks, :346-357 */
		SimlodChunk* chunk = sp->points;
		while (chunk) {
			SimlodChunk* next = chunk->next;
			chunk->next = NULL;
			e->stats->numAllocatedChunks -= 1;
			c->chunkQueue[e->stats->numAllocatedChunks] = chunk;
			chunk = next;
		}
		sp->numPoints = 0;
		sp->points = NULL;
		c->pointTail[sp - e->nodes] = c->pointCur[sp - e->nodes] = NULL;
		if (!sp->grid) sp->grid = (SimlodOccupancyGrid*)persistent_alloc(e->alloc, sizeof(SimlodOccupancyGrid));
	}
	/* the clear loop of :375-382 zeroes the grid of EVERY spilling node — also one that already had a
	 * grid (only the root can: it owns one since reset and spills with numVoxels possibly > 0). */
	for (uint32_t s = 0; s < c->numSpilling; s++) memset(c->spilling[s]->grid->values, 0, sizeof(SimlodOccupancyGrid));
}

/* sampleVoxel, voxels.cu:50-121 */
static void sample_voxel(BuildEnv* e, SimlodNode* node, uint32_t pXf, uint32_t pYf, uint32_t pZf, const SimlodPoint* p) {
	if (!node->grid) return;
	uint32_t div = 1u << ((SIMLOD_MAX_DEPTH + 1) - node->level);
	uint32_t pX = (pXf / div) % SIMLOD_GRID_SIZE;
	uint32_t pY = (pYf / div) % SIMLOD_GRID_SIZE;
	uint32_t pZ = (pZf / div) % SIMLOD_GRID_SIZE;
	uint32_t voxelIndex = pX + pY * SIMLOD_GRID_SIZE + pZ * SIMLOD_GRID_SIZE * SIMLOD_GRID_SIZE;
	uint32_t word = voxelIndex / 32u, mask = 1u << (voxelIndex % 32u);
	if (node->grid->values[word] & mask) return;
	node->grid->values[word] |= mask;
	node->numVoxels += 1;
	float nodeSize = e->size / ldexpf(1.0f, (int)node->level);   /* pow(2.0f, level) is exact */
	float nminx = ((float)node->X + 0.0f) * nodeSize + e->minx;
	float nminy = ((float)node->Y + 0.0f) * nodeSize + e->miny;
	float nminz = ((float)node->Z + 0.0f) * nodeSize + e->minz;
	SimlodPoint v;
	v.x = nminx + nodeSize * ((float)pX + 0.5f) / (float)SIMLOD_GRID_SIZE;
	v.y = nminy + nodeSize * ((float)pY + 0.5f) / (float)SIMLOD_GRID_SIZE;
	v.z = nminz + nodeSize * ((float)pZ + 0.5f) / (float)SIMLOD_GRID_SIZE;
	v.color = p->color;
	if (e->c->numBacklog >= VOXEL_BACKLOG_CAPACITY) { e->c->lastError = ORACLE_ERR_BACKLOG; return; }
	e->c->backlogVoxels[e->c->numBacklog] = v;
	e->c->backlogTargets[e->c->numBacklog] = node;
	e->c->numBacklog++;
}

/* voxelSampling::traverse, voxels.cu:426-470 — sampleVoxel on EVERY node of the root-to-leaf path */
static void sample_path(BuildEnv* e, const SimlodPoint* p) {
	const float fGrid = 1048576.0f, fFull = 268435456.0f; /* 2^20, MAX_DEPTH_GRIDSIZE = 2^28 */
	uint32_t X = quantize(fGrid, p->x, e->minx, e->size);
	uint32_t Y = quantize(fGrid, p->y, e->miny, e->size);
	uint32_t Z = quantize(fGrid, p->z, e->minz, e->size);
	uint32_t pX = quantize(fFull, p->x, e->minx, e->size);
	uint32_t pY = quantize(fFull, p->y, e->miny, e->size);
	uint32_t pZ = quantize(fFull, p->z, e->minz, e->size);
	SimlodNode* cur = &e->nodes[0];
	for (int level = 0; level < SIMLOD_MAX_DEPTH; level++) {
		sample_voxel(e, cur, pX, pY, pZ, p);
		SimlodNode* ch = cur->children[child_index(X, Y, Z, level)];
		if (!ch) break;
		cur = ch;
	}
}

static SimlodChunk* take_chunk(BuildEnv* e) { /* voxels.cu:506-517 */
	uint64_t idx = e->stats->numAllocatedChunks++;
	SimlodChunk* chunk = (idx >= e->stats->chunkPoolSize)
		? (SimlodChunk*)persistent_alloc(e->alloc, sizeof(SimlodChunk))
		: e->c->chunkQueue[idx];
	chunk->next = NULL;
	return chunk;
}

/* allocatePointChunks, voxels.cu:485-538 */
static void allocate_point_chunks(BuildEnv* e) {
	for (uint32_t i = 0; i < e->stats->numNodes; i++) {
		SimlodNode* n = &e->nodes[i];
		if (!node_is_leaf(n) || !(n->numPoints < n->counter)) continue;
		int required = (int)((n->counter + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK);
		int existing = (int)((n->numPoints + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK);
		for (int k = existing; k < required; k++) { /* append behind the tail, :505-526 */
			SimlodChunk* chunk = take_chunk(e);
			SimlodChunk* prev = e->c->pointTail[i];
			if (!prev) n->points = chunk; else prev->next = chunk;
			e->c->pointTail[i] = chunk;
		}
	}
	if (e->stats->numAllocatedChunks > e->stats->chunkPoolSize) e->stats->chunkPoolSize = e->stats->numAllocatedChunks; /* :535-537 */
}

/* allocateVoxelChunks, voxels.cu:641-672 */
static void allocate_voxel_chunks(BuildEnv* e) {
	for (uint32_t i = 0; i < e->stats->numNodes; i++) {
		SimlodNode* n = &e->nodes[i];
		int required = (int)((n->numVoxels + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK);
		if (required == 0) continue;
		if (!n->voxelChunks) {
			n->voxelChunks = (SimlodChunk*)persistent_alloc(e->alloc, sizeof(SimlodChunk));
			n->voxelChunks->next = NULL;
			e->c->voxelTail[i] = n->voxelChunks;
		}
		/* number of chunks already linked == ceil(numVoxelsStored / 1000), at least 1 */
		int have = (int)((n->numVoxelsStored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK);
		if (have < 1) have = 1;
		for (int k = have; k < required; k++) {
			SimlodChunk* nc = (SimlodChunk*)persistent_alloc(e->alloc, sizeof(SimlodChunk));
			nc->next = NULL;
			e->c->voxelTail[i]->next = nc;
			e->c->voxelTail[i] = nc;
		}
	}
}

/* insertPoint, voxels.cu:553-613.  The reference finds chunk #(slot/1000) by walking from the head; slots
 * of one node arrive in increasing order here, so a per-node cursor reaches the same chunk in O(1). */
static void insert_point(BuildEnv* e, const SimlodPoint* p) {
	SimlodNode* leaf = descend(e, p);
	size_t li = (size_t)(leaf - e->nodes);
	uint32_t slot = leaf->numPoints++;
	if (!leaf->points) { e->c->lastError = ORACLE_ERR_NULL_CHUNK; return; }
	SimlodChunk* chunk = e->c->pointCur[li];
	if (slot == 0) chunk = leaf->points;
	else if (slot % SIMLOD_POINTS_PER_CHUNK == 0) chunk = chunk ? chunk->next : NULL;
	if (!chunk) { e->c->lastError = ORACLE_ERR_NULL_CHUNK; return; }
	e->c->pointCur[li] = chunk;
	chunk->points[slot % SIMLOD_POINTS_PER_CHUNK] = *p;
}

/* addBatch, voxels.cu:700-802 */
static void add_batch(BuildEnv* e, const SimlodPoint* pts, uint32_t n, uint32_t batchIndex) {
	OracleCtx* c = e->c;
	c->numSpilled = 0;
	c->numBacklog = 0;
	/* expand, voxels.cu:385-415 */
	for (int round = 0; round < SIMLOD_MAX_EXPAND_ROUNDS; round++) {
		if (do_counting(e, pts, n, batchIndex + 1)) break;
		do_splitting(e);
		if (c->lastError == ORACLE_ERR_NODES) return;
	}
	/* voxelSampling, :417-483 */
	for (uint32_t i = 0; i < n; i++) sample_path(e, &pts[i]);
	for (uint32_t i = 0; i < c->numSpilled; i++) sample_path(e, &c->spilled[i]);
	allocate_point_chunks(e);
	allocate_voxel_chunks(e);
	/* insertPoints, :540-639 */
	for (uint32_t i = 0; i < n; i++) insert_point(e, &pts[i]);
	if (c->numSpilled > SPILLED_INSERT_LIMIT + 1u) c->lastError = ORACLE_ERR_SPILLED; /* reference drops them, :628 */
	for (uint32_t i = 0; i < c->numSpilled && i <= SPILLED_INSERT_LIMIT; i++) insert_point(e, &c->spilled[i]);
	/* insertVoxels, :674-698 */
	for (uint32_t i = 0; i < c->numBacklog; i++) {
		SimlodNode* t = c->backlogTargets[i];
		size_t ti = (size_t)(t - e->nodes);
		uint32_t slot = t->numVoxelsStored++;
		SimlodChunk* chunk = c->voxelCur[ti];   /* same O(1) cursor as insert_point */
		if (slot == 0) chunk = t->voxelChunks;
		else if (slot % SIMLOD_POINTS_PER_CHUNK == 0) chunk = chunk->next;
		c->voxelCur[ti] = chunk;
		chunk->points[slot % SIMLOD_POINTS_PER_CHUNK] = c->backlogVoxels[i];
	}
}

/* kernel_construct, voxels.cu:804-1010.  `momentaryBytes` is reported as Stats.allocatedBytes_momentary
 * (the reference's momentary bump allocator always ends at 408 800 192 B). */
void oracle_construct(OracleCtx* c, const SimlodUniforms* u, const SimlodPoint* ring, uint8_t* persistent,
                      SimlodNode* nodes, SimlodStats* stats, const uint32_t* numBatchesUploaded,
                      const uint32_t* batchSizes) {
	BuildEnv e;
	e.c = c; e.nodes = nodes; e.stats = stats; e.alloc = (SimlodAllocatorGlobal*)persistent;
	e.size = octree_size(u);
	e.minx = u->boxMin.x; e.miny = u->boxMin.y; e.minz = u->boxMin.z;
	uint32_t uploaded = *numBatchesUploaded;
	uint32_t numBatches = uploaded - stats->batchletIndex;
	if (numBatches > SIMLOD_MAX_BATCHES_PER_LAUNCH) numBatches = SIMLOD_MAX_BATCHES_PER_LAUNCH;
	uint32_t first = stats->batchletIndex, last = first + numBatches;
	for (uint32_t b = first; b < last; b++) {
		uint32_t slot = b % SIMLOD_BATCH_STREAM_SIZE;
		uint32_t batchSize = batchSizes[slot];
		const SimlodPoint* pts = ring + (size_t)slot * SIMLOD_MAX_BATCH_SIZE;
		int full = e.alloc->offset + SIMLOD_MEM_SAFETY_MARGIN >= u->persistentBufferCapacity; /* :896-912 */
		stats->memCapacityReached = (uint8_t)(full ? 1 : 0);
		if (full) break;
		add_batch(&e, pts, batchSize, stats->batchletIndex);
		stats->batchletIndex += 1;
		stats->numPointsProcessed += batchSize;
	}
	/* stats pass, :957-1009 */
	uint32_t inner = 0, leaves = 0, nonempty = 0, points = 0, voxels = 0, cp = 0, cv = 0;
	for (uint32_t i = 0; i < stats->numNodes; i++) {
		const SimlodNode* n = &nodes[i];
		if (node_is_leaf(n)) {
			leaves++; points += n->numPoints;
			cp += (n->numPoints + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
			if (n->numPoints > 0) nonempty++;
		} else {
			inner++; voxels += n->numVoxels;
			cv += (n->numVoxels + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
		}
	}
	stats->numInner = inner; stats->numLeaves = leaves; stats->numNonemptyLeaves = nonempty;
	stats->numPoints = points; stats->numVoxels = voxels;
	stats->numChunksPoints = cp; stats->numChunksVoxels = cv;
	stats->allocatedBytes_momentary = 408800192ull;
	stats->allocatedBytes_persistent = e.alloc->offset;
	stats->frameID = (uint32_t)u->frameCounter;
}

/* ---- rendering ------------------------------------------------------------------------------------------ */
static float dot4(const simlod_float4 r, float x, float y, float z, float w) { /* helper_math.h:1266-1269 */
	return r.x * x + r.y * y + r.z * z + r.w * w;
}

typedef struct { float x, y, z, w; } vec4;

static vec4 xform(const SimlodMat4* m, float x, float y, float z) { /* structures.cuh:53-60 */
	vec4 r;
	r.x = dot4(m->rows[0], x, y, z, 1.0f);
	r.y = dot4(m->rows[1], x, y, z, 1.0f);
	r.z = dot4(m->rows[2], x, y, z, 1.0f);
	r.w = dot4(m->rows[3], x, y, z, 1.0f);
	return r;
}

static int to_int_trunc(double v) { /* double -> int; out of range behaves like x86 cvttsd2si (INT_MIN) */
	if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
	return (int)v;
}

/* pixel of a sample, render.cu:62-70; returns 0 when rejected */
static int project(const SimlodUniforms* u, const SimlodPoint* p, int* px, int* py, float* depth) {
	vec4 ndc = xform(&u->transform, p->x, p->y, p->z);
	*depth = ndc.w;
	float nx = ndc.x / ndc.w, ny = ndc.y / ndc.w;
	int x = to_int_trunc(((double)nx * 0.5 + 0.5) * (double)u->width);
	int y = to_int_trunc(((double)ny * 0.5 + 0.5) * (double)u->height);
	if (!(x > 1 && (double)x < (double)u->width - 2.0)) return 0;
	if (!(y > 1 && (double)y < (double)u->height - 2.0)) return 0;
	*px = x; *py = y;
	return 1;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static uint32_t lod_color(int level) { /* render.cu:38-59 */
	static const uint32_t SPECTRAL[8] = {0x4f3ed5, 0x436df4, 0x61aefd, 0x8be0fe, 0x98f5e6, 0xa4ddab, 0xa5c266, 0xbd8832};
	int index = (int)((float)(8 - level) * 1.8f);
	return SPECTRAL[clampi(index, 0, 7)];
}

static uint64_t node_id(const SimlodNode* n) { /* structures.cuh:118-141, including its int-width shifts */
	uint64_t id = 0;
	id |= (uint64_t)(int64_t)(n->name[0] == 'r' ? 1 : 0);
	for (int i = 1; i <= 9; i++) id |= (uint64_t)(int64_t)(int32_t)((uint32_t)(n->name[i] - '0') << (3 * i));
	for (int i = 10; i <= 17; i++) id |= ((uint64_t)(int64_t)(n->name[i] - '0')) << (3 * i);
	id |= ((uint64_t)(int64_t)(n->name[18] - '0')) << 53;
	return id;
}

static uint32_t sample_color(const SimlodUniforms* u, const SimlodNode* node, uint32_t color, int hqs) {
	if (u->colorByNode) {
		/* render.cu:75 multiplies by 123456789ull, :460 by the int 123456789: same low 32 bits */
		(void)hqs;
		return (uint32_t)((node_id(node) % 127ull) * 123456789ull);
	}
	if (u->colorByLOD) return lod_color((int)node->level);
	return color;
}

typedef void (*sample_fn)(void* env, const SimlodNode* node, const SimlodPoint* p);

static void for_each_sample(const SimlodNode* node, const SimlodChunk* chunk, uint32_t count, sample_fn fn, void* env) {
	for (uint32_t i = 0; i < count; i++) { /* drawNode, render.cu:106-159 */
		if (i > 0 && i % SIMLOD_POINTS_PER_CHUNK == 0) chunk = chunk->next;
		fn(env, node, &chunk->points[i % SIMLOD_POINTS_PER_CHUNK]);
	}
}

typedef struct RenderEnv {
	const SimlodUniforms* u;
	uint64_t* fb;
	uint32_t* fbDepth;
	uint32_t* fbColor;
	int W, H;
} RenderEnv;

static void draw_point(void* env, const SimlodNode* node, const SimlodPoint* p) { /* drawPoint, render.cu:61-104 */
	RenderEnv* r = (RenderEnv*)env;
	int x, y; float depth;
	if (!project(r->u, p, &x, &y, &depth)) return;
	uint32_t color = sample_color(r->u, node, p->color, 0);
	uint32_t dbits; memcpy(&dbits, &depth, 4);
	uint64_t encoded = ((uint64_t)dbits << 32) | color;
	for (int ox = 0; ox < r->u->pointSize; ox++)
	for (int oy = 0; oy < r->u->pointSize; oy++) {
		uint32_t pxl = (uint32_t)clampi(x + ox, 0, r->W) + (uint32_t)r->W * (uint32_t)clampi(y + oy, 0, r->H);
		if (pxl >= (uint32_t)(r->W * r->H)) continue; /* out of the image for pointSize >= 4: UB in the reference */
		if (encoded < r->fb[pxl]) r->fb[pxl] = encoded;
	}
}

static void hqs_depth(void* env, const SimlodNode* node, const SimlodPoint* p) { /* render.cu:286-311, 362-388 */
	RenderEnv* r = (RenderEnv*)env; (void)node;
	int x, y; float depth;
	if (!project(r->u, p, &x, &y, &depth)) return;
	if (!(depth > 0.0f)) return;
	uint32_t dbits; memcpy(&dbits, &depth, 4);
	for (int ox = 0; ox < r->u->pointSize; ox++)
	for (int oy = 0; oy < r->u->pointSize; oy++) {
		uint32_t pxl = (uint32_t)clampi(x + ox, 0, r->W) + (uint32_t)r->W * (uint32_t)clampi(y + oy, 0, r->H);
		if (pxl >= (uint32_t)(r->W * r->H)) continue;
		if (dbits < r->fbDepth[pxl]) r->fbDepth[pxl] = dbits;
	}
}

static void hqs_color(void* env, const SimlodNode* node, const SimlodPoint* p) { /* render.cu:447-496, 549-599 */
	RenderEnv* r = (RenderEnv*)env;
	int x, y; float depth;
	if (!project(r->u, p, &x, &y, &depth)) return;
	if (!(depth > 0.0f)) return;
	uint32_t color = sample_color(r->u, node, p->color, 1);
	for (int ox = 0; ox < r->u->pointSize; ox++)
	for (int oy = 0; oy < r->u->pointSize; oy++) {
		uint32_t pxl = (uint32_t)clampi(x + ox, 0, r->W) + (uint32_t)r->W * (uint32_t)clampi(y + oy, 0, r->H);
		if (pxl >= (uint32_t)(r->W * r->H)) continue;
		float fbDepth; memcpy(&fbDepth, &r->fbDepth[pxl], 4);
		if (depth < fbDepth * 1.01f) {
			r->fbColor[4 * pxl + 0] += (color >> 0) & 0xff;
			r->fbColor[4 * pxl + 1] += (color >> 8) & 0xff;
			r->fbColor[4 * pxl + 2] += (color >> 16) & 0xff;
			r->fbColor[4 * pxl + 3] += 1;
		}
	}
}

/* intersectsFrustum, math.cuh:154-201 (+ createPlane :55-64); m = transform_updateBound */
static int intersects_frustum(const SimlodMat4* m, const float mn[3], const float mx[3]) {
	/* values[transposeIndex(k)]: element (row = k%4, col = k/4) of the row-major matrix */
	const simlod_float4* R = m->rows;
	float m0 = R[0].x, m1 = R[1].x, m2 = R[2].x, m3 = R[3].x;
	float m4 = R[0].y, m5 = R[1].y, m6 = R[2].y, m7 = R[3].y;
	float m8 = R[0].z, m9 = R[1].z, m10 = R[2].z, m11 = R[3].z;
	float m12 = R[0].w, m13 = R[1].w, m14 = R[2].w, m15 = R[3].w;
	float P[6][4] = {
		{m3 - m0, m7 - m4, m11 - m8, m15 - m12}, {m3 + m0, m7 + m4, m11 + m8, m15 + m12},
		{m3 + m1, m7 + m5, m11 + m9, m15 + m13}, {m3 - m1, m7 - m5, m11 - m9, m15 - m13},
		{m3 - m2, m7 - m6, m11 - m10, m15 - m14}, {m3 + m2, m7 + m6, m11 + m10, m15 + m14}};
	for (int i = 0; i < 6; i++) {
		float x = P[i][0], y = P[i][1], z = P[i][2], w = P[i][3];
		float len = sqrtf(x * x + y * y + z * z);
		float nx = x / len, ny = y / len, nz = z / len, c = w / len;
		float vx = (double)nx > 0.0 ? mx[0] : mn[0];
		float vy = (double)ny > 0.0 ? mx[1] : mn[1];
		float vz = (double)nz > 0.0 ? mx[2] : mn[2];
		float d = (nx * vx + ny * vy + nz * vz) + c;
		if (d < 0) return 0;
	}
	return 1;
}

static float min8(const float f[8]) { /* render.cu:705-716 */
	float m0 = fminf(f[0], f[1]), m1 = fminf(f[2], f[3]), m2 = fminf(f[4], f[5]), m3 = fminf(f[6], f[7]);
	return fminf(fminf(m0, m1), fminf(m2, m3));
}
static float max8(const float f[8]) { /* render.cu:718-729 */
	float m0 = fmaxf(f[0], f[1]), m1 = fmaxf(f[2], f[3]), m2 = fmaxf(f[4], f[5]), m3 = fmaxf(f[6], f[7]);
	return fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
}

/* ---- debug lines: node boxes + view frustum (only when Uniforms.showBoundingBox), render.cu:637-688, 1197-1233,
 *      rasterization.cuh:5-47 (drawLine, drawBoundingBox) and :90-183 (rasterizeLines), math.cuh:22-152 (Frustum) ---- */
typedef struct { float nx, ny, nz, c; } OPlane;
typedef struct { float x, y, z; uint32_t color; } OVertex;

static OPlane make_plane(float x, float y, float z, float w) { /* createPlane, math.cuh:55-64 */
	float len = sqrtf(x * x + y * y + z * z);
	OPlane p = {x / len, y / len, z / len, w / len};
	return p;
}

static void frustum_planes(const SimlodMat4* m, OPlane P[6]) { /* Frustum::fromWorldViewProj, math.cuh:69-108 */
	const simlod_float4* R = m->rows;
	float m0 = R[0].x, m1 = R[1].x, m2 = R[2].x, m3 = R[3].x, m4 = R[0].y, m5 = R[1].y, m6 = R[2].y, m7 = R[3].y;
	float m8 = R[0].z, m9 = R[1].z, m10 = R[2].z, m11 = R[3].z, m12 = R[0].w, m13 = R[1].w, m14 = R[2].w, m15 = R[3].w;
	P[0] = make_plane(m3 - m0, m7 - m4, m11 - m8, m15 - m12);
	P[1] = make_plane(m3 + m0, m7 + m4, m11 + m8, m15 + m12);
	P[2] = make_plane(m3 + m1, m7 + m5, m11 + m9, m15 + m13);
	P[3] = make_plane(m3 - m1, m7 - m5, m11 - m9, m15 - m13);
	P[4] = make_plane(m3 - m2, m7 - m6, m11 - m10, m15 - m14);
	P[5] = make_plane(m3 + m2, m7 + m6, m11 + m10, m15 + m14);
}

static float plane_dist(const OPlane* p, float x, float y, float z) { return (p->nx * x + p->ny * y + p->nz * z) + p->c; } /* math.cuh:22-25 */

static int frustum_contains(const OPlane P[6], float x, float y, float z) { /* math.cuh:138-151 */
	for (int i = 0; i < 6; i++) if (plane_dist(&P[i], x, y, z) < 0) return 0;
	return 1;
}

static float dist_to_plane(float ox, float oy, float oz, float dx, float dy, float dz, const OPlane* p) { /* math.cuh:27-53 */
	const float INF = 1.0f / 0.0f;
	float denom = p->nx * dx + p->ny * dy + p->nz * dz;
	if (denom < 0.0f) return INF;
	if (denom == 0.0f) return plane_dist(p, ox, oy, oz) == 0.0f ? 0.0f : INF;
	float t = -((ox * p->nx + oy * p->ny + oz * p->nz) + p->c) / denom;
	return ((double)t >= 0.0) ? t : INF;
}

static void frustum_intersect_ray(const OPlane P[6], float ox, float oy, float oz, float dx, float dy, float dz, float out[3]) { /* math.cuh:110-136 */
	const float INF = 1.0f / 0.0f;
	float farthest = -INF;
	for (int i = 0; i < 6; i++) {
		float d = dist_to_plane(ox, oy, oz, dx, dy, dz, &P[i]);
		if (d > 0 && d != INF) farthest = fmaxf(farthest, d);
	}
	out[0] = ox + dx * farthest; out[1] = oy + dy * farthest; out[2] = oz + dz * farthest;
}

static void emit_line(OVertex* v, uint32_t* count, uint32_t cap, const float a[3], const float b[3], uint32_t color) { /* drawLine */
	if (*count + 2 > cap) return;
	v[*count].x = a[0]; v[*count].y = a[1]; v[*count].z = a[2]; v[*count].color = color;
	v[*count + 1].x = b[0]; v[*count + 1].y = b[1]; v[*count + 1].z = b[2]; v[*count + 1].color = color;
	*count += 2;
}

static void emit_box(OVertex* v, uint32_t* count, uint32_t cap, const float pos[3], const float size[3], uint32_t color) { /* drawBoundingBox */
	float mn[3], mx[3];
	for (int k = 0; k < 3; k++) { mn[k] = pos[k] - size[k] / 2.0f; mx[k] = pos[k] + size[k] / 2.0f; }
	const int E[12][6] = { /* corner selectors (0 = min, 1 = max) in the order of rasterization.cuh:30-46 */
		{0,0,0, 1,0,0}, {1,0,0, 1,1,0}, {1,1,0, 0,1,0}, {0,1,0, 0,0,0},
		{0,0,1, 1,0,1}, {1,0,1, 1,1,1}, {1,1,1, 0,1,1}, {0,1,1, 0,0,1},
		{1,0,0, 1,0,1}, {1,1,0, 1,1,1}, {0,1,0, 0,1,1}, {0,0,0, 0,0,1}};
	for (int e = 0; e < 12; e++) {
		float a[3], b[3];
		for (int k = 0; k < 3; k++) { a[k] = E[e][k] ? mx[k] : mn[k]; b[k] = E[e][3 + k] ? mx[k] : mn[k]; }
		emit_line(v, count, cap, a, b, color);
	}
}

static void rasterize_line(const SimlodUniforms* u, const OPlane P[6], OVertex s, OVertex e, uint64_t* fb, int width, int height) { /* rasterization.cuh:98-180 */
	float dx = e.x - s.x, dy = e.y - s.y, dz = e.z - s.z;
	float inv = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);   /* normalize(): v * rsqrtf(dot(v, v)), helper_math.h:1323-1327 */
	dx = dx * inv; dy = dy * inv; dz = dz * inv;
	if (!frustum_contains(P, s.x, s.y, s.z)) { float I[3]; frustum_intersect_ray(P, s.x, s.y, s.z, dx, dy, dz, I); s.x = I[0]; s.y = I[1]; s.z = I[2]; }
	if (!frustum_contains(P, e.x, e.y, e.z)) { float I[3]; frustum_intersect_ray(P, e.x, e.y, e.z, dx * -1.0f, dy * -1.0f, dz * -1.0f, I); e.x = I[0]; e.y = I[1]; e.z = I[2]; }
	vec4 a = xform(&u->transform, s.x, s.y, s.z), b = xform(&u->transform, e.x, e.y, e.z);
	a.x = a.x / a.w; a.y = a.y / a.w; a.z = a.z / a.w;
	b.x = b.x / b.w; b.y = b.y / b.w; b.z = b.z / b.w;
	float sx0 = (a.x * 0.5f + 0.5f) * (float)width, sy0 = (a.y * 0.5f + 0.5f) * (float)height;
	float sx1 = (b.x * 0.5f + 0.5f) * (float)width, sy1 = (b.y * 0.5f + 0.5f) * (float)height;
	float ddx = sx1 - sx0, ddy = sy1 - sy0, ddz = 1.0f - 1.0f;
	float steps = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
	steps = fmaxf(0.0f, fminf(steps, 400.0f));
	float stepSize = (float)(1.0 / (double)steps);
	for (float t = 0; (double)t <= 1.0; t += stepSize) {
		float nx = (float)((1.0 - (double)t) * (double)a.x + (double)(t * b.x));
		float ny = (float)((1.0 - (double)t) * (double)a.y + (double)(t * b.y));
		float depth = (float)((1.0 - (double)t) * (double)a.w + (double)(t * b.w));
		if ((double)nx < -1.0 || (double)nx > 1.0) continue;
		if ((double)ny < -1.0 || (double)ny > 1.0) continue;
		int x = to_int_trunc(((double)nx * 0.5 + 0.5) * (double)width);
		int y = to_int_trunc(((double)ny * 0.5 + 0.5) * (double)height);
		x = clampi(x, 0, width - 1); y = clampi(y, 0, height - 1);
		uint32_t dbits; memcpy(&dbits, &depth, 4);
		uint64_t enc = ((uint64_t)dbits << 32) | s.color;
		if (enc < fb[x + width * y]) fb[x + width * y] = enc;
	}
}

static void draw_debug_lines(const SimlodUniforms* u, const SimlodNode* visible, uint32_t numVisible, float cubeSize, const float cmin[3],
                             uint64_t* fb, int W, int H) {
	const uint32_t cap = 1000000u; /* render.cu:1119 */
	OVertex* v = (OVertex*)malloc((size_t)cap * sizeof(OVertex));
	uint32_t count = 0;
	{ /* view frustum, render.cu:1197-1223 */
		const float fend = 0.99995f;
		const float C[8][2][3] = {{{1, 1, -1}, {1, 1, fend}}, {{1, -1, -1}, {1, -1, fend}}, {{-1, 1, -1}, {-1, 1, fend}}, {{-1, -1, -1}, {-1, -1, fend}},
		                          {{-1, -1, fend}, {1, -1, fend}}, {{-1, 1, fend}, {1, 1, fend}}, {{-1, -1, fend}, {-1, 1, fend}}, {{1, -1, fend}, {1, 1, fend}}};
		for (int l = 0; l < 8; l++) {
			float p[2][3];
			for (int k = 0; k < 2; k++) {
				vec4 q = xform(&u->transformInv_updateBound, C[l][k][0], C[l][k][1], C[l][k][2]);
				p[k][0] = q.x / q.w; p[k][1] = q.y / q.w; p[k][2] = q.z / q.w;
			}
			emit_line(v, &count, cap, p[0], p[1], 0x000000ffu);
		}
	}
	for (uint32_t i = 0; i < numVisible; i++) { /* drawNodesBoundingBoxes, render.cu:637-688: four coincident boxes per node */
		const SimlodNode* n = &visible[i];
		if (n->numPoints == 0 && n->numVoxels == 0) continue;
		float scale = cubeSize / ldexpf(1.0f, (int)n->level);
		float pos[3] = {cmin[0] + ((float)n->X + 0.5f) * scale, cmin[1] + ((float)n->Y + 0.5f) * scale, cmin[2] + ((float)n->Z + 0.5f) * scale};
		float size[3] = {scale, scale, scale};
		for (int r = 0; r < 4; r++) emit_box(v, &count, cap, pos, size, 0x0000ff00u);
	}
	OPlane P[6];
	frustum_planes(&u->transform, P);
	for (uint32_t l = 0; l + 1 < count; l += 2) rasterize_line(u, P, v[l], v[l + 1], fb, W, H);
	free(v);
}

/* test hook: rasterise an explicit vertex list (pairs) into fb */
void oracle_rasterize_lines(const SimlodUniforms* u, const OVertex* v, uint32_t count, uint64_t* fb) {
	OPlane P[6];
	frustum_planes(&u->transform, P);
	for (uint32_t l = 0; l + 1 < count; l += 2) rasterize_line(u, P, v[l], v[l + 1], fb, (int)u->width, (int)u->height);
}

/* kernel_render, render.cu:1084-1355.
 *   fb         out, W*H uint64: the pre-EDL framebuffer (depth bits << 32 | colour)
 *   colorOut   out, W*H uint32 or NULL: what surf2Dwrite would have stored (after EDL when edl != 0)
 *   visible    scratch/out, capacity SIMLOD_MAX_VISIBLE_NODES Node copies (render.cu:1108)
 *   edl        0: skip the EDL pass (matches oracle/_ref), 1: EDL over every full 16x16 tile
 */
/* kernel_render in the four parts of simlod_launch_render_part (include/simlod_hip.h): between the parts a multi-GPU frame reduces
 * depthPlane (MIN), sumPlanes (SUM) and fb (MIN) over the ranks.  depthPlane: W*H uint32, sumPlanes: W*H x {R,G,B,count} uint32. */
void oracle_render_part(OracleCtx* c, const SimlodUniforms* u, SimlodNode* nodes, SimlodStats* stats,
                        uint64_t* fb, uint32_t* colorOut, SimlodNode* visible, int edl, int part, uint32_t* depthPlane, uint32_t* sumPlanes) {
	int W = (int)u->width, H = (int)u->height;
	size_t numPixels = (size_t)W * (size_t)H;
	float cubeSize = octree_size(u);
	float cmin[3] = {u->boxMin.x, u->boxMin.y, u->boxMin.z};
	RenderEnv r; r.u = u; r.fb = fb; r.W = W; r.H = H; r.fbDepth = depthPlane; r.fbColor = sumPlanes;
	const int hqs = u->useHighQualityShading != 0;
	if (part == 0) {
		for (size_t i = 0; i < numPixels; i++) fb[i] = SIMLOD_CLEAR_PIXEL; /* :1126-1131 */
		/* compute_visibility_disjunct pass 1, render.cu:762-901 */
		for (uint32_t i = 0; i < stats->numNodes; i++) {
			SimlodNode* n = &nodes[i];
			float nodeSize = cubeSize / ldexpf(1.0f, (int)n->level);
			float mn[3], mx[3];
			uint32_t XYZ[3] = {n->X, n->Y, n->Z};
			for (int a = 0; a < 3; a++) {
				mn[a] = cmin[a] + ((float)XYZ[a] + 0.0f) * nodeSize;
				mx[a] = cmin[a] + ((float)XYZ[a] + 1.0f) * nodeSize;
			}
			float sx[8], sy[8];
			for (int k = 0; k < 8; k++) { /* order p000,p001,p010,p011,p100,p101,p110,p111 (:783-790) */
				float x = (k & 4) ? mx[0] : mn[0], y = (k & 2) ? mx[1] : mn[1], z = (k & 1) ? mx[2] : mn[2];
				vec4 ndc = xform(&u->transform_updateBound, x, y, z);
				sx[k] = ((ndc.x / ndc.w) * 0.5f + 0.5f) * u->width;
				sy[k] = ((ndc.y / ndc.w) * 0.5f + 0.5f) * u->height;
			}
			float dx = max8(sx) - min8(sx), dy = max8(sy) - min8(sy);
			int vis = intersects_frustum(&u->transform_updateBound, mn, mx) && (n->numPoints > 0 || n->numVoxels > 0);
			n->visible = (uint8_t)vis;
			n->isLarge = (uint8_t)(((double)dx > 2.0 * (double)u->minNodeSize) || ((double)dy > 2.0 * (double)u->minNodeSize));
		}
		/* pass 2, render.cu:906-933 + makeVisible :746-756 */
		uint32_t numVisible = 0, visPoints = 0, visVoxels = 0, visInner = 0, visLeaves = 0;
		for (uint32_t i = 0; i < stats->numNodes; i++) {
			SimlodNode* n = &nodes[i];
			SimlodNode* emit[8]; int ne = 0;
			if (n->isLarge && !node_is_leaf(n)) {
				for (int k = 0; k < 8; k++) {
					SimlodNode* ch = n->children[k];
					if (ch && !ch->isLarge && ch->visible) emit[ne++] = ch;
				}
			} else if (n->isLarge && node_is_leaf(n) && n->visible) {
				emit[ne++] = n;
			}
			for (int k = 0; k < ne; k++) {
				if (numVisible >= SIMLOD_MAX_VISIBLE_NODES) { if (c) c->lastError = ORACLE_ERR_VISIBLE; break; }
				visible[numVisible++] = *emit[k];
				if (emit[k]->numPoints > 0) { visLeaves++; visPoints += emit[k]->numPoints; }
				else if (emit[k]->numVoxels > 0) { visInner++; visVoxels += emit[k]->numVoxels; }
			}
		}
		stats->numVisibleNodes = numVisible; stats->numVisibleInner = visInner; stats->numVisibleLeaves = visLeaves;
		stats->numVisiblePoints = visPoints; stats->numVisibleVoxels = visVoxels;
		if (u->showPoints && !hqs) { /* drawNodes, render.cu:161-210 */
			for (uint32_t i = 0; i < numVisible; i++) {
				for_each_sample(&visible[i], visible[i].points, visible[i].numPoints, draw_point, &r);
				for_each_sample(&visible[i], visible[i].voxelChunks, visible[i].numVoxels, draw_point, &r);
			}
		}
		if (hqs) {
			for (size_t i = 0; i < numPixels; i++) { depthPlane[i] = 0x7f800000u; sumPlanes[4 * i] = sumPlanes[4 * i + 1] = sumPlanes[4 * i + 2] = sumPlanes[4 * i + 3] = 0; }
			if (u->showPoints) for (uint32_t i = 0; i < numVisible; i++) { /* drawNodesHQS depth pass, render.cu:212-388 */
				for_each_sample(&visible[i], visible[i].points, visible[i].numPoints, hqs_depth, &r);
				if (visible[i].numVoxels > 0) for_each_sample(&visible[i], visible[i].voxelChunks, visible[i].numVoxels, hqs_depth, &r);
			}
		} else if (u->showBoundingBox) draw_debug_lines(u, visible, numVisible, cubeSize, cmin, fb, W, H);   /* render.cu:1197-1235 */
		return;
	}
	const uint32_t numVisible = stats->numVisibleNodes;
	if (part == 1) {
		if (hqs && u->showPoints) for (uint32_t i = 0; i < numVisible; i++) { /* colour pass, render.cu:447-599 */
			for_each_sample(&visible[i], visible[i].points, visible[i].numPoints, hqs_color, &r);
			if (visible[i].numVoxels > 0) for_each_sample(&visible[i], visible[i].voxelChunks, visible[i].numVoxels, hqs_color, &r);
		}
		return;
	}
	if (part == 2) {
		if (!hqs) return;
		if (u->showPoints) for (size_t i = 0; i < numPixels; i++) { /* resolve, :607-632 */
			uint32_t C = r.fbColor[4 * i + 3];
			if (C == 0) continue;
			uint32_t color = ((r.fbColor[4 * i + 0] / C) & 0xff) | (((r.fbColor[4 * i + 1] / C) & 0xff) << 8)
			               | (((r.fbColor[4 * i + 2] / C) & 0xff) << 16) | (255u << 24);
			fb[i] = ((uint64_t)r.fbDepth[i] << 32) | color;
		}
		if (u->showBoundingBox) draw_debug_lines(u, visible, numVisible, cubeSize, cmin, fb, W, H);   /* render.cu:1197-1235 */
		return;
	}
	stats->frameID = (uint32_t)u->frameCounter;   /* stats, render.cu:1244-1252 (the visible counters were stored by part 0) */
	(void)c;
	if (!colorOut) return;
	for (size_t i = 0; i < numPixels; i++) colorOut[i] = (uint32_t)(fb[i] & 0xffffffffull);
	if (edl) { /* render.cu:1255-1325 on every full tile; reads depth from fb (unchanged by the pass) */
		const float PI = 3.1415f;
		const float us[4] = {0.0f, PI / 2.0f, PI, 3.0f * PI / 2.0f};
		int tilesX = W / 16, tilesY = H / 16;
		for (int ty = 0; ty < tilesY; ty++) for (int tx = 0; tx < tilesX; tx++)
		for (int t = 0; t < 256; t++) {
			int pixelID = tx * 16 + ty * W * 16 + (t % 16) + (t / 16) * W;
			uint32_t dbits = (uint32_t)(fb[pixelID] >> 32); float pdepth; memcpy(&pdepth, &dbits, 4);
			float sum = 0.0f;
			for (int k = 0; k < 4; k++) {
				int dx = (int)(1.5f * sinf(us[k])), dy = (int)(1.5f * cosf(us[k]));
				int index = pixelID + dx + W * dy;
				if (index < 0) index = 0;
				if (index > W * H - 1) index = W * H - 1;
				uint32_t nb = (uint32_t)(fb[index] >> 32); float ndepth; memcpy(&ndepth, &nb, 4);
				double diff = (double)(log2f(pdepth) - log2f(ndepth));
				double mxv = (diff > 0.0) ? diff : 0.0; /* fmax(NaN, 0) = 0 */
				sum = (float)((double)sum + mxv);
			}
			float response = sum / 50.0f;
			float shade = expf((float)((double)(-response) * 300.0 * (double)0.4f));
			uint32_t color = (uint32_t)(fb[pixelID] & 0xffffffffull);
			uint32_t R = (uint32_t)(shade * (float)((color >> 0) & 0xff));
			uint32_t G = (uint32_t)(shade * (float)((color >> 8) & 0xff));
			uint32_t B = (uint32_t)(shade * (float)((color >> 16) & 0xff));
			colorOut[pixelID] = R | (G << 8) | (B << 16) | (255u << 24);
		}
	}
}

void oracle_render(OracleCtx* c, const SimlodUniforms* u, SimlodNode* nodes, SimlodStats* stats,
                   uint64_t* fb, uint32_t* colorOut, SimlodNode* visible, int edl) {
	size_t numPixels = (size_t)(int)u->width * (size_t)(int)u->height;
	uint32_t* depth = u->useHighQualityShading ? (uint32_t*)malloc(numPixels * 4) : NULL;
	uint32_t* sums = u->useHighQualityShading ? (uint32_t*)malloc(numPixels * 16) : NULL;
	for (int part = 0; part < 4; part++) oracle_render_part(c, u, nodes, stats, fb, colorOut, visible, edl, part, depth, sums);
	free(depth); free(sums);
}
