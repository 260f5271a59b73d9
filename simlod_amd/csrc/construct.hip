// construct.hip — incremental octree/LOD builder for MI355X (gfx950): `kernel_construct`.  ONE builder with two granularities:
// EXACT (default: a group is one ring batch, the reference's granularity — every Node and Stats field after every batch is the
// reference's) and COALESCED (simlod_context_set_ingest_mode(1): a group is up to 20 pending batches; same octree content, fewer
// intermediate chunks).  The kernels are the same; the ones on the sample stream are compiled twice (SINGLE = one batch per group).
//
// Replaces modules/progressive_octree/progressive_octree_voxels.cu:804-1010 (one persistent cooperative CUDA kernel with ~40
// grid.sync() per batch) behind the same argument list and the same Node/Chunk/OccupancyGrid memory image.  Design:
//
//   * per group a CHAIN of ordinary launches — k_count, k_queue, k_hist, k_expand | k_insert, k_voxelize — because a dependent kernel
//     boundary costs ~3.7 us on this chip while a software grid barrier over 256 CUs / 8 XCDs costs 4-26 us; control flow stays on the
//     device (a control block at byte 0 of the momentary buffer), kernels of groups that do not exist exit at once, so the call is fully
//     asynchronous like the reference's.  The front half (the tree grows, every chunk the group needs is allocated) runs on the caller's
//     stream, the back half (points stored, voxels sampled and stored) on a second stream of the context, one group behind;
//   * every sample is read with one coalesced 16-byte load per pass and descends the tree ONCE (k_count): its leaf is cached (4 B/sample),
//     and when that leaf splits the word becomes (split slot, histogram bin) — the final leaf is one lookup in the slot's map;
//   * a leaf that crosses 50 000 gets a SLOT with a 512-bin histogram — three octree levels — of everything that lies in it (k_hist:
//     the group's samples and the leaf's stored points, which move to a spill buffer on the way).  The whole cascade of voxels.cu:245-415
//     is decided FROM THE COUNTS (k_expand): up to 584 nodes per slot and round, counters filled in, chunks allocated; only a
//     great-grandchild that is still too full costs another round (another histogram pass over the samples inside k_expand, one grid
//     barrier).  The reference needs ~8 grid.sync() per LEVEL and re-scans the batch in every one of them;
//   * per-leaf counters are aggregated per WORKGROUP in LDS hash tables (one global atomic per workgroup and leaf): device-scope
//     atomics on one word retire at ~88 M/s on this chip, and a spatially compact batch sends most of its samples to a few dozen leaves;
//   * voxel sampling walks the root path BOTTOM-UP (occupancy is hierarchical: a set bit implies the covering bits of all ancestors),
//     AFTER the insert, leaf by leaf, in LDS copies of the cubes of the ancestors' grids a leaf can touch (k_voxelize): the grids see one
//     atomicOr per touched word instead of one per sample.  The same kernel reserves the voxels' slots (one atomic per piece and
//     ancestor), allocates voxel chunks ON DEMAND (whoever reserves the first slot of a chunk allocates it and publishes it in a hash
//     directory) and stores the voxels: no second allocation / insertion pass;
//   * the O(list length) chunk walks of voxels.cu:606-610 / 688-692 / 500-503 are gone: the head chunk of a list remembers its tail (8
//     spare bytes of Chunk), a per-group chunk directory gives O(1) slot -> chunk, a per-node chunk table (also read by the rasteriser)
//     lets a split hand a leaf's whole list to the spill copy with one wave;
//   * no capacity limit loses a point: a split reserves its slot, its node slots and its spill space in ONE compare-and-swap or does
//     not happen yet (the leaf grows and is queued again by a later batch; Stats.dbg says so).
//
// The result after every batch is the reference's: same topology, same per-node sample multisets, same occupancy bitsets, same voxel
// positions (bit-exact fp32), same counters in Node and Stats, same allocator offset, same chunk-pool accounting.  What stays
// scheduling dependent is what is scheduling dependent in the reference too: node indices, chunk addresses, sample
// order inside a node, which point colours a voxel.
#include <mutex>
#include "simlod_device.hpp"
#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {
namespace build {

// Per-group state (BatchCtl), BATCH_COPIES copies in the control block, group #ordinal of a launch in copy ordinal & 3.
static constexpr uint32_t SLOT_CAP_GRIDS = 256;  // (memory guard's slack: grids one group's splits may allocate; more than that many splits per group and the guard is a group late)
static constexpr uint32_t BATCH_COPIES = 4;      // per-batch state of batch b lives in copy b & 3: the front half of batch b + 1 (count .. expand) and the back halves of
                                                 // batches b and b - 1 (insert, voxelize) are under way together, and batch b + 2 is being prepared
struct BatchCtl {
	uint32_t active, batchSize, ringSlot, batchIndex;
	uint32_t ordinal, tag, slotsRound0, numSpilled;   // tag = batch index + 1 (NodeDir); slotsRound0: slots handed out by k_count's tail, snapshot by k_hist: k_expand's first round
	uint32_t numWork, numClear, numTouched, numCross;    // spill-copy work items | grids k_insert has to clear | leaves with new samples (k_expand allocates their chunks) | leaves k_count saw cross the limit (k_queue)
	uint32_t barrierCount, rootPieces, numVoxItems, numVoxSmall;     // rootPieces: pieces of a root that is still a leaf (k_voxroot)
	uint32_t groupBatches, dirCount, acct, accounted;      // ring batches taken together: 1 in exact mode, up to groupMax in coalesced mode; batchSize = all their samples | acct: an EXACT group of several batches — counts and histograms are kept per batch, the allocator / chunk-pool counters are brought to what batch-by-batch ingestion leaves (account_group) | accounted: that has happened
	uint32_t start[SIMLOD_MAX_BATCHES_PER_LAUNCH + 1];   // sample index of the first sample of batch k of the group (start[groupBatches] = batchSize)
	uint32_t slot[SIMLOD_MAX_BATCHES_PER_LAUNCH];        // its ring slot
	uint32_t pad3;
	unsigned long long reserve;        // split slots << 52 | nodes in use << 32 | spilled points of this batch — ONE word, so a split reserves all or nothing
	unsigned long long reserve0;       // ... as the group began (k_count's first workgroup): what k_queue's entries count from
	// per-batch chunk accounting of an exact group (acct): point chunks batch k of the group would have taken / given back had the batches been ingested one
	// by one (voxels.cu:346-357, 485-538), filled by k_expand; the chunk counters as they stood when the group began (k_count's first workgroup)
	uint32_t acctAlloc0, acctPool0, rootSplitAt, pad5;      // rootSplitAt: an exact group in which the ROOT splits: the batch of the group it splits in (k_rootpre); NONE otherwise
	uint32_t acctD[SIMLOD_MAX_BATCHES_PER_LAUNCH], acctF[SIMLOD_MAX_BATCHES_PER_LAUNCH];
};

// Control block at byte 0 of kernel_construct's momentary buffer.  Lives only for the duration of one launch
// (the recycle stack behind it, like the reference's chunkQueue, must survive between launches).
struct Ctl {
	uint32_t uploaded, firstBatch, numBatches, stop;
	uint32_t errors, abortBatch, rebuildLeafChunks, debugFlags;   // rebuildLeafChunks: this launch found its side tables stale (first launch, reset, wiped or re-laid-out momentary buffer): k_rebuild / k_paths clear the tag words and refill parents, chunk table, paths and the top table — otherwise they are what the launch before left
	uint32_t processed, budgetUs, consumed, groupMax;   // groups completed in this launch | its time budget in us (voxels.cu:22: 10 ms; SIMLOD_DEBUG_BUDGET_US overrides) | ring batches taken so far | batches per group (1: exact mode)
	uint64_t startNs;
	uint32_t statCounters[8];
	uint32_t tableMagic, tableBatch;   // leaf chunk table is valid for the octree as it was after batch #tableBatch (k_finish) ...
	uint64_t tableNodes, tablePers;    // ... of THIS octree (node array, persistent buffer)
	uint64_t tableSig;                 // table_signature() of the Stats the table belongs to
	uint64_t tableLayout;              // layout_signature() of the momentary buffer the side tables were built in
	uint64_t pointsTaken;              // samples of all batches taken so far, this launch's included (Stats.numPointsProcessed follows when their back halves have run)
	uint64_t unused1[2];
	uint64_t expandNs[8];              // byte 152: k_expand phase times of workgroup 0 (in-kernel histogram pass, barrier, decide + build, barrier, [4] = groups ingested so far (bench.py), rounds, calls; tools/probe.py), [7] = spilled points so far (bench.py)
	uint64_t voxT[SIMLOD_MAX_BATCHES_PER_LAUNCH][3];   // byte 216: k_voxelize of group #ordinal of the last launch: first workgroup in, last piece done, last workgroup out (tools/probe.py)
	uint64_t phaseNs[48];              // byte 696: phase times of one workgroup per kernel, summed over the launches since the host last cleared them (tools/probe.py)
	BatchCtl batch[BATCH_COPIES];
	uint32_t tagOf[SIMLOD_MAX_BATCHES_PER_LAUNCH];     // tag of group #ordinal of this launch (whoever closes a group's voxel lists later needs it: its parity copy is recycled by then)
};
static_assert(offsetof(Ctl, voxT) == 216 && offsetof(Ctl, phaseNs) == 696, "tools/probe.py reads Ctl.voxT at byte 216, Ctl.phaseNs at byte 696");
static_assert(offsetof(Ctl, expandNs) == 152, "bench.py / tools read Ctl.expandNs at byte 152");
static_assert(offsetof(Ctl, batch) == 1080 && sizeof(BatchCtl) == 440, "tools/batch_shape.py reads Ctl.batch at byte 1080");
static_assert(sizeof(Ctl) <= 4096, "control block");

struct BuildArgs {
	SimlodPoint* ring;
	uint8_t*     mom;
	uint8_t*     pers;
	SimlodNode*  nodes;
	SimlodStats* stats;
	uint64_t*    frameStart;
	uint32_t*    numBatchesUploaded;
	uint32_t*    batchSizes;
	float        minx, miny, minz, size;
	uint64_t     persCapacity, frameCounter, scratchBytes;
	uint64_t     offQueue, offSlots, offHist, offMap, offClear, offTouched, offSplitTag, offRetryTag, offParent, offNodeDir, offChunkDir, offLeafChunks, offPaths, offWork, offLeafOf, offVoxItems, offSpilled, offHashDir, offTouchTag, offStartOf, offCross, offTop, offKid, leafOfStride;
	uint32_t     nodeCapacity, spilledCap, dirCap, workCap, voxItemCap, clearCap, hashCap, groupCap, groupMax, crossCap;   // groupCap = groupMax * 1 000 000: where the moved points' words start in leafOf
	uint64_t     trunkLo, trunkHi;   // simlod_context_set_trunk_mask: nodes of levels 0-2 that split whatever they hold (multi-GPU: the shared upper levels); zero on one GPU
	uint64_t     offCntB, offHistB;  // exact groups (acct): per node, samples of batch k of the group (groupMax words) | per (slot, bin), likewise
	uint32_t     acct, pad;          // exact mode with groups of several batches (groupMax > 1): see account_group
};

// ---- the shared upper levels of a multi-GPU job (include/simlod_hip.h simlod_context_set_trunk_mask; no counterpart in the reference, which is
// single-GPU) ----  Ranks own level-3 cells of one global cube; the nodes of levels 0-2 exist on every rank.  A rank that split them by ITS counts
// would keep one as a leaf where the single-GPU octree of the whole data set has an inner node (voxels.cu:209-217: a leaf splits when the count
// under it crosses 50 000 — the GLOBAL count there), and kernel_render would draw that rank's points where the single GPU draws voxels
// (render.cu:918-932).  So the host names the upper nodes whose global count exceeds the limit: such a node splits as soon as it exists.
// Bit 0: the root; 1 + c: the level-1 node with cell code c = x << 2 | y << 1 | z; 9 + c: the level-2 node, c = the level-1 octant << 3 | the octant below.
static constexpr uint32_t TRUNK_LEVELS = 3, TRUNK_NODES = 1 + 8 + 64;
__device__ __forceinline__ bool trunk_any(const BuildArgs& a) { return (a.trunkLo | a.trunkHi) != 0ull; }
__device__ __forceinline__ uint32_t trunk_index(uint32_t level, uint32_t X, uint32_t Y, uint32_t Z) {
	return level == 0u ? 0u : level == 1u ? 1u + ((X & 1u) << 2 | (Y & 1u) << 1 | (Z & 1u))
	                        : 9u + ((((X >> 1) & 1u) << 2 | ((Y >> 1) & 1u) << 1 | ((Z >> 1) & 1u)) << 3 | ((X & 1u) << 2 | (Y & 1u) << 1 | (Z & 1u)));
}
__device__ __forceinline__ bool trunk_forced(const BuildArgs& a, uint32_t level, uint32_t X, uint32_t Y, uint32_t Z) {
	if (level >= TRUNK_LEVELS) return false;
	const uint32_t i = trunk_index(level, X, Y, Z);
	return (((i < 64u ? a.trunkLo : a.trunkHi) >> (i & 63u)) & 1ull) != 0ull;
}


// ---- workgroup-level key -> count aggregation in LDS ------------------------------------------------------------------
// A device-scope atomic on ONE address retires at ~88 M/s on this chip (MI355X_MICROARCH.md, rows fanin / dequeue), and a
// spatially compact 1 M-point batch funnels most of its points into a few dozen leaves: per-wave aggregation still
// leaves ~16 k atomics per hot counter per batch.  So every counter update of the build (leaf arrival counters, slot
// reservations, voxel counters) is first combined per WORKGROUP in an open-addressing LDS table and flushed with one
// global atomic per (workgroup, node).  table_add returns the entry and the value the entry's counter had before
// (= rank of this caller inside the workgroup), or -1 when 16 probes found no room; a key that failed once keeps
// failing (entries are never removed), so callers can fall back to a direct global atomic consistently.
static constexpr uint32_t TBL_EMPTY = 0xffffffffu;
static constexpr int TBL_BITS = 10;
static constexpr int TBL_CAP = 1 << TBL_BITS;

struct BlockTable {
	uint32_t keys[TBL_CAP];
	uint32_t vals[TBL_CAP];
};

__device__ __forceinline__ void table_init(BlockTable& t) {
	for (uint32_t i = threadIdx.x; i < (uint32_t)TBL_CAP; i += blockDim.x) { t.keys[i] = TBL_EMPTY; t.vals[i] = 0u; }
}

__device__ __forceinline__ uint32_t table_hash(uint32_t key) { return (key * 2654435761u) >> (32 - TBL_BITS); }

__device__ __forceinline__ int table_add(BlockTable& t, uint32_t key, uint32_t inc, uint32_t* rank) {
	uint32_t h = table_hash(key);
#pragma unroll 1
	for (int probe = 0; probe < 16; ++probe) {
		uint32_t k = t.keys[h];
		if (k == TBL_EMPTY) { k = atomicCAS(&t.keys[h], TBL_EMPTY, key); if (k == TBL_EMPTY) k = key; }
		if (k == key) { *rank = atomicAdd(&t.vals[h], inc); return (int)h; }
		h = (h + 1) & (TBL_CAP - 1);
	}
	return -1;
}

// The same table with every counter REP times (k_count): a swath-ordered batch sends most lanes of a wave to a handful of leaves, and the LDS
// serialises atomics of one instruction on one address lane by lane (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.88 in k_count).  Lane l
// adds to copy l % REP — REP consecutive words, REP different banks — and the flush sums the copies.
template <uint32_t REP>
struct SpreadTable {
	uint32_t keys[TBL_CAP];
	uint32_t vals[TBL_CAP * REP];
};
template <uint32_t REP>
__device__ __forceinline__ void spread_init(SpreadTable<REP>& t) {
	for (uint32_t i = threadIdx.x; i < (uint32_t)TBL_CAP; i += blockDim.x) t.keys[i] = TBL_EMPTY;
	for (uint32_t i = threadIdx.x; i < (uint32_t)TBL_CAP * REP; i += blockDim.x) t.vals[i] = 0u;
}
template <uint32_t REP>
__device__ __forceinline__ bool spread_add(SpreadTable<REP>& t, uint32_t key, uint32_t copy) {
	uint32_t h = table_hash(key);
#pragma unroll 1
	for (int probe = 0; probe < 16; ++probe) {
		uint32_t k = t.keys[h];
		if (k == TBL_EMPTY) { k = atomicCAS(&t.keys[h], TBL_EMPTY, key); if (k == TBL_EMPTY) k = key; }
		if (k == key) { atomicAdd(&t.vals[h * REP + copy], 1u); return true; }
		h = (h + 1) & (TBL_CAP - 1);
	}
	return false;
}
template <uint32_t REP>
__device__ __forceinline__ uint32_t spread_sum(const SpreadTable<REP>& t, uint32_t e) {
	uint32_t s = 0;
#pragma unroll
	for (uint32_t r = 0; r < REP; r++) s += t.vals[e * REP + r];
	return s;
}

__device__ __forceinline__ int table_find(const BlockTable& t, uint32_t key) {
	uint32_t h = table_hash(key);
#pragma unroll 1
	for (int probe = 0; probe < 16; ++probe) {
		const uint32_t k = t.keys[h];
		if (k == key) return (int)h;
		if (k == TBL_EMPTY) return -1;
		h = (h + 1) & (TBL_CAP - 1);
	}
	return -1;
}



static constexpr uint32_t TPB = 256;
static constexpr float F_GRID = 1048576.0f;      // 2^MAX_DEPTH, progressive_octree_voxels.cu:139
static constexpr float F_FULL = 268435456.0f;    // MAX_DEPTH_GRIDSIZE, structures.cuh:26

struct NodeDir {          // per node, valid for the batch whose tag it carries: where the leaf's chunks stand in the batch's chunk directory
	uint32_t ptBase, ptFirst, ptTag, pad0;
};

// Leaf chunk table: slot k of leaf i's point list -> chunk, LEAF_SLOTS entries per node.  A leaf that can still split stores
// at most MAX_POINTS_PER_NODE points between batches (= 50 chunks), so the split reads its whole list from here with all
// lanes at once instead of chasing 50 `next` pointers (~1 us each) with one.  Kept up to date by alloc_points (k_expand); survives between
// launches like the recycle stack does, and is refilled by k_rebuild whenever k_begin finds its stamp stale.
static constexpr uint32_t LEAF_SLOTS = LEAF_ROW_SLOTS;      // (rows are packed: simlod_internal.hpp leaf_row_get / leaf_row_set)
static constexpr uint32_t TABLE_MAGIC = 0x51ab1e05u;
static_assert(LEAF_SLOTS <= 64, "queue_split hands a leaf's chunks out one per lane");

// Ancestor paths: PATH_WORDS 64-bit entries per node, entry k = the k-th ancestor (parent first), zero-terminated.
// An entry packs everything `sample` and `insert` need to know about that ancestor — its occupancy grid (offset into the
// persistent buffer), level and node index — so a sample reads its whole root path with independent loads instead of chasing
// parent -> node -> grid pointers level by level.
// Rebuilt for every node at the start of a launch (k_paths), extended for the eight children at a split (k_expand).
static constexpr uint32_t PATH_WORDS = SIMLOD_MAX_DEPTH + 1;
static constexpr unsigned long long PATH_VALID = 1ull << 63;

__device__ __forceinline__ unsigned long long path_pack(const uint8_t* pers, uint32_t nodeIdx, uint32_t level, const SimlodOccupancyGrid* grid) {
	const unsigned long long off = (unsigned long long)(reinterpret_cast<const uint8_t*>(grid) - pers) >> 4;     // grids are 16-byte aligned allocations
	return PATH_VALID | ((unsigned long long)nodeIdx << 41) | ((unsigned long long)level << 36) | off;
}
__device__ __forceinline__ uint32_t path_node(unsigned long long e) { return (uint32_t)(e >> 41) & 0x7ffffu; }
__device__ __forceinline__ uint32_t path_level(unsigned long long e) { return (uint32_t)(e >> 36) & 31u; }
__device__ __forceinline__ SimlodOccupancyGrid* path_grid(uint8_t* pers, unsigned long long e) {
	return reinterpret_cast<SimlodOccupancyGrid*>(pers + ((e & 0xfffffffffull) << 4));
}

// Top table: for every cell of the 32^3 grid of level 5, the deepest node at level <= 5 that contains it, as node | level << 19 — where k_count's
// descent STARTS (one load instead of up to five dependent ones; nodes are never removed, so an entry is always a valid starting point).
// Rebuilt at the start of every launch (k_paths), kept current by k_expand: only a split of a leaf at level <= 4 makes nodes that belong in
// it, which a stream does in its first batches over a region and hardly ever again (the 36 M terrain: leaves live at levels 6 to 8).
// (Round 3 measured the same idea one level deeper — 64^3, level 6 — and dropped it: every other split had to update it, +3.4 us in k_expand.)
static constexpr uint32_t TOP_LEVEL = 5, TOP_SIDE = 1u << TOP_LEVEL, TOP_CELLS = TOP_SIDE * TOP_SIDE * TOP_SIDE;
__device__ __forceinline__ uint32_t top_cell(uint32_t X, uint32_t Y, uint32_t Z) {      // X, Y, Z: 20-bit grid coordinates (bits 19..0 count, as in the descent)
	const uint32_t s = (uint32_t)SIMLOD_MAX_DEPTH - TOP_LEVEL;
	return (((X >> s) & (TOP_SIDE - 1u)) << (2u * TOP_LEVEL)) | (((Y >> s) & (TOP_SIDE - 1u)) << TOP_LEVEL) | ((Z >> s) & (TOP_SIDE - 1u));
}
// node `idx` at `level` <= TOP_LEVEL with coordinates (X, Y, Z) becomes the entry of every cell it covers
__device__ __forceinline__ void top_fill(uint32_t* top, uint32_t idx, uint32_t level, uint32_t X, uint32_t Y, uint32_t Z, uint32_t mark) {      // mark: TOP_LEAF for a node without children
	const uint32_t k = TOP_LEVEL - level, side = 1u << k, x0 = X << k, y0 = Y << k, z0 = Z << k, e = idx | (level << 19) | mark;      // (sides are powers of two: shifts, no division)
	for (uint32_t i = 0; i < (1u << (3u * k)); i++) {
		const uint32_t dx = i >> (2u * k), dy = (i >> k) & (side - 1u), dz = i & (side - 1u);
		top[((x0 + dx) << (2u * TOP_LEVEL)) | ((y0 + dy) << TOP_LEVEL) | (z0 + dz)] = e;
	}
}

// Child words: ONE 32-bit word per node instead of its eight 8-byte child pointers spread over a 152-byte record — what k_count's descent reads.  A node that
// splits gets all eight children at once, in eight consecutive node slots (voxels.cu:316-343: atomicAdd(&stats->numNodes, 8); here: k_queue / reserve()), so
// the word holds the FIRST child's index (19 bits) and, above it, which of the eight are leaves (8 bits): a sample that steps into a child marked as a leaf
// is done without looking at that child.  0: the node has no children.  KID_IRREGULAR: children that are not eight consecutive nodes (an image neither this
// builder nor the reference made): that node is descended through Node.children as before.  The 36 M terrain's 4 425 nodes: 17 KB, L1-resident, against
// 4 425 x 152 B of records of which a step used 8 bytes.  Kept current by k_expand (a node that splits clears its bit in its parent's word), restored from
// the node array with the other side tables (rebuild_side_tables).  The top table's entries carry the same mark (TOP_LEAF).
static constexpr uint32_t KID_IRREGULAR = 0xffffffffu, KID_LEAF_SHIFT = 19, TOP_LEAF = 0x80000000u;
__device__ __forceinline__ uint32_t octant_of(uint32_t X, uint32_t Y, uint32_t Z) { return ((X & 1u) << 2) | ((Y & 1u) << 1) | (Z & 1u); }      // a node's place among its parent's children (voxels.cu:320-322)

__device__ __forceinline__ Ctl* ctl_of(const BuildArgs& a) { return reinterpret_cast<Ctl*>(a.mom); }
template <class T> __device__ __forceinline__ T* at(const BuildArgs& a, uint64_t off) { return reinterpret_cast<T*>(a.mom + off); }

// the per-batch state of batch #ordinal of this launch, or nullptr when that batch does not exist (the copy of its parity may still
// hold an earlier batch: the launch enqueues kernels for 20 batches whether they exist or not)
__device__ __forceinline__ BatchCtl* batch_of(Ctl* ctl, uint32_t ordinal) {
	BatchCtl* bc = &ctl->batch[ordinal % BATCH_COPIES];
	return bc->active != 0u && bc->ordinal == ordinal ? bc : nullptr;
}

// phase timer of ONE thread of one workgroup per kernel: adds the time since `t` to slot k and restarts `t` (builds with SIMLOD_MEASURE only: `on` is a
// compile-time false in the product library and every mark() folds away)
struct Phase {
	Ctl* ctl; bool on; uint64_t t;
	__device__ __forceinline__ Phase(Ctl* c, bool who) : ctl(c), on(SIMLOD_MEASURE != 0 && who && threadIdx.x == 0), t(on ? wall_ns() : 0) {}
	__device__ __forceinline__ void mark(uint32_t k) { if (on) { const uint64_t n = wall_ns(); ctl->phaseNs[k] += n - t; t = n; } }
};

// The chunk directory, k_voxelize's work items and the cached-leaf words of a batch exist twice, by the parity of the batch's ordinal: batch
// b + 1's are written by its front half (k_count .. k_expand) while the back half of batch b (k_insert, k_voxelize) is still reading its own.
struct VoxItem;
__device__ __forceinline__ SimlodChunk** chunk_dir(const BuildArgs& a, const BatchCtl* bc) { return at<SimlodChunk*>(a, a.offChunkDir) + (uint64_t)(bc->ordinal & 1u) * a.dirCap; }
__device__ __forceinline__ VoxItem* vox_items(const BuildArgs& a, const BatchCtl* bc);

__device__ __forceinline__ void raise(Ctl* ctl, uint32_t bit) { atomicOr(&ctl->errors, bit); }
// conditions after which the batch cannot be completed: the rest of the chain does nothing, Stats.dbg keeps the bit until a reset
__device__ __forceinline__ void panic(Ctl* ctl, uint32_t bit) { atomicOr(&ctl->errors, bit); ctl->abortBatch = 1; ctl->stop = 1; }

// Worst case of what the voxel half of a batch can still add to the persistent buffer (voxel chunks: every sample can colour one voxel per
// level, every inner node can start one more chunk).  The memory guard of voxels.cu:896-912 looks at the allocator after the WHOLE previous
// batch; here a batch is prepared while the voxel halves of the TWO batches before it may still be running, so within this distance of the
// guard a launch takes one batch only: the next launch's k_begin runs after everything and decides exactly.
__host__ __device__ inline unsigned long long group_slack_bytes(unsigned long long samples, unsigned long long numNodes);
__device__ __forceinline__ unsigned long long slack_for(const BuildArgs& a, unsigned long long samples) { return group_slack_bytes(samples, a.stats->numNodes); }
__host__ __device__ inline unsigned long long group_slack_bytes(unsigned long long samples, unsigned long long numNodes) {
	// (+ the point chunks and grids the group before has yet to allocate: its k_expand runs after this look at the allocator)
	return (2ull * (samples * SIMLOD_MAX_DEPTH / SIMLOD_POINTS_PER_CHUNK + numNodes + 1ull) + samples / SIMLOD_POINTS_PER_CHUNK + 4096ull) * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk))
	       + (unsigned long long)SLOT_CAP_GRIDS * SIMLOD_ALLOC_ROUND(sizeof(SimlodOccupancyGrid));
}
__device__ __forceinline__ unsigned long long voxel_half_slack(const BuildArgs& a, const BatchCtl* prev) {
	return slack_for(a, (unsigned long long)prev->batchSize + prev->numSpilled);
}

// Make group #ordinal of this launch current (in copy ordinal & 3), or leave it inactive (progressive_octree_voxels.cu:890-912).  Runs on the
// FRONT stream: k_begin for the first group, one thread of k_hist of the group before for the others (so that k_count can follow k_expand
// without a kernel in between; what depends on that k_expand — the node array's fill, the chunk pool's high-water mark — is set by
// k_count's first workgroup).
// Exact mode: a group is ONE ring batch, the reference's granularity.  Coalesced mode (simlod_set_ingest_mode(1)): the next groupMax
// pending batches are counted, split, stored and voxelized as one; the memory guard is looked at once per group.
__device__ void prepare_batch(const BuildArgs& a, Ctl* ctl, uint32_t ordinal) {
	BatchCtl* bc = &ctl->batch[ordinal % BATCH_COPIES];
	bc->active = 0;
	if (ordinal >= SIMLOD_MAX_BATCHES_PER_LAUNCH || ctl->consumed >= ctl->numBatches) return;
	if (__hip_atomic_load(&ctl->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;      // (set by the back half of an earlier group: time budget, abort)
	const SimlodAllocatorGlobal* alloc = reinterpret_cast<const SimlodAllocatorGlobal*>(a.pers);
	if (ordinal > 0u && alloc->offset + SIMLOD_MEM_SAFETY_MARGIN + voxel_half_slack(a, &ctl->batch[(ordinal - 1u) % BATCH_COPIES]) >= a.persCapacity) { ctl->stop = 1; return; }
	const bool full = alloc->offset + SIMLOD_MEM_SAFETY_MARGIN >= a.persCapacity;
	a.stats->memCapacityReached = full ? 1 : 0;
	if (full) { ctl->stop = 1; return; }
	const uint32_t batchIndex = ctl->firstBatch + ctl->consumed;       // (Stats.batchletIndex itself is advanced by the back half, which may lag)
	uint32_t take = min(ctl->groupMax, ctl->numBatches - ctl->consumed);
	if (a.acct != 0u && take > 1u) {
		// An EXACT group of several batches is taken only where the reference's guard — looked at before EVERY batch (voxels.cu:896-912) — cannot
		// trip inside it: the allocator must be a worst-case group away from it.  Closer than that, the launch goes batch by batch as before.
		unsigned long long samples = 0;
		for (uint32_t k = 0; k < take; k++) samples += min(a.batchSizes[(batchIndex + k) % SIMLOD_BATCH_STREAM_SIZE], (uint32_t)SIMLOD_MAX_BATCH_SIZE);
		// (stored points a split moves are sampled again: at most what the octree holds, at most what the spill buffer takes)
		samples += min((unsigned long long)a.spilledCap, (unsigned long long)ctl->pointsTaken);
		const unsigned long long before = ordinal > 0u ? voxel_half_slack(a, &ctl->batch[(ordinal - 1u) % BATCH_COPIES]) : 0ull;
		if (alloc->offset + SIMLOD_MEM_SAFETY_MARGIN + before + slack_for(a, samples) >= a.persCapacity) take = 1u;
	}
	uint32_t total = 0;
	for (uint32_t k = 0; k < take; k++) {
		const uint32_t slot = (batchIndex + k) % SIMLOD_BATCH_STREAM_SIZE;
		uint32_t size = a.batchSizes[slot];
		if (size > SIMLOD_MAX_BATCH_SIZE) size = SIMLOD_MAX_BATCH_SIZE;
		bc->start[k] = total; bc->slot[k] = slot;
		total += size;
	}
	bc->start[take] = total;
	ctl->consumed += take;
	ctl->pointsTaken += total;
	bc->groupBatches = take;
	bc->batchIndex = batchIndex;
	bc->ringSlot = bc->slot[0];
	bc->batchSize = total;
	bc->ordinal = ordinal;
	bc->tag = batchIndex + 1u;
	ctl->tagOf[ordinal] = bc->tag;
	bc->slotsRound0 = 0;
	bc->numSpilled = 0;
	bc->numWork = 0;
	bc->numClear = 0;
	bc->numTouched = 0;
	bc->numCross = 0;
	bc->barrierCount = 0;      // every k_expand instance counts its barrier generations from zero
	bc->numVoxItems = 0;
	bc->numVoxSmall = 0;
	bc->rootPieces = 0;
	bc->dirCount = 0;
	bc->reserve = 0;           // (k_count's first workgroup: the node array as k_expand of the group before leaves it)
	bc->acct = a.acct != 0u && take > 1u ? 1u : 0u;
	bc->accounted = 0;
	bc->rootSplitAt = 0xffffffffu;
	for (uint32_t k = 0; k < SIMLOD_MAX_BATCHES_PER_LAUNCH; k++) { bc->acctD[k] = 0; bc->acctF[k] = 0; }
	bc->active = 1;
}

// The samples of the current group: batch k of the group holds samples [start[k], start[k + 1]) in its ring slot (a batch holds at most
// 1 000 000 samples, so batch i / 1 000 000 is the first candidate).  Exact mode — one batch per group — is a plain array.
// (SINGLE: the kernel was launched for exact mode — the host knows — and carries no trace of the group lookup)
template <bool SINGLE>
struct Samples {
	const float4* ring; const float4* only; const BatchCtl* bc; uint32_t batches;
	__device__ __forceinline__ Samples(const BuildArgs& a, const BatchCtl* b) : ring(reinterpret_cast<const float4*>(a.ring)), bc(b), batches(b->groupBatches) {
		only = ring + (size_t)b->ringSlot * SIMLOD_MAX_BATCH_SIZE;
	}
	__device__ __forceinline__ const float4* ptr(uint32_t i) const {
		if (SINGLE || batches == 1u) return only + i;
		uint32_t k = min(i / SIMLOD_MAX_BATCH_SIZE, batches - 1u);
		while (k + 1u < batches && i >= bc->start[k + 1u]) k++;
		return ring + (size_t)bc->slot[k] * SIMLOD_MAX_BATCH_SIZE + (i - bc->start[k]);
	}
	__device__ __forceinline__ float4 operator[](uint32_t i) const { return *ptr(i); }
	// Do the samples i0 .. i1 of the group lie in ONE ring batch (a workgroup's tile of consecutive samples does, as a rule: a batch has up to a million)?
	// Then sample i is base[i] and its batch of the group is k — without a lookup per sample (the walk through start[] in front of every load was a
	// dependent round trip in front of the kernels' first HBM access).
	__device__ __forceinline__ bool span(uint32_t i0, uint32_t i1, const float4*& base, uint32_t& k) const {
		k = 0; base = only;
		if (SINGLE || batches == 1u) return true;
		k = min(i0 / SIMLOD_MAX_BATCH_SIZE, batches - 1u);
		while (k + 1u < batches && i0 >= bc->start[k + 1u]) k++;
		base = ring + (size_t)bc->slot[k] * SIMLOD_MAX_BATCH_SIZE - bc->start[k];
		return i1 < bc->start[k + 1u];
	}
};

// what the stamp remembers of the momentary buffer's layout: side tables of another node capacity / buffer size / group size are not these
__host__ __device__ inline uint64_t layout_signature(const BuildArgs& a) {
	return a.scratchBytes ^ ((uint64_t)a.nodeCapacity << 40) ^ ((uint64_t)a.groupMax << 59) ^ (a.offSpilled * 0x9E3779B97F4A7C15ull);
}

// ---- begin: snapshot the upload counter, stamp the frame start (voxels.cu:823-825, 870-885); restore the side tables when they are stale -------
// The side tables — parents, ancestor paths, the top table, the chunk table, the recycle stack — and the per-node tag words survive between
// launches: k_expand keeps them current split by split, and every tag is a batch index + 1, which only grows while an octree lives.  They
// are rebuilt only when the stamp k_finish left does not name THIS octree in THIS state in THIS layout: the first launch, after a reset, an
// uploaded image, an aborted batch, a wiped or resized momentary buffer.  (Tags travel in 20 bits through the hash directory of voxel chunks: a
// full clear every 2^19 batches keeps them unambiguous.)  Every thread of the launch reads the stamp for itself — nobody writes those words
// while k_begin runs: the stamp is taken off by the launch's first k_count —, so the decision needs no second kernel (round 4: a memset and two
// kernels per launch; round 5, first: two kernels that exited at once; 9 us each on a chain that is one batch long).
__device__ __forceinline__ bool stamp_is_stale(const BuildArgs& a, const Ctl* ctl) {
	const uint32_t first = a.stats->batchletIndex;
	return ctl->tableMagic != TABLE_MAGIC || ctl->tableBatch != first || ctl->tableNodes != (uint64_t)a.nodes || ctl->tablePers != (uint64_t)a.pers ||
	       ctl->tableLayout != layout_signature(a) || ctl->tableSig != table_signature(a.stats) ||
	       (first >= 0x80000u && (first & 0x7ffffu) < SIMLOD_MAX_BATCHES_PER_LAUNCH);
}

// (the 8 spare bytes Chunk::size / padding_0 of a list's HEAD chunk: the address of the list's last chunk)
__device__ __forceinline__ SimlodChunk*& tail_of(SimlodChunk* head) { return *reinterpret_cast<SimlodChunk**>(&head->size); }

// The side tables of an octree this buffer does not describe.  The per-node tag words and the hash directory are zeroed (offSplitTag .. offParent:
// a stale word could pass for a tag of this octree's batches), parents come from the children pointers, the rows of the chunk table from the
// lists, the top table and every node's ancestor list from a descent from the root (a node knows its level and cell: the ancestor at level l is
// entry level - 1 - l of its list — no pass has to wait for the parent table).
__device__ void rebuild_side_tables(const BuildArgs& a) {
	const uint64_t first = (uint64_t)blockIdx.x * TPB + threadIdx.x, stride = (uint64_t)gridDim.x * TPB;
	{
		uint4* w = reinterpret_cast<uint4*>(a.mom + a.offSplitTag);
		const uint64_t n = (a.offParent - a.offSplitTag) / 16;                 // (the offsets are 256-byte aligned)
		for (uint64_t i = first; i < n; i += stride) w[i] = make_uint4(0, 0, 0, 0);
	}
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	uint32_t* parentOf = at<uint32_t>(a, a.offParent);
	for (uint64_t i = first; i < TOP_CELLS; i += stride) {      // the top table: cell i's deepest node at level <= TOP_LEVEL
		const uint32_t s = (uint32_t)SIMLOD_MAX_DEPTH - TOP_LEVEL;
		const uint32_t X = ((uint32_t)i >> (2u * TOP_LEVEL)) << s, Y = (((uint32_t)i >> TOP_LEVEL) & (TOP_SIDE - 1u)) << s, Z = ((uint32_t)i & (TOP_SIDE - 1u)) << s;
		uint32_t cur = 0, level = 0;
		while (level < TOP_LEVEL) {
			const SimlodNode* c = a.nodes[cur].children[child_index(X, Y, Z, (int)level)];
			if (c == nullptr) break;
			cur = (uint32_t)(c - a.nodes); level++;
		}
		at<uint32_t>(a, a.offTop)[i] = cur | (level << 19) | (node_is_leaf(a.nodes + cur) ? TOP_LEAF : 0u);
	}
	for (uint64_t i = first; i < numNodes; i += stride) {
		if (i == 0) parentOf[0] = 0xffffffffu;
		const SimlodNode* n = a.nodes + i;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const SimlodNode* c = n->children[k];
			if (c != nullptr) parentOf[(uint32_t)(c - a.nodes)] = (uint32_t)i;
		}
		{   // the node's child word
			const SimlodNode* c0 = n->children[0];
			uint32_t word = 0u, some = 0u;
			bool regular = c0 != nullptr;
#pragma unroll
			for (int k = 0; k < 8; k++) {
				const SimlodNode* c = n->children[k];
				if (c != nullptr) some++;
				if (c == nullptr || c != c0 + k) regular = false;
				else if (node_is_leaf(c)) word |= 1u << (KID_LEAF_SHIFT + (uint32_t)k);
			}
			at<uint32_t>(a, a.offKid)[i] = some == 0u ? 0u : regular ? (word | (uint32_t)(c0 - a.nodes)) : KID_IRREGULAR;
		}
		// a leaf's row: its point chunks; an inner node's row: its voxel chunks (for the rasteriser).  And the word this builder keeps in the spare
		// bytes of a list's HEAD chunk, the address of the list's last chunk (O(1) append): an image built elsewhere — by the reference — has
		// none; found by walking the list once, here.
		uint8_t* const table = a.mom + a.offLeafChunks;
		const bool leaf = node_is_leaf(n);
		SimlodChunk* const head = leaf ? n->points : n->voxelChunks;
		const uint32_t inList = ((leaf ? n->numPoints : n->numVoxelsStored) + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
		SimlodChunk* c = head;
		for (uint32_t k = 0; c != nullptr && k < max(inList, 1u); k++) {      // (not beyond the list's last chunk: whether its `next` is null is the other builder's business)
			if (k < LEAF_SLOTS) leaf_row_set(table, a.pers, i, k, c);
			if (k + 1u == inList) tail_of(head) = c;
			c = c->next;
		}
		// (a LEAF that also has a voxel list — the root while it is still a leaf samples itself, voxels.cu:449-463 —: that list's tail word too;
		// voxroot_pieces / voxelize_small append behind it.  ADVICE r5: an image built elsewhere with fewer than 50 000 points)
		if (leaf && n->voxelChunks != nullptr) {
			const uint32_t inVox = (n->numVoxelsStored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
			SimlodChunk* v = n->voxelChunks;
			for (uint32_t k = 0; v != nullptr && k < max(inVox, 1u); k++) {
				if (k + 1u >= inVox) { tail_of(n->voxelChunks) = v; break; }
				v = v->next;
			}
		}
		// the ancestors, parent first, zero-terminated
		unsigned long long* rec = at<unsigned long long>(a, a.offPaths) + i * PATH_WORDS;
		const uint32_t L = min(n->level, PATH_WORDS - 1u), s = (uint32_t)SIMLOD_MAX_DEPTH - L;
		const uint32_t X = n->X << s, Y = n->Y << s, Z = n->Z << s;        // (child_index() takes coordinates at full depth)
		uint32_t cur = 0, l = 0;
		for (; l < L; l++) {
			const SimlodNode* anc = a.nodes + cur;
			rec[L - 1u - l] = path_pack(a.pers, cur, anc->level, anc->grid);
			const SimlodNode* c = anc->children[child_index(X, Y, Z, (int)l)];
			if (c == nullptr) break;                                         // (an image whose node is not where its coordinates say: entries below stay as they are, the list ends)
			cur = (uint32_t)(c - a.nodes);
		}
		rec[L] = 0;
	}
}

__global__ __launch_bounds__(TPB) void k_begin(BuildArgs a, uint32_t momentaryTooSmall, uint32_t batchLimit, uint32_t debugFlags, uint32_t budgetUs, uint32_t groupMax) {
	Ctl* ctl = ctl_of(a);
	const bool stale = momentaryTooSmall != 0u || (debugFlags & 4u) != 0u || stamp_is_stale(a, ctl);      // (debugFlags bit 2: the host knows the image was replaced — simlod_octree_image_replaced, a reset)
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		const uint32_t fatal = a.stats->dbg & (SIMLOD_ERR_BARRIER_TIMEOUT | SIMLOD_ERR_DIRECTORY_FULL);   // sticky until the host resets the octree
		ctl->errors = momentaryTooSmall ? SIMLOD_ERR_MOMENTARY_TOO_SMALL : 0u;
		ctl->stop = (momentaryTooSmall || fatal) ? 1u : 0u;
		ctl->abortBatch = 0;
		ctl->processed = 0;
		ctl->consumed = 0;
		ctl->pointsTaken = a.stats->numPointsProcessed;
		ctl->groupMax = groupMax;
		ctl->debugFlags = debugFlags;
		ctl->budgetUs = budgetUs != 0u ? budgetUs : (uint32_t)(SIMLOD_MAX_PROCESSING_MS * 1000.0f);
		ctl->startNs = wall_ns();
		*a.frameStart = ctl->startNs;
		// written concurrently by the upload stream (main_progressive_octree.cpp:1047-1050): device-scope load
		const uint32_t uploaded = __hip_atomic_load(a.numBatchesUploaded, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t first = a.stats->batchletIndex;
		uint32_t n = uploaded - first;
		if ((int32_t)n < 0) n = 0;
		if (n > SIMLOD_MAX_BATCHES_PER_LAUNCH) n = SIMLOD_MAX_BATCHES_PER_LAUNCH;
		if (n > batchLimit) n = batchLimit;
		ctl->uploaded = uploaded;
		ctl->firstBatch = first;
		ctl->numBatches = n;
		for (int i = 0; i < 8; i++) ctl->statCounters[i] = 0;
		ctl->rebuildLeafChunks = stale ? 1u : 0u;
		if (momentaryTooSmall != 0u) ctl->tableMagic = 0;    // (a launch that does nothing: its k_finish leaves no stamp)
		for (uint32_t i = 0; i < BATCH_COPIES; i++) ctl->batch[i].active = 0;
		for (uint32_t i = 0; i < SIMLOD_MAX_BATCHES_PER_LAUNCH; i++) { ctl->voxT[i][0] = ~0ull; ctl->voxT[i][1] = 0; ctl->voxT[i][2] = 0; }
		prepare_batch(a, ctl, 0);
	}
	if (stale && momentaryTooSmall == 0u) rebuild_side_tables(a);
}

// ---- count: leaf lookup + per-leaf arrival counters + spill detection (voxels.cu:124-229) ---------------------
static constexpr uint32_t PPT = 4;                 // points per thread per chunk
static constexpr uint32_t PPB = TPB * PPT;         // points per workgroup chunk

// Split slots.  A leaf that has to split in this batch owns a SLOT: a record of what was reserved for it and a 512-bin histogram
// — the three octree levels below it — of everything that lies in it (its stored points and the batch's samples).  The cascade is
// decided from the histogram alone (k_expand), three levels per round.  Afterwards the same 512 words are the slot's MAP: bin ->
// the node that bin's samples ended up in.  A sample of a slot is relabelled by rewriting its cached-leaf word as
// LEAF_FLAG | slot << 9 | bin: whoever needs its leaf later (k_insert, the next round) reads ONE word of the map.
// the cached-leaf words of a group: one per sample of the group (two copies by the group's parity — k_count of the next group fills
// its copy while k_insert of this one still reads) and one per moved point (one copy: k_hist writes them after k_insert of the group before)
struct LeafWords {
	uint32_t* grp; uint32_t* mov; uint32_t cap;
	__device__ __forceinline__ LeafWords(const BuildArgs& a, uint32_t ordinal)
		: grp(at<uint32_t>(a, a.offLeafOf) + (uint64_t)(ordinal & 1u) * a.leafOfStride), mov(at<uint32_t>(a, a.offLeafOf) + 2 * a.leafOfStride), cap(a.groupCap) {}
	__device__ __forceinline__ uint32_t& operator[](uint32_t i) const { return i < cap ? grp[i] : mov[i - cap]; }     // i: sample of the group, or groupCap + moved point
};
static constexpr uint32_t SLOT_CAP = 2048;                  // slots per batch (12 bits of a relabelled word and of the reservation word)
static constexpr uint32_t HIST_BINS = 512;
// The histograms exist HIST_SHARDS times: a workgroup flushes its LDS counts into copy blockIdx & 3 (consecutive workgroups run on different XCDs), the
// readers (k_expand) add the copies.  A batch's splitting leaves are a dozen, their hot bins a few hundred words, and five hundred workgroups flush
// into them within the same microseconds: a memory-side atomic of k_hist spent ~3 000 cycles in flight against 500-700 in every other kernel
// (profiles/r04, TCC_EA0_ATOMIC_LEVEL / TCC_EA0_ATOMIC: same-address atomics retire one after the other).
// Only the first HIST_SHARDED slots of a batch have the copies — a batch of a stream splits a dozen leaves; a batch that splits hundreds
// (a coalesced group, scattered points) spreads its adds over that many histograms anyway —: 1.5 MB of the momentary buffer instead of 12.
#ifndef HIST_SHARDS_N
#define HIST_SHARDS_N 4          // measured on one box, ms per 36 M ingest: 1 copy 3.69, 4 copies 3.65, 8 copies 3.79 (cycles in flight per memory-side atomic of k_hist: 2 965 / 1 150 / 846)
#endif
static constexpr uint32_t HIST_SHARDS = HIST_SHARDS_N, HIST_SHARDED = 256;
static_assert(HIST_SHARDS >= 1u && (HIST_SHARDS & (HIST_SHARDS - 1u)) == 0u, "a workgroup picks its copy with blockIdx & (HIST_SHARDS - 1)");
static constexpr uint64_t HIST_EXTRA_WORDS = (uint64_t)(HIST_SHARDS - 1u) * HIST_SHARDED * HIST_BINS;      // copies 1..3 of slots 0..255, behind the SLOT_CAP x HIST_BINS words of copy 0
// word of (slot << 9 | bin) in copy `shard`
__device__ __forceinline__ uint64_t hist_word(uint32_t key, uint32_t shard) {
	return (shard == 0u || (key >> 9) >= HIST_SHARDED) ? (uint64_t)key : (uint64_t)SLOT_CAP * HIST_BINS + (uint64_t)(shard - 1u) * HIST_SHARDED * HIST_BINS + key;
}
static constexpr uint32_t LEAF_FLAG = 0x80000000u;          // cached-leaf word: FLAG | slot << 9 | bin   (else: node index | bin below that node << 19, as k_count left it)
static constexpr uint32_t LEAF_BIN_SHIFT = 19;              // node indices travel in 19 bits (simlod_context_set_node_capacity: <= 2^19 nodes)
static constexpr uint32_t LEAF_NODE_MASK = (1u << LEAF_BIN_SHIFT) - 1u;
static constexpr uint32_t MAP_LISTED = 0x80000000u;         // map entry: LISTED | level << 16 | slot of the NEXT round   (else: a node index)
static constexpr uint32_t NONE = 0xffffffffu;
struct SlotRec { uint32_t node, level, childBase, spillBase, stored, born, pad1, pad2; };   // node == NONE: nothing could be reserved, the leaf stays as it is | born: NONE for a leaf that existed when the group began; else (an exact group, a node the cascade queued for its next round) the batch of the group in which the node split — when its children were created
__device__ __forceinline__ SlotRec* slot_recs(const BuildArgs& a, uint32_t ordinal) { return at<SlotRec>(a, a.offSlots) + (uint64_t)(ordinal & 1u) * SLOT_CAP; }   // (by parity, as the clear list)

// the three child choices below a node at `level`, most significant first (levels beyond MAX_DEPTH contribute zero bits)
__device__ __forceinline__ uint32_t bin_of(uint32_t X, uint32_t Y, uint32_t Z, uint32_t level) {
	uint32_t b = 0;
#pragma unroll
	for (uint32_t k = 0; k < 3; k++) {
		const uint32_t lv = level + k;
		b = (b << 3) | (lv < (uint32_t)SIMLOD_MAX_DEPTH ? (uint32_t)child_index(X, Y, Z, (int)lv) : 0u);
	}
	return b;
}

// One arrival-counter update for `cnt` samples (voxels.cu:203-218).  Returns CROSSED for exactly one caller per leaf and batch: the one
// that has to queue the leaf for splitting — whoever sees its counter cross the limit, or, if it is already over the limit because
// an earlier batch could not split it (spill space, node array or slots exhausted: the split is deferred, nothing is lost), whoever
// touches it first in this batch; the exchange on the per-node tag decides.  And FIRST for exactly one caller per leaf and batch too
// (another per-node tag): that caller puts the leaf on the batch's list of leaves with new samples (k_expand allocates their chunks from it).
// Node.numPoints is NOT looked at: the back half of the batch before (k_insert) may still be advancing it.  What the leaf held when this
// batch began — its counter before anybody's add — is the smallest `old` any caller sees: kept per node as tag << 32 | ~old under an
// atomic max (a newer batch's tag beats an older one, a smaller `old` a larger one).
static constexpr uint32_t CROSSED = 1u, FIRST = 2u;
// (in two halves, so that a caller with several leaves — the samples of a thread that found no room in the workgroup's table: a batch scattered
// over thousands of leaves — has all its loads and adds in flight before it looks at any of them)
struct CountPending { uint32_t touchSeen, old; unsigned long long startSeen; };
__device__ __forceinline__ CountPending count_issue(const BuildArgs& a, uint32_t leafIdx, uint32_t cnt) {
	// A batch that is scattered over thousands of leaves (BASELINE config 5: 4 096 leaves, every workgroup meets most of them) makes every workgroup
	// a caller for every leaf: the add is the work, the two tag words are bookkeeping that only the FIRST callers of a leaf change.  So the tag
	// words are looked at with plain loads first (in flight beside the add): a word that already carries this batch's tag — and, for the counter
	// at batch start, a value no larger than this caller's — cannot be changed by this caller, and its atomic is skipped.  A stale read (the
	// XCDs' L2s are not coherent) can only show an OLDER state: the atomic is then issued as before.  Config 5, 200 M points: k_count issued
	// 2.0 M memory-side atomics per batch (250 us) before, 0.73 M after; profiles/r05/config5_*.
	CountPending p;
	p.touchSeen = __hip_atomic_load(at<uint32_t>(a, a.offTouchTag) + leafIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	p.startSeen = __hip_atomic_load(at<unsigned long long>(a, a.offStartOf) + leafIdx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	p.old = atomicAdd(&a.nodes[leafIdx].counter, cnt);
	return p;
}
__device__ __forceinline__ uint32_t count_finish(const BuildArgs& a, const BatchCtl* bc, uint32_t leafIdx, uint32_t cnt, const CountPending& p) {
	SimlodNode* leaf = a.nodes + leafIdx;
	const uint32_t old = p.old;
	const uint32_t before = p.touchSeen == bc->tag ? bc->tag : atomicExch(at<uint32_t>(a, a.offTouchTag) + leafIdx, bc->tag);
	const unsigned long long mine = ((unsigned long long)bc->tag << 32) | (0xffffffffu - old);
	if (p.startSeen < mine) atomicMax(at<unsigned long long>(a, a.offStartOf) + leafIdx, mine);
	uint32_t flags = before != bc->tag ? FIRST : 0u;
	bool over = old + cnt > SIMLOD_MAX_POINTS_PER_NODE;
	if (!over && trunk_any(a)) over = trunk_forced(a, leaf->level, leaf->X, leaf->Y, leaf->Z);      // (a multi-GPU job's shared upper node: splits by the global count)
	// A node at MAX_DEPTH cannot be subdivided (the descent stops there): it keeps growing instead of spilling.
	if (over && leaf->level < SIMLOD_MAX_DEPTH && atomicExch(at<uint32_t>(a, a.offRetryTag) + leafIdx, bc->tag) != bc->tag) flags |= CROSSED;
	return flags;
}
__device__ __forceinline__ uint32_t count_into(const BuildArgs& a, const BatchCtl* bc, uint32_t leafIdx, uint32_t cnt) {
	return count_finish(a, bc, leafIdx, cnt, count_issue(a, leafIdx, cnt));
}
// what a leaf held when batch `tag` began (valid once the batch's k_count is complete)
__device__ __forceinline__ uint32_t stored_at_start(const BuildArgs& a, uint32_t tag, uint32_t leafIdx) {
	const unsigned long long v = at<const unsigned long long>(a, a.offStartOf)[leafIdx];
	return (uint32_t)(v >> 32) == tag ? 0xffffffffu - (uint32_t)v : 0u;
}

struct SpillWork {
	const SimlodChunk* chunk;
	uint32_t slot, dstBase, count, level;
	uint32_t pad0, pad1;
};

// Reserve `slots` split slots, `nodes` node slots and `spill` points of spill space TOGETHER (one 64-bit word: slots << 52 | nodes in use
// << 32 | spill in use), before anything is modified: a leaf that cannot be served now stays a leaf — too full, but intact — and is
// queued again by a later batch.
static constexpr int RSV_SLOT_SHIFT = 52;
__device__ __forceinline__ uint32_t slots_in_use(const BatchCtl* bc) {
	return (uint32_t)(__hip_atomic_load(&bc->reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> RSV_SLOT_SHIFT);
}
__device__ __forceinline__ bool reserve(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, uint32_t slots, uint32_t nodes, uint32_t spill, uint32_t& slotBase, uint32_t& nodeBase, uint32_t& spillBase) {
	// ONE add on the packed word hands out the three ranges; a caller that finds any of them beyond its capacity takes its add back and fails.  While such
	// a failed add stands, every other caller sees that field beyond its capacity too and fails as well (conservatively: the leaf stays as it is and is
	// queued again later) — so a caller that SUCCEEDS never got its ranges on top of amounts that are taken back afterwards.  The fields have room for
	// what can stand at once: a round has at most (group + moved samples) / 50 000 nodes that ask for a slot, and at most one cascade per workgroup
	// (<= 8 x 72 nodes each) that asks for nodes.  (Rounds 2-5: a compare-and-swap loop — forty cascades of a group of batches queued behind each other
	// on this word, a round trip each: 35 of k_expand's 56 us per slot.)
	const unsigned long long inc = ((unsigned long long)slots << RSV_SLOT_SHIFT) + ((unsigned long long)nodes << 32) + spill;
	const unsigned long long old = atomicAdd(&bc->reserve, inc);
	slotBase = (uint32_t)(old >> RSV_SLOT_SHIFT); nodeBase = (uint32_t)(old >> 32) & 0xfffffu; spillBase = (uint32_t)old;
	const bool okSlots = slotBase + slots <= SLOT_CAP, okNodes = nodeBase + nodes <= a.nodeCapacity, okSpill = (unsigned long long)spillBase + spill <= a.spilledCap;
	if (okSlots && okNodes && okSpill) { atomicAdd(&a.stats->numNodes, nodes); return true; }       // voxels.cu:317
	atomicAdd(&bc->reserve, 0ull - inc);
	raise(ctl, !okSlots ? SIMLOD_ERR_SPILLING_OVERFLOW : !okNodes ? SIMLOD_ERR_NODES_EXHAUSTED : SIMLOD_ERR_SPILLED_OVERFLOW);      // more leaves cross the limit at once than a batch has slots for | node array full | spill space
	return false;
}

// the occupancy grid of a node that splits in this batch: allocated if the node has none (voxels.cu:363-365), cleared in any case
// (:371-382, also the root's, which has one from the reset on) — by k_insert, through this list; the grids are first read by k_voxelize
// (two copies of the list, by the batch's parity: k_queue of the next batch fills its list while k_insert of this one is still clearing)
__device__ __forceinline__ SimlodOccupancyGrid** clear_list(const BuildArgs& a, uint32_t ordinal) { return at<SimlodOccupancyGrid*>(a, a.offClear) + (uint64_t)(ordinal & 1u) * a.clearCap; }
__device__ __forceinline__ void note_clear(const BuildArgs& a, const BatchCtl* bc, uint32_t c, SimlodOccupancyGrid* g) {
	if (c < a.clearCap) clear_list(a, bc->ordinal)[c] = g;
	else {                                                        // (never: the list holds a grid per node slot a batch can create)
		uint4* w = reinterpret_cast<uint4*>(g->values);
		for (uint32_t i = 0; i < SIMLOD_GRID_NUM_WORDS / 4; i++) w[i] = make_uint4(0, 0, 0, 0);
	}
}
__device__ __forceinline__ SimlodOccupancyGrid* grid_for_split(const BuildArgs& a, BatchCtl* bc) {
	uint8_t* mem = persistent_alloc(a.pers, sizeof(SimlodOccupancyGrid), 1);          // (two independent atomics with a return value: one round trip)
	const uint32_t c = atomicAdd(&bc->numClear, 1u);
	SimlodOccupancyGrid* g = reinterpret_cast<SimlodOccupancyGrid*>(mem);
	note_clear(a, bc, c, g);
	return g;
}

// Queue leaf `nodeIdx` for splitting: ONE WAVE (k_queue).  Everything the split needs is reserved here, before anything is modified: a slot (and
// with it a histogram), eight node slots and the spill space for the stored points together, the occupancy grid.  Then the leaf's
// chunk list becomes spill-copy work items (chunk k comes from the leaf chunk table, not from a walk) and goes back to the recycle
// stack (voxels.cu:346-357; nothing pops before alloc_points in k_expand).  (voxels.cu:308-383 doSplitting, first half)
// sum over the wave and the sum of the lanes below (every lane of the wave calls)
__device__ __forceinline__ uint32_t wave_exclusive(uint32_t v, uint32_t& total) {
	const uint32_t lane = (uint32_t)lane_id();
	uint32_t x = v;
#pragma unroll
	for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o, 64); if (lane >= o) x += y; }
	total = (uint32_t)__shfl((int)x, 63, 64);
	return x - v;
}
__device__ __forceinline__ unsigned long long shfl64(unsigned long long v, int src) {
	return ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src, 64);
}

// What a crossing leaf needs reserved — a slot, eight node slots, spill space for its stored points, places in the work list and on the recycle
// stack for its chunks — follows from ITS PLACE IN THE LIST: entry e takes slot e, the nodes from 8 e on, and the spill / work / stack ranges behind
// those of the entries before it (a prefix sum over the list, which every wave computes for itself from the leaves' stored counts: plain loads, no
// atomic).  Round 5 reserved per leaf with a compare-and-swap on one word: forty leaves of a group of batches queued behind each other, a round trip
// each (k_queue: 55 us per group of five batches).  An entry is served iff everything up to and including it fits (slots, node array, spill
// space): once one does not, none behind it does — those leaves stay as they are, too full but intact, and are queued again by a later batch.
struct CrossPrefix { uint32_t stored, chunks, storedBefore, chunksBefore; bool ok; };
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
	return v;
}
// entry e of the list: its own numbers and the sums over the entries before it (every lane of the wave calls; all get the result)
__device__ CrossPrefix cross_prefix(const BuildArgs& a, const BatchCtl* bc, const uint32_t* crossList, uint32_t numCross, uint32_t e, uint32_t slots0, uint32_t nodes0, uint32_t spill0) {
	const uint32_t lane = (uint32_t)lane_id();
	CrossPrefix r{0u, 0u, 0u, 0u, false};
	for (uint32_t c0 = 0; c0 <= e; c0 += 64u) {
		const uint32_t idx = c0 + lane;
		uint32_t st = 0, ch = 0;
		if (idx < numCross && idx <= e) {
			const uint32_t leaf = crossList[idx];
			// (not Node.numPoints, which the back half of the group before may still be advancing — and which is reset, with the list's head,
			// by k_expand, after that back half: this kernel runs beside it); between batches stored == counter, so the list holds exactly ceil(stored / 1000) chunks
			st = stored_at_start(a, bc->tag, leaf);
			ch = a.nodes[leaf].points != nullptr ? (st + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK : 0u;
		}
		if (e - c0 < 64u) {        // the chunk that holds entry e
			r.stored = (uint32_t)__shfl((int)st, (int)(e - c0), 64); r.chunks = (uint32_t)__shfl((int)ch, (int)(e - c0), 64);
			r.storedBefore += wave_sum_u32(idx < e ? st : 0u); r.chunksBefore += wave_sum_u32(idx < e ? ch : 0u);
		} else { r.storedBefore += wave_sum_u32(st); r.chunksBefore += wave_sum_u32(ch); }
	}
	r.ok = slots0 + e + 1u <= SLOT_CAP && nodes0 + 8u * (e + 1u) <= a.nodeCapacity && (unsigned long long)spill0 + r.storedBefore + r.stored <= a.spilledCap;
	return r;
}

__device__ void queue_split(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, const uint32_t* crossList, uint32_t numCross, uint32_t e) {
	const uint32_t lane = (uint32_t)lane_id();
	const uint32_t nodeIdx = crossList[e];
	SimlodNode* node = a.nodes + nodeIdx;
	// the leaf's row of the chunk table, a lane per chunk: asked for now, needed after the prefix (one round trip less on a kernel that is a chain of them)
	SimlodChunk* const rowChunk = lane < LEAF_SLOTS ? const_cast<SimlodChunk*>(leaf_row_get(a.mom + a.offLeafChunks, a.pers, nodeIdx, lane)) : nullptr;
	const uint32_t level = node->level;
	SimlodOccupancyGrid* grid = node->grid;
	const unsigned long long base = bc->reserve0;          // as k_count's first workgroup left it (bc->reserve itself: the list's totals, written below by the wave of entry 0)
	const uint32_t slots0 = (uint32_t)(base >> RSV_SLOT_SHIFT), nodes0 = (uint32_t)(base >> 32) & 0xfffffu, spill0 = (uint32_t)base;
	const CrossPrefix cp = cross_prefix(a, bc, crossList, numCross, e, slots0, nodes0, spill0);
	if (!cp.ok) {
		if (lane == 0u) raise(ctl, slots0 + e + 1u > SLOT_CAP ? SIMLOD_ERR_SPILLING_OVERFLOW : nodes0 + 8u * (e + 1u) > a.nodeCapacity ? SIMLOD_ERR_NODES_EXHAUSTED : SIMLOD_ERR_SPILLED_OVERFLOW);
		return;
	}
	const uint32_t slot = slots0 + e, childBase = nodes0 + 8u * e, spillBase = spill0 + cp.storedBefore, stored = cp.stored, numChunks = cp.chunks, w0 = cp.chunksBefore;
	const unsigned long long top = (unsigned long long)bc->acctAlloc0 - cp.chunksBefore;      // Stats.numAllocatedChunks as this leaf's turn finds it (voxels.cu:346-357)
	if (lane == 0u) {
		if (grid == nullptr) {                                                               // voxels.cu:363-365
			SimlodAllocatorGlobal* alloc = reinterpret_cast<SimlodAllocatorGlobal*>(a.pers);
			grid = reinterpret_cast<SimlodOccupancyGrid*>(a.pers + atomicAdd(reinterpret_cast<unsigned long long*>(&alloc->offset), (unsigned long long)SIMLOD_ALLOC_ROUND(sizeof(SimlodOccupancyGrid))));
			node->grid = grid;
		}
		note_clear(a, bc, e, grid);
		slot_recs(a, bc->ordinal)[slot] = SlotRec{nodeIdx, level, childBase, spillBase, stored, NONE, 0u, 0u};
		at<unsigned long long>(a, a.offSplitTag)[nodeIdx] = ((unsigned long long)bc->tag << 32) | (level << 16) | slot;      // (the group's tag: unique while the octree lives, like every tag word)
	}
	// the slot's histogram starts from zero
	{
		for (uint32_t sd = 0; sd < (slot < HIST_SHARDED ? HIST_SHARDS : 1u); sd++) {
			uint4* h = reinterpret_cast<uint4*>(at<uint32_t>(a, a.offHist) + hist_word(slot << 9, sd));
			h[lane] = make_uint4(0, 0, 0, 0); h[lane + 64] = make_uint4(0, 0, 0, 0);
		}
		if (bc->acct != 0u) {      // (an exact group: the slot's histograms per batch)
			uint4* hb = reinterpret_cast<uint4*>(at<uint32_t>(a, a.offHistB) + (uint64_t)slot * HIST_BINS * a.groupMax);
			for (uint32_t i = lane; i < HIST_BINS / 4u * a.groupMax; i += 64u) hb[i] = make_uint4(0, 0, 0, 0);
		}
	}
	SimlodChunk** chunkQueue = at<SimlodChunk*>(a, a.offQueue);
	SpillWork* work = at<SpillWork>(a, a.offWork);
	auto emit = [&](uint32_t ci, SimlodChunk* chunk) {
		if (w0 + ci < a.workCap) {
			SpillWork w;
			w.chunk = chunk; w.slot = slot; w.dstBase = spillBase + ci * SIMLOD_POINTS_PER_CHUNK;
			w.count = min(stored - ci * SIMLOD_POINTS_PER_CHUNK, SIMLOD_POINTS_PER_CHUNK); w.level = level; w.pad0 = 0; w.pad1 = 0;
			work[w0 + ci] = w;
		} else raise(ctl, SIMLOD_ERR_SPILLED_OVERFLOW);
		const unsigned long long q = top - numChunks + ci;
		if (q < CHUNK_QUEUE_CAPACITY) chunkQueue[q] = chunk; else raise(ctl, SIMLOD_ERR_CHUNK_QUEUE_OVERFLOW);
	};
	SimlodChunk* beyond = nullptr;                      // chunk #LEAF_SLOTS of a leaf whose split was deferred and that kept growing
	if (lane == 0 && numChunks > LEAF_SLOTS) beyond = leaf_row_get(a.mom + a.offLeafChunks, a.pers, nodeIdx, LEAF_SLOTS - 1)->next;
	if (lane < min(numChunks, LEAF_SLOTS)) {                          // (LEAF_SLOTS <= 64: one chunk per lane)
		emit(lane, rowChunk);
		rowChunk->next = nullptr;
	}
	if (lane == 0) {
		for (uint32_t ci = LEAF_SLOTS; ci < numChunks && beyond != nullptr; ci++) {   // the table has no slot for these: walk
			SimlodChunk* next = beyond->next;
			emit(ci, beyond);
			beyond->next = nullptr;
			beyond = next;
		}
	}
}
// the list's totals (the wave that takes entry 0): how many entries are served — everything up to the first that does not fit — and what they take
__device__ void queue_totals(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, const uint32_t* crossList, uint32_t numCross) {
	const uint32_t lane = (uint32_t)lane_id();
	const unsigned long long base = bc->reserve0;
	const uint32_t slots0 = (uint32_t)(base >> RSV_SLOT_SHIFT), nodes0 = (uint32_t)(base >> 32) & 0xfffffu, spill0 = (uint32_t)base;
	uint32_t served = 0, stored = 0, chunks = 0;
	bool open = true;
	for (uint32_t c0 = 0; c0 < numCross && open; c0 += 64u) {
		const uint32_t idx = c0 + lane;
		uint32_t st = 0, ch = 0;
		if (idx < numCross) {
			const uint32_t leaf = crossList[idx];
			st = stored_at_start(a, bc->tag, leaf);
			ch = a.nodes[leaf].points != nullptr ? (st + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK : 0u;
		}
		uint32_t totSt, totCh;
		const uint32_t exSt = wave_exclusive(st, totSt), exCh = wave_exclusive(ch, totCh);
		const bool ok = idx < numCross && slots0 + idx + 1u <= SLOT_CAP && nodes0 + 8u * (idx + 1u) <= a.nodeCapacity && (unsigned long long)spill0 + stored + exSt + st <= a.spilledCap;
		// the served entries of this chunk are a prefix of it: the lanes below the first that does not fit
		const unsigned long long okMask = __ballot(ok), valid = __ballot(idx < numCross), firstFail = ~okMask & valid;
		const uint32_t here = firstFail != 0ull ? (uint32_t)__ffsll((long long)firstFail) - 1u : (uint32_t)__popcll(valid);
		served += here;
		stored += here < 64u ? (uint32_t)__shfl((int)exSt, (int)here, 64) : totSt;
		chunks += here < 64u ? (uint32_t)__shfl((int)exCh, (int)here, 64) : totCh;
		if (firstFail != 0ull) open = false;
	}
	if (lane == 0u) {
		bc->reserve = base + ((unsigned long long)served << RSV_SLOT_SHIFT) + ((unsigned long long)(8u * served) << 32) + stored;
		if (served != 0u) atomicAdd(&a.stats->numNodes, 8u * served);                        // voxels.cu:317
		if (chunks != 0u) atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats->numAllocatedChunks), (unsigned long long)(-(long long)chunks));   // voxels.cu:346-357
		bc->numClear = served; bc->numWork = chunks; bc->numSpilled = stored;
	}
}

// ---- exact groups of several batches: the allocator / chunk-pool counters of batch-by-batch ingestion -----------------------------------
// In EXACT mode a launch that finds several pending batches ingests up to groupMax of them as ONE group — one descent, one insert, one voxel
// pass over all their samples, the throughput regime of the chip — where the reference ingests them one after the other (voxels.cu:883-949).
// Topology, per-node sample multisets, occupancy grids, voxel positions and counts do not depend on that granularity (a leaf splits iff what
// lies in its cell exceeds 50 000; a grid holds the cells of every sample that passed through the node).  What does:
//   * Node.counter of an inner node — what lay in its cell after the batch in which it split (voxels.cu:203-218: later batches descend past it);
//   * Stats.numAllocatedChunks' excursions, hence Stats.chunkPoolSize (its high-water mark, voxels.cu:535-537) and how many point chunks came
//     FRESH from the allocator instead of the recycle stack, hence Stats.allocatedBytes_persistent: a node that is created by one batch of the
//     group and split by a later one held chunks in between; a leaf takes its chunks batch by batch.
// Both follow from COUNTS PER BATCH alone: k_count keeps the samples per (leaf, batch of the group), k_hist / k_expand the 512-bin histograms
// per batch, and k_expand derives, for every node of a cascade, the batch it was created in (= the batch its parent split in), the batch it
// split in (the first in which its running count exceeds the limit), its counter at that moment, and the chunks it took and gave back in every
// batch: BatchCtl.acctD[k] / acctF[k] = point chunks batch k of the group takes / returns, summed over all nodes.  account_group() then replays
// voxels.cu:346-357 / 505-516 on those numbers — a stack pointer and a high-water mark — and makes the real counters agree: the chunks the
// batch-by-batch run would have taken fresh beyond what the group did are allocated now and put on the recycle stack, where that run would have
// left them.  The replay must end at the chunk count the octree really has (every chunk in use is in use in both runs): SIMLOD_ERR_ACCOUNTING otherwise.
struct Phantom { uint32_t base, count; unsigned long long mem; };      // recycle-stack entries [base, base + count) = fresh chunks from `mem` on (written by the whole workgroup)
__device__ Phantom account_group(const BuildArgs& a, Ctl* ctl, BatchCtl* g) {
	SimlodStats* s = a.stats;
	Phantom ph{0u, 0u, 0ull};
	if (g != nullptr && g->acct != 0u && g->accounted == 0u && ctl->abortBatch == 0u) {
		long long A = (long long)g->acctAlloc0, P = (long long)g->acctPool0;
		for (uint32_t k = 0; k < g->groupBatches; k++) {
			A -= (long long)g->acctF[k];            // voxels.cu:346-357: the batch's splits return their leaves' chunks first ...
			A += (long long)g->acctD[k];            // ... then its leaves take what they need (voxels.cu:505-516), from the stack while it has any
			if (A > P) P = A;                       // voxels.cu:535-537
		}
		const long long Areal = (long long)s->numAllocatedChunks;
		if (A != Areal) raise(ctl, SIMLOD_ERR_ACCOUNTING);
		const long long Preal = max((long long)g->acctPool0, Areal);          // what the group's own allocations made of the pool
		if (P > Preal && P <= (long long)CHUNK_QUEUE_CAPACITY) {
			ph.base = (uint32_t)Preal; ph.count = (uint32_t)(P - Preal);
			ph.mem = (unsigned long long)persistent_alloc(a.pers, sizeof(SimlodChunk), ph.count);
		}
		s->chunkPoolSize = (uint64_t)max(P, Preal);
		g->accounted = 1u;
	} else if (s->numAllocatedChunks > s->chunkPoolSize) s->chunkPoolSize = s->numAllocatedChunks;      // voxels.cu:535-537
	return ph;
}
__device__ __forceinline__ void phantom_fill(const BuildArgs& a, const Phantom& ph) {      // (every thread of the workgroup)
	SimlodChunk** chunkQueue = at<SimlodChunk*>(a, a.offQueue);
	for (uint32_t i = threadIdx.x; i < ph.count; i += blockDim.x) {
		SimlodChunk* c = reinterpret_cast<SimlodChunk*>(ph.mem + (unsigned long long)i * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
		c->next = nullptr;
		chunkQueue[ph.base + i] = c;
	}
}
// batch of the group that sample i belongs to
__device__ __forceinline__ uint32_t batch_of_sample(const BatchCtl* bc, uint32_t i) {
	uint32_t k = min(i / (uint32_t)SIMLOD_MAX_BATCH_SIZE, bc->groupBatches - 1u);
	while (k + 1u < bc->groupBatches && i >= bc->start[k + 1u]) k++;
	return k;
}
static constexpr uint32_t ACCT_BATCH_SHIFT = 21;       // keys of an exact group's LDS tables: (slot << 9 | bin) | batch << 21  (k_hist, k_expand); leaf | batch << 19 (k_count)
static constexpr uint32_t ACCT_MOVED = 31u;            // "batch" of a stored point a split moves: it was there before the group

// k_count takes more points per thread than k_insert (PPT): its cost is the flush of the per-workgroup counts into a few dozen
// hot leaf counters, and fewer, fatter workgroups mean fewer same-address atomics (measured: 8 -> -3.5 us, in k_insert +14 us)
#ifndef COUNT_CPT
#define COUNT_CPT 8
#endif
#ifndef HIST_CPT
#define HIST_CPT 8
#endif
static constexpr uint32_t CPT = COUNT_CPT;
static constexpr uint32_t HCPT = HIST_CPT, CPB = TPB * HCPT;         // (k_hist)

static constexpr uint32_t TOUCH_CAP = 512;         // leaves one workgroup can be the first to touch in one batch (more: appended one by one)

// The descents of a thread's P samples in lockstep, one level per step, through the child words (KID_*): P independent 4-byte loads in flight per step, and a
// step into a child that its parent's word marks as a leaf ends the descent without a load of its own.
template <int P>
__device__ __forceinline__ void descend_kids(const SimlodNode* nodes, const uint32_t* kid, uint32_t (&cur)[P], uint32_t (&level)[P], const uint32_t (&X)[P], const uint32_t (&Y)[P],
                                             const uint32_t (&Z)[P], bool (&walking)[P]) {
	bool any = true;
#pragma unroll 1
	for (int step = 0; step < SIMLOD_MAX_DEPTH && any; ++step) {
		uint32_t w[P];
#pragma unroll
		for (int j = 0; j < P; j++) w[j] = walking[j] && level[j] < (uint32_t)SIMLOD_MAX_DEPTH ? kid[cur[j]] : 0u;
		any = false;
#pragma unroll
		for (int j = 0; j < P; j++) {
			if (w[j] == 0u) { walking[j] = false; continue; }
			const uint32_t ci = (uint32_t)child_index(X[j], Y[j], Z[j], (int)level[j]);
			if (w[j] == KID_IRREGULAR) {                                      // (an image whose children are not eight consecutive nodes)
				const SimlodNode* c = nodes[cur[j]].children[ci];
				if (c == nullptr) { walking[j] = false; continue; }
				cur[j] = (uint32_t)(c - nodes); level[j] += 1u; any = true;
				continue;
			}
			cur[j] = (w[j] & LEAF_NODE_MASK) + ci; level[j] += 1u;
			if (((w[j] >> (KID_LEAF_SHIFT + ci)) & 1u) != 0u) walking[j] = false; else any = true;
		}
	}
}

template <uint32_t BT, bool SINGLE>
__global__ __launch_bounds__(BT) void k_count(BuildArgs a, uint32_t ordinal) {
	constexpr uint32_t CPB = BT * CPT;
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr) return;
	constexpr uint32_t REP = 8;
	__shared__ SpreadTable<REP> tbl;
	__shared__ uint32_t sh_touch[TOUCH_CAP];
	__shared__ uint32_t sh_numTouch, sh_touchBase;
	const uint32_t n = bc->batchSize;
	const Samples<SINGLE> pts(a, bc);
	const LeafWords leafOf(a, ordinal);
	uint32_t* touched = at<uint32_t>(a, a.offTouched);
	uint32_t* crossList = at<uint32_t>(a, a.offCross);
	const uint32_t numChunks = (n + CPB - 1) / CPB;
	if (blockIdx.x == 0) {
		__shared__ Phantom sh_phantom;
		if (threadIdx.x == 0) {
			// what had to wait for k_expand of the group before: voxels.cu:535-537 — the chunk pool's high-water mark follows that group's
			// allocations, before this one recycles or takes a chunk (an exact group of several batches: account_group) — and the node array's fill,
			// where this group's reservations start (k_queue)
			sh_phantom = account_group(a, ctl, ordinal > 0u ? batch_of(ctl, ordinal - 1u) : nullptr);
			bc->acctAlloc0 = (uint32_t)a.stats->numAllocatedChunks; bc->acctPool0 = (uint32_t)a.stats->chunkPoolSize;
			bc->reserve = (unsigned long long)a.stats->numNodes << 32; bc->reserve0 = bc->reserve;
			if (ordinal == 0u) ctl->tableMagic = 0;     // the octree changes from here on: the side tables' stamp is valid again once k_finish has run (k_begin only reads it)
		}
		__syncthreads();
		phantom_fill(a, sh_phantom);
	}
	const bool acct = bc->acct != 0u;
	uint32_t* cntB = at<uint32_t>(a, a.offCntB);
	const bool trunkPass = blockIdx.x == 0 && trunk_any(a);     // (also for a group without samples: how a host flushes a mask it has just widened)
	if (blockIdx.x >= numChunks && !trunkPass) return;
	Phase ph(ctl, blockIdx.x == 0);
	auto counted = [&](uint32_t leafIdx, uint32_t flags) {
		if ((flags & FIRST) != 0u) {
			const uint32_t k = atomicAdd(&sh_numTouch, 1u);
			if (k < TOUCH_CAP) sh_touch[k] = leafIdx;
			else touched[atomicAdd(&bc->numTouched, 1u)] = leafIdx;       // (at most one entry per node and batch: the list has room for every node)
		}
		if ((flags & CROSSED) != 0u) {                                     // (rare: a handful per batch) k_queue reserves, lists and empties them
			const uint32_t k = atomicAdd(&bc->numCross, 1u);
			if (k < a.crossCap) crossList[k] = leafIdx;
			else { at<uint32_t>(a, a.offRetryTag)[leafIdx] = 0u; raise(ctl, SIMLOD_ERR_SPILLING_OVERFLOW); }   // deferred: a later batch queues it again
		}
	};
	// The LDS table lives for the whole workgroup: no barrier inside the chunk loop, so the waves never wait for each
	// other's slowest descent; one flush at the end.
	spread_init(tbl);
	if (threadIdx.x == 0) sh_numTouch = 0;
	__syncthreads();
	if (trunkPass && threadIdx.x < TRUNK_NODES) {
		// the upper nodes the host's mask names (simlod_context_set_trunk_mask): one that exists and is still a leaf is queued for splitting
		// whether this group has a sample for it or not — "counted" with zero samples, so the exchange on its tag makes ONE caller queue it
		const uint32_t t = threadIdx.x, level = t == 0u ? 0u : t < 9u ? 1u : 2u, code = t == 0u ? 0u : t < 9u ? t - 1u : t - 9u;
		if ((((t < 64u ? a.trunkLo : a.trunkHi) >> (t & 63u)) & 1ull) != 0ull) {
			uint32_t cur = 0;
			for (uint32_t lv = 0; lv < level && cur != NONE; lv++) {
				const SimlodNode* c = a.nodes[cur].children[(code >> (3u * (level - 1u - lv))) & 7u];
				cur = c != nullptr ? (uint32_t)(c - a.nodes) : NONE;
			}
			if (cur != NONE && node_is_leaf(a.nodes + cur)) counted(cur, count_into(a, bc, cur, 0u));
		}
	}
	for (uint32_t chunk = blockIdx.x; chunk < numChunks; chunk += gridDim.x) {
		float4 p[CPT];
		const float4* base; uint32_t kspan;
		const bool oneBatch = pts.span(chunk * CPB, min(n, (chunk + 1u) * CPB) - 1u, base, kspan);      // (workgroup-uniform)
#pragma unroll
		for (uint32_t j = 0; j < CPT; j++) {
			const uint32_t i = chunk * CPB + j * BT + threadIdx.x;
			p[j] = i < n ? (oneBatch ? base[i] : pts[i]) : make_float4(0, 0, 0, 0);
		}
		// the eight descents of a thread in lockstep, one level per step: eight L2 round trips in flight instead of eight chains of 5-8
		// dependent loads one after the other (that was the kernel: 21 us)
		uint32_t X[CPT], Y[CPT], Z[CPT], cur[CPT], level[CPT];
		bool walking[CPT];
#pragma unroll
		for (uint32_t j = 0; j < CPT; j++) {
			X[j] = quantize(F_GRID, p[j].x, a.minx, a.size); Y[j] = quantize(F_GRID, p[j].y, a.miny, a.size); Z[j] = quantize(F_GRID, p[j].z, a.minz, a.size);
			walking[j] = chunk * CPB + j * BT + threadIdx.x < n;
		}
		{   // the descent starts at the deepest node of level <= 5 above the sample (the top table: the eight loads go out together)
			const uint32_t* top = at<const uint32_t>(a, a.offTop);
			uint32_t e[CPT];
#pragma unroll
			for (uint32_t j = 0; j < CPT; j++) e[j] = walking[j] ? top[top_cell(X[j], Y[j], Z[j])] : 0u;
#pragma unroll
			for (uint32_t j = 0; j < CPT; j++) { cur[j] = e[j] & LEAF_NODE_MASK; level[j] = (e[j] >> 19) & 31u; walking[j] = walking[j] && (e[j] & TOP_LEAF) == 0u; }
		}
		descend_kids<(int)CPT>(a.nodes, at<const uint32_t>(a, a.offKid), cur, level, X, Y, Z, walking);
#pragma unroll
		for (uint32_t j = 0; j < CPT; j++) {
			const uint32_t i = chunk * CPB + j * BT + threadIdx.x;
			if (i >= n) continue;
			const uint32_t leafIdx = cur[j];
			leafOf.grp[i] = leafIdx | (bin_of(X[j], Y[j], Z[j], level[j]) << LEAF_BIN_SHIFT);      // the bin is what k_hist needs should this leaf split: it never reads the sample
			// (no room in the workgroup's table — a batch scattered over more leaves than it has keys —: the leaf's counters directly, one sample after the
			// other.  Round 5 measured all of a thread's spilled samples with their loads and adds in flight together: config 5 went from 60 to 71 ms)
			const uint32_t kb = !acct ? 0u : oneBatch ? kspan : batch_of_sample(bc, i);              // (an exact group: counts per leaf AND batch)
			if (!spread_add(tbl, leafIdx | (kb << LEAF_BIN_SHIFT), threadIdx.x & (REP - 1u))) {
				counted(leafIdx, count_into(a, bc, leafIdx, 1u));
				if (acct) atomicAdd(cntB + (uint64_t)leafIdx * a.groupMax + kb, 1u);
			}
		}
	}
	__syncthreads();
	ph.mark(0);
	for (uint32_t e = threadIdx.x; e < (uint32_t)TBL_CAP; e += BT) {
		const uint32_t key = tbl.keys[e];
		if (key == TBL_EMPTY) continue;
		const uint32_t leafIdx = key & LEAF_NODE_MASK, cnt = spread_sum(tbl, e);
		counted(leafIdx, count_into(a, bc, leafIdx, cnt));
		if (acct) atomicAdd(cntB + (uint64_t)leafIdx * a.groupMax + (key >> LEAF_BIN_SHIFT), cnt);
	}
	__syncthreads();
	ph.mark(1);
	// the leaves this workgroup was the first to touch in this batch go on the batch's list (one reservation per workgroup)
	const uint32_t numTouch = min(sh_numTouch, TOUCH_CAP);
	if (threadIdx.x == 0 && numTouch != 0u) sh_touchBase = atomicAdd(&bc->numTouched, numTouch);
	__syncthreads();
	for (uint32_t e = threadIdx.x; e < numTouch; e += BT) touched[sh_touchBase + e] = sh_touch[e];
	if (ph.on) ctl->phaseNs[3] += 1;
}

// ---- queue: the leaves k_count saw cross the limit are reserved, listed and emptied (voxels.cu:308-383 doSplitting, first half) -------------
// One wave per leaf.  Runs beside the back half of the batch BEFORE (whose k_insert may still be storing points into these very leaves): it
// reads what k_count and the allocator know — the counter at batch start, the chunk table — and writes only chunk links, the recycle stack
// and reservations; k_hist, which moves the points, is the kernel that waits for that k_insert.
__global__ __launch_bounds__(TPB) void k_queue(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr || ctl->abortBatch) return;
	const uint32_t numCross = min(bc->numCross, a.crossCap);
	if (numCross == 0u) return;
	// SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT: behave as if k_expand's grid barrier had given up, before anything is modified (tests the abort path)
	if ((ctl->debugFlags & 1u) != 0u) { if (blockIdx.x == 0 && threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }
	const uint32_t* crossList = at<const uint32_t>(a, a.offCross);
	const uint32_t wave = (blockIdx.x * TPB + threadIdx.x) / 64u, numWaves = gridDim.x * TPB / 64u;
	for (uint32_t e = wave; e < numCross; e += numWaves) queue_split(a, ctl, bc, crossList, numCross, e);
	if (wave == 0u) queue_totals(a, ctl, bc, crossList, numCross);
}

// ---- k_voxelize's work items (filled by the chunk allocation below) -----------------------------------------------------
static constexpr uint32_t VTPB = 1024;
#ifndef VOX_SPT_N
#define VOX_SPT_N 8
#endif
static constexpr uint32_t VOX_SPT = VOX_SPT_N;                  // samples per thread, kept in registers across both passes
static constexpr uint32_t VOX_PIECE = VTPB * VOX_SPT;           // 8192 samples per workgroup
static constexpr uint32_t VOX_BIG_ITEMS = 65536;                // entries of the item array for k_voxelize's pieces; the rest: one small item per leaf
static constexpr uint32_t VOX_SMALL = 512;                      // a leaf with fewer new samples than this takes the wave-per-leaf path ...
static constexpr uint32_t VOX_SMALL_PIECE = 128;                // ... in items of at most this many samples (two steps of a wave)
static constexpr uint32_t LDS_LEVELS = 7;                       // ancestors d = 1..7 own a cube of side 128 >> d; from d = 8 on: one cell
static constexpr uint32_t CUBE_WORDS = 8192 + 1024 + 256 + 64 + 16 + 4 + 4;
struct VoxItem { uint32_t leaf, s0, s1, ptBase, ptFirst; };   // samples [s0, s1) of the leaf's storage; its chunk directory; leaf = node index | level << 24   (20 bytes: the leaf's coordinates come from its node)
__device__ __forceinline__ VoxItem* vox_items(const BuildArgs& a, const BatchCtl* bc) { return at<VoxItem>(a, a.offVoxItems) + (uint64_t)(bc->ordinal & 1u) * a.voxItemCap; }

// ---- chunks for the leaves with new samples ----------------------------------------------------------------------------
// The point chunks of the leaves with new samples and their share of k_voxelize's work list (voxels.cu:485-538), for ALLOC_LEAVES entries
// of the batch's list: ONE WORKGROUP.
//   phase 1, wave 0, one leaf per lane: how many chunks, directory entries and work items each leaf needs; the reservations of the wave's
//     leaves — directory entries, chunks off the recycle stack (voxels.cu:505-516), work items, memory for the chunks the stack cannot
//     serve — are summed over the wave and made with ONE atomic each (the words they advance are shared by every leaf of the batch, and
//     device-scope atomics on one word retire at ~88 M/s here);
//   phase 2, all waves, one NEW CHUNK per lane: fetch it (stack or fresh memory), link it, enter it in the chunk directory and the leaf
//     chunk table.  A leaf the batch has filled from nothing needs 50 chunks; taken one after the other by the leaf's lane that was 50
//     dependent round trips (the whole of round 2's allocation kernel: 13 us), taken side by side it is two.
#ifndef ALLOC_WAVES_N
#define ALLOC_WAVES_N 1      /* (8: a wave per 64 leaves — measured: no gain, k_expand's build phase is not bound by this loop; and a fault in plain exact mode with the trunk mask that was not understood) */
#endif
static constexpr uint32_t ALLOC_WAVES = ALLOC_WAVES_N, ALLOC_LEAVES = 64 * ALLOC_WAVES;      // leaves per call: a wave per 64 (a group of several batches hands a cascade's 584 nodes over at once: one wave took nine turns, 45 us of k_expand's 61 per slot)
struct AllocRec {
	uint32_t node, existing, additional, fromPool;
	uint32_t dirNew, prefix;                                // directory entry of the leaf's first new chunk | new chunks of the lanes below
	unsigned long long firstIdx, mem;                       // recycle-stack index of the first new chunk | memory of the first one the stack could not serve
	SimlodChunk* head; SimlodChunk* tail;                   // the list as it is (nullptr: empty)
};
struct AllocShared { uint32_t waveTotal[ALLOC_WAVES]; };
struct FreshLeaf { uint32_t node, samples, level, X, Y, Z; };      // a leaf a cascade has just made: what alloc_points would otherwise read back from the node it was written to a moment ago

// Entry k is taken when firstEntry + lane < numEntries.  `touched` (global memory): the leaves k_count found new samples for — what
// each held when the batch began comes from stored_at_start().  `fresh` (LDS): {node, samples} of the empty leaves a cascade has
// just made (nothing about them has to be read back).  One of the two lists is given.
// (rec: ALLOC_LEAVES records in LDS — k_expand lends the words of its hash table)
__device__ void alloc_points(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, AllocShared& sh, AllocRec* rec, const uint32_t* touched, const FreshLeaf* fresh, uint32_t firstEntry, uint32_t numEntries) {
	const bool fresh_leaves = fresh != nullptr;
	NodeDir* nodeDir = at<NodeDir>(a, a.offNodeDir);
	SimlodChunk** chunkDir = chunk_dir(a, bc);
	SimlodChunk** chunkQueue = at<SimlodChunk*>(a, a.offQueue);
	uint8_t* const leafChunks = a.mom + a.offLeafChunks;
	if (threadIdx.x < 64u * ALLOC_WAVES) {
		const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
		firstEntry += wv * 64u;                                         // (this wave's 64 entries)
		// (Node.numPoints is not looked at: the back half of the batch before may still be advancing it)
		const unsigned long long pool = a.stats->chunkPoolSize;         // raised only by prepare_batch, between the groups' allocations (asked for here: in flight beside everything below)
		FreshLeaf fl = FreshLeaf{NONE, 0u, 0u, 0u, 0u, 0u};
		if (firstEntry + lane < numEntries) { if (fresh_leaves) fl = fresh[firstEntry + lane]; else fl.node = touched[firstEntry + lane]; }
		const uint2 entry = make_uint2(fl.node, fl.samples);
		const uint32_t i = entry.x, stored = (fresh_leaves || i == NONE) ? 0u : stored_at_start(a, bc->tag, i);
		SimlodNode* node = a.nodes + (i != NONE ? i : 0u);
		uint32_t counter = 0;
		bool need = false;
		SimlodChunk* head = nullptr;
		if (i != NONE && fresh_leaves) { counter = entry.y; need = counter != 0u; }
		else if (i != NONE) {
			// (a leaf that k_count's tail has queued for splitting still looks like a leaf until k_expand gives it children: not this one's business)
			const bool queued = (uint32_t)(at<const unsigned long long>(a, a.offSplitTag)[i] >> 32) == bc->tag;
			counter = node->counter; head = node->points; need = !queued && stored < counter && node_is_leaf(node);
		}
		if (bc->acct != 0u && !fresh_leaves) {
			// An exact group of several batches: a leaf takes its chunks batch by batch (voxels.cu:485-538) — batch k of the group brings it from
			// ceil(count before / 1000) to ceil(count after / 1000) chunks; k_count kept the leaf's samples per batch.  Summed over the wave, one add per
			// batch (account_group replays them).  The row is left as it was found: zero.  (The nodes of a cascade: k_expand, from the histograms.)
			uint32_t* row = at<uint32_t>(a, a.offCntB) + (uint64_t)(i != NONE ? i : 0u) * a.groupMax;
			uint32_t c = stored, have = (stored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
			for (uint32_t k = 0; k < bc->groupBatches; k++) {
				uint32_t delta = 0;
				if (i != NONE) {
					c += row[k]; row[k] = 0u;
					const uint32_t want = (c + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
					if (need) delta = want - have;
					have = want;
				}
				uint32_t sum;
				(void)wave_exclusive(delta, sum);
				if (lane == 0u && sum != 0u) atomicAdd(&bc->acctD[k], sum);
			}
		}
		const uint32_t required = (counter + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
		const uint32_t existing = need ? (stored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK : 0u;
		const uint32_t first = stored / SIMLOD_POINTS_PER_CHUNK;       // chunk that receives slot `stored`
		const uint32_t entries = need ? required - first : 0u;
		const uint32_t additional = need ? required - existing : 0u;
		const uint32_t fresh = need ? counter - stored : 0u;
		SimlodChunk* tail = existing > 0u ? tail_of(head) : nullptr;
		if (existing == 0u) head = nullptr;
		// the leaf's new samples [stored, counter) are k_voxelize's work, in pieces one workgroup takes — or, when they are few,
		// voxelize_small's, one wave per leaf (big items: the first VOX_BIG_ITEMS entries of the item array — a piece has at least
		// VOX_SMALL samples or is the last of its leaf, so they cover 33 M samples; small items behind them: one per leaf at most)
		const uint32_t pieces = fresh < VOX_SMALL ? 0u : (fresh + VOX_PIECE - 1) / VOX_PIECE;
		if (i == 0u && pieces != 0u) bc->rootPieces = pieces;          // (the root as a leaf: k_voxroot's pieces)
		const uint32_t small = need && pieces == 0u ? (fresh + VOX_SMALL_PIECE - 1) / VOX_SMALL_PIECE : 0u;
		uint32_t totEntries, totAdditional, totPieces, totSmall;
		const uint32_t exEntries = wave_exclusive(entries, totEntries), exAdditional = wave_exclusive(additional, totAdditional);
		const uint32_t exPieces = wave_exclusive(pieces, totPieces), exSmall = wave_exclusive(small, totSmall);
		uint32_t dirBase = 0, itemBase = 0, smallBase = 0;
		unsigned long long chunkBase = 0;
		if (lane == 0u) {                                              // (four independent atomics with a return value: one round trip)
			if (totEntries != 0u) dirBase = atomicAdd(&bc->dirCount, totEntries);
			if (totAdditional != 0u) chunkBase = atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats->numAllocatedChunks), (unsigned long long)totAdditional);
			if (totPieces != 0u) itemBase = atomicAdd(&bc->numVoxItems, totPieces);
			if (totSmall != 0u) smallBase = atomicAdd(&bc->numVoxSmall, totSmall);
			sh.waveTotal[wv] = totAdditional;
		}
		dirBase = (uint32_t)__shfl((int)dirBase, 0, 64); itemBase = (uint32_t)__shfl((int)itemBase, 0, 64); smallBase = (uint32_t)__shfl((int)smallBase, 0, 64);
		chunkBase = shfl64(chunkBase, 0);
		// pop from the recycle stack, allocate what the stack cannot serve
		const unsigned long long firstIdx = chunkBase + exAdditional;
		const uint32_t fromPool = firstIdx >= pool ? 0u : (uint32_t)min((unsigned long long)additional, pool - firstIdx);
		uint32_t totNew;
		const uint32_t exNew = wave_exclusive(additional - fromPool, totNew);
		unsigned long long mem = 0;
		if (lane == 0u && totNew != 0u) mem = (unsigned long long)persistent_alloc(a.pers, sizeof(SimlodChunk), totNew);
		mem = shfl64(mem, 0) + (unsigned long long)exNew * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk));
		const uint32_t base = dirBase + exEntries;
		const uint32_t itemAt = pieces != 0u ? itemBase + exPieces : VOX_BIG_ITEMS + smallBase + exSmall;
		bool ok = need;
		if (need && base + entries > a.dirCap) { panic(ctl, SIMLOD_ERR_DIRECTORY_FULL); ok = false; }
		if (need && (pieces != 0u ? itemAt + pieces > VOX_BIG_ITEMS : itemAt + small > a.voxItemCap)) { panic(ctl, SIMLOD_ERR_DIRECTORY_FULL); ok = false; }   // a batch + moved points beyond 33 M samples
		uint32_t e = 0;
		if (ok) {
			if (first < existing) chunkDir[base + e++] = tail;          // the partially filled tail chunk
			NodeDir& d = nodeDir[i];
			d.ptBase = base; d.ptFirst = first; d.ptTag = bc->tag;
			VoxItem* items = vox_items(a, bc);
			const uint32_t nl = fresh_leaves ? fl.level : node->level;
			if (pieces == 0u) for (uint32_t q = 0; q < small; q++) items[itemAt + q] = VoxItem{i | nl << 24, stored + q * VOX_SMALL_PIECE, min(stored + (q + 1u) * VOX_SMALL_PIECE, counter), base, first};
			else for (uint32_t q = 0; q < pieces; q++) items[itemAt + q] = VoxItem{i | nl << 24, stored + q * VOX_PIECE, min(stored + (q + 1u) * VOX_PIECE, counter), base, first};
		}
		AllocRec& r = rec[threadIdx.x];
		r.node = i; r.existing = existing; r.additional = ok ? additional : 0u; r.fromPool = fromPool; r.dirNew = base + e; r.prefix = exAdditional;
		r.firstIdx = firstIdx; r.mem = mem; r.head = head; r.tail = tail;
	}
	__syncthreads();
	// phase 2: new chunk q of the workgroup = chunk k of the leaf whose prefix covers q
	uint32_t total = 0;
	for (uint32_t w = 0; w < ALLOC_WAVES; w++) total += sh.waveTotal[w];
	for (uint32_t q0 = threadIdx.x; q0 < total; q0 += blockDim.x) {
		uint32_t q = q0, wv = 0;                                         // the wave whose leaves new chunk q0 belongs to, and its number among that wave's
		while (wv + 1u < ALLOC_WAVES && q >= sh.waveTotal[wv]) { q -= sh.waveTotal[wv]; wv++; }
		uint32_t lo = wv * 64u, hi = lo + 64u;                           // the last leaf with prefix <= q (leaves without new chunks share their successor's prefix)
		while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (rec[mid].prefix <= q) lo = mid; else hi = mid; }
		const AllocRec r = rec[lo];
		const uint32_t k = q - r.prefix;
		if (k >= r.additional) continue;                                 // (a leaf that could not be served: its reservation stays unused)
		auto chunk_at = [&](uint32_t j) -> SimlodChunk* {
			return j < r.fromPool ? chunkQueue[r.firstIdx + j] : reinterpret_cast<SimlodChunk*>(r.mem + (unsigned long long)(j - r.fromPool) * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
		};
		SimlodChunk* c = chunk_at(k);
		SimlodChunk* next = k + 1u < r.additional ? chunk_at(k + 1u) : nullptr;
		SimlodChunk* first = r.head != nullptr ? r.head : (k == 0u ? c : (k + 1u == r.additional ? chunk_at(0u) : nullptr));
		c->next = next;
		chunkDir[r.dirNew + k] = c;
		if (r.existing + k < LEAF_SLOTS) leaf_row_set(leafChunks, a.pers, r.node, r.existing + k, c);
		if (k == 0u) { if (r.tail == nullptr) a.nodes[r.node].points = c; else r.tail->next = c; }
		if (k + 1u == r.additional) tail_of(first) = c;
	}
}

// ---- hist: round 0 of the split cascade's histograms (voxels.cu:245-289) ---------------------------------------------------------
// Every CU takes part (an ordinary launch; the rounds that follow, if any, run inside k_expand).  Index space: first the stored points
// of the queued leaves — element e = point (e % 1000) of work item (e / 1000), one chunk per item: they move into the spill buffer
// (voxels.cu:253-289) — then the batch's samples.  Whatever lies in a queued leaf is added to the leaf's histogram (per workgroup in LDS
// first, one global add per workgroup and bin) and its cached-leaf word is relabelled FLAG | slot | bin.  Exits at once when k_queue
// queued nothing.
__global__ __launch_bounds__(TPB) void k_hist(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr || ctl->abortBatch) return;
	const uint32_t n = bc->batchSize, tag = bc->tag, numWork = bc->numWork;      // (asked for beside the slot count: one round trip, not two)
	const uint32_t slots0 = slots_in_use(bc);
	if (blockIdx.x == 0 && threadIdx.x == 0) bc->slotsRound0 = slots0;
	if (blockIdx.x + 1u == gridDim.x && threadIdx.x == 0) prepare_batch(a, ctl, ordinal + 1u);     // (a workgroup without samples, as a rule)
	if (slots0 == 0u) return;
	__shared__ BlockTable tbl;
	const LeafWords leafOf(a, ordinal);
	const unsigned long long* slotOf = at<const unsigned long long>(a, a.offSplitTag);   // per node: batch tag << 32 | level << 16 | slot
	uint32_t* hist = at<uint32_t>(a, a.offHist);
	const uint32_t shard = blockIdx.x & (HIST_SHARDS - 1u);                                    // this workgroup's copy
	const SpillWork* work = at<const SpillWork>(a, a.offWork);
	float4* spilled = at<float4>(a, a.offSpilled);
	const uint32_t moved = min(numWork, a.workCap) * SIMLOD_POINTS_PER_CHUNK;
	const uint32_t total = moved + n;
	const uint32_t numChunks = (total + CPB - 1) / CPB;
	if (blockIdx.x >= numChunks) return;
	Phase ph(ctl, blockIdx.x == 0);
	table_init(tbl);
	__syncthreads();
	// An exact group of several batches (acct) keeps a slot's histogram PER BATCH: a sample of batch k of the group goes into histB[(slot, bin)][k], a
	// stored point that moves — it was there before the group — into the plain histogram; a bin's count is the sum of both (bin_total).  The LDS
	// table's key carries the batch above the (slot, bin) bits.
	const bool acct = bc->acct != 0u;
	uint32_t* histB = at<uint32_t>(a, a.offHistB);
	const uint32_t GB = a.groupMax;
	auto flush = [&](uint32_t key, uint32_t cnt) {
		const uint32_t kb = key >> ACCT_BATCH_SHIFT, k21 = key & 0x1fffffu;
		if (acct && kb != ACCT_MOVED) atomicAdd(histB + (uint64_t)k21 * GB + kb, cnt);
		else atomicAdd(hist + hist_word(k21, shard), cnt);
	};
	auto add = [&](uint32_t key, uint32_t kb) {
		uint32_t rank;
		if (acct) key |= kb << ACCT_BATCH_SHIFT;
		if (table_add(tbl, key, 1u, &rank) < 0) flush(key, 1u);
	};
	const Samples<false> grp(a, bc);
	for (uint32_t chunk = blockIdx.x; chunk < numChunks; chunk += gridDim.x) {
		// (an exact group: which batch of the group the tile's samples belong to — one lookup per tile where the tile lies in one batch)
		const float4* unusedBase; uint32_t kspan = 0;
		const uint32_t e0 = max(chunk * CPB, moved), e1 = min(total, (chunk + 1u) * CPB);
		const bool oneBatch = acct && e0 < e1 && grp.span(e0 - moved, e1 - 1u - moved, unusedBase, kspan);      // (workgroup-uniform)
		// stage by stage, eight elements per thread.  A SAMPLE of the group is never read here: its word holds the leaf k_count found and the
		// bin below that leaf (node | bin << 19) — if the leaf was queued, the word becomes FLAG | slot | bin and the bin is counted.  A STORED
		// point of a queued leaf is read, binned and moved to the spill buffer.
		uint32_t ent[HCPT], dst[HCPT], v[HCPT];              // moved points: ent = level << 16 | slot (or NONE), dst = spill index; samples: v = the cached-leaf word (or NONE)
		const float4* src[HCPT];
#pragma unroll
		for (uint32_t j = 0; j < HCPT; j++) {
			const uint32_t e = chunk * CPB + j * TPB + threadIdx.x;
			ent[j] = NONE; src[j] = nullptr; dst[j] = 0; v[j] = NONE;
			if (e < moved) {
				const SpillWork item = work[e / SIMLOD_POINTS_PER_CHUNK];
				const uint32_t k = e % SIMLOD_POINTS_PER_CHUNK;
				if (k < item.count) { ent[j] = (item.level << 16) | item.slot; src[j] = reinterpret_cast<const float4*>(item.chunk->points) + k; dst[j] = item.dstBase + k; }
			} else if (e < total) v[j] = leafOf.grp[e - moved];
		}
		unsigned long long info[HCPT];
#pragma unroll
		for (uint32_t j = 0; j < HCPT; j++) info[j] = v[j] != NONE ? slotOf[v[j] & LEAF_NODE_MASK] : 0ull;
		float4 p[HCPT];
#pragma unroll
		for (uint32_t j = 0; j < HCPT; j++) p[j] = ent[j] != NONE ? *src[j] : make_float4(0, 0, 0, 0);
#pragma unroll
		for (uint32_t j = 0; j < HCPT; j++) {
			if (v[j] == NONE || (uint32_t)(info[j] >> 32) != tag) continue;
			const uint32_t key = (((uint32_t)info[j] & 0xffffu) << 9) | (v[j] >> LEAF_BIN_SHIFT);
			const uint32_t i = chunk * CPB + j * TPB + threadIdx.x - moved;
			leafOf.grp[i] = LEAF_FLAG | key;
			add(key, !acct ? 0u : oneBatch ? kspan : batch_of_sample(bc, i));
		}
#pragma unroll
		for (uint32_t j = 0; j < HCPT; j++) {
			if (ent[j] == NONE) continue;
			const uint32_t X = quantize(F_GRID, p[j].x, a.minx, a.size), Y = quantize(F_GRID, p[j].y, a.miny, a.size), Z = quantize(F_GRID, p[j].z, a.minz, a.size);
			const uint32_t key = ((ent[j] & 0xffffu) << 9) | bin_of(X, Y, Z, ent[j] >> 16);
			spilled[dst[j]] = p[j]; leafOf.mov[dst[j]] = LEAF_FLAG | key;                       // a stored point moves
			add(key, ACCT_MOVED);
		}
	}
	__syncthreads();
	ph.mark(4);
	for (uint32_t e = threadIdx.x; e < (uint32_t)TBL_CAP; e += TPB) {
		const uint32_t key = tbl.keys[e];
		if (key != TBL_EMPTY) flush(key, tbl.vals[e]);
	}
	ph.mark(5);
	if (ph.on) ctl->phaseNs[6] += 1;
}

// ---- expand: split the queued leaves, cascades included (voxels.cu:385-415, 245-289, 308-383) ---------------------------
// Persistent, hand-rolled grid barrier; exits at once when `count` queued nothing.  The reference counts, splits ONE level, counts
// again, ... with ~8 grid.sync() per level.  Here a round settles THREE levels:
//   H) all workgroups: the stored points of the round's slots move to the spill buffer (work items, round 0 only) and every batch
//      sample / moved point that lies in a slot's node is added to the slot's 512-bin histogram (LDS first, one global add per
//      workgroup and bin); its cached-leaf word becomes FLAG | slot | bin;
//   -- barrier --
//   D) one workgroup per slot: from the histogram alone, which children, grandchildren hold more than 50 000 and split in turn;
//      all their nodes at once (counters filled in, grids allocated, ancestor paths, parents), the slot's map; great-grandchildren
//      that are still too full get a slot of their own for the next round.
//   Every workgroup can tell from the histograms whether a next round is possible; if not, the kernel ends after D without
//   another barrier (the common case: one barrier per batch).
static constexpr uint32_t ETPB = 1024;             // k_expand: at most one workgroup per CU (grid barrier participants), 16 waves each
static constexpr int HT_BITS = 12;
static constexpr uint32_t HT_CAP = 1u << HT_BITS;
static constexpr uint32_t LOCAL_NODES = 8 + 64 + 512;          // nodes a slot can create: local numbering t = 0..7 | 8..71 | 72..583

__device__ __forceinline__ uint32_t hist_sum(const uint32_t* hist, uint64_t word) {      // a bin's count: the sum of its copies
	uint32_t v = hist[word];
	if ((word >> 9) < HIST_SHARDED) {
#pragma unroll
		for (uint32_t sd = 1; sd < HIST_SHARDS; sd++) v += hist[hist_word((uint32_t)word, sd)];
	}
	return v;
}

// ... of an exact group of several batches (acct): the stored points that moved (the plain histogram) + the group's samples, kept per batch
__device__ __forceinline__ uint32_t bin_total(const BuildArgs& a, const uint32_t* hist, uint64_t word, bool acct) {
	uint32_t v = hist_sum(hist, word);
	if (acct) {
		const uint32_t* hb = at<const uint32_t>(a, a.offHistB) + word * a.groupMax;
		for (uint32_t k = 0; k < a.groupMax; k++) v += hb[k];
	}
	return v;
}
static constexpr uint32_t ACCT_MAX_GROUP = 12;     // batches an exact group can have: k_expand keeps a slot's 512 + 64 + 8 per-batch rows in the LDS words of its hash table
static constexpr uint32_t NEVER = 0xffu;

struct ExpandShared {
	union {
		struct { uint32_t keys[HT_CAP], vals[HT_CAP]; };      // H: (slot << 9 | bin) -> count.  D, exact groups: rows of per-batch counts (keys and vals as one array: acct_rows)
		AllocRec allocRec[ALLOC_LEAVES];                       // ... and, when those are done with, alloc_points' records
	};
	uint32_t bins[HIST_BINS], c2[64], c1[8];
	uint32_t base2[8], base3[64];                  // first child of split child j / grandchild jk
	uint32_t listed[LOCAL_NODES];                  // map entry override of a node that got a slot for the next round, or NONE
	FreshLeaf fresh[LOCAL_NODES];                  // the cascade's nodes that hold samples — they get their chunks before the kernel ends
	uint32_t LX, LY, LZ;                           // the slot node's own coordinates, name and grid, read once per slot
	uint8_t nameL[20];
	SimlodOccupancyGrid* gridL;
	uint32_t numFresh, numFill;
	// new leaves at level <= 3 whose cells of the top table the whole workgroup fills: {node | level << 19, X, Y, Z}.  The ROOT's cascade can make all of its
	// 8 + 64 + 512 nodes such leaves (rounds 3-6 had 72 entries here: a terrain's first batch lists ~120, the entries beyond the array landed in whatever
	// followed it — alloc_points' records while those stood there, harmlessly; the grids' pointers once they did not: a wild pointer in a path entry, a fault
	// in k_voxelize once in a hundred ingests)
	uint4 fill[LOCAL_NODES];
	AllocShared alloc;
	SimlodOccupancyGrid* grid[8 + 64];             // grids of the children / grandchildren that split here
	unsigned long long pathL[PATH_WORDS];          // the slot node's own ancestor path
	uint32_t mask1, extraBase, ok, more;
	unsigned long long mask2;
	// exact groups of several batches (acct): per local node, the batch of the group in which it split (NEVER: it did not) and its counter after that batch;
	// [LOCAL_NODES] = the slot's own node.  The upper nodes the trunk mask names (they split as soon as they exist).  Chunks taken / returned per batch.
	uint8_t splitAt[LOCAL_NODES + 8];
	uint32_t counterAt[LOCAL_NODES + 1];
	uint32_t forced1;
	unsigned long long forced2;
	uint32_t accD[SIMLOD_MAX_BATCHES_PER_LAUNCH], accF[SIMLOD_MAX_BATCHES_PER_LAUNCH];
};
static_assert(offsetof(ExpandShared, vals) == offsetof(ExpandShared, keys) + sizeof(uint32_t) * HT_CAP && (HIST_BINS + 64u + 8u) * ACCT_MAX_GROUP <= 2u * HT_CAP, "acct_rows");

// one table entry of k_expand's in-kernel histogram pass -> global memory.  key: slot << 9 | bin, and in an exact group (acct) above them the batch the
// samples belong to (ACCT_MOVED: stored points that moved): a sample of batch k counts in histB[(slot, bin)][k], a moved point in the plain histogram
__device__ __forceinline__ void hist_flush(const BuildArgs& a, uint32_t key, uint32_t cnt, bool acct) {
	const uint32_t kb = key >> ACCT_BATCH_SHIFT, k21 = key & 0x1fffffu;
	if (acct && kb != ACCT_MOVED) atomicAdd(at<uint32_t>(a, a.offHistB) + (uint64_t)k21 * a.groupMax + kb, cnt);
	else atomicAdd(at<uint32_t>(a, a.offHist) + hist_word(k21, blockIdx.x & (HIST_SHARDS - 1u)), cnt);
}
__device__ __forceinline__ void hist_add(const BuildArgs& a, ExpandShared& sh, uint32_t key, uint32_t cnt, bool acct) {
	uint32_t h = (key * 2654435761u) >> (32 - HT_BITS);
#pragma unroll 1
	for (int probe = 0; probe < 16; ++probe) {
		uint32_t k = sh.keys[h];
		if (k == TBL_EMPTY) { k = atomicCAS(&sh.keys[h], TBL_EMPTY, key); if (k == TBL_EMPTY) k = key; }
		if (k == key) { atomicAdd(&sh.vals[h], cnt); return; }
		h = (h + 1) & (HT_CAP - 1);
	}
	hist_flush(a, key, cnt, acct);          // no room in the table: straight to the histogram (this workgroup's copy)
}
// local node number t of a slot -> depth below the slot's node (1..3) and the octants chosen on the way
__device__ __forceinline__ uint32_t local_depth(uint32_t t) { return t < 8u ? 1u : t < 72u ? 2u : 3u; }

#ifndef EXPAND_U
#define EXPAND_U 8
#endif
__global__ __launch_bounds__(ETPB) void k_expand(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr || ctl->abortBatch) return;
	__shared__ ExpandShared sh;
	static_assert(sizeof(AllocRec) * ALLOC_LEAVES <= sizeof(uint32_t) * 2u * HT_CAP, "alloc_points' records in the hash table's words");
	{
		// The chunks of the leaves that k_count found new samples for and that do not split (voxels.cu:485-538 allocatePointChunks; the nodes of
		// a cascade get theirs below, from the workgroup that builds them): 64 leaves per list, list #k to the k-th workgroup FROM THE END —
		// the first workgroups are the ones that build the cascades — beside the cascade and off k_insert's path.
		const uint32_t numTouched = min(bc->numTouched, a.nodeCapacity);
		const uint32_t allocBlocks = (numTouched + ALLOC_LEAVES - 1) / ALLOC_LEAVES;
		for (uint32_t blk = gridDim.x - 1u - blockIdx.x; blk < allocBlocks; blk += gridDim.x) {
			__syncthreads();
			alloc_points(a, ctl, bc, sh.alloc, sh.allocRec, at<const uint32_t>(a, a.offTouched), nullptr, blk * ALLOC_LEAVES, numTouched);
		}
		__syncthreads();
	}
	if (bc->slotsRound0 == 0u) return;         // slots handed out by k_count's tail, as k_hist found them: stable while this kernel hands out more
	// SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT: behave as if the grid barrier had given up (tests the abort path); either way the octree is
	// not to be trusted any more (k_count's tail has already emptied the queued leaves): fatal, sticky until a reset
	if ((ctl->debugFlags & 1u) != 0u) { if (threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }

	const LeafWords leafOf(a, ordinal);
	uint32_t* parentOf = at<uint32_t>(a, a.offParent);
	uint32_t* kidOf = at<uint32_t>(a, a.offKid);
	unsigned long long* paths = at<unsigned long long>(a, a.offPaths);
	SlotRec* slots = slot_recs(a, ordinal);
	uint32_t* hist = at<uint32_t>(a, a.offHist);
	uint32_t* map = at<uint32_t>(a, a.offMap);        // (not the histogram's words: other workgroups may still be peeking at those)
	const float4* spilled = at<const float4>(a, a.offSpilled);
	const Samples<false> pts(a, bc);
	const uint32_t n = bc->batchSize;
	uint32_t generation = 0;
	const bool acct = bc->acct != 0u;                      // an exact group of several batches: histograms per batch, counters and chunk accounting as batch-by-batch ingestion leaves them (account_group)
	const uint32_t GB = bc->groupBatches, GBS = a.groupMax;      // batches of the group | stride of the per-batch rows

	const bool timer = SIMLOD_MEASURE != 0 && blockIdx.x == 0 && threadIdx.x == 0;
	if (timer) ctl->expandNs[6] += 1;

	uint32_t sb = 0, se = min(bc->slotsRound0, SLOT_CAP);
	for (uint32_t round = 0; round < SIMLOD_MAX_EXPAND_ROUNDS && sb < se; ++round) {
		uint64_t t0 = timer ? wall_ns() : 0, t1;
		// -- H: histograms (round 0: k_hist has built them) --------------------------------------------------------------------------
		if (round > 0) {
			for (uint32_t i = threadIdx.x; i < HT_CAP; i += ETPB) { sh.keys[i] = TBL_EMPTY; sh.vals[i] = 0u; }
			__syncthreads();
		}
		if (round > 0) {
			// the batch's samples and the moved points, eight per thread at a time, stage by stage: the cached-leaf words are in flight
			// together, then the map words of those that were relabelled, then the points of those whose node was queued again
			const uint32_t stride = gridDim.x * ETPB;
			const uint32_t total = n + min(bc->numSpilled, a.spilledCap);
			constexpr uint32_t U = EXPAND_U;
			for (uint32_t first = blockIdx.x * ETPB + threadIdx.x; first < total; first += U * stride) {
				uint32_t idx[U], v[U], ent[U];
				float4 p[U];
#pragma unroll
				for (uint32_t q = 0; q < U; q++) {
					const uint32_t t = first + q * stride;
					idx[q] = t < n ? t : a.groupCap + (t - n);
					v[q] = t < total ? leafOf[idx[q]] : 0u;
				}
#pragma unroll
				for (uint32_t q = 0; q < U; q++) {
					// ent = level << 16 | slot of the round's slot this sample lies in, or NONE
					ent[q] = NONE;
					if ((v[q] & LEAF_FLAG) == 0u) continue;
					const uint32_t e = map[v[q] & 0x1fffffu];                         // the map of an earlier round
					if ((e & MAP_LISTED) != 0u) ent[q] = e & 0x7fffffffu;
				}
#pragma unroll
				for (uint32_t q = 0; q < U; q++) {
					const uint32_t t = first + q * stride;
					p[q] = ent[q] != NONE ? (t < n ? pts[t] : spilled[t - n]) : make_float4(0, 0, 0, 0);
				}
#pragma unroll
				for (uint32_t q = 0; q < U; q++) {
					if (ent[q] == NONE) continue;
					const uint32_t X = quantize(F_GRID, p[q].x, a.minx, a.size), Y = quantize(F_GRID, p[q].y, a.miny, a.size), Z = quantize(F_GRID, p[q].z, a.minz, a.size);
					const uint32_t key = ((ent[q] & 0xffffu) << 9) | bin_of(X, Y, Z, ent[q] >> 16);
					leafOf[idx[q]] = LEAF_FLAG | key;
					const uint32_t t = first + q * stride;
					hist_add(a, sh, acct ? key | ((t < n ? batch_of_sample(bc, t) : ACCT_MOVED) << ACCT_BATCH_SHIFT) : key, 1u, acct);
				}
			}
		}
		if (round > 0) {
			__syncthreads();
			for (uint32_t e = threadIdx.x; e < HT_CAP; e += ETPB) {
				const uint32_t key = sh.keys[e];
				if (key != TBL_EMPTY) hist_flush(a, key, sh.vals[e], acct);
			}
			if (timer) { t1 = wall_ns(); ctl->expandNs[0] += t1 - t0; t0 = t1; }
			if (!grid_barrier(&bc->barrierCount, generation, gridDim.x)) { if (threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }
			if (timer) { t1 = wall_ns(); ctl->expandNs[1] += t1 - t0; t0 = t1; }
		}

		// -- can this round queue anything for a next one?  Only a great-grandchild bin above the limit can (everybody looks at all the
		//    round's histograms: a few KB from L2); if none, the kernel ends after D without meeting again
		if (threadIdx.x == 0) sh.more = 0;
		__syncthreads();
		if (se - sb > 64u) { if (threadIdx.x == 0) sh.more = 1; }
		else {
			bool mine = false;
			for (uint32_t i = threadIdx.x; i < (se - sb) * HIST_BINS; i += ETPB) {
				const uint32_t s = sb + i / HIST_BINS;
				if (bin_total(a, hist, (uint64_t)s * HIST_BINS + (i % HIST_BINS), acct) > SIMLOD_MAX_POINTS_PER_NODE && slots[s].node != NONE && slots[s].level + 3u < (uint32_t)SIMLOD_MAX_DEPTH) mine = true;
			}
			if (mine) sh.more = 1;
		}
		__syncthreads();
		const bool more = sh.more != 0u;
		__syncthreads();

		// -- D: decide and build ---------------------------------------------------------------------------------------------
		Phase pd(ctl, blockIdx.x == 0);      // (measure builds: slots 32..39 of Ctl.phaseNs — where workgroup 0's D phase goes; tools/probe.py)
		for (uint32_t s = sb + blockIdx.x; s < se; s += gridDim.x) {
			const uint32_t t = threadIdx.x;
			const uint32_t myBin = t < HIST_BINS ? hist_sum(hist, (uint64_t)s * HIST_BINS + t) : 0u;      // (needs the slot's number only: in flight beside its record)
			const SlotRec rec = slots[s];
			if (rec.node == NONE) continue;                                     // nothing could be reserved for this leaf
			const uint32_t L = rec.node, l = rec.level;
			const uint32_t K = min(3u, (uint32_t)SIMLOD_MAX_DEPTH - l);         // levels below L that exist
			__syncthreads();
			uint32_t* const B3 = sh.keys; uint32_t* const B2 = B3 + HIST_BINS * GBS; uint32_t* const B1 = B2 + 64u * GBS;      // acct_rows: per-batch counts of the 512 bins, the 64 grandchildren, the 8 children
			uint32_t binTotal = myBin;
			if (acct && t < HIST_BINS) {       // an exact group: myBin = the stored points that moved; the group's samples per batch
				const uint32_t* hb = at<const uint32_t>(a, a.offHistB) + ((uint64_t)s * HIST_BINS + t) * GBS;
				for (uint32_t k = 0; k < GB; k++) { const uint32_t r = hb[k]; B3[t * GBS + k] = r; binTotal += r; }
			}
			if (t < HIST_BINS) sh.bins[t] = binTotal;
			if (t < SIMLOD_MAX_BATCHES_PER_LAUNCH) { sh.accD[t] = 0; sh.accF[t] = 0; }
			if (t < PATH_WORDS) sh.pathL[t] = t + 1 < PATH_WORDS ? paths[(uint64_t)L * PATH_WORDS + t] : 0ull;
			// the slot node's coordinates, name and grid: one round trip here, beside its path, instead of one in every phase that wants them
			if (t == 64u) { const SimlodNode* nl = a.nodes + L; sh.LX = nl->X; sh.LY = nl->Y; sh.LZ = nl->Z; sh.gridL = nl->grid; }
			if (t >= 96u && t < 116u) sh.nameL[t - 96u] = a.nodes[L].name[t - 96u];
			for (uint32_t i = t; i < LOCAL_NODES; i += ETPB) sh.listed[i] = NONE;
			if (t < 72u) sh.grid[t] = nullptr;
			if (t == 0u) { sh.numFresh = 0; sh.numFill = 0; }
			__syncthreads();
			if (t < 64u) { uint32_t c = 0; for (uint32_t k = 0; k < 8; k++) c += sh.bins[t * 8 + k]; sh.c2[t] = c; }
			__syncthreads();
			if (t < 8u) { uint32_t c = 0; for (uint32_t k = 0; k < 8; k++) c += sh.c2[t * 8 + k]; sh.c1[t] = c; }
			__syncthreads();
			if (t < 64u) {
				// a child splits when it holds more than 50 000 and is above MAX_DEPTH (K >= 2 <=> level l + 1 <= 19); a grandchild likewise
				// (or when it is an upper node of a multi-GPU job that the host's mask names: trunk_forced)
				bool f1 = false, f2 = false;
				if (trunk_any(a) && l + 1u < TRUNK_LEVELS) {
					// lane t < 8 as child t; every lane as grandchild t = child (t >> 3), octant (t & 7) below it
					const uint32_t LX = sh.LX, LY = sh.LY, LZ = sh.LZ, j = t >> 3, k = t & 7u;
					f1 = t < 8u && trunk_forced(a, l + 1u, 2u * LX + ((t >> 2) & 1u), 2u * LY + ((t >> 1) & 1u), 2u * LZ + (t & 1u));
					f2 = trunk_forced(a, l + 2u, 4u * LX + 2u * ((j >> 2) & 1u) + ((k >> 2) & 1u), 4u * LY + 2u * ((j >> 1) & 1u) + ((k >> 1) & 1u), 4u * LZ + 2u * (j & 1u) + (k & 1u));
				}
				{ const unsigned long long bf1 = __ballot(f1), bf2 = __ballot(f2); if (t == 0u) { sh.forced1 = (uint32_t)bf1 & 0xffu; sh.forced2 = bf2; } }
				const bool s1 = t < 8u && K >= 2u && (sh.c1[t & 7u] > SIMLOD_MAX_POINTS_PER_NODE || f1);
				uint32_t mask1 = (uint32_t)__ballot(s1) & 0xffu;
				const bool s2 = K >= 3u && ((mask1 >> (t >> 3)) & 1u) != 0u && (sh.c2[t] > SIMLOD_MAX_POINTS_PER_NODE || f2);
				unsigned long long mask2 = __ballot(s2);
				uint32_t n1 = (uint32_t)__popc(mask1), n2 = (uint32_t)__popcll(mask2);
				uint32_t extraBase = 0, granted = n1 + n2;
				if (t == 0 && n1 + n2 > 0u) {
					uint32_t noSlot, noSpill;
					if (!reserve(a, ctl, bc, 0u, 8u * (n1 + n2), 0u, noSlot, extraBase, noSpill)) {
						// no room in the node array for the whole cascade: as many of its splits as still fit, children first (the others stay too full: deferred)
						const uint32_t inUse = (uint32_t)(__hip_atomic_load(&bc->reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32) & 0xfffffu;
						granted = a.nodeCapacity > inUse ? min((a.nodeCapacity - inUse) / 8u, n1 + n2) : 0u;
						if (granted == 0u || !reserve(a, ctl, bc, 0u, 8u * granted, 0u, noSlot, extraBase, noSpill)) { granted = 0u; extraBase = NONE; }
					}
				}
				extraBase = __shfl(extraBase, 0); granted = __shfl(granted, 0);
				if (granted < n1 + n2) {
					// the first `granted` splits in order: the children by octant, then the grandchildren of the children that do split
					uint32_t keep1 = 0, left = granted;
					for (uint32_t j = 0; j < 8u && left != 0u; j++) if (((mask1 >> j) & 1u) != 0u) { keep1 |= 1u << j; left--; }
					unsigned long long keep2 = 0ull;
					for (uint32_t jk = 0; jk < 64u && left != 0u; jk++) if (((mask2 >> jk) & 1ull) != 0ull && ((keep1 >> (jk >> 3)) & 1u) != 0u) { keep2 |= 1ull << jk; left--; }
					mask1 = keep1; mask2 = keep2; n1 = (uint32_t)__popc(mask1); n2 = (uint32_t)__popcll(mask2);
				}
				if (extraBase == NONE) { mask1 = 0; mask2 = 0ull; n1 = 0; n2 = 0; }
				if (t < 8u) sh.base2[t] = extraBase + 8u * (uint32_t)__popc(mask1 & ((1u << t) - 1u));
				sh.base3[t] = extraBase + 8u * (n1 + (uint32_t)__popcll(mask2 & ((1ull << t) - 1ull)));
				if (t == 0) { sh.mask1 = mask1; sh.mask2 = mask2; }
			}
			__syncthreads();
			pd.mark(32);      // histograms in, counts summed, splits decided, node slots reserved
			const uint32_t mask1 = sh.mask1;
			const unsigned long long mask2 = sh.mask2;
			// local node t: does it exist, how many samples, does it split here
			auto exists = [&](uint32_t u) { return u < 8u ? true : u < 72u ? ((mask1 >> ((u - 8u) >> 3)) & 1u) != 0u : ((mask2 >> ((u - 72u) >> 3)) & 1ull) != 0ull; };
			auto countOf = [&](uint32_t u) { return u < 8u ? sh.c1[u] : u < 72u ? sh.c2[u - 8u] : sh.bins[u - 72u]; };
			auto splits = [&](uint32_t u) { return u < 8u ? ((mask1 >> u) & 1u) != 0u : u < 72u ? ((mask2 >> (u - 8u)) & 1ull) != 0ull : false; };
			auto indexOf = [&](uint32_t u) { return u < 8u ? rec.childBase + u : u < 72u ? sh.base2[(u - 8u) >> 3] + ((u - 8u) & 7u) : sh.base3[(u - 72u) >> 3] + ((u - 72u) & 7u); };
			// ---- an exact group of several batches: WHEN each node of the cascade split, what it held then, the chunks it took and returned meanwhile ----
			// Local node u's row: its samples per batch of the group.  The node is created in the batch its parent split in (`born`), holds what lay in its
			// cell up to and including that batch (the stored points that moved + the group's samples so far), and — if it splits at all — splits in the first
			// batch from then on after which it holds more than 50 000 (a node the trunk mask names: at once).  While it is a leaf it has ceil(count / 1000)
			// chunks (voxels.cu:485-538), all of which it returns when it splits (voxels.cu:346-357).
			auto rowOf = [&](uint32_t u) -> const uint32_t* { return u < 8u ? B1 + u * GBS : u < 72u ? B2 + (u - 8u) * GBS : B3 + (u - 72u) * GBS; };
			// -> the batch in which u splits (`willSplit`: it does, in this round or as a slot of the next) and its counter after that batch
			auto split_time = [&](uint32_t u, uint32_t born, bool forced, uint32_t& counterThen) -> uint32_t {
				const uint32_t* row = rowOf(u);
				uint32_t c = countOf(u);
				for (uint32_t k = 0; k < GB; k++) c -= row[k];                     // the stored points that moved into u's cell
				for (uint32_t k = 0; k <= born; k++) c += row[k];                   // what u is created with
				uint32_t k = born;
				while (!forced && c <= SIMLOD_MAX_POINTS_PER_NODE && k + 1u < GB) { k++; c += row[k]; }
				counterThen = c;
				return k;
			};
			// the chunks u takes, batch by batch, while it is a leaf: from `born` until `splitAt` (NEVER: the end of the group), and returns then
			auto leaf_chunks = [&](uint32_t u, uint32_t born, uint32_t splitAt) {
				const uint32_t* row = rowOf(u);
				uint32_t c = countOf(u);
				for (uint32_t k = 0; k < GB; k++) c -= row[k];
				for (uint32_t k = 0; k < born; k++) c += row[k];
				uint32_t have = 0;
				for (uint32_t k = born; k < min(splitAt, GB); k++) {
					c += row[k];
					const uint32_t want = (c + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
					if (want != have) atomicAdd(&sh.accD[k], want - have);
					have = want;
				}
				if (splitAt < GB && have != 0u) atomicAdd(&sh.accF[splitAt], have);
			};
			if (acct) {
				for (uint32_t i = t; i < 64u * GB; i += ETPB) { const uint32_t jk = i / GB, k = i % GB; uint32_t c = 0; for (uint32_t q = 0; q < 8; q++) c += B3[(jk * 8u + q) * GBS + k]; B2[jk * GBS + k] = c; }
				__syncthreads();
				for (uint32_t i = t; i < 8u * GB; i += ETPB) { const uint32_t j = i / GB, k = i % GB; uint32_t c = 0; for (uint32_t q = 0; q < 8; q++) c += B2[(j * 8u + q) * GBS + k]; B1[j * GBS + k] = c; }
				__syncthreads();
				if (t == 0u) {
					// the slot's own node.  One the cascade queued for this round (born != NONE): settled by the round that queued it.  A leaf of the octree as the
					// group found it: it holds rec.stored points in ceil(stored / 1000) chunks, grows batch by batch, and splits in the first batch that takes it
					// over the limit (a leaf already over it — a split that had to wait —: in the first batch that touches it; one the trunk mask names: at once)
					uint32_t sL = rec.born, cThen = 0;
					if (rec.born == NONE) {
						const bool forcedL = trunk_any(a) && trunk_forced(a, l, sh.LX, sh.LY, sh.LZ);
						uint32_t c = rec.stored;
						sL = GB - 1u;
						for (uint32_t k = 0; k < GB; k++) {
							uint32_t g = 0;
							for (uint32_t j = 0; j < 8; j++) g += B1[j * GBS + k];
							if (forcedL || (g != 0u && c + g > SIMLOD_MAX_POINTS_PER_NODE)) { sL = k; break; }
							c += g;
						}
						c = rec.stored;
						uint32_t have = (c + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
						for (uint32_t k = 0; k <= sL; k++) {
							uint32_t g = 0;
							for (uint32_t j = 0; j < 8; j++) g += B1[j * GBS + k];
							c += g;
							if (k == sL) break;
							const uint32_t want = (c + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
							if (want != have) atomicAdd(&sh.accD[k], want - have);
							have = want;
						}
						if (have != 0u) atomicAdd(&sh.accF[sL], have);
						cThen = c;
					}
					sh.splitAt[LOCAL_NODES] = (uint8_t)sL; sh.counterAt[LOCAL_NODES] = cThen;
					if (L == 0u && rec.born == NONE) bc->rootSplitAt = sL;      // (the batches before it found the root a leaf: k_rootpre)
				}
				__syncthreads();
				if (t < 8u) {                                                      // the children: created when the slot's node split
					const uint32_t born = sh.splitAt[LOCAL_NODES];
					uint32_t cThen = 0;
					const uint32_t at = splits(t) ? split_time(t, born, ((sh.forced1 >> t) & 1u) != 0u, cThen) : NEVER;
					sh.splitAt[t] = (uint8_t)at; sh.counterAt[t] = cThen;
					leaf_chunks(t, born, at);
				}
				__syncthreads();
				if (t >= 8u && t < 72u && exists(t)) {                             // the grandchildren: created when their parent split
					const uint32_t born = sh.splitAt[(t - 8u) >> 3];
					uint32_t cThen = 0;
					const uint32_t at = splits(t) ? split_time(t, born, ((sh.forced2 >> (t - 8u)) & 1ull) != 0ull, cThen) : NEVER;
					sh.splitAt[t] = (uint8_t)at; sh.counterAt[t] = cThen;
					leaf_chunks(t, born, at);
				}
				__syncthreads();
			}
			pd.mark(33);      // exact groups: when every node split, chunks taken and returned per batch
			if (t < LOCAL_NODES && exists(t)) {
				const uint32_t level = l + local_depth(t);
				if (splits(t)) sh.grid[t] = grid_for_split(a, bc);
				else if (countOf(t) > SIMLOD_MAX_POINTS_PER_NODE && level < (uint32_t)SIMLOD_MAX_DEPTH && local_depth(t) == 3u) {
					// still too full after three levels: a slot of its own for the next round (its eight children reserved now, no stored points)
					uint32_t slot = 0, childBase = 0, dummy;
					if (reserve(a, ctl, bc, 1u, 8u, 0u, slot, childBase, dummy)) {
						for (uint32_t sd = 0; sd < (slot < HIST_SHARDED ? HIST_SHARDS : 1u); sd++) {
							uint4* h = reinterpret_cast<uint4*>(hist + hist_word(slot << 9, sd));
							for (uint32_t i = 0; i < HIST_BINS / 4; i++) h[i] = make_uint4(0, 0, 0, 0);
						}
						uint32_t born = NONE, cThen = 0;
						if (acct) {      // (an exact group: the slot's histograms per batch; the batch in which this node splits — when its children are created)
							uint4* hb = reinterpret_cast<uint4*>(at<uint32_t>(a, a.offHistB) + (uint64_t)slot * HIST_BINS * GBS);
							for (uint32_t i = 0; i < HIST_BINS / 4u * GBS; i++) hb[i] = make_uint4(0, 0, 0, 0);
							born = split_time(t, sh.splitAt[8u + ((t - 72u) >> 3)], false, cThen);
						}
						slots[slot] = SlotRec{indexOf(t), level, childBase, 0u, 0u, born, 0u, 0u};
						sh.listed[t] = MAP_LISTED | (level << 16) | slot;
					}
				}
			}
			__syncthreads();
			pd.mark(34);      // grids, slots of the next round
			if (t < LOCAL_NODES && exists(t)) {
				const uint32_t depth = local_depth(t), level = l + depth, idx = indexOf(t);
				// octants chosen below L, first to last
				const uint32_t rel = t < 8u ? t : t < 72u ? t - 8u : t - 72u;
				uint32_t oct[3] = {0, 0, 0};
				for (uint32_t k = 0; k < depth; k++) oct[k] = (rel >> (3u * (depth - 1u - k))) & 7u;
				uint32_t X = sh.LX, Y = sh.LY, Z = sh.LZ;
				for (uint32_t k = 0; k < depth; k++) { X = 2u * X + ((oct[k] >> 2) & 1u); Y = 2u * Y + ((oct[k] >> 1) & 1u); Z = 2u * Z + (oct[k] & 1u); }
				const bool split = splits(t);
				const bool nextRound = sh.listed[t] != NONE;
				// written field by field straight to the node array (a 152-byte local would live in scratch memory); voxels.cu:318-343
				SimlodNode& c = a.nodes[idx];
				const uint32_t firstChild = !split ? 0u : t < 8u ? sh.base2[t] : sh.base3[t - 8u];
				for (uint32_t k = 0; k < 8; k++) c.children[k] = split ? a.nodes + firstChild + k : nullptr;
				// its child word: the first child and which of the eight stay leaves in this round (a child's children: t < 8: local nodes 8 + 8 t + k, whose
				// splits are bits 8 t + k of mask2; the great-grandchildren never split in the round that makes them)
				kidOf[idx] = !split ? 0u : firstChild | ((t < 8u ? ~(uint32_t)(mask2 >> (8u * t)) & 0xffu : 0xffu) << KID_LEAF_SHIFT);
				uint32_t counter = countOf(t);
				if (acct) {
					// what the node held after the batch in which it split (later batches of the group went past it, voxels.cu:169-187); a great-grandchild
					// is settled here (whether it got a slot for the next round is known now), the levels above it were on the way
					if (depth == 3u) {
						const uint32_t born = sh.splitAt[8u + ((t - 72u) >> 3)];
						uint32_t cThen = 0;
						const uint32_t at = nextRound ? split_time(t, born, false, cThen) : NEVER;
						sh.splitAt[t] = (uint8_t)at; sh.counterAt[t] = cThen;
						leaf_chunks(t, born, at);
					}
					if (sh.splitAt[t] != NEVER) counter = sh.counterAt[t];
				}
				c.counter = counter; c.numPoints = 0;
				c.level = level; c.X = X; c.Y = Y; c.Z = Z;
				c.countIteration = 0; c.countFlag = 0;
				for (int k = 0; k < 20; k++) c.name[k] = sh.nameL[k];
				for (uint32_t k = 0; k < depth; k++) if (l + 1u + k < 20u) c.name[l + 1u + k] = (uint8_t)('0' + oct[k]);
				c.visible = 0; c.isFiltered = 0; c.isLeaf = 1; c.isLarge = 0;
				c.grid = split ? sh.grid[t] : nullptr; c.points = nullptr; c.voxelChunks = nullptr;
				c.numVoxels = 0; c.numVoxelsStored = 0;
				if (nextRound) {                                                   // (its grid: like a queued leaf's, before its children exist)
					c.grid = grid_for_split(a, bc);
				}
				// parent, and the ancestor path: parent first, ..., then L, then L's own ancestors
				const uint32_t parentLocal = depth == 1u ? NONE : depth == 2u ? (t - 8u) >> 3 : 8u + ((t - 72u) >> 3);
				const uint32_t parentIdx = depth == 1u ? L : indexOf(parentLocal);
				parentOf[idx] = parentIdx;
				unsigned long long* mine = paths + (uint64_t)idx * PATH_WORDS;
				uint32_t w = 0;
				if (depth >= 3u) { const uint32_t g = 8u + ((t - 72u) >> 3); mine[w++] = path_pack(a.pers, indexOf(g), l + 2u, sh.grid[g]); }
				if (depth >= 2u) { const uint32_t ch = depth == 2u ? (t - 8u) >> 3 : (t - 72u) >> 6; mine[w++] = path_pack(a.pers, indexOf(ch), l + 1u, sh.grid[ch]); }
				mine[w++] = path_pack(a.pers, L, l, sh.gridL);
				for (uint32_t k = 0; w < PATH_WORDS; k++) {
					const unsigned long long e = w + 1 < PATH_WORDS ? sh.pathL[k] : 0ull;
					mine[w++] = e;
					if (e == 0ull) break;
				}
				if (t < 8u) a.nodes[L].children[t] = a.nodes + idx;
				// the nodes of the cascade that hold samples and stay leaves (one that was queued again is none by the time its chunks would be used)
				if (!split && countOf(t) != 0u && !nextRound) sh.fresh[atomicAdd(&sh.numFresh, 1u)] = FreshLeaf{idx, countOf(t), level, X, Y, Z};
				// the top table (where k_count's descent starts): a new node at level <= 5 that has no children in the table's range takes over the cells it covers
				// (a node two or more levels above the table's — 64 to 4096 cells, the first batches over a region — is filled by the whole workgroup, below)
				if (level <= TOP_LEVEL && (!split || level == TOP_LEVEL)) {
					const uint32_t mark = split ? 0u : TOP_LEAF;
					if (TOP_LEVEL - level <= 1u) top_fill(at<uint32_t>(a, a.offTop), idx, level, X, Y, Z, mark);
					else sh.fill[atomicAdd(&sh.numFill, 1u)] = make_uint4(idx | (level << 19) | mark, X, Y, Z);
				}
			}
			if (t == 0u) {
				// the slot node's child word; it is no leaf any more: its bit in its parent's word goes (siblings may be splitting in other workgroups: an atomic;
				// the parent's word was written by an earlier launch or an earlier round), and a top-table entry that names it (a node at the table's level) loses its mark
				kidOf[L] = rec.childBase | ((~mask1 & 0xffu) << KID_LEAF_SHIFT);
				if (L != 0u) {
					const uint32_t P = parentOf[L];
					if (P != NONE && kidOf[P] != KID_IRREGULAR) atomicAnd(&kidOf[P], ~(1u << (KID_LEAF_SHIFT + octant_of(sh.LX, sh.LY, sh.LZ))));
				}
				if (l == TOP_LEVEL) at<uint32_t>(a, a.offTop)[(sh.LX << (2u * TOP_LEVEL)) | (sh.LY << TOP_LEVEL) | sh.LZ] = L | (l << 19);
				a.nodes[L].numPoints = 0; a.nodes[L].points = nullptr;          // voxels.cu:359-360 (its points are in the spill buffer, its chunks on the recycle stack: k_queue, k_hist)
				if (acct && rec.born == NONE) a.nodes[L].counter = sh.counterAt[LOCAL_NODES];      // (k_count added the whole group's samples: the batches after the split went to the children)
			}
			// the slot's map: bin -> the deepest node that exists above it (or the slot that node got for the next round)
			if (t < HIST_BINS) {
				const uint32_t j = t >> 6, jk = t >> 3;
				const uint32_t u = ((mask1 >> j) & 1u) == 0u ? j : ((mask2 >> jk) & 1ull) == 0ull ? 8u + jk : 72u + t;
				const uint32_t ls = sh.listed[u];
				map[(uint64_t)s * HIST_BINS + t] = ls != NONE ? ls : indexOf(u);
			}
			// ... and their chunks, 64 leaves at a time (voxels.cu:485-538; the nodes written above are this workgroup's own stores: visible after the barrier)
			__syncthreads();
			pd.mark(35);      // nodes, paths, map
			if (acct && t < GB) {
				if (sh.accD[t] != 0u) atomicAdd(&bc->acctD[t], sh.accD[t]);
				if (sh.accF[t] != 0u) atomicAdd(&bc->acctF[t], sh.accF[t]);
			}
			for (uint32_t j = 0; j < sh.numFill; j++) {                          // the top table's cells under the big new leaves, all threads
				const uint4 f = sh.fill[j];
				const uint32_t flevel = (f.x >> 19) & 31u, k = TOP_LEVEL - flevel, side = 1u << k;
				uint32_t* top = at<uint32_t>(a, a.offTop);
				for (uint32_t i = t; i < (1u << (3u * k)); i += ETPB) {
					const uint32_t dx = i >> (2u * k), dy = (i >> k) & (side - 1u), dz = i & (side - 1u);
					top[(((f.y << k) + dx) << (2u * TOP_LEVEL)) | (((f.z << k) + dy) << TOP_LEVEL) | ((f.w << k) + dz)] = f.x;
				}
			}
			pd.mark(36);      // top table
			for (uint32_t first = 0; first < sh.numFresh; first += ALLOC_LEAVES) {
				alloc_points(a, ctl, bc, sh.alloc, sh.allocRec, nullptr, sh.fresh, first, sh.numFresh);
				__syncthreads();
			}
			pd.mark(37);      // the fresh leaves' chunks
			if (pd.on) { ctl->phaseNs[38] += sh.numFresh; ctl->phaseNs[39] += 1; }
		}
		if (timer) { t1 = wall_ns(); ctl->expandNs[2] += t1 - t0; t0 = t1; ctl->expandNs[5] += 1; }
		if (!more) break;
		if (!grid_barrier(&bc->barrierCount, generation, gridDim.x)) { if (threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }
		if (timer) { t1 = wall_ns(); ctl->expandNs[3] += t1 - t0; }
		sb = se;
		se = min(slots_in_use(bc), SLOT_CAP);
	}
}

// cell-centre position of a voxel, voxels.cu:103-114, operation by operation (no contraction): cell (cx, cy, cz) of the 128^3 grid of the
// level-`level` node with coordinates (nX, nY, nZ)
__device__ __forceinline__ float4 voxel_at(const BuildArgs& a, int level, uint32_t nX, uint32_t nY, uint32_t nZ, uint32_t cx, uint32_t cy, uint32_t cz, float colorBits) {
	const float nodeSize = a.size / exp2_int((uint32_t)level);
	const float nminx = ((float)nX + 0.0f) * nodeSize + a.minx;
	const float nminy = ((float)nY + 0.0f) * nodeSize + a.miny;
	const float nminz = ((float)nZ + 0.0f) * nodeSize + a.minz;
	float4 v;
	v.x = nminx + (nodeSize * ((float)cx + 0.5f)) / 128.0f;
	v.y = nminy + (nodeSize * ((float)cy + 0.5f)) / 128.0f;
	v.z = nminz + (nodeSize * ((float)cz + 0.5f)) / 128.0f;
	v.w = colorBits;                       // colour of the claiming point
	return v;
}
// ... of the cell a sample with 28-bit coordinates (pX, pY, pZ) falls into
__device__ __forceinline__ float4 voxel_of(const BuildArgs& a, int level, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits) {
	const uint32_t sh = (uint32_t)(SIMLOD_MAX_DEPTH + 1 - level);
	const uint32_t cx = (pX >> sh) & 127u, cy = (pY >> sh) & 127u, cz = (pZ >> sh) & 127u;
	// Node.X/Y/Z of the level-`level` node that contains the sample: the top `level` bits of its 28-bit coordinate (the
	// 2^20 grid the nodes are indexed in is the same fp32 quotient scaled by an exact power of two, simlod_device.hpp quantize)
	const uint32_t nsh = 28u - (uint32_t)level;
	// masked to `level` bits: a coordinate exactly on the max face quantises to 2^20 (2^28 here) and the reference's descent, which
	// looks at bits 19..0 only, files it under node coordinate 0 on that axis (voxels.cu:171-179) — the voxel sits at the LOW face
	const uint32_t nmask = (1u << (uint32_t)level) - 1u;
	return voxel_at(a, level, (pX >> nsh) & nmask, (pY >> nsh) & nmask, (pZ >> nsh) & nmask, cx, cy, cz, colorBits);
}

// ---- voxel chunks, on demand (voxels.cu:641-698 allocateVoxelChunks + insertVoxels in one pass) ------------------------------------------
// A reservation in a node's voxel list is the return value of the add to Node.numVoxels (voxels.cu:101): slots [old, old + n).  Whoever
// holds slot k * 1000 allocates chunk k of the list (voxel chunks never come from the recycle stack, voxels.cu:656-659), publishes it in
// a hash directory of the batch ((node, k) -> chunk) and links it behind its predecessor; whoever holds another slot of chunk k looks it up —
// or takes the list's old tail, if the chunk existed when the batch began (Node.numVoxelsStored and the tail pointer in the head chunk stay
// as they were until k_voxdone).  An allocator publishes everything it owns BEFORE it waits for anybody (its predecessor's allocator, whose
// add came first and who is therefore already running), so every wait ends.
struct DirEntry { unsigned long long key; SimlodChunk* ptr; };
static constexpr unsigned long long DIR_BUSY = 1ull << 62;
__device__ __forceinline__ unsigned long long dir_key(uint32_t tag, uint32_t node, uint32_t k) {
	return (1ull << 63) | ((unsigned long long)(tag & 0xfffffu) << 42) | ((unsigned long long)node << 22) | (k & 0x3fffffu);
}
__device__ __forceinline__ uint32_t dir_tag(unsigned long long key) { return (uint32_t)(key >> 42) & 0xfffffu; }
__device__ __forceinline__ uint32_t dir_hash(const BuildArgs& a, unsigned long long key) {
	key ^= key >> 29; key *= 0x9e3779b97f4a7c15ull; key ^= key >> 32;
	return (uint32_t)key & (a.hashCap - 1u);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void dir_insert(const BuildArgs& a, Ctl* ctl, uint32_t tag, uint32_t node, uint32_t k, SimlodChunk* c) {
	DirEntry* dir = at<DirEntry>(a, a.offHashDir);
	const unsigned long long key = dir_key(tag, node, k);
	uint32_t h = dir_hash(a, key);
	for (uint32_t probe = 0; probe < a.hashCap; probe++, h = (h + 1u) & (a.hashCap - 1u)) {
		unsigned long long cur = ld_agent(&dir[h].key);
		while (cur == 0ull || ((cur >> 63) != 0ull && dir_tag(cur) != (tag & 0xfffffu))) {           // free, or left over from an earlier batch
			const unsigned long long prev = atomicCAS(&dir[h].key, cur, DIR_BUSY);
			if (prev == cur) {
				// pointer first, key second: a reader that sees the key must see the pointer (write-through stores, drained in between)
				__hip_atomic_store(reinterpret_cast<unsigned long long*>(&dir[h].ptr), (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__hip_atomic_store(&dir[h].key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				return;
			}
			cur = prev;
		}
	}
	panic(ctl, SIMLOD_ERR_DIRECTORY_FULL);
}
// non-blocking lookup; nullptr when the entry is not (yet) there
__device__ SimlodChunk* dir_find(const BuildArgs& a, uint32_t tag, uint32_t node, uint32_t k) {
	const DirEntry* dir = at<const DirEntry>(a, a.offHashDir);
	const unsigned long long key = dir_key(tag, node, k);
	uint32_t h = dir_hash(a, key);
	for (uint32_t probe = 0; probe < a.hashCap; probe++, h = (h + 1u) & (a.hashCap - 1u)) {
		const unsigned long long cur = ld_agent(&dir[h].key);
		if (cur == key) return reinterpret_cast<SimlodChunk*>(ld_agent(reinterpret_cast<const unsigned long long*>(&dir[h].ptr)));
		if (cur == 0ull) return nullptr;
		if ((cur >> 63) != 0ull && dir_tag(cur) != (tag & 0xfffffu)) return nullptr;
	}
	return nullptr;
}
// blocking lookup (see above: it ends; the bound is a guard against a broken device)
__device__ SimlodChunk* dir_wait(const BuildArgs& a, Ctl* ctl, uint32_t tag, uint32_t node, uint32_t k) {
	for (uint32_t spin = 0;; spin++) {
		SimlodChunk* c = dir_find(a, tag, node, k);
		if (c != nullptr) return c;
		__builtin_amdgcn_s_sleep(2);
		if ((spin & 255u) == 255u && __hip_atomic_load(&ctl->abortBatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return nullptr;
		if (spin > (1u << 22)) { panic(ctl, SIMLOD_ERR_DIRECTORY_FULL); return nullptr; }
	}
}
// chunk k of node's voxel list exists from now on (the caller holds slot k * 1000)
__device__ __forceinline__ void vox_chunk_publish(const BuildArgs& a, Ctl* ctl, uint32_t tag, uint32_t node, uint32_t k, SimlodChunk* c) {
	// an inner node's row of the leaf chunk table lists its voxel chunks: the rasteriser reads the list from there (render.hip r_visible) — never
	// the root's: its row may still be read as a LEAF's by the next batch's k_count, which splits a root that was still a leaf
	if (k < LEAF_SLOTS && node != 0u) leaf_row_set(a.mom + a.offLeafChunks, a.pers, node, k, c);
	dir_insert(a, ctl, tag, node, k, c);
}
// ... and hangs behind its predecessor: the head pointer, the old tail, or a chunk of this batch (which may have to be waited for).
// Every `next` field has ONE writer per kernel — the allocator of the chunk behind it, or, for the list's last chunk, k_voxdone (the L2s of
// the eight XCDs are not coherent with each other for plain stores: an allocator that cleared its own chunk's `next` could overwrite, at
// write-back time, the link its successor's allocator has made from another XCD).
__device__ __forceinline__ void vox_chunk_link(const BuildArgs& a, Ctl* ctl, uint32_t tag, uint32_t node, uint32_t k, uint32_t existing, SimlodChunk* oldTail, SimlodChunk* c) {
	if (k == 0u) a.nodes[node].voxelChunks = c;
	else {
		SimlodChunk* pred = k - 1u < existing ? oldTail : dir_wait(a, ctl, tag, node, k - 1u);
		if (pred != nullptr) pred->next = c;
	}
}
// A wave stores one voxel per `go` lane: lane's slot in `node`'s voxel list (lanes may name different nodes).  All allocations of the
// wave come before any of its waits (a lane may wait for a chunk another lane of the same wave opens).
__device__ __forceinline__ void store_voxels_wave(const BuildArgs& a, Ctl* ctl, uint32_t tag, bool go, uint32_t node, uint32_t slot, const float4& vox) {
	const uint32_t k = slot / SIMLOD_POINTS_PER_CHUNK, r = slot % SIMLOD_POINTS_PER_CHUNK;
	const uint32_t existing = go ? (a.nodes[node].numVoxelsStored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK : 0u;
	SimlodChunk* mine = nullptr;
	if (go && r == 0u) {
		mine = reinterpret_cast<SimlodChunk*>(persistent_alloc(a.pers, sizeof(SimlodChunk), 1));
		vox_chunk_publish(a, ctl, tag, node, k, mine);             // (its `next`: see vox_chunk_link)
	}
	if (go) {
		SimlodChunk* oldTail = existing > 0u ? tail_of(a.nodes[node].voxelChunks) : nullptr;
		if (mine != nullptr) vox_chunk_link(a, ctl, tag, node, k, existing, oldTail, mine);
		SimlodChunk* c = mine != nullptr ? mine : k < existing ? oldTail : dir_wait(a, ctl, tag, node, k);
		if (c != nullptr) reinterpret_cast<float4*>(c->points)[r] = vox;
	}
}

// ---- voxelize: 128^3 occupancy test-and-set on every inner node of the root-to-leaf path (voxels.cu:50-121, 417-483) ----------------
// The reference offers every sample to the grid of EVERY node of its path, root first.  Two properties make that cheap here:
//  * Occupancy is hierarchical: a cell of a node covers exactly 2x2x2 cells of the child below it, and every sample that ever set a bit
//    in a node had, in the same pass, been offered to all its ancestors — so "bit set in node N" implies "covering bit set in every
//    ancestor of N".  A sample therefore climbs BOTTOM-UP and stops at the first cell that is already set.  The same argument covers the
//    points a split moved into new leaves (voxels.cu:325-415 re-samples them from the split node's level down): above the split node
//    their cells were set when they first arrived, so their climb ends there by itself.
//  * After k_insert the new samples of a leaf lie together in its chunks, and the part of an ancestor's grid a leaf can touch is a
//    small cube: 64^3 cells of the parent, 32^3 of the grandparent, ... one cell from the 8th ancestor on — 37 KB of bits in all.
// So this kernel runs AFTER k_insert, one workgroup per (leaf, range of <= 8192 new samples): it loads the leaf's cubes into LDS, keeps
// its 8 samples per thread in registers, and does the whole test-and-set in LDS.  Round 1 sampled BEFORE the insert, per batch sample,
// with device-scope atomicOr: a 1 M-point batch is spatially compact, nearly every sample of a freshly entered region found its cell
// clear at the same moment, and a thousand workgroups queued their atomics on the same few hundred words (measured: one wave waited
// 55-190 us for a single level's atomics; the kernel took 80-90 us with the memory system idle; more samples in flight per thread, or
// fewer workgroups, both made it SLOWER).  Here the global memory sees one atomicOr per touched WORD and piece — it returns which of
// the new cells are really new (the pieces of one leaf share cubes; different leaves never share a cell below the 8th ancestor) —
// and nothing in the two passes over the samples leaves the CU.
//   level 1    every sample (eight per thread, in registers): test-and-set of its cell in the parent's cube; the samples that found their cell
//              clear are the level's WINNERS: they go on a list (LDS) — cell and sample number in one word
//   level d    the winners of level d - 1, DENSELY (a list entry per lane, not a sample per lane: three samples in four lose at level 1 and
//              would idle through six more levels): test-and-set in cube d, winners on the list of level d.  The lists shrink with the cubes
//              (<= 4096, 512, 64, 8, 1 winners from d = 3 on); from the 8th ancestor on the leaf is ONE cell: the last winner climbs alone
//   write-back  old = atomicOr(grid word, bits set here); won = those & ~old; Node.numVoxels += popcount(won)   (voxels.cu:96-101).  What a
//              thread set = its LDS words now minus the same words as it loaded them (kept in registers): no second bit plane
//   reserve    per ancestor with won cells: slot range from atomicAdd(Node.numVoxels); chunks the range starts are allocated here and now
//   store      a winner whose cell is still marked won colours the voxel — position from the CELL (list entry + the leaf's coordinates),
//              colour from the sample (which sample of a cell does is scheduling dependent in the reference too) —
//              into slot base + rank (voxels.cu:674-698), through the hash directory of chunks
static constexpr uint32_t VOX_CHUNKS = VOX_PIECE / SIMLOD_POINTS_PER_CHUNK + 2;   // chunks a piece's voxels of one ancestor can span
// winner lists: level d holds at most min(8192, (128 >> d)^3) entries
static constexpr uint32_t VOX_LIST_WORDS = 2 * VOX_PIECE + 4096 + 512 + 64 + 8 + 8;
__device__ __forceinline__ uint32_t list_offset(uint32_t d) { return d == 1u ? 0u : d == 2u ? VOX_PIECE : d == 3u ? 2u * VOX_PIECE : d == 4u ? 2u * VOX_PIECE + 4096u : d == 5u ? 2u * VOX_PIECE + 4608u : d == 6u ? 2u * VOX_PIECE + 4672u : 2u * VOX_PIECE + 4680u; }
struct VoxShared {
	uint32_t occ[CUBE_WORDS];                                   // cubes d = 1..7: rows of (128 >> d) x-bits; d = 1: two words per row.  After the write-back: the cells this piece WON
	uint32_t list[VOX_LIST_WORDS];                              // winner = cell inside the leaf's level-1 cube (6 bits per axis) | sample << 18
	uint32_t listCount[8];                                      // winners of level d
	uint32_t hiOcc[PATH_WORDS], hiFresh[PATH_WORDS];            // ancestors d >= 8: the ONE cell the whole leaf falls into
	uint32_t hiSample;                                          // the sample that climbs beyond the 7th ancestor
	unsigned long long anc[PATH_WORDS];
	uint32_t cnt[PATH_WORDS];
	uint32_t first[PATH_WORDS], rank[PATH_WORDS];               // the slots this piece reserved in ancestor d's voxel list: [first, first + cnt); how many of them are taken
	SimlodChunk* chunkOf[PATH_WORDS][VOX_CHUNKS];               // ... and the chunks they lie in, from chunk first / 1000 on
	float color[VOX_PIECE];                                     // the samples' colours
};
__device__ __forceinline__ uint32_t cube_offset(uint32_t d) {          // word offset of cube d in VoxShared::occ / fresh
	return d == 1u ? 0u : d == 2u ? 8192u : d == 3u ? 9216u : d == 4u ? 9472u : d == 5u ? 9536u : d == 6u ? 9552u : 9556u;
}
__device__ __forceinline__ uint32_t grid_cell(uint32_t level, uint32_t pX, uint32_t pY, uint32_t pZ) {   // voxels.cu:78-92
	const uint32_t shf = (uint32_t)(SIMLOD_MAX_DEPTH + 1) - level;
	return ((pX >> shf) & 127u) + ((pY >> shf) & 127u) * SIMLOD_GRID_SIZE + ((pZ >> shf) & 127u) * SIMLOD_GRID_SIZE * SIMLOD_GRID_SIZE;
}
// A winner names its cell by the cell's position inside the leaf's level-1 cube: code = lx | ly << 6 | lz << 12 (six bits per axis: the
// leaf is 64^3 cells of its parent's grid).  The cell above it in ancestor d's cube (side 128 >> d, aligned to its side) is code >> (d - 1)
// per axis: word and bit in the LDS cubes.
__device__ __forceinline__ void cube_cell_from(uint32_t d, uint32_t code, uint32_t& word, uint32_t& bit) {
	const uint32_t s = d - 1u;
	const uint32_t lx = (code & 63u) >> s, row = (((code >> 6) & 63u) >> s) + ((((code >> 12) & 63u) >> s) << (7u - d));
	if (d == 1u) { word = row * 2u + (lx >> 5); bit = lx & 31u; }
	else { word = cube_offset(d) + row; bit = lx; }
}
// the voxel of that cell in ancestor d of the leaf (LX, LY, LZ) at level leafLevel: the cube starts at cell (L & (2^d - 1)) * side of the
// ancestor's grid, the ancestor's own coordinates are L >> d   (the same cell, hence the same voxel, as voxel_of() finds for any sample in it)
__device__ __forceinline__ float4 voxel_from(const BuildArgs& a, uint32_t d, uint32_t code, uint32_t leafLevel, uint32_t LX, uint32_t LY, uint32_t LZ, float colorBits) {
	const uint32_t s = d - 1u, side = 128u >> d, m = (1u << d) - 1u;
	const uint32_t cx = (LX & m) * side + ((code & 63u) >> s), cy = (LY & m) * side + (((code >> 6) & 63u) >> s), cz = (LZ & m) * side + (((code >> 12) & 63u) >> s);
	return voxel_at(a, (int)(leafLevel - d), LX >> d, LY >> d, LZ >> d, cx, cy, cz, colorBits);
}
// LDS word w of the cubes -> which ancestor's grid word it mirrors: d (0: none), the word's index in that grid, the bit offset of the
// cube's row inside the word, and the row's mask
__device__ __forceinline__ uint32_t cube_word(uint32_t w, uint32_t LX, uint32_t LY, uint32_t LZ, uint32_t& gridWord, uint32_t& shift, uint32_t& mask) {
	const uint32_t d = w < 8192u ? 1u : w < 9216u ? 2u : w < 9472u ? 3u : w < 9536u ? 4u : w < 9552u ? 5u : w < 9556u ? 6u : w < 9557u ? 7u : 0u;
	if (d == 0u) { gridWord = 0; shift = 0; mask = 0; return 0u; }
	const uint32_t side = 128u >> d, ox = (LX & ((1u << d) - 1u)) * side, oy = (LY & ((1u << d) - 1u)) * side, oz = (LZ & ((1u << d) - 1u)) * side;
	const uint32_t rel = w - cube_offset(d), row = d == 1u ? rel >> 1 : rel, ly = row % side, lz = row / side;
	const uint32_t cell = ox + 128u * (oy + ly) + 16384u * (oz + lz);
	gridWord = (cell >> 5) + (d == 1u ? (rel & 1u) : 0u);
	shift = d <= 2u ? 0u : (cell & 31u);
	mask = side >= 32u ? 0xffffffffu : (1u << side) - 1u;
	return d;
}

// Leaves that received only a few samples (a batch that is NOT spatially compact — uniformly scattered points, a sparse overview
// scan — touches tens of thousands of leaves with a few dozen samples each): loading 38 KB of cubes per leaf would cost far more
// than the samples, and there is nothing to contend for.  One WAVE per leaf, the round-1 way: every sample probes its ancestors' grids
// bottom-up with plain loads and claims with atomicOr, four samples per lane in flight; the winners of one level of one leaf share
// the ancestor, so Node.numVoxels takes one add per (leaf, level, step).  The waves of k_voxelize's workgroups do this after their
// pieces.  (Measured: without this path the uniformly scattered
// 350 M-point replay of the C++ harness, 40 000 leaves touched per batch, took 408 ms of kernel time instead of 157 ms.)
__device__ __forceinline__ void voxelize_small(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, const uint32_t wave, const uint32_t numWaves) {
	const uint32_t numSmall = min(bc->numVoxSmall, a.voxItemCap - VOX_BIG_ITEMS);
	if (numSmall == 0u) return;
	const VoxItem* items = vox_items(a, bc);
	SimlodChunk* const* chunkDir = chunk_dir(a, bc);
	const uint32_t tag = bc->tag;
	const uint32_t lane = (uint32_t)lane_id();
	constexpr uint32_t U = 4;                              // items a wave works on together when there are more items than waves: a scattered batch leaves ~25 samples in each of tens of thousands of leaves
	const uint32_t per = numSmall > numWaves / 4u ? U : 1u;          // (several items of a wave in flight once the items outnumber a quarter of the waves: config 5 has 8 192 items per batch for 8 192 waves of which half are resident — one item per wave took two rounds of ~80 us)
	for (uint32_t k0 = wave * per; k0 < numSmall; k0 += numWaves * per) {
		VoxItem it[U];
		unsigned long long mine[U];
		uint32_t itemIndex[U], count[U], depth[U];
		uint32_t maxCount = 0, maxDepth = 0;
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			itemIndex[u] = VOX_BIG_ITEMS + min(k0 + u, numSmall - 1u);
			it[u] = items[itemIndex[u]];
		}
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const uint32_t leafIdx = it[u].leaf & 0xffffffu;
			count[u] = u < per && k0 + u < numSmall ? it[u].s1 - it[u].s0 : 0u;
			// the leaf's path, one entry per lane (entry d - 1 = ancestor d; a root that is still a leaf samples itself, voxels.cu:449-463)
			mine[u] = 0ull;
			if (leafIdx == 0u) { SimlodOccupancyGrid* g = a.nodes[0].grid; if (lane == 0u && g != nullptr) mine[u] = path_pack(a.pers, 0u, 0u, g); }
			else if (lane < PATH_WORDS - 1) mine[u] = (at<const unsigned long long>(a, a.offPaths) + (uint64_t)leafIdx * PATH_WORDS)[lane];
		}
#pragma unroll
		for (uint32_t u = 0; u < U; u++) {
			const unsigned long long present = __ballot(mine[u] != 0ull);
			depth[u] = count[u] != 0u ? (uint32_t)__ffsll((long long)~present) - 1u : 0u;      // entries up to the terminator
			maxCount = max(maxCount, count[u]); maxDepth = max(maxDepth, depth[u]);
		}
		for (uint32_t base = 0; base < maxCount; base += 64u) {
			uint32_t pX[U], pY[U], pZ[U];
			float color[U];
			bool go[U];
			{
				float4 p[U];
#pragma unroll
				for (uint32_t u = 0; u < U; u++) {
					const uint32_t rel = base + lane, i = it[u].s0 + rel;
					go[u] = rel < count[u];
					p[u] = go[u] ? reinterpret_cast<const float4*>(chunkDir[it[u].ptBase + (i / SIMLOD_POINTS_PER_CHUNK - it[u].ptFirst)]->points)[i % SIMLOD_POINTS_PER_CHUNK] : make_float4(0, 0, 0, 0);
				}
#pragma unroll
				for (uint32_t u = 0; u < U; u++) {
					pX[u] = quantize(F_FULL, p[u].x, a.minx, a.size); pY[u] = quantize(F_FULL, p[u].y, a.miny, a.size); pZ[u] = quantize(F_FULL, p[u].z, a.minz, a.size);
					color[u] = p[u].w;
				}
			}
			for (uint32_t d = 1; d <= maxDepth; d++) {                                   // bottom-up, the whole wave one level at a time
				bool any = false;
#pragma unroll
				for (uint32_t u = 0; u < U; u++) { go[u] = go[u] && d <= depth[u]; any = any || go[u]; }
				if (__ballot(any) == 0ull) break;
				unsigned long long ent[U];
				uint32_t* word[U]; uint32_t bit[U], seen[U], level[U];
#pragma unroll
				for (uint32_t u = 0; u < U; u++) {
					ent[u] = ((unsigned long long)(uint32_t)__shfl((int)(mine[u] >> 32), (int)d - 1, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)mine[u], (int)d - 1, 64);
					level[u] = path_level(ent[u]);
					go[u] = go[u] && ent[u] != 0ull && level[u] < (uint32_t)SIMLOD_MAX_DEPTH;   // voxels.cu:449: levels 0..19 only
					const uint32_t cell = grid_cell(level[u], pX[u], pY[u], pZ[u]);
					word[u] = &path_grid(a.pers, ent[u])->values[cell >> 5]; bit[u] = cell & 31u;
					seen[u] = go[u] ? *word[u] : 0xffffffffu;
				}
#pragma unroll
				for (uint32_t u = 0; u < U; u++) { go[u] = go[u] && ((seen[u] >> bit[u]) & 1u) == 0u; seen[u] = go[u] ? atomicOr(word[u], 1u << bit[u]) : 0xffffffffu; }   // voxels.cu:93-96
#pragma unroll
				for (uint32_t u = 0; u < U; u++) {
					go[u] = go[u] && ((seen[u] >> bit[u]) & 1u) == 0u;                  // lost: the winner climbs on
					// the winners share the ancestor: one add per (leaf, level) reserves their slots in its voxel list (voxels.cu:101), and each stores
					// its voxel — the cell's centre in the sample's colour (voxels.cu:103-114, 674-698) — right away
					const unsigned long long wm = __ballot(go[u]);
					if (wm == 0ull) continue;
					const uint32_t node = path_node(ent[u]);
					uint32_t first = 0;
					if (lane == 0u) first = atomicAdd(&a.nodes[node].numVoxels, (uint32_t)__popcll(wm));
					first = (uint32_t)__shfl((int)first, 0, 64);
					store_voxels_wave(a, ctl, tag, go[u], node, first + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull)), voxel_of(a, (int)level[u], pX[u], pY[u], pZ[u], color[u]));
				}
			}
		}
	}
}

// ---- end of batch: bookkeeping (voxels.cu:535-537, 925-949), then make the next batch current ------------------
// One thread of k_voxelize, the kernel behind the group's k_insert: the group's points are stored and its ring slots no longer read
// (nothing the rest of that launch reads is touched here: the kernels know their group by its parity copy).
__device__ void end_of_batch(const BuildArgs& a, Ctl* ctl, BatchCtl* bc) {
	if (ctl->abortBatch) __hip_atomic_store(&ctl->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // scratch overflow: this batch is lost, report through Stats.dbg
	else {
		a.stats->batchletIndex += bc->groupBatches;
		a.stats->numPointsProcessed += bc->batchSize;
		ctl->processed += 1;
		ctl->expandNs[7] += min(bc->numSpilled, a.spilledCap);   // measurement aid: stored points moved by splits so far (bench.py)
		ctl->expandNs[4] += 1;                                   // ... and groups of batches ingested so far: the launches of a per-group kernel that had work (bench.py's roofline)
		// voxels.cu:936-949: no further batch once the launch has run for 10 ms.  The front half of the next batch (or two) may be under way
		// already; what has been prepared is completed, nothing more is prepared.
		const float elapsedMs = (float)(wall_ns() - ctl->startNs) / 1000000.0f;
		if (elapsedMs > (float)ctl->budgetUs / 1000.0f) __hip_atomic_store(&ctl->stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

// On the library's side stream, while the next batch is already counted and split on the caller's.  A root that is still a leaf (the whole
// octree holds fewer than 50 000 points) is sampled here like every other leaf — into its own grid: the next batch may split that root, but
// the grid is cleared by that batch's k_insert (voxels.cu:371-382), which waits for this kernel.  No other grid is touched by both: a leaf
// that k_expand splits gets a NEW grid, and nothing samples into a leaf's own grid.
static constexpr uint32_t VOXROOT_WGS = 8;         // the last workgroups of k_voxelize's grid: the pieces of a root that is still a leaf (voxroot_pieces)
__device__ __forceinline__ void voxroot_pieces(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, VoxShared& sh, uint32_t wg, uint32_t numWgs);

__global__ __launch_bounds__(VTPB) void k_voxelize(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr || ctl->abortBatch) return;
	__shared__ VoxShared sh;
	const uint32_t numWgs = gridDim.x - VOXROOT_WGS;          // (the host launches VOXROOT_WGS more than it has pieces for)
	if (blockIdx.x >= numWgs) {
		if (bc->rootPieces != 0u) voxroot_pieces(a, ctl, bc, sh, blockIdx.x - numWgs, VOXROOT_WGS);
		return;
	}
	// the group's points are stored (k_insert has ended: nobody reads its ring slots any more): voxels.cu:925-949
	if (blockIdx.x == 0 && threadIdx.x == 0) end_of_batch(a, ctl, bc);
	const uint32_t numItems = min(bc->numVoxItems, VOX_BIG_ITEMS);
	const VoxItem* items = vox_items(a, bc);
	SimlodChunk* const* chunkDir = chunk_dir(a, bc);
	const uint32_t tag = bc->tag;
	static_assert(VTPB == 1024 && CUBE_WORDS <= 9216 + VTPB, "k_voxelize: thread t owns the LDS cube words t + 1024 k (k < 8: cube 1), 8192 + t (cube 2), 9216 + t (cubes 3..7)");
	const uint32_t lane = (uint32_t)lane_id();
	Phase ph(ctl, blockIdx.x == ((ctl->debugFlags >> 8) & 0xffffu));     // (SIMLOD_DEBUG_PHASE_WG: whose phase times tools/probe.py prints; default workgroup 0)
	const bool clocked = SIMLOD_MEASURE != 0 && (ctl->debugFlags & 2u) != 0u;       // SIMLOD_DEBUG_VOXELIZE_CLOCK (tools/probe.py): when the first workgroup came, the last piece was done, the last workgroup left
	if (clocked && blockIdx.x == 0 && threadIdx.x == 0) ctl->voxT[ordinal][0] = wall_ns();
	for (uint32_t item = blockIdx.x; item < numItems; item += numWgs) {
		// Global memory is touched in six steps, each one round trip with everything it needs in flight together: the item; the leaf's
		// path; chunk addresses + cube words; the samples; the write-back atomics; the slot reservations and voxel chunks.
		VoxItem it = items[item];
		const uint32_t leafLevel = it.leaf >> 24;
		it.leaf &= 0xffffffu;
		if (it.leaf == 0u) continue;                              // (the root as a leaf: voxroot_pieces)
		const uint32_t LX = a.nodes[it.leaf].X, LY = a.nodes[it.leaf].Y, LZ = a.nodes[it.leaf].Z;      // (in flight beside the leaf's path)
		const unsigned long long* rec = at<const unsigned long long>(a, a.offPaths) + (uint64_t)it.leaf * PATH_WORDS;
		__syncthreads();                                       // the previous item's LDS state is no longer read
		if (threadIdx.x < PATH_WORDS) {
			// ancestor d (1 = parent) is anc[d - 1].  (A root that is still a leaf samples ITSELF, voxels.cu:449-463 — the whole octree holds fewer
			// than 50 000 points —: voxroot_pieces, skipped here.)
			const unsigned long long e = threadIdx.x < PATH_WORDS - 1 ? rec[threadIdx.x] : 0ull;
			sh.anc[threadIdx.x] = e; sh.cnt[threadIdx.x] = 0; sh.hiOcc[threadIdx.x] = 0; sh.hiFresh[threadIdx.x] = 0; sh.rank[threadIdx.x] = 0;
			if (threadIdx.x < 8u) sh.listCount[threadIdx.x] = 0;
		}
		const SimlodChunk* chunk[VOX_SPT];
		bool live[VOX_SPT];
#pragma unroll
		for (uint32_t j = 0; j < VOX_SPT; j++) {
			const uint32_t i = it.s0 + j * VTPB + threadIdx.x;
			live[j] = i < it.s1;
			chunk[j] = live[j] ? chunkDir[it.ptBase + (i / SIMLOD_POINTS_PER_CHUNK - it.ptFirst)] : nullptr;
		}
		__syncthreads();
		ph.mark(24);
		uint32_t depth = 0;
		while (depth < PATH_WORDS - 1 && sh.anc[depth] != 0ull) depth++;
		// voxels.cu:449: the traverse loop samples levels 0..19 only — an ancestor is at level 19 at most (leaves are at most at 20)
		const uint32_t ldsDepth = min(depth, LDS_LEVELS);

		// the leaf's cubes, as the grids hold them now (each thread keeps what it loaded: the write-back needs it); the single cells of the
		// ancestors above; the samples
		// LDS word w of the cubes belongs to thread w % 1024: eight words of cube 1 (the parent's grid: words g1 + k * 4096 — the 64 x-bits of one
		// (y, z) row are a pair of words —), one of cube 2 (the grandparent's: g2), and for w >= 9216 one of the small cubes 3..7 (cube_word).
		// (Round 4 ran the general cube_word() for all ten words, twice, and kept its results: 128 VGPRs and 40 bytes of scratch per lane.)
		const uint32_t g1 = ((((LX & 1u) * 64u) + 128u * ((LY & 1u) * 64u) + 16384u * ((LZ & 1u) * 64u)) >> 5) + ((threadIdx.x >> 1) & 63u) * 4u + (threadIdx.x >> 7) * 512u + (threadIdx.x & 1u);
		const uint32_t g2 = (((LX & 3u) * 32u) + 128u * ((LY & 3u) * 32u + (threadIdx.x & 31u)) + 16384u * ((LZ & 3u) * 32u + (threadIdx.x >> 5))) >> 5;
		const uint32_t w3 = 9216u + threadIdx.x;
		uint32_t gw3, sft3, msk3;
		const uint32_t d3 = cube_word(w3, LX, LY, LZ, gw3, sft3, msk3);
		const bool have3 = d3 != 0u && d3 <= ldsDepth;
		uint32_t* const grid1 = ldsDepth >= 1u ? path_grid(a.pers, sh.anc[0])->values : nullptr;
		uint32_t* const grid2 = ldsDepth >= 2u ? path_grid(a.pers, sh.anc[1])->values : nullptr;
		uint32_t* const grid3 = have3 ? path_grid(a.pers, sh.anc[d3 - 1u])->values : nullptr;
		uint32_t snap1[8], snap2, snap3;
		{
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) snap1[k] = ldsDepth >= 1u ? grid1[g1 + k * 4096u] : 0u;
			snap2 = ldsDepth >= 2u ? grid2[g2] : 0u;
			snap3 = have3 ? grid3[gw3] : 0u;
			uint32_t hi = 0;
			const bool hiMine = threadIdx.x >= LDS_LEVELS && threadIdx.x < depth;   // d = threadIdx.x + 1 >= 8: every sample of the leaf has the same cell
			if (hiMine) {
				const unsigned long long ent = sh.anc[threadIdx.x];
				const uint32_t d = threadIdx.x + 1u, level = path_level(ent);
				// the leaf's own corner stands for all its samples: 2^(28 - leafLevel) fine units per leaf, leafLevel = level + d
				const uint32_t s2 = 28u - (level + d), cell = grid_cell(level, LX << s2, LY << s2, LZ << s2);
				hi = (path_grid(a.pers, ent)->values[cell >> 5] >> (cell & 31u)) & 1u;
			}
			float4 p[VOX_SPT];
#pragma unroll
			for (uint32_t j = 0; j < VOX_SPT; j++)
				p[j] = live[j] ? reinterpret_cast<const float4*>(chunk[j]->points)[(it.s0 + j * VTPB + threadIdx.x) % SIMLOD_POINTS_PER_CHUNK] : make_float4(0, 0, 0, 0);
			snap3 = have3 ? (snap3 >> sft3) & msk3 : 0u;
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) sh.occ[k * VTPB + threadIdx.x] = snap1[k];
			sh.occ[8192u + threadIdx.x] = snap2;
			if (w3 < CUBE_WORDS) sh.occ[w3] = snap3;
			if (hiMine) sh.hiOcc[threadIdx.x] = hi;
			__syncthreads();
			ph.mark(25);

			// level 1: every sample, from registers.  Its cell inside the leaf's cube of the parent's grid: six bits per axis below the leaf's
			// own bits (the parent's grid cell is coordinate >> (21 - parentLevel), voxels.cu:78-92)
			{
				const uint32_t shf = (uint32_t)(SIMLOD_MAX_DEPTH + 2) - leafLevel;
				uint32_t code[VOX_SPT], word[VOX_SPT], bit[VOX_SPT], old[VOX_SPT];
#pragma unroll
				for (uint32_t j = 0; j < VOX_SPT; j++) {
					const uint32_t pX = quantize(F_FULL, p[j].x, a.minx, a.size), pY = quantize(F_FULL, p[j].y, a.miny, a.size), pZ = quantize(F_FULL, p[j].z, a.minz, a.size);
					code[j] = ((pX >> shf) & 63u) | (((pY >> shf) & 63u) << 6) | (((pZ >> shf) & 63u) << 12);
					sh.color[j * VTPB + threadIdx.x] = p[j].w;
					cube_cell_from(1u, code[j], word[j], bit[j]);
					old[j] = live[j] ? sh.occ[word[j]] : 0xffffffffu;
				}
#pragma unroll
				for (uint32_t j = 0; j < VOX_SPT; j++) if (((old[j] >> bit[j]) & 1u) == 0u) old[j] = atomicOr(&sh.occ[word[j]], 1u << bit[j]);   // voxels.cu:93-96
				// the thread's winners go on the list (the losers' cells had been set: their climb ends here): one reservation per wave
				uint32_t wins = 0;
#pragma unroll
				for (uint32_t j = 0; j < VOX_SPT; j++) if (((old[j] >> bit[j]) & 1u) == 0u) wins |= 1u << j;
				uint32_t total;
				const uint32_t before = wave_exclusive((uint32_t)__popc(wins), total);
				uint32_t base = 0;
				if (lane == 0u && total != 0u) base = atomicAdd(&sh.listCount[1], total);
				base = (uint32_t)__shfl((int)base, 0, 64) + before;
#pragma unroll
				for (uint32_t j = 0; j < VOX_SPT; j++) if (((wins >> j) & 1u) != 0u) sh.list[base++] = code[j] | ((j * VTPB + threadIdx.x) << 18);
			}
		}
		__syncthreads();
		ph.mark(26);
		// levels 2 .. 7: the winners of level 1, a list entry per lane, climb on INSIDE their wave — no workgroup barrier per level (six of them
		// cost more than the climbing): a wave takes four entries per lane, tests and sets their cells level by level while any of them still
		// finds its cell clear, and appends each level's winners to that level's list (one reservation per wave and level).  Waves never need
		// each other: the cubes of different levels are different words, and a cell has ONE winner whoever gets there first.
		{
			const uint32_t n1 = sh.listCount[1];
			const uint32_t* src = sh.list + list_offset(1u);
			for (uint32_t i0 = 0; i0 < n1 && ldsDepth >= 2u; i0 += 4u * VTPB) {
				uint32_t e[4];
				bool go[4];
#pragma unroll
				for (uint32_t q = 0; q < 4; q++) { const uint32_t i = i0 + q * VTPB + threadIdx.x; go[q] = i < n1; e[q] = go[q] ? src[i] : 0u; }
				for (uint32_t d = 2; d <= ldsDepth; d++) {
					if (__ballot(go[0] || go[1] || go[2] || go[3]) == 0ull) break;      // (wave-uniform)
					uint32_t* dst = sh.list + list_offset(d);
					uint32_t word[4], bit[4], old[4];
#pragma unroll
					for (uint32_t q = 0; q < 4; q++) { cube_cell_from(d, e[q] & 0x3ffffu, word[q], bit[q]); old[q] = go[q] ? sh.occ[word[q]] : 0xffffffffu; }
#pragma unroll
					for (uint32_t q = 0; q < 4; q++) if (((old[q] >> bit[q]) & 1u) == 0u) old[q] = atomicOr(&sh.occ[word[q]], 1u << bit[q]);
#pragma unroll
					for (uint32_t q = 0; q < 4; q++) {
						go[q] = ((old[q] >> bit[q]) & 1u) == 0u;                    // lost: the winner climbs on
						const unsigned long long wm = __ballot(go[q]);
						if (wm != 0ull) {
							uint32_t base = 0;
							if (lane == 0u) base = atomicAdd(&sh.listCount[d], (uint32_t)__popcll(wm));
							base = (uint32_t)__shfl((int)base, 0, 64);
							if (go[q]) dst[base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull))] = e[q];
						}
					}
				}
			}
			__syncthreads();
		}
		// from the 8th ancestor on the whole leaf is one cell: the winner of the 7th cube (there is at most one) climbs alone
		if (threadIdx.x == 0u && depth > LDS_LEVELS && sh.listCount[LDS_LEVELS] != 0u) {
			sh.hiSample = sh.list[list_offset(LDS_LEVELS)] >> 18;
			for (uint32_t d = LDS_LEVELS + 1u; d <= depth; d++) {
				if (sh.hiOcc[d - 1u] != 0u) break;
				sh.hiOcc[d - 1u] = 1u; sh.hiFresh[d - 1u] = 1u;
			}
		}
		__syncthreads();
		ph.mark(27);

		// write-back: the grids learn the new cells and tell which of them are new for everybody (pieces of one leaf share the cubes):
		// every thread's atomics are in flight together.  What this piece set in a word = the word now minus the word as it was loaded.
		{
			uint32_t f1[8], old1[8], f2, old2 = 0, f3, old3 = 0;
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) {
				f1[k] = sh.occ[k * VTPB + threadIdx.x] & ~snap1[k];
				old1[k] = f1[k] != 0u ? atomicOr(&grid1[g1 + k * 4096u], f1[k]) : 0u;                                  // voxels.cu:96
			}
			f2 = sh.occ[8192u + threadIdx.x] & ~snap2;
			if (f2 != 0u) old2 = atomicOr(&grid2[g2], f2);
			f3 = w3 < CUBE_WORDS ? sh.occ[w3] & ~snap3 : 0u;
			if (f3 != 0u) old3 = atomicOr(&grid3[gw3], f3 << sft3);
			uint32_t won1 = 0;
#pragma unroll
			for (uint32_t k = 0; k < 8; k++) {
				const uint32_t won = f1[k] & ~old1[k];
				sh.occ[k * VTPB + threadIdx.x] = won;                               // from here on: the cells this piece won
				won1 += (uint32_t)__popc(won);
			}
			if (won1 != 0u) atomicAdd(&sh.cnt[1], won1);
			{ const uint32_t won = f2 & ~old2; sh.occ[8192u + threadIdx.x] = won; if (won != 0u) atomicAdd(&sh.cnt[2], (uint32_t)__popc(won)); }
			if (w3 < CUBE_WORDS) { const uint32_t won = f3 & ~(old3 >> sft3); sh.occ[w3] = won; if (won != 0u) atomicAdd(&sh.cnt[d3], (uint32_t)__popc(won)); }
			const bool hiMine = threadIdx.x >= LDS_LEVELS && threadIdx.x < depth;
			if (hiMine && sh.hiFresh[threadIdx.x] != 0u) {
				const unsigned long long ent = sh.anc[threadIdx.x];
				const uint32_t d = threadIdx.x + 1u, level = path_level(ent), s2 = 28u - (level + d), cell = grid_cell(level, LX << s2, LY << s2, LZ << s2);
				const uint32_t o = atomicOr(&path_grid(a.pers, ent)->values[cell >> 5], 1u << (cell & 31u));
				const uint32_t won = ((o >> (cell & 31u)) & 1u) ^ 1u;
				sh.hiFresh[threadIdx.x] = won;
				sh.cnt[d] = won;
			}
		}
		__syncthreads();
		ph.mark(28);
		// The cells this piece won in ancestor d become voxels: the add to Node.numVoxels (voxels.cu:101) reserves their slots in d's voxel
		// list, and the chunks those slots lie in are made or found here (see "voxel chunks, on demand") — one lane per ancestor, all in
		// wave 0: first every lane allocates and publishes what it owns, then every lane links / looks up (which may wait for another piece).
		{
			const uint32_t d = threadIdx.x;
			const bool mineD = d >= 1u && d <= depth && sh.cnt[d] != 0u;
			uint32_t node = 0, first = 0, existing = 0, kFirst = 0, ownFirst = 0, own = 0;
			SimlodChunk* mem = nullptr;
			if (mineD) {
				node = path_node(sh.anc[d - 1u]);
				const uint32_t cnt = sh.cnt[d];
				first = atomicAdd(&a.nodes[node].numVoxels, cnt);
				existing = (a.nodes[node].numVoxelsStored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
				kFirst = first / SIMLOD_POINTS_PER_CHUNK;
				const uint32_t kLast = (first + cnt - 1u) / SIMLOD_POINTS_PER_CHUNK;
				ownFirst = first % SIMLOD_POINTS_PER_CHUNK == 0u ? kFirst : kFirst + 1u;           // chunk k is this piece's to make when it holds slot k * 1000
				own = kLast + 1u > ownFirst ? kLast + 1u - ownFirst : 0u;
				if (own != 0u) mem = reinterpret_cast<SimlodChunk*>(persistent_alloc(a.pers, sizeof(SimlodChunk), own));   // voxel chunks never come from the pool (voxels.cu:656-659)
				for (uint32_t q = 0; q < own; q++) {
					SimlodChunk* c = reinterpret_cast<SimlodChunk*>(reinterpret_cast<uint8_t*>(mem) + (uint64_t)q * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
					if (q + 1u < own) c->next = reinterpret_cast<SimlodChunk*>(reinterpret_cast<uint8_t*>(c) + SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));   // (the last one's: vox_chunk_link)
					vox_chunk_publish(a, ctl, tag, node, ownFirst + q, c);
					sh.chunkOf[d][ownFirst + q - kFirst] = c;
				}
				sh.first[d] = first;
			}
			if (mineD) {
				SimlodChunk* oldTail = existing > 0u ? tail_of(a.nodes[node].voxelChunks) : nullptr;
				if (own != 0u) vox_chunk_link(a, ctl, tag, node, ownFirst, existing, oldTail, mem);
				if (ownFirst != kFirst) sh.chunkOf[d][0] = kFirst < existing ? oldTail : dir_wait(a, ctl, tag, node, kFirst);
			}
		}
		__syncthreads();
		ph.mark(29);

		// store: every winner whose cell is still marked won (nobody else had it) becomes a voxel — the cell's centre in the winner's colour
		// (voxels.cu:103-114, 674-698) — in the next free slot of the piece's range in that ancestor's list.  Level 1: the thread's own samples;
		// levels 2..7: the lists, an entry per lane; beyond: the one sample that climbed alone.
		auto store = [&](uint32_t d, const float4& vox) {
			const uint32_t slot = sh.first[d] + atomicAdd(&sh.rank[d], 1u);
			SimlodChunk* c = sh.chunkOf[d][slot / SIMLOD_POINTS_PER_CHUNK - sh.first[d] / SIMLOD_POINTS_PER_CHUNK];
			if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % SIMLOD_POINTS_PER_CHUNK] = vox;
		};
		for (uint32_t d = 1; d <= ldsDepth; d++) {
			const uint32_t n = sh.listCount[d];
			const uint32_t* src = sh.list + list_offset(d);
			for (uint32_t i = threadIdx.x; i < n; i += VTPB) {
				const uint32_t e = src[i];
				uint32_t word, bit;
				cube_cell_from(d, e & 0x3ffffu, word, bit);
				if (((sh.occ[word] >> bit) & 1u) != 0u) store(d, voxel_from(a, d, e & 0x3ffffu, leafLevel, LX, LY, LZ, sh.color[e >> 18]));
			}
		}
		if (threadIdx.x >= LDS_LEVELS && threadIdx.x < depth && sh.hiFresh[threadIdx.x] != 0u) {
			const uint32_t d = threadIdx.x + 1u, level = leafLevel - d, s2 = 28u - leafLevel;
			store(d, voxel_of(a, (int)level, LX << s2, LY << s2, LZ << s2, sh.color[sh.hiSample]));
		}
		ph.mark(30);
		if (ph.on) ctl->phaseNs[31] += 1;
	}
	if (clocked && threadIdx.x == 0 && blockIdx.x < numItems) atomicMax(reinterpret_cast<unsigned long long*>(&ctl->voxT[ordinal][1]), (unsigned long long)wall_ns());
	// then, wave by wave, the leaves with few new samples — handed out from the LAST wave down: the workgroups that had no piece start at once
	voxelize_small(a, ctl, bc, numWgs * VTPB / 64u - 1u - (blockIdx.x * VTPB + threadIdx.x) / 64u, numWgs * VTPB / 64u);
	if (clocked) { __syncthreads(); if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned long long*>(&ctl->voxT[ordinal][2]), (unsigned long long)wall_ns()); }
}

// ---- voxroot: a root that is still a leaf (the whole octree holds fewer than 50 000 points) samples ITSELF (voxels.cu:449-463: every node of the
// path that has a grid is sampled, and the root has one from the reset on) — into its own grid, sample by sample with device-scope atomics: the
// first batch or two of an octree, at most seven pieces.  The LAST workgroups of k_voxelize's launch (a path of its own, not a case of the main
// path: that cost the main path scratch memory; as a kernel of its own behind k_voxelize it cost every batch a launch); they leave at once when
// the group has no such piece.  Uses entry 0 of the piece state (cnt, first, rank, chunkOf).
__device__ __forceinline__ void voxroot_pieces(const BuildArgs& a, Ctl* ctl, BatchCtl* bc, VoxShared& sh, uint32_t wg, uint32_t numWgs) {
	const uint32_t numItems = min(bc->numVoxItems, VOX_BIG_ITEMS);
	const VoxItem* items = vox_items(a, bc);
	SimlodChunk* const* chunkDir = chunk_dir(a, bc);
	const uint32_t tag = bc->tag;
	for (uint32_t item = wg; item < numItems; item += numWgs) {
		VoxItem it = items[item];
		it.leaf &= 0xffffffu;
		if (it.leaf != 0u) continue;
		__syncthreads();
		if (threadIdx.x == 0u) { sh.cnt[0] = 0; sh.rank[0] = 0; }
		__syncthreads();
		SimlodOccupancyGrid* const g = a.nodes[0].grid;
		const unsigned long long ent = g != nullptr ? path_pack(a.pers, 0u, 0u, g) : 0ull;
		// a root that is still a leaf (fewer than 50 000 points in the whole octree): its own grid, sample by sample, one after the other
		// (this path runs for the first batch of an octree at most: nothing here is worth a register of the main path); the winners
		// share ONE voxel list (the root's): one reservation for the whole piece
		auto sample = [&](uint32_t j, uint32_t& pX, uint32_t& pY, uint32_t& pZ) -> float {
			const uint32_t i = it.s0 + j * VTPB + threadIdx.x;
			const float4 q = reinterpret_cast<const float4*>(chunkDir[it.ptBase + (i / SIMLOD_POINTS_PER_CHUNK - it.ptFirst)]->points)[i % SIMLOD_POINTS_PER_CHUNK];
			pX = quantize(F_FULL, q.x, a.minx, a.size); pY = quantize(F_FULL, q.y, a.miny, a.size); pZ = quantize(F_FULL, q.z, a.minz, a.size);
			return q.w;
		};
		uint32_t wonMask = 0;
		if (ent != 0ull) {
#pragma unroll 1
			for (uint32_t j = 0; j < VOX_SPT; j++) {
				if (it.s0 + j * VTPB + threadIdx.x >= it.s1) continue;
				uint32_t pX, pY, pZ;
				(void)sample(j, pX, pY, pZ);
				const uint32_t cell = grid_cell(0u, pX, pY, pZ), bit = cell & 31u;
				uint32_t* word = &path_grid(a.pers, ent)->values[cell >> 5];
				if (((*word >> bit) & 1u) != 0u) continue;                                         // voxels.cu:93-94
				if (((atomicOr(word, 1u << bit) >> bit) & 1u) != 0u) continue;                     // voxels.cu:96
				wonMask |= 1u << j;
				atomicAdd(&sh.cnt[0], 1u);
			}
		}
		__syncthreads();
		if (threadIdx.x == 0u && sh.cnt[0] != 0u) {
			const uint32_t cnt = sh.cnt[0];
			const uint32_t first = atomicAdd(&a.nodes[0].numVoxels, cnt);
			const uint32_t existing = (a.nodes[0].numVoxelsStored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
			const uint32_t kFirst = first / SIMLOD_POINTS_PER_CHUNK, kLast = (first + cnt - 1u) / SIMLOD_POINTS_PER_CHUNK;
			const uint32_t ownFirst = first % SIMLOD_POINTS_PER_CHUNK == 0u ? kFirst : kFirst + 1u;
			const uint32_t own = kLast + 1u > ownFirst ? kLast + 1u - ownFirst : 0u;
			SimlodChunk* mem = own != 0u ? reinterpret_cast<SimlodChunk*>(persistent_alloc(a.pers, sizeof(SimlodChunk), own)) : nullptr;
			for (uint32_t q = 0; q < own; q++) {
				SimlodChunk* c = reinterpret_cast<SimlodChunk*>(reinterpret_cast<uint8_t*>(mem) + (uint64_t)q * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
				if (q + 1u < own) c->next = reinterpret_cast<SimlodChunk*>(reinterpret_cast<uint8_t*>(c) + SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
				vox_chunk_publish(a, ctl, tag, 0u, ownFirst + q, c);
				sh.chunkOf[0][ownFirst + q - kFirst] = c;
			}
			SimlodChunk* oldTail = existing > 0u ? tail_of(a.nodes[0].voxelChunks) : nullptr;
			if (own != 0u) vox_chunk_link(a, ctl, tag, 0u, ownFirst, existing, oldTail, mem);
			if (ownFirst != kFirst) sh.chunkOf[0][0] = kFirst < existing ? oldTail : dir_wait(a, ctl, tag, 0u, kFirst);
			sh.first[0] = first;
		}
		__syncthreads();
#pragma unroll 1
		for (uint32_t j = 0; j < VOX_SPT; j++) {
			if (((wonMask >> j) & 1u) == 0u) continue;
			uint32_t pX, pY, pZ;
			const float colour = sample(j, pX, pY, pZ);
			const uint32_t slot = sh.first[0] + atomicAdd(&sh.rank[0], 1u);
			SimlodChunk* c = sh.chunkOf[0][slot / SIMLOD_POINTS_PER_CHUNK - sh.first[0] / SIMLOD_POINTS_PER_CHUNK];
			if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % SIMLOD_POINTS_PER_CHUNK] = voxel_of(a, 0, pX, pY, pZ, colour);
		}
	}
}

// ---- alloc: grow the chunk lists to their new lengths, build the per-batch chunk directory ----------------------
// (voxels.cu:485-538 allocatePointChunks, :641-672 allocateVoxelChunks, :298-300 countIteration stamp)

// ---- voxdone: the voxel lists are complete (voxels.cu:674-698: numVoxelsStored catches up with numVoxels) -------------------------------------
// After a batch's k_voxelize.  A node whose list grew gets its new tail: the last chunk's `next` is cleared and the head chunk remembers it
// (O(1) append next time).  `tag`: the batch whose hash directory names the new chunks.  Runs as part of the NEXT batch's k_insert (the
// first kernel on the caller's stream that has waited for the side stream; some workgroups at the end of its grid, which have no samples)
// and once more, as a kernel of its own, at the end of the launch.  Nodes the next batch has created meanwhile have no voxels: skipped.
__device__ void voxdone_nodes(const BuildArgs& a, Ctl* ctl, uint32_t tag, uint32_t numNodes, uint32_t first, uint32_t stride, bool skipRoot = false) {
	for (uint32_t i = first; i < numNodes; i += stride) {
		if (skipRoot && i == 0u) continue;
		SimlodNode* node = a.nodes + i;
		const uint32_t numVoxels = node->numVoxels, stored = node->numVoxelsStored;
		if (numVoxels == stored) continue;
		const uint32_t existing = (stored + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK, last = (numVoxels - 1u) / SIMLOD_POINTS_PER_CHUNK;
		if (last >= existing) {
			SimlodChunk* tail = dir_find(a, tag, i, last);
			if (tail != nullptr) { tail->next = nullptr; tail_of(node->voxelChunks) = tail; } else raise(ctl, SIMLOD_ERR_NULL_CHUNK);
		}
		node->numVoxelsStored = numVoxels;
	}
}
__global__ __launch_bounds__(TPB) void k_voxdone(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (ctl->processed == 0u || ctl->abortBatch) return;
	voxdone_nodes(a, ctl, ctl->tagOf[ctl->processed - 1u], min(a.stats->numNodes, a.nodeCapacity), blockIdx.x * TPB + threadIdx.x, gridDim.x * TPB);   // (the launch's last group)
}

// ---- rootpre: an exact group of several batches in which the ROOT splits ------------------------------------------------------------------
// A root that is still a leaf samples ITSELF (voxels.cu:449-463: every node of the path that has a grid, and the root has one from the reset on), and the
// batch that splits it clears that grid and samples everything again from nothing (voxels.cu:371-382): the root's voxel list keeps what the batches
// before the split put there AND gets every cell again.  Ingested batch by batch that is what happens; in a group, the batches in front of the
// splitting one (k_expand knows which: BatchCtl.rootSplitAt) have to meet the root's grid as those batches would have — before k_insert clears it.
// One workgroup on the back stream in front of k_insert; leaves at once in every group but the one or two of an octree's life that split its root.
__global__ __launch_bounds__(1024) void k_rootpre(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr || ctl->abortBatch || bc->acct == 0u) return;
	const uint32_t sR = bc->rootSplitAt;
	if (sR == NONE || sR == 0u) return;
	// the root's voxel list as the group before left it (its k_voxelize has ended: stream order): closed here, in front of this group's first voxels
	if (threadIdx.x == 0u && ordinal > 0u) voxdone_nodes(a, ctl, ctl->tagOf[ordinal - 1u], 1u, 0u, 1u);
	__threadfence();
	__syncthreads();
	const uint32_t n = bc->start[min(sR, bc->groupBatches)], lane = (uint32_t)lane_id();
	const Samples<false> pts(a, bc);
	SimlodOccupancyGrid* const g = a.nodes[0].grid;
	if (g == nullptr) return;
	for (uint32_t base = 0; base < n; base += blockDim.x) {
		const uint32_t i = base + threadIdx.x;
		bool go = i < n;
		const float4 p = go ? pts[i] : make_float4(0, 0, 0, 0);
		const uint32_t pX = quantize(F_FULL, p.x, a.minx, a.size), pY = quantize(F_FULL, p.y, a.miny, a.size), pZ = quantize(F_FULL, p.z, a.minz, a.size);
		const uint32_t cell = grid_cell(0u, pX, pY, pZ), bit = cell & 31u;
		uint32_t* word = &g->values[cell >> 5];
		go = go && ((*word >> bit) & 1u) == 0u;                                                  // voxels.cu:93-94
		go = go && ((atomicOr(word, go ? 1u << bit : 0u) >> bit) & 1u) == 0u;                      // voxels.cu:96
		const unsigned long long wm = __ballot(go);
		if (wm == 0ull) continue;
		uint32_t first = 0;
		if (lane == 0u) first = atomicAdd(&a.nodes[0].numVoxels, (uint32_t)__popcll(wm));       // voxels.cu:101
		first = (uint32_t)__shfl((int)first, 0, 64);
		store_voxels_wave(a, ctl, bc->tag, go, 0u, first + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull)), voxel_of(a, 0, pX, pY, pZ, p.w));
	}
}

// ---- insert: points into leaf chunks, regenerated voxels into voxel chunks (voxels.cu:540-639, 674-698) --------
struct InsertShared {
	BlockTable tbl;                       // node -> count (step 1), then node -> running cursor (step 3)
	uint32_t base[TBL_CAP];               // first slot of the range this workgroup reserved in the node
	uint32_t dirBase[TBL_CAP];            // chunk-directory base of the node for this batch, or 0xffffffff
	uint32_t dirFirst[TBL_CAP];
};

// The points go into their leaves (before k_voxelize, which reads them back leaf by leaf), and with them what used to be kernels of their
// own: the clearing of the grids of the nodes this batch split, the end-of-batch bookkeeping.  The chunks are there already (k_hist,
// k_expand).  In steps: (1) everybody counts its samples per leaf in an LDS table — a relabelled sample finds its leaf in its slot's map —
// (2) reserves one slot range per (workgroup, leaf) with one global atomic each, (3) stores: the slot inside the range comes from an
// LDS cursor.  Barriers are paid per workgroup, not per chunk.
template <bool SINGLE>
__global__ __launch_bounds__(TPB) void k_insert(BuildArgs a, uint32_t ordinal) {
	Ctl* ctl = ctl_of(a);
	BatchCtl* bc = batch_of(ctl, ordinal);
	if (bc == nullptr) return;
	if (ctl->abortBatch) {                                  // an earlier kernel gave up: the batch is not counted (Stats.dbg says why)
		if (blockIdx.x == 0 && threadIdx.x == 0) end_of_batch(a, ctl, bc);
		return;
	}
	__shared__ InsertShared sh;
	const uint32_t n = bc->batchSize;
	const uint32_t total = n + min(bc->numSpilled, a.spilledCap);
	const Samples<SINGLE> pts(a, bc);
	const float4* spilled = at<const float4>(a, a.offSpilled);
	const LeafWords leafOf(a, ordinal);
	const NodeDir* nodeDir = at<const NodeDir>(a, a.offNodeDir);
	SimlodChunk* const* chunkDir = chunk_dir(a, bc);
	const uint32_t tag = bc->tag;
	const uint32_t numChunks = (total + PPB - 1) / PPB;

	{
		Phase ph(ctl, blockIdx.x == 0 || blockIdx.x + 1u == numChunks);
		const uint32_t pb = blockIdx.x == 0 ? 8u : 16u;
		// (The end-of-batch bookkeeping — voxels.cu:925-949: Stats.batchletIndex, numPointsProcessed, the time budget — is k_voxelize's first act:
		// a host that watches Stats.batchletIndex outside stream order, as the reference's uploader does with its back-pressure rule
		// (main_progressive_octree.cpp:1012), may refill the group's ring slots the moment the index moves, and until this kernel has ENDED
		// its workgroups may still be reading them.  Counting the workgroups in with an atomic each was measured: 2 048 adds on one word,
		// 4.7 ms per ingest instead of 3.7.)
		// ... and the 32 before it close the voxel lists of the previous batch (voxdone_nodes: this kernel has waited for its k_voxelize)
		{
			constexpr uint32_t DONE_WGS = 32;
			const uint32_t back = gridDim.x - 1u - blockIdx.x;
			if (back >= 1u && back <= DONE_WGS && bc->ordinal != 0u)
				voxdone_nodes(a, ctl, ctl->tagOf[bc->ordinal - 1u], min(a.stats->numNodes, a.nodeCapacity), (back - 1u) * TPB + threadIdx.x, DONE_WGS * TPB,
				              bc->acct != 0u && bc->rootSplitAt != 0xffffffffu && bc->rootSplitAt != 0u);      // (k_rootpre has closed the root's list and appended to it since)
		}
		ph.mark(pb + 0);
		// the occupancy grids of the nodes this batch split (k_count's tail and k_expand listed them): cleared here, by everybody, before
		// k_voxelize samples into them (voxels.cu:371-382) — 256 KB each, the stores ride along with the loads below
		const uint32_t numClear = min(bc->numClear, a.clearCap);
		SimlodOccupancyGrid* const* clearList = clear_list(a, ordinal);
		constexpr uint32_t W4 = SIMLOD_GRID_NUM_WORDS / 4;
		for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < numClear * W4; i += gridDim.x * TPB)
			reinterpret_cast<uint4*>(clearList[i / W4]->values)[i % W4] = make_uint4(0, 0, 0, 0);
		if (blockIdx.x >= numChunks) return;

		ph.mark(pb + 1);
		// (1) samples per leaf
		table_init(sh.tbl);
		__syncthreads();
		const uint32_t* map = at<const uint32_t>(a, a.offMap);
		for (uint32_t chunk = blockIdx.x; chunk < numChunks; chunk += gridDim.x) {
			uint32_t v[PPT];
#pragma unroll
			for (uint32_t j = 0; j < PPT; j++) {
				const uint32_t t = chunk * PPB + j * TPB + threadIdx.x;
				v[j] = t < total ? leafOf[t < n ? t : a.groupCap + (t - n)] : NONE;
			}
#pragma unroll
			for (uint32_t j = 0; j < PPT; j++) {
				// a sample that k_hist / k_expand relabelled (slot, bin): its leaf is one word of the slot's map; any other: the leaf k_count found
				if (v[j] == NONE) continue;
				if ((v[j] & LEAF_FLAG) == 0u) { v[j] &= LEAF_NODE_MASK; continue; }
				uint32_t e = map[v[j] & 0x1fffffu];
				if ((e & MAP_LISTED) != 0u) e = slot_recs(a, ordinal)[e & 0xffffu].node;      // (a node that got a slot but no round any more: it stays a leaf)
				v[j] = e;
				const uint32_t t = chunk * PPB + j * TPB + threadIdx.x;
				leafOf[t < n ? t : a.groupCap + (t - n)] = e;
			}
#pragma unroll
			for (uint32_t j = 0; j < PPT; j++) {
				if (v[j] == NONE) continue;
				uint32_t rank;
				(void)table_add(sh.tbl, v[j], 1u, &rank);
			}
		}
		__syncthreads();
		ph.mark(pb + 2);
		// (2) one slot range per (workgroup, leaf)
		for (uint32_t e = threadIdx.x; e < (uint32_t)TBL_CAP; e += TPB) {
			const uint32_t key = sh.tbl.keys[e];
			if (key == TBL_EMPTY) continue;
			const NodeDir d = nodeDir[key];
			sh.base[e] = atomicAdd(&a.nodes[key].numPoints, sh.tbl.vals[e]);                      // voxels.cu:593
			sh.tbl.vals[e] = 0;                                                                    // becomes the cursor
			sh.dirBase[e] = d.ptTag == tag ? d.ptBase : 0xffffffffu;
			sh.dirFirst[e] = d.ptFirst;
		}
		__syncthreads();
		ph.mark(pb + 3);
		for (uint32_t chunk = blockIdx.x; chunk < numChunks; chunk += gridDim.x) {
			float4 p[PPT];
			const float4* base = nullptr; uint32_t kspan;
			const bool oneBatch = chunk * PPB < n && pts.span(chunk * PPB, min(n, (chunk + 1u) * PPB) - 1u, base, kspan);      // (workgroup-uniform)
#pragma unroll
			for (uint32_t j = 0; j < PPT; j++) {
				const uint32_t t = chunk * PPB + j * TPB + threadIdx.x;
				p[j] = t >= total ? make_float4(0, 0, 0, 0) : (t < n ? (oneBatch ? base[t] : pts[t]) : spilled[t - n]);
			}
#pragma unroll
			for (uint32_t j = 0; j < PPT; j++) {
				const uint32_t t = chunk * PPB + j * TPB + threadIdx.x;
				if (t >= total) continue;
				const uint32_t leafIdx = leafOf[t < n ? t : a.groupCap + (t - n)] & LEAF_NODE_MASK;
				const int e = table_find(sh.tbl, leafIdx);
				uint32_t slot, base, first;
				if (e >= 0) { slot = sh.base[e] + atomicAdd(&sh.tbl.vals[e], 1u); base = sh.dirBase[e]; first = sh.dirFirst[e]; }
				else {                                                                                // table had no room for this leaf
					const NodeDir d = nodeDir[leafIdx];
					slot = atomicAdd(&a.nodes[leafIdx].numPoints, 1u); base = d.ptTag == tag ? d.ptBase : 0xffffffffu; first = d.ptFirst;
				}
				if (base == 0xffffffffu) { raise(ctl, SIMLOD_ERR_NULL_CHUNK); continue; }           // voxels.cu:599-604
				SimlodChunk* c = chunkDir[base + (slot / SIMLOD_POINTS_PER_CHUNK - first)];
				reinterpret_cast<float4*>(c->points)[slot % SIMLOD_POINTS_PER_CHUNK] = p[j];
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		ph.mark(pb + 5);
		if (ph.on) ctl->phaseNs[pb + 6] += 1;
	}
}

// ---- stats pass (voxels.cu:957-1009) -------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

__global__ __launch_bounds__(TPB) void k_stats(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	uint32_t v[7] = {0, 0, 0, 0, 0, 0, 0};   // inner, leaves, nonempty, points, voxels, chunksP, chunksV
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < numNodes; i += gridDim.x * TPB) {      // (a grid for the nodes that exist, not for the node capacity)
		SimlodNode* n = a.nodes + i;
		// voxels.cu:298-300: every counting pass stamps every node with (index of the batch + 1); what the host can see is the last stamp
		if (ctl->processed != 0u) n->countIteration = a.stats->batchletIndex;
		if (node_is_leaf(n)) {
			v[1] += 1; v[3] += n->numPoints; v[5] += (n->numPoints + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
			v[2] += n->numPoints > 0 ? 1u : 0u;
		} else {
			v[0] += 1; v[4] += n->numVoxels; v[6] += (n->numVoxels + SIMLOD_POINTS_PER_CHUNK - 1) / SIMLOD_POINTS_PER_CHUNK;
		}
	}
	for (int k = 0; k < 7; k++) {
		const uint32_t s = wave_sum(v[k]);
		if (lane_id() == 0 && s != 0u) atomicAdd(&ctl->statCounters[k], s);
	}
}

__global__ void k_finish(BuildArgs a, uint32_t fits, uint32_t* feedback, const uint32_t* numBatchesUploaded, uint32_t launchSeq) {
	if (blockIdx.x != 0) return;
	Ctl* ctl = ctl_of(a);
	SimlodStats* s = a.stats;
	__shared__ Phantom sh_phantom;
	// voxels.cu:535-537 for the launch's last group (k_count's first workgroup: the others); an exact group of several batches: account_group
	if (threadIdx.x == 0) sh_phantom = account_group(a, ctl, ctl->processed != 0u ? batch_of(ctl, ctl->processed - 1u) : nullptr);
	__syncthreads();
	phantom_fill(a, sh_phantom);
	if (threadIdx.x != 0) return;
	if (feedback != nullptr) {      // what the next launch sizes itself by (groups_for_launch): page-locked host memory
		feedback[0] = s->batchletIndex; feedback[1] = *numBatchesUploaded;
		// ... and whether its batches can go in groups (exact mode, prepare_batch decides for every group; a launch whose groups would each be cut down to
		// one batch had better enqueue one group of kernels per batch): the allocator is a worst-case full group away from the guard
		const SimlodAllocatorGlobal* alloc = reinterpret_cast<const SimlodAllocatorGlobal*>(a.pers);
		feedback[2] = a.acct != 0u &&
		              alloc->offset + SIMLOD_MEM_SAFETY_MARGIN + slack_for(a, (unsigned long long)a.groupCap + min((unsigned long long)a.spilledCap, (unsigned long long)s->numPointsProcessed)) < a.persCapacity ? 1u : 0u;
		__hip_atomic_store(feedback + 3, launchSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // (last: a reader that sees this launch's number sees its report)
	}
	s->numInner = ctl->statCounters[0];
	s->numLeaves = ctl->statCounters[1];
	s->numNonemptyLeaves = ctl->statCounters[2];
	s->numPoints = ctl->statCounters[3];
	s->numVoxels = ctl->statCounters[4];
	s->numChunksPoints = ctl->statCounters[5];
	s->numChunksVoxels = ctl->statCounters[6];
	s->allocatedBytes_momentary = a.scratchBytes;
	s->allocatedBytes_persistent = reinterpret_cast<const SimlodAllocatorGlobal*>(a.pers)->offset;
	s->frameID = (uint32_t)a.frameCounter;
	s->dbg |= ctl->errors;
	if (fits && !ctl->abortBatch) {              // the side tables describe THIS octree as it is after THIS batch
		ctl->tableBatch = s->batchletIndex;
		ctl->tableNodes = (uint64_t)a.nodes;
		ctl->tablePers = (uint64_t)a.pers;
		ctl->tableSig = table_signature(s);
		ctl->tableLayout = layout_signature(a);
		ctl->tableMagic = TABLE_MAGIC;
	}
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

// exactGroup > 1: EXACT mode in groups of that many batches (account_group) — the per-batch count rows are part of the layout; false when the buffer
// does not hold them beside ACCT_SPILL_FLOOR moved points (the caller then lays out plain exact mode: one batch per group)
static constexpr uint64_t ACCT_SPILL_FLOOR = 3500000;      // moved points an exact group must have room for: what one BATCH has in the plain exact layout of the reference host's 300 MB
bool layout_construct(BuildArgs& a, uint64_t capacity, bool coalesce, uint32_t groupLimit = SIMLOD_MAX_BATCHES_PER_LAUNCH, uint32_t exactGroup = 1) {
	a.dirCap = 2 * a.nodeCapacity + 65536;
	a.acct = exactGroup > 1u ? 1u : 0u;
	uint64_t off = 4096;
	a.offQueue = off;    off += align_up((uint64_t)CHUNK_QUEUE_CAPACITY * 8, 256);
	a.offSlots = off;    off += align_up((uint64_t)2 * SLOT_CAP * sizeof(SlotRec), 256);
	a.offHist = off;     off += align_up(((uint64_t)SLOT_CAP * HIST_BINS + HIST_EXTRA_WORDS) * 4, 256);      // (+ three more copies of the first 256 slots)
	a.offMap = off;      off += align_up((uint64_t)SLOT_CAP * HIST_BINS * 4, 256);
	a.offHistB = off;    off += align_up((uint64_t)(a.acct ? SLOT_CAP : 0u) * HIST_BINS * exactGroup * 4, 256);
	a.clearCap = 65536;
	a.offClear = off;    off += align_up((uint64_t)2 * a.clearCap * 8, 256);
	a.offTouched = off;  off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.crossCap = 16384;
	a.offCross = off;    off += align_up((uint64_t)a.crossCap * 4, 256);
	// (cleared by the host-enqueued memset of every launch, offSplitTag .. offParent: split records, retry / touch tags, the counters at batch
	// start, the hash directory of voxel chunks)
	a.offSplitTag = off; off += align_up((uint64_t)a.nodeCapacity * 8, 256);
	a.offRetryTag = off; off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offTouchTag = off; off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offStartOf = off;  off += align_up((uint64_t)a.nodeCapacity * 8, 256);
	a.offCntB = off;     off += align_up((uint64_t)(a.acct ? a.nodeCapacity : 0u) * exactGroup * 4, 256);      // (zero between groups: whoever reads a row clears it)
	a.hashCap = 1u << 17;
	a.offHashDir = off;  off += align_up((uint64_t)a.hashCap * sizeof(DirEntry), 256);
	a.offParent = off;   off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offNodeDir = off;  off += align_up((uint64_t)a.nodeCapacity * sizeof(NodeDir), 256);
	a.offChunkDir = off; off += align_up(2ull * a.dirCap * 8, 256);            // (two copies, by batch parity)
	a.offLeafChunks = off; off += (uint64_t)a.nodeCapacity * LEAF_ROW_BYTES;
	a.offPaths = off; off += align_up((uint64_t)a.nodeCapacity * PATH_WORDS * 8, 256);
	a.offTop = off;   off += align_up((uint64_t)TOP_CELLS * 4, 256);
	a.offKid = off;   off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.voxItemCap = min(a.nodeCapacity + 2u * VOX_BIG_ITEMS, 1u << 20);         // VOX_BIG_ITEMS pieces + small items: a leaf has one more than its new samples / 128, and 65 536 x 128 = 8 M samples
	a.offVoxItems = off; off += align_up(2ull * a.voxItemCap * sizeof(VoxItem), 256);   // (two copies, by batch parity)
	// what is left is shared by the per-sample arrays: the 4-byte cached-leaf word of the group's and of the moved samples, 16 B per moved point.
	// Exact mode: a group is one ring batch.  Coalesced mode: as many batches per group (up to 20) as leave room for a million moved points
	// per batch (with the reference host's 300 MB that is 2-3 batches; the mode wants ~700 MB for groups of 20).
	const uint64_t perBatch = (uint64_t)SIMLOD_MAX_BATCH_SIZE * 4;
	const uint64_t fixedWork = ((uint64_t)SPILLING_CAPACITY + a.nodeCapacity / 8) * 32;
	a.groupMax = 1; a.groupCap = SIMLOD_MAX_BATCH_SIZE;
	if (capacity < off + 2 * perBatch + fixedWork + 4096 + 25ull * 65536) { a.spilledCap = 0; a.scratchBytes = off + 2 * perBatch + fixedWork; return false; }
	const uint64_t freeBytes = capacity - off - fixedWork - 4096;
	if (coalesce) a.groupMax = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(groupLimit, SIMLOD_MAX_BATCHES_PER_LAUNCH), freeBytes / (2 * perBatch + 20ull * SIMLOD_MAX_BATCH_SIZE)));
	if (a.acct) {
		if (freeBytes < 2ull * exactGroup * perBatch + 512 + ACCT_SPILL_FLOOR * 20032 / 1000) return false;
		a.groupMax = exactGroup;
	}
	a.groupCap = a.groupMax * SIMLOD_MAX_BATCH_SIZE;
	// (the group samples' cached-leaf words exist twice, by group parity; 4 + 16 B per moved point)
	uint64_t cap = (freeBytes - 2ull * a.groupMax * perBatch - 512) * 1000 / (20 * 1000 + 32);   // + one 32-byte work item per 1000 moved points
	if (cap > 0x7fffffffull - a.groupCap) cap = 0x7fffffffull - a.groupCap;
	a.spilledCap = (uint32_t)cap;
	a.workCap = a.spilledCap / SIMLOD_POINTS_PER_CHUNK + a.nodeCapacity / 8 + SPILLING_CAPACITY;   // one item per 1000 moved points + one partial chunk per split
	a.offWork = off;     off += align_up((uint64_t)a.workCap * 32, 256);
	a.leafOfStride = align_up((uint64_t)a.groupCap * 4, 256) / 4;
	a.offLeafOf = off;   off += 2 * a.leafOfStride * 4 + align_up((uint64_t)a.spilledCap * 4, 256);
	a.offSpilled = off;  off += (uint64_t)a.spilledCap * 16;
	a.scratchBytes = off;
	return off <= capacity;
}

}  // namespace build
// a context's second stream for the back halves of the batches, and the events that tie it to the caller's stream (one pair per group of a launch)
struct SideStream {
	hipStream_t stream;
	hipEvent_t expanded[SIMLOD_MAX_BATCHES_PER_LAUNCH], inserted[SIMLOD_MAX_BATCHES_PER_LAUNCH], tailDone;
	std::mutex enqueue;          // the events are reused by every launch of the context: one launch's records and waits are enqueued as a block
};
void destroy_side_stream(SideStream* s) {
	(void)hipStreamSynchronize(s->stream);
	for (uint32_t i = 0; i < SIMLOD_MAX_BATCHES_PER_LAUNCH; i++) { (void)hipEventDestroy(s->expanded[i]); (void)hipEventDestroy(s->inserted[i]); }
	(void)hipEventDestroy(s->tailDone);
	(void)hipStreamDestroy(s->stream);
	delete s;
}
namespace build {
// The events order kernels of ONE device: no timing, and no system-scope fence when one is recorded (its cache write-back and invalidation
// are for the host and for other devices; a kernel's own end makes its stores visible to the kernels that follow on this device).
#ifndef SYNC_EVENT_FLAGS
#define SYNC_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
static SideStream* side_stream(Context& ctx) {
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev < 0 || dev >= 64) return nullptr;
	std::lock_guard<std::mutex> hold(ctx.sideLock);
	if (ctx.side[dev] == nullptr) {
		SideStream* s = new SideStream();
		bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
		// SIMLOD_EVENT_SYSTEM_FENCE=1 (read when the context's second stream is made): events WITH the system-scope release — the fallback should a
		// ROCm release ever stop publishing a kernel's stores to the other stream's kernels without it (the parity tests and
		// test_repeated_ingests_leave_identical_counters would show it)
		const unsigned flags = ctx.tune(KNOB_EVENT_SYSTEM_FENCE, 0) != 0 ? (unsigned)hipEventDisableTiming : (unsigned)SYNC_EVENT_FLAGS;
		for (uint32_t i = 0; ok && i < SIMLOD_MAX_BATCHES_PER_LAUNCH; i++)
			ok = hipEventCreateWithFlags(&s->expanded[i], flags) == hipSuccess && hipEventCreateWithFlags(&s->inserted[i], flags) == hipSuccess;
		ok = ok && hipEventCreateWithFlags(&s->tailDone, flags) == hipSuccess;
		if (!ok) { (void)hipGetLastError(); delete s; return nullptr; }     // no side stream: everything stays on the caller's
		ctx.side[dev] = s;
	}
	return ctx.side[dev];
}

int launch_construct(Context& ctx, const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream) {
	BuildArgs a{};
	a.ring = points; a.mom = reinterpret_cast<uint8_t*>(buffer); a.pers = pers; a.nodes = nodes; a.stats = stats;
	a.frameStart = frameStart; a.numBatchesUploaded = numBatchesUploaded; a.batchSizes = batchSizes;
	const float bx = u->boxMax.x - u->boxMin.x, by = u->boxMax.y - u->boxMin.y, bz = u->boxMax.z - u->boxMin.z;
	a.size = fmaxf(fmaxf(bx, by), bz);                                         // voxels.cu:860-863
	a.minx = u->boxMin.x; a.miny = u->boxMin.y; a.minz = u->boxMin.z;
	a.persCapacity = u->persistentBufferCapacity;
	a.frameCounter = u->frameCounter;
	a.nodeCapacity = ctx.nodeCapacity.load();
	a.trunkLo = ctx.trunkLo.load(); a.trunkHi = ctx.trunkHi.load();
	const bool coalesce = ctx.ingestMode.load() != 0u;
	// EXACT mode: a launch that finds several batches pending ingests them in groups of up to SIMLOD_EXACT_GROUP (default 5; 1: one by one) wherever the
	// momentary buffer holds the largest such layout that leaves room for ACCT_SPILL_FLOOR moved points — every Node and Stats field comes out as batch-by-batch
	// ingestion leaves it (account_group).  Not with a forced time budget (the budget is looked at per group: voxels.cu:936-949 looks per batch).
	uint32_t exactGroup = coalesce || ctx.tune(KNOB_DEBUG_BUDGET_US, 0) > 0 ? 1u : (uint32_t)std::min<int>(std::max(1, ctx.tune(KNOB_EXACT_GROUP, 5)), (int)ACCT_MAX_GROUP);
	// (a persistent buffer so small that even an empty octree is closer to the reference's memory guard than a worst-case group of two batches: prepare_batch
	// would cut every group down to one batch — plain exact mode, without the groups' bookkeeping)
	if (exactGroup > 1u && SIMLOD_MEM_SAFETY_MARGIN + group_slack_bytes(2ull * SIMLOD_MAX_BATCH_SIZE, 0u) >= a.persCapacity) exactGroup = 1u;
	bool fits = false;
	for (; exactGroup > 1u && !fits; exactGroup -= fits ? 0u : 1u) fits = layout_construct(a, u->momentaryBufferCapacity, false, 1, exactGroup);
	if (!fits) fits = layout_construct(a, u->momentaryBufferCapacity, coalesce, (uint32_t)std::max(1, ctx.tune(KNOB_GROUP_BATCHES, 10)));   // coalesced mode: groups of 10 (36 M terrain: 20: 3.17 ms, 10: 2.95, 5: 3.10, 2: 3.66 — two groups per launch overlap front and back halves)
	if (fits && ((uint64_t)a.mom & (LEAF_ROW_BYTES - 1u)) == 0ull) {   // the rasteriser reads leaf lists through the table while its stamp matches the octree (render.hip r_visible; a draw item names a row by its address, 256-byte aligned)
		const Ctl* ctl = reinterpret_cast<const Ctl*>(a.mom);
		note_leaf_table(ctx, LeafTableRef{nodes, a.mom, a.mom + a.offLeafChunks, a.pers, &ctl->tableMagic, &ctl->tableBatch,
		                             &ctl->tableNodes, &ctl->tableSig, TABLE_MAGIC, LEAF_SLOTS, a.nodeCapacity});
	} else forget_leaf_table(ctx, nodes);
	const DeviceInfo& dev = device_info();

	note_upload_counter(numBatchesUploaded, 0u, false, true);
	const LaunchPlan plan = launch_plan(ctx, stats, numBatchesUploaded);
	const uint32_t limit = plan.batches;
	// batches per group this launch aims at: the layout's, or — exact mode, when the latest launch reported that groups of several batches are out of the
	// question for now (LaunchPlan::mayGroup) — one, with one group of kernels per batch
	const uint32_t take = a.acct != 0u && !plan.mayGroup ? 1u : a.groupMax;
	// (one workgroup does the launch's bookkeeping; all of them restore the side tables when the stamp is stale: the first launch of an octree, as a rule)
	SIMLOD_LAUNCH(k_begin, dim3(fits ? dev.numCUs * 2 : 1u), dim3(TPB), stream, a, fits ? 0u : 1u, limit, ((uint32_t)ctx.tune(KNOB_DEBUG_FORCE_BARRIER_TIMEOUT, 0) & 1u) | (ctx.tune(KNOB_DEBUG_VOXELIZE_CLOCK, 0) != 0 ? 2u : 0u) | (ctx.sideTablesStale.exchange(false) ? 4u : 0u) | (((uint32_t)ctx.tune(KNOB_DEBUG_PHASE_WG, 0) & 0xffffu) << 8),
	              (uint32_t)std::max(0, ctx.tune(KNOB_DEBUG_BUDGET_US, 0)), take);
	if (fits) {
		const uint32_t gridPoints = dev.numCUs * (uint32_t)ctx.tune(KNOB_GRID_MULT, 8);
		// k_expand's workgroups meet at grid barriers: never more than one per CU (all must be resident).  One per TWO CUs is the
		// measured optimum on MI355X (36 M terrain, us per batch: 256 -> 104, 192 -> 93, 128 -> 83, 96 -> 82, 64 -> 84, 32 -> 107):
		// the barrier's agent-scope release / acquire and the polling cost grow with the participants, the work does not need them
		// (with the previous batch's voxel half running beside it on the side stream: one per FOUR CUs — 256: 8.3 ms per ingest, 128: 7.8,
		// 96: 7.5, 64: 7.2, 48: 7.3, 32: 7.5)
		// kernel groups to enqueue: one per ring batch, or per groupMax of them (coalesced mode, exact mode in groups)
		const uint32_t numGroups = (limit + take - 1u) / take;
		const bool overlap = ctx.tune(KNOB_OVERLAP_TAIL, 1) != 0 && !profile_enabled() && numGroups > 1u;   // (two streams: see below)
		// (coalesced mode: a group's later rounds pass over tens of millions of samples inside k_expand — every CU takes part: 3.49 -> 2.78 ms per 36 M)
		const bool single = take == 1u || limit <= 1u;      // every group of this launch is ONE ring batch (k_begin takes no more than `limit` batches)
		const uint32_t expandWgs = (uint32_t)max(1, min(ctx.tune(KNOB_EXPAND_WGS, !single ? (int)dev.numCUs : (int)dev.numCUs / (overlap ? 4 : 2)), (int)dev.numCUs));
		// A batch has a FRONT half — k_count, k_queue, k_hist, k_expand: the tree grows, every chunk the batch's points need is
		// allocated — and a BACK half — k_insert, k_voxelize: the points are stored, the voxels sampled and stored.  The front half runs on
		// the caller's stream, the back half on a second stream of the library, two dependencies per batch between them:
		//     k_insert(b) after k_expand(b);                 k_hist(b + 1) after k_insert(b)   (it moves points k_insert(b) has stored).
		// So k_count and k_queue(b + 1) run beside k_insert(b), k_hist and k_expand(b + 1) beside k_voxelize(b), and k_insert(b + 1) follows k_voxelize(b)
		// by stream order (it overwrites chunks that k_expand(b + 1) recycled and k_voxelize(b) may still be reading).  What the halves share
		// exists per batch: the control state in copies b & 3, the chunk directory, the work items and the cached-leaf words by parity;
		// k_count does not look at Node.numPoints (stored_at_start()).  Per batch the chain is as long as its longest cycle — k_insert,
		// event, k_queue + k_hist + k_expand, event: ~90 us — instead of the sum of all seven kernels (~190 us on one stream).
		// Off while per-kernel profiling is on (one stream, one timeline) or with SIMLOD_OVERLAP_TAIL=0.
		// A launch of ONE group has nothing to overlap: its seven kernels on the caller's stream, without the two event hops (13 + 12 us) and the stop
		// event's gap (6 us) — a launch that finds one batch: 166 -> ~135 us (tools/launch_cost.py; the reference's frame loop while the loader is the bottleneck).
		SideStream* side = overlap ? side_stream(ctx) : nullptr;
		std::unique_lock<std::mutex> block;
		if (side != nullptr) block = std::unique_lock<std::mutex>(side->enqueue);      // (host threads building two octrees on one device)
		const int countTpb = ctx.tune(KNOB_COUNT_TPB, 512);   // fewer, fatter workgroups: fewer adds on the hot leaf counters (flush 7.5 -> 2.7 us at 512, main loop 11.1 -> 12.7)
		hipStream_t back = side != nullptr ? side->stream : stream;
		// an enqueue that fails in the middle of the chain: the second stream may hold kernels that read and write the caller's buffers — the
		// call does not return before they have ended (the caller may free or reset those buffers next)
		auto fail = [&](hipError_t e) { if (side != nullptr) (void)hipStreamSynchronize(side->stream); (void)hipGetLastError(); return (int)(e != hipSuccess ? e : hipErrorUnknown); };
		const uint32_t voxWgs = (uint32_t)max(1, ctx.tune(KNOB_VOXELIZE_WGS, (int)dev.numCUs * 2)) + VOXROOT_WGS;     // (+ the workgroups of a root that is still a leaf)
		for (uint32_t b = 0; b < numGroups; b++) {
			if (single) {
				if (countTpb == 256) SIMLOD_LAUNCH((k_count<TPB, true>), dim3(gridPoints), dim3(TPB), stream, a, b);
				else SIMLOD_LAUNCH((k_count<512, true>), dim3(gridPoints / 2), dim3(512), stream, a, b);
			} else SIMLOD_LAUNCH((k_count<512, false>), dim3(gridPoints / 2), dim3(512), stream, a, b);
			SIMLOD_LAUNCH(k_queue, dim3(single ? 16 : 128), dim3(TPB), stream, a, b);   // one wave per crossing leaf: a couple per batch, hundreds per coalesced group (36 M terrain, groups of 10: 42 us on 16 workgroups)
			if (side != nullptr && b > 0) { const hipError_t e = hipStreamWaitEvent(stream, side->inserted[b - 1], 0); if (e != hipSuccess) return fail(e); }
			SIMLOD_LAUNCH(k_hist, dim3(gridPoints), dim3(TPB), stream, a, b);
			// (with two streams the kernels the other stream waits for carry their event as the launch's stop event: it is signalled by the
			// kernel's own completion, where hipEventRecord puts a marker of its own behind the kernel — 3.93 -> 3.85 ms per ingest)
			if (side != nullptr) {
				const bool gated = expand_gate_enter(ctx, stream);
				SIMLOD_LAUNCH_STOP(k_expand, dim3(expandWgs), dim3(ETPB), stream, side->expanded[b], a, b);
				expand_gate_leave(ctx, stream, side->expanded[b], gated);
				{ const hipError_t e = hipStreamWaitEvent(back, side->expanded[b], 0); if (e != hipSuccess) return fail(e); }
				if (!single && a.acct != 0u) SIMLOD_LAUNCH(k_rootpre, dim3(1), dim3(1024), back, a, b);
				if (single) SIMLOD_LAUNCH_STOP(k_insert<true>, dim3(gridPoints), dim3(TPB), back, side->inserted[b], a, b);   // grid clears, points, end-of-batch bookkeeping, the previous group's voxel lists
				else SIMLOD_LAUNCH_STOP(k_insert<false>, dim3(gridPoints), dim3(TPB), back, side->inserted[b], a, b);
			} else {
				const bool gated = expand_gate_enter(ctx, stream);
				SIMLOD_LAUNCH(k_expand, dim3(expandWgs), dim3(ETPB), stream, a, b);
				expand_gate_leave(ctx, stream, nullptr, gated);
				if (!single && a.acct != 0u) SIMLOD_LAUNCH(k_rootpre, dim3(1), dim3(1024), back, a, b);
				if (single) SIMLOD_LAUNCH(k_insert<true>, dim3(gridPoints), dim3(TPB), back, a, b);
				else SIMLOD_LAUNCH(k_insert<false>, dim3(gridPoints), dim3(TPB), back, a, b);
			}
			if (profile_dominant()) {      // bench.py's roofline: the dominant kernel timed in the headline configuration, by the launch's own start / stop events
				hipEvent_t e0, e1;
				profile_kernel_events("k_voxelize", &e0, &e1);
				hipExtLaunchKernelGGL(k_voxelize, dim3(voxWgs), dim3(VTPB), 0, back, e0, e1, 0, a, b);
				{ const hipError_t e = hipGetLastError(); if (e != hipSuccess) return fail(e); }
			} else
			SIMLOD_LAUNCH(k_voxelize, dim3(voxWgs), dim3(VTPB), back, a, b);
		}
		if (side != nullptr) {
			hipError_t e = hipEventRecord(side->tailDone, back);
			if (e == hipSuccess) e = hipStreamWaitEvent(stream, side->tailDone, 0);
			if (e != hipSuccess) return fail(e);
		}
		// (kernels that follow each other on one queue start where the one before ends: three small kernels cost what one does — merged into one,
		// with an arrival counter and its fence, this tail took 22 us instead of 19)
		const uint32_t gridNodes = (a.nodeCapacity + TPB - 1) / TPB;
		SIMLOD_LAUNCH(k_voxdone, dim3(min(gridNodes, dev.numCUs)), dim3(TPB), stream, a);   // the last batch's voxel lists (the others: by the following batch's k_insert)
		SIMLOD_LAUNCH(k_stats, dim3(min(gridNodes, dev.numCUs)), dim3(TPB), stream, a);
	}
	SIMLOD_LAUNCH(k_finish, dim3(1), dim3(64), stream, a, fits ? 1u : 0u, plan.feedback, (const uint32_t*)numBatchesUploaded, plan.seq);
	if (profile_enabled()) profile_close(stream);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return (int)e;
	return fits ? 0 : (int)hipErrorInvalidValue;
}

uint64_t construct_min_bytes(uint32_t nodeCapacity) {
	BuildArgs a{};
	a.nodeCapacity = nodeCapacity;
	layout_construct(a, 0, false);
	return a.scratchBytes + 4096 + 26ull * 65536;   // the smallest capacity layout_construct accepts, plus a page of slack
}

}  // namespace build
}  // namespace simlod
