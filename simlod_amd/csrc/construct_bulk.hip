// construct_bulk.hip — incremental octree/LOD builder for MI355X (gfx950): `kernel_construct` in the opt-in COALESCED mode
// (simlod_set_ingest_mode(1)); construct_batch.hip is the chain of the default exact mode.  This chain is exact as well when it is given
// one batch at a time (SIMLOD_EXACT_CHAIN=bulk, used by the parity tests), it is built for many.
//
// Replaces modules/progressive_octree/progressive_octree_voxels.cu:804-1010 (one persistent cooperative CUDA
// kernel with ~40 grid.sync() per batch) behind the same argument list and the same Node/Chunk/OccupancyGrid
// memory image.  Design:
//
//   * per batch a short CHAIN of launches on the caller's stream — ingest, expand, place, link, nodes, end — because a
//     dependent kernel boundary costs ~1.5-1.9 us on this chip while a software grid barrier over 256 CUs / 8 XCDs costs 4-26 us;
//     control flow stays on the device (a control block at byte 0 of the momentary buffer), inactive kernels exit at once, so the
//     call is fully asynchronous like the reference's;
//   * `k_ingest` reads every point ONCE (one coalesced 16-byte load) and, for a sample whose leaf does not overflow — the common
//     case — does everything the reference spreads over three passes and three tree descents: descent, arrival count, slot
//     reservation, the 16-byte store into the leaf's chunk, voxel sampling of the root path, and the stores of the voxels it won;
//   * counters are aggregated per WORKGROUP in LDS hash tables (one global atomic per workgroup and counter): device-scope atomics
//     on one word retire at ~88 M/s on this chip and a spatially compact batch sends most of its points to a few dozen leaves;
//   * chunks are allocated ON DEMAND by whoever reserves the first slot of a chunk (slot % 1000 == 0) and published through a
//     directory (the leaf chunk table for point lists, a hash directory for voxel lists); everybody else looks the chunk up.
//     Allocators never wait, so lookups always terminate.  The O(list length) walks of voxels.cu:500-503 / 606-610 / 688-692 are gone;
//   * samples of an overflowing leaf (and the leaf's stored points) take the slow path: `k_expand` builds, per round, a 512-bin
//     histogram of them three octree levels below the leaf, decides up to three generations of the split cascade from it in one
//     step (one grid barrier per round instead of the reference's ~8 grid.sync() per level), and `k_place` inserts and samples
//     them in their final leaves;
//   * voxel sampling walks the root path BOTTOM-UP (occupancy is hierarchical: a set bit implies the covering bits of all
//     ancestors) and reads the path from a per-node ancestor table instead of chasing parent -> node -> grid pointers;
//   * no capacity limit loses a point: a split reserves its node slots and spill space before anything is modified, or it does not
//     happen yet (the leaf grows and is queued again by a later batch).
//
// The result after every batch is the reference's: same topology, same per-node sample multisets, same occupancy
// bitsets, same voxel positions (bit-exact fp32), same counters in Node and Stats, same allocator offset, same
// chunk-pool accounting.  What stays scheduling dependent is what is scheduling dependent in the reference too
// (SURVEY.md H6): node indices, chunk addresses, sample order inside a node, which point colours a voxel.
//
// Opt-in COALESCED mode (simlod_set_ingest_mode(1)): all pending batches of a launch (<= 20, as many as the momentary buffer
// holds) are ingested as one group.  Topology, multisets, bitsets and voxel positions do not depend on the batch granularity;
// the allocator / chunk-pool counters of Stats do (fewer intermediate chunks are ever allocated), so the default stays exact.
#include "simlod_device.hpp"
#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {
namespace bulk {

static constexpr uint32_t TPB = 256;
static constexpr float F_GRID = 1048576.0f;      // 2^MAX_DEPTH, progressive_octree_voxels.cu:139
static constexpr float F_FULL = 268435456.0f;    // MAX_DEPTH_GRIDSIZE, structures.cuh:26
static constexpr uint32_t MAXPTS = SIMLOD_MAX_POINTS_PER_NODE;
static constexpr uint32_t CHUNK = SIMLOD_POINTS_PER_CHUNK;
static constexpr uint32_t NONE = 0xffffffffu;
static constexpr uint32_t STORED = 0xfffffffeu;

// Leaf chunk table: slot k of leaf i's point list -> chunk, LEAF_SLOTS entries per node.  A leaf that can still split stores
// at most MAX_POINTS_PER_NODE points (= 50 chunks).  It is the DIRECTORY of the on-demand allocation (whoever reserves slot
// k * 1000 of a leaf allocates chunk k and publishes it here; everybody else polls the entry) and it lets a split read a leaf's
// whole list with all lanes at once instead of chasing 50 `next` pointers with one.  Entries behind the end of a list are null.
// Survives between launches like the recycle stack does; refilled by k_parents whenever k_begin finds its stamp stale.
static constexpr uint32_t LEAF_SLOTS = MAXPTS / CHUNK;
static constexpr uint32_t TABLE_MAGIC = 0x51ab1e06u;

// Ancestor paths: PATH_WORDS 64-bit entries per node, entry k = the k-th ancestor (parent first), zero-terminated.
// An entry packs everything sampling needs to know about that ancestor — its occupancy grid (offset into the persistent
// buffer), level and node index — so a sample reads its whole root path with independent loads instead of chasing
// parent -> node -> grid pointers level by level.  Rebuilt for every node at the start of a launch (k_paths), extended for new
// nodes at a split (k_expand).
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

__device__ __forceinline__ Ctl* ctl_of(const BuildArgs& a) { return reinterpret_cast<Ctl*>(a.mom); }
template <class T> __device__ __forceinline__ T* at(const BuildArgs& a, uint64_t off) { return reinterpret_cast<T*>(a.mom + off); }

__device__ __forceinline__ void raise(Ctl* ctl, uint32_t bit) { atomicOr(&ctl->errors, bit); }
// conditions after which the octree image cannot be trusted: everything stops, pollers bail out, Stats.dbg keeps the bit until a reset
__device__ __forceinline__ void panic(Ctl* ctl, uint32_t bit) {
	atomicOr(&ctl->errors, bit);
	__hip_atomic_store(&ctl->panic, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	ctl->abortBatch = 1; ctl->stop = 1;
}
__device__ __forceinline__ bool panicked(Ctl* ctl) { return __hip_atomic_load(&ctl->panic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }

__device__ __forceinline__ SimlodChunk*& tail_of(SimlodChunk* head) { return *reinterpret_cast<SimlodChunk**>(&head->size); }

// ---- the group of batches that is being ingested ----------------------------------------------------------------------------
// A group is ONE batch in exact mode and all pending batches of the launch in coalesced mode.  Its samples live in the ring slots
// batchSlot[0..groupBatches); the virtual index of sample i of batch b is b * MAX_BATCH_SIZE + i.
__device__ __forceinline__ const float4* ring_slot(const BuildArgs& a, uint32_t slot) {
	return reinterpret_cast<const float4*>(a.ring + (size_t)slot * SIMLOD_MAX_BATCH_SIZE);
}
__device__ __forceinline__ float4 point_of(const BuildArgs& a, const Ctl* ctl, uint32_t v) {
	const uint32_t b = v / SIMLOD_MAX_BATCH_SIZE;
	return ring_slot(a, ctl->batchSlot[b])[v - b * SIMLOD_MAX_BATCH_SIZE];
}
// tile #t of `tileSize` samples -> (batch, first sample inside the batch); false behind the last tile
__device__ __forceinline__ bool tile_lookup(const Ctl* ctl, uint32_t tileSize, uint32_t t, uint32_t& b, uint32_t& first) {
	const uint32_t nb = ctl->groupBatches;
	for (b = 0; b < nb; b++) {
		const uint32_t nt = (ctl->batchSize[b] + tileSize - 1) / tileSize;
		if (t < nt) { first = t * tileSize; return true; }
		t -= nt;
	}
	return false;
}

__device__ __forceinline__ uint32_t spilled_end(const Ctl* ctl) { return (uint32_t)__hip_atomic_load(&ctl->reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Make group #ordinal of this launch current, or deactivate (progressive_octree_voxels.cu:890-912).
__device__ void prepare_group(const BuildArgs& a, Ctl* ctl, uint32_t ordinal) {
	ctl->active = 0;
	if (ctl->consumed >= ctl->numBatches || ctl->stop) return;
	const SimlodAllocatorGlobal* alloc = reinterpret_cast<const SimlodAllocatorGlobal*>(a.pers);
	const bool full = alloc->offset + SIMLOD_MEM_SAFETY_MARGIN >= a.persCapacity;
	a.stats->memCapacityReached = full ? 1 : 0;
	if (full) { ctl->stop = 1; return; }
	const uint32_t batchIndex = a.stats->batchletIndex;
	const uint32_t g = ctl->coalesce ? min(ctl->numBatches - ctl->consumed, a.groupMax) : 1u;
	uint32_t total = 0;
	for (uint32_t b = 0; b < g; b++) {
		const uint32_t slot = (batchIndex + b) % SIMLOD_BATCH_STREAM_SIZE;
		uint32_t size = a.batchSizes[slot];
		if (size > SIMLOD_MAX_BATCH_SIZE) size = SIMLOD_MAX_BATCH_SIZE;
		ctl->batchSize[b] = size; ctl->batchSlot[b] = slot;
		total += size;
	}
	ctl->batchIndex = batchIndex;
	ctl->groupBatches = g;
	ctl->groupPoints = total;
	ctl->ordinal = ordinal;
	ctl->numPending = 0;
	ctl->numSpilling = 0;
	ctl->roundSpill[0] = 0; ctl->roundSpill[1] = 0;
	ctl->dirUsed = 0;
	ctl->numVoxLeaves = 0;
	ctl->reserve = (unsigned long long)a.stats->numNodes << 32;
	ctl->nodesAtStart = a.stats->numNodes;
	ctl->treeModified = 0;
	ctl->abortBatch = 0;
	ctl->barrierCount[0] = 0; ctl->barrierCount[1] = 0;      // every k_expand launch counts its barrier generations from zero
	ctl->active = 1;
}

// ---- begin: snapshot the upload counter, stamp the frame start (voxels.cu:823-825, 870-885) -------------------
__global__ void k_begin(BuildArgs a, uint32_t momentaryTooSmall, uint32_t coalesce, uint32_t batchLimit, uint32_t debugFlags) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	Ctl* ctl = ctl_of(a);
	const uint32_t fatal = a.stats->dbg & (SIMLOD_ERR_BARRIER_TIMEOUT | SIMLOD_ERR_DIRECTORY_FULL);   // sticky until the host resets the octree
	ctl->errors = momentaryTooSmall ? SIMLOD_ERR_MOMENTARY_TOO_SMALL : 0u;
	ctl->stop = (momentaryTooSmall || fatal) ? 1u : 0u;
	ctl->panic = 0;
	ctl->abortBatch = 0;
	ctl->coalesce = coalesce;
	ctl->debugFlags = debugFlags;
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
	ctl->consumed = 0;
	for (int i = 0; i < 8; i++) ctl->statCounters[i] = 0;
	const bool valid = ctl->tableMagic == TABLE_MAGIC && ctl->tableBatch == first && ctl->tableNodes == (uint64_t)a.nodes && ctl->tablePers == (uint64_t)a.pers;
	ctl->rebuildLeafChunks = valid ? 0u : 1u;
	ctl->tableMagic = 0;                        // valid again once k_finish has run
	prepare_group(a, ctl, 0);
}

// ---- parents: node index -> parent index, rebuilt at the start of every launch from the children pointers; snapshots of the
// list lengths; the leaf chunk table and the tail pointers when the table's stamp is stale -----------------------------------------
__global__ __launch_bounds__(TPB) void k_parents(BuildArgs a) {
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= numNodes) return;
	uint32_t* parentOf = at<uint32_t>(a, a.offParent);
	if (i == 0) parentOf[0] = NONE;
	SimlodNode* n = a.nodes + i;
#pragma unroll
	for (int k = 0; k < 8; k++) {
		const SimlodNode* c = n->children[k];
		if (c != nullptr) parentOf[(uint32_t)(c - a.nodes)] = i;
	}
	at<uint32_t>(a, a.offPtStart)[i] = n->numPoints;
	at<uint32_t>(a, a.offVoxStart)[i] = n->numVoxelsStored;
	if (ctl_of(a)->rebuildLeafChunks) {
		SimlodChunk** slots = at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)i * LEAF_SLOTS;
		const bool leaf = node_is_leaf(n);
		SimlodChunk* c = leaf ? n->points : nullptr;
		SimlodChunk* last = nullptr;
		for (uint32_t k = 0; k < LEAF_SLOTS; k++) {
			slots[k] = c;
			if (c != nullptr) { last = c; c = c->next; }
		}
		if (!leaf) {                           // an inner node's row lists its voxel chunks (for the rasteriser, render.hip r_visible)
			SimlodChunk* v = n->voxelChunks;
			for (uint32_t k = 0; k < LEAF_SLOTS && v != nullptr; k++) { slots[k] = v; v = v->next; }
		}
		// the tail pointers (8 spare bytes of a head chunk) are this implementation's own: an image built elsewhere has none
		while (c != nullptr) { last = c; c = c->next; }
		if (leaf && n->points != nullptr) tail_of(n->points) = last;
		if (n->voxelChunks != nullptr) {
			SimlodChunk* t = n->voxelChunks;
			while (t->next != nullptr) t = t->next;
			tail_of(n->voxelChunks) = t;
		}
	}
}

// ---- paths: every node's ancestor list, from the parent table (one thread per node, depth <= 20 steps) ---------------------
__global__ __launch_bounds__(TPB) void k_paths(BuildArgs a) {
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= numNodes) return;
	const uint32_t* parentOf = at<const uint32_t>(a, a.offParent);
	unsigned long long* rec = at<unsigned long long>(a, a.offPaths) + (uint64_t)i * PATH_WORDS;
	uint32_t k = 0;
	for (uint32_t cur = parentOf[i]; cur != NONE && k < PATH_WORDS - 1; cur = parentOf[cur]) {
		const SimlodNode* n = a.nodes + cur;
		rec[k++] = path_pack(a.pers, cur, n->level, n->grid);
	}
	rec[k] = 0;
}

// ---- on-demand chunk allocation -------------------------------------------------------------------------------------------------
// Point chunks come from the recycle stack first (voxels.cu:505-516).  Free chunks are chunkQueue[numAllocatedChunks ..
// chunkPoolSize); chunkPoolSize is constant while a kernel pops (k_ingest, k_place) and is raised to the high-water mark of
// numAllocatedChunks before anything is pushed back (k_expand) and at the end of the batch (k_end) — the same totals as the
// reference's "recycle, then allocate, then max" (:346-357, :535-537), because the chunks in use never exceed their end-of-batch number.
__device__ __forceinline__ SimlodChunk* take_point_chunk(const BuildArgs& a) {
	const unsigned long long idx = atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats->numAllocatedChunks), 1ull);
	SimlodChunk* c = idx < a.stats->chunkPoolSize ? at<SimlodChunk*>(a, a.offQueue)[idx]
	                                              : reinterpret_cast<SimlodChunk*>(persistent_alloc(a.pers, sizeof(SimlodChunk), 1));
	c->next = nullptr;
	return c;
}

// Hash directory of the chunks allocated in the current group: (kind, node, chunk index) -> chunk.  Cleared by the host-enqueued
// memset of every launch; entries of earlier groups of the launch carry another tag and count as free.
struct DirEntry {
	unsigned long long key;
	SimlodChunk* ptr;
};
static constexpr unsigned long long DIR_BUSY = 1ull << 62;
enum : uint32_t { KIND_PT = 0, KIND_VOX = 1 };
__device__ __forceinline__ unsigned long long dir_key(uint32_t tag, uint32_t kind, uint32_t node, uint32_t k) {
	return (1ull << 63) | ((unsigned long long)(tag & 0xfffffu) << 42) | ((unsigned long long)kind << 41) | ((unsigned long long)node << 22) | (k & 0x3fffffu);
}
__device__ __forceinline__ uint32_t dir_tag(unsigned long long key) { return (uint32_t)(key >> 42) & 0xfffffu; }
__device__ __forceinline__ uint32_t dir_hash(const BuildArgs& a, unsigned long long key) {
	key ^= key >> 29; key *= 0x9e3779b97f4a7c15ull; key ^= key >> 32;
	return (uint32_t)key & (a.dirCap - 1u);
}
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void dir_insert(const BuildArgs& a, Ctl* ctl, uint32_t kind, uint32_t node, uint32_t k, SimlodChunk* c) {
	DirEntry* dir = at<DirEntry>(a, a.offDir);
	const uint32_t tag = ctl->ordinal + 1u;
	const unsigned long long key = dir_key(tag, kind, node, k);
	atomicAdd(&ctl->dirUsed, 1u);
	uint32_t h = dir_hash(a, key);
	for (uint32_t probe = 0; probe < a.dirCap; probe++, h = (h + 1u) & (a.dirCap - 1u)) {
		unsigned long long cur = ld_agent(&dir[h].key);
		while (cur == 0ull || ((cur >> 63) != 0ull && dir_tag(cur) != tag)) {           // free, or left over from an earlier group
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
__device__ SimlodChunk* dir_find(const BuildArgs& a, const Ctl* ctl, uint32_t kind, uint32_t node, uint32_t k) {
	const DirEntry* dir = at<const DirEntry>(a, a.offDir);
	const uint32_t tag = ctl->ordinal + 1u;
	const unsigned long long key = dir_key(tag, kind, node, k);
	uint32_t h = dir_hash(a, key);
	for (uint32_t probe = 0; probe < a.dirCap; probe++, h = (h + 1u) & (a.dirCap - 1u)) {
		const unsigned long long cur = ld_agent(&dir[h].key);
		if (cur == key) return reinterpret_cast<SimlodChunk*>(ld_agent(reinterpret_cast<const unsigned long long*>(&dir[h].ptr)));
		if (cur == 0ull) return nullptr;
		if ((cur >> 63) != 0ull && dir_tag(cur) != tag) return nullptr;
	}
	return nullptr;
}

// blocking lookup: the allocator of the chunk has already reserved its slot range (its atomicAdd precedes the caller's) and it
// never waits for anything, so this terminates; the bound is a guard against a broken device
__device__ SimlodChunk* dir_wait(const BuildArgs& a, Ctl* ctl, uint32_t kind, uint32_t node, uint32_t k) {
	for (uint32_t spin = 0;; spin++) {
		SimlodChunk* c = dir_find(a, ctl, kind, node, k);
		if (c != nullptr) return c;
		__builtin_amdgcn_s_sleep(2);
		if ((spin & 255u) == 255u && panicked(ctl)) return nullptr;
		if (spin > (1u << 22)) { panic(ctl, SIMLOD_ERR_DIRECTORY_FULL); return nullptr; }
	}
}

// chunk k of a node's point list: allocate (the caller reserved slot k * 1000) ...
__device__ void make_point_chunk(const BuildArgs& a, Ctl* ctl, uint32_t node, uint32_t k) {
	if (k < LEAF_SLOTS && at<SimlodChunk*>(a, a.offLeafChunks)[(uint64_t)node * LEAF_SLOTS + k] != nullptr) return;   // k_prealloc was here
	SimlodChunk* c = take_point_chunk(a);
	if (k == 0u) { a.nodes[node].points = c; tail_of(c) = c; }
	if (k < LEAF_SLOTS) {
		unsigned long long* slot = reinterpret_cast<unsigned long long*>(at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)node * LEAF_SLOTS + k);
		__hip_atomic_store(slot, (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else dir_insert(a, ctl, KIND_PT, node, k, c);
}
// ... or look up (the caller holds some other slot of the chunk)
__device__ SimlodChunk* wait_point_chunk(const BuildArgs& a, Ctl* ctl, uint32_t node, uint32_t k) {
	if (k < LEAF_SLOTS) {
		const unsigned long long* slot = reinterpret_cast<const unsigned long long*>(at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)node * LEAF_SLOTS + k);
		for (uint32_t spin = 0;; spin++) {
			const unsigned long long c = ld_agent(slot);
			if (c != 0ull) return reinterpret_cast<SimlodChunk*>(c);
			__builtin_amdgcn_s_sleep(2);
			if ((spin & 255u) == 255u && panicked(ctl)) return nullptr;
			if (spin > (1u << 22)) { panic(ctl, SIMLOD_ERR_DIRECTORY_FULL); return nullptr; }
		}
	}
	// an over-full leaf (its split was deferred, or it sits at MAX_DEPTH): the only older chunk anyone can still write to is the tail
	if (k * CHUNK < at<const uint32_t>(a, a.offPtStart)[node]) return tail_of(a.nodes[node].points);
	return dir_wait(a, ctl, KIND_PT, node, k);
}
__device__ void make_voxel_chunk(const BuildArgs& a, Ctl* ctl, uint32_t node, uint32_t k) {
	SimlodChunk* c = reinterpret_cast<SimlodChunk*>(persistent_alloc(a.pers, sizeof(SimlodChunk), 1));   // voxel chunks never come from the pool (voxels.cu:656-659)
	c->next = nullptr;
	if (k == 0u) { a.nodes[node].voxelChunks = c; tail_of(c) = c; }
	// an inner node's row of the leaf chunk table lists its voxel chunks: the rasteriser reads the list from there (render.hip r_visible)
	if (k < LEAF_SLOTS && !node_is_leaf(a.nodes + node)) at<SimlodChunk*>(a, a.offLeafChunks)[(uint64_t)node * LEAF_SLOTS + k] = c;
	dir_insert(a, ctl, KIND_VOX, node, k, c);
}
__device__ SimlodChunk* wait_voxel_chunk(const BuildArgs& a, Ctl* ctl, uint32_t node, uint32_t k) {
	if (k * CHUNK < at<const uint32_t>(a, a.offVoxStart)[node]) return tail_of(a.nodes[node].voxelChunks);   // the partially filled tail of earlier batches
	return dir_wait(a, ctl, KIND_VOX, node, k);
}

// ---- split bookkeeping ---------------------------------------------------------------------------------------------------------------
// Reserve `nodes` node slots and `spill` points of spill space TOGETHER (one 64-bit word), before anything is modified: a leaf that
// cannot be served now stays a leaf — too full, but intact — and is queued again by a later batch.
__device__ bool reserve(const BuildArgs& a, Ctl* ctl, uint32_t nodes, uint32_t spill, uint32_t& nodeBase, uint32_t& spillBase) {
	unsigned long long cur = __hip_atomic_load(&ctl->reserve, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	for (;;) {
		nodeBase = (uint32_t)(cur >> 32); spillBase = (uint32_t)cur;
		if (nodeBase + nodes > a.nodeCapacity) { raise(ctl, SIMLOD_ERR_NODES_EXHAUSTED); return false; }
		if ((unsigned long long)spillBase + spill > a.spilledCap) { raise(ctl, SIMLOD_ERR_SPILLED_OVERFLOW); return false; }
		const unsigned long long prev = atomicCAS(&ctl->reserve, cur, cur + ((unsigned long long)nodes << 32) + spill);
		if (prev == cur) { atomicAdd(&a.stats->numNodes, nodes); return true; }   // voxels.cu:317
		cur = prev;
	}
}

__device__ __forceinline__ uint32_t round_tag(const Ctl* ctl, uint32_t round) { return ctl->ordinal * 64u + round + 1u; }

// Queue `leafIdx` for splitting in round `round` of the current group (round 0: by k_ingest, with room for `stored` points to move;
// later rounds: by k_expand for freshly created, empty nodes).  Everything the split needs is reserved here, by ONE thread per leaf:
// a place in the round's work list (and with it a histogram), eight node slots, the spill space, the occupancy grid.
__device__ void queue_split(const BuildArgs& a, Ctl* ctl, uint32_t leafIdx, uint32_t stored, uint32_t round, SpillEntry* list, uint32_t* count) {
	const uint32_t s = atomicAdd(count, 1u);
	if (s >= a.histCap) { raise(ctl, SIMLOD_ERR_SPILLING_OVERFLOW); return; }            // more leaves cross the limit at once than a round can hold: deferred
	uint32_t nodeBase, spillBase;
	if (!reserve(a, ctl, 8u, stored, nodeBase, spillBase)) { list[s] = SpillEntry{NONE, 0u, 0u, 0u}; return; }
	SimlodNode* leaf = a.nodes + leafIdx;
	if (leaf->grid == nullptr)                              // voxels.cu:363-365
		leaf->grid = reinterpret_cast<SimlodOccupancyGrid*>(persistent_alloc(a.pers, sizeof(SimlodOccupancyGrid), 1));
	list[s] = SpillEntry{leafIdx, nodeBase, spillBase, stored};
	at<unsigned long long>(a, a.offSplitTag)[leafIdx] = ((unsigned long long)round_tag(ctl, round) << 32) | s;
}

// One arrival-counter update for `cnt` samples (voxels.cu:203-218); returns the counter's previous value.  A leaf is queued for
// splitting by the arrival that crosses the limit — everything that arrived before it was stored directly, so `old` bounds the
// stored points that will have to move — or, if the leaf is already over the limit because an earlier batch could not split it
// (spill space, node array or work list exhausted: the split was deferred, nothing was lost), by whoever touches it first in this
// group.  A node at MAX_DEPTH cannot be subdivided (the descent stops there): it keeps growing instead.
// `withPoints`: counter and numPoints sit side by side in Node (byte 64 / 68): ONE 64-bit atomic advances both — the caller takes the
// numPoints part back if the range turns out not to fit (device-scope atomics on one word retire at ~88 M/s: every hot leaf sees one
// atomic per workgroup instead of two).
__device__ __forceinline__ uint32_t count_into(const BuildArgs& a, Ctl* ctl, uint32_t leafIdx, uint32_t cnt, bool withPoints) {
	SimlodNode* leaf = a.nodes + leafIdx;
	uint32_t old;
	if (withPoints) old = (uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(&leaf->counter), (unsigned long long)cnt | ((unsigned long long)cnt << 32));
	else old = atomicAdd(&leaf->counter, cnt);
	if (old + cnt > MAXPTS && leaf->level < SIMLOD_MAX_DEPTH) {
		SpillEntry* list = at<SpillEntry>(a, a.offSpillA);
		if (old <= MAXPTS) queue_split(a, ctl, leafIdx, old, 0u, list, &ctl->numSpilling);
		else {
			const uint32_t start = at<const uint32_t>(a, a.offPtStart)[leafIdx];
			const uint32_t tag = ctl->ordinal + 1u;
			if (start > MAXPTS && atomicExch(at<uint32_t>(a, a.offRetryTag) + leafIdx, tag) != tag) queue_split(a, ctl, leafIdx, start, 0u, list, &ctl->numSpilling);
		}
	}
	return old;
}

// k_peek counted every PEEK-th sample of the group per leaf.  A leaf that is on course to overflow in this group takes no direct
// inserts: its samples would be stored only to be moved again by the split (a prediction — wrong either way it only shifts work
// between k_ingest and k_place).  The root as a leaf never does: whether its grid survives the batch is not known yet.
static constexpr uint32_t PEEK = 8;
__device__ __forceinline__ bool leaf_is_hot(const BuildArgs& a, uint32_t leafIdx) {
	if (leafIdx == 0u) return true;
	const uint32_t start = at<const uint32_t>(a, a.offPtStart)[leafIdx], est = at<const uint32_t>(a, a.offEst)[leafIdx];
	return start + PEEK * est > MAXPTS - MAXPTS / 16u;
}

// k_place stores samples in `leafIdx`: the first to do so in this group puts the leaf on k_voxelize's work list, one entry
// (leaf | piece << 19) per VOX_PIECE samples
static constexpr uint32_t VOX_SPT = 8, VOX_PIECE = 1024u * VOX_SPT;
__device__ __forceinline__ void note_placed(const BuildArgs& a, Ctl* ctl, uint32_t leafIdx) {
	const uint32_t tag = ctl->ordinal + 1u;
	if (atomicExch(at<uint32_t>(a, a.offPlacedTag) + leafIdx, tag) != tag) {
		// pieces of VOX_PIECE samples: Node.counter is final since k_expand — that many samples the leaf holds when k_place is done
		const uint32_t start = at<const uint32_t>(a, a.offPtStart)[leafIdx], end = a.nodes[leafIdx].counter;
		const uint32_t pieces = end > start ? (end - start + VOX_PIECE - 1u) / VOX_PIECE : 1u;
		const uint32_t at0 = atomicAdd(&ctl->numVoxLeaves, pieces);
		uint32_t* list = at<uint32_t>(a, a.offVoxList);
		for (uint32_t q = 0; q < pieces && at0 + q < a.voxListCap; q++) list[at0 + q] = leafIdx | q << 19;
		if (at0 + pieces > a.voxListCap) panic(ctl, SIMLOD_ERR_DIRECTORY_FULL);
	}
}

// ---- one tile of samples: slots, stores, voxel sampling -----------------------------------------------------------------------------
static constexpr int LT_BITS = 9, VT_BITS = 8, SET_BITS = 11;
static constexpr uint32_t LT_CAP = 1u << LT_BITS, VT_CAP = 1u << VT_BITS, SET_CAP = 1u << SET_BITS;

struct TileShared {
	Tab<LT_BITS> lt;                       // leaf -> samples of this workgroup (count; the old value of the count is a sample's rank)
	uint32_t ltBase[LT_CAP];               // first slot of the range this workgroup reserved in the leaf, or NONE (samples wait for k_place)
	SimlodChunk* ltPtr[LT_CAP][2];         // the chunk that holds slot ltBase and the one behind it
	Tab<VT_BITS> vt;                       // inner node -> voxels created by this workgroup (count, later the store cursor)
	uint32_t vtBase[VT_CAP];
	SimlodChunk* vtPtr[VT_CAP][2];
	uint32_t claimed[SET_CAP];             // (node, cell) pairs this workgroup already claimed
	uint32_t pendCount, pendBase;
};

__device__ __forceinline__ void tile_reset(TileShared& sh) {
	tab_init(sh.lt);
	tab_init(sh.vt);
	for (uint32_t i = threadIdx.x; i < SET_CAP; i += blockDim.x) sh.claimed[i] = TBL_EMPTY;
	if (threadIdx.x == 0) { sh.pendCount = 0; sh.pendBase = 0; }
}

// Reserve the slot ranges of this workgroup's samples, one global atomic per (workgroup, leaf), allocate the chunks whose first slot
// falls into a range, then look up the others.  ALL allocations of a thread come before its first lookup: allocators never wait.
// INGEST: ranges come from the arrival counter and are only taken while the leaf stays within its limit (and is not a root that
// is still a leaf: whether its grid survives the batch is not known yet) — otherwise the samples are left to k_place.
template <bool INGEST>
__device__ void flush_points(const BuildArgs& a, Ctl* ctl, TileShared& sh) {
	for (uint32_t e = threadIdx.x; e < LT_CAP; e += blockDim.x) {
		const uint32_t key = sh.lt.keys[e];
		if (key == TBL_EMPTY) continue;
		const uint32_t cnt = sh.lt.vals[e];
		uint32_t old;
		bool direct = true;
		if (INGEST) {
			const bool hot = leaf_is_hot(a, key);
			old = count_into(a, ctl, key, cnt, !hot);                                  // arrivals and, speculatively, stored points (voxels.cu:203-218, :593)
			direct = !hot && old + cnt <= MAXPTS;
			if (!hot && !direct) atomicSub(&a.nodes[key].numPoints, cnt);
		} else { old = atomicAdd(&a.nodes[key].numPoints, cnt); note_placed(a, ctl, key); }
		if (direct) for (uint32_t k = (old + CHUNK - 1) / CHUNK; k * CHUNK < old + cnt; k++) make_point_chunk(a, ctl, key, k);
		sh.ltBase[e] = direct ? old : NONE;
	}
	for (uint32_t e = threadIdx.x; e < LT_CAP; e += blockDim.x) {
		const uint32_t key = sh.lt.keys[e];
		if (key == TBL_EMPTY || sh.ltBase[e] == NONE) continue;
		const uint32_t old = sh.ltBase[e], cnt = sh.lt.vals[e], k0 = old / CHUNK;
		sh.ltPtr[e][0] = wait_point_chunk(a, ctl, key, k0);
		sh.ltPtr[e][1] = (old + cnt - 1) / CHUNK > k0 ? wait_point_chunk(a, ctl, key, k0 + 1) : nullptr;
	}
}

__device__ __forceinline__ void store_point(const BuildArgs& a, Ctl* ctl, TileShared& sh, uint32_t e, uint32_t rank, const float4& p) {
	const uint32_t base = sh.ltBase[e], slot = base + rank, k = slot / CHUNK, d = k - base / CHUNK;
	SimlodChunk* c = d == 0u ? sh.ltPtr[e][0] : d == 1u ? sh.ltPtr[e][1] : wait_point_chunk(a, ctl, sh.lt.keys[e], k);
	if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % CHUNK] = p;
}

// a sample whose leaf found no room in the LDS table (k_place only): its own slot, its own chunk bookkeeping
__device__ void store_point_direct(const BuildArgs& a, Ctl* ctl, uint32_t leafIdx, const float4& p) {
	const uint32_t slot = atomicAdd(&a.nodes[leafIdx].numPoints, 1u), k = slot / CHUNK;
	note_placed(a, ctl, leafIdx);
	if (slot % CHUNK == 0u) make_point_chunk(a, ctl, leafIdx, k);
	SimlodChunk* c = wait_point_chunk(a, ctl, leafIdx, k);
	if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % CHUNK] = p;
}

// cell-centre position of a voxel, voxels.cu:103-114, operation by operation (no contraction)
__device__ __forceinline__ float4 voxel_of(const BuildArgs& a, int level, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits) {
	const uint32_t sh = (uint32_t)(SIMLOD_MAX_DEPTH + 1 - level);
	const uint32_t cx = (pX >> sh) & 127u, cy = (pY >> sh) & 127u, cz = (pZ >> sh) & 127u;
	// Node.X/Y/Z of the level-`level` node that contains the sample: the top `level` bits of its 28-bit coordinate (the
	// 2^20 grid the nodes are indexed in is the same fp32 quotient scaled by an exact power of two, simlod_device.hpp quantize),
	// masked to `level` bits: a coordinate exactly on the max face quantises to 2^20 (2^28 here) and the reference's descent, which
	// looks at bits 19..0 only, files it under node coordinate 0 on that axis (voxels.cu:171-179) — the voxel sits at the LOW face
	const uint32_t nsh = 28u - (uint32_t)level;
	const uint32_t nmask = (1u << (uint32_t)level) - 1u;
	const uint32_t nX = (pX >> nsh) & nmask, nY = (pY >> nsh) & nmask, nZ = (pZ >> nsh) & nmask;
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

// a voxel whose node found no room in the LDS table: its own slot, its own chunk bookkeeping
__device__ void store_voxel_direct(const BuildArgs& a, Ctl* ctl, uint32_t nodeIdx, int level, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits) {
	const uint32_t slot = atomicAdd(&a.nodes[nodeIdx].numVoxels, 1u), k = slot / CHUNK;          // voxels.cu:101
	if (slot % CHUNK == 0u) make_voxel_chunk(a, ctl, nodeIdx, k);
	SimlodChunk* c = wait_voxel_chunk(a, ctl, nodeIdx, k);
	if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % CHUNK] = voxel_of(a, level, pX, pY, pZ, colorBits);
}

// true: the caller is the first of its workgroup to claim `key` (or the set has no room: claim anyway, merely redundant)
__device__ __forceinline__ bool set_insert(uint32_t* set, uint32_t key) {
	uint32_t h = (key * 2654435761u) >> (32 - SET_BITS);
#pragma unroll 1
	for (int probe = 0; probe < 8; ++probe) {
		uint32_t k = set[h];
		if (k == TBL_EMPTY) k = atomicCAS(&set[h], TBL_EMPTY, key);
		if (k == TBL_EMPTY) return true;
		if (k == key) return false;
		h = (h + 1) & (SET_CAP - 1);
	}
	return true;
}

// the k-th ancestor entry of a leaf; a root that is still a leaf samples itself (voxels.cu:449-463: every node of the path that
// has a grid is sampled, and the root has one from the reset on)
__device__ __forceinline__ unsigned long long path_entry(const BuildArgs& a, const unsigned long long* rec, uint32_t leafIdx, uint32_t k) {
	if (leafIdx == 0u) { SimlodOccupancyGrid* g = a.nodes[0].grid; return (k == 0u && g != nullptr) ? path_pack(a.pers, 0u, 0u, g) : 0ull; }
	return k < PATH_WORDS - 1 ? rec[k] : 0ull;
}

// 128^3 occupancy test-and-set on the inner nodes of the root-to-leaf path (voxels.cu:50-121, 417-483), BOTTOM-UP.
// The reference tests the sample's cell in EVERY node of the path, root first.  Occupancy is hierarchical, though: a cell of a node
// covers exactly 2x2x2 cells of the child below it, and every sample that ever set a bit in a node had, in the same pass, been
// offered to all its ancestors — so "bit set in node N" implies "covering bit set in every ancestor of N".  The walk therefore
// starts at the deepest inner node and stops at the first level whose bit is already set, or where its own atomicOr lost the race
// (the winner keeps climbing).  Same bitsets, same voxel counts, one winner per cell as in the reference; what it removes is the
// contention: a top-down pass issues 2-4 atomicOr per sample, thousands of them on the same still-clear upper-level words of newly
// entered territory; bottom-up issues about one per NEW voxel, and steady-state samples cost one 4-byte probe.
// Only the FIRST sample of a workgroup that sees a clear cell issues the global atomicOr (per-workgroup claim set in LDS): the
// others know the cell is being taken care of and stop, exactly as if they had lost the race.
// Returns the mask of levels this sample won; the voxels are stored later (store_voxels) behind ranges reserved per workgroup.
// `startLevel`: stored points that move because their leaf splits were offered to the levels above it when they first arrived.
static constexpr int WIN = 3;                         // ancestors fetched and probed together
struct Probe {
	unsigned long long ent[WIN];                      // path entries, 0 = nothing (more) to do
	uint32_t seen[WIN];                               // their occupancy words as read
};

__device__ __forceinline__ uint32_t cell_of(uint32_t level, uint32_t pX, uint32_t pY, uint32_t pZ) {
	const uint32_t shf = (uint32_t)(SIMLOD_MAX_DEPTH + 1) - level;           // voxels.cu:78-85
	const uint32_t cx = (pX >> shf) & 127u, cy = (pY >> shf) & 127u, cz = (pZ >> shf) & 127u;
	return cx + cy * SIMLOD_GRID_SIZE + cz * SIMLOD_GRID_SIZE * SIMLOD_GRID_SIZE;
}

// ancestors k0 .. k0+WIN-1 of the leaf and the sample's occupancy word in each: independent loads, no side effects — a thread issues
// them for ALL its samples before it starts claiming for the first (the claims are dependent chains of LDS and global atomics)
__device__ __forceinline__ void probe_load(const BuildArgs& a, uint32_t leafIdx, uint32_t startLevel, uint32_t pX, uint32_t pY, uint32_t pZ, uint32_t k0, Probe& pr) {
	const unsigned long long* rec = at<const unsigned long long>(a, a.offPaths) + (uint64_t)leafIdx * PATH_WORDS;
#pragma unroll
	for (int w = 0; w < WIN; w++) pr.ent[w] = path_entry(a, rec, leafIdx, k0 + w);
#pragma unroll
	for (int w = 1; w < WIN; w++) if (pr.ent[w - 1] == 0ull) pr.ent[w] = 0ull;                   // what lies behind the terminator was never written
#pragma unroll
	for (int w = 0; w < WIN; w++) {
		const uint32_t level = path_level(pr.ent[w]);
		// voxels.cu:449: the traverse loop samples levels 0..19 only
		if (pr.ent[w] == 0ull || level < startLevel || level >= (uint32_t)SIMLOD_MAX_DEPTH) { pr.ent[w] = 0ull; pr.seen[w] = 0u; }
		else pr.seen[w] = path_grid(a.pers, pr.ent[w])->values[cell_of(level, pX, pY, pZ) >> 5];   // a plain load on purpose: measured, device-scope probes of the hot occupancy lines cost 20 % more
	}
}

// bottom-up over one window: claim while winning; false = the walk ends here
__device__ __forceinline__ bool probe_claim(const BuildArgs& a, Ctl* ctl, TileShared& sh, const Probe& pr, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits, uint32_t& wins) {
#pragma unroll
	for (int w = 0; w < WIN; w++) {
		if (pr.ent[w] == 0ull) return false;
		const uint32_t level = path_level(pr.ent[w]), cell = cell_of(level, pX, pY, pZ), bit = cell & 31u;
		if (((pr.seen[w] >> bit) & 1u) != 0u) return false;                      // voxels.cu:93-94; the ancestors are set as well
		const uint32_t nodeIdx = path_node(pr.ent[w]);
		uint32_t rank;
		const int e = tab_add(sh.vt, nodeIdx, 0u, &rank);
		if (e >= 0 && !set_insert(sh.claimed, ((uint32_t)e << 21) | cell)) return false;          // a sample of this workgroup already claims the cell
		uint32_t* word = &path_grid(a.pers, pr.ent[w])->values[cell >> 5];
		if (((atomicOr(word, 1u << bit) >> bit) & 1u) != 0u) return false;                         // voxels.cu:96; lost: the winner climbs on
		if (e >= 0) { wins |= 1u << level; atomicAdd(&sh.vt.vals[e], 1u); }                        // first point in the cell, voxels.cu:99
		else store_voxel_direct(a, ctl, nodeIdx, (int)level, pX, pY, pZ, colorBits);
	}
	return true;
}

__device__ __forceinline__ uint32_t sample_path(const BuildArgs& a, Ctl* ctl, TileShared& sh, uint32_t leafIdx, uint32_t startLevel, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits, const Probe& first) {
	uint32_t wins = 0;
	bool go = probe_claim(a, ctl, sh, first, pX, pY, pZ, colorBits, wins);
#pragma unroll 1
	for (uint32_t k0 = WIN; go && k0 < PATH_WORDS - 1; k0 += WIN) {
		Probe pr;
		probe_load(a, leafIdx, startLevel, pX, pY, pZ, k0, pr);
		go = probe_claim(a, ctl, sh, pr, pX, pY, pZ, colorBits, wins);
	}
	return wins;
}

// voxel slot ranges of this workgroup, one global atomic per (workgroup, node): numVoxels is counter and cursor in one
// (voxels.cu:101 and :685; numVoxelsStored catches up in k_nodes)
__device__ void flush_voxels(const BuildArgs& a, Ctl* ctl, TileShared& sh, unsigned long long* tally = nullptr) {
	for (uint32_t e = threadIdx.x; e < VT_CAP; e += blockDim.x) {
		const uint32_t key = sh.vt.keys[e];
		if (key == TBL_EMPTY) continue;
		const uint32_t cnt = sh.vt.vals[e];
		if (cnt == 0u) { sh.vtBase[e] = NONE; continue; }
		const uint32_t old = atomicAdd(&a.nodes[key].numVoxels, cnt);
		for (uint32_t k = (old + CHUNK - 1) / CHUNK; k * CHUNK < old + cnt; k++) make_voxel_chunk(a, ctl, key, k);
		sh.vtBase[e] = old;
		if (tally != nullptr) atomicAdd(tally, (unsigned long long)cnt);
	}
	for (uint32_t e = threadIdx.x; e < VT_CAP; e += blockDim.x) {
		const uint32_t key = sh.vt.keys[e];
		if (key == TBL_EMPTY || sh.vtBase[e] == NONE) continue;
		const uint32_t old = sh.vtBase[e], cnt = sh.vt.vals[e], k0 = old / CHUNK;
		sh.vtPtr[e][0] = wait_voxel_chunk(a, ctl, key, k0);
		sh.vtPtr[e][1] = (old + cnt - 1) / CHUNK > k0 ? wait_voxel_chunk(a, ctl, key, k0 + 1) : nullptr;
		sh.vt.vals[e] = 0;                  // becomes the store cursor
	}
}

// the samples that won cells regenerate their voxel(s) from (level, cell) and store them behind the reserved bases
__device__ void store_voxels(const BuildArgs& a, Ctl* ctl, TileShared& sh, uint32_t leafIdx, uint32_t wins, uint32_t pX, uint32_t pY, uint32_t pZ, float colorBits) {
	const unsigned long long* rec = at<const unsigned long long>(a, a.offPaths) + (uint64_t)leafIdx * PATH_WORDS;
#pragma unroll 1
	for (uint32_t k = 0; wins != 0u && k < PATH_WORDS - 1; k++) {
		const unsigned long long ent = path_entry(a, rec, leafIdx, k);
		if (ent == 0ull) break;
		const int level = (int)path_level(ent);
		if (((wins >> level) & 1u) == 0u) continue;
		wins &= ~(1u << level);
		const uint32_t nodeIdx = path_node(ent);
		const int e = tab_find(sh.vt, nodeIdx);
		if (e < 0) continue;                               // cannot happen: a win bit is only set for a node that has an entry
		const uint32_t base = sh.vtBase[e], slot = base + atomicAdd(&sh.vt.vals[e], 1u), kk = slot / CHUNK, d = kk - base / CHUNK;
		SimlodChunk* c = d == 0u ? sh.vtPtr[e][0] : d == 1u ? sh.vtPtr[e][1] : wait_voxel_chunk(a, ctl, nodeIdx, kk);
		if (c != nullptr) reinterpret_cast<float4*>(c->points)[slot % CHUNK] = voxel_of(a, level, pX, pY, pZ, colorBits);
	}
}

// ---- peek: every PEEK-th sample of the group, counted per leaf — k_ingest's forecast of which leaves will overflow ------------------
__global__ __launch_bounds__(TPB) void k_peek(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active) return;
	__shared__ Tab<LT_BITS> tab;
	tab_init(tab);
	__syncthreads();
	uint32_t* est = at<uint32_t>(a, a.offEst);
	const uint32_t nb = ctl->groupBatches;
	for (uint32_t b = 0; b < nb; b++) {
		const uint32_t n = ctl->batchSize[b];
		const float4* pts = ring_slot(a, ctl->batchSlot[b]);
		for (uint32_t i = (blockIdx.x * TPB + threadIdx.x) * PEEK; i < n; i += gridDim.x * TPB * PEEK) {
			const float4 p = pts[i];
			const uint32_t X = quantize(F_GRID, p.x, a.minx, a.size), Y = quantize(F_GRID, p.y, a.miny, a.size), Z = quantize(F_GRID, p.z, a.minz, a.size);
			const uint32_t leafIdx = (uint32_t)(descend(a.nodes, 0, X, Y, Z) - a.nodes);
			uint32_t rank;
			if (tab_add(tab, leafIdx, 1u, &rank) < 0) atomicAdd(&est[leafIdx], 1u);
		}
	}
	__syncthreads();
	for (uint32_t e = threadIdx.x; e < LT_CAP; e += TPB) if (tab.keys[e] != TBL_EMPTY) atomicAdd(&est[tab.keys[e]], tab.vals[e]);
}

// SIMLOD_PHASE_TIMERS=1: thread 0 of every workgroup adds the wall time of each phase to Ctl.phaseNs (tools/kprof.py prints them)
struct PhaseTimer {
	uint64_t t; unsigned long long* slots; bool on;
	__device__ PhaseTimer(Ctl* ctl, uint32_t first) : t(0), slots(reinterpret_cast<unsigned long long*>(ctl->phaseNs) + first), on((ctl->debugFlags & 2u) != 0u && threadIdx.x == 0) { if (on) t = wall_ns(); }
	__device__ void lap(uint32_t k) { if (on) { const uint64_t n = wall_ns(); atomicAdd(&slots[k], (unsigned long long)(n - t)); t = n; } }
};

// ---- ingest: the whole job for samples whose leaf stays within its limit (voxels.cu:124-229, 417-483, 485-639, 674-698) ----------
template <int P>
__global__ __launch_bounds__(TPB, 4) void k_ingest(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active) return;
	__shared__ TileShared sh;
	constexpr uint32_t TILE = TPB * P;
	uint32_t* pendIdx = at<uint32_t>(a, a.offPendIdx);
	uint32_t* pendLeaf = at<uint32_t>(a, a.offPendLeaf);
	for (uint32_t tile = blockIdx.x;; tile += gridDim.x) {
		uint32_t b, first;
		if (!tile_lookup(ctl, TILE, tile, b, first)) break;
		const uint32_t cnt = min(TILE, ctl->batchSize[b] - first);
		const float4* pts = ring_slot(a, ctl->batchSlot[b]) + first;
		const uint32_t vbase = b * SIMLOD_MAX_BATCH_SIZE + first;
		PhaseTimer timer(ctl, 0);
		__syncthreads();
		tile_reset(sh);
		__syncthreads();

		// 1: one coalesced 16-byte load per sample, ONE root -> leaf descent, count per (workgroup, leaf) in LDS
		float4 p[P];
		uint32_t leafOf[P], er[P];                     // er: table entry << 16 | rank inside the workgroup, NONE without an entry
#pragma unroll
		for (int j = 0; j < P; j++) {
			const uint32_t i = j * TPB + threadIdx.x;
			p[j] = i < cnt ? pts[i] : make_float4(0, 0, 0, 0);
		}
		{
			uint32_t X[P], Y[P], Z[P], level[P];
			bool walking[P];
#pragma unroll
			for (int j = 0; j < P; j++) {
				walking[j] = (uint32_t)j * TPB + threadIdx.x < cnt;
				X[j] = quantize(F_GRID, p[j].x, a.minx, a.size); Y[j] = quantize(F_GRID, p[j].y, a.miny, a.size); Z[j] = quantize(F_GRID, p[j].z, a.minz, a.size);
				leafOf[j] = walking[j] ? 0u : NONE; level[j] = 0; er[j] = NONE;
			}
			descend_lockstep<P>(a.nodes, leafOf, level, X, Y, Z, walking);
		}
#pragma unroll
		for (int j = 0; j < P; j++) {
			const uint32_t i = j * TPB + threadIdx.x;
			if (i >= cnt) { leafOf[j] = NONE; continue; }
			const uint32_t leafIdx = leafOf[j];
			uint32_t rank;
			const int e = tab_add(sh.lt, leafIdx, 1u, &rank);
			if (e >= 0) er[j] = ((uint32_t)e << 16) | rank;
			else {
				// no room in the table: this sample is its own (workgroup, leaf) entry.  Its slot comes from the arrival counter like
				// everybody's, so the leaf's storage stays gap-free; whoever allocates its chunk reserved an earlier slot and never waits
				const bool hot = leaf_is_hot(a, leafIdx);
				const uint32_t old = count_into(a, ctl, leafIdx, 1u, !hot);
				if (!hot && old + 1u > MAXPTS) atomicSub(&a.nodes[leafIdx].numPoints, 1u);
				if (!hot && old + 1u <= MAXPTS) {
					if (old % CHUNK == 0u) make_point_chunk(a, ctl, leafIdx, old / CHUNK);
					SimlodChunk* c = wait_point_chunk(a, ctl, leafIdx, old / CHUNK);
					if (c != nullptr) reinterpret_cast<float4*>(c->points)[old % CHUNK] = p[j];
					er[j] = STORED;
				}
			}
		}
		__syncthreads();
		timer.lap(0);

		// 2: arrival counters, slot ranges, chunks
		flush_points<true>(a, ctl, sh);
		__syncthreads();
		timer.lap(1);

		// 3: store, sample; what cannot be placed yet is queued for k_place
		constexpr uint32_t WAITS = 0x80000000u;             // wins[j]: levels won (bits 0..19), or WAITS | index in the workgroup's share of the k_place queue
		uint32_t wins[P];
		Probe pr[P];
#pragma unroll
		for (int j = 0; j < P; j++) {                       // the probes of all samples of the thread are in flight together
			wins[j] = 0;
			if (leafOf[j] == NONE) continue;
			const bool placed = er[j] == STORED || (er[j] != NONE && sh.ltBase[er[j] >> 16] != NONE);
			if (!placed) { wins[j] = WAITS; continue; }
			probe_load(a, leafOf[j], 0u, quantize(F_FULL, p[j].x, a.minx, a.size), quantize(F_FULL, p[j].y, a.miny, a.size), quantize(F_FULL, p[j].z, a.minz, a.size), 0u, pr[j]);
		}
#pragma unroll
		for (int j = 0; j < P; j++) {
			if (leafOf[j] == NONE) continue;
			if (wins[j] == 0u) {
				if (er[j] != STORED) store_point(a, ctl, sh, er[j] >> 16, er[j] & 0xffffu, p[j]);
				const uint32_t pX = quantize(F_FULL, p[j].x, a.minx, a.size), pY = quantize(F_FULL, p[j].y, a.miny, a.size), pZ = quantize(F_FULL, p[j].z, a.minz, a.size);
				wins[j] = sample_path(a, ctl, sh, leafOf[j], 0u, pX, pY, pZ, p[j].w, pr[j]);
			} else {
				// one LDS atomic per wave: the lanes that wait are numbered by their rank among the waiting lanes
				const unsigned long long waiting = __ballot(1);
				const uint32_t lane = (uint32_t)lane_id(), leader = (uint32_t)__ffsll((long long)waiting) - 1u;
				uint32_t first = 0;
				if (lane == leader) first = atomicAdd(&sh.pendCount, (uint32_t)__popcll(waiting));
				first = __shfl(first, (int)leader);
				wins[j] = WAITS | (first + (uint32_t)__popcll(waiting & ((1ull << lane) - 1ull)));
			}
		}
		__syncthreads();
		timer.lap(2);
		if (threadIdx.x == 0 && sh.pendCount != 0u) sh.pendBase = atomicAdd(&ctl->numPending, sh.pendCount);

		// 4: voxel slot ranges and chunks
		flush_voxels(a, ctl, sh);
		__syncthreads();
		timer.lap(3);

		// 5: voxel stores; the queue entries of the samples left to k_place
#pragma unroll
		for (int j = 0; j < P; j++) {
			if (wins[j] != 0u && (wins[j] & WAITS) == 0u) {
				const uint32_t pX = quantize(F_FULL, p[j].x, a.minx, a.size), pY = quantize(F_FULL, p[j].y, a.miny, a.size), pZ = quantize(F_FULL, p[j].z, a.minz, a.size);
				store_voxels(a, ctl, sh, leafOf[j], wins[j], pX, pY, pZ, p[j].w);
			}
			if ((wins[j] & WAITS) != 0u) {
				const uint32_t q = sh.pendBase + (wins[j] & ~WAITS);
				if (q < a.pendCap) { pendIdx[q] = vbase + j * TPB + threadIdx.x; pendLeaf[q] = leafOf[j]; }
			}
		}
		timer.lap(4);
	}
}

// ---- expand: the split cascade (voxels.cu:385-415, 245-289, 308-383) ------------------------------------------------------------
// Persistent, one workgroup per two CUs, launched cooperatively (all workgroups resident), hand-rolled grid barrier.
// Per round, for the leaves of the round's work list (round 0: the leaves k_ingest saw overflow; later: nodes created by the
// previous round that are still too full):
//   H) histogram, per listed leaf, of everything that has to go below it — its stored points (moved to the spill buffer on the
//      way, round 0) and the samples waiting for k_place — over the 8^nl cells nl = 3 levels further down (2 when the list is
//      longer than the histogram space at 512 bins each);
//   -- barrier --
//   D) one workgroup per (leaf, child): from the histogram alone it creates the child, and while a descendant holds more than
//      MAX_POINTS_PER_NODE samples, that one's children too, down to nl generations, arrival counters filled in, occupancy grids
//      allocated and cleared, ancestor paths written; descendants of the last generation that are still too full go on the next
//      round's list.  The leaf's chunks return to the recycle stack.
// Nothing waits for a barrier while the tree is half modified, and k_ingest reserved everything a round-0 split needs, so a round
// either happens completely or — a barrier that gives up — not at all.
static constexpr uint32_t ETPB = 1024;             // at most one workgroup per CU (grid barrier participants), 16 waves each
static constexpr int HT_BITS = 12;

struct ExpandShared {
	Tab<HT_BITS> ht;                               // (list entry << 9 | bin) -> count, for the scan of the waiting samples
	uint32_t dense[512];                           // one stored chunk's histogram
	uint32_t hc[64];                               // D: bins below one child
	uint32_t childSplit, gcBase, split2Mask, ggBase[8], queueMask2, queueMask3[8];
	SimlodOccupancyGrid* gridC;
	SimlodOccupancyGrid* gridG[8];
};

__device__ __forceinline__ int child_at(uint32_t X, uint32_t Y, uint32_t Z, uint32_t level) {
	return level < (uint32_t)SIMLOD_MAX_DEPTH ? child_index(X, Y, Z, (int)level) : 0;
}
// histogram bin of a sample below a node of `level`: its child, grand-child (and great-grand-child) octant
__device__ __forceinline__ uint32_t bin_of(const BuildArgs& a, const float4& p, uint32_t level, uint32_t nl) {
	const uint32_t X = quantize(F_GRID, p.x, a.minx, a.size), Y = quantize(F_GRID, p.y, a.miny, a.size), Z = quantize(F_GRID, p.z, a.minz, a.size);
	uint32_t bin = (uint32_t)child_at(X, Y, Z, level) * 8u + (uint32_t)child_at(X, Y, Z, level + 1);
	if (nl == 3u) bin = bin * 8u + (uint32_t)child_at(X, Y, Z, level + 2);
	return bin;
}

__device__ __forceinline__ const SimlodChunk* leaf_chunk(const BuildArgs& a, uint32_t leafIdx, uint32_t k) {
	SimlodChunk* const* slots = at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)leafIdx * LEAF_SLOTS;
	if (k < LEAF_SLOTS) return slots[k];
	const SimlodChunk* c = slots[LEAF_SLOTS - 1];                  // a leaf whose split was deferred and that kept growing: walk
	for (uint32_t i = LEAF_SLOTS - 1; i < k && c != nullptr; i++) c = c->next;
	return c;
}

// write one freshly created node, field by field straight to the node array (a 152-byte local would live in scratch memory)
__device__ void write_node(const BuildArgs& a, uint32_t idx, uint32_t parentIdx, uint32_t octant, uint32_t counter, SimlodNode* firstChild, SimlodOccupancyGrid* grid) {
	SimlodNode& c = a.nodes[idx];
	const SimlodNode& par = a.nodes[parentIdx];
	const uint32_t level = par.level + 1u;
	for (int k = 0; k < 8; k++) c.children[k] = firstChild != nullptr ? firstChild + k : nullptr;
	c.counter = counter; c.numPoints = 0;
	c.level = level;
	c.X = 2 * par.X + ((octant >> 2) & 1u);
	c.Y = 2 * par.Y + ((octant >> 1) & 1u);
	c.Z = 2 * par.Z + (octant & 1u);
	c.countIteration = 0; c.countFlag = 0;
	for (int k = 0; k < 20; k++) c.name[k] = par.name[k];
	if (level < 20u) c.name[level] = (uint8_t)('0' + octant);
	c.visible = 0; c.isFiltered = 0; c.isLeaf = 1; c.isLarge = 0;
	c.grid = grid; c.points = nullptr; c.voxelChunks = nullptr;
	c.numVoxels = 0; c.numVoxelsStored = 0;
	at<uint32_t>(a, a.offParent)[idx] = parentIdx;
	at<uint32_t>(a, a.offPtStart)[idx] = 0;
	at<uint32_t>(a, a.offVoxStart)[idx] = 0;
	// its ancestors: the parent (whose grid is final), then the parent's own ancestors
	unsigned long long* paths = at<unsigned long long>(a, a.offPaths);
	const unsigned long long* mine = paths + (uint64_t)parentIdx * PATH_WORDS;
	unsigned long long* theirs = paths + (uint64_t)idx * PATH_WORDS;
	theirs[0] = path_pack(a.pers, parentIdx, par.level, par.grid);
	for (uint32_t k = 0; k + 1 < PATH_WORDS; k++) {
		const unsigned long long e = k + 2 < PATH_WORDS ? mine[k] : 0ull;
		theirs[k + 1] = e;
		if (e == 0ull) break;
	}
	SimlodChunk** slots = at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)idx * LEAF_SLOTS;
	for (uint32_t k = 0; k < LEAF_SLOTS; k++) slots[k] = nullptr;
}

__device__ __forceinline__ void clear_grid(SimlodOccupancyGrid* g, uint32_t firstWord4, uint32_t numWords4) {
	uint4* w = reinterpret_cast<uint4*>(g->values) + firstWord4;
	const uint4 z = make_uint4(0, 0, 0, 0);
	for (uint32_t i = threadIdx.x; i < numWords4; i += ETPB) w[i] = z;
}

// roundFirst == 0: round 0 only (the common case is a cascade that ends within three levels: one barrier, then the kernel boundary
// does the rest).  roundFirst == 1: the remaining rounds, until a round leaves no node over the limit.
__global__ __launch_bounds__(ETPB) void k_expand(BuildArgs a, uint32_t roundFirst) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || ctl->abortBatch) return;
	if (roundFirst == 0u ? ctl->numSpilling == 0u : ctl->roundSpill[0] == 0u) return;      // stable: written before this launch, never modified by it (roundSpill[0]: see the zeroing below)

	__shared__ ExpandShared sh;
	uint32_t* hist = at<uint32_t>(a, a.offHist);
	uint32_t* pendIdx = at<uint32_t>(a, a.offPendIdx);
	uint32_t* pendLeaf = at<uint32_t>(a, a.offPendLeaf);
	uint32_t* spMeta = at<uint32_t>(a, a.offSpMeta);
	float4* spilled = at<float4>(a, a.offSpilled);
	const unsigned long long* splitInfo = at<const unsigned long long>(a, a.offSplitTag);
	SimlodChunk** chunkQueue = at<SimlodChunk*>(a, a.offQueue);
	SimlodChunk** leafChunks = at<SimlodChunk*>(a, a.offLeafChunks);
	const uint32_t numPending = min(ctl->numPending, a.pendCap);
	const uint32_t spEnd = spilled_end(ctl);
	uint32_t generation = 0;
	const bool forceTimeout = (ctl->debugFlags & 1u) != 0u;

	// before anything returns to the recycle stack: free chunks are chunkQueue[numAllocatedChunks .. chunkPoolSize)
	if (roundFirst == 0u && blockIdx.x == 0 && threadIdx.x == 0) {
		if (a.stats->numAllocatedChunks > a.stats->chunkPoolSize) a.stats->chunkPoolSize = a.stats->numAllocatedChunks;
		ctl->expandNs[5] += 1;
	}

	for (uint32_t round = roundFirst;; ++round) {
		SpillEntry* listCur = at<SpillEntry>(a, (round & 1u) ? a.offSpillB : a.offSpillA);
		SpillEntry* listNext = at<SpillEntry>(a, (round & 1u) ? a.offSpillA : a.offSpillB);
		uint32_t* countCur = round == 0u ? &ctl->numSpilling : &ctl->roundSpill[(round - 1u) & 1u];
		uint32_t* countNext = &ctl->roundSpill[round & 1u];
		const uint32_t n = min(*countCur, a.histCap);
		if (n == 0u) break;
		const uint32_t nl = n * 512u <= a.histCap * 64u ? 3u : 2u;     // the histogram space holds histCap entries at 64 bins
		const uint32_t bins = nl == 3u ? 512u : 64u;
		const uint32_t tag = round_tag(ctl, round);
		if (blockIdx.x == 0 && threadIdx.x == 0) { if (round >= 2u) *countNext = 0; ctl->expandNs[4] += 1; }   // last read one round ago, appended to only after the barrier below

		// -- H1: stored points of the listed leaves -> spill buffer, histogram per chunk (entries of later rounds are empty nodes)
		if (round == 0u) for (uint32_t s = 0; s < n; s++) {
			const SpillEntry en = listCur[s];
			if (en.leaf == NONE || en.stored == 0u) continue;
			const SimlodNode* L = a.nodes + en.leaf;
			const uint32_t actual = min(L->numPoints, en.stored), lvl = L->level;
			const uint32_t numChunks = (en.stored + CHUNK - 1) / CHUNK;
			for (uint32_t k = blockIdx.x; k < numChunks; k += gridDim.x) {
				__syncthreads();
				for (uint32_t i = threadIdx.x; i < 512u; i += ETPB) sh.dense[i] = 0;
				__syncthreads();
				const SimlodChunk* c = k * CHUNK < actual ? leaf_chunk(a, en.leaf, k) : nullptr;
				const uint32_t j = threadIdx.x, idx = k * CHUNK + j;
				if (j < CHUNK && idx < en.stored) {
					const uint32_t dst = en.spillBase + idx;
					if (idx < actual && c != nullptr) {
						const float4 p = reinterpret_cast<const float4*>(c->points)[j];
						spilled[dst] = p;
						spMeta[dst] = en.leaf | (lvl << 19);           // sampling restarts at the spilling node's level
						atomicAdd(&sh.dense[bin_of(a, p, lvl, nl)], 1u);
					} else spMeta[dst] = NONE;                        // reserved for arrivals that were not stored after all
				}
				__syncthreads();
				for (uint32_t i = threadIdx.x; i < bins; i += ETPB) if (sh.dense[i] != 0u) atomicAdd(&hist[(uint64_t)s * bins + i], sh.dense[i]);
			}
		}

		// -- H2: the samples waiting for k_place (and, from round 1 on, the moved points): follow the tree as far as it goes now,
		//        count those that end in a listed leaf
		{
			__syncthreads();
			tab_init(sh.ht);
			__syncthreads();
			const uint32_t total = numPending + (round == 0u ? 0u : spEnd);
			for (uint32_t q = blockIdx.x * ETPB + threadIdx.x; q < total; q += gridDim.x * ETPB) {
				const bool isPend = q < numPending;
				const uint32_t meta = isPend ? pendLeaf[q] : spMeta[q - numPending];
				if (meta == NONE) continue;
				uint32_t cur = meta & 0x7ffffu;
				float4 p = make_float4(0, 0, 0, 0);
				bool have = false;
				// a cached node that an earlier round of this group split has a record with that round's tag: one 8-byte load tells
				const uint32_t cachedTag = (uint32_t)(splitInfo[cur] >> 32);
				if (round != 0u && cachedTag >= round_tag(ctl, 0u) && cachedTag < tag) {
					p = isPend ? point_of(a, ctl, pendIdx[q]) : spilled[q - numPending]; have = true;
					const uint32_t X = quantize(F_GRID, p.x, a.minx, a.size), Y = quantize(F_GRID, p.y, a.miny, a.size), Z = quantize(F_GRID, p.z, a.minz, a.size);
					cur = (uint32_t)(descend(a.nodes + cur, (int)a.nodes[cur].level, X, Y, Z) - a.nodes);
					if (isPend) pendLeaf[q] = cur; else spMeta[q - numPending] = cur | (meta & ~0x7ffffu);
				}
				const unsigned long long info = splitInfo[cur];
				if ((uint32_t)(info >> 32) != tag) continue;
				const uint32_t s = (uint32_t)info;
				if (s >= n) continue;
				if (!have) p = isPend ? point_of(a, ctl, pendIdx[q]) : spilled[q - numPending];
				const uint32_t key = s * bins + bin_of(a, p, a.nodes[cur].level, nl);
				uint32_t rank;
				if (tab_add(sh.ht, key, 1u, &rank) < 0) atomicAdd(&hist[key], 1u);
			}
			__syncthreads();
			for (uint32_t e = threadIdx.x; e < (uint32_t)Tab<HT_BITS>::CAP; e += ETPB) {
				const uint32_t key = sh.ht.keys[e];
				if (key != TBL_EMPTY) atomicAdd(&hist[key], sh.ht.vals[e]);
			}
		}

		if (!grid_barrier(&ctl->barrierCount[roundFirst], generation, gridDim.x, forceTimeout)) { if (threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }

		// -- D: decide and build, one workgroup per (listed leaf, child octant)
		const uint32_t sub = bins / 8u;
		if (threadIdx.x == 0) ctl->treeModified = 1;
		for (uint32_t item = blockIdx.x; item < n * 8u; item += gridDim.x) {
			const uint32_t s = item >> 3, c1 = item & 7u;
			const SpillEntry en = listCur[s];
			if (en.leaf == NONE) continue;
			SimlodNode* L = a.nodes + en.leaf;
			const uint32_t lvl = L->level;
			const uint32_t childIdx = en.childBase + c1;
			__syncthreads();
			if (threadIdx.x < 64u) {
				uint32_t* h = hist + (uint64_t)s * bins + c1 * sub;
				sh.hc[threadIdx.x] = threadIdx.x < sub ? h[threadIdx.x] : 0u;
				if (threadIdx.x < sub) h[threadIdx.x] = 0;           // the histogram space is clean again for the next round / batch
			}
			__syncthreads();
			if (threadIdx.x == 0) {
				// which descendants exist: a node splits while it holds more than MAX_POINTS_PER_NODE samples and is above MAX_DEPTH
				// (voxels.cu:203-218) and eight node slots can still be had — otherwise it stays a leaf, over-full, and is queued
				// again by a later batch
				uint32_t count1 = 0;
				for (uint32_t i = 0; i < sub; i++) count1 += sh.hc[i];
				sh.childSplit = 0; sh.gcBase = 0; sh.split2Mask = 0; sh.queueMask2 = 0; sh.gridC = nullptr;
				for (int i = 0; i < 8; i++) { sh.ggBase[i] = 0; sh.queueMask3[i] = 0; sh.gridG[i] = nullptr; }
				uint32_t nb, sb;
				if (count1 > MAXPTS && lvl + 1u < (uint32_t)SIMLOD_MAX_DEPTH && reserve(a, ctl, 8u, 0u, nb, sb)) {
					sh.childSplit = 1; sh.gcBase = nb;
					sh.gridC = reinterpret_cast<SimlodOccupancyGrid*>(persistent_alloc(a.pers, sizeof(SimlodOccupancyGrid), 1));
					for (uint32_t c2 = 0; c2 < 8u; c2++) {
						uint32_t count2 = 0;
						if (nl == 3u) for (uint32_t i = 0; i < 8u; i++) count2 += sh.hc[c2 * 8u + i]; else count2 = sh.hc[c2];
						if (count2 <= MAXPTS || lvl + 2u >= (uint32_t)SIMLOD_MAX_DEPTH) continue;
						if (nl == 2u) { sh.queueMask2 |= 1u << c2; continue; }     // its histogram is the next round's business
						if (!reserve(a, ctl, 8u, 0u, nb, sb)) continue;
						sh.split2Mask |= 1u << c2; sh.ggBase[c2] = nb;
						sh.gridG[c2] = reinterpret_cast<SimlodOccupancyGrid*>(persistent_alloc(a.pers, sizeof(SimlodOccupancyGrid), 1));
						for (uint32_t c3 = 0; c3 < 8u; c3++)
							if (sh.hc[c2 * 8u + c3] > MAXPTS && lvl + 3u < (uint32_t)SIMLOD_MAX_DEPTH) sh.queueMask3[c2] |= 1u << c3;
					}
				}
				L->children[c1] = a.nodes + childIdx;
			}
			__syncthreads();
			// the nodes: thread 0 the child, 1..8 its children, 9..72 theirs — parents first (a node copies its parent's record)
			if (threadIdx.x == 0) {
				uint32_t count1 = 0;
				for (uint32_t i = 0; i < sub; i++) count1 += sh.hc[i];
				write_node(a, childIdx, en.leaf, c1, count1, sh.childSplit ? a.nodes + sh.gcBase : nullptr, sh.gridC);
			}
			__syncthreads();
			if (sh.childSplit && threadIdx.x >= 1u && threadIdx.x < 9u) {
				const uint32_t c2 = threadIdx.x - 1u;
				uint32_t count2 = 0;
				if (nl == 3u) for (uint32_t i = 0; i < 8u; i++) count2 += sh.hc[c2 * 8u + i]; else count2 = sh.hc[c2];
				const bool split = ((sh.split2Mask >> c2) & 1u) != 0u;
				write_node(a, sh.gcBase + c2, childIdx, c2, count2, split ? a.nodes + sh.ggBase[c2] : nullptr, sh.gridG[c2]);
			}
			__syncthreads();
			if (threadIdx.x >= 9u && threadIdx.x < 73u) {
				const uint32_t c2 = (threadIdx.x - 9u) >> 3, c3 = (threadIdx.x - 9u) & 7u;
				if (((sh.split2Mask >> c2) & 1u) != 0u) write_node(a, sh.ggBase[c2] + c3, sh.gcBase + c2, c3, sh.hc[c2 * 8u + c3], nullptr, nullptr);
			}
			// the occupancy grids: this workgroup's eighth of the leaf's own — of EVERY split node, also one that already had a grid
			// (the root), voxels.cu:371-382 — and the whole grid of every descendant it split
			clear_grid(L->grid, c1 * (SIMLOD_GRID_NUM_WORDS / 32u), SIMLOD_GRID_NUM_WORDS / 32u);
			if (sh.childSplit) clear_grid(sh.gridC, 0u, SIMLOD_GRID_NUM_WORDS / 4u);
			for (uint32_t c2 = 0; c2 < 8u; c2++) if (((sh.split2Mask >> c2) & 1u) != 0u) clear_grid(sh.gridG[c2], 0u, SIMLOD_GRID_NUM_WORDS / 4u);
			if (c1 == 0u && threadIdx.x >= 64u && threadIdx.x < 128u) {
				// Wave 1 hands the leaf's chunks back to the recycle stack (voxels.cu:346-357; nothing pops before k_place).  Chunk k
				// comes from the leaf chunk table, not from a walk.
				const uint32_t lane = threadIdx.x - 64u;
				const uint32_t stored = L->numPoints;
				const uint32_t numChunks = L->points != nullptr ? (stored + CHUNK - 1) / CHUNK : 0u;
				unsigned long long top = 0;
				if (lane == 0 && numChunks > 0) top = atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats->numAllocatedChunks), (unsigned long long)(-(long long)numChunks));
				top = ((unsigned long long)__shfl((uint32_t)(top >> 32), 0) << 32) | __shfl((uint32_t)top, 0);
				SimlodChunk** slots = leafChunks + (uint64_t)en.leaf * LEAF_SLOTS;
				SimlodChunk* beyond = nullptr;                      // chunk #LEAF_SLOTS of a leaf whose split was deferred and that kept growing
				if (lane == 0 && numChunks > LEAF_SLOTS) beyond = slots[LEAF_SLOTS - 1]->next;
				for (uint32_t ci = lane; ci < min(numChunks, LEAF_SLOTS); ci += 64) {
					SimlodChunk* chunk = slots[ci];
					const unsigned long long q = top - numChunks + ci;
					if (q < CHUNK_QUEUE_CAPACITY) chunkQueue[q] = chunk; else raise(ctl, SIMLOD_ERR_CHUNK_QUEUE_OVERFLOW);
					chunk->next = nullptr;
				}
				if (lane == 0) for (uint32_t ci = LEAF_SLOTS; ci < numChunks && beyond != nullptr; ci++) {   // the table has no slot for these: walk
					SimlodChunk* next = beyond->next;
					const unsigned long long q = top - numChunks + ci;
					if (q < CHUNK_QUEUE_CAPACITY) chunkQueue[q] = beyond; else raise(ctl, SIMLOD_ERR_CHUNK_QUEUE_OVERFLOW);
					beyond->next = nullptr;
					beyond = next;
				}
				for (uint32_t ci = lane; ci < LEAF_SLOTS; ci += 64) slots[ci] = nullptr;
				if (lane == 0) {
					atomicAdd(reinterpret_cast<unsigned long long*>(&ctl->spilledTotal), (unsigned long long)stored);
					L->numPoints = 0;
					L->points = nullptr;
				}
			}
			__syncthreads();
			// descendants of the last generation that are still too full: next round (reserves their slots and grids now)
			if (threadIdx.x == 0) {
				for (uint32_t c2 = 0; c2 < 8u; c2++) {
					if (((sh.queueMask2 >> c2) & 1u) != 0u) queue_split(a, ctl, sh.gcBase + c2, 0u, round + 1u, listNext, countNext);
					for (uint32_t c3 = 0; c3 < 8u; c3++)
						if (((sh.queueMask3[c2] >> c3) & 1u) != 0u) queue_split(a, ctl, sh.ggBase[c2] + c3, 0u, round + 1u, listNext, countNext);
				}
			}
		}
		if (roundFirst == 0u) break;                            // the kernel boundary is the barrier
		if (!grid_barrier(&ctl->barrierCount[roundFirst], generation, gridDim.x, false)) { if (threadIdx.x == 0) panic(ctl, SIMLOD_ERR_BARRIER_TIMEOUT); return; }
	}
}

// ---- prealloc: the leaves the cascade created know how many samples k_place will bring them: all their chunks with ONE pop of the
// recycle stack and ONE allocation per leaf (voxels.cu:485-538), instead of a thousand on-demand allocations on two hot words while
// every workgroup waits -------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_prealloc(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || ctl->abortBatch) return;
	const uint32_t last = min(a.stats->numNodes, a.nodeCapacity);
	SimlodChunk** queue = at<SimlodChunk*>(a, a.offQueue);
	for (uint32_t i = ctl->nodesAtStart + blockIdx.x * TPB + threadIdx.x; i < last; i += gridDim.x * TPB) {
		SimlodNode* node = a.nodes + i;
		const uint32_t cnt = node->counter;
		if (cnt == 0u || !node_is_leaf(node)) continue;
		const uint32_t n = min((cnt + CHUNK - 1) / CHUNK, LEAF_SLOTS);           // an over-full new leaf gets the rest on demand
		const unsigned long long idx = atomicAdd(reinterpret_cast<unsigned long long*>(&a.stats->numAllocatedChunks), (unsigned long long)n);
		const unsigned long long pool = a.stats->chunkPoolSize;
		const uint32_t fromPool = idx >= pool ? 0u : (uint32_t)min((unsigned long long)n, pool - idx);
		uint8_t* fresh = n > fromPool ? persistent_alloc(a.pers, sizeof(SimlodChunk), n - fromPool) : nullptr;
		SimlodChunk** slots = at<SimlodChunk*>(a, a.offLeafChunks) + (uint64_t)i * LEAF_SLOTS;
		SimlodChunk* prev = nullptr;
		SimlodChunk* head = nullptr;
		for (uint32_t k = 0; k < n; k++) {
			SimlodChunk* c = k < fromPool ? queue[idx + k] : reinterpret_cast<SimlodChunk*>(fresh + (uint64_t)(k - fromPool) * SIMLOD_ALLOC_ROUND(sizeof(SimlodChunk)));
			c->next = nullptr;
			if (prev != nullptr) prev->next = c; else head = c;
			slots[k] = c;
			prev = c;
		}
		node->points = head;
		tail_of(head) = prev;
	}
}

// ---- place: samples that k_ingest could not place — their leaf overflowed, or was forecast to — and the stored points of the split
// leaves go into the leaves that exist now: slot ranges per (workgroup, leaf), 16-byte stores (voxels.cu:540-639).  Their voxel
// sampling is k_voxelize's job --------------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(TPB) void k_place(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || ctl->abortBatch) return;
	const uint32_t numPending = min(ctl->numPending, a.pendCap);
	const uint32_t total = numPending + spilled_end(ctl);
	if (total == 0u) return;
	__shared__ TileShared sh;
	constexpr uint32_t TILE = TPB * P;
	const uint32_t* pendIdx = at<const uint32_t>(a, a.offPendIdx);
	const uint32_t* pendLeaf = at<const uint32_t>(a, a.offPendLeaf);
	const uint32_t* spMeta = at<const uint32_t>(a, a.offSpMeta);
	const float4* spilled = at<const float4>(a, a.offSpilled);
	const uint32_t numTiles = (total + TILE - 1) / TILE;
	for (uint32_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
		PhaseTimer timer(ctl, 8);
		__syncthreads();
		tab_init(sh.lt);
		__syncthreads();
		float4 p[P];
		uint32_t leafOf[P], er[P];
#pragma unroll
		for (int j = 0; j < P; j++) {
			const uint32_t q = tile * TILE + j * TPB + threadIdx.x;
			leafOf[j] = NONE; er[j] = NONE; p[j] = make_float4(0, 0, 0, 0);
			if (q >= total) continue;
			const uint32_t meta = q < numPending ? pendLeaf[q] : spMeta[q - numPending];
			if (meta == NONE) continue;
			p[j] = q < numPending ? point_of(a, ctl, pendIdx[q]) : spilled[q - numPending];
			leafOf[j] = meta & 0x7ffffu;
		}
		{
			uint32_t X[P], Y[P], Z[P], level[P];
			bool walking[P];
#pragma unroll
			for (int j = 0; j < P; j++) {
				walking[j] = leafOf[j] != NONE;
				X[j] = quantize(F_GRID, p[j].x, a.minx, a.size); Y[j] = quantize(F_GRID, p[j].y, a.miny, a.size); Z[j] = quantize(F_GRID, p[j].z, a.minz, a.size);
				level[j] = walking[j] ? a.nodes[leafOf[j]].level : 0u;              // from the cached node down
				if (!walking[j]) leafOf[j] = 0u;
			}
			bool valid[P];
#pragma unroll
			for (int j = 0; j < P; j++) valid[j] = walking[j];
			descend_lockstep<P>(a.nodes, leafOf, level, X, Y, Z, walking);
#pragma unroll
			for (int j = 0; j < P; j++) {
				if (!valid[j]) { leafOf[j] = NONE; continue; }
				uint32_t rank;
				const int e = tab_add(sh.lt, leafOf[j], 1u, &rank);
				if (e >= 0) er[j] = ((uint32_t)e << 16) | rank;
			}
		}
		__syncthreads();
		timer.lap(0);
		flush_points<false>(a, ctl, sh);
		__syncthreads();
		timer.lap(1);
#pragma unroll
		for (int j = 0; j < P; j++) {                       // every allocation before any lookup: first the samples that own their slot bookkeeping
			if (leafOf[j] != NONE && er[j] == NONE) store_point_direct(a, ctl, leafOf[j], p[j]);
		}
#pragma unroll
		for (int j = 0; j < P; j++) if (leafOf[j] != NONE && er[j] != NONE) store_point(a, ctl, sh, er[j] >> 16, er[j] & 0xffffu, p[j]);
		timer.lap(2);
	}
}

// ---- voxelize: 128^3 occupancy sampling of what k_place stored (voxels.cu:50-121, 417-483), ONE workgroup per leaf ---------------------
// k_place's samples are the contended ones: a batch that enters new territory puts its points into a few dozen fresh leaves under a
// handful of fresh inner nodes whose grids are empty — a million test-and-set attempts on a few hundred 128-byte lines, twelve
// candidates per cell from twelve different workgroups when samples are taken in arrival order.  But placed, the samples of a leaf
// sit side by side in the leaf's chunks, and a leaf owns a CUBE of every ancestor's grid: 64^3 cells of its parent's (32 KB of
// bits), 32^3 of the grand-parent's, ... down to one cell seven levels up — cubes of different leaves are disjoint.  So the
// workgroup that owns a leaf copies those cubes into LDS, runs the whole test-and-set cascade of the leaf's new samples there
// (bottom-up, climbing while a cell is new, as everywhere in this file), and writes the cubes back: no global atomic per sample, none
// contended at all.  Only above the seventh ancestor, where several leaves share a cell, the global atomicOr decides.
// Pass A marks the new cells; their number per ancestor reserves a voxel slot range with ONE atomic per (leaf, ancestor); pass B
// walks the samples again and whoever finds its cell still marked new takes the mark and stores the voxel with its own colour (which
// point of a cell colours the voxel is scheduling dependent in the reference too).
// Leaves with few new samples (and a root that is still a leaf: its cube is the whole grid) take the per-sample path with global
// atomics; there is nothing to contend for.
static constexpr uint32_t VTPB = 1024;
static constexpr uint32_t BULK_MIN = 768;
static constexpr uint32_t LDS_LEVELS = 7;           // ancestors whose cube of this leaf has at least one whole cell: side 128 >> d
static constexpr uint32_t CUBE_WORDS = 8192 + 1024 + 256 + 64 + 16 + 4 + 4;
static_assert(VOX_PIECE == VTPB * VOX_SPT, "a piece = VOX_SPT samples per thread");
struct VoxShared {
	uint32_t occ[CUBE_WORDS];                        // cubes d = 1..7: rows of (128 >> d) x-bits, one row per word from d = 2 on
	uint32_t fresh[CUBE_WORDS];                      // pass A: cells this piece set; after the write-back: cells it won
	unsigned long long anc[PATH_WORDS];
	uint32_t cnt[PATH_WORDS], base[PATH_WORDS], cursor[PATH_WORDS];
	SimlodChunk* ptr[PATH_WORDS][2];
};
__device__ __forceinline__ uint32_t cube_offset(uint32_t d) {          // word offset of cube d in VoxShared::occ / fresh
	return d == 1u ? 0u : d == 2u ? 8192u : d == 3u ? 9216u : d == 4u ? 9472u : d == 5u ? 9536u : d == 6u ? 9552u : 9556u;
}
// word and bit of grid cell `cell` of ancestor d inside the leaf's cube (side = 128 >> d; the cube is aligned to its side)
__device__ __forceinline__ void cube_cell(uint32_t d, uint32_t cell, uint32_t& word, uint32_t& bit) {
	const uint32_t side = 128u >> d, lx = (cell & 127u) & (side - 1u), ly = ((cell >> 7) & 127u) & (side - 1u), lz = (cell >> 14) & (side - 1u);
	const uint32_t row = ly + side * lz;
	if (d == 1u) { word = row * 2u + (lx >> 5); bit = lx & 31u; }
	else { word = cube_offset(d) + row; bit = lx; }
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

__device__ __forceinline__ const SimlodChunk* placed_chunk(const BuildArgs& a, Ctl* ctl, uint32_t leafIdx, uint32_t k, uint32_t start) {
	if (k < LEAF_SLOTS) return at<SimlodChunk*>(a, a.offLeafChunks)[(uint64_t)leafIdx * LEAF_SLOTS + k];
	if (k * CHUNK < start) return tail_of(a.nodes[leafIdx].points);       // an over-full leaf: the tail it had when the batch began
	return dir_find(a, ctl, KIND_PT, leafIdx, k);
}

__device__ __forceinline__ float4 voxel_at(const BuildArgs& a, uint32_t level, uint32_t nX, uint32_t nY, uint32_t nZ, uint32_t cell, float colorBits) {
	const uint32_t cx = cell & 127u, cy = (cell >> 7) & 127u, cz = cell >> 14;
	const float nodeSize = a.size / exp2_int(level);                        // voxels.cu:103-114, operation by operation
	const float nminx = ((float)nX + 0.0f) * nodeSize + a.minx;
	const float nminy = ((float)nY + 0.0f) * nodeSize + a.miny;
	const float nminz = ((float)nZ + 0.0f) * nodeSize + a.minz;
	float4 v;
	v.x = nminx + (nodeSize * ((float)cx + 0.5f)) / 128.0f;
	v.y = nminy + (nodeSize * ((float)cy + 0.5f)) / 128.0f;
	v.z = nminz + (nodeSize * ((float)cz + 0.5f)) / 128.0f;
	v.w = colorBits;
	return v;
}

// One workgroup per work-list entry: a leaf's piece of VOX_PIECE new samples, eight per thread, kept in registers through both passes.
// Global memory is touched in a few steps, each one round trip with everything it needs in flight together: the leaf and its path; chunk
// addresses + cube words; the samples; the write-back atomics (old = atomicOr(word, fresh): the pieces of one leaf share its cubes, the
// returned word tells which cells are new for everybody); the voxel slot reservations; the voxel stores.
__global__ __launch_bounds__(VTPB) void k_voxelize(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || ctl->abortBatch) return;
	const uint32_t numItems = min(ctl->numVoxLeaves, a.voxListCap);
	if (numItems == 0u) return;
	__shared__ VoxShared sh;
	const uint32_t* list = at<const uint32_t>(a, a.offVoxList);
	const uint32_t* ptStart = at<const uint32_t>(a, a.offPtStart);
	constexpr uint32_t WPT = (CUBE_WORDS + VTPB - 1) / VTPB;        // cube words per thread
	for (uint32_t item = blockIdx.x; item < numItems; item += gridDim.x) {
		const uint32_t leafIdx = list[item] & 0x7ffffu, piece = list[item] >> 19;
		const SimlodNode* leaf = a.nodes + leafIdx;
		const uint32_t first = ptStart[leafIdx], last = leaf->numPoints;
		const uint32_t s0 = first + piece * VOX_PIECE, s1 = min(s0 + VOX_PIECE, last);
		if (s1 <= s0 || !node_is_leaf(leaf)) continue;
		const uint32_t lvl = leaf->level, LX = leaf->X, LY = leaf->Y, LZ = leaf->Z;
		const unsigned long long* rec = at<const unsigned long long>(a, a.offPaths) + (uint64_t)leafIdx * PATH_WORDS;
		PhaseTimer timer(ctl, 16);
		__syncthreads();
		if (threadIdx.x < PATH_WORDS) { sh.anc[threadIdx.x] = path_entry(a, rec, leafIdx, threadIdx.x); sh.cnt[threadIdx.x] = 0; sh.cursor[threadIdx.x] = 0; sh.base[threadIdx.x] = NONE; }
		const SimlodChunk* chunk[VOX_SPT];
		bool live[VOX_SPT];
#pragma unroll
		for (uint32_t j = 0; j < VOX_SPT; j++) {
			const uint32_t i = s0 + j * VTPB + threadIdx.x;
			live[j] = i < s1;
			chunk[j] = live[j] ? placed_chunk(a, ctl, leafIdx, i / CHUNK, first) : nullptr;
			live[j] = live[j] && chunk[j] != nullptr;
		}
		__syncthreads();
		// ancestor d (1 = parent) is sh.anc[d - 1]; a root that is still a leaf has itself as "ancestor 1" and no cube
		uint32_t depth = 0;
		while (depth < PATH_WORDS - 1 && sh.anc[depth] != 0ull) depth++;
		const bool bulk = leafIdx != 0u && last - first >= BULK_MIN;
		const uint32_t ldsDepth = bulk ? min(depth, LDS_LEVELS) : 0u;

		// the leaf's cubes, as the grids hold them now; the samples
		uint32_t sft[WPT], msk[WPT];
		float4 p[VOX_SPT];
		{
			uint32_t raw[WPT];
#pragma unroll
			for (uint32_t k = 0; k < WPT; k++) {
				uint32_t gw;
				const uint32_t d = cube_word(k * VTPB + threadIdx.x, LX, LY, LZ, gw, sft[k], msk[k]);
				if (d == 0u || d > ldsDepth) { msk[k] = 0u; raw[k] = 0u; }
				else raw[k] = path_grid(a.pers, sh.anc[d - 1])->values[gw];
			}
#pragma unroll
			for (uint32_t j = 0; j < VOX_SPT; j++)
				p[j] = live[j] ? reinterpret_cast<const float4*>(chunk[j]->points)[(s0 + j * VTPB + threadIdx.x) % CHUNK] : make_float4(0, 0, 0, 0);
			if (bulk) {
#pragma unroll
				for (uint32_t k = 0; k < WPT; k++) {
					const uint32_t w = k * VTPB + threadIdx.x;
					if (w < CUBE_WORDS) { sh.occ[w] = (raw[k] >> sft[k]) & msk[k]; sh.fresh[w] = 0u; }
				}
			}
		}
		__syncthreads();
		timer.lap(0);
		uint32_t pX[VOX_SPT], pY[VOX_SPT], pZ[VOX_SPT];
#pragma unroll
		for (uint32_t j = 0; j < VOX_SPT; j++) { pX[j] = quantize(F_FULL, p[j].x, a.minx, a.size); pY[j] = quantize(F_FULL, p[j].y, a.miny, a.size); pZ[j] = quantize(F_FULL, p[j].z, a.minz, a.size); }

		// pass A: test-and-set, bottom-up, climbing while the cell is new
#pragma unroll
		for (uint32_t j = 0; j < VOX_SPT; j++) {
			if (!live[j]) continue;
			for (uint32_t d = 1; d <= depth; d++) {
				const unsigned long long ent = sh.anc[d - 1];
				const uint32_t level = path_level(ent);
				if (level >= (uint32_t)SIMLOD_MAX_DEPTH) continue;                 // voxels.cu:449: levels 0..19 only
				const uint32_t cell = cell_of(level, pX[j], pY[j], pZ[j]);
				if (d <= ldsDepth) {
					uint32_t word, bit;
					cube_cell(d, cell, word, bit);
					if (((sh.occ[word] >> bit) & 1u) != 0u) break;
					if (((atomicOr(&sh.occ[word], 1u << bit) >> bit) & 1u) != 0u) break;
					atomicOr(&sh.fresh[word], 1u << bit);
				} else {
					uint32_t* word = &path_grid(a.pers, ent)->values[cell >> 5];
					const uint32_t bit = cell & 31u;
					if (((*word >> bit) & 1u) != 0u) break;                         // voxels.cu:93-94
					if (((atomicOr(word, 1u << bit) >> bit) & 1u) != 0u) break;     // voxels.cu:96
					// a cell that several leaves share, or a leaf with few samples: the voxel is stored right away
					const uint32_t sd = lvl - level;
					const uint32_t nodeIdx = path_node(ent);
					const uint32_t slot = atomicAdd(&a.nodes[nodeIdx].numVoxels, 1u), k = slot / CHUNK;
					if (slot % CHUNK == 0u) make_voxel_chunk(a, ctl, nodeIdx, k);
					SimlodChunk* vc = wait_voxel_chunk(a, ctl, nodeIdx, k);
					if (vc != nullptr) reinterpret_cast<float4*>(vc->points)[slot % CHUNK] = voxel_at(a, level, LX >> sd, LY >> sd, LZ >> sd, cell, p[j].w);
				}
			}
		}
		if (!bulk) { timer.lap(5); continue; }
		__syncthreads();
		timer.lap(1);

		// write-back: the grids learn the new cells and tell which of them are new for everybody; every thread's atomics in flight together
		{
			uint32_t f[WPT], old[WPT], gw[WPT];
#pragma unroll
			for (uint32_t k = 0; k < WPT; k++) {
				const uint32_t w = k * VTPB + threadIdx.x;
				const uint32_t d = cube_word(w, LX, LY, LZ, gw[k], sft[k], msk[k]);
				f[k] = (d != 0u && d <= ldsDepth) ? sh.fresh[w] : 0u;
				old[k] = f[k] != 0u ? atomicOr(&path_grid(a.pers, sh.anc[d - 1])->values[gw[k]], f[k] << sft[k]) : 0u;   // voxels.cu:96
			}
#pragma unroll
			for (uint32_t k = 0; k < WPT; k++) {
				if (f[k] == 0u) continue;
				const uint32_t w = k * VTPB + threadIdx.x;
				const uint32_t won = f[k] & ~(old[k] >> sft[k]);
				sh.fresh[w] = won;
				if (won != 0u) atomicAdd(&sh.cnt[w < 8192u ? 1u : w < 9216u ? 2u : w < 9472u ? 3u : w < 9536u ? 4u : w < 9552u ? 5u : w < 9556u ? 6u : 7u], (uint32_t)__popc(won));
			}
		}
		__syncthreads();
		timer.lap(2);
		// voxel slot ranges: one atomic per (piece, ancestor); chunks whose first slot falls into a range; then the lookups
		if (threadIdx.x >= 1u && threadIdx.x <= ldsDepth && sh.cnt[threadIdx.x] != 0u) {
			const uint32_t d = threadIdx.x, nodeIdx = path_node(sh.anc[d - 1]), cnt = sh.cnt[d];
			const uint32_t old = atomicAdd(&a.nodes[nodeIdx].numVoxels, cnt);                  // voxels.cu:101
			for (uint32_t k = (old + CHUNK - 1) / CHUNK; k * CHUNK < old + cnt; k++) make_voxel_chunk(a, ctl, nodeIdx, k);
			sh.base[d] = old;
		}
		__syncthreads();
		if (threadIdx.x >= 1u && threadIdx.x <= ldsDepth && sh.base[threadIdx.x] != NONE) {
			const uint32_t d = threadIdx.x, nodeIdx = path_node(sh.anc[d - 1]), old = sh.base[d], cnt = sh.cnt[d], k0 = old / CHUNK;
			sh.ptr[d][0] = wait_voxel_chunk(a, ctl, nodeIdx, k0);
			sh.ptr[d][1] = (old + cnt - 1) / CHUNK > k0 ? wait_voxel_chunk(a, ctl, nodeIdx, k0 + 1) : nullptr;
		}
		__syncthreads();
		timer.lap(3);

		// pass B: every cell this piece won becomes a voxel, coloured by whichever of its samples gets there first
		uint32_t levelsWithNew = 0;
		for (uint32_t d = 1; d <= ldsDepth; d++) if (sh.cnt[d] != 0u) levelsWithNew |= 1u << d;
		if (levelsWithNew == 0u) { timer.lap(4); continue; }
#pragma unroll
		for (uint32_t j = 0; j < VOX_SPT; j++) {
			if (!live[j]) continue;
			for (uint32_t left = levelsWithNew; left != 0u; left &= left - 1u) {       // only the cubes that gained cells
				const uint32_t d = (uint32_t)__ffs((int)left) - 1u;
				const unsigned long long ent = sh.anc[d - 1];
				const uint32_t level = path_level(ent);
				if (level >= (uint32_t)SIMLOD_MAX_DEPTH) continue;
				const uint32_t cell = cell_of(level, pX[j], pY[j], pZ[j]);
				uint32_t word, bit;
				cube_cell(d, cell, word, bit);
				if (((sh.fresh[word] >> bit) & 1u) == 0u) continue;
				if (((atomicAnd(&sh.fresh[word], ~(1u << bit)) >> bit) & 1u) == 0u) continue;     // somebody else took the mark
				const uint32_t base = sh.base[d], slot = base + atomicAdd(&sh.cursor[d], 1u), kk = slot / CHUNK, dk = kk - base / CHUNK;
				const uint32_t nodeIdx = path_node(ent);
				SimlodChunk* vc = dk == 0u ? sh.ptr[d][0] : dk == 1u ? sh.ptr[d][1] : wait_voxel_chunk(a, ctl, nodeIdx, kk);
				if (vc != nullptr) reinterpret_cast<float4*>(vc->points)[slot % CHUNK] = voxel_at(a, level, LX >> d, LY >> d, LZ >> d, cell, p[j].w);
			}
		}
		timer.lap(4);
	}
}

// ---- link: chunks allocated in this group hang behind their predecessors (the `next` pointers the reference sets while it walks,
// voxels.cu:517-531, 660-669) — one thread per directory slot; the leaf lists below LEAF_SLOTS are linked by k_nodes ------------------
__global__ __launch_bounds__(TPB) void k_link(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || ctl->dirUsed == 0u) return;       // also after an aborted batch: the lists must stay walkable
	const DirEntry* dir = at<const DirEntry>(a, a.offDir);
	const uint32_t tag = ctl->ordinal + 1u;
	SimlodChunk* const* leafChunks = at<SimlodChunk*>(a, a.offLeafChunks);
	for (uint32_t h = blockIdx.x * TPB + threadIdx.x; h < a.dirCap; h += gridDim.x * TPB) {
		const unsigned long long key = dir[h].key;
		if ((key >> 63) == 0ull || dir_tag(key) != tag) continue;
		const uint32_t kind = (uint32_t)(key >> 41) & 1u, node = (uint32_t)(key >> 22) & 0x7ffffu, k = (uint32_t)key & 0x3fffffu;
		if (k == 0u) continue;
		SimlodChunk* prev;
		if (kind == KIND_VOX) prev = (k - 1u) * CHUNK < at<const uint32_t>(a, a.offVoxStart)[node] ? tail_of(a.nodes[node].voxelChunks) : dir_find(a, ctl, KIND_VOX, node, k - 1u);
		else if (!node_is_leaf(a.nodes + node)) continue;         // split since: its point chunks went back to the pool, its row now lists voxel chunks
		else if (k - 1u < LEAF_SLOTS) prev = leafChunks[(uint64_t)node * LEAF_SLOTS + k - 1u];
		else prev = (k - 1u) * CHUNK < at<const uint32_t>(a, a.offPtStart)[node] ? tail_of(a.nodes[node].points) : dir_find(a, ctl, KIND_PT, node, k - 1u);
		if (prev != nullptr) prev->next = dir[h].ptr;
	}
}

// ---- nodes: per node, what the batch leaves behind — countIteration stamp (voxels.cu:298-300), numVoxelsStored (:685), the links
// of a leaf's new chunks, tail pointers, and the list lengths the next batch starts from --------------------------------------------
__global__ __launch_bounds__(TPB) void k_nodes(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	if (!ctl->active || (ctl->abortBatch && ctl->treeModified)) return;   // a cascade that gave up half way: node slots without nodes
	const uint32_t numNodes = min(ctl->abortBatch ? ctl->nodesAtStart : a.stats->numNodes, a.nodeCapacity);
	uint32_t* ptStart = at<uint32_t>(a, a.offPtStart);
	uint32_t* voxStart = at<uint32_t>(a, a.offVoxStart);
	SimlodChunk* const* leafChunks = at<SimlodChunk*>(a, a.offLeafChunks);
	uint32_t* est = at<uint32_t>(a, a.offEst);
	const uint32_t stamp = ctl->batchIndex + ctl->groupBatches;
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < numNodes; i += gridDim.x * TPB) {
		SimlodNode* node = a.nodes + i;
		node->countIteration = stamp;
		est[i] = 0;
		const uint32_t np = node->numPoints, ps = ptStart[i];
		if (np > ps) {
			const uint32_t kOld = (ps + CHUNK - 1) / CHUNK, kNew = (np + CHUNK - 1) / CHUNK;
			SimlodChunk* const* slots = leafChunks + (uint64_t)i * LEAF_SLOTS;
			for (uint32_t k = max(kOld, 1u); k < min(kNew, LEAF_SLOTS); k++) slots[k - 1]->next = slots[k];
			if (kNew > kOld) {
				SimlodChunk* t = kNew - 1u < LEAF_SLOTS ? slots[kNew - 1u] : dir_find(a, ctl, KIND_PT, i, kNew - 1u);
				if (t != nullptr) tail_of(node->points) = t;
			}
		}
		ptStart[i] = np;
		const uint32_t nv = node->numVoxels, vs = voxStart[i];
		if (nv > vs) {
			node->numVoxelsStored = nv;
			const uint32_t kOld = (vs + CHUNK - 1) / CHUNK, kNew = (nv + CHUNK - 1) / CHUNK;
			if (kNew > kOld) {
				SimlodChunk* t = dir_find(a, ctl, KIND_VOX, i, kNew - 1u);
				if (t != nullptr) tail_of(node->voxelChunks) = t;
			}
			voxStart[i] = nv;
		}
	}
}

// ---- end of group: bookkeeping (voxels.cu:535-537, 925-949), then make the next group current ------------------
__global__ void k_end(BuildArgs a, uint32_t ordinal) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	Ctl* ctl = ctl_of(a);
	if (ctl->active && ctl->abortBatch) {                    // this batch is lost: report through Stats.dbg (fatal, sticky until a reset)
		ctl->stop = 1;
		if (!ctl->treeModified) a.stats->numNodes = ctl->nodesAtStart;   // node slots reserved for splits that never began
	}
	if (ctl->active && !ctl->abortBatch) {
		if (a.stats->numAllocatedChunks > a.stats->chunkPoolSize) a.stats->chunkPoolSize = a.stats->numAllocatedChunks;
		a.stats->batchletIndex += ctl->groupBatches;
		a.stats->numPointsProcessed += ctl->groupPoints;
		ctl->consumed += ctl->groupBatches;
		ctl->pendingTotal += min(ctl->numPending, a.pendCap);
		const float elapsedMs = (float)(wall_ns() - ctl->startNs) / 1000000.0f;
		if (elapsedMs > SIMLOD_MAX_PROCESSING_MS) ctl->stop = 1;
	}
	prepare_group(a, ctl, ordinal + 1);
}

// ---- stats pass (voxels.cu:957-1009) -------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
	for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
	return v;
}

__global__ __launch_bounds__(TPB) void k_stats(BuildArgs a) {
	Ctl* ctl = ctl_of(a);
	const uint32_t numNodes = min(a.stats->numNodes, a.nodeCapacity);
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	uint32_t v[7] = {0, 0, 0, 0, 0, 0, 0};   // inner, leaves, nonempty, points, voxels, chunksP, chunksV
	if (i < numNodes) {
		const SimlodNode* n = a.nodes + i;
		if (node_is_leaf(n)) {
			v[1] = 1; v[3] = n->numPoints; v[5] = (n->numPoints + CHUNK - 1) / CHUNK;
			v[2] = n->numPoints > 0 ? 1u : 0u;
		} else {
			v[0] = 1; v[4] = n->numVoxels; v[6] = (n->numVoxels + CHUNK - 1) / CHUNK;
		}
	}
	for (int k = 0; k < 7; k++) {
		const uint32_t s = wave_sum(v[k]);
		if (lane_id() == 0 && s != 0u) atomicAdd(&ctl->statCounters[k], s);
	}
}

__global__ void k_finish(BuildArgs a, uint32_t fits) {
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	Ctl* ctl = ctl_of(a);
	SimlodStats* s = a.stats;
	if (fits) {
		s->numInner = ctl->statCounters[0];
		s->numLeaves = ctl->statCounters[1];
		s->numNonemptyLeaves = ctl->statCounters[2];
		s->numPoints = ctl->statCounters[3];
		s->numVoxels = ctl->statCounters[4];
		s->numChunksPoints = ctl->statCounters[5];
		s->numChunksVoxels = ctl->statCounters[6];
	}
	s->allocatedBytes_momentary = a.scratchBytes;
	s->allocatedBytes_persistent = reinterpret_cast<const SimlodAllocatorGlobal*>(a.pers)->offset;
	s->frameID = (uint32_t)a.frameCounter;
	s->dbg |= ctl->errors;
	if (fits && !panicked(ctl)) {               // the side tables describe THIS octree as it is after THIS batch
		ctl->tableBatch = s->batchletIndex;
		ctl->tableNodes = (uint64_t)a.nodes;
		ctl->tablePers = (uint64_t)a.pers;
		ctl->tableSig = table_signature(s);
		ctl->tableMagic = TABLE_MAGIC;
	}
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static inline uint64_t align_up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }
static constexpr uint64_t HIST_BYTES = 16ull << 20;        // 8192 leaves per round at 512 bins, 65536 at 64
static constexpr uint32_t DIR_CAP = 1u << 17;

bool layout_construct(BuildArgs& a, uint64_t capacity) {
	uint64_t off = 4096;
	a.histCap = (uint32_t)std::min<uint64_t>(SPILLING_CAPACITY, HIST_BYTES / (64 * 4));
	a.dirCap = DIR_CAP;
	a.offQueue = off;    off += align_up((uint64_t)CHUNK_QUEUE_CAPACITY * 8, 256);
	a.offSpillA = off;   off += align_up((uint64_t)a.histCap * sizeof(SpillEntry), 256);
	a.offSpillB = off;   off += align_up((uint64_t)a.histCap * sizeof(SpillEntry), 256);
	a.offSplitTag = off; off += align_up((uint64_t)a.nodeCapacity * 8, 256);
	a.offRetryTag = off; off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offEst = off;      off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offPlacedTag = off; off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offParent = off;   off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offPtStart = off;  off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offVoxStart = off; off += align_up((uint64_t)a.nodeCapacity * 4, 256);
	a.offLeafChunks = off; off += align_up((uint64_t)a.nodeCapacity * LEAF_SLOTS * 8, 256);
	a.offPaths = off;    off += align_up((uint64_t)a.nodeCapacity * PATH_WORDS * 8, 256);
	a.voxListCap = a.nodeCapacity + 8192u;                     // one piece per leaf that received samples + one per 8192 samples beyond (a group has at most 20 M)
	a.offVoxList = off;  off += align_up((uint64_t)a.voxListCap * 4, 256);
	a.offHist = off;     off += HIST_BYTES;
	a.offDir = off;      off += align_up((uint64_t)a.dirCap * sizeof(DirEntry), 256);
	// what is left is shared by the per-sample arrays: 8 B per sample of a group that may have to wait for k_place, 20 B per moved point
	const uint64_t perBatch = (uint64_t)SIMLOD_MAX_BATCH_SIZE * 8;
	const uint64_t minSpill = 100000ull * 20;                                     // two leaves' worth of moved points
	if (capacity < off + perBatch + minSpill + 4096) { a.spilledCap = 0; a.pendCap = 0; a.groupMax = 1; a.scratchBytes = off + perBatch + minSpill + 4096; return false; }
	const uint64_t left = capacity - off - 4096;
	// coalesced mode: as many batches per group as fit beside a quarter of the space kept for moved points; exact mode: one
	uint32_t groupMax = ingest_mode() ? (uint32_t)std::min<uint64_t>(SIMLOD_MAX_BATCHES_PER_LAUNCH, std::max<uint64_t>(1, (left - std::max<uint64_t>(minSpill, left / 4)) / perBatch)) : 1u;
	a.groupMax = groupMax;
	a.pendCap = groupMax * SIMLOD_MAX_BATCH_SIZE;
	a.offPendIdx = off;  off += align_up((uint64_t)a.pendCap * 4, 256);
	a.offPendLeaf = off; off += align_up((uint64_t)a.pendCap * 4, 256);
	uint64_t cap = (capacity - off - 1024) / 20;
	if (cap > 0x7fffffffull) cap = 0x7fffffffull;
	a.spilledCap = (uint32_t)cap;
	a.offSpMeta = off;   off += align_up((uint64_t)a.spilledCap * 4, 256);
	a.offSpilled = off;  off += (uint64_t)a.spilledCap * 16;
	a.scratchBytes = off;
	return off <= capacity;
}

static int launch_expand(const BuildArgs& a, uint32_t roundFirst, uint32_t wgs, bool cooperative, hipStream_t stream) {
	if (profile_enabled()) profile_mark(roundFirst ? "k_expand_more" : "k_expand", stream);
	if (!cooperative) { hipLaunchKernelGGL(k_expand, dim3(wgs), dim3(ETPB), 0, stream, a, roundFirst); return (int)hipGetLastError(); }
	BuildArgs args = a;
	void* params[] = {&args, &roundFirst};
	return (int)hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_expand), dim3(wgs), dim3(ETPB), params, 0, stream);
}

int launch_construct(const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream) {
	BuildArgs a{};
	a.ring = points; a.mom = reinterpret_cast<uint8_t*>(buffer); a.pers = pers; a.nodes = nodes; a.stats = stats;
	a.frameStart = frameStart; a.numBatchesUploaded = numBatchesUploaded; a.batchSizes = batchSizes;
	const float bx = u->boxMax.x - u->boxMin.x, by = u->boxMax.y - u->boxMin.y, bz = u->boxMax.z - u->boxMin.z;
	a.size = fmaxf(fmaxf(bx, by), bz);                                         // voxels.cu:860-863
	a.minx = u->boxMin.x; a.miny = u->boxMin.y; a.minz = u->boxMin.z;
	a.persCapacity = u->persistentBufferCapacity;
	a.frameCounter = u->frameCounter;
	a.nodeCapacity = node_capacity();
	const bool fits = layout_construct(a, u->momentaryBufferCapacity);
	if (fits) {   // the rasteriser reads leaf lists through the table while its stamp matches the octree (render.hip r_visible)
		const Ctl* ctl = reinterpret_cast<const Ctl*>(a.mom);
		note_leaf_table(LeafTableRef{nodes, a.mom, reinterpret_cast<const SimlodChunk* const*>(a.mom + a.offLeafChunks), &ctl->tableMagic, &ctl->tableBatch,
		                             &ctl->tableNodes, &ctl->tableSig, TABLE_MAGIC, LEAF_SLOTS});
	} else forget_leaf_table(nodes);
	const DeviceInfo& dev = device_info();
	const uint32_t coalesce = ingest_mode();
	const uint32_t limit = std::min<uint32_t>(batch_limit(), SIMLOD_MAX_BATCHES_PER_LAUNCH);
	const uint32_t debugFlags = ((uint32_t)tune("SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT", 0) & 1u) | (tune("SIMLOD_PHASE_TIMERS", 0) ? 2u : 0u);

	SIMLOD_LAUNCH(k_begin, dim3(1), dim3(64), stream, a, fits ? 0u : 1u, coalesce, limit, debugFlags);
	if (fits) {
		hipError_t e = hipMemsetAsync(a.mom + a.offSplitTag, 0, (size_t)(a.offParent - a.offSplitTag), stream);   // split records, retry tags, arrival estimates
		if (e == hipSuccess) e = hipMemsetAsync(a.mom + a.offHist, 0, (size_t)(HIST_BYTES + (uint64_t)a.dirCap * sizeof(DirEntry)), stream);
		if (e != hipSuccess) return (int)e;
		const uint32_t gridNodes = (a.nodeCapacity + TPB - 1) / TPB;
		SIMLOD_LAUNCH(k_parents, dim3(gridNodes), dim3(TPB), stream, a);
		SIMLOD_LAUNCH(k_paths, dim3(gridNodes), dim3(TPB), stream, a);
		const uint32_t gridPoints = dev.numCUs * (uint32_t)tune("SIMLOD_GRID_MULT", 8);
		// k_expand's workgroups meet at grid barriers: never more than one per CU (all must be resident).  One per TWO CUs is the
		// measured optimum on MI355X: the barrier's agent-scope release / acquire and the polling cost grow with the participants
		// (one batch at a time; a coalesced group scans tens of millions of waiting samples per round: every CU takes part)
		const uint32_t expandWgs = (uint32_t)std::max(1, std::min(tune("SIMLOD_EXPAND_WGS", coalesce ? (int)dev.numCUs : (int)dev.numCUs / 2), (int)dev.numCUs));
		const bool cooperative = tune("SIMLOD_EXPAND_COOPERATIVE", 0) != 0;
		const bool peek = tune("SIMLOD_PEEK", coalesce ? 0 : 1) != 0;    // the forecast pays when batches come one at a time (measured: coalesced 6.49 vs 6.60 ms without)
		const uint32_t groups = coalesce ? (limit + a.groupMax - 1) / a.groupMax : limit;
		for (uint32_t g = 0; g < groups; g++) {
			if (peek) SIMLOD_LAUNCH(k_peek, dim3(dev.numCUs * 2), dim3(TPB), stream, a);
			SIMLOD_LAUNCH(k_ingest<4>, dim3(gridPoints), dim3(TPB), stream, a);
			int rc = launch_expand(a, 0u, expandWgs, cooperative, stream);
			if (rc == 0) rc = launch_expand(a, 1u, expandWgs, cooperative, stream);
			if (rc != 0) return rc;
			SIMLOD_LAUNCH(k_prealloc, dim3(dev.numCUs), dim3(TPB), stream, a);
			SIMLOD_LAUNCH(k_place<4>, dim3(gridPoints), dim3(TPB), stream, a);
			SIMLOD_LAUNCH(k_voxelize, dim3(dev.numCUs * 2), dim3(VTPB), stream, a);
			SIMLOD_LAUNCH(k_link, dim3(std::min<uint32_t>(dev.numCUs, a.dirCap / TPB)), dim3(TPB), stream, a);
			SIMLOD_LAUNCH(k_nodes, dim3(std::min(gridNodes, dev.numCUs)), dim3(TPB), stream, a);
			SIMLOD_LAUNCH(k_end, dim3(1), dim3(64), stream, a, g);
		}
		SIMLOD_LAUNCH(k_stats, dim3(gridNodes), dim3(TPB), stream, a);
	}
	SIMLOD_LAUNCH(k_finish, dim3(1), dim3(64), stream, a, fits ? 1u : 0u);
	if (profile_enabled()) profile_close(stream);
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) return (int)e;
	return fits ? 0 : (int)hipErrorInvalidValue;
}

uint64_t construct_min_bytes() {
	BuildArgs a{};
	a.nodeCapacity = node_capacity();
	layout_construct(a, 0);
	return a.scratchBytes + 4096;
}

}  // namespace bulk
}  // namespace simlod
