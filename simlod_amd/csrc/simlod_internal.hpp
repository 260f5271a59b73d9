// simlod_internal.hpp — declarations shared by the translation units of libsimlod_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "simlod_abi.h"

namespace simlod {

static constexpr uint32_t CHUNK_QUEUE_CAPACITY = 1000000u;   // progressive_octree_voxels.cu:856
static constexpr uint32_t SPILLING_CAPACITY = 100000u;       // progressive_octree_voxels.cu:847

namespace bulk {   // construct_bulk.hip: the chain of the coalesced ingest mode

// One entry of a split round's work list: the leaf that has to split, the eight node slots reserved for its children and — for
// leaves that hold stored points (round 0 only: nodes created inside a cascade are empty) — where those points go in the spill buffer.
struct SpillEntry {
	uint32_t leaf, childBase, spillBase, stored;
};

// Control block at byte 0 of kernel_construct's momentary buffer.  Lives only for the duration of one launch
// (the recycle stack behind it, like the reference's chunkQueue, must survive between launches; so does the leaf chunk table).
struct Ctl {
	uint32_t uploaded, firstBatch, numBatches, stop;      // launch: snapshot of the upload counter, first batch, batches to take
	uint32_t consumed;           // batches of this launch already ingested
	uint32_t active;             // the current group exists
	uint32_t ordinal;            // group number inside this launch (tags)
	uint32_t batchIndex;         // Stats.batchletIndex of the group's first batch
	uint32_t groupBatches;       // 1 in exact mode, up to 20 in coalesced mode
	uint32_t groupPoints;
	uint32_t numPending;         // samples waiting for the place pass (their leaf overflowed, or is the root, or the LDS table was full)
	uint32_t numSpilling;        // round-0 list length (written by k_ingest, never modified by k_expand: stable early-exit test)
	uint32_t dirUsed;            // chunks published in the hash directory by this group
	uint32_t numVoxLeaves;       // leaves that k_place stored samples in: k_voxelize's work list
	uint32_t errors, abortBatch, panic;
	uint32_t barrierCount[2];    // one monotonic counter per k_expand launch of a group (round 0 | the later rounds)
	uint32_t rebuildLeafChunks;  // this launch found the leaf chunk table stale (first launch, reset, wiped momentary buffer)
	uint32_t roundSpill[2];      // list length of rounds >= 1: round r appends to roundSpill[r & 1]
	uint32_t coalesce, debugFlags;
	uint32_t nodesAtStart;       // Stats.numNodes when the group began
	uint32_t treeModified;       // k_expand has started to build nodes in this group
	uint64_t startNs;
	uint32_t statCounters[8];
	unsigned long long reserve;  // nodes in use << 32 | spill space in use — ONE word, so a split reserves both or neither
	uint32_t tableMagic, tableBatch;   // leaf chunk table is valid for the octree as it was after batch #tableBatch (k_finish) ...
	uint64_t tableNodes, tablePers;    // ... of THIS octree (node array, persistent buffer)
	uint64_t spilledTotal;       // byte 176: measurement aid — stored points moved by splits since the host last cleared it (bench.py)
	uint64_t pendingTotal;       // byte 184: samples that went through k_place (their leaf overflowed) ...
	uint64_t placeVoxels;        // byte 192: ... and the voxels k_place created, since the host last cleared them
	uint64_t expandNs[8];        // byte 200: k_expand phase times of workgroup 0 (hist, barrier, decide, barrier; rounds; calls)
	uint32_t batchSize[SIMLOD_MAX_BATCHES_PER_LAUNCH], batchSlot[SIMLOD_MAX_BATCHES_PER_LAUNCH];
	uint64_t tableSig;           // table_signature() of the Stats the table belongs to (k_finish)
	uint64_t phaseNs[24];        // SIMLOD_PHASE_TIMERS=1 — wall time per phase summed over workgroups: k_ingest [0..7], k_place [8..15], k_voxelize [16..23]
};
static_assert(offsetof(Ctl, spilledTotal) == 176, "bench.py reads Ctl.spilledTotal at byte 176");
static_assert(offsetof(Ctl, phaseNs) == 432, "tools/phases.py reads Ctl.phaseNs at byte 432");
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
	uint64_t     offQueue, offSpillA, offSpillB, offSplitTag, offRetryTag, offEst, offPlacedTag, offParent, offPtStart, offVoxStart, offLeafChunks, offPaths, offHist, offDir,
	             offPendIdx, offPendLeaf, offSpMeta, offSpilled, offVoxList;
	uint32_t     nodeCapacity, spilledCap, pendCap, histCap, dirCap, groupMax, voxListCap;
};

bool layout_construct(BuildArgs& a, uint64_t capacity);
int launch_construct(const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream);
uint64_t construct_min_bytes();

}  // namespace bulk

namespace batch {  // construct_batch.hip: exact mode, one ring batch at a time

int launch_construct(const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream);
uint64_t construct_min_bytes();

}  // namespace batch

// The builder's leaf chunk table, as the rasteriser may use it (render.hip r_visible): row i holds the first chunks of node i's list in order (a leaf: points; an inner node: voxels).
// The three stamp words live in the builder's control block on the device; the table describes the octree `nodes` as it is NOW only
// while *magic == magicValue, *batch == Stats.batchletIndex and *tableNodes == nodes and *sig == table_signature(Stats) (the builder clears the stamp while it works and
// re-stamps in k_finish), which r_visible checks on the device every frame.
struct LeafTableRef {
	const void*               nodes;
	const void*               block;       // start of the buffer the table lives in (the construct kernel's momentary buffer)
	const SimlodChunk* const* table;
	const uint32_t*           magic;
	const uint32_t*           batch;
	const uint64_t*           tableNodes;
	const uint64_t*           sig;         // table_signature() of the Stats the table was stamped for
	uint32_t                  magicValue, slots;
};
// what the stamp remembers of the Stats block: an octree image that reached the buffers some other way (a host upload) differs here
__host__ __device__ inline uint64_t table_signature(const SimlodStats* s) {
	return ((uint64_t)s->numNodes | (uint64_t)s->numPoints << 32) ^ (s->allocatedBytes_persistent * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)s->numVoxels << 20);
}
void note_leaf_table(const LeafTableRef& ref);                 // construct chains: after a launch whose layout fits
void forget_leaf_table(const void* nodes);                     // reset
bool find_leaf_table(const void* nodes, LeafTableRef& ref);    // false also when the table's buffer is no longer a live device allocation

// How many per-batch kernel groups a kernel_construct launch should enqueue (the host cannot see how many batches are pending: the
// upload counter lives on the device).  Every launch ends with two 4-byte copies — Stats.batchletIndex and the upload counter — into
// page-locked host memory; the next launch reads whatever has arrived (no synchronisation) and enqueues what was left pending + what
// the recent launches processed + 2, at least 2, at most 20.  Unknown octree, or just reset: 20.  An idle frame loop pays for 2
// groups instead of 20 (0.84 ms -> 0.1 ms per launch on MI355X, tools/idle_launch.py); a burst is picked up one launch late.
uint32_t groups_for_launch(const SimlodStats* stats);
int note_launch_end(const SimlodStats* stats, const uint32_t* numBatchesUploaded, hipStream_t stream);
void forget_launch_history(const SimlodStats* stats);

struct DeviceInfo {
	int      device;
	uint32_t numCUs;
};

const DeviceInfo& device_info();

// Optional per-kernel timing with HIP events on the launch stream (simlod_profile_* in simlod_hip.h).  Disabled by
// default: LAUNCH() then is a bare hipLaunchKernelGGL.
bool profile_enabled();
void profile_mark(const char* kernelName, hipStream_t stream);   // records "kernelName starts now"
void profile_close(hipStream_t stream);                           // records the end of the last kernel of a call

#define SIMLOD_LAUNCH(kernel, grid, block, stream, ...)                           \
	do {                                                                          \
		if (::simlod::profile_enabled()) ::simlod::profile_mark(#kernel, stream); \
		hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);          \
		if (::simlod::debug_sync()) ::simlod::debug_synced(#kernel);              \
	} while (0)
bool debug_sync();                           // SIMLOD_DEBUG_SYNC=1: synchronise the device after every kernel and name it on stderr (fault hunting)
void debug_synced(const char* kernelName);
uint32_t node_capacity();
uint32_t ingest_mode();                      // 0 = exact (one batch at a time, the reference's granularity), 1 = coalesced
uint32_t batch_limit();                      // host hint: at most this many batches are pending (<= 20)
int tune(const char* envName, int dflt);     // integer tuning knob from the environment (read once per call site)

int launch_reset(const SimlodUniforms* u, uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                 uint32_t* batchSizes, hipStream_t stream);
int launch_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format, const double* scale,
                      const double* offset, SimlodPoint* out, hipStream_t stream);
int launch_colorfilter(const SimlodUniforms* u, uint32_t* buffer, SimlodNode* nodes, const uint32_t* numNodes, SimlodStats* stats, hipStream_t stream);
uint64_t colorfilter_min_bytes(uint32_t nodeCapacity);
int launch_generate_terrain(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed, uint32_t tilesX,
                            const float tileExtent[3], float swathWidth, hipStream_t stream);
enum : uint32_t { RENDER_FIRST = 1u, RENDER_COLOR = 2u, RENDER_RESOLVE = 4u, RENDER_OUTPUT = 8u, RENDER_ALL = 15u };
int launch_render(uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                  uint64_t* frameStart, hipStream_t stream, uint32_t parts);
uint64_t render_depth_plane_offset(uint32_t width, uint32_t height);
uint64_t render_sum_planes_offset(uint32_t width, uint32_t height);

}  // namespace simlod
