// simlod_internal.hpp — declarations shared by the translation units of libsimlod_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <climits>
#include <mutex>
#include <vector>

#include "simlod_abi.h"

namespace simlod {

static constexpr uint32_t CHUNK_QUEUE_CAPACITY = 1000000u;   // progressive_octree_voxels.cu:856
static constexpr uint32_t SPILLING_CAPACITY = 100000u;       // progressive_octree_voxels.cu:847

// The builder's leaf chunk table, as the rasteriser may use it (render.hip r_visible): row i holds the first chunks of node i's list in order (a leaf: points; an inner node: voxels).
// The three stamp words live in the builder's control block on the device; the table describes the octree `nodes` as it is NOW only
// while *magic == magicValue, *batch == Stats.batchletIndex and *tableNodes == nodes and *sig == table_signature(Stats) (the builder clears the stamp while it works and
// re-stamps in k_finish), which r_visible checks on the device every frame.
// A row is 256 bytes: LEAF_ROW_SLOTS chunk addresses as offsets into the persistent buffer in units of 16 bytes (every allocation there is a multiple of 16
// from a base that is: utils.h.cu:185-197), 40 bits each — the low words as 50 x u32, the high bytes as 50 x u8 behind them; 0 = no chunk (offset 0 is the
// allocator's own header).  8-byte pointers cost 105 MB of the momentary buffer for the reference host's 263 157 nodes, this 67 MB: the difference is
// what lets EXACT mode ingest a launch's batches in groups inside the reference host's 300 MB (construct.hip account_group).
static constexpr uint32_t LEAF_ROW_SLOTS = SIMLOD_MAX_POINTS_PER_NODE / SIMLOD_POINTS_PER_CHUNK, LEAF_ROW_BYTES = 256;
static_assert(LEAF_ROW_SLOTS * 5u <= LEAF_ROW_BYTES && LEAF_ROW_SLOTS * 4u % 4u == 0u, "leaf chunk table row");
__host__ __device__ inline const SimlodChunk* leaf_row_get(const uint8_t* table, const uint8_t* pers, uint64_t row, uint32_t k) {
	const uint8_t* r = table + row * LEAF_ROW_BYTES;
	const uint64_t v = (uint64_t)reinterpret_cast<const uint32_t*>(r)[k] | (uint64_t)r[LEAF_ROW_SLOTS * 4u + k] << 32;
	return v != 0ull ? reinterpret_cast<const SimlodChunk*>(pers + (v << 4)) : nullptr;
}
__host__ __device__ inline void leaf_row_set(uint8_t* table, const uint8_t* pers, uint64_t row, uint32_t k, const SimlodChunk* c) {
	uint8_t* r = table + row * LEAF_ROW_BYTES;
	const uint64_t v = c != nullptr ? (uint64_t)(reinterpret_cast<const uint8_t*>(c) - pers) >> 4 : 0ull;
	reinterpret_cast<uint32_t*>(r)[k] = (uint32_t)v;
	r[LEAF_ROW_SLOTS * 4u + k] = (uint8_t)(v >> 32);
}
struct LeafTableRef {
	const void*               nodes;
	const void*               block;       // start of the buffer the table lives in (the construct kernel's momentary buffer)
	const uint8_t*            table;       // rows of LEAF_ROW_BYTES (leaf_row_get)
	const uint8_t*            pers;        // the persistent buffer the rows' offsets refer to
	const uint32_t*           magic;
	const uint32_t*           batch;
	const uint64_t*           tableNodes;
	const uint64_t*           sig;         // table_signature() of the Stats the table was stamped for
	uint32_t                  magicValue, slots;
	uint32_t                  rows;        // rows the table has (the node capacity of the launch that registered it)
};
// what the stamp remembers of the Stats block: an octree image that reached the buffers some other way (a host upload) differs here
__host__ __device__ inline uint64_t table_signature(const SimlodStats* s) {
	return ((uint64_t)s->numNodes | (uint64_t)s->numPoints << 32) ^ (s->allocatedBytes_persistent * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)s->numVoxels << 20);
}
// ---- per-octree state (include/simlod_hip.h, simlod_context_*) -----------------------------------------------------------------------------
// Everything the library keeps between launches belongs to a context: the ingest mode, the node capacity, the host's batch limit, the
// tuning knobs (read from the environment ONCE, when the context is made; simlod_context_set_knob overrides one), the second stream and
// its events, the registry of leaf chunk tables, the launch feedback.  The reference's launch signatures carry no handle, so a launch
// finds its context through the node array it is given (simlod_context_attach); node arrays nobody attached share the default context.
enum Knob : int {
	KNOB_OVERLAP_TAIL, KNOB_EXPAND_WGS, KNOB_GRID_MULT, KNOB_COUNT_TPB, KNOB_VOXELIZE_WGS, KNOB_ADAPTIVE_GROUPS,
	KNOB_RASTER_LEAF_TABLE, KNOB_RASTER_LDS_TILES, KNOB_DRAW_MULT, KNOB_RASTER_FUSED_RESOLVE,
	KNOB_DEBUG_FORCE_BARRIER_TIMEOUT, KNOB_DEBUG_VOXELIZE_CLOCK, KNOB_DEBUG_BUDGET_US, KNOB_GROUP_BATCHES, KNOB_DEBUG_PHASE_WG, KNOB_EVENT_SYSTEM_FENCE, KNOB_RASTER_SCREEN_BINS, KNOB_DEBUG_BIN_POOL, KNOB_EXACT_GROUP, KNOB_DEBUG_IRREGULAR_CHILDREN, KNOB_COUNT_
};
static constexpr int KNOB_UNSET = INT_MIN;
extern const char* const KNOB_NAMES[KNOB_COUNT_];            // "SIMLOD_OVERLAP_TAIL", ...

struct LaunchHistory {      // seen (page-locked, written by the launches' last kernels and by k_reset): [0] batchletIndex, [1] the upload counter, [2] fit for exact groups, [3] sequence number of the launch that wrote them
	const void* stats; volatile uint32_t* seen; uint32_t prevIndex, prevUploaded, arrivals; bool havePrev;
	uint32_t seq, resetSeq; bool resetKnown;      // launches (and resets) of this octree so far | the latest reset's number | ... and it ran through this library (index 0 from there on)
	uint32_t enq[32];                             // batches launch #seq was sized for
};
struct LaunchPlan { uint32_t batches; bool mayGroup; uint32_t* feedback; uint32_t seq; };   // batches this launch can find (0: none — an idle frame) | exact groups allowed | where its last kernel reports, and as which launch
struct FrameFeedback { const void* buffer; volatile uint32_t* seen; bool bins; uint64_t bytes; bool possible, open; };                       // render.hip launch_render: seen[0] = nodes of the buffer's latest frame that sort (or would)
struct SideStream;                                           // construct.hip: the second stream of kernel_construct and its events
void destroy_side_stream(SideStream* s);

struct Context {
	std::atomic<uint32_t> nodeCapacity{263157u};             // 40 000 000 B / 152 B, main_progressive_octree.cpp:552
	std::atomic<uint32_t> ingestMode{0u};                    // 0 = exact (one batch at a time, the reference's granularity), 1 = coalesced
	std::atomic<uint32_t> batchLimit{SIMLOD_MAX_BATCHES_PER_LAUNCH};   // simlod_context_set_construct_batch_limit: a launch never takes more than this many batches (<= 20)
	std::atomic<bool>     sideTablesStale{false};                       // simlod_octree_image_replaced / a reset: the next kernel_construct rebuilds its side tables whatever the stamp in the buffer says (an uploaded image with the same counters as the one it replaces: ADVICE r5)
	std::atomic<int>      hintPending{-1};                              // simlod_context_hint_pending_batches: that many batches are pending NOW — for the next launch, then forgotten
	std::atomic<uint64_t> trunkLo{0u}, trunkHi{0u};          // simlod_context_set_trunk_mask: upper nodes (levels 0-2) that split whatever they hold; zero: the reference's rule alone
	int knob[KNOB_COUNT_];
	std::mutex sideLock;
	SideStream* side[64] = {};                               // per device ordinal, made by the first launch that wants it
	std::mutex tablesLock;
	std::vector<LeafTableRef> tables;
	std::mutex historyLock;
	std::vector<LaunchHistory> history;
	std::mutex framesLock;
	std::vector<FrameFeedback> frames;
	hipEvent_t gateEvent[64] = {};                            // per device ordinal: the end of this context's latest k_expand, when it runs without a second stream (expand_gate)
	Context();
	~Context();
	void reload_env();
	int tune(Knob k, int dflt) const { return knob[k] == KNOB_UNSET ? dflt : knob[k]; }
};
// The frame's feedback word (page-locked, written by the frame's first draw pass) of the render buffer, and whether this frame sorts the
// samples of its large nodes into the screen bins: decided by the frame's first part from what the buffer's previous frame found, kept for
// the frame's other parts.  nullptr (no page-locked memory): every frame sorts.
uint32_t* frame_feedback(Context& ctx, const void* buffer, uint32_t parts, bool& possible, bool& bins, uint64_t& bufferBytes);   // parts: RENDER_* of this launch; bufferBytes: what the allocation behind `buffer` holds from `buffer` on
void frame_feedback_no_bins(Context& ctx, const void* buffer);
Context& context_of(const void* nodes);                      // the context `nodes` is attached to, else the default one
uint32_t live_contexts();                                    // contexts that exist right now (the default one included once it has been used)

// k_expand's workgroups meet at a hand-rolled grid barrier and must all be resident.  Two such kernels of two contexts, running at once on
// one device, can each hold part of the CUs and wait for workgroups of their own that the other keeps from becoming resident.  So, while more
// than one context is alive, the k_expand launches of a device form ONE chain: a launch waits for the end of the latest k_expand of any OTHER
// context (its own are ordered by its stream).  One context — the reference host's case — pays nothing.
bool expand_gate_enter(Context& ctx, hipStream_t stream);                      // before the launch: wait for the other contexts' latest k_expand; true: the gate is held until expand_gate_leave
void expand_gate_leave(Context& ctx, hipStream_t stream, hipEvent_t ended, bool held);   // after it: `ended` is signalled by its end (nullptr: an event of the context is recorded behind it)
void expand_gate_forget(Context& ctx);                                         // the context goes away

namespace build {  // construct.hip: kernel_construct — one ring batch at a time (exact mode) or groups of pending batches (coalesced mode)

int launch_construct(Context& ctx, const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream);
uint64_t construct_min_bytes(uint32_t nodeCapacity);

}  // namespace build

void note_leaf_table(Context& ctx, const LeafTableRef& ref);                 // kernel_construct: after a launch whose layout fits
void forget_leaf_table(Context& ctx, const void* nodes);                     // reset
bool find_leaf_table(Context& ctx, const void* nodes, LeafTableRef& ref);    // false also when the table's buffer is no longer a live device allocation

// How many batches a kernel_construct launch should enqueue kernels for (simlod_hip.cpp: launch sizing).
LaunchPlan launch_plan(Context& ctx, const SimlodStats* stats, const void* uploadCounter);
void forget_launch_history(Context& ctx, const SimlodStats* stats, const void* uploadCounter = nullptr, uint32_t** words = nullptr, uint32_t* seq = nullptr);   // reset (words: where k_reset reports, as launch *seq)
void note_upload_counter(const void* counter, uint32_t value, bool written, bool create);     // written: the host has enqueued a write of `value` to the upload counter at `counter`; else: `counter` is named as one

struct DeviceInfo {
	int      device;
	uint32_t numCUs;
};

const DeviceInfo& device_info();

// Optional per-kernel timing with HIP events on the launch stream (simlod_profile_* in simlod_hip.h).  Disabled by
// default: LAUNCH() then is a bare hipLaunchKernelGGL.
bool profile_enabled();
bool profile_dominant();                      // simlod_profile_enable(2): events around k_voxelize alone, on ITS stream — the two-stream pipeline stays as it is in production
void profile_mark(const char* kernelName, hipStream_t stream);   // records "kernelName starts now"
void profile_close(hipStream_t stream);                           // records the end of the last kernel of a call
void profile_kernel_events(const char* kernelName, hipEvent_t* start, hipEvent_t* stop);   // events for a launch's own start / stop slots (hipExtLaunchKernelGGL)

#define SIMLOD_LAUNCH(kernel, grid, block, stream, ...)                           \
	do {                                                                          \
		if (::simlod::profile_enabled()) ::simlod::profile_mark(#kernel, stream); \
		hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);          \
		if (::simlod::debug_sync()) ::simlod::debug_synced(#kernel);              \
	} while (0)
// a launch whose completion signals `stopEvent` (hipExtLaunchKernelGGL): for kernels another stream waits for
#define SIMLOD_LAUNCH_STOP(kernel, grid, block, stream, stopEvent, ...)                       \
	do {                                                                                      \
		hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, nullptr, stopEvent, 0, __VA_ARGS__); \
		if (::simlod::debug_sync()) ::simlod::debug_synced(#kernel);                          \
	} while (0)
bool debug_sync();                           // SIMLOD_DEBUG_SYNC=1: synchronise the device after every kernel and name it on stderr (fault hunting)
void debug_synced(const char* kernelName);

int launch_reset(Context& ctx, const SimlodUniforms* u, uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                 uint32_t* batchSizes, hipStream_t stream);
int launch_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format, const double* scale,
                      const double* offset, SimlodPoint* out, hipStream_t stream);
int launch_colorfilter(Context& ctx, const SimlodUniforms* u, uint32_t* buffer, SimlodNode* nodes, const uint32_t* numNodes, SimlodStats* stats, hipStream_t stream);
uint64_t colorfilter_min_bytes(uint32_t nodeCapacity);
int launch_generate_terrain(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed, uint32_t tilesX,
                            const float tileExtent[3], float swathWidth, hipStream_t stream);
enum : uint32_t { RENDER_FIRST = 1u, RENDER_COLOR = 2u, RENDER_RESOLVE = 4u, RENDER_OUTPUT = 8u, RENDER_ALL = 15u };
int launch_render(Context& ctx, uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                  uint64_t* frameStart, hipStream_t stream, uint32_t parts);
uint64_t render_depth_plane_offset(uint32_t width, uint32_t height);
uint64_t render_sum_planes_offset(uint32_t width, uint32_t height);

}  // namespace simlod
