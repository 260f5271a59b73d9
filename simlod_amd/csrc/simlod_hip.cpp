// simlod_hip.cpp — the C ABI of libsimlod_hip.so (include/simlod_hip.h): typed launches plus the
// CudaModularProgram / cuLaunchCooperativeKernel shaped surface of the reference host
// (include/CudaModularProgram.h:140-264, modules/progressive_octree/main_progressive_octree.cpp:333-546).
#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {

const char* const KNOB_NAMES[KNOB_COUNT_] = {
	"SIMLOD_OVERLAP_TAIL", "SIMLOD_EXPAND_WGS", "SIMLOD_GRID_MULT", "SIMLOD_COUNT_TPB", "SIMLOD_VOXELIZE_WGS", "SIMLOD_ADAPTIVE_GROUPS",
	"SIMLOD_RASTER_LEAF_TABLE", "SIMLOD_RASTER_LDS_TILES", "SIMLOD_DRAW_MULT", "SIMLOD_RASTER_FUSED_RESOLVE",
	"SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT", "SIMLOD_DEBUG_VOXELIZE_CLOCK", "SIMLOD_DEBUG_BUDGET_US", "SIMLOD_GROUP_BATCHES", "SIMLOD_DEBUG_PHASE_WG", "SIMLOD_EVENT_SYSTEM_FENCE", "SIMLOD_RASTER_SCREEN_BINS", "SIMLOD_DEBUG_BIN_POOL", "SIMLOD_EXACT_GROUP", "SIMLOD_DEBUG_IRREGULAR_CHILDREN",
};

static std::atomic<uint32_t> g_liveContexts{0};
uint32_t live_contexts() { return g_liveContexts.load(); }

Context::Context() { reload_env(); g_liveContexts.fetch_add(1); }

// ---- one chain of k_expand launches per device while several contexts are alive (simlod_internal.hpp) ----------------------------
namespace {
struct ExpandGate { std::mutex lock; hipEvent_t last = nullptr; const Context* owner = nullptr; };
ExpandGate g_gate[64];
int gate_device() { int dev = 0; (void)hipGetDevice(&dev); return dev < 0 || dev >= 64 ? 0 : dev; }
}  // namespace

bool expand_gate_enter(Context& ctx, hipStream_t stream) {
	ExpandGate& g = g_gate[gate_device()];
	g.lock.lock();                                            // held until expand_gate_leave: the launch and the gate's new state change hands together
	if (g.last != nullptr && g.owner != &ctx) (void)hipStreamWaitEvent(stream, g.last, 0);
	return true;
}

void expand_gate_leave(Context& ctx, hipStream_t stream, hipEvent_t ended, bool held) {
	if (!held) return;
	const int dev = gate_device();
	ExpandGate& g = g_gate[dev];
	if (ended == nullptr) {           // one stream: an event of the context behind the kernel — also while this is the only context (~1 us): one made
	                                  // a moment later must find the k_expand that is still in flight
		if (ctx.gateEvent[dev] == nullptr && hipEventCreateWithFlags(&ctx.gateEvent[dev], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx.gateEvent[dev] = nullptr; }
		ended = ctx.gateEvent[dev];
		if (ended != nullptr) (void)hipEventRecord(ended, stream);
	}
	if (ended != nullptr) { g.last = ended; g.owner = &ctx; }
	g.lock.unlock();
}

void expand_gate_forget(Context& ctx) {
	for (ExpandGate& g : g_gate) {
		std::lock_guard<std::mutex> hold(g.lock);
		if (g.owner == &ctx) { g.last = nullptr; g.owner = nullptr; }
	}
}

void Context::reload_env() {
	for (int k = 0; k < KNOB_COUNT_; k++) {
		const char* v = std::getenv(KNOB_NAMES[k]);
		knob[k] = v != nullptr && *v != 0 ? std::atoi(v) : KNOB_UNSET;
	}
}

Context::~Context() {
	expand_gate_forget(*this);
	g_liveContexts.fetch_sub(1);
	for (SideStream*& s : side) { if (s != nullptr) destroy_side_stream(s); s = nullptr; }
	// the page-locked feedback words: a copy enqueued by the context's last launches (on the CALLER's streams) may still be on its way
	(void)hipDeviceSynchronize();
	for (LaunchHistory& h : history) (void)hipHostFree(const_cast<uint32_t*>(h.seen));
	for (FrameFeedback& f : frames) (void)hipHostFree(const_cast<uint32_t*>(f.seen));
	for (hipEvent_t& e : gateEvent) if (e != nullptr) { (void)hipEventDestroy(e); e = nullptr; }
	(void)hipGetLastError();
}

// node array -> context (simlod_context_attach); everything else runs in the default context
static std::mutex g_attachLock;
static std::vector<std::pair<const void*, Context*>> g_attached;
static Context& default_context() { static Context* c = new Context(); return *c; }   // (never destroyed: launches may race with process exit)

Context& context_of(const void* nodes) {
	std::lock_guard<std::mutex> hold(g_attachLock);
	for (auto& a : g_attached) if (a.first == nodes) return *a.second;
	return default_context();
}

bool debug_sync() {
	static const bool on = [] { const char* v = std::getenv("SIMLOD_DEBUG_SYNC"); return v != nullptr && std::atoi(v) != 0; }();
	return on;
}
void debug_synced(const char* kernelName) {
	const hipError_t e = hipDeviceSynchronize();
	std::fprintf(stderr, "[simlod] %s -> %d\n", kernelName, (int)e);
	std::fflush(stderr);
}

const DeviceInfo& device_info() {
	// one entry per device ordinal; a process drives one GPU (one rank per GPU), but stay correct if it switches
	static DeviceInfo cache[64];
	static std::atomic<bool> valid[64];
	int dev = 0;
	(void)hipGetDevice(&dev);
	if (dev < 0 || dev >= 64) dev = 0;
	if (!valid[dev].load(std::memory_order_acquire)) {
		hipDeviceProp_t prop;
		DeviceInfo info{dev, 256u};
		if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) info.numCUs = (uint32_t)prop.multiProcessorCount;
		cache[dev] = info;
		valid[dev].store(true, std::memory_order_release);
	}
	return cache[dev];
}

uint64_t render_buffer_bytes(uint32_t width, uint32_t height);

// ---- builder -> rasteriser: where each octree's leaf chunk table lives (simlod_internal.hpp LeafTableRef) --------
void note_leaf_table(Context& ctx, const LeafTableRef& ref) {
	std::lock_guard<std::mutex> hold(ctx.tablesLock);
	std::vector<LeafTableRef>& g_leafTables = ctx.tables;
	for (LeafTableRef& r : g_leafTables)
		if (r.nodes == ref.nodes) { r = ref; return; }
	if (g_leafTables.size() >= 64) g_leafTables.erase(g_leafTables.begin());
	g_leafTables.push_back(ref);
}

void forget_leaf_table(Context& ctx, const void* nodes) {
	std::lock_guard<std::mutex> hold(ctx.tablesLock);
	std::vector<LeafTableRef>& g_leafTables = ctx.tables;
	for (size_t i = 0; i < g_leafTables.size(); i++)
		if (g_leafTables[i].nodes == nodes) { g_leafTables.erase(g_leafTables.begin() + (long)i); return; }
}

bool find_leaf_table(Context& ctx, const void* nodes, LeafTableRef& ref) {
	std::lock_guard<std::mutex> hold(ctx.tablesLock);
	std::vector<LeafTableRef>& g_leafTables = ctx.tables;
	for (size_t i = 0; i < g_leafTables.size(); i++) {
		if (g_leafTables[i].nodes != nodes) continue;
		// the host may have freed the construct buffer since: the kernel must not touch an address that is no longer mapped
		hipPointerAttribute_t attr;
		if (hipPointerGetAttributes(&attr, g_leafTables[i].block) != hipSuccess || attr.type != hipMemoryTypeDevice) {
			(void)hipGetLastError();
			g_leafTables.erase(g_leafTables.begin() + (long)i);
			return false;
		}
		ref = g_leafTables[i];
		return true;
	}
	return false;
}

// ---- launch sizing: how many batches a kernel_construct launch can find (simlod_internal.hpp launch_plan) ---------------------------------
// The reference's kernel loops over whatever has been uploaded when it starts (voxels.cu:870-885); here every group of batches is a handful of kernel
// launches the HOST enqueues before it can read that number (the upload counter lives on the device).  Kernels of a group without a batch leave at
// once, but its event hops stand in line behind the real work (~45 us per empty group).  What the host can know:
//   * the upload counter itself: the reference's uploader publishes it with cuMemsetD32Async(cptr_numBatchesUploaded, n, 1, stream_upload)
//     (main_progressive_octree.cpp:1047-1050) — shim/cuda.h passes every such write on (simlod_upload_counter_written), the Python mirror does the same;
//   * Stats.batchletIndex as the latest launch whose end the host has seen left it (k_finish stores it, the upload counter, whether the octree is
//     fit for exact groups, and the launch's sequence number into page-locked memory: no synchronisation), and how many batches the launches
//     enqueued since then were sized for.
// pending = uploaded - index - (what those launches will take).  A host that does not say (no shim, no hint): the counter the latest report saw + the
// uploader's pace, and never less than one group — a host that uploads, launches once and waits must not wait for ever.
static constexpr uint32_t NOTHING_SEEN = 0xffffffffu;

namespace {
struct UploadCounter { const void* ptr; uint32_t value; bool known; };      // known: the host has said what it wrote there (a reset: zero)
std::mutex g_countersLock;
std::vector<UploadCounter> g_counters;          // the upload counters the host has told about (simlod_upload_counter_written), by address
bool host_uploaded(const void* counter, uint32_t& value) {
	std::lock_guard<std::mutex> hold(g_countersLock);
	for (const UploadCounter& c : g_counters) if (c.ptr == counter) { value = c.value; return c.known; }
	return false;
}
}  // namespace
// written: the host has enqueued a write of `value` (remembered only for addresses a reset or a launch has named as an upload counter: the shims pass on
// every one-word memset); else: a launch names `counter` as its upload counter (what it holds stays unknown until the host says)
void note_upload_counter(const void* counter, uint32_t value, bool written, bool create) {
	std::lock_guard<std::mutex> hold(g_countersLock);
	for (UploadCounter& c : g_counters) if (c.ptr == counter) { if (written) { c.value = value; c.known = true; } return; }
	if (!create) return;
	if (g_counters.size() >= 64) g_counters.erase(g_counters.begin());
	g_counters.push_back(UploadCounter{counter, value, written});
}

static LaunchHistory* history_of(Context& ctx, const void* stats, bool create) {
	std::vector<LaunchHistory>& g_history = ctx.history;
	for (LaunchHistory& h : g_history) if (h.stats == stats) return &h;
	if (!create) return nullptr;
	volatile uint32_t* seen = nullptr;
	if (g_history.size() >= 64) {
		// the oldest entry makes room — its page-locked words are handed on, never freed: a copy enqueued by an earlier launch may still
		// be on its way into them (what arrives late is at worst a stale hint for the new owner: one launch with too many or too few groups)
		seen = g_history.front().seen;
		g_history.erase(g_history.begin());
	} else {
		void* pinned = nullptr;
		if (hipHostMalloc(&pinned, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		seen = static_cast<volatile uint32_t*>(pinned);
	}
	seen[0] = NOTHING_SEEN; seen[1] = NOTHING_SEEN; seen[2] = 1u; seen[3] = 0u;
	LaunchHistory h{};
	h.stats = stats; h.seen = seen;
	g_history.push_back(h);
	return &g_history.back();
}

LaunchPlan launch_plan(Context& ctx, const SimlodStats* stats, const void* uploadCounter) {
	LaunchPlan plan{SIMLOD_MAX_BATCHES_PER_LAUNCH, true, nullptr, 0u};
	const uint32_t limit = std::min<uint32_t>(ctx.batchLimit.load(), SIMLOD_MAX_BATCHES_PER_LAUNCH);
	const int hinted = ctx.hintPending.exchange(-1);                      // simlod_context_hint_pending_batches: for THIS launch
	std::lock_guard<std::mutex> hold(ctx.historyLock);
	LaunchHistory* h = ctx.tune(KNOB_ADAPTIVE_GROUPS, 1) != 0 ? history_of(ctx, stats, true) : nullptr;
	if (h == nullptr) { plan.batches = hinted >= 0 ? std::min<uint32_t>((uint32_t)hinted, limit) : limit; return plan; }
	const uint32_t seq = ++h->seq;
	plan.feedback = const_cast<uint32_t*>(h->seen); plan.seq = seq;
	const uint32_t index = h->seen[0], uploadedSeen = h->seen[1], reportSeq = h->seen[3];
	// a launch of this octree, or its reset, has ended and said so.  (A report from BEFORE the latest reset — a launch that was still running when the host
	// enqueued the reset writes its words after forget_launch_history cleared them — describes an octree that is gone: ignored; k_reset's own report follows it)
	const bool reported = index != NOTHING_SEEN && uploadedSeen != NOTHING_SEEN && (int32_t)(reportSeq - h->resetSeq) >= 0;
	plan.mayGroup = !reported || h->seen[2] != 0u;
	// what the launches enqueued behind the reporting one (or behind the reset, while nothing has reported) were sized for
	const uint32_t since = reported ? reportSeq : h->resetSeq;
	uint32_t inflight = 0;
	if (seq - since > 32u) inflight = NOTHING_SEEN;
	else for (uint32_t q = since + 1u; q != seq; q++) inflight += h->enq[q % 32u];
	const bool knowsStart = reported || h->resetKnown;                                  // index is known: the report's, or zero (a reset is in the stream ahead of this launch)
	const uint32_t indexNow = reported ? index : 0u;
	uint32_t want, uploadedHost = 0;
	if (hinted >= 0) want = (uint32_t)hinted;
	else if (knowsStart && inflight != NOTHING_SEEN && host_uploaded(uploadCounter, uploadedHost) && !(reported && uploadedSeen > uploadedHost)) {
		// (... unless a launch has SEEN more on the device than the host has told: a host that does not pass its counter writes on — the only value this
		// library knows is the zero of its own reset.  Such a host gets the prediction below)
		const uint32_t pending = uploadedHost > indexNow ? uploadedHost - indexNow : 0u;
		want = pending > inflight ? pending - inflight : 0u;
		// (a launch may take less than it was sized for — the memory guard, the time budget —: while the launches in flight have not reported, they
		// cannot be counted on to have taken everything, and this one enqueues one group: a frame loop must not stand still with batches pending)
		if (want == 0u && pending != 0u) want = 1u;
		// ... and nothing at all is enqueued only on the word of a host that has been HEARD since the reset or whose launches have reported: "zero uploaded"
		// right after a reset is also what a host that tells nothing looks like (it uploads, launches once and waits)
		if (want == 0u && !reported) want = 1u;
	} else if (!reported) want = SIMLOD_MAX_BATCHES_PER_LAUNCH;                          // nothing is known: everything
	else {
		// the host says nothing: what the latest report saw pending + what was uploaded between the last two reports (the uploader's pace per launch)
		const uint32_t pending = uploadedSeen > index ? uploadedSeen - index : 0u;
		if (!h->havePrev) h->arrivals = SIMLOD_MAX_BATCHES_PER_LAUNCH;                  // (one report says nothing about the pace)
		else if (index != h->prevIndex || uploadedSeen != h->prevUploaded)
			h->arrivals = uploadedSeen >= h->prevUploaded && index >= h->prevIndex ? uploadedSeen - h->prevUploaded : SIMLOD_MAX_BATCHES_PER_LAUNCH;   // (counters that went back: reset by other means)
		else if (pending == 0u) h->arrivals = 0u;                                       // the same numbers again and nothing pending: the loader is idle or done
		h->prevIndex = index; h->prevUploaded = uploadedSeen; h->havePrev = true;
		want = std::max<uint32_t>(1u, pending + h->arrivals);                           // (never nothing: such a host may upload, launch once and wait)
	}
	plan.batches = std::min<uint32_t>(want, limit);
	h->enq[seq % 32u] = plan.batches;
	return plan;
}

void forget_launch_history(Context& ctx, const SimlodStats* stats, const void* uploadCounter, uint32_t** words, uint32_t* seq) {
	if (uploadCounter != nullptr) note_upload_counter(uploadCounter, 0u, true, true);        // reset.cu:84: the reset zeroes the upload counter
	std::lock_guard<std::mutex> hold(ctx.historyLock);
	LaunchHistory* h = history_of(ctx, stats, words != nullptr);
	if (words != nullptr) { *words = nullptr; *seq = 0u; }
	if (h == nullptr) return;
	h->seen[0] = NOTHING_SEEN; h->seen[1] = NOTHING_SEEN; h->seen[2] = 1u;
	h->havePrev = false;
	h->resetSeq = ++h->seq; h->resetKnown = words != nullptr;
	h->enq[h->seq % 32u] = 0u;
	if (words != nullptr) { *words = const_cast<uint32_t*>(h->seen); *seq = h->seq; }  // (k_reset reports in stream order: index 0, uploaded 0)
}

// ---- frame feedback: did the render buffer's previous frame have nodes that sort into the screen bins (render.hip r_overflow) ----------
static uint64_t bytes_behind(const void* buffer) {
	hipDeviceptr_t base = nullptr;
	size_t size = 0;
	if (hipMemGetAddressRange(&base, &size, const_cast<void*>(buffer)) != hipSuccess || base == nullptr) { (void)hipGetLastError(); return 200000000ull; }   // main_progressive_octree.cpp:555
	const uint64_t off = (uint64_t)((const uint8_t*)buffer - (const uint8_t*)base);
	return off < size ? size - off : 0;
}

// The buffer's entry, made when the buffer is first seen.  The FIRST part of a frame decides (possible: the knobs and the buffer's size allow bins;
// bins: and the buffer's previous frame had nodes to sort) and the decision is kept for the frame's other parts, whatever the knobs say by then:
// r_visible has built the frame's draw items for it, and a colour pass that ran without the bins would drop the samples of the items that sort.
// An entry whose frame is between its first and its last part is not evicted.
uint32_t* frame_feedback(Context& ctx, const void* buffer, uint32_t parts, bool& possible, bool& bins, uint64_t& bufferBytes) {
	std::lock_guard<std::mutex> hold(ctx.framesLock);
	FrameFeedback* f = nullptr;
	for (FrameFeedback& g : ctx.frames) if (g.buffer == buffer) f = &g;
	if (f == nullptr) {
		volatile uint32_t* seen = nullptr;
		if (ctx.frames.size() >= 64) {                       // the oldest entry without a frame in progress makes room; its word is handed on (a late store into it: a stale hint)
			size_t victim = 0;
			for (size_t i = 0; i < ctx.frames.size(); i++) if (!ctx.frames[i].open) { victim = i; break; }
			seen = ctx.frames[victim].seen;
			ctx.frames.erase(ctx.frames.begin() + (long)victim);
		} else {
			void* pinned = nullptr;
			if (hipHostMalloc(&pinned, 64, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); bufferBytes = bytes_behind(buffer); bins = possible; return nullptr; }
			seen = static_cast<volatile uint32_t*>(pinned);
		}
		seen[0] = 1u;                                        // a buffer's first frame sorts
		ctx.frames.push_back(FrameFeedback{buffer, seen, false, 0, false, false});
		f = &ctx.frames.back();
	}
	if ((parts & RENDER_FIRST) != 0u) {
		f->bytes = bytes_behind(buffer);                     // (a host may free a buffer and get the address back with another size)
		f->possible = possible;
		f->bins = possible && f->seen[0] != 0u;
		f->open = true;
	}
	possible = f->possible; bins = f->bins; bufferBytes = f->bytes;
	if ((parts & RENDER_OUTPUT) != 0u) f->open = false;
	return const_cast<uint32_t*>(f->seen);
}

// the first part found that the buffer has no room for the bins after all (launch_render: the pool is what the allocation has left)
void frame_feedback_no_bins(Context& ctx, const void* buffer) {
	std::lock_guard<std::mutex> hold(ctx.framesLock);
	for (FrameFeedback& g : ctx.frames) if (g.buffer == buffer) { g.possible = false; g.bins = false; }
}

// ---- optional per-kernel profiling ------------------------------------------------------------------------------
struct ProfileMark { const char* name; hipEvent_t ev; };
static std::atomic<int> g_profile{0};          // 0 off | 1 every kernel, one stream | 2 the builder's dominant kernel only, streams as in production
static std::vector<ProfileMark> g_marks;
static std::vector<hipEvent_t> g_eventPool;

bool profile_enabled() { return g_profile.load(std::memory_order_relaxed) == 1; }
bool profile_dominant() { return g_profile.load(std::memory_order_relaxed) == 2; }

static hipEvent_t take_event() {
	if (!g_eventPool.empty()) { hipEvent_t e = g_eventPool.back(); g_eventPool.pop_back(); return e; }
	hipEvent_t e; (void)hipEventCreate(&e); return e;
}

void profile_mark(const char* kernelName, hipStream_t stream) {
	hipEvent_t e = take_event();
	(void)hipEventRecord(e, stream);
	g_marks.push_back({kernelName, e});
}

void profile_close(hipStream_t stream) { profile_mark(nullptr, stream); }

// a pair of events for hipExtLaunchKernelGGL's start / stop slots: the kernel's own begin and end, as the command processor stamps them (what rocprofv3
// reports) — an event RECORDED in front of a launch is stamped when the stream reaches it, several microseconds before the kernel starts
void profile_kernel_events(const char* kernelName, hipEvent_t* start, hipEvent_t* stop) {
	*start = take_event(); *stop = take_event();
	g_marks.push_back({kernelName, *start});
	g_marks.push_back({nullptr, *stop});
}

enum KernelKind { KIND_RESET = 0, KIND_CONSTRUCT = 1, KIND_RENDER = 2, KIND_FILTER = 3 };

}  // namespace simlod

struct SimlodFunction {
	simlod::KernelKind kind;
	std::string name;
};

struct SimlodProgram {
	std::vector<std::string> modules;
	std::vector<SimlodFunction> functions;
};

using namespace simlod;

extern "C" {

struct SimlodContext { simlod::Context ctx; };
static Context& ctx_or_default(SimlodContext* c) { return c != nullptr ? c->ctx : default_context(); }

int simlod_context_create(SimlodContext** out) {
	if (!out) return (int)hipErrorInvalidValue;
	*out = new SimlodContext();
	return 0;
}

int simlod_context_destroy(SimlodContext* c) {
	if (!c) return (int)hipErrorInvalidValue;
	{
		std::lock_guard<std::mutex> hold(g_attachLock);
		for (size_t i = 0; i < g_attached.size();) { if (g_attached[i].second == &c->ctx) g_attached.erase(g_attached.begin() + (long)i); else i++; }
	}
	delete c;                                    // (synchronises and destroys the context's second stream)
	return 0;
}

int simlod_context_attach(SimlodContext* c, const SimlodNode* nodes) {
	if (!nodes) return (int)hipErrorInvalidValue;
	std::lock_guard<std::mutex> hold(g_attachLock);
	for (size_t i = 0; i < g_attached.size(); i++)
		if (g_attached[i].first == nodes) { g_attached.erase(g_attached.begin() + (long)i); break; }
	if (c != nullptr) g_attached.emplace_back(nodes, &c->ctx);           // NULL: back to the default context
	return 0;
}

int simlod_context_set_node_capacity(SimlodContext* c, uint32_t numNodes) {
	// node indices travel in 19-bit fields (ancestor path entries, split records, claim-set keys: construct.hip)
	if (numNodes < 9 || numNodes > (1u << 19)) return (int)hipErrorInvalidValue;
	ctx_or_default(c).nodeCapacity.store(numNodes);
	return 0;
}
int simlod_context_set_ingest_mode(SimlodContext* c, uint32_t mode) {
	if (mode > 1u) return (int)hipErrorInvalidValue;
	ctx_or_default(c).ingestMode.store(mode);
	return 0;
}
int simlod_context_set_construct_batch_limit(SimlodContext* c, uint32_t maxBatches) {
	if (maxBatches == 0u) return (int)hipErrorInvalidValue;
	ctx_or_default(c).batchLimit.store(maxBatches > SIMLOD_MAX_BATCHES_PER_LAUNCH ? SIMLOD_MAX_BATCHES_PER_LAUNCH : maxBatches);
	return 0;
}
int simlod_context_hint_pending_batches(SimlodContext* c, uint32_t pending) {
	ctx_or_default(c).hintPending.store((int)std::min<uint32_t>(pending, SIMLOD_MAX_BATCHES_PER_LAUNCH));
	return 0;
}
int simlod_upload_counter_written(const void* numBatchesUploaded, uint32_t value) {
	if (!numBatchesUploaded) return (int)hipErrorInvalidValue;
	note_upload_counter(numBatchesUploaded, value, true, false);      // (only addresses a reset or a launch has named as an upload counter are remembered)
	return 0;
}
int simlod_context_set_trunk_mask(SimlodContext* c, uint64_t lo, uint64_t hi) {
	if ((hi >> 9) != 0ull) return (int)hipErrorInvalidValue;                  // 73 upper nodes: 1 + 8 + 64
	// a node can only split once it exists: every node the mask names below the root must have its parent named too
	for (uint32_t i = 1; i < 73u; i++) {
		const bool set = (((i < 64u ? lo : hi) >> (i & 63u)) & 1ull) != 0ull;
		const uint32_t parent = i < 9u ? 0u : 1u + ((i - 9u) >> 3);
		if (set && ((lo >> parent) & 1ull) == 0ull) return (int)hipErrorInvalidValue;
	}
	Context& ctx = ctx_or_default(c);
	ctx.trunkLo.store(lo); ctx.trunkHi.store(hi);
	return 0;
}
int simlod_context_set_knob(SimlodContext* c, const char* name, int value, int set) {
	if (!name) return (int)hipErrorInvalidValue;
	for (int k = 0; k < KNOB_COUNT_; k++)
		if (std::strcmp(name, KNOB_NAMES[k]) == 0) { ctx_or_default(c).knob[k] = set ? value : KNOB_UNSET; return 0; }
	return (int)hipErrorNotFound;
}
int simlod_context_reload_env(SimlodContext* c) { ctx_or_default(c).reload_env(); return 0; }

int simlod_set_node_capacity(uint32_t numNodes) { return simlod_context_set_node_capacity(nullptr, numNodes); }

uint64_t simlod_render_framebuffer_offset(void) {
	return (uint64_t)SIMLOD_MAX_VISIBLE_NODES * sizeof(SimlodNode) + 7 * 16 + 32 + 16000000ull;
}

uint64_t simlod_render_buffer_bytes(uint32_t width, uint32_t height) { return render_buffer_bytes(width, height); }

uint64_t simlod_construct_buffer_min_bytes(void) { return build::construct_min_bytes(default_context().nodeCapacity.load()); }
uint64_t simlod_context_construct_buffer_min_bytes(SimlodContext* c) { return build::construct_min_bytes(ctx_or_default(c).nodeCapacity.load()); }

int simlod_set_ingest_mode(uint32_t mode) { return simlod_context_set_ingest_mode(nullptr, mode); }

int simlod_set_construct_batch_limit(uint32_t maxBatches) { return simlod_context_set_construct_batch_limit(nullptr, maxBatches); }

int simlod_octree_image_replaced(const SimlodNode* nodes) {
	Context& ctx = context_of(nodes);
	forget_leaf_table(ctx, nodes);
	ctx.sideTablesStale.store(true);
	return 0;
}

int simlod_launch_reset(const SimlodUniforms* uniforms, uint8_t* buffer_octree, SimlodNode* nodes, SimlodStats* stats,
                        void* cudaprint, uint32_t* numBatchesUploaded, uint32_t* batchSizes, void* stream) {
	(void)cudaprint;   // CudaPrint's device side is a no-op (modules/CudaPrint/CudaPrint.cuh:49-51): accepted, unused
	if (!uniforms || !buffer_octree || !nodes || !stats || !numBatchesUploaded || !batchSizes) return (int)hipErrorInvalidValue;
	return launch_reset(context_of(nodes), uniforms, buffer_octree, nodes, stats, numBatchesUploaded, batchSizes, (hipStream_t)stream);
}

int simlod_launch_construct(const SimlodUniforms* uniforms, SimlodPoint* points, uint32_t* buffer, uint8_t* buffer_persistent,
                            SimlodNode* nodes, SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint,
                            uint32_t* numBatchesUploaded_volatile, uint32_t* batchSizes, void* stream) {
	(void)cudaprint;
	if (!uniforms || !points || !buffer || !buffer_persistent || !nodes || !stats || !frameStartTimestamp ||
	    !numBatchesUploaded_volatile || !batchSizes) return (int)hipErrorInvalidValue;
	return build::launch_construct(context_of(nodes), uniforms, points, buffer, buffer_persistent, nodes, stats, frameStartTimestamp, numBatchesUploaded_volatile, batchSizes, (hipStream_t)stream);
}

int simlod_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format, const double scale[3],
                      const double offset[3], SimlodPoint* out, void* stream) {
	return launch_decode_las(records, numPoints, bytesPerPoint, format, scale, offset, out, (hipStream_t)stream);
}

int simlod_launch_colorfilter(const SimlodUniforms* uniforms, uint32_t* buffer, SimlodNode* nodes, uint32_t* numNodes, SimlodStats* stats, void* stream) {
	if (!uniforms || !buffer || !nodes || (!numNodes && !stats)) return (int)hipErrorInvalidValue;
	return launch_colorfilter(context_of(nodes), uniforms, buffer, nodes, numNodes, stats, (hipStream_t)stream);
}

uint64_t simlod_colorfilter_buffer_min_bytes(void) { return colorfilter_min_bytes(default_context().nodeCapacity.load()); }

int simlod_generate_terrain(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed, uint32_t tilesX,
                            const float tileExtent[3], void* stream) {
	return launch_generate_terrain(out, numPoints, firstIndex, pointsPerTile, seed, tilesX, tileExtent, 0.0f, (hipStream_t)stream);
}

int simlod_generate_terrain_scan(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed, uint32_t tilesX,
                                 const float tileExtent[3], float swathWidth, void* stream) {
	return launch_generate_terrain(out, numPoints, firstIndex, pointsPerTile, seed, tilesX, tileExtent, swathWidth, (hipStream_t)stream);
}

int simlod_launch_render(uint32_t* buffer, const SimlodUniforms* uniforms, SimlodNode* nodes, uint32_t* colorbuffer,
                         SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint, void* stream) {
	(void)cudaprint;
	if (!uniforms || !buffer || !nodes || !stats || !frameStartTimestamp) return (int)hipErrorInvalidValue;
	return launch_render(context_of(nodes), buffer, uniforms, nodes, colorbuffer, stats, frameStartTimestamp, (hipStream_t)stream, RENDER_ALL);
}

int simlod_launch_render_part(uint32_t part, uint32_t* buffer, const SimlodUniforms* uniforms, SimlodNode* nodes, uint32_t* colorbuffer,
                              SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint, void* stream) {
	(void)cudaprint;
	if (!uniforms || !buffer || !nodes || !stats || !frameStartTimestamp || part > 3) return (int)hipErrorInvalidValue;
	return launch_render(context_of(nodes), buffer, uniforms, nodes, colorbuffer, stats, frameStartTimestamp, (hipStream_t)stream, 1u << part);
}

// ---- a frame composed across ranks in one call (include/simlod_hip.h) -------------------------------------------------------------------
int simlod_render_frame_composed(uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                                 uint64_t* frameStartTimestamp, void* cudaprint, void* stream, SimlodReduceFn reduce, void* user) {
	(void)cudaprint;
	if (!u || !buffer || !nodes || !stats || !frameStartTimestamp) return (int)hipErrorInvalidValue;
	Context& ctx = context_of(nodes);
	hipStream_t st = (hipStream_t)stream;
	const uint32_t W = (uint32_t)u->width, H = (uint32_t)u->height;
	const uint64_t px = (uint64_t)W * H;
	uint8_t* base = reinterpret_cast<uint8_t*>(buffer);
	auto part = [&](uint32_t k) { return launch_render(ctx, buffer, u, nodes, colorbuffer, stats, frameStartTimestamp, st, 1u << k); };
	auto red = [&](uint32_t plane, uint64_t offset, uint64_t count, uint32_t elemBytes, uint32_t op) { return reduce ? reduce(user, plane, base + offset, count, elemBytes, op, stream) : 0; };
	const bool hqs = u->useHighQualityShading != 0, boxes = u->showBoundingBox != 0;
	int rc = part(0);
	if (rc == 0 && hqs) rc = red(SIMLOD_PLANE_DEPTH, render_depth_plane_offset(W, H), px, 4, SIMLOD_REDUCE_MIN);
	if (rc == 0 && hqs) rc = part(1);
	if (rc == 0 && hqs) rc = red(SIMLOD_PLANE_SUMS, render_sum_planes_offset(W, H), px * 4, 4, SIMLOD_REDUCE_SUM);
	if (rc == 0 && hqs) rc = part(2);
	if (rc == 0 && (!hqs || boxes)) rc = red(SIMLOD_PLANE_FRAMEBUFFER, simlod_render_framebuffer_offset(), px, 8, SIMLOD_REDUCE_MIN);
	if (rc == 0) rc = part(3);
	return rc;
}

// RCCL, found at run time: ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream).
// The library that made the caller's ncclComm_t must be the one that reduces with it: the copy ALREADY LOADED in the process is looked up first
// (a PyTorch process carries its own torch/lib/librccl.so, loaded by path; dlopen("librccl.so") beside it would load a second instance and hand the
// communicator to the wrong one); only a process without one gets a dlopen by name.  The datatype / operator numbers below are the ABI of
// NCCL / RCCL 2.x (rccl.h ncclDataType_t: ncclUint32 = 3, ncclUint64 = 5; ncclRedOp_t: ncclSum = 0, ncclMin = 3): ncclGetVersion must name a 2.x
// release from 2.10 on (where the enums took that shape and have stayed) or the entry refuses with hipErrorNotSupported.
namespace {
using AllReduceFn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
using VersionFn = int (*)(int*);
struct Rccl { AllReduceFn allReduce = nullptr; int version = 0; bool supported = false; };
const Rccl* rccl() {
	static const Rccl r = [] {
		Rccl x;
		void* fn = dlsym(RTLD_DEFAULT, "ncclAllReduce");
		void* ver = fn != nullptr ? dlsym(RTLD_DEFAULT, "ncclGetVersion") : nullptr;
		if (fn == nullptr) {
			for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
				void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);             // loaded under that name already?
				if (h == nullptr) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
				if (h != nullptr && (fn = dlsym(h, "ncclAllReduce")) != nullptr) { ver = dlsym(h, "ncclGetVersion"); break; }
			}
		}
		x.allReduce = reinterpret_cast<AllReduceFn>(fn);
		if (ver != nullptr && reinterpret_cast<VersionFn>(ver)(&x.version) != 0) x.version = 0;
		// NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + ... before)
		x.supported = x.allReduce != nullptr && x.version >= 21000 && x.version < 30000;
		return x;
	}();
	return &r;
}
int reduce_over_rccl(void* comm, uint32_t plane, void* data, uint64_t count, uint32_t elemBytes, uint32_t op, void* stream) {
	(void)plane;
	const Rccl& r = *rccl();
	if (!r.supported) return (int)hipErrorNotSupported;
	enum { ncclUint32 = 3, ncclUint64 = 5, ncclSum = 0, ncclMin = 3 };       // rccl.h: ncclDataType_t, ncclRedOp_t (2.x ABI, checked above)
	return r.allReduce(data, data, (size_t)count, elemBytes == 8 ? ncclUint64 : ncclUint32, op == SIMLOD_REDUCE_SUM ? ncclSum : ncclMin, comm, (hipStream_t)stream) == 0 ? 0 : (int)hipErrorUnknown;
}
}  // namespace

int simlod_rccl_version(void) { return rccl()->version; }

int simlod_render_frame_rccl(uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                             uint64_t* frameStartTimestamp, void* cudaprint, void* stream, void* ncclComm) {
	if (!ncclComm) return (int)hipErrorInvalidValue;
	if (!rccl()->supported) return (int)hipErrorNotSupported;           // no RCCL in the process, or one whose enum ABI this entry was not written for
	return simlod_render_frame_composed(buffer, u, nodes, colorbuffer, stats, frameStartTimestamp, cudaprint, stream, reduce_over_rccl, ncclComm);
}

uint64_t simlod_render_depth_plane_offset(uint32_t width, uint32_t height) { return render_depth_plane_offset(width, height); }
uint64_t simlod_render_sum_planes_offset(uint32_t width, uint32_t height) { return render_sum_planes_offset(width, height); }

static bool ends_with(const std::string& s, const char* suffix) {
	const size_t n = std::strlen(suffix);
	return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int simlod_program_create(SimlodProgram** out, const char* const* modules, int numModules, const char* const* kernels, int numKernels) {
	if (!out) return (int)hipErrorInvalidValue;
	*out = nullptr;
	SimlodProgram* p = new SimlodProgram();
	bool hasReset = false, hasUpdate = false, hasRender = false, hasFilter = false;
	for (int i = 0; i < numModules; i++) {
		p->modules.emplace_back(modules[i] ? modules[i] : "");
		const std::string& m = p->modules.back();
		hasReset |= ends_with(m, "reset.cu");
		hasUpdate |= ends_with(m, "progressive_octree_voxels.cu");
		hasRender |= ends_with(m, "render.cu");
		hasFilter |= ends_with(m, "colorfilter.cu");
	}
	for (int i = 0; i < numKernels; i++) {
		const std::string k = kernels[i] ? kernels[i] : "";
		SimlodFunction f;
		f.name = k;
		if (k == "kernel" && hasReset) f.kind = KIND_RESET;
		else if (k == "kernel" && hasFilter) f.kind = KIND_FILTER;
		else if (k == "kernel_construct" && hasUpdate) f.kind = KIND_CONSTRUCT;
		else if (k == "kernel_render" && hasRender) f.kind = KIND_RENDER;
		else { delete p; return (int)hipErrorNotFound; }
		p->functions.push_back(f);
	}
	*out = p;
	return 0;
}

void simlod_program_destroy(SimlodProgram* program) { delete program; }

SimlodFunction* simlod_program_kernel(SimlodProgram* program, const char* name) {
	if (!program || !name) return nullptr;
	for (auto& f : program->functions) if (f.name == name) return &f;
	return nullptr;
}

int simlod_function_max_active_blocks(SimlodFunction* fn, int blockSize, int* numBlocks) {
	if (!fn || !numBlocks || blockSize <= 0) return (int)hipErrorInvalidValue;
	*numBlocks = 2048 / blockSize > 0 ? (2048 / blockSize > 8 ? 8 : 2048 / blockSize) : 1;   // 32 waves per CU
	return 0;
}

int simlod_launch_cooperative(SimlodFunction* fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                              unsigned sharedMemBytes, void* stream, void** args) {
	(void)gx; (void)gy; (void)gz; (void)bx; (void)by; (void)bz; (void)sharedMemBytes;
	if (!fn || !args) return (int)hipErrorInvalidValue;
	switch (fn->kind) {
	case KIND_RESET:      // main_progressive_octree.cpp:337-345
		return simlod_launch_reset((const SimlodUniforms*)args[0], *(uint8_t**)args[1], *(SimlodNode**)args[2], *(SimlodStats**)args[3],
		                           *(void**)args[4], *(uint32_t**)args[5], *(uint32_t**)args[6], stream);
	case KIND_CONSTRUCT:  // main_progressive_octree.cpp:374-382
		return simlod_launch_construct((const SimlodUniforms*)args[0], *(SimlodPoint**)args[1], *(uint32_t**)args[2], *(uint8_t**)args[3],
		                               *(SimlodNode**)args[4], *(SimlodStats**)args[5], *(uint64_t**)args[6], *(void**)args[7],
		                               *(uint32_t**)args[8], *(uint32_t**)args[9], stream);
	case KIND_FILTER:     // colorfilter.cu:164-169: (Uniforms, buffer, nodes, numNodes, stats)
		return simlod_launch_colorfilter((const SimlodUniforms*)args[0], *(uint32_t**)args[1], *(SimlodNode**)args[2], *(uint32_t**)args[3], *(SimlodStats**)args[4], stream);
	case KIND_RENDER:     // main_progressive_octree.cpp:499-507
		return simlod_launch_render(*(uint32_t**)args[0], (const SimlodUniforms*)args[1], *(SimlodNode**)args[2], *(uint32_t**)args[3],
		                            *(SimlodStats**)args[4], *(uint64_t**)args[5], *(void**)args[6], stream);
	}
	return (int)hipErrorInvalidValue;
}

int simlod_profile_enable(int on) {
	g_profile.store(on == 2 ? 2 : on != 0 ? 1 : 0);
	return 0;
}

// Synchronises the device, folds the marks recorded since the last call into per-kernel totals and clears them.
int simlod_profile_collect(SimlodProfileEntry* out, int capacity, int* count) {
	if (!out || !count) return (int)hipErrorInvalidValue;
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) return (int)e;
	int n = 0;
	for (size_t i = 0; i + 1 < g_marks.size(); i++) {
		if (g_marks[i].name == nullptr) continue;       // end-of-call marker
		float ms = 0.f;
		if (hipEventElapsedTime(&ms, g_marks[i].ev, g_marks[i + 1].ev) != hipSuccess) continue;
		int k = 0;
		for (; k < n; k++) if (std::strcmp(out[k].name, g_marks[i].name) == 0) break;
		if (k == n) {
			if (n >= capacity) continue;
			std::memset(&out[n], 0, sizeof(out[n]));
			std::strncpy(out[n].name, g_marks[i].name, sizeof(out[n].name) - 1);
			n++;
		}
		out[k].launches += 1;
		out[k].total_ms += ms;
	}
	for (auto& m : g_marks) g_eventPool.push_back(m.ev);
	g_marks.clear();
	*count = n;
	return 0;
}

const char* simlod_build_info(void) { return "simlod_hip gfx950 (-ffp-contract=off, IEEE div/sqrt) " __DATE__ " " __TIME__; }

}  // extern "C"
