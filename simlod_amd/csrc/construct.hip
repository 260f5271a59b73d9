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

// The device code, stage by stage (textual parts of THIS translation unit, inside namespace simlod::build; each names what it holds in its first line):
#include "construct_state.inc"      // BatchCtl / Ctl / BuildArgs, trunk mask, LDS count tables, side-table formats, prepare_batch, Samples
#include "construct_begin.inc"      // k_begin, rebuild_side_tables
#include "construct_count.inc"      // k_count, k_queue, reserve(), account_group
#include "construct_expand.inc"     // alloc_points, k_hist, k_expand
#include "construct_voxelize.inc"   // voxel chunks on demand, k_voxelize, voxelize_small, voxroot_pieces, end_of_batch
#include "construct_insert.inc"     // k_voxdone, k_rootpre, k_insert, k_stats, k_finish

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
	SIMLOD_LAUNCH(k_begin, dim3(fits ? dev.numCUs * 2 : 1u), dim3(TPB), stream, a, fits ? 0u : 1u, limit, ((uint32_t)ctx.tune(KNOB_DEBUG_FORCE_BARRIER_TIMEOUT, 0) & 1u) | (ctx.tune(KNOB_DEBUG_VOXELIZE_CLOCK, 0) != 0 ? 2u : 0u) | (ctx.sideTablesStale.exchange(false) ? 4u : 0u) | (ctx.tune(KNOB_DEBUG_IRREGULAR_CHILDREN, 0) != 0 ? 8u : 0u) | (((uint32_t)ctx.tune(KNOB_DEBUG_PHASE_WG, 0) & 0xffffu) << 8),
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
		// (groups of several batches: one workgroup per TWO CUs — 2.70 against 2.75 ms per 36 M with one per CU, and, what matters more, two PROCESSES that
		// build octrees on one GPU can both have their k_expand resident: with a workgroup per CU each, the two launches held half the chip each and waited
		// for the other half until the barrier gave up — Stats.dbg 0x40 in test_bench_n2..., once in a few runs)
		const uint32_t expandWgs = (uint32_t)max(1, min(ctx.tune(KNOB_EXPAND_WGS, !single ? (int)dev.numCUs / 2 : (int)dev.numCUs / (overlap ? 4 : 2)), (int)dev.numCUs));
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
				if (!single) {      // groups: round 0, the next round's histogram pass over the whole chip, then round 1 and whatever follows (k_expand's comment)
					SIMLOD_LAUNCH(k_expand, dim3(expandWgs), dim3(ETPB), stream, a, b, 0u, 1u);
					SIMLOD_LAUNCH(k_hist2, dim3(gridPoints), dim3(TPB), stream, a, b);
				}
				const bool gated = expand_gate_enter(ctx, stream);
				SIMLOD_LAUNCH_STOP(k_expand, dim3(expandWgs), dim3(ETPB), stream, side->expanded[b], a, b, single ? 0u : 1u, (uint32_t)SIMLOD_MAX_EXPAND_ROUNDS);
				expand_gate_leave(ctx, stream, side->expanded[b], gated);
				{ const hipError_t e = hipStreamWaitEvent(back, side->expanded[b], 0); if (e != hipSuccess) return fail(e); }
				if (!single && a.acct != 0u) SIMLOD_LAUNCH(k_rootpre, dim3(1), dim3(1024), back, a, b);
				if (single) SIMLOD_LAUNCH_STOP(k_insert<true>, dim3(gridPoints), dim3(TPB), back, side->inserted[b], a, b);   // grid clears, points, end-of-batch bookkeeping, the previous group's voxel lists
				else SIMLOD_LAUNCH_STOP(k_insert<false>, dim3(gridPoints), dim3(TPB), back, side->inserted[b], a, b);
			} else {
				if (!single) {
					SIMLOD_LAUNCH(k_expand, dim3(expandWgs), dim3(ETPB), stream, a, b, 0u, 1u);
					SIMLOD_LAUNCH(k_hist2, dim3(gridPoints), dim3(TPB), stream, a, b);
				}
				const bool gated = expand_gate_enter(ctx, stream);
				SIMLOD_LAUNCH(k_expand, dim3(expandWgs), dim3(ETPB), stream, a, b, single ? 0u : 1u, (uint32_t)SIMLOD_MAX_EXPAND_ROUNDS);
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
