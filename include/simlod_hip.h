/* simlod_hip.h — C ABI of libsimlod_hip.so, the MI355X (gfx950) implementation of SimLOD's two hot paths.
 *
 * Drop-in boundary.  The reference host reaches its device code through exactly one surface:
 *
 *     CudaModularProgram({.modules = {...cu paths}, .kernels = {names}})      include/CudaModularProgram.h:166-190
 *     program->kernels["name"]  -> CUfunction                                 include/CudaModularProgram.h:241-256
 *     cuLaunchCooperativeKernel(fn, gx,gy,gz, bx,by,bz, smem, stream, void** args)
 *                                         modules/progressive_octree/main_progressive_octree.cpp:351, :396, :507
 *
 * with three programs / kernels:
 *     reset  : {reset.cu, utils.cu}                    -> "kernel"             main_progressive_octree.cpp:620-626
 *     update : {progressive_octree_voxels.cu, utils.cu} -> "kernel_construct"  main_progressive_octree.cpp:603-610
 *     render : {render.cu, utils.cu}                   -> "kernel_render"      main_progressive_octree.cpp:612-618
 *
 * This header exports that surface 1:1 (simlod_program_* / simlod_launch_cooperative, same argument arrays as
 * the reference builds at main_progressive_octree.cpp:337-345, :374-382, :499-507) plus typed entry points for
 * hosts that prefer not to build void* arrays.  All pointers are DEVICE pointers unless stated otherwise; all
 * structs are the ones of simlod_abi.h.  Every function returns 0 on success or a hipError_t value; launches are
 * asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream), like the reference's.
 *
 * Device-side conditions the reference reports by printf or by silently dropping data are
 * reported in Stats.dbg (a field the reference never writes) as a bit mask of SIMLOD_ERR_*.
 */
#ifndef SIMLOD_HIP_H
#define SIMLOD_HIP_H

#include "simlod_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Stats.dbg bits */
#define SIMLOD_ERR_MOMENTARY_TOO_SMALL 0x001u /* Uniforms.momentaryBufferCapacity cannot hold the scratch layout   */
#define SIMLOD_ERR_SPILLED_OVERFLOW    0x002u /* spill space exhausted: some splits were DEFERRED to a later batch (no point lost) */
#define SIMLOD_ERR_SPILLING_OVERFLOW   0x004u /* more leaves cross the limit at once than one split round holds (65 536; voxels.cu:847: 100 000): the rest is deferred */
#define SIMLOD_ERR_NODES_EXHAUSTED     0x008u /* node array full (main_progressive_octree.cpp:552: 263 157 nodes): leaves stop splitting */
#define SIMLOD_ERR_DIRECTORY_FULL      0x010u /* FATAL: directory of the chunks allocated in a batch full (sticky until reset) */
#define SIMLOD_ERR_NULL_CHUNK          0x020u /* insert into a leaf without storage (voxels.cu:599-604)             */
#define SIMLOD_ERR_BARRIER_TIMEOUT     0x040u /* FATAL: in-kernel grid barrier of the split cascade gave up (sticky until reset) */
#define SIMLOD_ERR_CHUNK_QUEUE_OVERFLOW 0x080u /* > 1 000 000 recycled chunks (voxels.cu:856)                        */
#define SIMLOD_ERR_VISIBLE_OVERFLOW    0x100u /* > 100 000 visible nodes (render.cu:1108)                           */
#define SIMLOD_ERR_ACCOUNTING          0x200u /* exact mode, launches of several batches: the per-batch chunk accounting of a group did not end at the chunk count the octree has (Stats.allocatedBytes_persistent / chunkPoolSize may differ from the reference's; the octree itself is sound) */

/* ---- per-octree contexts --------------------------------------------------------------------------------------------------------
 * The reference host keeps ONE octree per process and its launch signatures carry no handle (main_progressive_octree.cpp:337-345,
 * :374-382, :499-507).  What this library keeps between launches — ingest mode, node capacity, batch limit, tuning knobs, its second
 * stream and events, the table registry that links kernel_construct to kernel_render, the launch feedback — lives in a context; a
 * launch finds its context through the NODE ARRAY it is given.  Node arrays that were never attached share the default context, which
 * is what the simlod_set_* calls below configure: a host with one octree never needs these functions.  A host with several octrees
 * (one per tile, per data set, per thread) makes a context per octree and attaches the octree's node array to it.
 *
 * Tuning knobs (SIMLOD_OVERLAP_TAIL, SIMLOD_EXPAND_WGS, SIMLOD_GRID_MULT, SIMLOD_COUNT_TPB, SIMLOD_VOXELIZE_WGS, SIMLOD_ADAPTIVE_GROUPS,
 * SIMLOD_RASTER_LEAF_TABLE, SIMLOD_RASTER_LDS_TILES, SIMLOD_DRAW_MULT, SIMLOD_RASTER_FUSED_RESOLVE, SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT,
 * SIMLOD_DEBUG_VOXELIZE_CLOCK, SIMLOD_DEBUG_BUDGET_US, SIMLOD_GROUP_BATCHES, SIMLOD_DEBUG_PHASE_WG, SIMLOD_EVENT_SYSTEM_FENCE — 1: the
 * events between the builder's two streams keep the system-scope fence HIP gives an event by default —, SIMLOD_RASTER_SCREEN_BINS — 0:
 * no screen bins; n: nodes whose screen box exceeds n x 1024 pixels sort their samples into the bins (default 32) —,
 * SIMLOD_DEBUG_BIN_POOL — entries of the bin pool, for tests —, SIMLOD_EXACT_GROUP — batches an EXACT-mode group may have (default 5, at most 12;
 * 1: one batch per group as before round 6; the momentary buffer's size decides how many fit) —, SIMLOD_DEBUG_IRREGULAR_CHILDREN — 1: every inner node's child word
 * sends k_count's descent through Node.children (the path of an image whose children are not eight consecutive nodes; for tests)) are read from the environment ONCE, when a context is made (the default
 * context: at its first use); simlod_context_set_knob overrides one by name (set = 0: back to the built-in default),
 * simlod_context_reload_env reads the environment again.  ctx == NULL means the default context everywhere. */
typedef struct SimlodContext SimlodContext;
int simlod_context_create(SimlodContext** out);
int simlod_context_destroy(SimlodContext* ctx);                               /* synchronises the device, detaches the context's node arrays; no launch with one of them may be in progress on another thread */
int simlod_context_attach(SimlodContext* ctx, const SimlodNode* nodes);       /* launches given `nodes` run in ctx from now on (NULL: default again) */
int simlod_context_set_node_capacity(SimlodContext* ctx, uint32_t numNodes);
int simlod_context_set_ingest_mode(SimlodContext* ctx, uint32_t mode);
int simlod_context_set_construct_batch_limit(SimlodContext* ctx, uint32_t maxBatches);
/* "That many ring batches are pending right now" (uploaded and not ingested, not counting what launches already enqueued will take): for the NEXT
 * kernel_construct launch of the context only — it enqueues kernels for that many batches (0: none, an idle frame), whatever the library would have
 * guessed.  Optional: hosts whose upload-counter writes reach simlod_upload_counter_written (shim/cuda.h does that for the reference's) need not call it. */
int simlod_context_hint_pending_batches(SimlodContext* ctx, uint32_t pending);
/* The host has enqueued a write of `value` to the 4-byte upload counter at `numBatchesUploaded` (main_progressive_octree.cpp:1047-1050:
 * cuMemsetD32Async(cptr_numBatchesUploaded, batchStreamUploadIndex + 1, 1, stream_upload)).  kernel_construct reads the counter on the device when it
 * runs (voxels.cu:870-885); the library, which has to enqueue a group of kernels per batch BEFORE that, sizes its launches by what it is told here and by
 * what its earlier launches reported.  Addresses that no reset / construct launch has been given as `numBatchesUploaded` are ignored (the shims forward
 * every 4-byte memset).  Without it the library predicts from its launches' own reports alone: at least one group per launch, a burst picked up a launch late. */
int simlod_upload_counter_written(const void* numBatchesUploaded, uint32_t value);
/* Multi-GPU jobs (no counterpart in the reference, which is single-GPU: main_progressive_octree.cpp:274, CudaModularProgram.h:215).  Ranks own
 * level-3 cells of ONE global cube; the nodes of levels 0-2 exist on every rank.  The single-GPU octree of the whole data set splits such a
 * node when the GLOBAL count under it crosses 50 000 (progressive_octree_voxels.cu:209-217); a rank that looked at its own count would keep it
 * as a leaf and kernel_render would draw its points where one GPU draws the node's voxels (render.cu:918-932).  So the host names the upper
 * nodes whose global count exceeds the limit — 73 bits: bit 0 the root; bit 1 + c the level-1 node with cell code c = x << 2 | y << 1 | z;
 * bit 9 + c the level-2 node, c = its level-1 octant << 3 | the octant below (lo: bits 0..63, hi: bits 64..72) — and kernel_construct splits
 * such a node as soon as it exists, whatever it holds: in the first batch ingested after the call, or, when no batch is pending, by a batch
 * of ZERO points (publish batchSizes[slot] = 0 and bump numBatchesUploaded).  A named node's parent must be named too (hipErrorInvalidValue).
 * Mask 0 (the default): the reference's rule alone — every Node and Stats field is the reference's.  simlod_amd/distributed.py trunk_mask
 * derives the mask from the all-reduced histogram over the 512 level-3 cells; with it the composed frame of N ranks
 * (simlod_render_frame_composed) has the single-GPU frame's depth at every pixel. */
int simlod_context_set_trunk_mask(SimlodContext* ctx, uint64_t lo, uint64_t hi);
int simlod_context_set_knob(SimlodContext* ctx, const char* name, int value, int set);
int simlod_context_reload_env(SimlodContext* ctx);
uint64_t simlod_context_construct_buffer_min_bytes(SimlodContext* ctx);

/* Number of Node records the host's node buffer holds (default 263 157 = 40 000 000 / 152,
 * main_progressive_octree.cpp:552).  Default context; set before the first reset if the host allocates differently. */
int simlod_set_node_capacity(uint32_t numNodes);

/* Ingest granularity of kernel_construct.  0 (default) = EXACT: one ring batch at a time, as progressive_octree_voxels.cu:883-949 does
 * — every Node and Stats field after every batch is the reference's.  1 = COALESCED: all pending batches of a launch (<= 20, as many
 * as the momentary buffer holds) are ingested as one batch.  Topology, per-node sample multisets, occupancy bitsets, voxel positions
 * and counts do not depend on the granularity; the allocator / chunk-pool accounting (Stats.allocatedBytes_persistent,
 * numAllocatedChunks, chunkPoolSize) does: fewer intermediate chunks are ever allocated.  Default context. */
int simlod_set_ingest_mode(uint32_t mode);

/* Optional host hint: no more than `maxBatches` (1..20, default 20) ring batches are pending when kernel_construct is launched, so
 * no more than that many per-batch kernel groups need to be enqueued (the reference host knows its upload counter,
 * main_progressive_octree.cpp:1012-1050).  A launch never ingests more than this many batches.  Default context. */
int simlod_set_construct_batch_limit(uint32_t maxBatches);

/* kernel_render reads the chunk lists of visible nodes through a table kernel_construct keeps in ITS momentary buffer (one row of chunk
 * addresses per node), for as long as a stamp in that buffer says the table describes the octree in `nodes` as it is now: same node
 * array, same Stats.batchletIndex / numNodes / numPoints / numVoxels / allocatedBytes_persistent as after the last kernel_construct.
 * kernel_reset drops the association.  A host that writes an octree image into `nodes` / the persistent buffer by other means
 * (memcpy of a saved image) calls this afterwards; kernel_render then walks the `next` pointers, as the reference does
 * (render.cu:106-159), until kernel_construct has run again. */
int simlod_octree_image_replaced(const SimlodNode* nodes);

/* Byte offset of the uint64 framebuffer inside kernel_render's momentary `buffer` (identical to where the
 * reference's bump allocator places it, render.cu:1108-1123) and the size of that buffer's full layout: planes, draw items, chunk directory
 * and, last, the screen bins (a 48 MB pool of 16-byte entries + per-bin tables) that frames with very large nodes sort their samples into.
 * 1920 x 1080: 193.5 MB — inside the 200 000 000 bytes the reference host allocates whatever its window's size (main_progressive_octree.cpp:555).
 * Larger frames need more for the full layout (1920 x 1200: 202 MB, 2560 x 1440: 255 MB); a buffer that is smaller is still fine as long as it
 * holds everything in front of the bin pool (2560 x 1440: 199.2 MB): kernel_render asks the runtime how large the ALLOCATION behind `buffer` is
 * (hipMemGetAddressRange) and sizes the pool by what is left — with less than 1 MB left it draws without bins (same frame, the samples of
 * screen-filling nodes take device-scope atomics).  A host that carves `buffer` out of a larger allocation of its own must therefore give it
 * simlod_render_buffer_bytes(width, height): what lies behind `buffer` inside that allocation is taken for the pool. */
uint64_t simlod_render_framebuffer_offset(void);
uint64_t simlod_render_buffer_bytes(uint32_t width, uint32_t height);
/* Minimum size of kernel_construct's momentary `buffer` (the host allocates 300 MB, main_progressive_octree.cpp:554). */
uint64_t simlod_construct_buffer_min_bytes(void);

/* ---- typed launches -------------------------------------------------------------------------------------- */
/* reset.cu:20-29  `kernel` */
int simlod_launch_reset(const SimlodUniforms* uniforms /*host*/, uint8_t* buffer_octree, SimlodNode* nodes,
                        SimlodStats* stats, void* cudaprint, uint32_t* numBatchesUploaded, uint32_t* batchSizes,
                        void* stream);

/* progressive_octree_voxels.cu:804-816  `kernel_construct` */
int simlod_launch_construct(const SimlodUniforms* uniforms /*host*/, SimlodPoint* points, uint32_t* buffer,
                            uint8_t* buffer_persistent, SimlodNode* nodes, SimlodStats* stats,
                            uint64_t* frameStartTimestamp, void* cudaprint, uint32_t* numBatchesUploaded_volatile,
                            uint32_t* batchSizes, void* stream);

/* render.cu:1084-1093  `kernel_render`; `colorbuffer` stands for the GL surface: a linear width*height RGBA8 image */
int simlod_launch_render(uint32_t* buffer, const SimlodUniforms* uniforms /*host*/, SimlodNode* nodes,
                         uint32_t* colorbuffer, SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint,
                         void* stream);

/* colorfilter.cu:163-169  `kernel` of the colour-filter module (its host call is commented out in the reference,
 * main_progressive_octree.cpp:430-462): every voxel of every inner node becomes the average colour of the child samples in its cell,
 * bottom-up, ten levels of inner nodes per call.  `buffer` is the momentary buffer (Uniforms.momentaryBufferCapacity bytes, at least
 * simlod_colorfilter_buffer_min_bytes()); `numNodes` is a DEVICE pointer to the node count, or NULL for Stats.numNodes.  Voxel
 * positions, counts and list structure stay as they are; Node.isFiltered is set on every node that was processed. */
int simlod_launch_colorfilter(const SimlodUniforms* uniforms /*host*/, uint32_t* buffer, SimlodNode* nodes, uint32_t* numNodes,
                              SimlodStats* stats, void* stream);
uint64_t simlod_colorfilter_buffer_min_bytes(void);

/* kernel_render in four parts, for frames composed across GPUs (the reference is single-GPU).  Every rank calls the
 * parts in order on its own octree and reduces the named plane of the render buffer over all ranks in between:
 *   part 0  clear, visibility, first pass — plain: the 64-bit atomicMin pass and the debug lines; HQS: the depth pass
 *           HQS: all-reduce(MIN, 32-bit) of the depth plane   [simlod_render_depth_plane_offset, width*height uint32]
 *   part 1  HQS only: colour pass, sums folded into the sum planes
 *           HQS: all-reduce(SUM, 32-bit) of the sum planes     [simlod_render_sum_planes_offset, width*height x {R,G,B,count} uint32]
 *   part 2  HQS only: resolve (render.cu:607-632), then the debug lines
 *           all-reduce(MIN, 64-bit) of the framebuffer          [simlod_render_framebuffer_offset, width*height uint64; the stored
 *           words never have the sign bit set, so a signed MIN will do] — for HQS only needed when showBoundingBox is set
 *   part 3  Stats, EDL, RGBA8 output
 * With one rank and no reductions the four parts produce exactly the frame of simlod_launch_render. */
int simlod_launch_render_part(uint32_t part, uint32_t* buffer, const SimlodUniforms* uniforms /*host*/, SimlodNode* nodes,
                              uint32_t* colorbuffer, SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint,
                              void* stream);
uint64_t simlod_render_depth_plane_offset(uint32_t width, uint32_t height);
uint64_t simlod_render_sum_planes_offset(uint32_t width, uint32_t height);

/* The four parts and the reductions between them in ONE call, for hosts that are not Python (simlod_amd/distributed.py render_frame is the
 * same sequence over torch.distributed).  `reduce` is called on the launch stream's timeline, between the parts, with the plane to reduce
 * IN PLACE over all ranks: `data` (device memory inside `buffer`), `count` elements of `elemBytes` bytes, `op`; it returns 0 or an error
 * code, which ends the frame.  Plane and order, per frame:
 *     HQS:   SIMLOD_PLANE_DEPTH (uint32, MIN) after part 0;  SIMLOD_PLANE_SUMS (uint32 x 4 per pixel, SUM) after part 1;
 *            SIMLOD_PLANE_FRAMEBUFFER (uint64, MIN) after part 2 only when Uniforms.showBoundingBox is set
 *     plain: SIMLOD_PLANE_FRAMEBUFFER (uint64, MIN) after part 0
 * reduce == NULL: no reduction — the frame of simlod_launch_render.  The all-gather of the visible-node records (the first Stats.numVisibleNodes
 * records of `buffer`, 152 bytes each) is the host's own business: nothing in the frame depends on it. */
#define SIMLOD_PLANE_DEPTH       0u
#define SIMLOD_PLANE_SUMS        1u
#define SIMLOD_PLANE_FRAMEBUFFER 2u
#define SIMLOD_REDUCE_MIN        0u
#define SIMLOD_REDUCE_SUM        1u
typedef int (*SimlodReduceFn)(void* user, uint32_t plane, void* data, uint64_t count, uint32_t elemBytes, uint32_t op, void* stream);
int simlod_render_frame_composed(uint32_t* buffer, const SimlodUniforms* uniforms /*host*/, SimlodNode* nodes, uint32_t* colorbuffer,
                                 SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint, void* stream,
                                 SimlodReduceFn reduce, void* user);
/* ... with the reductions as ncclAllReduce calls on `ncclComm` (an ncclComm_t of RCCL: one rank per GPU over xGMI), enqueued on `stream`.
 * RCCL is looked up when this is first called — the copy already loaded in the process (the one that made `ncclComm`: a PyTorch process has its own
 * torch/lib/librccl.so), else dlopen("librccl.so"); the library itself does not link against it.  hipErrorNotSupported if there is none, or if
 * ncclGetVersion names a release outside 2.10 .. 2.x: the call passes ncclDataType_t / ncclRedOp_t by their 2.x numbers (ncclUint32 = 3,
 * ncclUint64 = 5, ncclSum = 0, ncclMin = 3).  simlod_rccl_version(): the NCCL_VERSION_CODE found (0: none). */
int simlod_render_frame_rccl(uint32_t* buffer, const SimlodUniforms* uniforms /*host*/, SimlodNode* nodes, uint32_t* colorbuffer,
                             SimlodStats* stats, uint64_t* frameStartTimestamp, void* cudaprint, void* stream, void* ncclComm);
int simlod_rccl_version(void);

/* ---- CudaModularProgram-shaped surface ------------------------------------------------------------------- */
typedef struct SimlodProgram SimlodProgram;
typedef struct SimlodFunction SimlodFunction;

/* Mirrors CudaModularProgram's constructor: module paths are matched by file name (reset.cu,
 * progressive_octree_voxels.cu, render.cu, utils.cu); the device code is precompiled for gfx950, nothing is
 * compiled at run time (colorfilter.cu -> `kernel` is known as well).  Unknown kernel names make the call fail with hipErrorNotFound. */
int simlod_program_create(SimlodProgram** out, const char* const* modules, int numModules,
                          const char* const* kernels, int numKernels);
void simlod_program_destroy(SimlodProgram* program);
/* program->kernels[name]; NULL when absent */
SimlodFunction* simlod_program_kernel(SimlodProgram* program, const char* name);
/* cuOccupancyMaxActiveBlocksPerMultiprocessor stand-in used by main_progressive_octree.cpp:494-496 */
int simlod_function_max_active_blocks(SimlodFunction* fn, int blockSize, int* numBlocks);
/* cuLaunchCooperativeKernel stand-in.  `args` holds pointers to the kernel arguments in declaration order, the
 * Uniforms struct by value (i.e. args[k] points at a host Uniforms).  The requested geometry is accepted and
 * ignored: the implementation sizes its own launches for the 256 CUs / 8 XCDs of the device. */
int simlod_launch_cooperative(SimlodFunction* fn, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by,
                              unsigned bz, unsigned sharedMemBytes, void* stream, void** args);

/* ---- measurement aid (not part of the reference surface) ------------------------------------------------------
 * When enabled, every internal kernel launch is bracketed by HIP events recorded on the launch stream, so that
 * bench.py can attribute device time to the kernels rocprofv3 lists.  Off by default (zero overhead). */
typedef struct SimlodProfileEntry {
	char     name[48];     /* internal kernel name, e.g. "k_insert", "r_draw<MODE_MIN64>" */
	uint32_t launches;
	uint32_t pad;
	double   total_ms;
} SimlodProfileEntry;
int simlod_profile_enable(int on);             /* 0 off | 1 every kernel (the builder's two streams become one) | 2 k_voxelize only, on its own stream: the builder's pipeline as in production */
int simlod_profile_collect(SimlodProfileEntry* out, int capacity, int* count);

/* ---- loader side -------------------------------------------------------------------------------
 * Replaces the parse loop of loadLasNative (modules/progressive_octree/LasLoader.cpp:169-227, called by the loader threads at
 * main_progressive_octree.cpp:866-870): `records` = numPoints raw LAS point records of bytesPerPoint bytes each, in DEVICE
 * memory (16-byte aligned), exactly the bytes the reference reads from offsetToPointData + bytesPerPoint * firstPoint;
 * `format` = LasHeader.format (RGB is taken for 2, 3, 5 and 7, as in the reference); scale = LasHeader.scale;
 * offset[k] = LasHeader.offset[k] + translation[k] (the sum the reference forms at LasLoader.cpp:197-199, translation = -boxMin).
 * out = numPoints Points, e.g. a slot of the batch ring.  Positions and r,g,b are bit-identical to the reference's; alpha,
 * which the reference leaves uninitialised, is 255.  Returns 0 or a hipError_t (invalid value: bytesPerPoint outside [12, 255],
 * RGB beyond the record, misaligned pointers). */
int simlod_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format,
                      const double scale[3], const double offset[3], SimlodPoint* out, void* stream);

/* ---- workload generator (BASELINE config 4; no counterpart in the reference) ------------------------------------------------------------
 * Points firstIndex .. firstIndex + numPoints - 1 of a procedurally generated, tiled terrain, written to `out` (device memory): the
 * stream is tile after tile (`pointsPerTile` points each, tiles laid out row-major with `tilesX` tiles per row, each tileExtent[0] x
 * tileExtent[1] metres, heights within [0, tileExtent[2])), inside a tile swath by swath like an airborne LAS scan.  One continuous
 * surface over all tiles; a pure function of (seed, index), so every rank of a multi-GPU job can generate any part of the stream. */
int simlod_generate_terrain(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed,
                            uint32_t tilesX, const float tileExtent[3], void* stream);
/* The same stream with flight lines: inside a tile the points come in strips of `swathWidth` metres (along x), each strip row by row —
 * what an airborne scanner with a finite swath writes (BASELINE config 3's stand-in file; simlod_amd/synthetic.terrain_scan is the host
 * twin).  swathWidth <= 0 or >= the tile's width: one strip, i.e. simlod_generate_terrain. */
int simlod_generate_terrain_scan(SimlodPoint* out, uint64_t numPoints, uint64_t firstIndex, uint64_t pointsPerTile, uint32_t seed,
                                 uint32_t tilesX, const float tileExtent[3], float swathWidth, void* stream);

/* Version / build info string (static storage). */
const char* simlod_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* SIMLOD_HIP_H */
