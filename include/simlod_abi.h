/* simlod_abi.h — memory-layout contract of the SimLOD octree hot path (clean-room restatement).
 *
 * Every struct here is shared, byte for byte, between
 *   - the host (which only ever touches Uniforms and Stats),
 *   - the HIP kernels in simlod_amd/csrc/,
 *   - the CPU oracle in oracle/,
 *   - and an octree image built by the reference's own kernels.
 *
 * Reference definitions this header restates (sizes/offsets are pinned by the static asserts
 * below and were measured by compiling the reference headers):
 *   Point, Chunk, OccupancyGrid, Node, constants .. modules/progressive_octree/structures.cuh:21-143
 *   mat4, Uniforms, Stats ....................... modules/progressive_octree/HostDeviceInterface.h:6-71
 *   AllocatorGlobal ............................. modules/progressive_octree/utils.h.cu:180-197
 *
 * Plain C99 / C++11; no CUDA or HIP types appear in it.
 */
#ifndef SIMLOD_ABI_H
#define SIMLOD_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define SIMLOD_STATIC_ASSERT(c, m) static_assert(c, m)
#else
#define SIMLOD_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* ---- constants (structures.cuh:21-28, progressive_octree_voxels.cu:21-22, reset.cu) ---------- */
#define SIMLOD_MAX_POINTS_PER_NODE 50000u     /* leaf spills when its arrival counter crosses this   */
#define SIMLOD_POINTS_PER_CHUNK    1000u      /* samples per linked-list chunk                       */
#define SIMLOD_GRID_SIZE           128u       /* voxel sampling grid per inner node: 128^3 bits       */
#define SIMLOD_GRID_NUM_CELLS      (128u * 128u * 128u)
#define SIMLOD_GRID_NUM_WORDS      (SIMLOD_GRID_NUM_CELLS / 32u)
#define SIMLOD_MAX_DEPTH           20         /* node-coordinate precision: 2^20 cells per axis       */
#define SIMLOD_BATCH_STREAM_SIZE   50u        /* ring slots of the upload ring                        */
#define SIMLOD_MAX_BATCH_SIZE      1000000u   /* points per ring slot (voxels.cu:881)                 */
#define SIMLOD_MAX_BATCHES_PER_LAUNCH 20u     /* voxels.cu:883                                        */
#define SIMLOD_MAX_EXPAND_ROUNDS   20         /* voxels.cu:394                                        */
#define SIMLOD_MAX_VISIBLE_NODES   100000u    /* render.cu:1108                                       */
#define SIMLOD_BACKGROUND_COLOR    0x00332211u /* render.cu:32                                        */
#define SIMLOD_CLEAR_PIXEL         ((0x7f800000ull << 32) | (uint64_t)SIMLOD_BACKGROUND_COLOR) /* render.cu:1130 */
#define SIMLOD_MEM_SAFETY_MARGIN   200000000ull /* voxels.cu:898                                      */
#define SIMLOD_MAX_PROCESSING_MS   10.0f      /* voxels.cu:22                                         */

typedef struct { float x, y, z; } simlod_float3;
typedef struct { float x, y, z, w; } __attribute__((aligned(16))) simlod_float4;   /* CUDA float4: 16-byte aligned */

/* structures.cuh:30-35 — XYZ + RGBA8, the 16-byte record of the .simlod format */
typedef struct SimlodPoint {
	float    x, y, z;
	uint32_t color;
} SimlodPoint;

/* structures.cuh:62-67 — `size` and `padding_0` are never written nor read by the reference kernels.
 * This implementation keeps, in the HEAD chunk of each list only, the address of the list's tail chunk
 * in those 8 bytes (see DESIGN.md "O(1) append"); all other chunks leave them untouched. */
typedef struct SimlodChunk {
	SimlodPoint         points[SIMLOD_POINTS_PER_CHUNK];
	int32_t             size;
	int32_t             padding_0;
	struct SimlodChunk* next;
} SimlodChunk;

/* structures.cuh:69-72 — bit index = x + 128*y + 128*128*z (voxels.cu:88-92) */
typedef struct SimlodOccupancyGrid {
	uint32_t values[SIMLOD_GRID_NUM_WORDS];
} SimlodOccupancyGrid;

/* structures.cuh:74-143 */
typedef struct SimlodNode {
	struct SimlodNode*   children[8];
	uint32_t             counter;        /* arrivals (stored + pending); spill trigger          */
	uint32_t             numPoints;      /* points stored in `points`                           */
	uint32_t             level;
	uint32_t             X, Y, Z;        /* cell coordinate at `level`                          */
	uint32_t             countIteration; /* == batchletIndex+1 of the last batch that counted    */
	uint32_t             countFlag;      /* unused by the reference                             */
	uint8_t              name[20];       /* 'r' + child digits                                  */
	uint8_t              visible;        /* written by render pass 1                            */
	uint8_t              isFiltered;
	uint8_t              isLeaf;         /* never maintained — use simlod_node_is_leaf()        */
	uint8_t              isLarge;        /* written by render pass 1                            */
	SimlodOccupancyGrid* grid;
	SimlodChunk*         points;
	SimlodChunk*         voxelChunks;
	uint32_t             numVoxels;
	uint32_t             numVoxelsStored;
} SimlodNode;

/* HostDeviceInterface.h:6-8 — rows[i] is matrix ROW i (host stores the transpose, main_progressive_octree.cpp:290-298) */
typedef struct SimlodMat4 { simlod_float4 rows[4]; } SimlodMat4;

/* HostDeviceInterface.h:10-44 — passed BY VALUE to every kernel */
typedef struct SimlodUniforms {
	float      width;
	float      height;
	float      time;
	float      fovy_rad;
	SimlodMat4 world;
	SimlodMat4 view;
	SimlodMat4 proj;
	SimlodMat4 transform;
	SimlodMat4 transform_updateBound;
	SimlodMat4 transformInv_updateBound;
	uint64_t   persistentBufferCapacity;
	uint64_t   momentaryBufferCapacity;
	uint64_t   frameCounter;
	simlod_float3 boxMin;
	simlod_float3 boxMax;
	uint8_t    showBoundingBox;
	uint8_t    showPoints;
	uint8_t    colorByNode;
	uint8_t    colorByLOD;
	uint8_t    colorWhite;
	uint8_t    doUpdateVisibility;
	uint8_t    doProgressive;
	float      LOD;
	uint8_t    useHighQualityShading;
	float      minNodeSize;
	int32_t    pointSize;
	uint8_t    updateStats;
	uint8_t    enableEDL;
	float      edlStrength;
} SimlodUniforms;

/* HostDeviceInterface.h:46-71 */
typedef struct SimlodStats {
	uint32_t frameID;
	uint32_t numNodes;
	uint32_t numInner;
	uint32_t numLeaves;
	uint32_t numNonemptyLeaves;
	uint32_t numPoints;
	uint32_t numVoxels;
	uint64_t allocatedBytes_momentary;
	uint64_t allocatedBytes_persistent;
	uint32_t numVisibleNodes;
	uint32_t numVisibleInner;
	uint32_t numVisibleLeaves;
	uint32_t numVisiblePoints;
	uint32_t numVisibleVoxels;
	uint32_t numChunksPoints;
	uint32_t numChunksVoxels;
	uint32_t batchletIndex;
	uint64_t numPointsProcessed;
	uint64_t numAllocatedChunks;
	uint64_t chunkPoolSize;
	uint32_t dbg;
	uint8_t  memCapacityReached;
} SimlodStats;

/* utils.h.cu:180-197 — lives at byte 0 of the persistent buffer; alloc(size) advances `offset`
 * by 16*((size+16)/16), i.e. always at least size+1 (utils.h.cu:190). */
typedef struct SimlodAllocatorGlobal {
	uint8_t* buffer;
	uint64_t offset;
} SimlodAllocatorGlobal;

#define SIMLOD_ALLOC_ROUND(size) (16ull * (((uint64_t)(size) + 16ull) / 16ull))

/* ---- layout pins ---------------------------------------------------------------------------- */
SIMLOD_STATIC_ASSERT(sizeof(SimlodPoint) == 16, "Point");
SIMLOD_STATIC_ASSERT(sizeof(SimlodChunk) == 16016, "Chunk");
SIMLOD_STATIC_ASSERT(offsetof(SimlodChunk, size) == 16000, "Chunk.size");
SIMLOD_STATIC_ASSERT(offsetof(SimlodChunk, next) == 16008, "Chunk.next");
SIMLOD_STATIC_ASSERT(sizeof(SimlodOccupancyGrid) == 262144, "OccupancyGrid");
SIMLOD_STATIC_ASSERT(sizeof(SimlodNode) == 152, "Node");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, counter) == 64, "Node.counter");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, numPoints) == 68, "Node.numPoints");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, level) == 72, "Node.level");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, X) == 76, "Node.X");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, countIteration) == 88, "Node.countIteration");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, countFlag) == 92, "Node.countFlag");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, name) == 96, "Node.name");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, visible) == 116, "Node.visible");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, isLarge) == 119, "Node.isLarge");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, grid) == 120, "Node.grid");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, points) == 128, "Node.points");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, voxelChunks) == 136, "Node.voxelChunks");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, numVoxels) == 144, "Node.numVoxels");
SIMLOD_STATIC_ASSERT(offsetof(SimlodNode, numVoxelsStored) == 148, "Node.numVoxelsStored");
SIMLOD_STATIC_ASSERT(sizeof(SimlodMat4) == 64, "mat4");
SIMLOD_STATIC_ASSERT(sizeof(SimlodUniforms) == 480, "Uniforms");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, world) == 16, "Uniforms.world");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, transform) == 208, "Uniforms.transform");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, transform_updateBound) == 272, "Uniforms.transform_updateBound");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, persistentBufferCapacity) == 400, "Uniforms.persistentBufferCapacity");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, frameCounter) == 416, "Uniforms.frameCounter");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, boxMin) == 424, "Uniforms.boxMin");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, boxMax) == 436, "Uniforms.boxMax");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, showBoundingBox) == 448, "Uniforms.showBoundingBox");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, LOD) == 456, "Uniforms.LOD");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, useHighQualityShading) == 460, "Uniforms.useHighQualityShading");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, minNodeSize) == 464, "Uniforms.minNodeSize");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, pointSize) == 468, "Uniforms.pointSize");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, updateStats) == 472, "Uniforms.updateStats");
SIMLOD_STATIC_ASSERT(offsetof(SimlodUniforms, edlStrength) == 476, "Uniforms.edlStrength");
SIMLOD_STATIC_ASSERT(sizeof(SimlodStats) == 112, "Stats");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, allocatedBytes_momentary) == 32, "Stats.allocatedBytes_momentary");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, numVisibleNodes) == 48, "Stats.numVisibleNodes");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, batchletIndex) == 76, "Stats.batchletIndex");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, numPointsProcessed) == 80, "Stats.numPointsProcessed");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, numAllocatedChunks) == 88, "Stats.numAllocatedChunks");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, chunkPoolSize) == 96, "Stats.chunkPoolSize");
SIMLOD_STATIC_ASSERT(offsetof(SimlodStats, memCapacityReached) == 108, "Stats.memCapacityReached");
SIMLOD_STATIC_ASSERT(sizeof(SimlodAllocatorGlobal) == 16, "AllocatorGlobal");

#endif /* SIMLOD_ABI_H */
