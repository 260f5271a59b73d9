// simlod_internal.hpp — declarations shared by the translation units of libsimlod_hip.so (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simlod_abi.h"

namespace simlod {

static constexpr uint32_t CHUNK_QUEUE_CAPACITY = 1000000u;   // progressive_octree_voxels.cu:856
static constexpr uint32_t SPILLING_CAPACITY = 100000u;       // progressive_octree_voxels.cu:847

// Control block at byte 0 of kernel_construct's momentary buffer.  Lives only for the duration of one launch
// (the recycle stack behind it, like the reference's chunkQueue, must survive between launches).
struct Ctl {
	uint32_t uploaded, firstBatch, numBatches, stop;
	uint32_t active, batchSize, ringSlot, batchIndex;
	uint32_t numSpilling;        // spilling leaves found by k_count; NOT modified by k_expand (its early-exit test must be stable)
	uint32_t numSpilled, dirCount, errors;
	uint32_t ordinal, abortBatch, barrierCount;
	uint32_t rebuildLeafChunks;  // this launch found the leaf chunk table stale (first launch, reset, wiped momentary buffer): k_parents refills it
	uint32_t roundSpill[2];      // spilling leaves found by expand round r live in roundSpill[r & 1]
	uint32_t numWork;            // spill-copy work items appended so far in this batch (monotonic)
	uint32_t pad1;
	uint32_t spilledSnap[2];     // numSpilled / numWork as they were BEFORE round r's split phase: slot [r & 1]
	uint32_t workSnap[2];
	uint64_t startNs;
	uint32_t statCounters[8];
	unsigned long long reserve;        // k_expand: nodes in use << 32 | spilled points of this batch — ONE word, so a split reserves both or neither
	uint32_t tableMagic, tableBatch;   // leaf chunk table is valid for the octree as it was after batch #tableBatch (k_finish)
	uint64_t expandNs[8];              // byte 152: k_expand phase times of workgroup 0 (split, barrier, copy, recount, barrier, rounds, calls; tools/kprof.py), [7] = spilled points so far (bench.py)
};

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
	uint64_t     offQueue, offSpillA, offSpillB, offSplitTag, offRetryTag, offParent, offNodeDir, offChunkDir, offLeafChunks, offPaths, offWork, offLeafOf, offWin, offSpilled;
	uint32_t     nodeCapacity, spilledCap, dirCap, workCap;
};

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
	} while (0)
uint32_t node_capacity();
int tune(const char* envName, int dflt);     // integer tuning knob from the environment (read once per call site)

bool layout_construct(BuildArgs& a, uint64_t capacity);
int launch_construct(const SimlodUniforms* u, SimlodPoint* points, uint32_t* buffer, uint8_t* pers, SimlodNode* nodes,
                     SimlodStats* stats, uint64_t* frameStart, uint32_t* numBatchesUploaded, uint32_t* batchSizes, hipStream_t stream);
int launch_reset(const SimlodUniforms* u, uint8_t* pers, SimlodNode* nodes, SimlodStats* stats, uint32_t* numBatchesUploaded,
                 uint32_t* batchSizes, hipStream_t stream);
int launch_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format, const double* scale,
                      const double* offset, SimlodPoint* out, hipStream_t stream);
enum : uint32_t { RENDER_FIRST = 1u, RENDER_COLOR = 2u, RENDER_RESOLVE = 4u, RENDER_OUTPUT = 8u, RENDER_ALL = 15u };
int launch_render(uint32_t* buffer, const SimlodUniforms* u, SimlodNode* nodes, uint32_t* colorbuffer, SimlodStats* stats,
                  uint64_t* frameStart, hipStream_t stream, uint32_t parts);
uint64_t render_depth_plane_offset(uint32_t width, uint32_t height);
uint64_t render_sum_planes_offset(uint32_t width, uint32_t height);

}  // namespace simlod
