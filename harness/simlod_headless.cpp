// simlod_headless.cpp — headless replay of the reference host's launch sequence on MI355X.
//
// What modules/progressive_octree/main_progressive_octree.cpp does around its three kernels, without window, GL or ImGui,
// written against the SAME driver-API calls (shim/cuda.h, shim/CudaModularProgram.h):
//
//   initCuda()          main.cpp:272-281     context, upload stream, CU count
//   initCudaProgram()   main.cpp:549-642     device buffers with the reference's sizes, three CudaModularPrograms
//   getUniforms()       main.cpp:283-331     Uniforms from a camera (orbit controls, include/OrbitControls.h:140-159); in the _ref builds the
//                                            reference's own text over its vendored glm (see SIMLOD_REF_UNIFORMS_EXTRACT below)
//   resetCUDA()         main.cpp:333-361     `kernel`
//   uploader            main.cpp:963-1063    pinned batch -> ring slot, batchSizes[slot], numBatchesUploaded on stream_upload,
//                                            back-pressure: at most 50 M points ahead of Stats.numPointsProcessed (:1012)
//   updateOctree()      main.cpp:364-428     `kernel_construct`, numSMs x 256 cooperative geometry
//   renderCUDA()        main.cpp:465-546     `kernel_render`, occupancy x numSMs workgroups; linear RGBA8 instead of a GL surface
//   frame loop          main.cpp:1159-1226   render, update, async Stats copy, until numPointsProcessed == numPointsTotal
//
// Built a second time as harness/_ref/ref_host_replay (make ref_host, only where /root/reference exists): resetCUDA, updateOctree,
// renderCUDA and initCudaProgram are then the REFERENCE'S OWN TEXT, cut by line range out of main_progressive_octree.cpp at build
// time (-DSIMLOD_REF_HOST_EXTRACT=...; nothing of it is stored in this repository) and compiled against shim/cuda.h — the proof that
// the unchanged host functions drive libsimlod_hip.so.
//
// Usage: simlod_headless <file.simlod | synthetic:N> [out.ppm] [width height]
// Prints the Stats the reference shows in its UI and the update/render kernel timings of "benchmark mode" (main.cpp:411-422).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <iterator>
#include <memory>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "CudaModularProgram.h"
#include "cuda.h"
#include "simlod_abi.h"
#ifdef SIMLOD_REF_UNIFORMS_EXTRACT
#include <glm/glm.hpp>                 // the reference's vendored copy (libs/glm, header-only), -DGLM_FORCE_CTOR_INIT: see getUniforms below
#endif

using namespace std;
using Point = SimlodPoint;
using Uniforms = SimlodUniforms;
using Stats = SimlodStats;

// what the host functions see of the GL renderer (src/GLRenderer.h): the colour attachment they hand to CUDA
struct GLTexture { GLuint handle = 1; };
struct GLFramebuffer { vector<shared_ptr<GLTexture>> colorAttachments; };
struct GLView { shared_ptr<GLFramebuffer> framebuffer; };
#ifdef SIMLOD_REF_UNIFORMS_EXTRACT
struct Camera { glm::dmat4 view, proj; double fovy = 60.0; };      // include/GLRenderer.h:130-163, what getUniforms reads of it
struct GLRenderer { GLView view; int width = 0, height = 0, frameCount = 0; shared_ptr<Camera> camera = make_shared<Camera>(); };
#else
struct GLRenderer { GLView view; int width = 0, height = 0, frameCount = 0; };
#endif
template <class... A> static void printfmt(const char*, A&&...) {}      // the reference's fmt-style log lines are dropped

constexpr uint64_t BATCH_STREAM_SIZE = SIMLOD_BATCH_STREAM_SIZE;
constexpr uint64_t MAX_BATCH_SIZE = SIMLOD_MAX_BATCH_SIZE;

static CUdevice device;
static CUcontext context;
static int numSMs;
static CUstream stream_upload;
static CUdeviceptr cptr_buffer, cptr_buffer_persistent, cptr_nodes, cptr_renderbuffer, cptr_stats, cptr_numBatchesUploaded, cptr_batchSizes,
    cptr_frameStart, cptr_colorbuffer;
static struct { CUdeviceptr cptr = 0; } cudaprint;      // CudaPrint's device side is a no-op (modules/CudaPrint/CudaPrint.cuh:49-51)
static CUgraphicsResource cugl_colorbuffer;
static CUdeviceptr cptr_points_ring[BATCH_STREAM_SIZE];
static CUevent ce_render_start, ce_render_end, ce_update_start, ce_update_end;
static CudaModularProgram *cuda_program_update, *cuda_program_render, *cuda_program_reset;
static uint64_t momentaryBufferCapacity, persistentBufferCapacity, frameCounter = 0;
static Stats stats;
static Stats* h_stats_pinned;
static simlod_float3 boxSize;
static int width = 1920, height = 1080;
static float viewProj[16];   // row-major world-view-projection (what glm::transpose leaves in Uniforms.transform)
static double cameraView[16], cameraProj[16];   // row-major view and projection (fp64, as the reference's Camera keeps them)

struct {
	bool useHighQualityShading = true;
	bool showBoundingBox = false;
	bool doUpdateVisibility = true;
	bool showPoints = true;
	bool colorByNode = false;
	bool colorByLOD = false;
	bool colorWhite = false;
	bool benchmarkRendering = false;
	float LOD = 0.2f;
	float minNodeSize = 64.0f;
	int pointSize = 1;
	bool enableEDL = true;
	float edlStrength = 0.8f;
} settings;   // main.cpp:123-139
static bool requestBenchmark = true;                           // "benchmark mode": kernel durations are accumulated (main.cpp:411-422)
static std::atomic_bool requestStep = false;
static double kernelUpdateDuration = 0, minKernelUpdateDuration = 1e30, maxKernelUpdateDuration = 0, avgKernelUpdateDuration = 0, cntKernelUpdateDuration = 0;
static double kernelRenderDuration = 0, minKernelRenderDuration = 1e30, maxKernelRenderDuration = 0, avgKernelRenderDuration = 0, cntKernelRenderDuration = 0;
static float renderingDuration = 0;
static int numBatchesProcessed = 0;

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void initCuda() {   // main.cpp:272-281
	cuInit(0);
	cuDeviceGet(&device, 0);
	cuCtxCreate(&context, 0, device);
	cuStreamCreate(&stream_upload, CU_STREAM_NON_BLOCKING);
	cuCtxGetDevice(&device);
	cuDeviceGetAttribute(&numSMs, CU_DEVICE_ATTRIBUTE_MULTIPROCESSOR_COUNT, device);
}

#ifdef SIMLOD_REF_UNIFORMS_EXTRACT
Uniforms getUniforms(shared_ptr<GLRenderer> renderer);      // (defined by the reference's text)
#else
static Uniforms getUniforms(shared_ptr<GLRenderer> renderer);
#endif

#ifdef SIMLOD_REF_UPLOADER_EXTRACT
// ---- SIMLOD_REF_UPLOADER_EXTRACT: the reference's own PinnedMemorySlot / PinnedMemPool (main.cpp:48-55, 141-222), reset() (:775-809) and
// spawnUploader() (:963-1063) — the code that races with kernel_construct — cut out of the checkout at build time like the
// four host functions.  What they need around them is declared here the way main.cpp declares it (:67-93, :225-262); the loader threads
// (spawnLoader :811-958) and reload() (:644-773) are restated further down: they read files, the uploader and the kernels never see how.
constexpr uint64_t PINNED_MEM_POOL_SIZE = 200;
static CUstream stream_download;
#include SIMLOD_REF_TYPES_EXTRACT     // PinnedMemorySlot, PinnedMemPool
struct PointBatch {                   // main.cpp:67-75 (file / LAS header: the restated loader keeps them to itself)
	int first = 0;
	int count = 0;
	PinnedMemorySlot pinnedMem;
};
static deque<PointBatch> batchesInPinnedMemory;
static deque<PinnedMemorySlot> pinnedMemoryInUpload;
static atomic_bool resetInProgress;
static mutex mtx_uploader;
static vector<unique_ptr<mutex>> mtx_loader;
static mutex mtx_batchesInPinnedMemory;
static mutex mtx_pinnedMemoryInUpload;
static int batchStreamUploadIndex = 0;
static bool requestReset = false;
static atomic_bool requestStepthrough = false;
static uint32_t numPointsUploaded = 0;
static int numBatchesTotal = 0;
static bool lastBatchFinishedDevice = false;
static PinnedMemPool pinnedMemPool;
static void reload();
void resetCUDA(shared_ptr<GLRenderer> renderer);      // (defined by the host-function extract)
template <class T> static void setThreadPriorityHigh(T&) {}   // unsuck.hpp: a Windows scheduling hint
#endif

#ifdef SIMLOD_REF_HOST_EXTRACT
#include SIMLOD_REF_HOST_EXTRACT      // resetCUDA, updateOctree, renderCUDA, initCudaProgram: the reference's own lines
#else
static void initCudaProgram(shared_ptr<GLRenderer> renderer) {   // main.cpp:549-642
	uint64_t nodesCapacity = 200000, estimatedNodeSize = 200;
	uint64_t cptr_buffer_bytes = 300000000, cptr_nodes_bytes = nodesCapacity * estimatedNodeSize, cptr_renderbuffer_bytes = 200000000;
	momentaryBufferCapacity = cptr_buffer_bytes;
	cuMemAlloc(&cptr_buffer, cptr_buffer_bytes);
	cuMemAlloc(&cptr_nodes, cptr_nodes_bytes);
	cuMemAlloc(&cptr_renderbuffer, cptr_renderbuffer_bytes);
	cuMemAlloc(&cptr_stats, sizeof(Stats));
	cuMemAlloc(&cptr_numBatchesUploaded, 4);
	cuMemAlloc(&cptr_batchSizes, 4 * BATCH_STREAM_SIZE);
	cuMemAlloc(&cptr_frameStart, 8);
	cuMemAllocHost((void**)&h_stats_pinned, sizeof(Stats));
	uint64_t cptr_points_bytes = MAX_BATCH_SIZE * sizeof(Point);
	CUdeviceptr devicemem = 0;
	cuMemAlloc(&devicemem, BATCH_STREAM_SIZE * cptr_points_bytes);
	for (uint64_t i = 0; i < BATCH_STREAM_SIZE; i++) cptr_points_ring[i] = devicemem + i * cptr_points_bytes;
	size_t availableMem = 0, totalMem = 0;
	cuMemGetInfo(&availableMem, &totalMem);
	size_t cptr_buffer_persistent_bytes = std::min<size_t>((size_t)((double)availableMem * 0.80), (size_t)64 << 30);
	persistentBufferCapacity = cptr_buffer_persistent_bytes;
	void* p = nullptr;   // 80 % of a 288 GB device: not zero-filled (the allocator header is written by `kernel`)
	if (hipMalloc(&p, cptr_buffer_persistent_bytes) != hipSuccess) { std::fprintf(stderr, "persistent buffer allocation failed\n"); std::exit(1); }
	cptr_buffer_persistent = (CUdeviceptr)(uintptr_t)p;

	cuda_program_update = new CudaModularProgram({.modules = {"./modules/progressive_octree/progressive_octree_voxels.cu", "./modules/progressive_octree/utils.cu"},
	                                              .kernels = {"kernel_construct"}});
	cuda_program_render = new CudaModularProgram({.modules = {"./modules/progressive_octree/render.cu", "./modules/progressive_octree/utils.cu"},
	                                              .kernels = {"kernel_render"}});
	cuda_program_reset = new CudaModularProgram({.modules = {"./modules/progressive_octree/reset.cu", "./modules/progressive_octree/utils.cu"},
	                                             .kernels = {"kernel"}});
	cuEventCreate(&ce_render_start, 0); cuEventCreate(&ce_render_end, 0);
	cuEventCreate(&ce_update_start, 0); cuEventCreate(&ce_update_end, 0);
	cuGraphicsGLRegisterImage(&cugl_colorbuffer, renderer->view.framebuffer->colorAttachments[0]->handle, GL_TEXTURE_2D, CU_GRAPHICS_REGISTER_FLAGS_WRITE_DISCARD);
}
#endif

#ifdef SIMLOD_REF_UNIFORMS_EXTRACT
// getUniforms as the reference wrote it (main.cpp:283-331, cut at build time like the other host functions).  One thing has to be decided
// for it: it multiplies by a default-constructed `glm::mat4 world`, which the vendored glm 0.9.9 leaves UNINITIALISED unless
// GLM_FORCE_CTOR_INIT is defined; the reference's tree does not define it and multiplies by whatever its stack holds.  The _ref builds
// define it: `world` is the identity — the one value with which the author's proj * view * world is the camera's transform.  (The octree
// does not depend on the camera; the frames do.)
static glm::mat4 transform_updatebound;                     // main.cpp:116
#define float3 simlod_float3                                // (Uniforms.boxMin / boxMax are this layout struct here, HIP's float3 is another type)
#include SIMLOD_REF_UNIFORMS_EXTRACT
#undef float3
#else
static Uniforms getUniforms(shared_ptr<GLRenderer>) {   // main.cpp:283-331
	Uniforms u;
	std::memset(&u, 0, sizeof(u));
	const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	std::memcpy(&u.world, ident, 64); std::memcpy(&u.view, ident, 64); std::memcpy(&u.proj, ident, 64);
	std::memcpy(&u.transform, viewProj, 64);
	std::memcpy(&u.transform_updateBound, viewProj, 64);
	std::memcpy(&u.transformInv_updateBound, ident, 64);
	u.width = (float)width; u.height = (float)height;
	u.fovy_rad = 3.1415f * 60.0f / 180.0f;
	u.boxMin = {0.0f, 0.0f, 0.0f};
	u.boxMax = boxSize;
	u.frameCounter = frameCounter;
	u.showBoundingBox = settings.showBoundingBox;
	u.doUpdateVisibility = true;
	u.showPoints = settings.showPoints;
	u.LOD = 0.2f;
	u.minNodeSize = settings.minNodeSize;
	u.pointSize = settings.pointSize;
	u.useHighQualityShading = settings.useHighQualityShading;
	u.persistentBufferCapacity = persistentBufferCapacity;
	u.momentaryBufferCapacity = momentaryBufferCapacity;
	u.enableEDL = true;
	u.edlStrength = 0.8f;
	return u;
}
#endif

#ifndef SIMLOD_REF_HOST_EXTRACT
static void resetCUDA(shared_ptr<GLRenderer> renderer) {   // main.cpp:333-361
	Uniforms uniforms = getUniforms(renderer);
	void* args[] = {&uniforms, &cptr_buffer_persistent, &cptr_nodes, &cptr_stats, &cudaprint.cptr, &cptr_numBatchesUploaded, &cptr_batchSizes};
	auto res_launch = cuLaunchCooperativeKernel(cuda_program_reset->kernels["kernel"], 1, 1, 1, 1, 1, 1, 0, 0, args);
	if (res_launch != CUDA_SUCCESS) std::printf("CUDA kernel 'reset' failed.\n");
	cuCtxSynchronize();
}

static void updateOctree(shared_ptr<GLRenderer> renderer) {   // main.cpp:364-428
	Uniforms uniforms = getUniforms(renderer);
	int workgroupSize = 256, numGroups = 1 * numSMs;
	auto ptrPoints = cptr_points_ring[0];
	void* args[] = {&uniforms, &ptrPoints, &cptr_buffer, &cptr_buffer_persistent, &cptr_nodes, &cptr_stats, &cptr_frameStart, &cudaprint.cptr,
	                &cptr_numBatchesUploaded, &cptr_batchSizes};
	cuEventRecord(ce_update_start, 0);
	auto res_launch = cuLaunchCooperativeKernel(cuda_program_update->kernels["kernel_construct"], numGroups, 1, 1, workgroupSize, 1, 1, 0, 0, args);
	if (res_launch != CUDA_SUCCESS) { const char* str; cuGetErrorString(res_launch, &str); std::printf("error: %s \n", str); }
	cuEventRecord(ce_update_end, 0);
	cuCtxSynchronize();   // benchmark mode, main.cpp:411-422
	float duration;
	cuEventElapsedTime(&duration, ce_update_start, ce_update_end);
	kernelUpdateDuration += duration;
	cntKernelUpdateDuration += 1.0;
}

static void renderCUDA(shared_ptr<GLRenderer> renderer) {   // main.cpp:465-546
	Uniforms uniforms = getUniforms(renderer);
	// the colour attachment, through the interop calls the reference makes per frame (main.cpp:472-486)
	vector<CUgraphicsResource> dynamic_resources = {cugl_colorbuffer};
	cuGraphicsMapResources((unsigned)dynamic_resources.size(), dynamic_resources.data(), (CUstream)CU_STREAM_DEFAULT);
	CUDA_RESOURCE_DESC res_desc = {};
	res_desc.resType = CU_RESOURCE_TYPE_ARRAY;
	cuGraphicsSubResourceGetMappedArray(&res_desc.res.array.hArray, cugl_colorbuffer, 0, 0);
	CUsurfObject output_surf;
	cuSurfObjectCreate(&output_surf, &res_desc);
	int workgroupSize = 256, numGroups;
	cuOccupancyMaxActiveBlocksPerMultiprocessor(&numGroups, cuda_program_render->kernels["kernel_render"], workgroupSize, 0);
	numGroups *= numSMs;
	void* args[] = {&cptr_renderbuffer, &uniforms, &cptr_nodes, &output_surf, &cptr_stats, &cptr_frameStart, &cudaprint.cptr};
	cuEventRecord(ce_render_start, 0);
	auto res_launch = cuLaunchCooperativeKernel(cuda_program_render->kernels["kernel_render"], numGroups, 1, 1, workgroupSize, 1, 1, 0, 0, args);
	if (res_launch != CUDA_SUCCESS) { const char* str; cuGetErrorString(res_launch, &str); std::printf("error: %s \n", str); }
	cuEventRecord(ce_render_end, 0);
	cuCtxSynchronize();
	float duration;
	cuEventElapsedTime(&duration, ce_render_start, ce_render_end);
	kernelRenderDuration += duration;
	cntKernelRenderDuration += 1.0;
	cuSurfObjectDestroy(output_surf);
	cuGraphicsUnmapResources((unsigned)dynamic_resources.size(), dynamic_resources.data(), (CUstream)CU_STREAM_DEFAULT);
}
#endif

#ifdef SIMLOD_REF_UPLOADER_EXTRACT
#include SIMLOD_REF_UPLOADER_EXTRACT  // reset(), spawnUploader(): the reference's own lines
// reload() (main.cpp:644-773), what is left of it without files to list: the counters the uploader and the frame loop look at (:764-772)
static uint64_t reloadNumBatches = 0;
static void reload() {
	numBatchesTotal = (int)reloadNumBatches;
	stats = Stats();
	numPointsUploaded = 0;
	numBatchesProcessed = 0;
	lastBatchFinishedDevice = false;
	batchStreamUploadIndex = 0;
}
#endif

// ---- camera: OrbitControls::update (include/OrbitControls.h:140-159) + glm::perspective + main.cpp:286-298 ------------------
static void mul4(const double a[16], const double b[16], double out[16]) {
	for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { double s = 0; for (int k = 0; k < 4; k++) s += a[4 * i + k] * b[4 * k + j]; out[4 * i + j] = s; }
}

static void setCamera(double yaw, double pitch, double radius, const double target[3]) {
	const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
	// world = T(target) * Rz(yaw) * Rx(pitch) * flip * T(0,0,radius); its rotation part R and translation t:
	const double Rz[16] = {cy, -sy, 0, 0, sy, cy, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
	const double Rx[16] = {1, 0, 0, 0, 0, cp, -sp, 0, 0, sp, cp, 0, 0, 0, 0, 1};
	const double flip[16] = {1, 0, 0, 0, 0, 0, -1, 0, 0, 1, 0, 0, 0, 0, 0, 1};
	double a[16], R[16];
	mul4(Rz, Rx, a); mul4(a, flip, R);
	const double eye[3] = {target[0] + R[2] * radius, target[1] + R[6] * radius, target[2] + R[10] * radius};
	double view[16] = {0};   // inverse of a rigid transform: R^T, -R^T eye
	for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) view[4 * i + j] = R[4 * j + i]; view[4 * i + 3] = -(R[i] * eye[0] + R[4 + i] * eye[1] + R[8 + i] * eye[2]); }
	view[15] = 1;
	const double fovy = 3.14159265358979323846 * 60.0 / 180.0, aspect = (double)width / height, zn = 0.1, zf = 2000000.0, t = std::tan(fovy / 2);
	double proj[16] = {0};
	proj[0] = 1 / (aspect * t); proj[5] = 1 / t; proj[10] = -(zf + zn) / (zf - zn); proj[11] = -(2 * zf * zn) / (zf - zn); proj[14] = -1;
	float v32[16], p32[16];
	for (int i = 0; i < 16; i++) { v32[i] = (float)view[i]; p32[i] = (float)proj[i]; cameraView[i] = view[i]; cameraProj[i] = proj[i]; }
	for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { float s = 0; for (int k = 0; k < 4; k++) s += p32[4 * i + k] * v32[4 * k + j]; viewProj[4 * i + j] = s; }
}

int main(int argc, char** argv) {
	if (argc < 2) { std::printf("usage: %s <file.simlod | file.las | synthetic:N> [out.ppm] [width height]\n", argv[0]); return 2; }
	const std::string path = argv[1];
	const char* outPath = argc > 2 ? argv[2] : nullptr;
	if (argc > 4) { width = std::atoi(argv[3]); height = std::atoi(argv[4]); }

	// reload(): a .simlod file is a 24-byte bounding box followed by 16-byte XYZRGBA records (main.cpp:722-745, tools/las2simlod.mjs:96-147);
	// a .las file is kept as raw records and decoded on the device (loadHeader LasLoader.h:21-55; simlod_decode_las replaces the
	// loader threads' parse loop LasLoader.cpp:169-227)
	std::vector<Point> points;
	std::vector<uint8_t> lasRecords;
	uint32_t lasBytesPerPoint = 0, lasFormat = 0;
	double lasScale[3] = {0, 0, 0}, lasOffset[3] = {0, 0, 0};
	uint64_t lasNumPoints = 0;
	const bool isLas = path.size() > 4 && (path.substr(path.size() - 4) == ".las" || path.substr(path.size() - 4) == ".LAS");
	if (isLas) {
		FILE* f = std::fopen(path.c_str(), "rb");
		if (!f) { std::perror(path.c_str()); return 1; }
		uint8_t h[375] = {0};
		if (std::fread(h, 1, 375, f) < 227) { std::fprintf(stderr, "short LAS header\n"); return 1; }
		auto rd = [&](auto& v, size_t off) { std::memcpy(&v, h + off, sizeof(v)); };
		uint8_t vMajor = h[24], vMinor = h[25]; uint32_t offsetToPointData, legacyCount; uint16_t bpp; uint64_t count64;
		rd(offsetToPointData, 96); lasFormat = h[104]; rd(bpp, 105); rd(legacyCount, 107); rd(count64, 247);
		lasBytesPerPoint = bpp;
		lasNumPoints = (vMajor == 1 && vMinor <= 3) ? legacyCount : count64;
		double mn[3], mx[3];
		for (int k = 0; k < 3; k++) { rd(lasScale[k], 131 + 8 * k); rd(lasOffset[k], 155 + 8 * k); rd(mx[k], 179 + 16 * k); rd(mn[k], 187 + 16 * k); }
		for (int k = 0; k < 3; k++) lasOffset[k] += -mn[k];                                  // translation = -boxMin, main.cpp:868, LasLoader.cpp:197-199
		boxSize = {(float)(mx[0] - mn[0]), (float)(mx[1] - mn[1]), (float)(mx[2] - mn[2])};
		lasRecords.resize((size_t)lasNumPoints * lasBytesPerPoint);
		std::fseek(f, (long)offsetToPointData, SEEK_SET);
		if (std::fread(lasRecords.data(), 1, lasRecords.size(), f) != lasRecords.size()) { std::fprintf(stderr, "short LAS point data\n"); return 1; }
		std::fclose(f);
	} else if (path.rfind("synthetic:", 0) == 0) {
		const size_t n = std::strtoull(path.c_str() + 10, nullptr, 10);
		points.resize(n);
		uint32_t s = 1234567u;
		auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
		for (auto& p : points) { p.x = rnd(); p.y = rnd(); const float h = 0.25f + 0.2f * std::sin(6.0f * p.x) * std::cos(5.0f * p.y); p.z = h + 0.002f * rnd();
			p.color = (uint32_t)(255 * p.x) | ((uint32_t)(255 * p.y) << 8) | ((uint32_t)(255 * p.z * 2) << 16) | (255u << 24); }
		boxSize = {1.0f, 1.0f, 1.0f};
	} else {
		FILE* f = std::fopen(path.c_str(), "rb");
		if (!f) { std::perror(path.c_str()); return 1; }
		float bbox[6];
		if (std::fread(bbox, 4, 6, f) != 6) { std::fprintf(stderr, "short header\n"); return 1; }
		std::fseek(f, 0, SEEK_END); const long bytes = std::ftell(f); std::fseek(f, 24, SEEK_SET);
		points.resize((size_t)(bytes - 24) / 16);
		if (std::fread(points.data(), 16, points.size(), f) != points.size()) { std::fprintf(stderr, "short read\n"); return 1; }
		std::fclose(f);
		boxSize = {bbox[3] - bbox[0], bbox[4] - bbox[1], bbox[5] - bbox[2]};
		for (auto& p : points) { p.x -= bbox[0]; p.y -= bbox[1]; p.z -= bbox[2]; }   // main.cpp:868
	}
	const uint64_t numPointsTotal = isLas ? lasNumPoints : points.size();
	const uint64_t numBatchesTotal = (numPointsTotal + MAX_BATCH_SIZE - 1) / MAX_BATCH_SIZE;

	initCuda();
	auto renderer = make_shared<GLRenderer>();               // headless: a colour attachment that is a linear RGBA8 image on the device
	renderer->width = width; renderer->height = height;
	renderer->view.framebuffer = make_shared<GLFramebuffer>();
	renderer->view.framebuffer->colorAttachments.push_back(make_shared<GLTexture>());
	cuMemAlloc(&cptr_colorbuffer, (size_t)width * height * 4);
	simlod_shim_register_surface(renderer->view.framebuffer->colorAttachments[0]->handle, cptr_colorbuffer, (size_t)width * height * 4);
	initCudaProgram(renderer);
	const double target[3] = {boxSize.x * 0.5, boxSize.y * 0.5, boxSize.z * 0.3};
	setCamera(-0.207, -0.797, 1.1 * std::max(boxSize.x, std::max(boxSize.y, boxSize.z)), target);
#ifdef SIMLOD_REF_UNIFORMS_EXTRACT
	for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { renderer->camera->view[c][r] = cameraView[4 * r + c]; renderer->camera->proj[c][r] = cameraProj[4 * r + c]; }   // glm: column-major
#endif
	resetCUDA(renderer);

#ifdef SIMLOD_REF_UPLOADER_EXTRACT
	// ---- the reference's uploader thread (spawnUploader, its own text) fed by a restated loader thread (spawnLoader, main.cpp:811-958):
	// acquire a pinned slot from the reference's pool, fill it with the batch's points — a LAS batch is parsed on the CPU here, as
	// loadLasNative does (LasLoader.cpp:169-227) — and queue it; the uploader copies it into the ring and publishes it (H10)
	const bool sourcePinned = false;
	pinnedMemPool.reserveSlots(24);
	mtx_loader.push_back(make_unique<mutex>());
	reloadNumBatches = numBatchesTotal;
	spawnUploader(renderer);                                     // idles until reload() announces batches
	reset(renderer);                                             // locks everybody out, resetCUDA, Stats read-back, reload()
	std::atomic<bool> quitUploader{false};
	std::thread uploader([&]() {                                 // (the loader; the variable keeps its name for the code below)
		for (uint64_t index = 0; index < numBatchesTotal && !quitUploader.load(); index++) {
			for (;;) {                                           // MAX_LOADQUEUE_SIZE, main.cpp:37: do not run ahead of the uploader without bound
				lock_guard<mutex> lock(mtx_batchesInPinnedMemory);
				if (batchesInPinnedMemory.size() < 16) break;
				std::this_thread::sleep_for(std::chrono::microseconds(50));
			}
			lock_guard<mutex> lock_loader(*mtx_loader[0]);
			PinnedMemorySlot slot = pinnedMemPool.acquire();
			Point* dst = (Point*)slot.memLocation;
			const uint64_t first = index * MAX_BATCH_SIZE;
			const uint32_t count = (uint32_t)std::min<uint64_t>(MAX_BATCH_SIZE, numPointsTotal - first);
			if (isLas) {
				const uint32_t rgbOffset = lasFormat == 2 ? 20u : (lasFormat == 3 || lasFormat == 5) ? 28u : lasFormat == 7 ? 30u : 0u;   // LasLoader.cpp:177-185
				for (uint32_t i = 0; i < count; i++) {
					const uint8_t* rec = lasRecords.data() + (first + i) * lasBytesPerPoint;
					int32_t X, Y, Z;
					std::memcpy(&X, rec, 4); std::memcpy(&Y, rec + 4, 4); std::memcpy(&Z, rec + 8, 4);
					double x = (double)X * lasScale[0]; x = x + lasOffset[0];          // LasLoader.cpp:212-214 (offset and translation added up front)
					double y = (double)Y * lasScale[1]; y = y + lasOffset[1];
					double z = (double)Z * lasScale[2]; z = z + lasOffset[2];
					Point p; p.x = (float)x; p.y = (float)y; p.z = (float)z;
					uint32_t color = 0xff000000u;
					if (rgbOffset > 0) {                                               // LasLoader.cpp:216-221
						uint16_t c[3]; std::memcpy(c, rec + rgbOffset, 6);
						color |= (uint32_t)(c[0] > 255 ? c[0] / 256 : c[0]) | ((uint32_t)(c[1] > 255 ? c[1] / 256 : c[1]) << 8) | ((uint32_t)(c[2] > 255 ? c[2] / 256 : c[2]) << 16);
					}
					p.color = color;
					dst[i] = p;
				}
			} else std::memcpy(dst, points.data() + first, (size_t)count * sizeof(Point));
			PointBatch batch;
			batch.first = (int)first; batch.count = (int)count; batch.pinnedMem = slot;
			lock_guard<mutex> lock(mtx_batchesInPinnedMemory);
			batchesInPinnedMemory.push_back(batch);
		}
	});
	std::atomic<uint64_t> numPointsProcessedSeen{0};             // (the reference's uploader reads stats.numPointsProcessed itself)
#else
	// pinned staging slots, as the reference's pinnedMemPool (main.cpp:141-222)
	void* pinned[4];
	CUevent uploadEnd[4];
	const size_t slotBytes = MAX_BATCH_SIZE * std::max<size_t>(sizeof(Point), lasBytesPerPoint);
	for (int i = 0; i < 4; i++) { cuMemAllocHost(&pinned[i], slotBytes); cuEventCreate(&uploadEnd[i], 0); cuEventRecord(uploadEnd[i], stream_upload); }
	CUdeviceptr lasStage[4] = {0, 0, 0, 0};                            // raw records of a batch on the device, one per pinned slot
	if (isLas) for (int i = 0; i < 4; i++) cuMemAlloc(&lasStage[i], slotBytes);

	// SIMLOD_HARNESS_PINNED=1: the point array itself is page-locked, as if the loader threads had already read every batch into its
	// pinned slot (main.cpp:846-935) — the uploader then issues the H2D copy straight from it and the staging memcpy disappears.
	const bool sourcePinned = std::getenv("SIMLOD_HARNESS_PINNED") != nullptr &&
	                          ((points.size() > 0 && hipHostRegister(points.data(), points.size() * sizeof(Point), hipHostRegisterDefault) == hipSuccess) ||
	                           (lasRecords.size() > 0 && hipHostRegister(lasRecords.data(), lasRecords.size(), hipHostRegisterDefault) == hipSuccess));

	// ---- spawnUploader (main.cpp:963-1063): ITS OWN THREAD, its own stream.  It publishes batchSizes[slot] and numBatchesUploaded
	// with stream-ordered memsets WHILE kernel_construct launches run on the frame thread's stream; the frame thread
	// publishes its Stats read-back through an atomic the back-pressure rule reads (main.cpp:1012).
	std::atomic<uint64_t> numPointsProcessedSeen{0};
	std::atomic<bool> quitUploader{false};
	std::atomic<uint64_t> batchStreamUploadIndex{0};
	const int device_ = device;
	std::thread uploader([&]() {
		(void)hipSetDevice(device_);
		uint64_t numPointsUploaded = 0;
		while (!quitUploader.load()) {
			const bool everythingIsDone = batchStreamUploadIndex.load() == numBatchesTotal;
			const bool processingLagsBehind = numPointsUploaded > numPointsProcessedSeen.load() + BATCH_STREAM_SIZE * MAX_BATCH_SIZE;
			if (everythingIsDone) break;
			if (processingLagsBehind) { std::this_thread::sleep_for(std::chrono::microseconds(50)); continue; }
			const uint64_t index = batchStreamUploadIndex.load();
			const int slot = (int)(index % 4);
			cuEventSynchronize(uploadEnd[slot]);
			const uint64_t first = index * MAX_BATCH_SIZE;
			const uint32_t count = (uint32_t)std::min<uint64_t>(MAX_BATCH_SIZE, numPointsTotal - first);
			const int uploadRingIndex = (int)(index % BATCH_STREAM_SIZE);
			if (isLas) {
				// the loader thread's job shrinks to moving bytes: raw records -> pinned slot -> device, decoded into the ring slot there
				const size_t bytes = (size_t)count * lasBytesPerPoint;
				const void* src = lasRecords.data() + first * lasBytesPerPoint;
				if (!sourcePinned) { std::memcpy(pinned[slot], src, bytes); src = pinned[slot]; }
				cuMemcpyHtoDAsync(lasStage[slot], src, bytes, stream_upload);
				simlod_decode_las((const void*)(uintptr_t)lasStage[slot], count, lasBytesPerPoint, lasFormat, lasScale, lasOffset,
				                  (SimlodPoint*)(uintptr_t)cptr_points_ring[uploadRingIndex], (void*)stream_upload);
			} else {
				const void* src = points.data() + first;
				if (!sourcePinned) { std::memcpy(pinned[slot], src, (size_t)count * sizeof(Point)); src = pinned[slot]; }
				cuMemcpyHtoDAsync(cptr_points_ring[uploadRingIndex], src, (size_t)count * sizeof(Point), stream_upload);
			}
			cuEventRecord(uploadEnd[slot], stream_upload);
			cuMemsetD32Async(cptr_batchSizes + 4 * uploadRingIndex, count, 1, stream_upload);
			cuMemsetD32Async(cptr_numBatchesUploaded, (unsigned)(index + 1), 1, stream_upload);
			batchStreamUploadIndex.store(index + 1);
			numPointsUploaded += count;
		}
	});

#endif
#ifndef SIMLOD_REF_UPLOADER_EXTRACT
	bool lastBatchFinishedDevice = false;
#endif
	const double loadStart = now();
	while (!lastBatchFinishedDevice) {
		// ---- frame (main.cpp:1159-1226): render first, then update, then the Stats copy
		renderCUDA(renderer);
		updateOctree(renderer);
		cuMemcpyDtoHAsync(h_stats_pinned, cptr_stats, sizeof(Stats), 0);
		cuCtxSynchronize();
		std::memcpy(&stats, h_stats_pinned, sizeof(Stats));
		numPointsProcessedSeen.store(stats.numPointsProcessed);
		lastBatchFinishedDevice = stats.numPointsProcessed == numPointsTotal || stats.memCapacityReached || (stats.dbg & 0x50u) != 0u;
		frameCounter++;
		if (frameCounter > 2000000) { std::fprintf(stderr, "no progress\n"); quitUploader.store(true); uploader.join(); return 1; }
	}
	quitUploader.store(true);
	uploader.join();
	const double totalUpdateDuration = 1000.0 * (now() - loadStart);
	renderCUDA(renderer);   // the frame that shows the finished octree
	cuMemcpyDtoH(&stats, cptr_stats, sizeof(Stats));

	std::printf("points %llu batches %llu frames %d\n", (unsigned long long)numPointsTotal, (unsigned long long)numBatchesTotal, (int)cntKernelRenderDuration);
	std::printf("numNodes %u numInner %u numLeaves %u numPoints %u numVoxels %u persistentBytes %llu chunkPoolSize %llu dbg %u\n", stats.numNodes, stats.numInner,
	            stats.numLeaves, stats.numPoints, stats.numVoxels, (unsigned long long)stats.allocatedBytes_persistent, (unsigned long long)stats.chunkPoolSize, stats.dbg);
	std::printf("visible nodes %u points %u voxels %u\n", stats.numVisibleNodes, stats.numVisiblePoints, stats.numVisibleVoxels);
	std::printf("%s", sourcePinned ? "source array page-locked: no staging memcpy\n" : "");
	std::printf("load+build wall %.1f ms (incl. H2D), update kernel %.2f ms total over %d launches = %.1f M points/s, render kernel %.3f ms/frame\n", totalUpdateDuration,
	            kernelUpdateDuration, (int)cntKernelUpdateDuration, numPointsTotal / (kernelUpdateDuration * 1e-3) / 1e6, kernelRenderDuration / std::max(1.0, cntKernelRenderDuration));
	if (outPath) {
		std::vector<uint32_t> img((size_t)width * height);
		cuMemcpyDtoH(img.data(), cptr_colorbuffer, img.size() * 4);
		FILE* f = std::fopen(outPath, "wb");
		std::fprintf(f, "P6\n%d %d\n255\n", width, height);
		for (int y = height - 1; y >= 0; y--) for (int x = 0; x < width; x++) { const uint32_t c = img[(size_t)y * width + x]; const unsigned char rgb[3] = {(unsigned char)c, (unsigned char)(c >> 8), (unsigned char)(c >> 16)}; std::fwrite(rgb, 1, 3, f); }
		std::fclose(f);
	}
	// SIMLOD_HARNESS_DUMP=<file>: the octree image as the kernels left it — {numNodes, persistent bytes in use, device address of the
	// node array, device address of the persistent buffer} as four uint64, the node records, the used part of the persistent buffer
	// — so that a test can compare the replay's octree with the oracle's node by node, not by seven counters
	if (const char* dumpPath = std::getenv("SIMLOD_HARNESS_DUMP")) {
		const uint64_t head[4] = {stats.numNodes, stats.allocatedBytes_persistent, (uint64_t)cptr_nodes, (uint64_t)cptr_buffer_persistent};
		std::vector<uint8_t> nodesHost((size_t)stats.numNodes * sizeof(SimlodNode)), persHost((size_t)stats.allocatedBytes_persistent);
		cuMemcpyDtoH(nodesHost.data(), cptr_nodes, nodesHost.size());
		cuMemcpyDtoH(persHost.data(), cptr_buffer_persistent, persHost.size());
		FILE* f = std::fopen(dumpPath, "wb");
		if (!f) { std::perror(dumpPath); return 1; }
		std::fwrite(head, 8, 4, f); std::fwrite(nodesHost.data(), 1, nodesHost.size(), f); std::fwrite(persHost.data(), 1, persHost.size(), f);
		std::fclose(f);
	}
	return (stats.dbg & ~0xeu) == 0 ? 0 : 3;   // 0x2 | 0x4 | 0x8: splits were deferred for lack of scratch space / node slots — nothing lost
}
