// shim/cuda.h — the slice of the CUDA driver API that the reference host uses around its hot path
// (modules/progressive_octree/main_progressive_octree.cpp, include/CudaModularProgram.h), expressed on HIP and on the C ABI of
// libsimlod_hip.so.  With this header and shim/CudaModularProgram.h on the include path, host code written against the reference's
// calls (harness/simlod_headless.cpp replays them) builds for an MI355X without source changes to those calls.
//
//   launch surface     cuLaunchCooperativeKernel, cuOccupancyMaxActiveBlocksPerMultiprocessor  -> simlod_launch_cooperative, ...
//   memory / streams   cuMemAlloc (zero-filled: the reference relies on fresh VRAM reading 0), cuMemAllocHost,
//                      cuMemcpyHtoDAsync, cuMemcpyDtoHAsync, cuMemsetD32(Async), cuMemGetInfo, cuStreamCreate, cuEvent*
//   GL interop         cuGraphicsGLRegisterImage / MapResources / SubResourceGetMappedArray / UnmapResources / UnregisterResource,
//                      cuSurfObjectCreate / Destroy (main.cpp:472-486, 542-545, 641): headless — a GL texture handle stands for a
//                      linear RGBA8 device image (simlod_shim_register_surface, or allocated on first use); the "surface object"
//                      handed to kernel_render is that image's device address
//   module loading     cuModuleLoadData / cuModuleGetFunction (include/CudaModularProgram.h:245-249): the "image" is ignored, a
//                      function is looked up by name among the three precompiled kernels
//   not provided       NVRTC / nvJitLink (kernels are precompiled for gfx950)
#pragma once

#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "simlod_hip.h"

using CUresult = int;
using CUdeviceptr = unsigned long long;
using CUstream = hipStream_t;
using cudaStream_t = hipStream_t;
using CUevent = hipEvent_t;
using CUdevice = int;
using CUcontext = void*;
using CUfunction = SimlodFunction*;

constexpr CUresult CUDA_SUCCESS = 0;
constexpr unsigned CU_STREAM_NON_BLOCKING = hipStreamNonBlocking;
enum CUdevice_attribute { CU_DEVICE_ATTRIBUTE_MULTIPROCESSOR_COUNT = 16, CU_DEVICE_ATTRIBUTE_COMPUTE_CAPABILITY_MAJOR = 75, CU_DEVICE_ATTRIBUTE_COMPUTE_CAPABILITY_MINOR = 76 };

inline CUresult cuInit(unsigned flags) { return (CUresult)hipInit(flags); }
inline CUresult cuDeviceGet(CUdevice* dev, int ordinal) { *dev = ordinal; return (CUresult)hipSetDevice(ordinal); }
inline CUresult cuCtxCreate(CUcontext* ctx, unsigned, CUdevice dev) { *ctx = nullptr; return (CUresult)hipSetDevice(dev); }
inline CUresult cuCtxGetDevice(CUdevice* dev) { return (CUresult)hipGetDevice(dev); }
inline CUresult cuCtxSetCurrent(CUcontext) { return CUDA_SUCCESS; }
inline CUresult cuCtxSynchronize() { return (CUresult)hipDeviceSynchronize(); }
inline CUresult cuDeviceGetAttribute(int* value, CUdevice_attribute attr, CUdevice dev) {
	hipDeviceProp_t p;
	hipError_t e = hipGetDeviceProperties(&p, dev);
	if (e != hipSuccess) return (CUresult)e;
	*value = attr == CU_DEVICE_ATTRIBUTE_MULTIPROCESSOR_COUNT ? p.multiProcessorCount : attr == CU_DEVICE_ATTRIBUTE_COMPUTE_CAPABILITY_MAJOR ? p.major : p.minor;
	return CUDA_SUCCESS;
}
inline CUresult cuStreamCreate(CUstream* s, unsigned flags) { return (CUresult)hipStreamCreateWithFlags(s, flags); }
inline CUresult cuMemAlloc(CUdeviceptr* p, size_t bytes) {
	void* q = nullptr;
	hipError_t e = hipMalloc(&q, bytes);
	if (e == hipSuccess) e = hipMemset(q, 0, bytes);      // H11: the host renders before its first reset
	*p = (CUdeviceptr)(uintptr_t)q;
	return (CUresult)e;
}
inline CUresult cuMemAllocHost(void** p, size_t bytes) { return (CUresult)hipHostMalloc(p, bytes, hipHostMallocDefault); }
inline CUresult cuMemGetInfo(size_t* freeBytes, size_t* total) { return (CUresult)hipMemGetInfo(freeBytes, total); }
inline CUresult cuMemcpyHtoDAsync(CUdeviceptr dst, const void* src, size_t n, CUstream s) { return (CUresult)hipMemcpyAsync((void*)(uintptr_t)dst, src, n, hipMemcpyHostToDevice, s); }
inline CUresult cuMemcpyDtoHAsync(void* dst, CUdeviceptr src, size_t n, CUstream s) { return (CUresult)hipMemcpyAsync(dst, (const void*)(uintptr_t)src, n, hipMemcpyDeviceToHost, s); }
inline CUresult cuMemcpyDtoH(void* dst, CUdeviceptr src, size_t n) { return (CUresult)hipMemcpy(dst, (const void*)(uintptr_t)src, n, hipMemcpyDeviceToHost); }
// (a one-word memset may be the uploader publishing its batch count, main_progressive_octree.cpp:1047-1050: the library sizes its launches by it — it ignores every other address)
inline CUresult cuMemsetD32(CUdeviceptr dst, unsigned v, size_t count) { if (count == 1) (void)simlod_upload_counter_written((const void*)(uintptr_t)dst, v); return (CUresult)hipMemsetD32((hipDeviceptr_t)(uintptr_t)dst, (int)v, count); }
inline CUresult cuMemsetD32Async(CUdeviceptr dst, unsigned v, size_t count, CUstream s) { if (count == 1) (void)simlod_upload_counter_written((const void*)(uintptr_t)dst, v); return (CUresult)hipMemsetD32Async((hipDeviceptr_t)(uintptr_t)dst, (int)v, count, s); }
inline CUresult cuMemsetD8(CUdeviceptr dst, unsigned char v, size_t count) { return (CUresult)hipMemset((void*)(uintptr_t)dst, v, count); }
inline CUresult cuEventCreate(CUevent* e, unsigned) { return (CUresult)hipEventCreate(e); }
inline CUresult cuEventRecord(CUevent e, CUstream s) { return (CUresult)hipEventRecord(e, s); }
inline CUresult cuEventQuery(CUevent e) { return (CUresult)hipEventQuery(e); }
inline CUresult cuEventSynchronize(CUevent e) { return (CUresult)hipEventSynchronize(e); }
inline CUresult cuEventElapsedTime(float* ms, CUevent a, CUevent b) { return (CUresult)hipEventElapsedTime(ms, a, b); }
inline CUresult cuStreamSynchronize(CUstream s) { return (CUresult)hipStreamSynchronize(s); }
inline CUresult cuGetErrorString(CUresult r, const char** str) { *str = hipGetErrorString((hipError_t)r); return CUDA_SUCCESS; }
inline CUresult cuOccupancyMaxActiveBlocksPerMultiprocessor(int* n, CUfunction f, int blockSize, size_t) { return simlod_function_max_active_blocks(f, blockSize, n); }
inline CUresult cuLaunchCooperativeKernel(CUfunction f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                          unsigned sharedMemBytes, CUstream stream, void** kernelParams) {
	return simlod_launch_cooperative(f, gx, gy, gz, bx, by, bz, sharedMemBytes, (void*)stream, kernelParams);
}
inline CUresult cuMemFree(CUdeviceptr p) { return (CUresult)hipFree((void*)(uintptr_t)p); }
inline CUresult cuMemFreeHost(void* p) { return (CUresult)hipHostFree(p); }
inline CUresult cuStreamDestroy(CUstream s) { return (CUresult)hipStreamDestroy(s); }
inline CUresult cuEventDestroy(CUevent e) { return (CUresult)hipEventDestroy(e); }
constexpr CUresult cudaSuccess = 0;                      // the reference compares a driver result with the runtime constant (main.cpp:1004)

// ---- GL interop, headless ------------------------------------------------------------------------------------------------------------
using GLuint = unsigned;
using GLenum = unsigned;
#ifndef GL_TEXTURE_2D
constexpr GLenum GL_TEXTURE_2D = 0x0DE1;
#endif
constexpr unsigned CU_GRAPHICS_REGISTER_FLAGS_WRITE_DISCARD = 2;
constexpr unsigned CU_GRAPHICS_REGISTER_FLAGS_SURFACE_LDST = 4;
#define CU_STREAM_DEFAULT 0
struct SimlodShimSurface { CUdeviceptr image; size_t bytes; int mapped; };
using CUgraphicsResource = SimlodShimSurface*;
using CUarray = SimlodShimSurface*;
using CUsurfObject = unsigned long long;
enum CUresourcetype { CU_RESOURCE_TYPE_ARRAY = 0, CU_RESOURCE_TYPE_MIPMAPPED_ARRAY = 1, CU_RESOURCE_TYPE_LINEAR = 2, CU_RESOURCE_TYPE_PITCH2D = 3 };
struct CUDA_RESOURCE_DESC {
	CUresourcetype resType;
	union { struct { CUarray hArray; } array; struct { int reserved[32]; } reserved; } res;
	unsigned flags;
};
inline std::unordered_map<GLuint, SimlodShimSurface>& simlod_shim_surfaces() { static std::unordered_map<GLuint, SimlodShimSurface> m; return m; }
// what the GL texture `handle` stands for: width * height RGBA8 pixels in device memory (row 0 first)
inline void simlod_shim_register_surface(GLuint handle, CUdeviceptr image, size_t bytes) { simlod_shim_surfaces()[handle] = SimlodShimSurface{image, bytes, 0}; }
inline CUresult cuGraphicsGLRegisterImage(CUgraphicsResource* resource, GLuint image, GLenum, unsigned) {
	auto& m = simlod_shim_surfaces();
	auto it = m.find(image);
	if (it == m.end()) {                                 // nobody said what the texture is: a 4K image of its own
		CUdeviceptr p = 0;
		const size_t bytes = (size_t)3840 * 2160 * 4;
		CUresult r = cuMemAlloc(&p, bytes);
		if (r != CUDA_SUCCESS) return r;
		it = m.emplace(image, SimlodShimSurface{p, bytes, 0}).first;
	}
	*resource = &it->second;
	return CUDA_SUCCESS;
}
inline CUresult cuGraphicsMapResources(unsigned count, CUgraphicsResource* resources, CUstream) { for (unsigned i = 0; i < count; i++) resources[i]->mapped++; return CUDA_SUCCESS; }
inline CUresult cuGraphicsUnmapResources(unsigned count, CUgraphicsResource* resources, CUstream) { for (unsigned i = 0; i < count; i++) resources[i]->mapped--; return CUDA_SUCCESS; }
inline CUresult cuGraphicsSubResourceGetMappedArray(CUarray* array, CUgraphicsResource resource, unsigned, unsigned) { *array = resource; return resource->mapped > 0 ? CUDA_SUCCESS : (CUresult)hipErrorNotMapped; }
inline CUresult cuGraphicsUnregisterResource(CUgraphicsResource) { return CUDA_SUCCESS; }
inline CUresult cuSurfObjectCreate(CUsurfObject* surf, const CUDA_RESOURCE_DESC* desc) {
	if (desc->resType != CU_RESOURCE_TYPE_ARRAY || desc->res.array.hArray == nullptr) return (CUresult)hipErrorInvalidValue;
	*surf = desc->res.array.hArray->image;               // kernel_render's `gl_colorbuffer` argument: the linear image
	return CUDA_SUCCESS;
}
inline CUresult cuSurfObjectDestroy(CUsurfObject) { return CUDA_SUCCESS; }

// ---- module loading (CudaModularProgram::link, include/CudaModularProgram.h:227-256) ----------------------------------------------------
struct SimlodShimModule { SimlodProgram* program; };
using CUmodule = SimlodShimModule*;
// `image` would be the cubin nvJitLink produced; here every module holds the three precompiled kernels
inline CUresult cuModuleLoadData(CUmodule* module, const void*) {
	static const char* mods[] = {"reset.cu", "progressive_octree_voxels.cu", "render.cu", "utils.cu"};
	*module = new SimlodShimModule{nullptr};
	(void)mods;
	return CUDA_SUCCESS;
}
inline CUresult cuModuleGetFunction(CUfunction* fn, CUmodule module, const char* name) {
	// "kernel" -> reset.cu, "kernel_construct" -> progressive_octree_voxels.cu, "kernel_render" -> render.cu
	const char* mod = std::strcmp(name, "kernel") == 0 ? "reset.cu" : std::strcmp(name, "kernel_construct") == 0 ? "progressive_octree_voxels.cu" : "render.cu";
	const char* mods[] = {mod, "utils.cu"};
	const char* kernels[] = {name};
	CUresult r = simlod_program_create(&module->program, mods, 2, kernels, 1);
	if (r != CUDA_SUCCESS) return r;
	*fn = simlod_program_kernel(module->program, name);
	return *fn != nullptr ? CUDA_SUCCESS : (CUresult)hipErrorNotFound;
}
inline CUresult cuModuleUnload(CUmodule) { return CUDA_SUCCESS; }   // functions handed out stay valid
