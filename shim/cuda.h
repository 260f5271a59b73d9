// shim/cuda.h — the slice of the CUDA driver API that the reference host uses around its hot path
// (modules/progressive_octree/main_progressive_octree.cpp, include/CudaModularProgram.h), expressed on HIP and on the C ABI of
// libsimlod_hip.so.  With this header and shim/CudaModularProgram.h on the include path, host code written against the reference's
// calls (harness/simlod_headless.cpp replays them) builds for an MI355X without source changes to those calls.
//
//   launch surface     cuLaunchCooperativeKernel, cuOccupancyMaxActiveBlocksPerMultiprocessor  -> simlod_launch_cooperative, ...
//   memory / streams   cuMemAlloc (zero-filled: the reference relies on fresh VRAM reading 0), cuMemAllocHost,
//                      cuMemcpyHtoDAsync, cuMemcpyDtoHAsync, cuMemsetD32(Async), cuMemGetInfo, cuStreamCreate, cuEvent*
//   not provided       NVRTC / nvJitLink (kernels are precompiled for gfx950), GL interop (the colour buffer is a linear image)
#pragma once

#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>

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
inline CUresult cuMemsetD32(CUdeviceptr dst, unsigned v, size_t count) { return (CUresult)hipMemsetD32((hipDeviceptr_t)(uintptr_t)dst, (int)v, count); }
inline CUresult cuMemsetD32Async(CUdeviceptr dst, unsigned v, size_t count, CUstream s) { return (CUresult)hipMemsetD32Async((hipDeviceptr_t)(uintptr_t)dst, (int)v, count, s); }
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
