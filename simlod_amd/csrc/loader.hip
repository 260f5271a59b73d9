// loader.hip — LAS point records -> SimLOD `Point` records, on the device.
//
// Replaces the CPU parse loop of the reference's loader thread (modules/progressive_octree/LasLoader.cpp:169-227): the
// host copies the raw record bytes of a batch to the device and this kernel writes the 16-byte XYZRGBA points straight
// into a slot of the batch ring, so the loader threads only move bytes.  Arithmetic, operation by operation
// (LasLoader.cpp:212-221):   x = float( double(int32 X) * scale.x + (header.offset.x + translation.x) )   — one fp64
// multiply, one fp64 add (never contracted: -ffp-contract=off), one round-to-nearest conversion to fp32; colour
// channel = c > 255 ? c / 256 : c for the formats that carry RGB (2, 3, 5, 7 — the reference's list, :177-185).
// Where the reference leaves bytes undefined (alpha always, r/g/b for formats without colour: its local `point` is
// never initialised) this kernel writes alpha = 255 — what tools/las2simlod.mjs:144 writes — and r = g = b = 0.
//
// HBM-bound byte work: (bytesPerPoint + 16) B per point.  Records are 20-67 bytes and unaligned, so a workgroup stages the
// 256 records of its tile in LDS with coalesced 16-byte loads (256 * bytesPerPoint is a multiple of 16) and every lane
// then picks its fields out of LDS; the 16-byte stores are coalesced by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simlod_abi.h"
#include "simlod_hip.h"
#include "simlod_internal.hpp"

namespace simlod {

static constexpr uint32_t LTPB = 256;

struct LasArgs {
	const uint8_t* records;
	SimlodPoint*   out;
	uint64_t       numPoints;
	uint32_t       bytesPerPoint, rgbOffset;
	double         scale[3], offset[3];
};

__device__ __forceinline__ uint32_t lds_u16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ int32_t lds_i32(const uint8_t* p) {
	return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

__global__ __launch_bounds__(LTPB) void k_decode_las(LasArgs a) {
	extern __shared__ uint4 stage[];
	uint8_t* bytes = reinterpret_cast<uint8_t*>(stage);
	const uint32_t bpp = a.bytesPerPoint;
	const uint64_t totalBytes = a.numPoints * bpp;
	const uint64_t numTiles = (a.numPoints + LTPB - 1) / LTPB;
	const uint32_t tileVecs = LTPB * bpp / 16;
	for (uint64_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
		const uint64_t tileByte = tile * LTPB * bpp;
		__syncthreads();
		for (uint32_t v = threadIdx.x; v < tileVecs; v += LTPB) {
			const uint64_t b = tileByte + (uint64_t)v * 16;
			if (b + 16 <= totalBytes) stage[v] = *reinterpret_cast<const uint4*>(a.records + b);
			else for (uint32_t k = 0; k < 16 && b + k < totalBytes; k++) bytes[v * 16 + k] = a.records[b + k];   // ragged end of the buffer
		}
		__syncthreads();
		const uint64_t i = tile * LTPB + threadIdx.x;
		if (i >= a.numPoints) continue;
		const uint8_t* rec = bytes + threadIdx.x * bpp;
		const int32_t X = lds_i32(rec), Y = lds_i32(rec + 4), Z = lds_i32(rec + 8);
		float4 o;
		o.x = (float)((double)X * a.scale[0] + a.offset[0]);      // LasLoader.cpp:212-214
		o.y = (float)((double)Y * a.scale[1] + a.offset[1]);
		o.z = (float)((double)Z * a.scale[2] + a.offset[2]);
		uint32_t color = 0xff000000u;
		if (a.rgbOffset > 0) {                                    // LasLoader.cpp:216-221
			const uint32_t r = lds_u16(rec + a.rgbOffset), g = lds_u16(rec + a.rgbOffset + 2), b = lds_u16(rec + a.rgbOffset + 4);
			color |= (r > 255u ? r / 256u : r) | ((g > 255u ? g / 256u : g) << 8) | ((b > 255u ? b / 256u : b) << 16);
		}
		o.w = __uint_as_float(color);
		reinterpret_cast<float4*>(a.out)[i] = o;
	}
}

int launch_decode_las(const void* records, uint64_t numPoints, uint32_t bytesPerPoint, uint32_t format, const double* scale,
                      const double* offset, SimlodPoint* out, hipStream_t stream) {
	if (numPoints == 0) return 0;
	if (records == nullptr || out == nullptr || scale == nullptr || offset == nullptr) return (int)hipErrorInvalidValue;
	uint32_t rgb = 0;                                             // LasLoader.cpp:177-185
	if (format == 2) rgb = 20; else if (format == 3) rgb = 28;
	if (format == 5) rgb = 28;
	if (format == 7) rgb = 30;
	// a record must hold XYZ and, where the format has it, the RGB triple; 255 keeps the stage within 64 KiB of LDS
	if (bytesPerPoint < 12 || bytesPerPoint > 255 || (rgb > 0 && rgb + 6 > bytesPerPoint)) return (int)hipErrorInvalidValue;
	if ((reinterpret_cast<uintptr_t>(records) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 15u) != 0) return (int)hipErrorInvalidValue;
	LasArgs a{};
	a.records = static_cast<const uint8_t*>(records); a.out = out; a.numPoints = numPoints;
	a.bytesPerPoint = bytesPerPoint; a.rgbOffset = rgb;
	for (int k = 0; k < 3; k++) { a.scale[k] = scale[k]; a.offset[k] = offset[k]; }
	const uint64_t numTiles = (numPoints + LTPB - 1) / LTPB;
	const uint32_t grid = (uint32_t)(numTiles < (uint64_t)device_info().numCUs * 16 ? numTiles : (uint64_t)device_info().numCUs * 16);
	if (profile_enabled()) profile_mark("k_decode_las", stream);
	hipLaunchKernelGGL(k_decode_las, dim3(grid), dim3(LTPB), (size_t)LTPB * bytesPerPoint, stream, a);
	if (profile_enabled()) profile_close(stream);
	return (int)hipGetLastError();
}

}  // namespace simlod
