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
// HBM-bound byte work: (bytesPerPoint + 16) B per point.  Records are 20-67 bytes and unaligned, so a workgroup stages the records of
// its tile in LDS with coalesced 16-byte loads (tile * bytesPerPoint is a multiple of 16) and every lane then picks its fields out of LDS —
// as DWORDS: X, Y, Z are three unaligned dwords that share one byte shift, so four aligned LDS dwords and three v_alignbyte_b32 give
// them, three more dwords and two alignbytes the RGB triple (round 4 assembled every field from single bytes: 18 ds_read_u8 per record; the
// kernel reached 0.64 of the device's copy rate).  A lane decodes TWO records of a 512-record tile (records up to 64 bytes: a batch is half
// as many workgroups, all resident at once); the 16-byte stores are coalesced by construction.
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

// the unaligned little-endian dword at byte `off` of the staged tile: two aligned dwords, one funnel shift
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t shiftBytes) { return __builtin_amdgcn_alignbyte(hi, lo, shiftBytes); }

template <uint32_t RPT>                                   // records per thread
__global__ __launch_bounds__(LTPB) void k_decode_las(LasArgs a) {
	extern __shared__ uint4 stage[];
	const uint32_t* words = reinterpret_cast<const uint32_t*>(stage);
	uint8_t* bytes = reinterpret_cast<uint8_t*>(stage);
	constexpr uint32_t TILE = LTPB * RPT;
	const uint32_t bpp = a.bytesPerPoint;
	const uint64_t totalBytes = a.numPoints * bpp;
	const uint64_t numTiles = (a.numPoints + TILE - 1) / TILE;
	const uint32_t tileVecs = TILE * bpp / 16;
	for (uint64_t tile = blockIdx.x; tile < numTiles; tile += gridDim.x) {
		const uint64_t tileByte = tile * TILE * bpp;
		__syncthreads();
		for (uint32_t v = threadIdx.x; v < tileVecs; v += LTPB) {
			const uint64_t b = tileByte + (uint64_t)v * 16;
			if (b + 16 <= totalBytes) stage[v] = *reinterpret_cast<const uint4*>(a.records + b);
			else for (uint32_t k = 0; k < 16 && b + k < totalBytes; k++) bytes[v * 16 + k] = a.records[b + k];   // ragged end of the buffer
		}
		__syncthreads();
#pragma unroll
		for (uint32_t r = 0; r < RPT; r++) {
			const uint32_t local = r * LTPB + threadIdx.x;
			const uint64_t i = tile * TILE + local;
			if (i >= a.numPoints) continue;
			const uint32_t off = local * bpp, w = off >> 2, sh = off & 3u;
			const uint32_t w0 = words[w], w1 = words[w + 1], w2 = words[w + 2], w3 = words[w + 3];          // (w + 3: at most the tile's padding dword)
			const int32_t X = (int32_t)funnel(w1, w0, sh), Y = (int32_t)funnel(w2, w1, sh), Z = (int32_t)funnel(w3, w2, sh);
			float4 o;
			o.x = (float)((double)X * a.scale[0] + a.offset[0]);      // LasLoader.cpp:212-214
			o.y = (float)((double)Y * a.scale[1] + a.offset[1]);
			o.z = (float)((double)Z * a.scale[2] + a.offset[2]);
			uint32_t color = 0xff000000u;
			if (a.rgbOffset > 0) {                                    // LasLoader.cpp:216-221
				const uint32_t co = off + a.rgbOffset, c = co >> 2, cs = co & 3u;
				const uint32_t c0 = words[c], c1 = words[c + 1], c2 = words[c + 2];
				const uint32_t rg = funnel(c1, c0, cs), bx = funnel(c2, c1, cs);
				const uint32_t rr = rg & 0xffffu, gg = rg >> 16, bb = bx & 0xffffu;
				color |= (rr > 255u ? rr / 256u : rr) | ((gg > 255u ? gg / 256u : gg) << 8) | ((bb > 255u ? bb / 256u : bb) << 16);
			}
			o.w = __uint_as_float(color);
			reinterpret_cast<float4*>(a.out)[i] = o;
		}
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
#ifndef LAS_RPT
#define LAS_RPT 2
#endif
	const uint32_t rpt = bytesPerPoint <= 64 ? (uint32_t)LAS_RPT : 1u;        // (several records per lane while the tile's stage stays within 32 KiB of LDS)
	const uint64_t numTiles = (numPoints + LTPB * rpt - 1) / (LTPB * rpt);
	const uint32_t grid = (uint32_t)(numTiles < (uint64_t)device_info().numCUs * 16 ? numTiles : (uint64_t)device_info().numCUs * 16);
	const size_t lds = (size_t)LTPB * rpt * bytesPerPoint + 16;  // + the dword a record's last funnel shift may touch behind the tile
	if (profile_enabled()) profile_mark("k_decode_las", stream);
	if (rpt != 1) hipLaunchKernelGGL(k_decode_las<LAS_RPT>, dim3(grid), dim3(LTPB), lds, stream, a);
	else hipLaunchKernelGGL(k_decode_las<1>, dim3(grid), dim3(LTPB), lds, stream, a);
	if (profile_enabled()) profile_close(stream);
	return (int)hipGetLastError();
}

}  // namespace simlod
