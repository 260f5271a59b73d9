// oracle/las_ref_glue.cpp — TEST INFRASTRUCTURE ONLY.
// C entry points over the reference's own LAS reader (modules/progressive_octree/LasLoader.h:21-55 loadHeader,
// LasLoader.cpp:169-227 loadLasNative), which oracle/Makefile compiles in place next to this file.
#include "LasLoader.h"

extern "C" {

struct RefLasHeader {
	int32_t  versionMajor, versionMinor;
	uint64_t headerSize, offsetToPointData, format, bytesPerPoint, numPoints;
	double   scale[3], offset[3], min[3], max[3];
};

void ref_las_header(const char* path, RefLasHeader* out) {
	LasHeader h = loadHeader(path);
	out->versionMajor = h.versionMajor; out->versionMinor = h.versionMinor;
	out->headerSize = h.headerSize; out->offsetToPointData = h.offsetToPointData;
	out->format = h.format; out->bytesPerPoint = h.bytesPerPoint; out->numPoints = h.numPoints;
	for (int i = 0; i < 3; i++) { out->scale[i] = h.scale[i]; out->offset[i] = h.offset[i]; out->min[i] = h.min[i]; out->max[i] = h.max[i]; }
}

// target: numPoints 16-byte records.  The reference never writes the alpha byte nor, for formats without colour, r/g/b
// (its local `point` is uninitialised): callers compare those bytes only where the reference defines them.
void ref_las_load(const char* path, uint64_t firstPoint, uint64_t numPoints, void* target, const double* translation) {
	LasHeader h = loadHeader(path);
	double t[3] = {translation[0], translation[1], translation[2]};
	loadLasNative(path, h, firstPoint, numPoints, target, t);
}

}
