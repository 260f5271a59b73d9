// oracle/shim_las/unsuck.hpp — TEST INFRASTRUCTURE ONLY.
// Stand-in for the reference's include/unsuck.hpp (which needs <format> and fmt, absent from this toolchain) so that
// modules/progressive_octree/LasLoader.{h,cpp} compile unmodified, in place, as host C++.  It provides exactly what those two
// files use: `string`, `fs`, Buffer::get<T>(offset), the two readBinaryFile overloads (semantics of unsuck.hpp:443-498:
// clamp to the file size) and now().  Written from scratch; nothing is copied from the reference.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <memory>
#include <string>

using std::string;
using std::shared_ptr;
namespace fs = std::filesystem;

struct Buffer {
	uint8_t* data = nullptr;
	uint64_t size = 0;
	explicit Buffer(uint64_t n) : data((uint8_t*)calloc(n ? n : 1, 1)), size(n) {}
	~Buffer() { free(data); }
	template <class T> T get(uint64_t offset) const { T v; memcpy(&v, data + offset, sizeof(T)); return v; }
};

inline void readBinaryFile(string path, uint64_t start, uint64_t size, void* target) {
	FILE* f = fopen(path.c_str(), "rb");
	if (!f) return;
	const uint64_t total = fs::file_size(path);
	if (start < total) {
		if (start + size > total) size = total - start;
		fseeko(f, (off_t)start, SEEK_SET);
		size_t got = fread(target, 1, size, f);
		(void)got;
	}
	fclose(f);
}

inline shared_ptr<Buffer> readBinaryFile(string path, uint64_t start, uint64_t size) {
	const uint64_t total = fs::file_size(path);
	if (start >= total) return std::make_shared<Buffer>(0);
	if (start + size > total) size = total - start;
	auto b = std::make_shared<Buffer>(size);
	readBinaryFile(path, start, size, b->data);
	return b;
}

inline double now() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
