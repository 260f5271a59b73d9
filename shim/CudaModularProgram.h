// shim/CudaModularProgram.h — stand-in for the reference's include/CudaModularProgram.h:140-264 on top of libsimlod_hip.so.
// Same constructor shape ({.modules, .kernels}), same public members (`kernels[name]`, `onCompile`).  Nothing is compiled at run
// time: the kernels are precompiled for gfx950, so onCompile callbacks never fire and there is no hot reload (CMP.h:181-184).
#pragma once

#include <cstdlib>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "cuda.h"

struct CudaModularProgramArgs {
	std::vector<std::string> modules;
	std::vector<std::string> kernels;
};

struct CudaModularProgram {
	SimlodProgram* program = nullptr;
	std::unordered_map<std::string, CUfunction> kernels;
	std::vector<std::function<void(void)>> compileCallbacks;

	CudaModularProgram(CudaModularProgramArgs args) {
		std::vector<const char*> m, k;
		for (auto& s : args.modules) m.push_back(s.c_str());
		for (auto& s : args.kernels) k.push_back(s.c_str());
		if (simlod_program_create(&program, m.data(), (int)m.size(), k.data(), (int)k.size()) != 0) {
			std::fprintf(stderr, "CudaModularProgram: unknown module/kernel combination\n");
			std::exit(1);                                   // the reference exits on a link failure as well (CMP.h:15-33)
		}
		for (auto& s : args.kernels) kernels[s] = simlod_program_kernel(program, s.c_str());
	}
	~CudaModularProgram() { simlod_program_destroy(program); }
	void onCompile(std::function<void(void)> callback) { compileCallbacks.push_back(callback); }
};
