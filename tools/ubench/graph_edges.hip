// What does a dependency between the builder's two streams cost as a hipGraph edge, against the event hops the library uses?
// The ingest's chain per group — caller's stream: count, queue, [wait inserted(g-1)] hist, expand(signals expanded(g)); second stream: [wait expanded(g)] insert
// (signals inserted(g)), voxelize — with kernels that spin for a fixed time, (1) enqueued as the library does (hipExtLaunchKernelGGL stop events + hipStreamWaitEvent),
// (2) captured once into a graph (fork / join through the same events) and replayed.  Prints the time per group beyond the kernels' own spin times.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/graph_edges.hip -o tools/ubench/graph_edges && tools/ubench/graph_edges
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks, unsigned* sink) {      // ticks of the 100 MHz wall clock
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) { }
	if (sink != nullptr && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1u);
}
int main() {
	const int G = 20, REPS = 50;
	const unsigned long long us = 100;      // ticks per microsecond
	// spin times per kernel (us): roughly the one-batch-per-group chain of round 5
	const unsigned long long tCount = 20 * us, tQueue = 10 * us, tHist = 18 * us, tExpand = 21 * us, tInsert = 21 * us, tVox = 38 * us;
	hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
	std::vector<hipEvent_t> expanded(G), inserted(G);
	const unsigned flags = hipEventDisableTiming | hipEventDisableSystemFence;
	for (int i = 0; i < G; i++) { CK(hipEventCreateWithFlags(&expanded[i], flags)); CK(hipEventCreateWithFlags(&inserted[i], flags)); }
	hipEvent_t tail, t0, t1; CK(hipEventCreateWithFlags(&tail, flags)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
	unsigned* sink; CK(hipMalloc(&sink, 4)); CK(hipMemset(sink, 0, 4));
	auto enqueue = [&](bool ext) -> int {      // ext: stop events on the launches (as the library); else hipEventRecord (what stream capture understands)
		for (int g = 0; g < G; g++) {
			hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, tCount, (unsigned*)nullptr);
			hipLaunchKernelGGL(spin, dim3(16), dim3(256), 0, A, tQueue, (unsigned*)nullptr);
			if (g > 0) CK(hipStreamWaitEvent(A, inserted[g - 1], 0));
			hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, A, tHist, (unsigned*)nullptr);
			if (ext) hipExtLaunchKernelGGL(spin, dim3(64), dim3(1024), 0, A, nullptr, expanded[g], 0, tExpand, (unsigned*)nullptr);
			else { hipLaunchKernelGGL(spin, dim3(64), dim3(1024), 0, A, tExpand, (unsigned*)nullptr); CK(hipEventRecord(expanded[g], A)); }
			CK(hipStreamWaitEvent(B, expanded[g], 0));
			if (ext) hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, B, nullptr, inserted[g], 0, tInsert, (unsigned*)nullptr);
			else { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, B, tInsert, (unsigned*)nullptr); CK(hipEventRecord(inserted[g], B)); }
			hipLaunchKernelGGL(spin, dim3(256), dim3(1024), 0, B, tVox, sink);
		}
		CK(hipEventRecord(tail, B)); CK(hipStreamWaitEvent(A, tail, 0));
		hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, A, 4 * us, (unsigned*)nullptr);
		return 0;
	};
	// ideal: the front loop's kernels back to back, the back half hidden: per group max(front, cross loop)
	const double frontUs = (double)(tCount + tQueue + tHist + tExpand) / us, crossUs = (double)(tHist + tExpand + tInsert) / us;
	printf("kernels spin %.0f us per group on the caller's stream (count + queue + hist + expand); the loop through both streams (hist + expand + insert): %.0f us\n", frontUs, crossUs);
	for (int ext = 1; ext >= 0; ext--) {
		for (int w = 0; w < 3; w++) { if (enqueue(ext != 0)) return 1; } CK(hipDeviceSynchronize());
		CK(hipEventRecord(t0, A));
		for (int r = 0; r < REPS; r++) if (enqueue(ext != 0)) return 1;
		CK(hipEventRecord(t1, A)); CK(hipDeviceSynchronize());
		float ms; CK(hipEventElapsedTime(&ms, t0, t1));
		printf("streams + events (%s): %.1f us per group  -> %.1f us beyond the front loop's kernels\n", ext ? "stop events on the launches, as the library" : "hipEventRecord", ms * 1e3 / (REPS * G), ms * 1e3 / (REPS * G) - frontUs);
	}
	// the same chain captured into a graph
	hipGraph_t graph; hipGraphExec_t exec;
	CK(hipStreamBeginCapture(A, hipStreamCaptureModeRelaxed));
	// B joins the capture through an event recorded on A
	hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, flags)); CK(hipEventRecord(fork, A)); CK(hipStreamWaitEvent(B, fork, 0));
	if (enqueue(false)) return 1;
	CK(hipStreamEndCapture(A, &graph));
	CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
	size_t nodes = 0; CK(hipGraphGetNodes(graph, nullptr, &nodes));
	for (int w = 0; w < 3; w++) CK(hipGraphLaunch(exec, A)); CK(hipDeviceSynchronize());
	CK(hipEventRecord(t0, A));
	for (int r = 0; r < REPS; r++) CK(hipGraphLaunch(exec, A));
	CK(hipEventRecord(t1, A)); CK(hipDeviceSynchronize());
	float ms; CK(hipEventElapsedTime(&ms, t0, t1));
	printf("hipGraph of the same chain (%zu nodes), replayed: %.1f us per group  -> %.1f us beyond the front loop's kernels\n", nodes, ms * 1e3 / (REPS * G), ms * 1e3 / (REPS * G) - frontUs);
	return 0;
}
