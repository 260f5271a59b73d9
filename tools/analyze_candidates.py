"""CPU-only analysis: how are the NEW voxel cells of one batch distributed over 128-byte lines of the occupancy grids?"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from simlod_amd import abi, camera, synthetic
n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pts, box = synthetic.terrain(36_000_000, seed=7)
W, H = 256, 256
u = abi.make_uniforms(W, H, np.eye(4, dtype=np.float32), box, persistent_capacity=4 << 30, momentary_capacity=300_000_000)
o = oracle.HostOctree("port", persistent_bytes=2 << 30, ring_slots=50)
o.reset(u)
B = 1_000_000
for b in range(n_batches):
    o.upload(pts[b * B:(b + 1) * B]); o.construct(u)
batch = np.ascontiguousarray(pts[n_batches * B:(n_batches + 1) * B])
node = np.zeros(B, np.uint32); addr = np.zeros(B, np.uint64); isset = np.zeros(B, np.uint8)
L = oracle.port_lib()
L.oracle_probe_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_float] * 4 + [ctypes.c_void_p] * 3
L.oracle_probe_batch(o.nodes.ctypes.data, batch.ctypes.data, B, 0.0, 0.0, 0.0, float(max(box)), node.ctypes.data, addr.ctypes.data, isset.ctypes.data)
cand = isset == 0
print("batch", n_batches, "points", B, "candidates (deepest cell clear before the batch)", int(cand.sum()), "distinct deepest nodes", len(np.unique(node)))
lines = addr[cand] // 128
ul, cnt = np.unique(lines, return_counts=True)
print("distinct 128-B lines", len(ul), "max candidates per line", int(cnt.max()), "mean", float(cnt.mean()), "p99", float(np.percentile(cnt, 99)))
uw, cw = np.unique(addr[cand], return_counts=True)
print("distinct words", len(uw), "max per word", int(cw.max()), "mean per word", float(cw.mean()))
cells = np.unique(np.stack([addr[cand]], 1), axis=0)
alll, cl = np.unique(addr // 128, return_counts=True)
print("probes: distinct lines", len(alll), "max probes per line", int(cl.max()), "mean", float(cl.mean()))
