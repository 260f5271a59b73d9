"""Exploration: the 200 M-point hotspot (BASELINE config 5) on the device with a momentary buffer big enough that no split is deferred, and
through the oracle: does the reference's own algorithm stay inside its capacity limits on this input?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from simlod_amd import abi, camera
from simlod_amd.runtime import DeviceOctree
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
mom = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000_000
Wd, Hd = 1920, 1080
dev = DeviceOctree("cuda:0", persistent_bytes=48 * n, momentary_bytes=mom, max_pixels=Wd * Hd)
g = torch.Generator(device=dev.device); g.manual_seed(11)
src = torch.empty((n, 4), dtype=torch.int32, device=dev.device)
cell = torch.tensor([21.0, 40.0, 13.0], device=dev.device) / 64.0
for first in range(0, n, 50_000_000):
    r = torch.rand((min(50_000_000, n - first), 3), generator=g, device=dev.device, dtype=torch.float32)
    src[first: first + len(r), :3] = (cell + r * (0.999 / 64.0)).view(torch.int32)
    c = (r * 255.0).to(torch.int32)
    src[first: first + len(r), 3] = c[:, 0] + c[:, 1] * 256 + c[:, 2] * 65536 - 16777216
box = np.array([1.0, 1.0, 1.0], dtype=np.float32)
center = np.array([21, 40, 13], dtype=np.float64) / 64 + 1 / 128
dist = (1 / 64) * (Hd / 128.0) / (2 * np.tan(np.radians(30)))
T = camera.lookat_transform(center + np.array([0.6, -0.7, 0.4]) / np.linalg.norm([0.6, -0.7, 0.4]) * dist, center, Wd, Hd)
u = dev.uniforms(Wd, Hd, T, box, min_node_size=8.0)
dev.reset(u)
t0 = time.time(); launches = dev.stream(u, src.view(torch.uint8).reshape(-1), n); torch.cuda.synchronize(); t1 = time.time()
ds = dev.read_stats()
print("device:", {k: int(ds[k]) for k in ("dbg", "numNodes", "numPoints", "numVoxels", "batchletIndex", "allocatedBytes_persistent", "chunkPoolSize")}, "launches", launches, "s", round(t1 - t0, 2), flush=True)
pts = src.cpu().numpy().view(np.uint8).reshape(-1).view(abi.point_dtype)
ref = oracle.HostOctree("port", persistent_bytes=48 * n, ring_slots=abi.BATCH_STREAM_SIZE)
ref.reset(u)
t0 = time.time()
firstErr = None
for i in range(0, n, abi.MAX_BATCH_SIZE):
    ref.upload(pts[i:i + abi.MAX_BATCH_SIZE])
    ref.construct(u)
    if firstErr is None and ref.last_error() != 0:
        firstErr = (i // abi.MAX_BATCH_SIZE, ref.last_error(), int(ref.stats["numPoints"][0]), int(ref.stats["numPointsProcessed"][0]))
print("oracle:", {k: int(ref.stats[k][0]) for k in ("numNodes", "numPoints", "numVoxels", "batchletIndex", "allocatedBytes_persistent", "chunkPoolSize")}, "last_error", ref.last_error(), "first error at (batch, code, numPoints, processed)", firstErr, "s", round(time.time() - t0, 1), flush=True)
