"""Race hunt at the parity tests' own shape: a fresh, poisoned DeviceOctree per pass (as tests/test_gpu_parity.py _device), a small terrain in two ring
batches taken by ONE launch (an exact group in which the root splits), then frames with and without screen bins; every Stats counter compared with pass 0.

    python tools/stress_small.py [--passes 300] [--kind terrain|hotspot|uniform]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=300)
ap.add_argument("--kind", default="terrain")
ap.add_argument("--keep", action="store_true", help="one DeviceOctree for all passes")
args = ap.parse_args()
FIELDS = ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksPoints", "numChunksVoxels", "batchletIndex",
          "numPointsProcessed", "numAllocatedChunks", "chunkPoolSize", "allocatedBytes_persistent", "dbg", "memCapacityReached", "numVisibleNodes"]
Wd, Hd = 1000, 562
if args.kind == "terrain":
    pts, box = synthetic.terrain(1_500_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
    ex, ey = 0.5 * float(box[0]), 0.3 * float(box[1])
    ground = synthetic.terrain_height(ex, ey, seed=3, box=(600.0, 400.0, 40.0))
    eye, target = (ex, ey, ground + 6.0), (ex + 20.0, ey + 200.0, ground - 4.0)
elif args.kind == "hotspot":
    pts, box = synthetic.hotspot(1_200_000, seed=11, level=4, cell=(5, 9, 6))
    c = (np.array([5, 9, 6], dtype=np.float32) + 0.5) / 16.0
    eye, target = tuple(c + np.float32(0.09) * np.array([1.4, -1.1, 0.9], dtype=np.float32)), tuple(c)
else:
    pts, box = synthetic.uniform_cube(1_000_000, seed=77)
    eye, target = (1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2])
T = camera.lookat_transform(eye, target, Wd, Hd)
first, bad, dev = None, 0, None
for p in range(args.passes):
    if dev is None or not args.keep:
        dev = DeviceOctree("cuda:0", persistent_bytes=8192 << 20, ring_slots=2, max_pixels=1920 * 1080)
        dev.momentary.fill_(0xA5); dev.render_buffer.fill_(0xA5); dev.persistent.fill_(0xA5)
        if p < 3:
            print("buffers:", {k: hex(getattr(dev, k).data_ptr()) for k in ("persistent", "momentary", "render_buffer", "ring", "nodes") if hasattr(getattr(dev, k, None), "data_ptr")}, flush=True)
    if p % 3 == 2:
        dev.tune("SIMLOD_RASTER_SCREEN_BINS", 0)
    u = dev.uniforms(Wd, Hd, T, box)
    dev.reset(u)
    for i in range(0, len(pts), 1_000_000):
        if dev.uploaded_host - dev.processed() >= dev.ring_slots:
            dev.drain(u)
        dev.upload(pts[i:i + 1_000_000])
    dev.drain(u)
    u["useHighQualityShading"] = p % 2
    dev.render(u)
    dev.render(u)
    st = dev.read_stats()
    got = {f: int(st[f]) for f in FIELDS}
    if first is None:
        first = got
        print("pass 0:", got, flush=True)
    elif got != first:
        bad += 1
        print(f"pass {p} differs:", {f: (first[f], got[f]) for f in FIELDS if got[f] != first[f]}, flush=True)
    if p % 50 == 49:
        print(f"pass {p}: {bad} differing so far", flush=True)
print(f"{args.passes} passes, {bad} differ from the first")
sys.exit(1 if bad else 0)
