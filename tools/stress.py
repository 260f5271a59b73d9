"""Race hunt: the same ingest over and over, every build counter of Stats (which in exact mode are the reference's, hence deterministic)
compared with the first pass.  The two-stream pipeline's bugs showed up as a few thousand missing voxels once in a while.

    python tools/stress.py [--passes 200] [--points 36000000] [--stream] [--coalesce]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=200)
ap.add_argument("--points", type=int, default=36_000_000)
ap.add_argument("--stream", action="store_true", help="device-generated points through the ring with the uploader (DeviceOctree.stream)")
ap.add_argument("--coalesce", action="store_true")
args = ap.parse_args()
FIELDS = ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksPoints", "numChunksVoxels", "batchletIndex",
          "numPointsProcessed", "numAllocatedChunks", "chunkPoolSize", "allocatedBytes_persistent", "dbg", "memCapacityReached"]
if args.coalesce:
    FIELDS = [f for f in FIELDS if f not in ("chunkPoolSize", "allocatedBytes_persistent")]
W, H = 1920, 1080
batch = abi.MAX_BATCH_SIZE
dev = DeviceOctree("cuda:0", persistent_bytes=max(8 << 30, 48 * args.points), momentary_bytes=(700 if args.coalesce else 300) * 1_000_000, max_pixels=W * H, coalesce=args.coalesce)
if args.stream:
    box = (6000.0, 4000.0, 400.0)
    src = torch.empty(args.points * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(src, 0, args.points, 7, 1, box)
else:
    pts, box = synthetic.terrain(args.points, seed=7)
    nb = (args.points + batch - 1) // batch
    rv = dev.ring.view(torch.uint8)
    for i in range(nb):
        c = pts[i * batch:(i + 1) * batch]
        rv[i * batch * 16: i * batch * 16 + len(c) * 16].copy_(torch.from_numpy(c.view(np.uint8).reshape(-1)))
    sizes = torch.tensor([min(batch, args.points - i * batch) for i in range(nb)], dtype=torch.int32, device=dev.device)
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
u = dev.uniforms(W, H, T, box, hqs=True)
first, bad = None, 0
for p in range(args.passes):
    dev.reset(u)
    if args.stream:
        dev.stream(u, src, args.points)
    else:
        dev.batch_sizes[:nb] = sizes
        dev.publish(nb)
        dev.uploaded_host = nb
        dev.drain(u)
    st = dev.read_stats()
    got = {f: int(st[f]) for f in FIELDS}
    if first is None:
        first = got
        print("pass 0:", got, flush=True)
    elif got != first:
        bad += 1
        print(f"pass {p} differs:", {f: (first[f], got[f]) for f in FIELDS if got[f] != first[f]}, flush=True)
print(f"{args.passes} passes, {bad} differ from the first")
sys.exit(1 if bad else 0)
