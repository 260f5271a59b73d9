"""GPU time of kernel_construct launches on the bench terrain (36 M points resident in the ring): (a) ONE batch per launch — the reference's frame
loop when the loader is the bottleneck (main_progressive_octree.cpp:364-428: one launch per frame, whatever has been uploaded) —, events around
every launch; (b) the bench's step (two launches, 20 + 16 batches) with the gap between the two launches' kernels taken from the launch times.

    python tools/launch_cost.py [--points 36000000] ["ENV=V ..." ...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=36_000_000)
ap.add_argument("variants", nargs="*", default=[""])
args = ap.parse_args()
W, H = 1920, 1080
pts, box = synthetic.terrain(args.points, seed=7)
batch = abi.MAX_BATCH_SIZE
nb = (args.points + batch - 1) // batch
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box, hqs=True)
rv = dev.ring.view(torch.uint8)
for i in range(nb):
    c = pts[i * batch:(i + 1) * batch]
    rv[i * batch * 16: i * batch * 16 + len(c) * 16].copy_(torch.from_numpy(c.view(np.uint8).reshape(-1)))
sizes = torch.tensor([min(batch, args.points - i * batch) for i in range(nb)], dtype=torch.int32, device=dev.device)
for var in args.variants:
    saved = {}
    for kv in var.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k); os.environ[k] = v
    dev.reload_env()
    res = []
    for rep in range(3):
        dev.reset(u)
        dev.batch_sizes[:nb] = sizes
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nb)]
        for b in range(nb):
            dev.publish(b + 1)
            dev.uploaded_host = b + 1
            ev[b][0].record()
            dev.construct(u)
            ev[b][1].record()
        torch.cuda.synchronize()
        st = dev.read_stats()
        assert int(st["numPoints"]) == args.points and int(st["dbg"]) == 0
        res.append([e0.elapsed_time(e1) * 1e3 for e0, e1 in ev])
    r = np.array(res[1:])
    total = r.sum(axis=1).mean()
    print(f"{var or 'default':40s} one batch per launch: median {np.median(r):6.1f} us, mean {r.mean():6.1f} us per launch (first {r[:, 0].mean():6.1f}), {nb} launches {total / 1e3:6.3f} ms = {args.points / total:6.0f} M pts/s", flush=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
