"""Per-batch kernel times of one ingest: uploads one batch, launches kernel_construct, collects the profile; repeats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree, lib
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "terrain"
pts, box = (synthetic.terrain(n, seed=7) if kind == "terrain" else synthetic.terrain_scan(n, seed=7))
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box, hqs=True)
L = lib()
for rep in range(2):
    dev.reset(u)
    rows = []
    for b in range(0, n, 1_000_000):
        dev.upload(pts[b:b + 1_000_000])
        torch.cuda.synchronize()
        L.simlod_profile_enable(1)
        dev.drain(u)
        torch.cuda.synchronize()
        p = bench.collect_profile(L)
        L.simlod_profile_enable(0)
        st = dev.read_stats()
        rows.append((b // 1_000_000, int(st["numNodes"]), int(st["numVoxels"]), {k.split("<")[0]: round(ms * 1e3) for k, (c, ms) in p.items() if k.startswith("k_")}))
prev = 0
for b, nn, nv, p in rows:
    print("batch %2d nodes %5d newvox %7d | count %3d queue %3d hist %3d expand %4d insert %3d voxelize %4d" % (b, nn, nv - prev, p.get("k_count", 0), p.get("k_queue", 0), p.get("k_hist", 0), p.get("k_expand", 0), p.get("k_insert", 0), p.get("k_voxelize", 0)))
    prev = nv
