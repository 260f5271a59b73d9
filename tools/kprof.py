import sys, os, json, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree, lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "terrain"
pts, box = (synthetic.terrain(n, seed=7) if kind == "terrain" else synthetic.terrain_scan(n, seed=7) if kind == "scan" else synthetic.hotspot(n) if kind == "hotspot" else synthetic.uniform_cube(n))
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box, hqs=True)
L = lib()
for rep in range(2):
    dev.reset(u)
    if rep == 1:
        L.simlod_profile_enable(1)
        dev.momentary[152:264].zero_()
    dev.add_points(u, pts)
    torch.cuda.synchronize()
prof = bench.collect_profile(L)
st = dev.read_stats()
nb = (n + 999999) // 1000000
print("variant", os.environ.get("SIMLOD_VARIANT", "0"), "pts", int(st["numPoints"]), "voxels", int(st["numVoxels"]), "dbg", int(st["dbg"]))
print({k: (c, round(ms, 2), "%.0f us/batch" % (ms * 1e3 / nb)) for k, (c, ms) in prof.items() if "k_" in k})
if os.environ.get("SIMLOD_EXACT_CHAIN") == "bulk":     # construct_bulk.hip: Ctl.spilledTotal at byte 176, expandNs behind it
    c = dev.momentary[176:264].cpu().numpy().view(np.uint64)
    print("moved points %d, samples through k_place %d, voxels by k_place %d | k_expand: %d rounds in %d calls with spills" % (c[0], c[1], c[2], c[3 + 4], c[3 + 5]))
else:                                                   # construct_batch.hip: Ctl.expandNs at byte 152, [7] = moved points
    c = dev.momentary[152:216].cpu().numpy().view(np.uint64)
    print("moved points %d | k_expand: %d rounds in %d calls with spills" % (c[7], c[5], c[6]))
for hq in (1, 0):
    u["useHighQualityShading"] = hq
    dev.render(u); torch.cuda.synchronize()
    dev.render(u)
    pr = bench.collect_profile(L)
    print("hqs" if hq else "plain", {k: round(ms, 3) for k, (c, ms) in pr.items()})
