"""Per-phase wall time inside k_ingest / k_place (SIMLOD_PHASE_TIMERS=1), summed over workgroups by the kernels themselves."""
import os, sys
os.environ["SIMLOD_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree
n = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
coalesce = len(sys.argv) > 2 and sys.argv[2] == "coalesce"
pts, box = synthetic.terrain(n, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, momentary_bytes=700_000_000 if coalesce else 300_000_000, max_pixels=W * H, coalesce=coalesce)
u = dev.uniforms(W, H, T, box)
for rep in range(2):
    dev.reset(u)
    dev.momentary[432:432 + 192].zero_()
    dev.add_points(u, pts)
    torch.cuda.synchronize()
ph = dev.momentary[432:432 + 192].cpu().numpy().view(np.uint64).astype(np.float64) / 1e3
nb = (n + 999999) // 1000000
names = ["load+descend+count", "flush points (atomics, chunk alloc, lookups)", "store + sample", "flush voxels", "store voxels (+ queue entries)"]
vn = ["cube load", "pass A", "write-back", "slot ranges + chunks", "pass B", "light path (whole)"]
tot = ph[16:22].sum()
print("k_voxelize sum over workgroups %.0f us per batch" % (tot / nb), {vn[i]: "%.0f us (%.0f%%)" % (ph[16 + i] / nb, 100 * ph[16 + i] / max(tot, 1)) for i in range(6)})
for base, k in ((0, "k_ingest"), (8, "k_place")):
    tot = ph[base:base + 5].sum()
    print(k, "sum over workgroups %.0f us per batch" % (tot / nb), {names[i]: "%.0f us (%.0f%%)" % (ph[base + i] / nb, 100 * ph[base + i] / max(tot, 1)) for i in range(5)})
