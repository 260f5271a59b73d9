"""k_expand phase times of workgroup 0 (construct_batch.hip: Ctl.expandNs at byte 152) on the bench workload."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree
pts, box = synthetic.terrain(36_000_000, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box, hqs=True)
dev.reset(u)
dev.momentary[152:216].zero_()
dev.add_points(u, pts)
torch.cuda.synchronize()
c = dev.momentary[152:216].cpu().numpy().view(np.uint64)
print("calls with spills %d, rounds %d | us total: split %.0f barrier %.0f copy %.0f recount %.0f barrier %.0f | moved points %d" % (c[6], c[5], c[0] / 1e3, c[1] / 1e3, c[2] / 1e3, c[3] / 1e3, c[4] / 1e3, c[7]))
print("per round us: split %.1f barrier %.1f copy %.1f recount %.1f barrier %.1f" % tuple(c[:5] / 1e3 / max(int(c[5]), 1)))
