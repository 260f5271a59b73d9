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
n = c[4]
m=int(c[5]); print("max wait %.1f us batch %d slot %d | node level %d spilled-sample %d root-leaf %d startLevel %d | attempts/chunk l0 %.1f l1-2 %.1f" % ((m>>40)/100.0, (m>>32)&255, (m>>30)&3, (m>>12)&31, (m>>11)&1, (m>>10)&1, (m>>5)&31, c[6]/n, c[7]/n))
print("wave-chunks", n, "avg us per wave-chunk: stage1 %.2f stage2 %.2f stage3 %.2f claim %.2f" % tuple(c[:4] / n / 100.0))
