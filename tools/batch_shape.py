"""What one batch of the bench terrain looks like to the builder: per 1 M-point batch (one per launch) the leaves it touches, the leaves that
cross the limit, the stored points the splits move, k_voxelize's pieces (>= 512 new samples of one leaf, <= VOX_PIECE each) and small items —
read from the control block at byte 0 of the momentary buffer (construct.hip Ctl / BatchCtl)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
pts, box = synthetic.terrain(n, seed=7)
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=1920 * 1080)
u = dev.uniforms(1920, 1080, np.eye(4), box)
dev.reset(u)
BATCH0 = 1080          # offsetof(Ctl, batch): construct.hip static_asserts the fields before it
rows = []
for b in range(0, n, abi.MAX_BATCH_SIZE):
    dev.upload(pts[b:b + abi.MAX_BATCH_SIZE])
    dev.drain(u)
    torch.cuda.synchronize()
    w = dev.momentary[BATCH0: BATCH0 + 64].cpu().numpy().view(np.uint32)
    rows.append(dict(spilled=int(w[7]), work=int(w[8]), clear=int(w[9]), touched=int(w[10]), cross=int(w[11]), pieces=int(w[14]), small=int(w[15])))
for i, r in enumerate(rows):
    print("batch %2d touched %4d cross %3d moved %6d | pieces %4d small items %5d" % (i, r["touched"], r["cross"], r["spilled"], r["pieces"], r["small"]))
print("mean pieces %.1f small %.1f touched %.1f cross %.1f" % tuple(np.mean([r[k] for r in rows]) for k in ("pieces", "small", "touched", "cross")))
