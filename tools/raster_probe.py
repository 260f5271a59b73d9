"""Phase clock of r_visible on the bench octree (a library built with -DVAR_PROBE: SIMLOD_HIP_LIB=...): when, after the first workgroup
started, the last wave passed each point of the kernel.

    SIMLOD_HIP_LIB=/path/probe.so python tools/raster_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simlod_amd import camera, synthetic, abi
from simlod_amd.runtime import DeviceOctree

pts, box = synthetic.terrain(36_000_000, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u0 = dev.uniforms(W, H, T, box, hqs=False)
dev.reset(u0)
dev.add_points(u0, pts)
off = abi.MAX_VISIBLE_NODES * abi.node_dtype.itemsize + 7 * 16 + 32 + 8_000_000
names = ["first start", "numNodes loaded (last)", "node fields + geometry (last)", "reservations returned (last)", "items stored (last)", "visible_nodes done (last)",
         "kernel end (last)", "kernel end (first)", "frame-ready seen (last)", "box on screen (last)", "before reservations (last)", "had to wait for frame-ready (last)"]
for hqs in (False, True):
    u = dev.uniforms(W, H, T, box, hqs=hqs)
    for rep in range(4):
        dev.render_buffer[off: off + 12 * 8192 * 8].zero_()
        torch.cuda.synchronize()
        dev.render(u)
        torch.cuda.synchronize()
        v = dev.render_buffer[off: off + 12 * 8192 * 8].cpu().numpy().view(np.uint64).reshape(12, 8192).astype(np.int64)
        t0 = v[0][v[0] > 0].min()
        if rep >= 2:
            print("hqs  " if hqs else "plain", " | ".join(f"{names[k]} {((v[k][v[k] > 0].min() if k in (0, 7) else max(v[k].max(), t0)) - t0) / 100.0:6.1f}" for k in (0, 1, 2, 9, 10, 11, 8, 3, 4, 5, 7, 6)),
                  f"| last start {(v[0].max() - t0) / 100.0:6.1f} | waves that emit {(v[3] > 0).sum()}")
