"""Per-bin view of the screen bins of one close-up frame on the bench octree (render.hip r_overflow's measurement record): how many
entries and segments each 32 x 32-pixel bin held and how long its workgroup took.

    python tools/raster_bins.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree

pts, box = synthetic.terrain(36_000_000, seed=7)
W, H = 1920, 1080
cx, cy = 2750.218 * float(box[0]) / 6000.0, 974.775 * float(box[1]) / 4000.0
T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, synthetic.terrain_height(cx, cy, seed=7, box=tuple(float(v) for v in box)))), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u0 = dev.uniforms(W, H, T_close, box, hqs=False)
dev.reset(u0)
dev.add_points(u0, pts)
for _ in range(3):
    dev.render(u0)
torch.cuda.synchronize()
px = W * H
a16 = lambda v: (v + 15) // 16 * 16
tx, ty = (W >> 5) + 1, (H >> 5) + 1
tiles = tx * ty
off = int(dev.L.simlod_render_framebuffer_offset()) + a16(px * 8) + 256 + 150000 * 4 * 32 + a16(px * 4) + a16(px * 8) + px * 16 + 2_000_000 * 8 + 3_000_000 * 16 + tiles * 256 * 8 + a16(tiles * 4)
st = dev.render_buffer[off: off + tiles * 8].cpu().numpy().view(np.uint32).reshape(tiles, 2)
entries, segs, took = st[:, 0].astype(np.int64), st[:, 1] >> 20, (st[:, 1] & 0xfffff) / 100.0
print(f"{tiles} bins ({tx} x {ty}); with entries: {(entries > 0).sum()}; entries {entries.sum()}; binned {dev.samples_binned(W, H)}; outside {dev.samples_outside_tiles()}")
order = np.argsort(-took)[:12]
for i in order:
    print(f"  bin {i % tx:2d},{i // tx:2d}: {entries[i]:7d} entries in {segs[i]:3d} segments, {took[i]:6.1f} us")
print("entries per bin row (thousands):", " ".join(f"{int(entries[r * tx:(r + 1) * tx].sum()) // 1000}" for r in range(ty)))
