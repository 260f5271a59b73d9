"""Per-item view of one frame of the rasteriser on the bench octree: how long each draw item's workgroup took over it (render.hip
DrawItem::took, written by the frame's last draw pass), by item kind.

    PRESETS=close python tools/raster_items.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree

pts, box = synthetic.terrain(36_000_000, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
cx, cy = 2750.218 * float(box[0]) / 6000.0, 974.775 * float(box[1]) / 4000.0
T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, synthetic.terrain_height(cx, cy, seed=7, box=tuple(float(v) for v in box)))), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u0 = dev.uniforms(W, H, T, box, hqs=False)
dev.reset(u0)
dev.add_points(u0, pts)
item_dtype = np.dtype([("chunks", "<u8"), ("samples", "<u4"), ("visibleIdx", "<u4"), ("tileX", "<i4"), ("tileY", "<i4"), ("tileW", "<u2"), ("tileH", "<u2"), ("took", "<u4")])
for name, Tc in [q for q in (("bird", T), ("close", T_close)) if q[0] in os.environ.get("PRESETS", "bird,close").split(",")]:
    u = dev.uniforms(W, H, Tc, box, hqs=False)
    for _ in range(3):
        dev.render(u)
    torch.cuda.synchronize()
    off_work = int(dev.L.simlod_render_framebuffer_offset()) + (W * H * 8 + 15) // 16 * 16
    work = dev.render_buffer[off_work: off_work + 64].cpu().numpy().view(np.uint32)
    cap = 150000
    items = []
    for cl in range(4):
        n = int(work[8 + cl])
        a = dev.render_buffer[off_work + 256 + cl * cap * 32: off_work + 256 + (cl * cap + n) * 32].cpu().numpy().view(item_dtype)
        items.append(a)
    it = np.concatenate(items)
    us = it["took"] / 100.0
    print(f"== {name}: {len(it)} items, {int(it['samples'].sum())} samples; sum of item times {us.sum():.0f} us = {us.sum() / 256:.1f} us per workgroup of 256; longest item {us.max():.1f} us")
    kinds = {"sorting": it["tileX"] == -2, "no tile": it["tileX"] == -1, "tile 128x128": (it["tileX"] >= 0) & (it["tileW"].astype(int) * it["tileH"] >= 128 * 128), "smaller tile": (it["tileX"] >= 0) & (it["tileW"].astype(int) * it["tileH"] < 128 * 128)}
    for k, m in kinds.items():
        if m.any():
            print(f"   {k:14s} {int(m.sum()):5d} items, {int(it['samples'][m].sum()):9d} samples, {us[m].sum():8.0f} us in all, {1e3 * us[m].sum() / max(int(it['samples'][m].sum()), 1):6.2f} ns per sample, longest {us[m].max():6.1f} us, mean tile {it['tileW'][m].mean():.0f} x {it['tileH'][m].mean():.0f}")
    big = np.argsort(-us)[:8]
    for i in big:
        print(f"      {us[i]:6.1f} us  samples {int(it['samples'][i]):6d}  tile {int(it['tileX'][i]):5d},{int(it['tileY'][i]):5d} {int(it['tileW'][i]):3d}x{int(it['tileH'][i]):3d}")
