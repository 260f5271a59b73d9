"""Frames of a BIG octree — 500 M device-generated points streamed through the ring, what one rank of BASELINE config 4 holds —: both camera presets, plain and HQS,
frame time + per-kernel times (simlod_profile_enable(1): HIP events between the launches).   python tools/raster_big.py [points]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from simlod_amd import camera
from simlod_amd.runtime import DeviceOctree, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
W, H = 1920, 1080
tile = (6000.0, 4000.0, 400.0)
box = np.array(tile, dtype=np.float32)
dev = DeviceOctree("cuda:0", persistent_bytes=max(8 << 30, 48 * n), max_pixels=W * H)
src = torch.empty(n * 16, dtype=torch.uint8, device=dev.device)
dev.generate_terrain(src, 0, n, 7, 1, tile)
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
cx, cy = 2750.218, 974.775
T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, 0.35 * float(box[2]))), camera.perspective(aspect=W / H))
u = dev.uniforms(W, H, T, box, hqs=True)
dev.reset(u)
dev.stream(u, src, n)
del src
st = dev.read_stats()
print(f"{n} points: {int(st['numNodes'])} nodes, {int(st['numVoxels'])} voxels", flush=True)
L = lib()
for name, Tc in (("bird", T), ("close", T_close)):
    for hqs in (0, 1):
        uc = dev.uniforms(W, H, Tc, box, hqs=bool(hqs))
        for _ in range(3):
            dev.render(uc)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            dev.render(uc)
        e1.record(); torch.cuda.synchronize()
        s = dev.read_stats()
        L.simlod_profile_enable(1)
        for _ in range(5):
            dev.render(uc)
        p = bench.collect_profile(L)
        L.simlod_profile_enable(0)
        ks = "  ".join(f"{k.replace('MODE_', '')} {ms / c * 1e3:.1f}" for k, (c, ms) in p.items())
        print(f"{name:5s} {'hqs  ' if hqs else 'plain'} {e0.elapsed_time(e1) / 20:7.4f} ms/frame  visible {int(s['numVisiblePoints']) + int(s['numVisibleVoxels'])} samples in {int(s['numVisibleNodes'])} nodes | us per kernel (one stream, events): {ks}", flush=True)

# per draw item of the bird HQS frame's COLOUR pass (DrawItem::took: a library built with -DSIMLOD_MEASURE=1, SIMLOD_HIP_LIB=...; zeros otherwise)
item_dtype = np.dtype([("chunks", "<u8"), ("samples", "<u4"), ("visibleIdx", "<u4"), ("tileX", "<i4"), ("tileY", "<i4"), ("tileW", "<u2"), ("tileH", "<u2"), ("took", "<u4")])
for name, Tc in (("bird", T), ("close", T_close)):
    uc = dev.uniforms(W, H, Tc, box, hqs=True)
    for _ in range(2):
        dev.render(uc)
    torch.cuda.synchronize()
    off_work = int(dev.L.simlod_render_framebuffer_offset()) + (W * H * 8 + 15) // 16 * 16
    work = dev.render_buffer[off_work: off_work + 64].cpu().numpy().view(np.uint32)
    it = np.concatenate([dev.render_buffer[off_work + 256 + cl * 150000 * 32: off_work + 256 + (cl * 150000 + int(work[8 + cl])) * 32].cpu().numpy().view(item_dtype) for cl in range(4)])
    us = it["took"] / 100.0
    if us.sum() == 0:
        break
    area = it["tileW"].astype(int) * it["tileH"]
    print(f"== {name} hqs, colour pass: {len(it)} items, {int(it['samples'].sum())} samples, longest item {us.max():.1f} us")
    for k, m in {"sorting": it["tileX"] == -2, "no tile": it["tileX"] == -1, "exact tile (<= 8192 px)": (it["tileX"] >= 0) & (area <= 8192), "packed tile": (it["tileX"] >= 0) & (area > 8192)}.items():
        if m.any():
            print(f"   {k:24s} {int(m.sum()):5d} items, {int(it['samples'][m].sum()):9d} samples, {1e3 * us[m].sum() / max(int(it['samples'][m].sum()), 1):6.2f} ns per sample, longest {us[m].max():6.1f} us, samples per tile pixel {it['samples'][m].sum() / max(area[m].sum(), 1):5.1f}")
    for i in np.argsort(-us)[:6]:
        print(f"      {us[i]:6.1f} us  samples {int(it['samples'][i]):6d}  tile {int(it['tileX'][i]):5d},{int(it['tileY'][i]):5d} {int(it['tileW'][i]):3d}x{int(it['tileH'][i]):3d}")
