#!/usr/bin/env python3
"""Generates tests/golden/edl_uniform_3x40k.npz: the EDL-shaded RGBA8 output of the reference's OWN render.cu (EDL block :1255-1325,
surface write :1334-1343) for one seeded case, plain and HQS.

The host build of the reference runs one thread; the EDL block shades pixel #thread_rank of each 16x16 tile.  oracle/_ref/
libref_render_edl.so (render.cu compiled with -DSIMLOD_SHIM_EDL, oracle/shim/simlod_host_shim.h) lets one block take every full tile
and hands the block's thread rank in from outside, so 256 calls with rank 0..255 — each re-rendering the same frame — apply the
reference's EDL arithmetic to every pixel of every full tile.  After call #rank, pixel #rank of every tile is collected.
Needs /root/reference (make -C oracle ref); run from the repo root:  python tests/golden/make_golden_edl.py"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from cases import W, H, batches_of, case, uniforms_for  # noqa: E402

NAME = "uniform_3x40k"
pts, box, batch, T = case(NAME)
u = uniforms_for(box, T)
o = oracle.HostOctree("ref", persistent_bytes=1 << 30, ring_slots=8)
o.reset(u)
for b in batches_of(NAME, pts, batch):
    o.upload(b)
while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
    o.construct(u)

edl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_render_edl.so"))
rank = ctypes.c_int.in_dll(edl, "simlod_shim_edl_rank")
in_edl = ctypes.c_int.in_dll(edl, "simlod_shim_in_edl")
ctypes.c_int.in_dll(edl, "simlod_shim_surface_width").value = W
fn = ctypes.cast(edl.kernel_render, ctypes.c_void_p)
need = 15_200_000 + 64 + 32 + 16_000_000 + W * H * 8 + W * H * 20 + 4096
buf = np.zeros(need, dtype=np.uint8)
cudaprint = np.zeros(1024 * 1001, dtype=np.uint8)
p = lambda a: ctypes.c_void_p(a.ctypes.data)
out = {}
ty, tx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
in_tile = (tx % 16) + 16 * (ty % 16)                       # the thread rank that shades a pixel (render.cu:1284)
full = (tx < (W // 16) * 16) & (ty < (H // 16) * 16)
for mode, hqs in (("plain", False), ("hqs", True)):
    uu = np.ascontiguousarray(uniforms_for(box, T, hqs=hqs)).reshape(1)
    image = np.zeros(W * H, dtype=np.uint32)
    for r in range(256):
        rank.value, in_edl.value = r, 0
        color = np.zeros(W * H, dtype=np.uint32)
        stats = o.stats.copy()
        oracle.port_lib().ref_call_render(fn, p(buf), p(uu), p(o.nodes), p(color), p(stats), p(o.frame_start), p(cudaprint))
        sel = (full & (in_tile == r)).reshape(-1)
        image[sel] = color[sel]
        if r == 0:
            unshaded = ~full.reshape(-1)
            image[unshaded] = color[unshaded]              # outside the full tiles the surface gets the unshaded colour
    out[f"color_{mode}"] = image
    print(mode, "shaded pixels that differ from the unshaded colour:", int((image != (color & 0xffffffff)).sum()))
np.savez_compressed(os.path.join(HERE, f"edl_{NAME}.npz"), **out)
