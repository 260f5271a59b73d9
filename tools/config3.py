"""BASELINE config 3 as stated: "Morro Bay 350M (.las) streamed in 1M-point batches, incremental octree growth, 1xMI355X".

1. writes the stand-in as a REAL LAS 1.4 file (point format 2, 26-byte records) in acquisition order: the tiled-terrain generator of
   simlod_generate_terrain (swath by swath, the order an airborne scanner writes) makes the points on the device, this script packs
   them into LAS records (int32 X, Y, Z at scale 0.001 around a UTM-like offset, RGB16) and appends them to the file;
2. streams the file through harness/_ref/ref_host_replay — the reference's OWN resetCUDA / updateOctree / renderCUDA / initCudaProgram
   (cut out of main_progressive_octree.cpp at build time) around this library — with the uploader on its own thread and stream, the raw
   records page-locked, the LAS decode on the device in the upload stream (LasLoader.cpp:169-227 -> simlod_decode_las);
3. for comparison, the adversarial order of round 2: the same number of LCG-scattered points (`synthetic:N`), no LAS bytes;
4. writes profiles-ready text + JSON to gpurun_out/.

    python tools/config3.py [--points 350000000] [--out gpurun_out/config3_350m]"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import lasio
from simlod_amd.runtime import DeviceOctree

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=350_000_000)
ap.add_argument("--out", default="gpurun_out/config3_350m")
ap.add_argument("--las", default="/tmp/config3_scan.las")
ap.add_argument("--skip-adversarial", action="store_true")
ap.add_argument("--swath", type=float, default=250.0, help="width of a flight line in metres (synthetic.terrain_scan's default)")
ap.add_argument("--density", choices=["config2", "dense"], default="config2",
                help="config2: the 36 M stand-in's density (1.5 points/m2): the terrain grows to 18.7 km x 12.5 km; dense: 350 M points over the same 6 km x 4 km (14.6 points/m2)")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = args.points
k = (n / 36_000_000.0) ** 0.5 if args.density == "config2" else 1.0
extent = (6000.0 * k, 4000.0 * k, 400.0)
world_min = np.array([694000.0, 3915000.0, -3.0])
scale = 0.001

t0 = time.perf_counter()
dev = DeviceOctree("cuda:0", persistent_bytes=1 << 20, momentary_bytes=1 << 20, max_pixels=64, ring_slots=1)
lasio.write_las(args.las, np.zeros((0, 26), dtype=np.uint8), 2, (scale,) * 3, (0.0, 0.0, 0.0), world_min, world_min + np.array(extent), version=(1, 4), num_points=n)
chunk = 50_000_000
with open(args.las, "ab") as f:
    for first in range(0, n, chunk):
        m = min(chunk, n - first)
        pts = torch.empty(m * 16, dtype=torch.uint8, device=dev.device)
        dev.generate_terrain(pts, first, n, 7, 1, extent, swath_width=args.swath)
        rec16 = pts.view(m, 16)
        xyz = rec16[:, :12].contiguous().view(torch.float32).view(m, 3).to(torch.float64) + torch.tensor(world_min, dtype=torch.float64, device=dev.device)
        xi = torch.round(xyz / scale).to(torch.int32)
        rgb8 = rec16[:, 12:15].to(torch.int32)
        rgb16 = (rgb8 * 257).to(torch.int16)                        # 16-bit colour as scanners write it; decodes back to rgb8 (LasLoader.cpp:177-185)
        rec = torch.zeros((m, 26), dtype=torch.uint8, device=dev.device)
        rec[:, 0:12] = xi.contiguous().view(torch.uint8).view(m, 12)
        rec[:, 20:26] = rgb16.contiguous().view(torch.uint8).view(m, 6)
        f.write(rec.cpu().numpy().tobytes())
        del pts, rec16, xyz, xi, rgb8, rgb16, rec
t_file = time.perf_counter() - t0
del dev
torch.cuda.empty_cache()
print(f"wrote {args.las}: {n} points, {os.path.getsize(args.las) / 1e9:.2f} GB in {t_file:.1f} s", flush=True)


def run(cmd, env_extra):
    env = dict(os.environ, **env_extra)
    t = time.perf_counter()
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True)
    return p.returncode, p.stdout + p.stderr, time.perf_counter() - t


def parse(out):
    r = {}
    m = re.search(r"load\+build wall ([\d.]+) ms .*update kernel ([\d.]+) ms total over (\d+) launches = ([\d.]+) M points/s, render kernel ([\d.]+) ms/frame", out)
    if m:
        r.update(wall_ms_incl_h2d=float(m.group(1)), update_kernel_ms=float(m.group(2)), launches=int(m.group(3)), kernel_M_points_per_s=float(m.group(4)), render_ms_per_frame=float(m.group(5)))
        r["wall_M_points_per_s"] = n / (r["wall_ms_incl_h2d"] * 1e-3) / 1e6
    m = re.search(r"numNodes (\d+) numInner (\d+) numLeaves (\d+) numPoints (\d+) numVoxels (\d+) persistentBytes (\d+) chunkPoolSize (\d+) dbg (\d+)", out)
    if m:
        r.update(numNodes=int(m.group(1)), numInner=int(m.group(2)), numLeaves=int(m.group(3)), numPoints=int(m.group(4)), numVoxels=int(m.group(5)), persistentBytes=int(m.group(6)), dbg=int(m.group(8)))
    return r


from simlod_amd.fingerprint import csrc_sha16
text, result = [], {"_csrc_sha16": csrc_sha16(), "points": n, "terrain_extent_m": extent, "points_per_m2": n / (extent[0] * extent[1]), "las_file_bytes": os.path.getsize(args.las), "las_write_s": t_file}
text.append(f"# BASELINE config 3: {n} points, fractal terrain {extent[0]:.0f} m x {extent[1]:.0f} m ({n / (extent[0] * extent[1]):.1f} points/m2), LAS 1.4 format 2, scan order\n")
host = os.path.join("harness", "_ref", "ref_host_replay")
if not os.path.exists(os.path.join(ROOT, host)):
    host = os.path.join("harness", "simlod_headless")
for label, cmd, env in (("scan-ordered LAS 1.4 file, decode on the device in the upload stream, raw records page-locked", [host, args.las, "/tmp/config3.ppm", "1920", "1080"], {"SIMLOD_HARNESS_PINNED": "1"}),
                        ("the same from pageable memory (staging memcpy per batch)", [host, args.las, "/tmp/config3.ppm", "1920", "1080"], {})):
    rc, out, secs = run(cmd, env)
    text.append(f"## {label}\n$ {' '.join(f'{k}={v}' for k, v in env.items())} {' '.join(cmd)}   (exit {rc}, {secs:.1f} s incl. reading the file)\n{out}")
    result["las_scan_pinned" if env else "las_scan_pageable"] = dict(parse(out), exit=rc, host=host)
if not args.skip_adversarial:
    rc, out, secs = run([os.path.join("harness", "simlod_headless"), f"synthetic:{n}", "/tmp/config3_adv.ppm", "1920", "1080"], {"SIMLOD_HARNESS_PINNED": "1"})
    text.append(f"## adversarial order (round 2's run): {n} LCG-scattered points, no LAS bytes, every batch touches tens of thousands of leaves\n$ SIMLOD_HARNESS_PINNED=1 harness/simlod_headless synthetic:{n}   (exit {rc}, {secs:.1f} s)\n{out}")
    result["adversarial_scatter"] = dict(parse(out), exit=rc)
os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
open(os.path.join(ROOT, args.out + ".txt"), "w").write("\n".join(text))
json.dump(result, open(os.path.join(ROOT, args.out + ".json"), "w"), indent=1)
print("\n".join(text))
try:
    os.remove(args.las)
except OSError:
    pass
