"""Frame times of the rasteriser on the bench octree (36 M terrain) for both camera presets of bench.py — "Morro Bay - bird" and
"Morro Bay - close" — plain and HQS, with the share of samples that took the global-atomic path (outside their item's LDS tile).

    python tools/raster_close.py [frames] ["NAME=VALUE ..." ...]      (every further argument: one variant of the tuning knobs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 36_000_000
pts, box = synthetic.terrain(n, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
cx, cy = 2750.218 * float(box[0]) / 6000.0, 974.775 * float(box[1]) / 4000.0
T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, synthetic.terrain_height(cx, cy, seed=7, box=tuple(float(v) for v in box)))), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u0 = dev.uniforms(W, H, T, box, hqs=False)
dev.reset(u0)
dev.add_points(u0, pts)
for var in (sys.argv[2:] or [""]):
    for kv in var.split():
        os.environ[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    dev.reload_env()
    for name, Tc in [q for q in (("bird", T), ("close", T_close)) if q[0] in os.environ.get("PRESETS", "bird,close").split(",")]:
        for hqs in (0, 1):
            u = dev.uniforms(W, H, Tc, box, hqs=bool(hqs))
            for _ in range(3):
                dev.render(u)
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(frames):
                dev.render(u)
            t1.record(); torch.cuda.synchronize()
            st = dev.read_stats()
            vs = int(st["numVisiblePoints"]) + int(st["numVisibleVoxels"])
            ms = t0.elapsed_time(t1) / frames
            print(f"{var or 'default':32s} {name:5s} {'hqs  ' if hqs else 'plain'} {ms:7.4f} ms/frame  {vs / ms / 1e6:7.1f} G samples/s  visible {vs:9d} samples, {int(st['numVisibleNodes']):5d} nodes; "
                  f"outside tiles {dev.samples_outside_tiles():9d} ({100.0 * dev.samples_outside_tiles() / max(vs, 1):5.1f} %), binned {dev.samples_binned(W, H):9d}", flush=True)
    for kv in var.split():
        os.environ.pop(kv.split("=", 1)[0], None)
