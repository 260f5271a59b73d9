"""Build the bench octree (36 M terrain), then draw FRAMES plain and FRAMES HQS frames: run under
`rocprofv3 --kernel-trace --stats` to get the per-kernel times of the rasteriser alone (tools/raster_prof.sh).

    python tools/raster_prof.py [points] [frames] ["NAME=VALUE ..." ...]

Every further argument is one variant: environment settings for the library's tuning knobs (DeviceOctree.reload_env() before each), timed one after
the other on the same octree ("" = defaults)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simlod_amd import camera, synthetic
from simlod_amd.runtime import DeviceOctree

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pts, box = synthetic.terrain(n, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box, hqs=False)
dev.reset(u)
dev.add_points(u, pts)
variants = sys.argv[3:] or [""]
for var, hqs in [(v, h) for v in variants for h in (0, 1)]:
    for kv in var.split():
        os.environ[kv.split("=", 1)[0]] = kv.split("=", 1)[1]
    dev.reload_env()
    u["useHighQualityShading"] = hqs
    for _ in range(frames):
        dev.render(u)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(frames):
        dev.render(u)
    t1.record(); torch.cuda.synchronize()
    for kv in var.split():
        os.environ.pop(kv.split("=", 1)[0], None)
    print(f"{var or 'default':40s}", "hqs" if hqs else "plain", "ms/frame", t0.elapsed_time(t1) / frames, "lists through table", dev.lists_read_through_table(), flush=True)
