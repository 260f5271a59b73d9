"""GPU and host time of a kernel_construct launch that finds nothing to do (the frame loop of the reference host launches it every frame),
for a host that never sizes its launches (the reference's own: the library's prediction, which enqueues no group while the loader is idle), and for one that
does, with limits of 20 and 1 (simlod_set_construct_batch_limit: at least one group per launch)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simlod_amd import camera, synthetic, abi
from simlod_amd.runtime import DeviceOctree
pts, box = synthetic.terrain(2_000_000, seed=7)
W, H = 1920, 1080
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
plain = DeviceOctree("cuda:0", persistent_bytes=2 << 30, max_pixels=W * H, sizes_launches=False)
u = plain.uniforms(W, H, T, box)
plain.reset(u); plain.add_points(u, pts)
for _ in range(5): plain.construct(u); torch.cuda.synchronize()          # (the reports of two launches that found nothing: the loader is idle)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(50): plain.construct(u)
e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize()
print("host that never sizes its launches: idle launch %.0f us on the GPU, %.0f us of host time to enqueue" % (e0.elapsed_time(e1) * 1e3 / 50, th * 1e6 / 50))
del plain
dev = DeviceOctree("cuda:0", persistent_bytes=2 << 30, max_pixels=W * H)
u = dev.uniforms(W, H, T, box)
dev.reset(u); dev.add_points(u, pts)
for limit in (abi.MAX_BATCHES_PER_LAUNCH, 1):
    dev.set_batch_limit(limit)
    for ov in ("1", "0"):
        dev.tune("SIMLOD_OVERLAP_TAIL", int(ov))
        for _ in range(5): dev.construct(u)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(50): dev.construct(u)
        e1.record(); th = time.perf_counter() - t0; torch.cuda.synchronize()
        print("batch limit %2d overlap %s: idle launch %.0f us on the GPU, %.0f us of host time to enqueue" % (limit, ov, e0.elapsed_time(e1) * 1e3 / 50, th * 1e6 / 50))
dev.set_batch_limit(abi.MAX_BATCHES_PER_LAUNCH)
