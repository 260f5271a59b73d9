"""A/B probe of the exact ingest on the bench workload (36 M terrain, 36 resident ring batches): one terrain, several
environment settings, each timed like bench.py's step (reset + republish + kernel_construct launches until drained), plus
k_expand's phase timers of workgroup 0 (construct.hip Ctl.expandNs, byte 152).

    python tools/probe.py [--steps 5] [--points 36000000] "SIMLOD_EXPAND_WGS=64" "SIMLOD_EXPAND_WGS=128 SIMLOD_GRID_MULT=4" ...

Every positional argument is one variant: space-separated NAME=VALUE pairs put into the environment for that variant only
(a context reads its tuning knobs from the environment once; DeviceOctree.reload_env() reads them again).  The empty string "" is the default configuration."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from simlod_amd import abi, camera, synthetic
from simlod_amd.runtime import DeviceOctree

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--points", type=int, default=36_000_000)
ap.add_argument("--order", default="shuffled")
ap.add_argument("--coalesce", action="store_true")
ap.add_argument("--momentary-mb", type=int, default=None)
ap.add_argument("variants", nargs="*", default=[""])
args = ap.parse_args()

W, H = 1920, 1080
gen = synthetic.terrain if args.order == "shuffled" else synthetic.terrain_scan
pts, box = gen(args.points, seed=7)
batch = abi.MAX_BATCH_SIZE
nb = (args.points + batch - 1) // batch
T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=W / H))
dev = DeviceOctree("cuda:0", persistent_bytes=64 << 30, momentary_bytes=(args.momentary_mb or (700 if args.coalesce else 300)) * 1_000_000, max_pixels=W * H, coalesce=args.coalesce)
u = dev.uniforms(W, H, T, box, hqs=True)
rv = dev.ring.view(torch.uint8)
for i in range(nb):
    c = pts[i * batch:(i + 1) * batch]
    rv[i * batch * 16: i * batch * 16 + len(c) * 16].copy_(torch.from_numpy(c.view(np.uint8).reshape(-1)))
sizes = torch.tensor([min(batch, args.points - i * batch) for i in range(nb)], dtype=torch.int32, device=dev.device)


def step():
    dev.reset(u)
    dev.batch_sizes[:nb] = sizes
    dev.publish(nb)
    dev.uploaded_host = nb
    return dev.drain(u)


for var in args.variants:
    saved = {}
    for kv in var.split():
        k, v = kv.split("=", 1)
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    dev.reload_env()
    step()
    torch.cuda.synchronize()
    dev.momentary[152:216].zero_()
    dev.momentary[696:696 + 48 * 8].zero_()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    st = dev.read_stats()
    c = dev.momentary[152:216].cpu().numpy().view(np.uint64).astype(np.float64)
    calls, rounds = max(c[6], 1), max(c[5], 1)
    print(f"{var or 'default':60s} {ms:7.3f} ms/ingest  {args.points / ms / 1e3:7.0f} M pts/s  dbg={int(st['dbg'])} nodes={int(st['numNodes'])} | "
          f"k_expand wg0 per call us: H {c[0] / 1e3 / calls:5.1f} bar {c[1] / 1e3 / calls:5.1f} D {c[2] / 1e3 / calls:5.1f} bar2 {c[3] / 1e3 / calls:5.1f} "
          f"rounds/call {rounds / calls:4.2f} calls {int(c[6])}", flush=True)
    vt = dev.momentary[216:696].cpu().numpy().view(np.uint64).reshape(20, 3).astype(np.float64)
    vt = vt[vt[:, 2] > 0]
    if len(vt):
        print(f"    k_voxelize of the last launch, us from the first workgroup in: last piece done {np.mean(vt[:, 1] - vt[:, 0]) / 1e3:5.1f} (max {np.max(vt[:, 1] - vt[:, 0]) / 1e3:5.1f}), "
              f"last wave out {np.mean(vt[:, 2] - vt[:, 0]) / 1e3:5.1f} (max {np.max(vt[:, 2] - vt[:, 0]) / 1e3:5.1f}); batches {len(vt)}")
    ph = dev.momentary[696:696 + 48 * 8].cpu().numpy().view(np.uint64).astype(np.float64)
    f = lambda lo, n, cnt: " ".join(f"{ph[lo + i] / 1e3 / max(ph[cnt], 1):5.1f}" for i in range(n))
    print(f"    us per call, one workgroup: k_count [main loop, flush, queue_split] {f(0, 3, 3)} | k_hist [loop, flush] {f(4, 2, 6)} | k_insert wg0 [alloc, clear, count, reserve, wait, store] {f(8, 6, 14)}"
          f" | k_insert last wg {f(16, 6, 22)} | k_voxelize wg0 per piece [item+path+chunks, cubes+samples, level 1, levels 2+, write-back, reserve+chunks, store] {f(24, 7, 31)}", flush=True)
    if ph[39] > 0:
        print(f"    k_expand wg0 per slot us [hist+decide+reserve, acct timing, grids+next slots, nodes+paths+map, top table, fresh leaves' chunks] {f(32, 6, 39)}; fresh leaves per slot {ph[38] / ph[39]:5.1f}; slots (wg0) {int(ph[39])}", flush=True)
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
