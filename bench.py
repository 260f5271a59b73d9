#!/usr/bin/env python3
"""bench.py — SimLOD hot paths on MI355X: octree ingest (kernel_construct) + software raster (kernel_render).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.

Workload at N=1 = BASELINE.json configs[1]: "Morro Bay 36M (.simlod, 16 B/point) single-batch ingest + 1080p raster".
The real file is not available offline, so the input is the synthetic stand-in of simlod_amd/synthetic.terrain():
36 M XYZRGBA points over a 6 km x 4 km x 0.4 km fractal terrain, emitted swath by swath, split in 36 ring batches of
1 M points that are resident in HBM before the timed region starts.

One STEP = the reference's whole ingest of that file through its launch surface: `kernel` (reset) + republishing the
36 ring slots + two `kernel_construct` launches (20 + 16 batches, main_progressive_octree.cpp:364-428 /
progressive_octree_voxels.cu:883).  value = points inserted per second over exactly K such steps (max over ranks).
The raster half of the metric is measured right after, on the octree of the last step: K frames of `kernel_render` at
1920x1080 (HQS, the reference's default, and plain) and reported under "raster".

N>1 (weak scaling, BASELINE config 4's shape): ONE global cube over a terrain of N tiles of 36 M points each.  Every rank generates its
tile ON THE DEVICE (simlod_generate_terrain), the ranks histogram the points over the 512 level-3 cells of the global cube, the cells
are dealt to ranks by point count (distributed.balanced_owners), and the records are routed with one all-to-all over RCCL
(distributed.route_points) — all of that BEFORE the timed region: config 4 is "spatially pre-partitioned", the timed step is the
ingest of resident points into the rank's sub-octrees (no data-path collective), value = all points / max over ranks.  The routing
time is reported under "partition".  A frame is composed exactly (distributed.render_frame: MIN/SUM all-reduces between the passes)
plus an all-gather of the visible-node records, over RCCL.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s peak (~6.3 TB/s achievable)
W, H = 1920, 1080


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=None, help="points per GPU (default: 36 M resident in the ring at N=1 = BASELINE config 2; 500 M streamed through the ring at N>1 = config 4)")
    ap.add_argument("--raster-presets", default="bird,close", help="camera presets of the raster half (tools/profile.sh collects counters for the bird preset alone)")
    ap.add_argument("--stream", action="store_true", help="N=1 too: device-generated points streamed through the 50-slot ring with the reference's back-pressure (config 4's per-rank path)")
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--cpu-points", type=int, default=36_000_000, help="bounded sample for the CPU baseline (the port finishes all 36 M in a few seconds)")
    ap.add_argument("--b0-points", type=int, default=36_000_000, help="how much of the same terrain B0 — the reference's own sources as host code — is given (it stops after 25 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("SIMLOD_BENCH_BACKEND", "nccl"), help="torch.distributed backend of the N>1 path: nccl (= RCCL over xGMI) or gloo (dry runs: "
                                                                                          "tests/test_gpu_distributed.py drives the same code path with two processes on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (dry runs of the N>1 path on a one-GPU box)")
    ap.add_argument("--profiles", default=None, help="directory under profiles/ whose kept measurements (PMC traffic, rocprofv3 kernel table, config 3) may be quoted; default: the newest "
                                                       "profiles/r*/ — and only while its fingerprint.json names the kernel sources of THIS tree (simlod_amd/fingerprint.py)")
    ap.add_argument("--coalesce", action="store_true", help="opt-in coalesced ingest (simlod_set_ingest_mode(1)): all pending batches of a launch as one")
    ap.add_argument("--momentary-mb", type=int, default=None, help="size of kernel_construct's momentary buffer (the reference host gives 300 MB)")
    ap.add_argument("--persistent-gb", type=int, default=64, help="size of the persistent buffer (the reference host takes 80 %% of the free device memory)")
    ap.add_argument("--order", choices=["shuffled", "scan"], default="shuffled",
                    help="record order of the synthetic terrain: shuffled inside 250 m tiles (default, the harder case) or scan-line order as in a LAS file")
    a = ap.parse_args()
    if a.points is None:
        a.points = 36_000_000 if a.gpus == 1 and not a.stream else 500_000_000
    if a.momentary_mb is None:
        a.momentary_mb = 700 if a.coalesce else 300      # coalesced groups of 20 batches want room for 20 M waiting samples + moved points
    return a


def launches_idle(launches, n_batches, coalesced):
    """kernel_construct enqueues one kernel group per possible batch (20 per launch); the groups beyond the pending batches exit at once."""
    if coalesced:
        return 0 if launches <= 4 else launches - 2
    return max(0, launches - n_batches)


def collect_profile(L):
    from simlod_amd.runtime import lib  # noqa: F401

    class Entry(ctypes.Structure):
        _fields_ = [("name", ctypes.c_char * 48), ("launches", ctypes.c_uint32), ("pad", ctypes.c_uint32), ("total_ms", ctypes.c_double)]
    buf = (Entry * 64)()
    cnt = ctypes.c_int(0)
    L.simlod_profile_collect(ctypes.byref(buf), 64, ctypes.byref(cnt))
    return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(cnt.value)}


def relaunch_through_torchrun(args):
    """`python bench.py --gpus N` started bare (no WORLD_SIZE in the environment) with N > 1: start it again the way the driver does — one process per
    GPU through torch.distributed.run on 127.0.0.1 — instead of giving up: the first hardware SCALE run must not die on a launch detail."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_through_torchrun(args)
    import torch
    import torch.distributed as dist
    from simlod_amd import abi, camera, synthetic
    from simlod_amd.runtime import DeviceOctree, lib

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    use_dist = "WORLD_SIZE" in os.environ            # launched through torch.distributed.run (also with one rank)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on stdout when its communicator comes up; stdout must carry ONE JSON line only
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(args.backend)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)       # the C library's buffer too (the print came through printf), while fd 1 still points away
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run"

    def barrier():
        if use_dist:
            dist.barrier()

    n_points = args.points
    batch = abi.MAX_BATCH_SIZE
    partition = None
    source = None                          # N>1 / --stream: the rank's points, resident on the device, streamed through the ring every step
    # the persistent buffer: the reference host takes 80 % of the free device memory (main_progressive_octree.cpp:579-586: ~230 GB of an MI355X's 288); a quarter of
    # that here.  (Its size matters to the ingest: exact mode takes a launch's batches in groups only while the allocator is a worst-case group — every sample
    # colouring a voxel on every level: ~6 GB for four batches — away from the reference's memory guard, voxels.cu:896-912; with 8 GB the groups end mid-ingest.)
    persistent_bytes = max(args.persistent_gb << 30, 48 * n_points)
    if not use_dist and not args.stream:
        n_batches = (n_points + batch - 1) // batch
        assert n_batches <= abi.BATCH_STREAM_SIZE, "the resident workload must fit the 50-slot ring (use --stream for more)"
        gen = synthetic.terrain if args.order == "shuffled" else synthetic.terrain_scan
        pts, box = gen(n_points, seed=7)
        dev = DeviceOctree(f"cuda:{local}", persistent_bytes=persistent_bytes, momentary_bytes=args.momentary_mb * 1_000_000, max_pixels=W * H, coalesce=args.coalesce)
        L = lib()
        ring_view = dev.ring.view(torch.uint8)
        for i in range(n_batches):           # H2D once, outside every timed region: inputs are resident in HBM
            chunk = pts[i * batch:(i + 1) * batch]
            ring_view[i * batch * 16: i * batch * 16 + len(chunk) * 16].copy_(torch.from_numpy(chunk.view(np.uint8).reshape(-1)))
        my_points = n_points
    else:
        from simlod_amd import distributed
        tiles_x = int(np.ceil(np.sqrt(world)))
        tiles_y = (world + tiles_x - 1) // tiles_x
        tile_extent = (6000.0, 4000.0, 400.0)
        box = np.array([tiles_x * tile_extent[0], tiles_y * tile_extent[1], tile_extent[2]], dtype=np.float32)
        dev = DeviceOctree(f"cuda:{local}", persistent_bytes=persistent_bytes, momentary_bytes=args.momentary_mb * 1_000_000, max_pixels=W * H, coalesce=args.coalesce)
        L = lib()
        generated = torch.empty(n_points * 16, dtype=torch.uint8, device=dev.device)
        dev.generate_terrain(generated, rank * n_points, n_points, 7, tiles_x, tile_extent)       # rank r makes tile r of the global stream
        torch.cuda.synchronize(); barrier()
        if use_dist:
            t0 = time.perf_counter()
            # histogram, assignment and routing in slices of 32 M records: on top of the generated points and the routed result a rank holds two
            # slices of records and one slice's index arrays (~2.8 GB), whatever --points is (distributed.partition_and_route)
            mine, owner, counts, recv = distributed.partition_and_route(generated, box, world, level=3, slice_points=32_000_000)
            torch.cuda.synchronize(); barrier()
            t_part = time.perf_counter() - t0
            load = np.array([int(counts[owner.cpu().numpy() == r].sum()) for r in range(world)])
            partition = {"level": 3, "cells_occupied": int((counts > 0).sum()), "per_rank_points": load.tolist(), "max_over_mean": float(load.max() / load.mean()),
                         "histogram_assign_route_ms": t_part * 1e3, "kept_local": int(recv[rank]), "slice_points": 32_000_000,
                         "what": "all-reduce of 512-cell histograms, greedy by count, the records routed in slices (one all_to_all_single per slice of 32 M)"}
            del generated
            # the shared upper levels (0-2) split by the GLOBAL counts on every rank: composed frames equal the single-GPU frame (distributed.trunk_mask)
            trunk = distributed.trunk_mask(counts)
            dev.set_trunk_mask(*trunk)
            partition["trunk_mask"] = f"{trunk[0] | (trunk[1] << 64):#x}"
        else:
            mine = generated.reshape(-1, 16)
        my_points = int(mine.shape[0])
        n_batches = (my_points + batch - 1) // batch
        source = mine.reshape(-1)
        pts = None
    sizes = torch.tensor([min(batch, my_points - i * batch) for i in range(n_batches)], dtype=torch.int32, device=dev.device)
    # "Morro Bay - bird" and "Morro Bay - close" (main_progressive_octree.cpp:1314-1328), scaled to the stand-in's box: the bird looks at the
    # middle of the terrain from 3.9 km, the close one at the surface point under the reference's target from 94 m
    T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * float(box[0]) / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])),
                               camera.perspective(aspect=W / H))
    cx, cy = 2750.218 * float(box[0]) / 6000.0, 974.775 * float(box[1]) / 4000.0
    T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, synthetic.terrain_height(cx, cy, seed=7, box=tuple(float(v) for v in box)) if not use_dist and source is None else 0.35 * float(box[2]))),
                                     camera.perspective(aspect=W / H))
    u = dev.uniforms(W, H, T, box, hqs=True)

    def ingest_step():
        dev.reset(u)
        if source is not None:                        # config 4: through the 50-slot ring, uploader + back-pressure + one launch per frame
            return dev.stream(u, source, my_points)
        dev.batch_sizes[:n_batches] = sizes           # what the uploader's cuMemsetD32Async pair publishes
        dev.publish(n_batches)                        # (... and what shim/cuda.h tells the library about it: simlod_upload_counter_written)
        dev.uploaded_host = n_batches
        return dev.drain(u)      # ceil(36 / 20) launches back to back, one look at Stats.batchletIndex; more launches if a time budget cut one short

    for _ in range(args.warmup):
        ingest_step()
    torch.cuda.synchronize(); barrier()
    t0 = time.perf_counter()
    launches = 0
    for _ in range(args.steps):
        launches += ingest_step()
    torch.cuda.synchronize(); barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev.device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ms_per_step = float(tmax.item()) * 1e3 / args.steps
    stats = dev.read_stats()
    assert int(stats["numPointsProcessed"]) == my_points and int(stats["numPoints"]) == my_points, "ingest lost points"
    assert int(stats["dbg"]) == 0, f"device error bits {int(stats['dbg']):#x}"
    value = world * n_points / (ms_per_step * 1e-3) / 1e6
    # The headline is what the UNCHANGED reference host gets: the library sizes its launches by the upload-counter writes it is told about (shim/cuda.h
    # forwards the uploader's cuMemsetD32Async, main_progressive_octree.cpp:1047-1050; the Python mirror's publish() does the same) and by what its
    # earlier launches reported.  Beside it: the same steps with the host ALSO saying how many batches are pending in front of every launch
    # (simlod_context_hint_pending_batches) — the two must agree — and with a host that tells nothing at all (the library's prediction alone).
    without_hint = None
    if source is None and rank == 0 and not use_dist and not args.no_cpu_baseline:      # (not in the profiling passes: their per-launch figures count three ingests)
        def timed():
            ingest_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ingest_step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3 / args.steps
            return {"ms_per_step": ms, "value": n_points / (ms * 1e-3) / 1e6}
        was = dev.hint_pending, dev.notifies
        dev.hint_pending = True
        with_hint = timed()
        dev.hint_pending, dev.notifies = False, False
        blind = timed()
        dev.hint_pending, dev.notifies = was
        ingest_step(); torch.cuda.synchronize()
        without_hint = {"with_host_hint": with_hint, "host_tells_nothing": blind,
                        "what": "headline: launches sized by the upload-counter writes the shim forwards; with_host_hint: + simlod_context_hint_pending_batches per launch; host_tells_nothing: neither (prediction from the launches' own reports)"}
    collective = None
    if use_dist:
        # what the process group really was: every rank adds 1 (ranks_seen must be the world size) and its point count (the octrees of all
        # ranks together hold every generated point)
        seen = torch.tensor([1, int(stats["numPoints"])], dtype=torch.int64, device=dev.device)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        ranks_seen, total_points = int(seen[0].item()), int(seen[1].item())
        assert ranks_seen == world, f"the all-reduce saw {ranks_seen} ranks, WORLD_SIZE is {world}"
        assert total_points == world * n_points, f"the ranks' octrees hold {total_points} points, {world} x {n_points} were generated"
        ver = None
        if args.backend == "nccl":
            try:
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                ver = None
        collective = {"backend": dist.get_backend(), "is_rccl": args.backend == "nccl" and torch.version.hip is not None, "world_size": dist.get_world_size(),
                      "nccl_version": ver, "ranks_seen": ranks_seen, "points_in_all_octrees": total_points}

    # ---- raster ------------------------------------------------------------------------------------------------
    from simlod_amd import distributed

    raster = {}
    # Kept measurements (profiles/): quoted ONLY when they were taken on the kernel sources of this tree — the fingerprint the profiling tools
    # record must equal the one computed here; otherwise the fields stay null and `profiles_note` says why.
    from simlod_amd.fingerprint import kept_profiles
    pdir, tfile, profiles_note, sha_now = kept_profiles(args.profiles)
    rtraffic = json.load(open(tfile)) if tfile else {}
    presets = args.raster_presets.split(",")
    for name, hqs, Tcam in [m for m in (("hqs", 1, T), ("plain", 0, T), ("hqs_close", 1, T_close), ("plain_close", 0, T_close)) if ("close" if "close" in m[0] else "bird") in presets]:
        uc = dev.uniforms(W, H, Tcam, box, hqs=bool(hqs))

        def frame():
            """One frame: a single launch on one GPU; across ranks the exact composition of distributed.render_frame —
            HQS: all-reduce(MIN) of the depth plane and all-reduce(SUM) of the colour sums between the passes; plain: all-reduce(MIN)
            of the uint64 framebuffer; plus the all-gather of the visible-node records."""
            if use_dist:
                return distributed.render_frame(dev, uc, check_overflow=False)      # (no host synchronisation inside the frame loop)
            else:
                dev.render(uc)

        for _ in range(2):
            frame()
        torch.cuda.synchronize(); barrier()
        t0 = time.perf_counter()
        if use_dist and world > 1:
            # two frames in flight: the plane reductions of frame f travel while frame f + 1 is rasterised (distributed.render_frames_pipelined)
            distributed.render_frames_pipelined(dev, [uc] * args.frames, check_overflow=False)
        else:
            for _ in range(args.frames):
                frame()
        torch.cuda.synchronize(); barrier()
        dtf = time.perf_counter() - t0
        tm = torch.tensor([dtf], dtype=torch.float64, device=dev.device)
        st = dev.read_stats()
        samples = torch.tensor([float(int(st["numVisiblePoints"]) + int(st["numVisibleVoxels"]))], dtype=torch.float64, device=dev.device)
        if use_dist:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX); dist.all_reduce(samples, op=dist.ReduceOp.SUM)
        ms = float(tm.item()) * 1e3 / args.frames
        vs = float(samples.item())
        if use_dist and world > 1:
            # one more composed frame, its framebuffer folded into one word per rank: every rank must hold the same frame
            recs_c, cnts_c = distributed.render_frame(dev, uc, check_overflow=False)
            assert not distributed.visible_overflowed(cnts_c, 4096), "a rank has more visible nodes than the gather carried"
            fbw = dev.framebuffer_words()
            h = (fbw ^ (fbw >> 29)).sum().reshape(1) if not hqs else (dev.colorbuffer[: W * H].to(torch.int64) * 2654435761 % 1000003).sum().reshape(1)
            hs = [torch.zeros_like(h) for _ in range(world)]
            dist.all_gather(hs, h)
            assert all(int(x.item()) == int(hs[0].item()) for x in hs), f"{name}: the ranks hold different composed frames"
            collective.setdefault("frames_identical_on_all_ranks", []).append(name)
        # plain 24 B/sample (16 B read + 8 B framebuffer RMW) + 20 B/px (8 clear + 12 output); HQS 32 B/sample (two
        # reads) + 4 B depth RMW per sample + 48 B/px (20 clear + 28 resolve) — the 16 B colour RMW per ACCEPTED sample is left out
        # (the kernels do not count acceptances), so the HQS figure is a lower bound.  `frac` prices the frame by THAT definition; with
        # per-item LDS tiles the framebuffer RMW never reaches HBM, so `frac_by_traffic` prices the same frame by the bytes the PMC
        # counters saw the whole frame move (profiles/traffic_r0x.json, bird preset; None for the other preset / without the file).
        rb = (32.0 + 4.0) * vs + 48.0 * W * H if hqs else 24.0 * vs + 20.0 * W * H
        fkeys = (["r_visible", "r_draw<MODE_DEPTH>", "r_overflow<MODE_DEPTH>", "r_draw<MODE_COLOR>", "r_overflow<MODE_COLOR>", "r_output<true>"] if hqs
                 else ["r_visible", "r_draw<MODE_MIN64>", "r_overflow<MODE_MIN64>", "r_output<false>"])      # (r_overflow: frames that sort samples into the screen bins only)
        pre = "close/" if "close" in name else ""                      # (tools/fold_profiles.py: the close-up preset's passes are folded under "close/...")
        ftraffic = sum(rtraffic.get(pre + k, 0.0) for k in fkeys) if (rtraffic and (pre + "r_draw<MODE_MIN64>") in rtraffic) else None
        raster[name] = {"value": vs / (ms * 1e-3) / 1e6, "unit": "M samples/s @1920x1080", "ms_per_frame": ms,
                        "camera": "close" if "close" in name else "bird",          # "Morro Bay - close" / "- bird", main_progressive_octree.cpp:1314-1328
                        "visible_samples": int(vs), "visible_nodes": int(st["numVisibleNodes"]),
                        "roofline": {"bound": "hbm", "achieved": rb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "traffic": ftraffic, "frac_by_traffic": (ftraffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ftraffic else None}}

    # ---- per-kernel attribution (separate, untimed passes) ----------------------------------------------------------------
    # (1) the ROOFLINE kernel in the HEADLINE configuration: simlod_profile_enable(2) puts HIP events around k_voxelize — the builder's dominant
    #     kernel — on the stream it is launched on (the library's second stream), the two-stream pipeline stays as it is in the timed region;
    #     the average is over the launches that had a batch (a launch enqueues kernels for 20 batches; those without one exit in a few us).
    # (2) every kernel's time on ONE stream (simlod_profile_enable(1): HIP events between the launches need one timeline) — the `kernels` table.
    roofline, chain, kernels = None, None, {}
    # measurement aid in the control block at byte 0 of the momentary buffer: construct.hip Ctl.expandNs[7] (stored points moved by splits) at byte 208
    CTL_COUNTERS = slice(208, 216)
    CTL_GROUPS = slice(184, 192)           # Ctl.expandNs[4]: groups of batches ingested (every per-group kernel had that many launches with work)
    if rank == 0 and not args.no_profile:
        L.simlod_profile_enable(2)
        dev.momentary[CTL_GROUPS].zero_()
        ingest_step()
        torch.cuda.synchronize()
        prof_dom = collect_profile(L)
        groups_with_work = int(dev.momentary[CTL_GROUPS].cpu().numpy().view(np.uint64)[0])
        L.simlod_profile_enable(1)
        dev.momentary[CTL_COUNTERS].zero_()
        ingest_step()
        prof_c = collect_profile(L)
        counters = [int(v) for v in dev.momentary[CTL_COUNTERS].cpu().numpy().view(np.uint64)]
        moved = counters[0]
        dev.render(u)
        prof_r = collect_profile(L)
        L.simlod_profile_enable(0)
        st = dev.read_stats()
        new_voxels = int(st["numVoxels"])
        kernels = {k: round(ms / max(n, 1) * 1e3, 1) for k, (n, ms) in {**prof_c, **prof_r}.items()}
        kernels["_what"] = "avg us per launch, HIP events between launches on ONE stream (the side-stream overlap of the headline run is off in this pass; idle launches included)"
        chain_bytes = 32.0 * my_points + 16.0 * new_voxels                 # 32 B/point + 16 B/new voxel
        # PMC traffic of the kept profile (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE passes of this command), only when it was taken on THESE sources
        chain_keys = ("k_count", "k_queue", "k_hist", "k_expand", "k_insert", "k_voxelize")
        chain_traffic = sum(rtraffic.get(k, 0.0) for k in chain_keys) * n_batches if (rtraffic and "k_voxelize" in rtraffic) else None
        chain = {"bound": "hbm", "achieved": chain_bytes / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": chain_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": chain_bytes,
                 "traffic": chain_traffic, "wasted_traffic_ratio": (chain_traffic / chain_bytes) if chain_traffic else None,
                 "what": "32 B/point + 16 B/new voxel of the whole ingest over the headline ms_per_step"}
        # Algorithmic bytes per kernel for one whole ingest: k_voxelize reads every stored sample back (16 B; moved points too) and
        # stores a voxel per new cell (16 B); the cube words it loads and writes back are overhead, not algorithm.
        per_ingest = {"k_count": 16.0 * my_points, "k_hist": 32.0 * moved, "k_insert": 32.0 * (my_points + moved), "k_voxelize": 16.0 * (my_points + moved) + 16.0 * new_voxels}
        dom = "k_voxelize"
        n_dom, ms_dom = prof_dom.get(dom, (0, 0.0))
        # launches with work: the groups of batches the pass ingested (counted on the device: Ctl.expandNs[4]) — exact mode takes a launch's batches in
        # groups (construct.hip account_group), so a launch of the kernel covers several 1 M-point batches; what the events saw beyond that are early
        # exits of groups without a batch.  `avg_launch_us` is the RAW event time over the launches with work (the early exits' few us each are in it);
        # `avg_launch_us_less_idle` takes 4 us per early exit off (a model, reported beside the measurement, not instead of it).
        active = max(1, min(n_dom, groups_with_work))
        idle = max(0, n_dom - active)
        avg_ms = max(ms_dom, 1e-6) / active
        avg_ms_less_idle = max(ms_dom - idle * 0.004, 1e-6) / active
        bytes_per_launch = per_ingest[dom] / active
        traffic = rtraffic.get(dom) * n_batches / active if (tfile and dom in rtraffic) else None     # HBM bytes per launch from the PMC passes (kept profile, same sources; the file holds bytes per 1 M-point batch)
        roofline = {"bound": "hbm", "kernel": dom, "achieved": bytes_per_launch / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                    "wasted_traffic_ratio": (traffic / bytes_per_launch) if traffic else None,
                    "avg_launch_us": avg_ms * 1e3, "avg_launch_us_less_idle": avg_ms_less_idle * 1e3, "bytes_per_launch": bytes_per_launch, "launches_with_work": active, "launches_without": idle,
                    "batches_per_launch": n_batches / active,
                    "measured_with": "HIP events in the launch's own start/stop slots (hipExtLaunchKernelGGL) on the stream the kernel runs on; two-stream pipeline as in the timed region",
                    "per_kernel_wasted_traffic_ratio": {k: round(rtraffic[k] * n_batches / per_ingest[k], 2) for k in per_ingest if rtraffic.get(k) and per_ingest[k] > 0} or None,
                    "moved_points": moved, "new_voxels": new_voxels}

    # ---- loader row: LAS format-2 records (26 B) -> Points (16 B) on the device, one 1 M-point batch per launch ----
    loader = None
    if rank == 0 and not args.no_profile:
        from simlod_amd import lasio
        rs = np.random.RandomState(5)
        rec = lasio.las_records(rs.randint(0, 6_000_000, size=(batch, 3)).astype(np.int32), rs.randint(0, 65536, size=(batch, 3)).astype(np.uint16), 2)
        # a ROTATING set of 42 raw batches + 42 output slots = 1.76 GB per round: beyond the 256 MiB Infinity Cache, so the figure is an
        # HBM number, not a cache number (one batch decoded over and over would sit in the cache)
        NSET = 42
        d_raw = torch.from_numpy(rec.reshape(-1)).to(dev.device).repeat(NSET).reshape(NSET, -1)
        d_out = torch.empty((NSET, batch * 16), dtype=torch.uint8, device=dev.device)
        scale3, off3 = (ctypes.c_double * 3)(1e-3, 1e-3, 1e-3), (ctypes.c_double * 3)(0.0, 0.0, 0.0)
        call = lambda k: L.simlod_decode_las(ctypes.c_void_p(d_raw[k].data_ptr()), ctypes.c_uint64(batch), ctypes.c_uint32(26), ctypes.c_uint32(2), scale3, off3,
                                             ctypes.c_void_p(d_out[k].data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for k in range(NSET):
            call(k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)       # (the launches go to torch's current stream)
        e0.record()
        for rep in range(3):
            for k in range(NSET):
                call(k)
        e1.record(); torch.cuda.synchronize()
        msl = e0.elapsed_time(e1) / (3 * NSET)
        gbs = (26.0 + 16.0) * batch / (msl * 1e-3) / 1e9
        loader = {"kernel": "k_decode_las", "value": batch / (msl * 1e-3) / 1e6, "unit": "M points/s decoded (LAS format 2, 26 B records)",
                  "us_per_launch": msl * 1e3, "what": "126 back-to-back 1 M-point launches over 1.76 GB of rotating buffers, one event pair around all of them",
                  "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_point": 42.0}}
        del d_raw, d_out

    # ---- measured denominator next to the vendor peak: a large device-to-device copy on this box ---------
    if rank == 0 and roofline is not None:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev.device)
        dst = torch.empty_like(src)
        for _ in range(2):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        copy_ms = e0.elapsed_time(e1) / 10
        copy_gbs = 2.0 * src.numel() / (copy_ms * 1e-3) / 1e9          # 1 GiB device-to-device copy on this box, read + write bytes per second
        for r in (roofline, chain):
            r["measured_copy_GBs"] = copy_gbs
            r["frac_of_measured_copy"] = r["achieved"] / copy_gbs
        if loader is not None:
            loader["roofline"]["frac_of_measured_copy"] = loader["roofline"]["achieved"] / copy_gbs
        del src, dst

    # ---- CPU baseline: the oracle's serial C restatement on a bounded sample of the same workload ------------------
    cpu = None
    if rank == 0 and world == 1 and not use_dist and not args.no_cpu_baseline:   # (the torchrun path generates its points on the device)
        import subprocess
        import tempfile
        # the serial restatement rebuilt -march=native ON this box, into a scratch file (the library that ships stays portable)
        native = os.path.join(tempfile.gettempdir(), "libsimlod_oracle_native.so")
        try:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port", "MARCH=native", f"PORTLIB={native}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.environ["SIMLOD_ORACLE_LIB"] = native
            march = "native"
        except Exception:
            march = "x86-64-v2 (native rebuild failed)"
        import oracle
        m = min(args.cpu_points, n_points)
        host = oracle.HostOctree("port", persistent_bytes=2 << 30, ring_slots=max(1, min(abi.BATCH_STREAM_SIZE, (m + batch - 1) // batch)))
        uh = abi.make_uniforms(W, H, T, box, persistent_capacity=2 << 30, momentary_capacity=300_000_000, hqs=True)
        host.reset(uh)
        for i in range(0, m, batch):
            host.upload(pts[i:i + batch])
        t0 = time.perf_counter()
        while int(host.stats["batchletIndex"][0]) < int(host.num_uploaded[0]):
            host.construct(uh)
        tc = time.perf_counter() - t0
        t0 = time.perf_counter()
        host.render(uh)
        tr = time.perf_counter() - t0
        vs = int(host.stats["numVisiblePoints"][0]) + int(host.stats["numVisibleVoxels"][0])
        cpu = {"value": m / tc / 1e6, "unit": "M points/s inserted", "cores": 1, "kind": "port",
               "sample": f"first {m} points ({(m + batch - 1) // batch} batches) of the same terrain, oracle/simlod_oracle.c -O3 -march={march}, 1 thread",
               "raster_value": vs / tr / 1e6, "raster_unit": "M samples/s @1920x1080 (HQS)", "host_cores_available": os.cpu_count()}
        del host
        # B0: the reference's own sources compiled as host code (oracle/_ref, built where /root/reference exists and
        # shipped as binaries), timed on BASELINE config 1: 1 M uniform points, one batch, 512 x 512 frame.  Quadratic list walks
        #  make it a correctness reference, not a fast CPU implementation.
        sys.stdout.flush()
        saved_stdout = os.dup(1)                 # the reference's reset.cu prints ("resetting octree"): stdout must carry ONE JSON line only
        os.dup2(2, 1)
        try:
            if oracle.have_ref():
                p1, b1 = synthetic.uniform_cube(1_000_000, seed=1234)
                T1 = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), 512, 512)
                u1 = abi.make_uniforms(512, 512, T1, b1, persistent_capacity=1 << 30, momentary_capacity=oracle.REF_MOMENTARY_BYTES)
                res = {}
                for kind in ("ref", "port"):
                    h1 = oracle.HostOctree(kind, persistent_bytes=1 << 30, ring_slots=1)
                    h1.reset(u1); h1.upload(p1)
                    t0 = time.perf_counter(); h1.construct(u1); tc1 = time.perf_counter() - t0
                    t0 = time.perf_counter(); h1.render(u1); tr1 = time.perf_counter() - t0
                    v1 = int(h1.stats["numVisiblePoints"][0]) + int(h1.stats["numVisibleVoxels"][0])
                    res[kind] = {"insert_M_points_per_s": 1.0 / tc1, "raster_M_samples_per_s": v1 / tr1 / 1e6, "numNodes": int(h1.stats["numNodes"][0]), "numVoxels": int(h1.stats["numVoxels"][0])}
                    del h1
                cpu["config1_reference_B0"] = {"what": "BASELINE config 1 (1 M uniform points, one batch, 512x512 plain frame), 1 thread: oracle/_ref (the reference's sources as host code) vs the restatement",
                                               "kind": "reference", **{k: {kk: round(vv, 2) if isinstance(vv, float) else vv for kk, vv in v.items()} for k, v in res.items()}}
                # ... and on a prefix of THIS workload (its list walks are quadratic in a leaf's chunk count: the whole 36 M would take hours)
                mb0 = min(args.b0_points, n_points)
                if mb0 > 0:
                    hb = oracle.HostOctree("ref", persistent_bytes=4 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
                    ub = abi.make_uniforms(W, H, T, box, persistent_capacity=4 << 30, momentary_capacity=oracle.REF_MOMENTARY_BYTES, hqs=True)
                    hb.reset(ub)
                    tb, done = 0.0, 0
                    for i in range(0, mb0, batch):                               # batch by batch, as long as 25 s allow
                        hb.upload(pts[i:i + batch])
                        t0 = time.perf_counter()
                        hb.construct(ub)
                        tb += time.perf_counter() - t0
                        done = min(i + batch, mb0)
                        if tb > 25.0:
                            break
                    cpu["workload_reference_B0"] = {"kind": "reference", "value": done / tb / 1e6, "unit": "M points/s inserted", "cores": 1,
                                                    "sample": f"first {done} points of the same terrain through oracle/_ref, one batch per call" + (" (stopped after 25 s)" if done < mb0 else ""), "seconds": round(tb, 2)}
                    del hb
        except Exception as e:                     # the baseline must never take the bench down
            cpu["config1_reference_B0"] = {"error": repr(e)}
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)       # the C library's buffer too (the print came through printf), while fd 1 still points away
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    # ---- the opt-in coalesced ingest mode on the same resident input (reported beside the headline, never as `value`) ---------------------
    coalesced = None
    if rank == 0 and world == 1 and not args.coalesce and not args.no_profile:
        try:
            mb = max(args.momentary_mb, 700)
            dev2 = DeviceOctree(f"cuda:{local}", persistent_bytes=8 << 30, momentary_bytes=mb * 1_000_000, max_pixels=W * H, coalesce=True)
            dev2.ring.copy_(dev.ring)
            u2 = dev2.uniforms(W, H, T, box, hqs=True)

            def step2():
                dev2.reset(u2)
                dev2.batch_sizes[:n_batches] = sizes
                dev2.publish(n_batches)
                dev2.uploaded_host = n_batches
                return dev2.drain(u2)
            step2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            l2 = sum(step2() for _ in range(args.steps))
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) * 1e3 / args.steps
            st2 = dev2.read_stats()
            ok = int(st2["numPoints"]) == n_points and int(st2["dbg"]) == 0 and all(int(st2[k]) == int(stats[k]) for k in ("numNodes", "numInner", "numLeaves", "numVoxels"))
            coalesced = {"value": n_points / (ms2 * 1e-3) / 1e6, "unit": "M points/s", "ms_per_step": ms2, "momentary_mb": mb, "same_octree_content_counts_as_exact": bool(ok),
                         "what": "opt-in simlod_set_ingest_mode(1): pending batches in groups of 10; same octree content, other allocator accounting"}
            dev2.close()
            del dev2
        finally:
            pass                              # (the ingest mode belongs to dev2's context: nothing process-wide to restore)

    # ---- BASELINE config 3 as stated (350 M-point scan-ordered LAS 1.4 file through the reference's own host functions, device decode in the
    # upload stream): a 9 GB file does not belong in a run that has to finish within minutes, so the object quotes the kept measurement of
    # tools/config3.py (profiles/r03/) and says so
    config3 = None
    c3path = os.path.join(pdir, "config3_350m.json") if pdir else ""
    if rank == 0 and os.path.exists(c3path) and json.load(open(c3path)).get("_csrc_sha16") == sha_now:
        c3 = json.load(open(c3path))
        pick = lambda d: {k: d[k] for k in ("kernel_M_points_per_s", "wall_M_points_per_s", "launches", "update_kernel_ms", "wall_ms_incl_h2d") if k in d} if isinstance(d, dict) else d
        config3 = {"measured_by": f"tools/config3.py (kept under {os.path.relpath(pdir, ROOT)}/, same kernel sources): 350 M-point scan-ordered LAS file through the reference's own host functions",
                   "las_scan_page_locked": pick(c3.get("las_scan_pinned")), "las_scan_pageable": pick(c3.get("las_scan_pageable"))}
    if rank == 0:
        out = {
            "metric": "M points/sec inserted into octree (raster M samples/s @1080p under 'raster')",
            "value": value, "unit": "M points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+u32 (fp32 quantise/project, fp64 pixel coordinate, integer octree/atomics)", "data": "synthetic",
            "ingest_mode": "coalesced" if args.coalesce else "exact",
            "host_hint": "simlod_context_hint_pending_batches per launch" if dev.hint_pending else None,
            "ingest_other_hosts": without_hint,
            "config": {"workload": (f"BASELINE config 4 shape: tiled terrain, {n_points} device-generated 16 B points per GPU streamed through the 50-slot ring ({n_batches} x 1M batches), "
                                    if source is not None else f"Morro Bay 36M stand-in (BASELINE config 2): {n_points} 16 B points, fractal terrain, {n_batches} x 1M ring batches resident in HBM, ") +
                                   f"reset + {launches / max(args.steps, 1):.1f} kernel_construct launches per step; raster 1920x1080",
                       "points_per_gpu": n_points, "record_order": args.order if not use_dist else "device-generated tiles, swath order",
                       "parallelism": f"one global cube, level-3 cells dealt to {world} rank(s) by point count"},
            "csrc_sha16": sha_now, "profiles_note": profiles_note,
            "roofline": roofline, "roofline_chain": chain, "cpu_baseline": cpu, "raster": raster, "coalesced_ingest": coalesced, "loader": loader,
            "collective": collective, "partition": partition, "config3": config3, "kernels": kernels,
            "octree": {k: int(stats[k]) for k in ("numNodes", "numInner", "numLeaves", "numVoxels", "allocatedBytes_persistent")},
        }
        out = json.loads(json.dumps(out), parse_float=lambda x: float("%.5g" % float(x)))      # five significant digits: the driver keeps 8 KB of the line
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
