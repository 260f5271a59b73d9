"""The code path ranks 0 and 1 of a multi-GPU job execute, run as TWO PROCESSES ON ONE GPU (gloo: RCCL refuses two ranks on one device):
every rank generates its tile on the device, the level-3 cells of the global cube are dealt by point count, the records are routed with
one all-to-all, each rank streams what it owns through its 50-slot ring into its own octree and the ranks compose exact frames
(simlod_amd/distributed.py, runtime.DeviceOctree.stream).  The parent process checks every rank against the CPU oracle: ownership,
the rank's octree (built from the same records in the same batches), and the composed frames (the oracle's composition of the two
octree images with the same reductions)."""
import ctypes
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

W = H = 384
N_PER_RANK = 3_000_000
TILES_X, TILE = 2, (600.0, 400.0, 40.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _uniforms(dev_or_abi, box, hqs, persistent):
    from simlod_amd import abi, camera
    T = camera.lookat_transform((1.1 * box[0], -0.9 * box[1], 1.2 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    return abi.make_uniforms(W, H, T, box, persistent_capacity=persistent, momentary_capacity=300_000_000, hqs=hqs)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from simlod_amd import distributed
    from simlod_amd.runtime import DeviceOctree
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    box = np.array([TILES_X * TILE[0], TILE[1], TILE[2]], dtype=np.float32)
    persistent = 1 << 30
    dev = DeviceOctree("cuda:0", persistent_bytes=persistent, max_pixels=W * H)
    dev.momentary.fill_(0xA5); dev.persistent.fill_(0xA5); dev.render_buffer.fill_(0xA5)
    generated = torch.empty(N_PER_RANK * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(generated, rank * N_PER_RANK, N_PER_RANK, 7, TILES_X, TILE)          # rank r makes tile r of the global stream
    codes = distributed.cell_codes(generated, box, 3)
    owner, counts = distributed.balanced_owners(codes, world, 3)
    mine, recv = distributed.route_points(generated, codes, owner)
    assert int(sum(recv)) == mine.shape[0] and int(counts.sum()) == world * N_PER_RANK
    # the same in slices (what bench.py --gpus N does with 500 M points per rank): the same records in the same order
    mine2, owner2, counts2, recv2 = distributed.partition_and_route(generated, box, world, level=3, slice_points=700_000)
    assert torch.equal(mine, mine2) and torch.equal(owner, owner2) and recv2 == recv
    del mine2
    u = _uniforms(None, box, True, persistent)
    dev.reset(u)
    dev.set_trunk_mask(*distributed.trunk_mask(counts))          # the shared upper levels split by the GLOBAL counts: the single-GPU octree's topology
    launches = dev.stream(u, mine.reshape(-1), int(mine.shape[0]))
    st = dev.read_stats()
    assert int(st["dbg"]) == 0 and int(st["numPoints"]) == mine.shape[0] and launches >= 1
    frames = {}
    for name, hqs in (("hqs", True), ("plain", False)):
        uf = _uniforms(None, box, hqs, persistent)
        vis, cnt = distributed.render_frame(dev, uf)
        torch.cuda.synchronize()
        frames[name] = dev.framebuffer(W, H)
        frames[name + "_color"] = dev.color(W, H)
        frames[name + "_visible"] = cnt.cpu().numpy()
    # ... and with two frames in flight (the plane reductions of one frame beside the rasterisation of the next): the same frames
    seq = [_uniforms(None, box, h, persistent) for h in (True, False, True, True, False)]
    got = {}
    def keep(i, r, recs, cnts):
        torch.cuda.synchronize()
        got[i] = r.framebuffer(W, H)
    distributed.render_frames_pipelined(dev, seq, on_frame=keep)
    dev.select_frame(0)
    assert sorted(got) == [0, 1, 2, 3, 4]
    for i, h in enumerate((True, False, True, True, False)):
        assert np.array_equal(got[i], frames["hqs" if h else "plain"]), f"frame {i} differs when two frames are in flight"
    nodes, pers, n, nodes_base, pers_base = dev.download_image()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), mine=mine.cpu().numpy(), owner=owner.cpu().numpy(), counts=counts, nodes=nodes.view(np.uint8), pers=pers, n=n,
             nodes_base=nodes_base, pers_base=pers_base, stats=np.frombuffer(dev.stats.cpu().numpy().tobytes(), dtype=np.uint8), **frames)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_partition_ingest_and_compose_frames_like_the_oracle(built_libs, tmp_path):
    import torch.multiprocessing as mp
    import oracle
    from simlod_amd import abi, distributed
    from util import STATS_BUILD_FIELDS, assert_dumps_equal, assert_stats_equal
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) for r in range(world)]
    box = np.array([TILES_X * TILE[0], TILE[1], TILE[2]], dtype=np.float32)
    assert np.array_equal(ranks[0]["owner"], ranks[1]["owner"]) and np.array_equal(ranks[0]["counts"], ranks[1]["counts"])
    owner = ranks[0]["owner"]
    assert sum(len(r["mine"]) for r in ranks) == world * N_PER_RANK
    images, lib_o = [], oracle.port_lib()
    for r, d in enumerate(ranks):
        pts = np.ascontiguousarray(d["mine"]).reshape(-1).view(abi.point_dtype)
        # every record a rank received lies in a cell it owns — by the BUILDER's arithmetic (the level-3 node the builder files it under)
        size = np.float32(max(box))
        q = [((np.float32(2 ** 20) * pts[k]) / size).astype(np.uint32) & np.uint32(2 ** 20 - 1) for k in "xyz"]      # simlod_device.hpp quantize + child_index
        code = np.zeros(len(pts), dtype=np.int64)
        for lv in range(3):
            sft = np.uint32(19 - lv)
            code = (code << 3) | (((q[0] >> sft) & 1).astype(np.int64) << 2) | (((q[1] >> sft) & 1).astype(np.int64) << 1) | ((q[2] >> sft) & 1).astype(np.int64)
        assert (owner[code] == r).all(), f"rank {r} holds records of cells it does not own"
        load = np.array([int(d["counts"][owner == k].sum()) for k in range(world)])
        assert len(pts) == load[r] and load.max() <= 1.5 * load.mean()
        # the rank's octree == the restatement's, fed the same records in the same 1 M batches
        u = _uniforms(None, box, True, 1 << 30)
        ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
        ref.reset(u)
        ref.set_trunk_mask(*distributed.trunk_mask(d["counts"]))
        ref.add_points(u, pts)
        nodes = np.ascontiguousarray(d["nodes"]).view(abi.node_dtype).copy()
        pers, n = d["pers"].copy(), int(d["n"])
        oracle.rebase_image(nodes, n, pers, int(d["nodes_base"]), int(d["pers_base"]))
        stats = np.frombuffer(d["stats"].tobytes(), dtype=abi.stats_dtype)[0]
        assert_stats_equal(stats, ref.stats[0], STATS_BUILD_FIELDS, f"rank {r}")
        assert_dumps_equal(oracle.dump_image(nodes, n), ref.dump(), f"rank {r}")
        images.append((nodes, pers, n))
    # frames: the oracle composes the SAME two octree images with the reductions of distributed.render_frame
    p = lambda arr: ctypes.c_void_p(arr.ctypes.data)
    for name, hqs in (("hqs", True), ("plain", False)):
        uu = np.ascontiguousarray(_uniforms(None, box, hqs, 1 << 30)).reshape(1)
        state = []
        for nodes, pers, n in images:
            st = np.zeros(1, dtype=abi.stats_dtype); st["numNodes"] = n
            state.append(dict(nodes=nodes, stats=st, fb=np.zeros(W * H, dtype=np.uint64), color=np.zeros(W * H, dtype=np.uint32), vis=np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype),
                              depth=np.zeros(W * H, dtype=np.uint32), sums=np.zeros(W * H * 4, dtype=np.uint32)))

        def part(k):
            for s_ in state:
                lib_o.oracle_render_part(None, p(uu), p(s_["nodes"]), p(s_["stats"]), p(s_["fb"]), p(s_["color"]), p(s_["vis"]), 1, k, p(s_["depth"]), p(s_["sums"]))
        part(0)
        if hqs:
            dm = np.minimum(state[0]["depth"], state[1]["depth"])
            for s_ in state: s_["depth"][:] = dm
            part(1)
            sm = state[0]["sums"] + state[1]["sums"]
            for s_ in state: s_["sums"][:] = sm
            part(2)
        else:
            fm = np.minimum(state[0]["fb"], state[1]["fb"])
            for s_ in state: s_["fb"][:] = fm
        part(3)
        for r, (d, s_) in enumerate(zip(ranks, state)):
            bad = int((d[name] != s_["fb"]).sum())
            assert bad == 0, f"{name}: rank {r}: {bad} pixels differ from the frame the oracle composes from the same two octrees"
            assert int((d[name] != abi.CLEAR_PIXEL).sum()) > 2000
            assert np.abs(d[name + "_color"].view(np.uint8).astype(np.int16) - s_["color"].view(np.uint8).astype(np.int16)).max() <= 1      # EDL: +-1 per channel
            # the all-gathered visible-node counts: what every rank's own visibility pass found
            assert [int(v) for v in d[name + "_visible"]] == [int(t["stats"]["numVisibleNodes"][0]) for t in state]
        assert np.array_equal(ranks[0][name], ranks[1][name]), f"{name}: the ranks hold different composed frames"
    # ... and the composed frames show what ONE GPU holding every record shows: one octree of all 6 M records, in this
    # process, on the same device — the same depth at every pixel (which point colours a voxel is scheduling dependent)
    from simlod_amd.runtime import DeviceOctree
    single = DeviceOctree("cuda:0", persistent_bytes=1 << 30, max_pixels=W * H)
    u = _uniforms(None, box, True, 1 << 30)
    single.reset(u)
    for d in ranks:
        single.add_points(u, np.ascontiguousarray(d["mine"]).reshape(-1).view(abi.point_dtype))
    assert int(single.read_stats()["numPoints"]) == world * N_PER_RANK
    for name, hqs in (("hqs", True), ("plain", False)):
        single.render(_uniforms(None, box, hqs, 1 << 30))
        want = single.framebuffer(W, H)
        bad = int(((ranks[0][name] >> np.uint64(32)) != (want >> np.uint64(32))).sum())
        assert bad == 0, f"{name}: {bad} pixels of the two ranks' composed frame have another depth than the single-GPU frame"
    single.close()


def test_bench_started_bare_with_gpus_2_relaunches_itself_through_torch_distributed_run(built_libs):
    """`python bench.py --gpus 2 ...` without torch.distributed.run (no WORLD_SIZE in the environment): bench.py starts itself again through
    `python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1` and still prints ONE JSON line (a bare start used
    to die on an assert)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "2", "--points", "3000000", "--backend", "gloo", "--one-device",
           "--no-cpu-baseline", "--no-profile", "--persistent-gb", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["collective"]["ranks_seen"] == 2


def test_bench_n2_runs_its_multi_rank_path_with_two_processes_on_one_gpu_over_gloo(built_libs):
    """`bench.py --gpus 2` — generation per rank, partition in slices, streamed ingest of what each rank owns, composed frames with two in
    flight, max-over-ranks timing, ONE JSON line from rank 0 — launched the way the driver launches it (torch.distributed.run, one process
    per rank), but with --backend gloo --one-device: RCCL refuses two ranks on one device, and this box has one.  Everything above the
    process group is the code the 8-GPU run executes."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "3", "--points", "6000000", "--backend", "gloo", "--one-device",
           "--no-cpu-baseline", "--no-profile", "--persistent-gb", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["points_per_gpu"] == 6_000_000
    assert sum(out["partition"]["per_rank_points"]) == 12_000_000 and out["partition"]["max_over_mean"] <= 1.5
    assert set(out["raster"]) == {"hqs", "plain", "hqs_close", "plain_close"} and all(v["visible_samples"] > 0 for v in out["raster"].values())
