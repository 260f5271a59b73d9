"""world_size-2 gloo test of the multi-GPU layer (simlod_amd/distributed.py): spatial ownership, batch routing and the
MIN composition of per-rank framebuffers.  Each rank builds/rasterises its sub-octrees with the CPU oracle; the composed
frame must equal the single-process frame of the whole data set bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


FRAME_VARIANTS = [("plain", dict(useHighQualityShading=0)), ("hqs", dict(useHighQualityShading=1)),
                  ("hqs_boxes", dict(useHighQualityShading=1, showBoundingBox=1)), ("hqs_ps2", dict(useHighQualityShading=1, pointSize=2))]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from simlod_amd import abi, camera, distributed, synthetic
    W = H = 256
    pts, box = synthetic.uniform_cube(800_000, seed=21)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=2)
    o.reset(u)
    batch = 400_000
    for i in range(0, len(pts), batch):
        # every rank "reads" half of the mixed batch, then the batch is routed to the owners of the octants
        # (contiguous halves: concatenated in rank order they keep the file order, so the serial first-writer-wins voxel
        # colours are those of the single-process run)
        part = pts[i + rank * batch // world: i + (rank + 1) * batch // world]
        mine = distributed.exchange_points(part, distributed.owner_of(part, box, world))
        mine = mine.numpy().reshape(-1).view(abi.point_dtype)
        owners = distributed.owner_of(mine, box, world)
        assert (owners == rank).all()
        o.upload(mine); o.construct(u)
    fb, _ = o.render(u)
    t = torch.from_numpy(fb.view(np.int64).copy())
    distributed.compose_min(t)
    vis_bytes = torch.from_numpy(np.ascontiguousarray(o.visible).view(np.uint8).reshape(-1).copy())
    recs, cnts = distributed.gather_visible(vis_bytes, len(o.visible))
    # the count may also be a one-element tensor next to the records (what the GPU path passes: no host synchronisation)
    padded = torch.zeros(4096 * 152, dtype=torch.uint8)
    padded[: vis_bytes.numel()] = vis_bytes
    recs2, cnts2 = distributed.gather_visible(padded, torch.tensor([len(o.visible)], dtype=torch.int32))
    assert torch.equal(cnts, cnts2)
    for r in range(world):
        assert torch.equal(recs[r, : int(cnts[r])], recs2[r, : int(cnts2[r])])
    # ... and issued asynchronously (render_frame: behind part 0, beside the rest of the frame), collected later
    finish = distributed.gather_visible(padded, torch.tensor([len(o.visible)], dtype=torch.int32), async_op=True)
    recs3, cnts3 = finish()
    assert torch.equal(cnts, cnts3) and all(torch.equal(recs[r, : int(cnts[r])], recs3[r, : int(cnts3[r])]) for r in range(world))
    np.save(os.path.join(out_dir, f"fb_{rank}.npy"), t.numpy().view(np.uint64))
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([int(o.stats["numPoints"][0]), int(cnts.sum()), int(o.stats["numVisibleNodes"][0])]))
    # whole frames through render_frame: plain, HQS (depth MIN / colour SUM between the passes), HQS with bounding boxes
    for name, kw in FRAME_VARIANTS:
        uv = u.copy()
        for k, v in kw.items():
            uv[k] = v
        recs, cnts = distributed.render_frame(o, uv)
        np.save(os.path.join(out_dir, f"frame_{name}_{rank}.npy"), o._fb.copy())
        np.save(os.path.join(out_dir, f"color_{name}_{rank}.npy"), o._color.copy())
        assert int(cnts.sum()) == int(np.load(os.path.join(out_dir, f"meta_{rank}.npy"))[1])
    dist.destroy_process_group()


def test_two_rank_sharded_ingest_and_min_composition(built_libs, tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle
    from simlod_amd import abi, camera, synthetic
    W = H = 256
    pts, box = synthetic.uniform_cube(800_000, seed=21)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=2)
    o.reset(u)
    for i in range(0, len(pts), 400_000):
        o.upload(pts[i:i + 400_000]); o.construct(u)
    fb, _ = o.render(u)
    f0, f1 = np.load(tmp_path / "fb_0.npy"), np.load(tmp_path / "fb_1.npy")
    m0, m1 = np.load(tmp_path / "meta_0.npy"), np.load(tmp_path / "meta_1.npy")
    assert np.array_equal(f0, f1), "all-reduce must leave the same frame on every rank"
    assert m0[0] + m1[0] == len(pts), "every point is owned by exactly one rank"
    assert m0[1] == m1[1] == m0[2] + m1[2] == int(o.stats["numVisibleNodes"][0]), "merged visible-node list"
    diff = int((f0 != fb).sum())
    assert diff == 0, f"{diff} pixels of the composed frame differ from the single-process frame"
    assert int((fb != abi.CLEAR_PIXEL).sum()) > 5000
    # render_frame: every variant must be the single-process frame, bit for bit, on both ranks
    for name, kw in FRAME_VARIANTS:
        uv = u.copy()
        for k, v in kw.items():
            uv[k] = v
        want_fb, want_color = o.render(uv)
        for rank in range(world):
            got = np.load(tmp_path / f"frame_{name}_{rank}.npy")
            bad = int((got != want_fb).sum())
            assert bad == 0, f"{name}: {bad} pixels of rank {rank}'s composed frame differ from the single-process frame"
            assert np.array_equal(np.load(tmp_path / f"color_{name}_{rank}.npy"), want_color), name


def test_ownership_is_a_partition():
    sys.path.insert(0, ROOT)
    from simlod_amd import distributed, synthetic
    pts, box = synthetic.uniform_cube(100_000, seed=3)
    for world in (1, 2, 4, 8):
        own = distributed.owner_of(pts, box, world)
        assert own.min() >= 0 and own.max() < world
        if world == 8:      # level 1: octant k -> rank k, x is the most significant bit (progressive_octree_voxels.cu:179)
            exp = ((pts["x"] >= 0.5).astype(int) << 2) | ((pts["y"] >= 0.5).astype(int) << 1) | (pts["z"] >= 0.5).astype(int)
            assert np.array_equal(own, exp)


# ---- one global cube, cells dealt by point count, one all-to-all (BASELINE config 4's shape) ---------------------------------------------
def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from simlod_amd import abi, camera, distributed, synthetic
    W = H = 256
    pts, box = synthetic.terrain(1_200_000, seed=4, box=(3000.0, 2000.0, 200.0), tile=125.0)     # NOT uniform: a thin sheet in a cube
    share = len(pts) // world
    part = pts[rank * share:(rank + 1) * share if rank + 1 < world else len(pts)]                 # every rank reads a contiguous part of the stream
    rec = torch.from_numpy(np.ascontiguousarray(part).view(np.uint8).reshape(-1, 16))
    level = 3
    codes = distributed.cell_codes(rec, box, level)
    owner, counts = distributed.balanced_owners(codes, world, level)
    assert int(counts.sum()) == len(pts)
    load = np.array([int(counts[owner.numpy() == r].sum()) for r in range(world)])
    assert load.max() <= 1.5 * load.mean(), f"per-rank load {load.tolist()} exceeds 1.5 x the mean"
    mine, recv = distributed.route_points(rec, codes, owner)
    assert mine.shape[0] == load[rank] == sum(recv)
    assert bool((owner[distributed.cell_codes(mine, box, level)] == rank).all()), "a rank received a record it does not own"
    mine_np = mine.numpy().reshape(-1).view(abi.point_dtype)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=4)
    o.reset(u)
    o.add_points(u, mine_np, 300_000)
    assert int(o.stats["numPoints"][0]) == len(mine_np)
    for name, kw in (("plain", dict(useHighQualityShading=0)), ("hqs", dict(useHighQualityShading=1))):
        uv = u.copy()
        for k, v in kw.items():
            uv[k] = v
        distributed.render_frame(o, uv)
        np.save(os.path.join(out_dir, f"frame4_{name}_{rank}.npy"), o._fb.copy())
    np.save(os.path.join(out_dir, f"load4_{rank}.npy"), load)
    dist.destroy_process_group()


def test_four_ranks_one_global_cube_balanced_cells_all_to_all(built_libs, tmp_path):
    """Four ranks, a terrain (thin, uneven: half of the cube's cells are empty) in ONE global cube: level-3 cells dealt by point count —
    no rank carries more than 1.5 x the mean — records routed with one all-to-all, frames composed exactly: every rank ends with the
    same frame, and it covers what the single-process frame covers."""
    world = 4
    port = _free_port()
    mp.spawn(_worker4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    import oracle
    from simlod_amd import abi, camera, synthetic
    W = H = 256
    pts, box = synthetic.terrain(1_200_000, seed=4, box=(3000.0, 2000.0, 200.0), tile=125.0)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=4)
    o.reset(u)
    o.add_points(u, pts, 300_000)
    loads = [np.load(tmp_path / f"load4_{r}.npy") for r in range(world)]
    assert all(np.array_equal(loads[0], l) for l in loads) and int(loads[0].sum()) == len(pts)
    for name, hqs in (("plain", 0), ("hqs", 1)):
        uv = u.copy(); uv["useHighQualityShading"] = hqs
        want, _ = o.render(uv)
        frames = [np.load(tmp_path / f"frame4_{name}_{r}.npy") for r in range(world)]
        for r in range(1, world):
            assert np.array_equal(frames[0], frames[r]), f"{name}: rank {r} holds another frame than rank 0"
        covered, want_covered = frames[0] != abi.CLEAR_PIXEL, want != abi.CLEAR_PIXEL
        assert int(want_covered.sum()) > 3000
        # the ranks' octrees refine their shared upper levels on their own points only, so the LOD cut may differ in places from the
        # single-process octree's: same coverage up to a sliver, same depth wherever both drew the same level
        assert int((covered & want_covered).sum()) >= 0.97 * int(want_covered.sum()), name
        same_depth = (frames[0] >> np.uint64(32)) == (want >> np.uint64(32))
        assert int((same_depth & want_covered).sum()) >= 0.5 * int(want_covered.sum()), name


def _worker_hardening(rank, world, port, out_dir):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from simlod_amd import abi, camera, distributed, synthetic
    # 1. more visible nodes on one rank than gather_visible sends at once (BASELINE config 5: 4 097 on one GPU): nothing may be dropped
    n_mine = 5000 if rank == 0 else 10
    recs = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
    recs["level"][:n_mine] = 7; recs["X"][:n_mine] = np.arange(n_mine) + 1000 * rank
    vb = torch.from_numpy(recs.view(np.uint8).reshape(-1))
    for count in (n_mine, torch.tensor([n_mine], dtype=torch.int32)):
        for async_op in (False, True):
            r = distributed.gather_visible(vb, count, capacity=4096, async_op=async_op)
            got, cnts = r() if async_op else r
            assert [int(c) for c in cnts] == [5000, 10] and got.shape[1] >= 5000
            back = got[0, :5000].numpy().view(abi.node_dtype).reshape(-1)
            assert np.array_equal(back["X"], np.arange(5000)) and int(got[1, :10].numpy().view(abi.node_dtype).reshape(-1)["X"][9]) == 1009
    try:
        distributed.gather_visible(vb, abi.MAX_VISIBLE_NODES + 1, capacity=64)
        raise AssertionError("a count beyond the visible-node array must raise")
    except distributed.VisibleOverflow:
        pass
    # 2. routing in slices == routing at once (same records, same order), whatever the slice size; ranks with different input sizes
    pts, box = synthetic.terrain(300_000 + 50_000 * rank, seed=5 + rank)
    t = torch.from_numpy(np.ascontiguousarray(pts).view(np.uint8).reshape(-1, 16).copy())
    codes = distributed.cell_codes(t, box, 3)
    owner, counts = distributed.balanced_owners(codes, world, 3)
    want, want_rs = distributed.route_points(t, codes, owner)
    for slice_points in (70_000, 1_000_000):
        got, owner2, counts2, rs = distributed.partition_and_route(t, box, world, level=3, slice_points=slice_points)
        assert torch.equal(owner, owner2) and np.array_equal(counts, counts2) and rs == want_rs and torch.equal(got, want), slice_points
    # 3. frames with two in flight == the same frames one after the other
    W = H = 192
    upts, ubox = synthetic.uniform_cube(300_000, seed=33)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    u = abi.make_uniforms(W, H, T, ubox, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=2)
    o.reset(u)
    mine = upts[distributed.owner_of(upts, ubox, world) == rank]
    o.upload(mine); o.construct(u)
    seq = []
    for name, kw in FRAME_VARIANTS:
        uv = u.copy()
        for k, v in kw.items():
            uv[k] = v
        seq.append(uv)
    frames = seq + seq[::-1] + [seq[1]] * 3
    want = []
    for uv in frames:
        distributed.render_frame(o, uv)
        want.append((o._fb.copy(), o._color.copy()))
    got = {}
    distributed.render_frames_pipelined(o, frames, on_frame=lambda i, r, recs, cnts: got.__setitem__(i, (r._fb.copy(), r._color.copy(), int(cnts.sum()))))
    assert sorted(got) == list(range(len(frames)))
    for i, (fb, col) in enumerate(want):
        assert np.array_equal(got[i][0], fb) and np.array_equal(got[i][1], col), f"frame {i} differs when two frames are in flight"
    dist.destroy_process_group()


def test_gather_beyond_capacity_sliced_routing_and_pipelined_frames(built_libs, tmp_path):
    """The multi-GPU path's hardening (world 2, gloo): the visible-node gather never truncates (5 000 records through a 4 096-record
    exchange; VisibleOverflow beyond the array), routing in slices delivers exactly what routing at once delivers, and frames composed with
    two in flight are the frames composed one after the other."""
    mp.spawn(_worker_hardening, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)


def _rank_octrees(pts, box, world, u):
    """The per-rank octrees of the multi-GPU layer (level-3 cells dealt by point count), built by the oracle in one process."""
    import oracle
    from simlod_amd import abi, distributed
    t = torch.from_numpy(np.ascontiguousarray(pts).view(np.uint8).reshape(-1, 16).copy())
    codes = distributed.cell_codes(t, box, 3)
    owner, _ = distributed.balanced_owners(codes, world, 3)
    dest = owner[codes].numpy()
    trees = []
    for r in range(world):
        o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
        o.reset(u)
        o.add_points(u, pts[dest == r])
        trees.append(o)
    return trees


def test_composed_frame_equals_the_single_gpu_frame_unless_a_rank_keeps_an_upper_node_as_a_leaf(built_libs):
    """Pins DESIGN.md §9's "known limit": ranks own level-3 cells of ONE global cube, so below level 3 a rank's octree IS the single-GPU
    octree's subtree; the nodes above (levels 0-2) are shared, and each rank grows them from its own points only.  Their voxels are the
    same cells (a cell of an upper node's grid lies inside one level-3 cell: exactly one rank can set it), so the composed frame has the
    single-GPU frame's DEPTH at every pixel — unless an upper node that is INNER in the single-GPU octree stays a LEAF on some rank (the
    rank holds fewer than 50 000 points under it): that rank draws the node's points where a single GPU draws the node's voxels.  Then, and
    only then, the frames differ, and only inside the screen boxes of those nodes."""
    import oracle
    from simlod_amd import abi, camera, synthetic
    W = H = 256
    world = 2
    # uniform 1.6 M: the single octree's inner nodes above level 3 are the root and the eight level-1 nodes (200 000 points each, ~100 000 of them
    #                on either rank: inner there too); the level-2 nodes are leaves on one GPU and on the ranks.
    # terrain 3 M:   one level-2 node at the edge of the terrain is inner in the single octree and a leaf on the rank that owns few of its cells.
    for name, (pts, box), expect_equal in (("uniform 1.6 M", synthetic.uniform_cube(1_600_000, seed=8), True),
                                           ("terrain 3 M", synthetic.terrain(3_000_000, seed=4, box=(600.0, 400.0, 40.0)), False)):
        T = camera.lookat_transform((2.4 * box[0], -2.0 * box[1], 2.2 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)     # far: upper nodes are drawn by their voxels
        u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
        single = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
        single.reset(u)
        single.add_points(u, pts)
        fb_single, _ = single.render(u)
        trees = _rank_octrees(pts, box, world, u)
        fb = np.minimum.reduce([t.render(u)[0] for t in trees])                 # the composition of distributed.render_frame for a plain frame
        ds = single.dump()
        inner_single = {(int(d["level"]), int(d["X"]), int(d["Y"]), int(d["Z"])) for d in ds if d["level"] < 3 and not d["isLeaf"]}
        culprits = []                                                             # upper nodes that are inner in the single octree but a (non-empty) leaf on a rank
        for t in trees:
            for d in t.dump():
                key = (int(d["level"]), int(d["X"]), int(d["Y"]), int(d["Z"]))
                if d["level"] < 3 and d["isLeaf"] and d["numPoints"] > 0 and key in inner_single:
                    culprits.append(key)
        depth_equal = np.array_equal(fb >> np.uint64(32), fb_single >> np.uint64(32))
        if expect_equal:
            assert not culprits, (name, culprits)
            assert depth_equal, f"{name}: every rank refines the shared upper nodes: the composed frame must have the single-GPU frame's depth everywhere"
        else:
            assert culprits, f"{name}: expected a rank with fewer than 50 000 points under a shared node"
            # the frames may differ only inside the screen boxes of those nodes
            bad = np.nonzero((fb >> np.uint64(32)) != (fb_single >> np.uint64(32)))[0]
            size = float(max(box))
            M = np.asarray(T, dtype=np.float32).reshape(4, 4)
            allowed = np.zeros(W * H, dtype=bool)
            for (lv, X, Y, Z) in set(culprits):
                s = size / 2 ** lv
                cs = np.array([[(X + a) * s, (Y + b) * s, (Z + c) * s, 1.0] for a in (0, 1) for b in (0, 1) for c in (0, 1)], dtype=np.float32)
                clip = cs @ M.T
                px = (clip[:, 0] / clip[:, 3] * 0.5 + 0.5) * W
                py = (clip[:, 1] / clip[:, 3] * 0.5 + 0.5) * H
                x0, x1 = max(int(px.min()) - 2, 0), min(int(px.max()) + 3, W)
                y0, y1 = max(int(py.min()) - 2, 0), min(int(py.max()) + 3, H)
                m = np.zeros((H, W), dtype=bool); m[y0:y1, x0:x1] = True
                allowed |= m.reshape(-1)
            assert allowed[bad].all(), f"{name}: {int((~allowed[bad]).sum())} differing pixels lie outside the screen boxes of the nodes a rank kept as leaves"
