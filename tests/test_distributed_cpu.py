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
    o.set_trunk_mask(*distributed.trunk_mask(counts))          # the shared upper levels split by the GLOBAL counts: the single-GPU octree's topology
    assert distributed.global_trunk_mask(rec, box) == distributed.trunk_mask(counts)
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
    same frame, and it has the single-process frame's depth at every pixel."""
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
        assert int((want != abi.CLEAR_PIXEL).sum()) > 3000
        # the shared upper levels split by the global counts (trunk_mask): the composed frame has the single-process frame's depth at every
        # pixel (which point colours a voxel depends on batch boundaries, SURVEY.md H6 — on one GPU as well)
        bad = int(((frames[0] >> np.uint64(32)) != (want >> np.uint64(32))).sum())
        assert bad == 0, f"{name}: {bad} pixels of the composed frame have another depth than the single-process frame"


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


def _rank_octrees(pts, box, world, u, trunk=True):
    """The per-rank octrees of the multi-GPU layer (level-3 cells dealt by point count; the shared upper levels split by the GLOBAL counts:
    distributed.trunk_mask), built by the oracle in one process."""
    import oracle
    from simlod_amd import abi, distributed
    t = torch.from_numpy(np.ascontiguousarray(pts).view(np.uint8).reshape(-1, 16).copy())
    codes = distributed.cell_codes(t, box, 3)
    owner, counts = distributed.balanced_owners(codes, world, 3)
    dest = owner[codes].numpy()
    lo, hi = distributed.trunk_mask(counts)
    trees = []
    for r in range(world):
        o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
        o.reset(u)
        if trunk:
            o.set_trunk_mask(lo, hi)
        o.add_points(u, pts[dest == r])
        trees.append(o)
    return trees, (lo, hi)


def _compose_in_process(trees, uv):
    """distributed.render_frame's sequence — the four parts of kernel_render with the reductions between them — over octrees that live in
    ONE process: the reductions are numpy's (MIN of the depth planes, SUM of the colour sums, MIN of the framebuffers)."""
    u = np.ascontiguousarray(uv).reshape(1)
    hqs, boxes = bool(u["useHighQualityShading"][0]), bool(u["showBoundingBox"][0])
    for t in trees:
        t.render_part(uv, 0)
    if hqs:
        d = np.minimum.reduce([t._depth for t in trees])
        for t in trees:
            t._depth[:] = d
            t.render_part(uv, 1)
        sm = np.add.reduce([t._sums for t in trees], dtype=np.uint32)
        for t in trees:
            t._sums[:] = sm
            t.render_part(uv, 2)
    if not hqs or boxes:
        fb = np.minimum.reduce([t._fb for t in trees])
        for t in trees:
            t._fb[:] = fb
    for t in trees:
        t.render_part(uv, 3)
    for t in trees[1:]:
        assert np.array_equal(t._fb, trees[0]._fb) and np.array_equal(t._color, trees[0]._color), "the ranks hold different frames"
    return trees[0]._fb.copy(), trees[0]._color.copy()


# (eye in units of the box, Uniforms.minNodeSize): between them the views draw nodes of levels 1, 2 and 3, inner ones by their voxels
VIEWS = [((2.4, -2.0, 2.2), 24.0), ((1.8, -1.2, 1.4), 64.0), ((1.0, -0.6, 0.8), 64.0), ((1.8, -1.2, 1.4), 24.0)]


def _trunk_topology(dump):
    return {(int(d["level"]), int(d["X"]), int(d["Y"]), int(d["Z"])): bool(d["isLeaf"]) for d in dump if d["level"] < 3}


@pytest.mark.parametrize("world", [2, 4])
def test_composed_frame_equals_the_single_gpu_frame(built_libs, world):
    """VERDICT r4 item 1 / BASELINE north star ("rasterized framebuffers are bit-exact vs the reference for a fixed camera"), for N > 1 ranks.
    Ranks own level-3 cells of ONE global cube, so below level 3 a rank's octree IS the single-GPU octree's subtree.  The nodes above
    (levels 0-2) are shared; each rank splits them by the GLOBAL counts (distributed.trunk_mask -> set_trunk_mask), so on every rank they
    have the single-GPU octree's topology, hold the voxels of the rank's own cells (a voxel cell of an upper node lies inside one level-3
    cell), and the composition of distributed.render_frame is the single-GPU frame: the same DEPTH at every pixel for plain and HQS frames —
    and, when no colour is scheduling dependent (which point colours a voxel depends on batch boundaries and split times,
    on one GPU as well; a data set of ONE colour takes that out), the same 64-bit words and the same RGBA8 image.
    The single-GPU frame comes from the UNMODIFIED path: the reference's own sources where oracle/_ref is built, else the restatement with
    its mask at zero.  Without the mask the terrain's frames differ (a rank keeps an upper node as a leaf): asserted too, so that the test
    would notice a mask that does nothing."""
    import oracle
    from simlod_amd import abi, camera, synthetic
    W = H = 256
    # uniform 1.6 M: root and the eight level-1 nodes are inner everywhere, the level-2 nodes leaves (25 000 points each)
    # terrain 3 M:   level-2 nodes at the terrain's edge are inner in the single octree and hold fewer than 50 000 points on some rank
    for name, (pts, box) in (("uniform 1.6 M", synthetic.uniform_cube(1_600_000, seed=8)),
                             ("terrain 3 M", synthetic.terrain(3_000_000, seed=4, box=(600.0, 400.0, 40.0)))):
        for recolour in (False, True):
            if recolour:
                pts = pts.copy(); pts["color"] = 0xff4080c0
            u = abi.make_uniforms(W, H, np.eye(4), box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
            single = oracle.HostOctree("ref" if oracle.have_ref() else "port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
            single.reset(u)
            single.add_points(u, pts)
            trees, mask = _rank_octrees(pts, box, world, u)
            want_top = _trunk_topology(single.dump())
            inner = {k for k, leaf in want_top.items() if not leaf}
            assert mask[0] | (mask[1] << 64) == sum(1 << (0 if k[0] == 0 else 1 + (k[1] << 2 | k[2] << 1 | k[3]) if k[0] == 1 else
                                                       9 + (((k[1] >> 1) << 2 | (k[2] >> 1) << 1 | (k[3] >> 1)) << 3 | ((k[1] & 1) << 2 | (k[2] & 1) << 1 | (k[3] & 1)))) for k in inner), \
                f"{name}: trunk_mask does not name the single octree's inner upper nodes"
            for r, t in enumerate(trees):
                assert t.last_error() == 0
                assert _trunk_topology(t.dump()) == want_top, f"{name}: rank {r}'s upper levels differ from the single-GPU octree's"
            assert sum(int(t.stats["numPoints"][0]) for t in trees) == len(pts)
            drawn_levels = set()
            for eye, min_node_size in VIEWS:
                T = camera.lookat_transform((eye[0] * box[0], eye[1] * box[1], eye[2] * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
                u = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000, min_node_size=min_node_size)
                for variant, kw in FRAME_VARIANTS:
                    uv = u.copy()
                    for k, v in kw.items():
                        uv[k] = v
                    want_fb, want_color = single.render(uv)
                    drawn_levels |= {int(l) for l in single.visible["level"][single.visible["numVoxels"] > 0]}
                    fb, color = _compose_in_process(trees, uv)
                    assert int((want_fb != abi.CLEAR_PIXEL).sum()) > 1500, (name, eye, min_node_size)
                    bad = int(((fb >> np.uint64(32)) != (want_fb >> np.uint64(32))).sum())
                    assert bad == 0, f"{name}, {variant}, {world} ranks, view {eye}/{min_node_size}: {bad} pixels of the composed frame have another depth than the single-GPU frame"
                    if recolour:
                        assert np.array_equal(fb, want_fb) and np.array_equal(color, want_color), f"{name}, {variant}, {world} ranks: composed frame != single-GPU frame"
            assert ({1} if name.startswith("uniform") else {1, 2}) <= drawn_levels, f"{name}: the views must draw shared upper nodes by their voxels (levels drawn as voxels: {sorted(drawn_levels)})"
        # the control: ranks that refine the upper levels from their own points only show another LOD cut on the terrain
        if name.startswith("terrain") and world == 2:
            loose, _ = _rank_octrees(pts, box, world, u, trunk=False)
            assert any(_trunk_topology(t.dump()) != want_top for t in loose)
            differs = False
            for eye, min_node_size in VIEWS:
                T = camera.lookat_transform((eye[0] * box[0], eye[1] * box[1], eye[2] * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
                uv = abi.make_uniforms(W, H, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000, min_node_size=min_node_size)
                differs |= not np.array_equal(_compose_in_process(loose, uv)[0] >> np.uint64(32), single.render(uv)[0] >> np.uint64(32))
            assert differs, "without the mask a rank keeps an upper node as a leaf: some view must show it"
