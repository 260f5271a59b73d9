"""GPU parity: the HIP kernels, called through the C ABI of libsimlod_hip.so, against the CPU oracle and the golden
fixtures (generated from the reference's own sources) on the same seeded inputs.

Scheduling-dependent facts (node indices, chunk addresses, sample order inside a node, which point colours a
voxel) are compared through order-independent forms; everything else — topology, per-node point multisets, occupancy bitsets,
voxel positions, every counter, allocator accounting, pre-EDL framebuffers on one and the same image — must be identical."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import oracle
from cases import CASES, H, W, batches_of, case, uniforms_for
from simlod_amd import abi, camera, synthetic
from test_golden import load_golden
from util import (STATS_BUILD_FIELDS, STATS_RENDER_FIELDS, assert_dumps_equal, assert_stats_equal, host_image_of, points_multiset_hash,
                  voxel_colors_are_member)

pytestmark = pytest.mark.gpu


def _device(**kw):
    from simlod_amd.runtime import DeviceOctree
    # (SIMLOD_TEST_MOMENTARY_MB / SIMLOD_TEST_PERSISTENT_MB: the same tests with other default buffer sizes — exact mode ingests a launch's batches in
    # groups where the momentary buffer has room for it and the persistent buffer is far from the reference's guard: construct.hip account_group)
    kw.setdefault("persistent_bytes", int(os.environ.get("SIMLOD_TEST_PERSISTENT_MB", "8192")) << 20)
    if "SIMLOD_TEST_MOMENTARY_MB" in os.environ:
        kw.setdefault("momentary_bytes", int(os.environ["SIMLOD_TEST_MOMENTARY_MB"]) * 1_000_000)
    kw.setdefault("max_pixels", 1920 * 1080)
    if os.environ.get("SIMLOD_TEST_COALESCE") == "1":      # (debugging aid: the same tests in coalesced mode — only those that do not compare granularity-dependent counters can pass)
        kw["coalesce"] = True
    dev = DeviceOctree("cuda:0", **kw)
    # The reference host never clears its momentary / render / persistent buffers (main_progressive_octree.cpp:549-586 only
    # cuMemAlloc's them): poison them so that any kernel that trusts bytes it did not write itself faults or miscompares here.
    dev.momentary.fill_(0xA5)
    dev.render_buffer.fill_(0xA5)
    dev.persistent.fill_(0xA5)
    return dev


def _ingest(dev, u, batches):
    dev.reset(u)
    for b in batches:
        if dev.uploaded_host - dev.processed() >= dev.ring_slots:
            dev.drain(u)
        dev.upload(b)
    dev.drain(u)


def _oracle_render(nodes, nn, u):
    Wd, Hd = int(u["width"]), int(u["height"])
    fb = np.zeros(Wd * Hd, dtype=np.uint64)
    col = np.zeros(Wd * Hd, dtype=np.uint32)
    vis = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
    stats = np.zeros(1, dtype=abi.stats_dtype)
    stats["numNodes"] = nn
    uu = np.ascontiguousarray(u).reshape(1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    oracle.port_lib().oracle_render(None, p(uu), p(nodes), p(stats), p(fb), p(col), p(vis), 1)
    return fb, col, stats[0]


@pytest.mark.parametrize("name", CASES)
def test_construct_matches_golden_and_oracle(built_libs, name):
    g = load_golden(name)
    pts, box, batch, T = case(name)
    dev = _device(ring_slots=8)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, batches_of(name, pts, batch))
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0, f"device error bits {int(ds['dbg']):#x}"
    assert_stats_equal(ds, g["build_stats"][0], STATS_BUILD_FIELDS, name)
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), g["dump"], name)
    oracle.check_invariants(nodes, nn)
    assert voxel_colors_are_member(nodes, nn, pts, box) == int(nodes["numVoxelsStored"][:nn].sum())


@pytest.mark.parametrize("kind,n,batch", [("uniform", 1_000_000, 1_000_000), ("uniform", 3_000_000, 1_000_000),
                                          ("terrain", 4_000_000, 1_000_000), ("hotspot", 3_000_000, 1_000_000)])
def test_construct_full_batches_match_oracle(built_libs, kind, n, batch):
    pts, box = {"uniform": lambda: synthetic.uniform_cube(n, seed=1234), "terrain": lambda: synthetic.terrain(n, seed=7),
                "hotspot": lambda: synthetic.hotspot(n, seed=11)}[kind]()
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=8)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + batch] for i in range(0, n, batch)])
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, batch)
    assert ref.last_error() == 0
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, kind)
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), kind)


@pytest.mark.parametrize("stops_telling", [False, True])
def test_a_host_that_tells_nothing_about_its_uploads_is_served_all_the_same(built_libs, stops_telling):
    """Launch sizing (simlod_hip.cpp launch_plan): the library enqueues kernels for the batches it believes pending.  A host behind shim/cuda.h passes its
    upload-counter writes on; one that does not — or stops doing so — must still have every batch ingested: the only counter value the library then knows is the
    zero of its own reset, which must not read as "nothing uploaded" (bench.py's host_tells_nothing leg died on exactly this).  Resets in between,
    launches enqueued back to back, the octree equal to the oracle's at the end."""
    from simlod_amd.runtime import DeviceOctree
    pts, box = synthetic.terrain(7_000_000, seed=7)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = DeviceOctree("cuda:0", persistent_bytes=8 << 30, ring_slots=8, max_pixels=W * H, sizes_launches=stops_telling)
    u = dev.uniforms(W, H, T, box)
    batches = [pts[i:i + 1_000_000] for i in range(0, len(pts), 1_000_000)]
    for attempt in range(3):
        if stops_telling and attempt == 1:
            dev.notifies = False              # (the first ingest was told about, from here on the counter is written behind the library's back)
        dev.reset(u)
        for b in batches:
            dev.upload(b)
        dev.drain(u)
        ds = dev.read_stats()
        assert int(ds["numPointsProcessed"]) == len(pts) and int(ds["dbg"]) == 0, attempt
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, 1_000_000)
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "blind host")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "blind host")


def test_a_root_cascade_with_over_a_hundred_upper_leaves_again_and_again(built_libs):
    """The first batch of a small terrain splits the root three levels deep in one round: ~120 of the cascade's nodes are leaves at level <= 3, whose cells of
    the top table the whole workgroup fills from a list in LDS.  Rounds 3-6 gave that list 72 entries; the rest landed on the grid pointers behind it, and once
    in a hundred ingests a path entry got a wild grid address (a memory fault in k_voxelize; found by tools/stress_small.py).  Forty fresh, poisoned octrees:
    every one must end as the first did."""
    pts, box = synthetic.terrain(1_500_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    first = None
    for _ in range(40):
        dev = _device(ring_slots=2)
        u = dev.uniforms(W, H, T, box)
        _ingest(dev, u, [pts[:1_000_000], pts[1_000_000:]])
        ds = dev.read_stats()
        got = {f: int(ds[f]) for f in STATS_BUILD_FIELDS}
        assert int(ds["dbg"]) == 0
        first = first or got
        assert got == first
    nodes, pers, nn = host_image_of(dev)
    upper_leaves = int(((nodes["level"][:nn] <= 3) & (nodes["children"][:nn] == 0).all(axis=1)).sum())
    assert upper_leaves > 72, f"the case must list more upper leaves than the old array held ({upper_leaves})"
    oracle.check_invariants(nodes, nn)


@pytest.mark.parametrize("kind,n,batch", [("terrain", 6_000_000, 1_000_000), ("hotspot", 3_000_000, 400_000)])
def test_descent_through_node_children_where_the_child_words_say_irregular(built_libs, kind, n, batch):
    """k_count descends through one 32-bit word per node (first child + which children are leaves: eight consecutive node slots, as the reference's
    and this builder's splits make them, voxels.cu:316-343).  An image whose children are NOT eight consecutive nodes gets KID_IRREGULAR words and is
    descended through Node.children as before round 6; SIMLOD_DEBUG_IRREGULAR_CHILDREN marks every inner node so: same octree, same counters."""
    pts, box = {"terrain": lambda: synthetic.terrain(n, seed=7), "hotspot": lambda: synthetic.hotspot(n, seed=11)}[kind]()
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=8)
    dev.tune("SIMLOD_DEBUG_IRREGULAR_CHILDREN", 1)
    try:
        u = dev.uniforms(W, H, T, box)
        _ingest(dev, u, [pts[i:i + batch] for i in range(0, n, batch)])
    finally:
        dev.tune("SIMLOD_DEBUG_IRREGULAR_CHILDREN", None)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, batch)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, kind)
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), kind)


@pytest.mark.parametrize("total,first,batch", [(4_000_000, 2_000_000, 500_000), (90_000, 30_000, 15_000)])
def test_ingest_continues_into_an_image_this_library_did_not_build(built_libs, total, first, batch):
    """An octree image as another implementation leaves it — the reference's kernel_construct, say —: the momentary buffer knows nothing of it (no stamp,
    no side tables) and the spare bytes of the lists' head chunks (Chunk::size / padding_0, where this builder keeps the address of a list's last
    chunk) hold whatever was there.  The next launch restores all of it from the node array inside k_begin; the rest of the ingest then ends in the
    oracle's octree, allocator and chunk-pool counters included."""
    # (the second case: an image of 30 000 points — its root is still a LEAF and has a point list AND a voxel list, it samples itself
    # (voxels.cu:449-463): both lists' tail words have to be restored; the root then splits in the continued ingest)
    import torch
    pts, box = synthetic.terrain(total, seed=7)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=8)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    for i in range(0, first, batch):
        dev.upload(pts[i:i + batch])
    dev.drain(u)
    torch.cuda.synchronize()
    nodes, pers, nn, nodes_base, pers_base = dev.download_image()
    heads = np.concatenate([nodes["points"][:nn], nodes["voxelChunks"][:nn]]).astype(np.uint64)
    heads = heads[heads != 0]
    assert len(heads) > 50 if total > 1_000_000 else (nn == 1 and len(heads) == 2)
    where = torch.from_numpy(((heads - np.uint64(pers_base)).astype(np.int64) + 16000)[:, None] + np.arange(8, dtype=np.int64)[None, :]).reshape(-1).to(dev.device)
    dev.persistent[where] = 0xA5
    # (everything in the momentary buffer but the recycle stack of released chunks — bytes 4096 .. 4096 + 8 000 000 —, which Stats.numAllocatedChunks indexes:
    # like the reference's chunkQueue, progressive_octree_voxels.cu:856, it lives there and has to survive between launches)
    dev.momentary[:4096].fill_(0xA5)
    dev.momentary[4096 + 8_000_000:].fill_(0xA5)
    for i in range(first, len(pts), batch):
        dev.upload(pts[i:i + batch])
        if total < 1_000_000:
            dev.drain(u)                          # (batch by batch: the root keeps sampling itself, then splits)
    dev.drain(u)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, batch)
    assert ref.last_error() == 0
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0, f"device error bits {int(ds['dbg']):#x}"
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "continued")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "continued")
    oracle.check_invariants(nodes, nn)


def test_construct_without_pending_batches_is_a_noop_and_render_before_reset_is_harmless(built_libs):
    dev = _device(ring_slots=2)
    box = np.array([1, 1, 1], dtype=np.float32)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    u = dev.uniforms(W, H, T, box)
    dev.render(u)                           # H11: the reference renders before any reset; buffers are zero -> empty frame
    assert (dev.framebuffer(W, H) == abi.CLEAR_PIXEL).all()
    dev.reset(u)
    dev.construct(u)                        # nothing uploaded
    s = dev.read_stats()
    assert (int(s["numNodes"]), int(s["batchletIndex"]), int(s["numPoints"]), int(s["dbg"])) == (1, 0, 0, 0)
    assert int(s["allocatedBytes_persistent"]) == 16 + abi.alloc_round(abi.GRID_BYTES)
    dev.render(u)
    assert (dev.framebuffer(W, H) == abi.CLEAR_PIXEL).all()
    assert (dev.color(W, H)[: (W // 16) * 16] & 0xffffff == 0x332211).all()


def test_too_small_momentary_buffer_is_reported_not_silently_overrun(built_libs):
    from simlod_amd.runtime import SimlodError
    dev = _device(ring_slots=2, momentary_bytes=300_000_000)
    pts, box = synthetic.uniform_cube(10_000, seed=2)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    u = dev.uniforms(W, H, T, box)
    u["momentaryBufferCapacity"] = 1_000_000
    dev.reset(u)
    dev.upload(pts)
    with pytest.raises(SimlodError):
        dev.construct(u)
    s = dev.read_stats()
    assert int(s["dbg"]) & 0x1 and int(s["batchletIndex"]) == 0


def test_persistent_memory_guard_stops_ingest_like_the_reference(built_libs):
    """voxels.cu:896-912: a batch is taken only while allocator offset + 200 MB < persistentBufferCapacity; otherwise
    Stats.memCapacityReached is raised and the launch does nothing — and neither do later launches.  Same stopping batch,
    same octree as the restatement; nothing is written beyond the capacity the kernels were told."""
    import torch
    n = 6_000_000
    pts, box = synthetic.uniform_cube(n, seed=17)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    cap = 260_000_000                                             # 60 MB of octree, then the guard
    dev = _device(ring_slots=8, persistent_bytes=cap + 4096)
    dev.persistent[cap:].fill_(0x3C)                              # canary behind the capacity the uniforms announce
    u = dev.uniforms(W, H, T, box)
    u["persistentBufferCapacity"] = cap
    dev.persistent_bytes = cap
    dev.reset(u)
    for i in range(0, n, 1_000_000):
        dev.upload(pts[i:i + 1_000_000])
    for _ in range(4):                                            # the frame loop keeps launching; nothing may move any more
        dev.construct(u)
    torch.cuda.synchronize()
    ds = dev.read_stats()
    ref = oracle.HostOctree("port", persistent_bytes=cap, ring_slots=8)
    uh = u.copy()
    ref.reset(uh)
    for i in range(0, n, 1_000_000):
        ref.upload(pts[i:i + 1_000_000])
    for _ in range(4):
        ref.construct(uh)
    assert int(ds["memCapacityReached"]) == 1 == int(ref.stats["memCapacityReached"][0])
    assert 0 < int(ds["batchletIndex"]) < 6
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "memory guard")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "memory guard")
    assert int(ds["allocatedBytes_persistent"]) <= cap and bool((dev.persistent[cap:] == 0x3C).all())


def test_full_node_array_stops_splitting_but_keeps_every_point(built_libs):
    """The reference's node array holds 263 157 nodes (main_progressive_octree.cpp:552) and its kernel writes past the end when
    more are needed.  Here a split that finds no eight free slots is refused: the leaf keeps growing, Stats.dbg says so."""
    from simlod_amd.runtime import lib
    n = 1_000_000
    pts, box = synthetic.uniform_cube(n, seed=5)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    try:
        dev = _device(ring_slots=4, max_nodes=41)                     # room for five splits; the data wants nine (73 nodes)
        dev.nodes[41 * 152:].fill_(0x77) if dev.nodes.numel() > 41 * 152 else None
        u = dev.uniforms(W, H, T, box)
        _ingest(dev, u, [pts[i:i + 250_000] for i in range(0, n, 250_000)])
        ds = dev.read_stats()
        assert int(ds["dbg"]) == 0x8 and int(ds["numNodes"]) == 41
        assert int(ds["numPoints"]) == n == int(ds["numPointsProcessed"])
        nodes, pers, nn = host_image_of(dev)
        tot = oracle.check_invariants(nodes, nn, allow_overfull=True)
        assert tot["points"] == n
        d = oracle.dump_image(nodes, nn)
        hs, hx = points_multiset_hash(pts)
        with np.errstate(over="ignore"):
            assert hs == np.uint64(d["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(d["pointsXor"])
        dev.render(u)                                                 # and the octree is still drawable
        assert int((dev.framebuffer(W, H) != abi.CLEAR_PIXEL).sum()) > 1000
    finally:
        lib().simlod_set_node_capacity(263_157)


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_memory_guard_trips_while_launches_take_several_batches_each(built_libs, overlap, monkeypatch):
    """voxels.cu:896-912 looks at the allocator after the WHOLE previous batch.  Here a batch's front half starts while the back halves of
    the batches before it are still allocating voxel chunks, so a launch takes further batches only while the allocator is a worst-case
    voxel half away from the guard, and goes batch by batch from there on (construct.hip voxel_half_slack) — the stopping batch and
    every counter must still be the reference's.  14 M points in 100 000-point batches against 800 MB: launches of many batches up to
    ~370 MB, single-batch launches up to 600 MB, then the guard; two-stream pipeline on and off."""
    import torch
    from simlod_amd.runtime import SimlodError
    monkeypatch.setenv("SIMLOD_OVERLAP_TAIL", overlap)
    n, step, cap = 14_000_000, 100_000, 800_000_000
    pts, box = synthetic.uniform_cube(n, seed=17)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    dev = _device(ring_slots=abi.BATCH_STREAM_SIZE, persistent_bytes=cap + 4096)
    dev.persistent[cap:].fill_(0x3C)                              # canary behind the capacity the uniforms announce
    u = dev.uniforms(W, H, T, box)
    u["persistentBufferCapacity"] = cap
    dev.persistent_bytes = cap
    ref = oracle.HostOctree("port", persistent_bytes=cap, ring_slots=abi.BATCH_STREAM_SIZE)
    uh = u.copy()
    dev.reset(u); ref.reset(uh)
    per_launch = []
    for i in range(0, n, step):
        if dev.uploaded_host - dev.processed() >= abi.MAX_BATCHES_PER_LAUNCH:
            before = dev.processed()
            dev.construct(u); ref.construct(uh)
            per_launch.append(dev.processed() - before)
            if per_launch[-1] == 0:
                break
        dev.upload(pts[i:i + step]); ref.upload(pts[i:i + step])
    while per_launch[-1] != 0:                                    # the frame loop keeps launching until nothing moves any more
        before = dev.processed()
        dev.construct(u)
        per_launch.append(dev.processed() - before)
    for _ in range(3):
        dev.construct(u)
    while True:                                                   # (the restatement takes what its launches let it: until it stands still too)
        before = int(ref.stats["batchletIndex"][0])
        ref.construct(uh)
        if int(ref.stats["batchletIndex"][0]) == before:
            break
    torch.cuda.synchronize()
    ds = dev.read_stats()
    assert int(ds["memCapacityReached"]) == 1 == int(ref.stats["memCapacityReached"][0])
    assert max(per_launch) >= 10 and per_launch.count(1) >= 10, f"expected launches of many batches, then of one: {per_launch}"
    assert 60 < int(ds["batchletIndex"]) < n // step
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "memory guard")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "memory guard")
    assert (dev.persistent[cap:] == 0x3C).all()
    with pytest.raises(SimlodError):
        dev.drain(u)


def test_collisions_at_max_depth_and_points_on_the_box_faces(built_libs):
    """70 000 identical points force twenty split rounds inside one batch, down to level 20 where a node cannot split any more;
    8 000 points sit exactly on the faces / corners of the bounding box (coordinate == boxMax quantises to 2^20 and, as in the
    reference, wraps into the low child).  Tree shape, voxels, grids and counts must be the reference restatement's.  One
    deliberate difference: the reference does not count after its 20th split, allocates no chunks for the level-20 leaf and
    drops the points that land there (voxels.cu:394-412, 599-604 — the restatement reports NULL_CHUNK); here every point is stored."""
    rs = np.random.RandomState(4)
    base, box = synthetic.uniform_cube(60_000, seed=9)
    same = np.repeat(base[:1], 70_000)
    same["x"], same["y"], same["z"] = np.float32(0.3), np.float32(0.6), np.float32(0.2)
    corners = np.repeat(base[:1], 8_000)
    for k in "xyz":
        corners[k] = rs.choice(np.array([0.0, 1.0], dtype=np.float32), 8_000)
    pts = np.concatenate([base[:30_000], same, corners, base[30_000:]])
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    dev = _device(ring_slots=4)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + 50_000] for i in range(0, len(pts), 50_000)])
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=4)
    ref.reset(u)
    ref.add_points(u, pts, 50_000)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksVoxels",
                                          "batchletIndex", "numPointsProcessed"], "collisions")
    nodes, pers, nn = host_image_of(dev)
    got, want = oracle.dump_image(nodes, nn), ref.dump()
    got, want = got[np.argsort(got["key"])], want[np.argsort(want["key"])]
    assert int(got["level"].max()) == abi.MAX_DEPTH
    for f in ("key", "level", "X", "Y", "Z", "isLeaf", "numPoints", "numVoxels", "numVoxelsStored", "hasGrid", "gridPopcount", "gridHash",
              "voxelPosSum", "voxelPosXor", "voxelChunks", "childMask"):
        assert np.array_equal(got[f], want[f]), f
    shallow = got["level"] < abi.MAX_DEPTH                          # what the reference stores it stores identically
    for f in ("counter", "pointChunks", "pointsSum", "pointsXor"):
        assert np.array_equal(got[f][shallow], want[f][shallow]), f
    # the deviation, exactly: ONE node differs — the level-20 leaf of the collision chain.  Both count the k identical points that reach it
    # (numPoints equal, above); the reference allocates it no chunk and stores none of them, here it holds ceil(k / 1000) chunks with
    # exactly those k points and nothing else
    deep = ~shallow
    assert int(deep.sum()) == 8 and int((got["numPoints"][deep] > 0).sum()) == 1       # the eight children of the 20th split, one of them hit
    k = int(got["numPoints"][deep].sum())
    assert 0 < k <= 70_000 and k == int(want["numPoints"][deep].sum())
    assert int(want["pointChunks"][deep].sum()) == 0 and int(want["pointsSum"][deep].sum()) == 0 and int(np.bitwise_xor.reduce(want["pointsXor"][deep])) == 0
    assert int(got["pointChunks"][deep].sum()) == -(-k // abi.POINTS_PER_CHUNK)
    ks, kx = points_multiset_hash(same[:k])
    with np.errstate(over="ignore"):
        assert ks == np.uint64(got["pointsSum"][deep].sum()) and kx == np.bitwise_xor.reduce(got["pointsXor"][deep])
    tot = oracle.check_invariants(nodes, nn)
    assert tot["points"] == len(pts)
    hs, hx = points_multiset_hash(pts)
    with np.errstate(over="ignore"):
        assert hs == np.uint64(got["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(got["pointsXor"])
    deep = int(nodes["numVoxelsStored"][:nn][nodes["level"][:nn] > 12].sum())    # one voxel per level of the collision chain
    assert voxel_colors_are_member(nodes, nn, pts, box, max_level=12) == int(nodes["numVoxelsStored"][:nn].sum()) - deep


def test_scarce_scratch_defers_splits_without_losing_a_point(built_libs):
    """Scattered input makes hundreds of leaves cross the limit in the same batch.  With a momentary buffer that can hold only
    a fraction of their stored points (the reference drops points here) the splits that do not fit are deferred:
    the leaves stay intact, grow past 50 000, and split in a later batch.  Every point must be in the octree, the image must
    be structurally sound, and once scratch space is plentiful again the late splits must have caught up."""
    from simlod_amd.runtime import lib
    n = 6_000_000
    pts, box = synthetic.uniform_cube(n, seed=99)
    pts["z"] *= np.float32(0.02)                                  # a slab: ~2-D density, leaves fill up together
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    small = int(lib().simlod_construct_buffer_min_bytes()) + 6_000_000      # ~250 k spilled points per batch instead of millions
    dev = _device(ring_slots=8, momentary_bytes=small)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + 1_000_000] for i in range(0, n, 1_000_000)])
    ds = dev.read_stats()
    assert int(ds["dbg"]) & ~0x2 == 0, f"unexpected device error bits {int(ds['dbg']):#x}"
    assert int(ds["dbg"]) & 0x2, "the test is meant to exhaust the spill space"
    assert int(ds["numPoints"]) == n and int(ds["numPointsProcessed"]) == n
    nodes, pers, nn = host_image_of(dev)
    tot = oracle.check_invariants(nodes, nn, allow_overfull=True)
    assert tot["points"] == n
    assert voxel_colors_are_member(nodes, nn, pts, box) == int(nodes["numVoxelsStored"][:nn].sum())
    d = oracle.dump_image(nodes, nn)
    leaves = d[d["isLeaf"] == 1]
    assert (leaves["numPoints"] > abi.MAX_POINTS_PER_NODE).any(), "expected leaves whose split is still pending"
    # the multiset of stored points is the input
    hs, hx = points_multiset_hash(pts)
    with np.errstate(over="ignore"):
        assert hs == np.uint64(d["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(d["pointsXor"])


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_time_budget_stops_launches_early_without_changing_the_octree(built_libs, overlap, monkeypatch):
    """voxels.cu:22, 936-949: a launch stops taking batches once it has run for 10 ms; the host launches again next frame.  With the
    budget forced down to 100 us (SIMLOD_DEBUG_BUDGET_US) every launch stops after a batch or two — with the voxel half of the stopped
    batch still on the side stream (overlap 1) or behind it on the caller's (overlap 0) — and 60 batches take dozens of launches.  The
    octree and every counter must be what the restatement builds without ever running out of time."""
    monkeypatch.setenv("SIMLOD_OVERLAP_TAIL", overlap)
    monkeypatch.setenv("SIMLOD_DEBUG_BUDGET_US", "100")
    n, batch = 6_000_000, 100_000
    pts, box = synthetic.terrain(n, seed=21)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=50)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    launches = 0
    for i in range(0, n, batch):
        if dev.uploaded_host - dev.processed() >= dev.ring_slots:
            launches += dev.drain(u)
        dev.upload(pts[i:i + batch])
    launches += dev.drain(u)
    # (a launch that runs out of time still completes the batch or two whose front half is already under way)
    assert launches >= 10, f"the budget was meant to cut the launches short ({launches} launches for 60 batches)"
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=50)
    ref.reset(u)
    ref.add_points(u, pts, batch)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "budget")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "budget")


def test_deferred_splits_catch_up_to_the_oracles_octree(built_libs):
    """The regime BASELINE config 3's adversarial replay enters (Stats.dbg bit 0x2): with too little spill space hundreds of leaves grow past
    50 000 instead of splitting.  Once space is plentiful and every such leaf is touched again, the late splits must produce the octree
    the restatement builds without ever deferring: same topology, per-node multisets, occupancy bitsets, voxel positions and counts
    (everything that does not depend on WHEN a leaf split; the allocator / chunk-pool counters do)."""
    from simlod_amd.runtime import lib
    n = 6_000_000
    pts, box = synthetic.uniform_cube(n + 2_000_000, seed=99)
    pts["z"] *= np.float32(0.02)                                  # a slab: ~2-D density, leaves fill up together
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    small = int(lib().simlod_construct_buffer_min_bytes()) + 6_000_000
    dev = _device(ring_slots=8, momentary_bytes=300_000_000)
    u = dev.uniforms(W, H, T, box)
    us = u.copy()
    us["momentaryBufferCapacity"] = small                         # phase 1: room for ~250 k moved points per batch
    dev.reset(us)
    for i in range(0, n, 1_000_000):
        dev.upload(pts[i:i + 1_000_000])
        dev.drain(us)
    ds = dev.read_stats()
    assert int(ds["dbg"]) & 0x2 and int(ds["dbg"]) & ~0x2 == 0, f"phase 1 was meant to exhaust the spill space only ({int(ds['dbg']):#x})"
    nodes, pers, nn = host_image_of(dev)
    d = oracle.dump_image(nodes, nn)
    assert (d["numPoints"][d["isLeaf"] == 1] > abi.MAX_POINTS_PER_NODE).any(), "expected leaves whose split is pending"
    for i in range(n, n + 2_000_000, 1_000_000):                  # phase 2: the full buffer; 2 M more points over the same slab touch every leaf
        dev.upload(pts[i:i + 1_000_000])
        dev.drain(u)
    ds = dev.read_stats()
    nodes, pers, nn = host_image_of(dev)
    got = oracle.dump_image(nodes, nn)
    assert not (got["numPoints"][got["isLeaf"] == 1] > abi.MAX_POINTS_PER_NODE).any(), "a deferred split is still pending"
    ref = oracle.HostOctree("port", persistent_bytes=2 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, 1_000_000)
    want = ref.dump()
    got, want = got[np.argsort(got["key"])], want[np.argsort(want["key"])]
    assert len(got) == len(want)
    for f in GRANULARITY_FREE_FIELDS:
        assert np.array_equal(got[f], want[f]), f
    assert_stats_equal(ds, ref.stats[0], ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksPoints", "numChunksVoxels",
                                          "batchletIndex", "numPointsProcessed"], "deferred")
    assert voxel_colors_are_member(nodes, nn, pts, box) == int(nodes["numVoxelsStored"][:nn].sum())


def test_las_stream_of_60m_points_wraps_the_ring_and_equals_the_oracle(built_libs, tmp_path):
    """BASELINE config 3's path at a size the serial restatement finishes in seconds: a scan-ordered LAS 1.4 file of 60 M points is read in
    1 M-point batches, decoded on the device into the 50-slot ring (which wraps), and ingested incrementally; the full octree dump and
    Stats must equal the restatement's on the oracle-decoded points.  Stats.dbg must stay 0 (no deferred split on this input)."""
    from simlod_amd import lasio
    n = 60_000_000
    pts0, box = synthetic.terrain_scan(n, seed=7)
    path = str(tmp_path / "scan60m.las")
    h = lasio.points_to_las(path, pts0, box, fmt=2, scale=0.001, world_min=(694000.0, 3915000.0, -3.0), version=(1, 4))
    del pts0
    tr = tuple(-m for m in h.min)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(persistent_bytes=6 << 30)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    dev.add_las(u, path)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0 and int(ds["numPoints"]) == n and int(ds["batchletIndex"]) == 60
    ref = oracle.HostOctree("port", persistent_bytes=6 << 30, ring_slots=50)
    uh = u.copy()
    uh["persistentBufferCapacity"] = 6 << 30
    ref.reset(uh)
    for first, count in lasio.batches(h, abi.MAX_BATCH_SIZE):
        if int(ref.num_uploaded[0]) - int(ref.stats["batchletIndex"][0]) >= 50:
            while int(ref.stats["batchletIndex"][0]) < int(ref.num_uploaded[0]):
                ref.construct(uh)
        ref.upload(oracle.decode_las_port(lasio.read_records(path, h, first, count), h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tr)))
    while int(ref.stats["batchletIndex"][0]) < int(ref.num_uploaded[0]):
        ref.construct(uh)
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "las 60 M")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "las 60 M")


@pytest.mark.parametrize("batch,n", [(abi.MAX_BATCH_SIZE, 53_400_000), (130_000, 9_000_000)])
def test_resident_points_streamed_through_the_ring_by_the_uploader_equal_the_oracle(built_libs, batch, n):
    """BASELINE config 4's mechanism (bench.py --gpus N / --stream, DeviceOctree.stream): the rank's points lie in device memory; an uploader on
    its own stream copies runs of them into free ring slots and publishes sizes and counter behind the copies, never more than a ring ahead
    of what the host has seen processed; kernel_construct is launched once per frame and takes whatever has been published by then.  More
    batches than the ring has slots (54 of 1 M with a short last one; 70 of 130 000), device-generated terrain: octree and Stats == oracle."""
    import torch
    box = (6000.0, 4000.0, 400.0)
    dev = _device(persistent_bytes=6 << 30)
    src = torch.empty(n * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(src, 0, n, 11, 1, box, swath_width=250.0)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    launches = dev.stream(u, src, n, batch=batch)
    ds = dev.read_stats()
    nb = (n + batch - 1) // batch
    assert int(ds["dbg"]) == 0 and int(ds["numPoints"]) == n and int(ds["batchletIndex"]) == nb > abi.BATCH_STREAM_SIZE and launches >= 3
    pts = src.cpu().numpy().view(abi.point_dtype)
    ref = oracle.HostOctree("port", persistent_bytes=6 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
    uh = u.copy()
    uh["persistentBufferCapacity"] = 6 << 30
    ref.reset(uh)
    ref.add_points(uh, pts, batch)
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "streamed")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "streamed")


# ---- render --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("hqs", [False, True])
def test_render_of_a_reference_built_image_matches_golden_hash(built_libs, name, hqs):
    """feed kernel_render an octree image built by the oracle (pointer-rebased upload).  The image is the
    one the reference's own sources build (the restatement is byte-identical to it, test_oracle_pin.py), so the pre-EDL
    framebuffer must hash to the golden value."""
    g = load_golden(name)
    pts, box, batch, T = case(name)
    u = uniforms_for(box, T, hqs=hqs)
    o = oracle.HostOctree("port", persistent_bytes=1 << 28, ring_slots=8)
    o.reset(u)
    for b in batches_of(name, pts, batch):
        o.upload(b)
    while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
        o.construct(u)
    nn = int(o.stats["numNodes"][0])
    used = int(o.stats["allocatedBytes_persistent"][0])
    dev = _device(persistent_bytes=1 << 28, ring_slots=1)
    nodes, pers = o.nodes[:nn].copy(), o.persistent[:used].copy()
    oracle.rebase_image_to(nodes, nn, pers, o.nodes.ctypes.data, o.persistent.ctypes.data, dev.nodes.data_ptr(), dev.persistent.data_ptr())
    dev.upload_image(nodes, pers, nn)
    dev.render(u)
    fb = dev.framebuffer(W, H)
    mode = "hqs" if hqs else "plain"
    assert int((fb != abi.CLEAR_PIXEL).sum()) == int(g[f"fb_nonbg_{mode}"][0])
    assert hashlib.sha256(fb.tobytes()).digest() == g[f"fb_sha256_{mode}"].tobytes(), f"{name}/{mode}: framebuffer differs from the reference's"
    assert_stats_equal(dev.read_stats(), g[f"render_stats_{mode}"][0], STATS_RENDER_FIELDS, name)
    if name == "uniform_3x40k":
        # the EDL'd RGBA8 output against the reference's own EDL block (tests/golden/make_golden_edl.py): +-1 per channel (log2 / exp),
        # every full 16x16 tile, last row excepted (there the reference reads one pixel past the framebuffer, render.cu:1303)
        want = np.load(os.path.join(os.path.dirname(__file__), "golden", "edl_uniform_3x40k.npz"))[f"color_{mode}"]
        got = dev.color(W, H)
        d = np.abs(got.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
        rows = np.arange(W * H) // W
        assert int(d[rows < H - 1].max()) <= 1, f"{mode}: {int((d[rows < H - 1] > 1).sum())} pixels of the EDL output differ from the reference's by more than 1"


@pytest.mark.parametrize("variant", ["plain", "hqs", "plain_ps2", "hqs_ps3", "by_node", "by_lod_hqs", "plain_boxes", "hqs_boxes"])
def test_render_bit_exact_on_device_built_octree(built_libs, variant):
    """Build on the GPU, render on the GPU, then hand the downloaded image to the oracle's rasteriser: same image in, the
    pre-EDL uint64 framebuffer must be bit-identical; the EDL'd RGBA8 output within 1 per channel (log2/exp)."""
    pts, box = synthetic.uniform_cube(1_000_000, seed=1234)
    Wd = Hd = 512
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), Wd, Hd)
    dev = _device(ring_slots=2)
    u = dev.uniforms(Wd, Hd, T, box)
    _ingest(dev, u, [pts])
    u["useHighQualityShading"] = 1 if "hqs" in variant else 0
    u["pointSize"] = 2 if "ps2" in variant else 3 if "ps3" in variant else 1
    u["colorByNode"] = 1 if "by_node" in variant else 0
    u["colorByLOD"] = 1 if "by_lod" in variant else 0
    u["showBoundingBox"] = 1 if "boxes" in variant else 0        # debug lines: node boxes + frustum (rasterization.cuh:90-183)
    dev.render(u)
    fb_dev, col_dev, ds = dev.framebuffer(Wd, Hd), dev.color(Wd, Hd), dev.read_stats()
    nodes, pers, nn = host_image_of(dev)
    fb, col, st = _oracle_render(nodes, nn, u)
    assert_stats_equal(ds, st, STATS_RENDER_FIELDS, variant)
    diff = np.nonzero(fb_dev != fb)[0]
    assert len(diff) == 0, f"{len(diff)} pixels differ, first {diff[:5]}: dev {fb_dev[diff[:5]]} oracle {fb[diff[:5]]}"
    assert int((fb != abi.CLEAR_PIXEL).sum()) > 10_000
    assert int(np.abs(col_dev.view(np.uint8).astype(np.int16) - col.view(np.uint8).astype(np.int16)).max()) <= 1
    dev.render(u)                           # a frame is a pure function of (image, uniforms)
    assert np.array_equal(dev.framebuffer(Wd, Hd), fb_dev)


@pytest.mark.parametrize("variant", ["odd_size_hqs", "odd_size_plain_boxes", "tiny_frame", "inside_the_cloud_hqs", "inside_the_cloud_plain",
                                     "point_size_5", "fine_lod_hqs", "boxes_only", "terrain_close_hqs", "hotspot_tiles_hqs", "hotspot_tiles_plain",
                                     "grazing_plain", "grazing_hqs", "grazing_hqs_by_node", "grazing_plain_small_pool", "grazing_hqs_small_pool", "grazing_plain_no_bins"])
def test_render_edge_cases_bit_exact(built_libs, variant):
    """Frame sizes that are not multiples of the 16-pixel EDL tile (nor of the 32-pixel LDS tile), a camera inside the point cloud
    (samples behind the eye, w <= 0), point sizes that reach over the frame border, a LOD threshold that makes thousands of nodes
    visible, lines without points, nodes small enough on screen for the LDS-tile path, a camera that skims the terrain (leaves much larger
    on screen than an LDS tile: their samples are sorted into the screen bins — render.hip r_overflow —, also with a bin pool that runs
    out after a few thousand entries, and with the bins switched off): pre-EDL framebuffer bit-identical to the oracle on the same
    octree image, RGBA8 within 1 per channel."""
    Wd, Hd = (250, 131) if "odd_size" in variant else (40, 23) if variant == "tiny_frame" else (640, 360) if "hotspot" in variant else (1000, 562) if "grazing" in variant else (384, 256)
    if "terrain" in variant or "grazing" in variant:
        pts, box = synthetic.terrain(1_500_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
    elif "hotspot" in variant:
        pts, box = synthetic.hotspot(1_200_000, seed=11, level=4, cell=(5, 9, 6))
    else:
        pts, box = synthetic.uniform_cube(1_000_000, seed=77)
    if "inside" in variant:
        eye, target = (0.52 * box[0], 0.48 * box[1], 0.5 * box[2]), (0.9 * box[0], 0.6 * box[1], 0.45 * box[2])
    elif "grazing" in variant:
        ex, ey = 0.5 * float(box[0]), 0.3 * float(box[1])
        ground = synthetic.terrain_height(ex, ey, seed=3, box=(600.0, 400.0, 40.0))
        eye, target = (ex, ey, ground + 6.0), (ex + 20.0, ey + 200.0, ground - 4.0)
    elif "terrain" in variant:
        eye, target = (0.5 * box[0], 0.45 * box[1], 0.9 * box[2]), (0.52 * box[0], 0.6 * box[1], 0.4 * box[2])
    elif "hotspot" in variant:
        c = (np.array([5, 9, 6], dtype=np.float32) + 0.5) / 16.0
        eye, target = tuple(c + np.float32(0.09) * np.array([1.4, -1.1, 0.9], dtype=np.float32)), tuple(c)
    else:
        eye, target = (1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2])
    T = camera.lookat_transform(eye, target, Wd, Hd)
    dev = _device(ring_slots=2)
    if "small_pool" in variant:
        dev.tune("SIMLOD_DEBUG_BIN_POOL", 20_000)
    if "no_bins" in variant:
        dev.tune("SIMLOD_RASTER_SCREEN_BINS", 0)
    u = dev.uniforms(Wd, Hd, T, box)
    _ingest(dev, u, [pts[i:i + 1_000_000] for i in range(0, len(pts), 1_000_000)])
    u["useHighQualityShading"] = 1 if "hqs" in variant else 0
    u["colorByNode"] = 1 if "by_node" in variant else 0
    u["showBoundingBox"] = 1 if "boxes" in variant else 0
    u["showPoints"] = 0 if variant == "boxes_only" else 1
    u["pointSize"] = 5 if variant == "point_size_5" else 1
    if variant == "fine_lod_hqs" or "odd_size" in variant:
        u["minNodeSize"] = 8.0                                  # a node is drawn once its box spans 2 * minNodeSize pixels (render.cu:893-901)
    if variant == "tiny_frame":
        u["minNodeSize"] = 2.0
    dev.render(u)
    fb_dev, col_dev, ds = dev.framebuffer(Wd, Hd), dev.color(Wd, Hd), dev.read_stats()
    assert int(ds["dbg"]) == 0
    if "grazing" in variant:
        binned, outside = dev.samples_binned(Wd, Hd), dev.samples_outside_tiles()
        if "no_bins" in variant:
            assert binned == 0 and outside > 10_000, (binned, outside)
        elif "small_pool" in variant:
            assert 0 < binned <= 20_000 and outside > 10_000, (binned, outside)      # what the pool could not take went the global way
        else:
            assert binned > 50_000, (binned, outside)
        # the buffer's next frames (the second one still sorts or not as the first did; from the third on the first frame's finding decides)
        for _ in range(3):
            dev.render(u)
            assert np.array_equal(dev.framebuffer(Wd, Hd), fb_dev) and np.array_equal(dev.color(Wd, Hd), col_dev)
    nodes, pers, nn = host_image_of(dev)
    fb, col, st = _oracle_render(nodes, nn, u)
    assert_stats_equal(ds, st, STATS_RENDER_FIELDS, variant)
    diff = np.nonzero(fb_dev != fb)[0]
    assert len(diff) == 0, f"{variant}: {len(diff)} pixels differ, first {diff[:5]}: dev {fb_dev[diff[:5]]} oracle {fb[diff[:5]]}"
    assert int((fb != abi.CLEAR_PIXEL).sum()) > (50 if variant in ("tiny_frame", "boxes_only") else 2000), "the case must draw something"
    assert int(np.abs(col_dev.view(np.uint8).astype(np.int16) - col.view(np.uint8).astype(np.int16)).max()) <= 1


@pytest.mark.parametrize("size", [(1920, 1200), (2560, 1440)])
def test_reference_hosts_200_mb_render_buffer_above_1080p(built_libs, size):
    """the reference host allocates 200 000 000 bytes for kernel_render's buffer whatever the window's size
    (main_progressive_octree.cpp:555-556) and renders at the window's size.  The planes grow with the frame and the screen bins lie behind them:
    at 1920 x 1200 the buffer has room for part of the bin pool, at 2560 x 1440 for none of it.  The frame goes into a buffer of exactly that
    size — hipMalloc, as the host's cuMemAlloc: the library asks the runtime what the allocation holds —, must not touch a byte beyond it (the
    device faults on an unmapped page: the test would die here) and must equal the oracle's; a camera that skims the terrain, so that the bins
    are wanted."""
    import ctypes
    import torch
    Wd, Hd = size
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    pts, box = synthetic.terrain(1_500_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
    ex, ey = 0.5 * float(box[0]), 0.3 * float(box[1])
    ground = synthetic.terrain_height(ex, ey, seed=3, box=(600.0, 400.0, 40.0))
    T = camera.lookat_transform((ex, ey, ground + 6.0), (ex + 20.0, ey + 200.0, ground - 4.0), Wd, Hd)
    dev = _device(ring_slots=2, max_pixels=Wd * Hd)
    u = dev.uniforms(Wd, Hd, T, box)
    _ingest(dev, u, [pts[i:i + 1_000_000] for i in range(0, len(pts), 1_000_000)])
    HOST_BYTES = 200_000_000
    assert int(dev.L.simlod_render_buffer_bytes(Wd, Hd)) > HOST_BYTES, "the case: a frame whose full layout does not fit the host's buffer"
    raw = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(raw), HOST_BYTES) == 0
    try:
        assert hip.hipMemset(raw, 0xA5, HOST_BYTES) == 0
        uu, up = dev._u(u)
        nodes, pers, nn = host_image_of(dev)
        for hqs in (0, 1):
            uu["useHighQualityShading"] = hqs
            for _ in range(2):             # (a buffer's first frame decides about the bins; its second one runs with what the first found)
                rc = dev.L.simlod_launch_render(raw, up, dev._p(dev.nodes), dev._p(dev.colorbuffer), dev._p(dev.stats), dev._p(dev.frame_start), None, dev._stream())
                assert rc == 0
                torch.cuda.synchronize()
            fb_dev = np.zeros(Wd * Hd, dtype=np.uint64)
            assert hip.hipMemcpy(fb_dev.ctypes.data, ctypes.c_void_p(raw.value + int(dev.L.simlod_render_framebuffer_offset())), Wd * Hd * 8, 2) == 0
            ds = dev.read_stats()
            assert int(ds["dbg"]) == 0
            fb, col, st = _oracle_render(nodes, nn, uu[0])
            assert_stats_equal(ds, st, STATS_RENDER_FIELDS, str(size))
            bad = int((fb_dev != fb).sum())
            assert bad == 0, f"{size}, hqs={hqs}: {bad} pixels differ from the oracle's frame"
            assert int((fb != abi.CLEAR_PIXEL).sum()) > 20_000
    finally:
        hip.hipFree(raw)


# ---- the reference's launch surface ------------------------------------------------------------------------------------------
def test_cuda_modular_program_shaped_surface(built_libs):
    """Drive reset / construct / render exactly as main_progressive_octree.cpp does: program->kernels[name] and a
    cuLaunchCooperativeKernel-style void** argument array (:337-345, :374-382, :499-507)."""
    import torch
    from simlod_amd.runtime import Program, lib
    L = lib()
    pts, box, batch, T = case("uniform_3x40k")
    g = load_golden("uniform_3x40k")
    dev = _device(ring_slots=8)
    u = dev.uniforms(W, H, T, box)
    ub = np.ascontiguousarray(u).reshape(1)
    prog_reset = Program(["./modules/progressive_octree/reset.cu", "./modules/progressive_octree/utils.cu"], ["kernel"])
    prog_update = Program(["./modules/progressive_octree/progressive_octree_voxels.cu", "./modules/progressive_octree/utils.cu"], ["kernel_construct"])
    prog_render = Program(["./modules/progressive_octree/render.cu", "./modules/progressive_octree/utils.cu"], ["kernel_render"])
    occ = ctypes.c_int(0)
    assert L.simlod_function_max_active_blocks(prog_render.kernels["kernel_render"], 256, ctypes.byref(occ)) == 0 and occ.value >= 1

    def args(*vals):
        holders = [ctypes.c_void_p(v) if not isinstance(v, np.ndarray) else None for v in vals]
        arr = (ctypes.c_void_p * len(vals))()
        for i, v in enumerate(vals):
            arr[i] = v.ctypes.data if isinstance(v, np.ndarray) else ctypes.addressof(holders[i])
        return arr, holders

    cudaprint = 0
    a, keep = args(ub, dev.persistent.data_ptr(), dev.nodes.data_ptr(), dev.stats.data_ptr(), cudaprint, dev.num_uploaded.data_ptr(), dev.batch_sizes.data_ptr())
    assert L.simlod_launch_cooperative(prog_reset.kernels["kernel"], 1, 1, 1, 1, 1, 1, 0, None, a) == 0
    torch.cuda.synchronize()
    for b in batches_of("uniform_3x40k", pts, batch):
        dev.upload(b)
    a, keep = args(ub, dev.ring.data_ptr(), dev.momentary.data_ptr(), dev.persistent.data_ptr(), dev.nodes.data_ptr(), dev.stats.data_ptr(),
                   dev.frame_start.data_ptr(), cudaprint, dev.num_uploaded.data_ptr(), dev.batch_sizes.data_ptr())
    assert L.simlod_launch_cooperative(prog_update.kernels["kernel_construct"], 256, 1, 1, 256, 1, 1, 0, None, a) == 0
    a, keep = args(dev.render_buffer.data_ptr(), ub, dev.nodes.data_ptr(), dev.colorbuffer.data_ptr(), dev.stats.data_ptr(), dev.frame_start.data_ptr(), cudaprint)
    assert L.simlod_launch_cooperative(prog_render.kernels["kernel_render"], 256 * occ.value, 1, 1, 256, 1, 1, 0, None, a) == 0
    torch.cuda.synchronize()
    ds = dev.read_stats()
    assert_stats_equal(ds, g["build_stats"][0], STATS_BUILD_FIELDS, "surface")
    assert_stats_equal(ds, g["render_stats_plain"][0], STATS_RENDER_FIELDS, "surface")
    assert int(ds["frameID"]) == int(u["frameCounter"])


# ---- BASELINE.json full size: size-independent properties ------------------------------------------------------------------------
def test_full_size_36m_properties(built_libs):
    """configs[1] at full size (36 M points, 36 ring batches): structural invariants, conservation of points, the allocator and
    chunk-pool accounting identities, idempotent frames, and HQS/plain agreement on per-pixel depth."""
    n = 36_000_000
    pts, box = synthetic.terrain(n, seed=7)
    Wd, Hd = 1920, 1080
    T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=Wd / Hd))
    dev = _device(persistent_bytes=4 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
    u = dev.uniforms(Wd, Hd, T, box)
    _ingest(dev, u, [pts[i:i + abi.MAX_BATCH_SIZE] for i in range(0, n, abi.MAX_BATCH_SIZE)])
    s = dev.read_stats()
    assert int(s["dbg"]) == 0 and int(s["numPointsProcessed"]) == n and int(s["numPoints"]) == n and int(s["batchletIndex"]) == 36
    nodes, pers, nn = host_image_of(dev)
    # the same 36 batches through the CPU oracle (a few seconds): every Stats counter incl. allocator offset and chunk pool, and every
    # node — topology, counters, point multisets, occupancy bitsets, voxel positions — must be identical at full size too
    ref = oracle.HostOctree("port", persistent_bytes=4 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
    ref.reset(u)
    ref.add_points(u, pts)
    assert ref.last_error() == 0
    assert_stats_equal(s, ref.stats[0], STATS_BUILD_FIELDS, "36M")
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "36M")
    del ref
    inv = oracle.check_invariants(nodes, nn)
    assert inv["points"] == n and inv["voxels"] >= int(s["numVoxels"])        # Stats.numVoxels counts inner nodes only
    assert int(s["numNodes"]) == nn == 1 + 8 * int(s["numInner"]) and int(s["numLeaves"]) == nn - int(s["numInner"])
    assert inv["point_chunks"] == int(s["numChunksPoints"]) == int(s["numAllocatedChunks"]) <= int(s["chunkPoolSize"])
    # every persistent byte is a grid, a pooled point chunk or a voxel chunk (utils.h.cu:190 rounding)
    expect = 16 + inv["grids"] * abi.alloc_round(abi.GRID_BYTES) + (int(s["chunkPoolSize"]) + inv["voxel_chunks"]) * abi.alloc_round(abi.CHUNK_BYTES)
    assert int(s["allocatedBytes_persistent"]) == expect
    # multiset of stored points == multiset of input points (order-independent 128-bit hash, same mixer as oracle_dump)
    d = oracle.dump_image(nodes, nn)
    hs, hx = points_multiset_hash(pts)
    with np.errstate(over="ignore"):
        assert hs == np.uint64(d["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(d["pointsXor"])
    # frames
    dev.render(u); f1 = dev.framebuffer(Wd, Hd); r1 = dev.read_stats()
    dev.render(u); f2 = dev.framebuffer(Wd, Hd)
    assert np.array_equal(f1, f2)
    fo, _, so = _oracle_render(nodes, nn, u)
    assert np.array_equal(f1, fo), f"{int((f1 != fo).sum())} of {Wd * Hd} pixels differ from the oracle at 1080p"
    assert_stats_equal(r1, so, STATS_RENDER_FIELDS, "1080p")
    u["useHighQualityShading"] = 1
    dev.render(u); fh = dev.framebuffer(Wd, Hd)
    assert np.array_equal(fh >> np.uint64(32), f1 >> np.uint64(32)), "HQS resolves to the same nearest depth per pixel as the 64-bit min"
    fho, _, _ = _oracle_render(nodes, nn, u)
    assert np.array_equal(fh, fho), f"{int((fh != fho).sum())} HQS pixels differ from the oracle at 1080p (averaged colours incl. the >64-samples-per-pixel path)"
    # "Morro Bay - close" (main_progressive_octree.cpp:1323-1328, as bench.py scales it): 94 m above the surface, leaves of the deepest levels
    # on screen next to coarse voxel nodes at the horizon
    cx, cy = 2750.218 * float(box[0]) / 6000.0, 974.775 * float(box[1]) / 4000.0
    T_close = camera.world_view_proj(camera.orbit_view(-11.270, -0.225, 93.982, (cx, cy, synthetic.terrain_height(cx, cy, seed=7, box=tuple(float(v) for v in box)))),
                                     camera.perspective(aspect=Wd / Hd))
    for hqs in (0, 1):
        uc = dev.uniforms(Wd, Hd, T_close, box, hqs=bool(hqs))
        dev.render(uc); fc, rc = dev.framebuffer(Wd, Hd), dev.read_stats()
        fco, _, sco = _oracle_render(nodes, nn, uc)
        assert int(rc["numVisibleNodes"]) > 100
        assert np.array_equal(fc, fco), f"close preset, hqs={hqs}: {int((fc != fco).sum())} of {Wd * Hd} pixels differ from the oracle"
        assert_stats_equal(rc, sco, STATS_RENDER_FIELDS, f"close preset hqs={hqs}")


# ---- the reference host's launch sequence in C++ (harness/simlod_headless.cpp on shim/cuda.h) ------------------------------------
def test_headless_cpp_host_replay(built_libs, tmp_path):
    """A .simlod file goes through the C++ replay of main_progressive_octree.cpp's init / reset / uploader / frame loop; the octree
    it reports must be the one the oracle builds from the same file."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "simlod_headless")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness")])
    pts, box = synthetic.terrain(2_500_000, seed=5, box=(1500.0, 1000.0, 100.0), tile=125.0)
    path = str(tmp_path / "terrain.simlod")
    synthetic.write_simlod(path, pts, box)
    ppm = str(tmp_path / "frame.ppm")
    out = subprocess.run([exe, path, ppm, "640", "360"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"numNodes (\d+) numInner (\d+) numLeaves (\d+) numPoints (\d+) numVoxels (\d+) persistentBytes (\d+) chunkPoolSize (\d+) dbg (\d+)", out.stdout)
    assert m, out.stdout
    got = [int(v) for v in m.groups()]
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = uniforms_for(box, T, persistent=8 << 30)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts)
    s = ref.stats[0]
    assert got == [int(s[k]) for k in ("numNodes", "numInner", "numLeaves", "numPoints", "numVoxels", "allocatedBytes_persistent", "chunkPoolSize")] + [0]
    assert os.path.getsize(ppm) == 15 + 640 * 360 * 3
    img = np.fromfile(ppm, dtype=np.uint8, offset=15).reshape(360, 640, 3)
    assert (img != np.array([0x11, 0x22, 0x33], dtype=np.uint8)).any(axis=2).sum() > 5000      # something other than background was drawn


def test_headless_cpp_host_streams_a_las_file(built_libs, tmp_path):
    """The same C++ host fed a LAS 1.4 format-7 file (world coordinates far from the origin): the loader moves raw records, the
    device decodes them into the ring (simlod_decode_las).  The octree must be the one the oracle builds from the oracle-decoded
    points."""
    import re
    import subprocess
    from simlod_amd import lasio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "simlod_headless")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness")])
    pts0, box = synthetic.terrain(2_200_000, seed=8, box=(1500.0, 1000.0, 100.0), tile=125.0)
    path = str(tmp_path / "terrain.las")
    h = lasio.points_to_las(path, pts0, box, fmt=7, scale=0.001, world_min=(694000.0, 3915000.0, -3.0), version=(1, 4))
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    m = re.search(r"numNodes (\d+) numInner (\d+) numLeaves (\d+) numPoints (\d+) numVoxels (\d+) persistentBytes (\d+) chunkPoolSize (\d+) dbg (\d+)", out.stdout)
    assert m, out.stdout
    got = [int(v) for v in m.groups()]
    tr = tuple(-v for v in h.min)
    pts = oracle.decode_las_port(lasio.read_records(path, h, 0, h.numPoints), h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tr))
    hbox = (np.array(h.max) - np.array(h.min)).astype(np.float32)      # what the host derives from the header
    T = camera.lookat_transform((1.8 * hbox[0], -1.2 * hbox[1], 1.4 * max(hbox)), (0.5 * hbox[0], 0.5 * hbox[1], 0.3 * hbox[2]), W, H)
    u = uniforms_for(hbox, T, persistent=8 << 30)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts)
    s = ref.stats[0]
    assert got == [int(s[k]) for k in ("numNodes", "numInner", "numLeaves", "numPoints", "numVoxels", "allocatedBytes_persistent", "chunkPoolSize")] + [0]


# ---- loader row: LAS records -> Points on the device -----------------------------------------------------
def _gpu_decode(raw, n, bpp, fmt, scale, offset):
    import torch
    from simlod_amd.runtime import lib
    L = lib()
    d_raw = torch.from_numpy(np.array(raw, dtype=np.uint8)).to("cuda:0") if raw.size else torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    d_out = torch.full((max(n, 1) * 16 + 64,), 0x5A, dtype=torch.uint8, device="cuda:0")       # guard bytes behind the output
    rc = L.simlod_decode_las(ctypes.c_void_p(d_raw.data_ptr()), ctypes.c_uint64(n), ctypes.c_uint32(bpp), ctypes.c_uint32(fmt),
                             (ctypes.c_double * 3)(*scale), (ctypes.c_double * 3)(*offset), ctypes.c_void_p(d_out.data_ptr()),
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    host = d_out.cpu().numpy()
    assert np.all(host[n * 16:] == 0x5A), "decode wrote past its output"
    return rc, host[: n * 16].view(abi.point_dtype)


def test_las_decode_matches_reference_fixture_bit_exact(built_libs, tmp_path):
    from simlod_amd import lasio
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "las_decode.npz"))
    for ci in range(len([k for k in G.files if k.endswith("_fmt")])):
        fmt, vmaj, vmin, bpp, first, count = (int(v) for v in G[f"c{ci}_fmt"])
        path = str(tmp_path / f"c{ci}.las")
        G[f"c{ci}_file"].tofile(path)
        h = lasio.load_header(path)
        raw = lasio.read_records(path, h, first, count)
        rc, got = _gpu_decode(raw, count, bpp, fmt, h.scale, lasio.decode_offset(h, tuple(-m for m in h.min)))
        assert rc == 0
        want = G[f"c{ci}_points"]
        for k in "xyz":
            assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), (ci, k)
        if fmt in lasio.RGB_OFFSET:
            assert np.array_equal(got["color"] & 0xffffff, want["color"] & 0xffffff), ci
        port = oracle.decode_las_port(raw, bpp, fmt, h.scale, lasio.decode_offset(h, tuple(-m for m in h.min)))
        assert np.array_equal(got.view(np.uint8), port.view(np.uint8)), ci          # incl. alpha = 255 / colourless = 0


@pytest.mark.parametrize("n", [0, 1, 63, 255, 256, 257, 100_003, 1_000_000])
@pytest.mark.parametrize("fmt,bpp", [(2, 26), (3, 34), (7, 37), (1, 28), (10, 255)])
def test_las_decode_ragged_sizes_and_strides_match_oracle(built_libs, n, fmt, bpp):
    from simlod_amd import lasio
    rs = np.random.RandomState(n % 9973 + fmt)
    xyz = rs.randint(-2 ** 31, 2 ** 31 - 1, size=(n, 3), dtype=np.int64).astype(np.int32)
    rgb = rs.randint(0, 65536, size=(n, 3)).astype(np.uint16)
    raw = lasio.las_records(xyz, rgb, fmt, bytes_per_point=bpp, seed=n + 1).reshape(-1)
    scale, offset = (1e-3, 2.5e-3, 1e-7), (-694000.123, 3915000.456, -3.0)
    rc, got = _gpu_decode(raw, n, bpp, fmt, scale, offset)
    assert rc == 0
    assert np.array_equal(got.view(np.uint8), oracle.decode_las_port(raw, bpp, fmt, scale, offset).view(np.uint8))


def test_las_decode_rejects_what_it_cannot_read(built_libs):
    raw = np.zeros(26 * 4, dtype=np.uint8)
    assert _gpu_decode(raw, 4, 11, 0, (1, 1, 1), (0, 0, 0))[0] != 0          # record shorter than XYZ
    assert _gpu_decode(raw, 4, 24, 2, (1, 1, 1), (0, 0, 0))[0] != 0          # format 2 needs RGB at 20..25
    assert _gpu_decode(raw, 4, 256, 0, (1, 1, 1), (0, 0, 0))[0] != 0         # stride beyond the staging limit


def test_las_file_to_octree_equals_oracle_on_decoded_points(built_libs, tmp_path):
    """End to end: synthetic LAS file -> raw bytes -> device decode into the ring -> kernel_construct, against the oracle
    fed with the oracle-decoded points (world coordinates far from the origin, translated by -min as the reference does)."""
    from simlod_amd import lasio
    pts0, box = synthetic.terrain(300_000, seed=3, box=(600.0, 400.0, 40.0), tile=50.0)
    path = str(tmp_path / "t.las")
    h = lasio.points_to_las(path, pts0, box, fmt=3, scale=0.001, world_min=(694000.0, 3915000.0, -3.0), version=(1, 4))
    tr = tuple(-m for m in h.min)
    pts = oracle.decode_las_port(lasio.read_records(path, h, 0, h.numPoints), h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tr))
    assert np.abs(pts["x"] - pts0["x"]).max() < 2e-3
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=8)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    dev.add_las(u, path, batch=100_000)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts, 100_000)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "las")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "las")


# ---- frames in parts / composed across ranks -------------------------------------------------------------------
@pytest.mark.parametrize("variant", ["plain", "hqs", "hqs_boxes", "plain_boxes", "hqs_ps2"])
def test_render_parts_equal_the_whole_frame_and_two_emulated_ranks_compose_exactly(built_libs, variant):
    """(1) the four parts of simlod_launch_render_part, run back to back, are simlod_launch_render; (2) two sub-octrees that split the
    points by top-level octant (distributed.owner_of), rendered in parts with the reductions of distributed.render_frame done by hand
    on one GPU, give the frame of the single octree that holds every point — plain and HQS, bit for bit."""
    import torch
    from simlod_amd import distributed
    pts, box = synthetic.uniform_cube(800_000, seed=21)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    kw = dict(hqs="hqs" in variant)
    whole = _device(ring_slots=2)
    u = whole.uniforms(W, H, T, box, **kw)
    if "boxes" in variant:
        u["showBoundingBox"] = 1
    if "ps2" in variant:
        u["pointSize"] = 2
    _ingest(whole, u, [pts])
    whole.render(u)
    want_fb, want_color = whole.framebuffer(W, H), whole.color(W, H)
    for part in range(4):
        whole.render_part(u, part)
    assert np.array_equal(whole.framebuffer(W, H), want_fb) and np.array_equal(whole.color(W, H), want_color)

    own = distributed.owner_of(pts, box, 2)
    ranks = []
    for r in range(2):
        d = _device(ring_slots=2)
        _ingest(d, u, [pts[own == r]])
        ranks.append(d)
    for d in ranks:
        d.render_part(u, 0)
    if kw["hqs"]:
        m = torch.minimum(ranks[0].depth_plane(), ranks[1].depth_plane())
        for d in ranks:
            d.depth_plane().copy_(m)
            d.render_part(u, 1)
        ssum = ranks[0].sum_planes() + ranks[1].sum_planes()
        for d in ranks:
            d.sum_planes().copy_(ssum)
            d.render_part(u, 2)
    m = torch.minimum(ranks[0].framebuffer_words(), ranks[1].framebuffer_words())
    for d in ranks:
        d.framebuffer_words().copy_(m)
        d.render_part(u, 3)
    # The two rank octrees were built in runs of their own, so their voxel colours (first writer wins) need not be
    # those of `whole`: depth must match the single-octree frame everywhere, and the exact frame is what the CPU oracle composes
    # from the SAME two octree images with the same reductions.
    images = [host_image_of(d) for d in ranks]
    lib_o = oracle.port_lib()
    p = lambda arr: ctypes.c_void_p(arr.ctypes.data)
    uu = np.ascontiguousarray(u).reshape(1)
    state = []
    for nodes, pers, nn in images:
        st = np.zeros(1, dtype=abi.stats_dtype); st["numNodes"] = nn
        state.append(dict(nodes=nodes, stats=st, fb=np.zeros(W * H, dtype=np.uint64), color=np.zeros(W * H, dtype=np.uint32),
                          vis=np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype), depth=np.zeros(W * H, dtype=np.uint32), sums=np.zeros(W * H * 4, dtype=np.uint32)))
    def part(k):
        for s_ in state:
            lib_o.oracle_render_part(None, p(uu), p(s_["nodes"]), p(s_["stats"]), p(s_["fb"]), p(s_["color"]), p(s_["vis"]), 1, k, p(s_["depth"]), p(s_["sums"]))
    part(0)
    if kw["hqs"]:
        dm = np.minimum(state[0]["depth"], state[1]["depth"])       # positive float bits: unsigned MIN
        for s_ in state: s_["depth"][:] = dm
        part(1)
        sm = state[0]["sums"] + state[1]["sums"]
        for s_ in state: s_["sums"][:] = sm
        part(2)
    fm = np.minimum(state[0]["fb"], state[1]["fb"])
    for s_ in state: s_["fb"][:] = fm
    part(3)
    for d, s_ in zip(ranks, state):
        got = d.framebuffer(W, H)
        assert np.array_equal(got >> np.uint64(32), want_fb >> np.uint64(32)), f"{variant}: depth differs from the single-octree frame"
        bad = int((got != s_["fb"]).sum())
        assert bad == 0, f"{variant}: {bad} pixels differ from the frame the oracle composes from the same two octrees"
        c_got, c_want = d.color(W, H).view(np.uint8).astype(np.int16), s_["color"].view(np.uint8).astype(np.int16)
        assert np.abs(c_got - c_want).max() <= 1                                         # EDL: +-1 per channel
    assert sum(int(d.read_stats()["numVisiblePoints"]) for d in ranks) == int(whole.read_stats()["numVisiblePoints"])


# ---- BASELINE configs 3 and 5 at a size the oracle still finishes in seconds; the ring; ingest granularity ---------------------------
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_ring_wraps_around_more_than_twice(built_libs, overlap, monkeypatch):
    """Config 3's mechanism: batches stream through the 50-slot ring (slot = batchletIndex % 50, voxels.cu:883-925) with the host's
    back-pressure rule.  130 batches of 50 000 points wrap the ring 2.6 times; the octree must be the oracle's after the same 130 batches.
    With the voxel tail of a batch on the library's side stream (the default of the exact chain) and on the caller's stream."""
    monkeypatch.setenv("SIMLOD_OVERLAP_TAIL", overlap)
    pts, box = synthetic.terrain(6_500_000, seed=21, box=(2400.0, 1600.0, 160.0), tile=100.0)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=abi.BATCH_STREAM_SIZE)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + 50_000] for i in range(0, len(pts), 50_000)])
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0 and int(ds["batchletIndex"]) == 130 and int(ds["numPointsProcessed"]) == len(pts)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
    ref.reset(u)
    ref.add_points(u, pts, 50_000)
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "ring")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "ring")
    oracle.check_invariants(nodes, nn)


def test_config5_hotspot_20m_octree_and_frames_match_oracle_with_and_without_lds_tiles(built_libs):
    """BASELINE config 5 at 20 M points: every point in one level-6 cell (the first batch cascades six levels down and keeps splitting),
    camera on the cell.  Octree == oracle; plain and HQS frames bit-exact against the oracle's rasteriser on the same image, with the
    LDS-tile path of the rasteriser on and off."""
    n = 20_000_000
    pts, box = synthetic.hotspot(n, seed=11)
    Wd, Hd = 1920, 1080
    cell = np.array([21, 40, 13], dtype=np.float64) / 64 + 1 / 128
    dist = (1 / 64) * (Hd / 128.0) / (2 * np.tan(np.radians(30)))
    T = camera.lookat_transform(cell + np.array([0.6, -0.7, 0.4]) / np.linalg.norm([0.6, -0.7, 0.4]) * dist, cell, Wd, Hd)
    dev = _device(persistent_bytes=4 << 30, ring_slots=20)
    u = dev.uniforms(Wd, Hd, T, box, min_node_size=8.0)
    _ingest(dev, u, [pts[i:i + abi.MAX_BATCH_SIZE] for i in range(0, n, abi.MAX_BATCH_SIZE)])
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0 and int(ds["numPoints"]) == n
    ref = oracle.HostOctree("port", persistent_bytes=4 << 30, ring_slots=20)
    ref.reset(u)
    ref.add_points(u, pts)
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "config 5")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "config 5")
    del ref
    want = {}
    try:
        for tiles in ("1", "0"):
            dev.tune("SIMLOD_RASTER_LDS_TILES", int(tiles))
            for hqs in (0, 1):
                u["useHighQualityShading"] = hqs
                dev.render(u)
                fb, st = dev.framebuffer(Wd, Hd), dev.read_stats()
                if hqs not in want:
                    want[hqs] = _oracle_render(nodes, nn, u)
                fo, _, so = want[hqs]
                assert_stats_equal(st, so, STATS_RENDER_FIELDS, f"config 5 tiles={tiles} hqs={hqs}")
                assert np.array_equal(fb, fo), f"tiles={tiles} hqs={hqs}: {int((fb != fo).sum())} pixels differ from the oracle"
    finally:
        dev.tune("SIMLOD_RASTER_LDS_TILES", None)


def test_headless_cpp_host_replay_octree_dump_equals_oracle(built_libs, tmp_path):
    """The C++ replay of the reference host writes the octree image it built (SIMLOD_HARNESS_DUMP); node by node it must be the
    oracle's octree of the same file — not just the same seven counters."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "simlod_headless")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "harness")])
    pts, box = synthetic.terrain(3_300_000, seed=15, box=(1500.0, 1000.0, 100.0), tile=125.0)
    path, dump = str(tmp_path / "terrain.simlod"), str(tmp_path / "octree.bin")
    synthetic.write_simlod(path, pts, box)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300, env=dict(os.environ, SIMLOD_HARNESS_DUMP=dump))
    assert out.returncode == 0, out.stdout + out.stderr
    head = np.fromfile(dump, dtype=np.uint64, count=4)
    nn, used, nodes_base, pers_base = (int(v) for v in head)
    nodes = np.fromfile(dump, dtype=abi.node_dtype, count=nn, offset=32)
    pers = np.fromfile(dump, dtype=np.uint8, count=used, offset=32 + nn * 152)
    oracle.rebase_image(nodes, nn, pers, nodes_base, pers_base)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = uniforms_for(box, T, persistent=8 << 30)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "harness dump")
    oracle.check_invariants(nodes, nn)


@pytest.mark.parametrize("host", ["ref_host_replay", "ref_uploader_replay"])
@pytest.mark.parametrize("kind", ["simlod", "las"])
def test_reference_host_functions_drive_the_library(built_libs, tmp_path, kind, host):
    """harness/_ref/ref_host_replay = the headless host with the REFERENCE'S OWN resetCUDA / updateOctree / renderCUDA / initCudaProgram
    (cut out of main_progressive_octree.cpp at build time where the reference exists; the binary travels with the snapshot).  Uploader on
    its own thread and stream while kernel_construct runs.  ref_uploader_replay: that uploader, the pinned-memory pool
    and reset() are the reference's own text as well (main.cpp:141-222, 775-809, 963-1063).  The octree either leaves must be the
    oracle's, node by node."""
    import subprocess
    from simlod_amd import lasio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "harness", "_ref", host)
    if not os.path.exists(exe):
        pytest.skip(f"harness/_ref/{host} was not built (no reference checkout where the snapshot was made)")
    pts0, box = synthetic.terrain(2_700_000, seed=19, box=(1500.0, 1000.0, 100.0), tile=125.0)
    dump = str(tmp_path / "octree.bin")
    if kind == "simlod":
        path = str(tmp_path / "terrain.simlod")
        synthetic.write_simlod(path, pts0, box)
        pts, hbox = pts0, box
    else:
        path = str(tmp_path / "terrain.las")
        h = lasio.points_to_las(path, pts0, box, fmt=7, scale=0.001, world_min=(694000.0, 3915000.0, -3.0), version=(1, 4))
        tr = tuple(-v for v in h.min)
        pts = oracle.decode_las_port(lasio.read_records(path, h, 0, h.numPoints), h.bytesPerPoint, h.format, h.scale, lasio.decode_offset(h, tr))
        hbox = (np.array(h.max) - np.array(h.min)).astype(np.float32)
    out = subprocess.run([exe, path, str(tmp_path / "f.ppm"), "640", "360"], capture_output=True, text=True, timeout=600, env=dict(os.environ, SIMLOD_HARNESS_DUMP=dump))
    assert out.returncode == 0, out.stdout + out.stderr
    head = np.fromfile(dump, dtype=np.uint64, count=4)
    nn, used, nodes_base, pers_base = (int(v) for v in head)
    nodes = np.fromfile(dump, dtype=abi.node_dtype, count=nn, offset=32)
    pers = np.fromfile(dump, dtype=np.uint8, count=used, offset=32 + nn * 152)
    oracle.rebase_image(nodes, nn, pers, nodes_base, pers_base)
    T = camera.lookat_transform((1.8 * hbox[0], -1.2 * hbox[1], 1.4 * max(hbox)), (0.5 * hbox[0], 0.5 * hbox[1], 0.3 * hbox[2]), W, H)
    u = uniforms_for(hbox, T, persistent=8 << 30)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "reference host functions")
    img = np.fromfile(str(tmp_path / "f.ppm"), dtype=np.uint8, offset=15).reshape(360, 640, 3)
    drawn = int((img != np.array([0x11, 0x22, 0x33], dtype=np.uint8)).any(axis=2).sum())
    assert drawn > (5000 if kind == "simlod" else 1500), f"only {drawn} pixels drawn; harness said: {out.stdout[-600:]}"


def test_points_exactly_on_the_max_faces_give_the_reference_voxels(built_libs):
    """A coordinate equal to boxMax quantises to 2^20; the reference's descent looks at bits 19..0 and files the point (and the voxels
    it creates) under node coordinate 0 of that axis.  Only such points here, so they are the ones that win the cells."""
    rs = np.random.RandomState(8)
    base, box = synthetic.uniform_cube(120_000, seed=3)
    faces = base[:60_000].copy()
    for k in "xyz":
        on = rs.rand(len(faces)) < 0.5
        faces[k][on] = np.float32(1.0)
    pts = np.concatenate([faces, base[60_000:]])
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    dev = _device(ring_slots=4)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + 40_000] for i in range(0, len(pts), 40_000)])
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=4)
    ref.reset(u)
    ref.add_points(u, pts, 40_000)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "faces")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "faces")


def test_node_capacity_beyond_the_packed_index_width_is_rejected(built_libs):
    from simlod_amd.runtime import lib
    L = lib()
    try:
        assert L.simlod_set_node_capacity((1 << 19) + 1) != 0 and L.simlod_set_node_capacity(8) != 0
        assert L.simlod_set_node_capacity(1 << 19) == 0
    finally:
        L.simlod_set_node_capacity(263_157)


def test_forced_barrier_timeout_aborts_the_batch_and_stays_fatal_until_reset(built_libs):
    """The split cascade's grid barrier giving up (here: forced) must not let the rest of the chain run on a half-built state: the batch
    is not counted, Stats.dbg carries the fatal bit, later launches refuse to touch the octree, and a reset brings everything back."""
    from simlod_amd.runtime import SimlodError
    pts, box = synthetic.uniform_cube(300_000, seed=12)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    dev = _device(ring_slots=4)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    dev.upload(pts[:40_000])
    dev.drain(u)                                       # 40 000 points: no split yet
    dev.tune("SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT", 1)
    try:
        dev.upload(pts[40_000:])                       # crosses the limit: k_expand runs and its barrier "gives up"
        with pytest.raises(SimlodError):
            dev.drain(u)
    finally:
        dev.tune("SIMLOD_DEBUG_FORCE_BARRIER_TIMEOUT", None)
    s = dev.read_stats()
    assert int(s["dbg"]) & 0x40 and int(s["batchletIndex"]) == 1 and int(s["numNodes"]) == 1
    with pytest.raises(SimlodError):
        dev.drain(u)                                   # still refused: the bit is sticky
    assert int(dev.read_stats()["batchletIndex"]) == 1
    _ingest(dev, u, [pts[:40_000], pts[40_000:]])      # reset + the same two batches
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=4)
    ref.reset(u)
    ref.upload(pts[:40_000]); ref.construct(u); ref.upload(pts[40_000:]); ref.construct(u)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "after reset")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "after reset")


# what does not depend on the ingest granularity: topology, per-node counters and multisets, bitsets, voxel positions and counts
GRANULARITY_FREE_FIELDS = ["key", "level", "X", "Y", "Z", "isLeaf", "childMask", "numPoints", "numVoxels", "numVoxelsStored", "hasGrid",
                           "gridPopcount", "gridHash", "pointsSum", "pointsXor", "voxelPosSum", "voxelPosXor", "pointChunks", "voxelChunks", "name"]
GRANULARITY_FREE_STATS = ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksPoints", "numChunksVoxels",
                          "batchletIndex", "numPointsProcessed", "numAllocatedChunks", "memCapacityReached"]


@pytest.mark.parametrize("kind,n,batch", [("terrain", 7_300_000, abi.MAX_BATCH_SIZE), ("terrain", 7_300_000, 300_000), ("hotspot", 5_000_000, abi.MAX_BATCH_SIZE),
                                          ("uniform", 3_000_000, abi.MAX_BATCH_SIZE)])
def test_coalesced_ingest_builds_the_same_octree_content(built_libs, kind, n, batch):
    """Opt-in coalesced mode (simlod_set_ingest_mode(1)): all pending batches of a launch as ONE batch.  Everything that does not depend
    on the batch granularity must equal the oracle's batch-by-batch octree; the frame of that image must equal the oracle's rasteriser."""
    from simlod_amd.runtime import lib
    pts, box = {"uniform": lambda: synthetic.uniform_cube(n, seed=31), "terrain": lambda: synthetic.terrain(n, seed=9, box=(3000.0, 2000.0, 200.0), tile=125.0),
                "hotspot": lambda: synthetic.hotspot(n, seed=13, level=4, cell=(5, 9, 6))}[kind]()
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    try:
        # (25 batches of 300 000: a launch takes 20 of them as two groups of 10, the second group's front half beside the first's back half)
        dev = _device(ring_slots=abi.BATCH_STREAM_SIZE, coalesce=True, momentary_bytes=400_000_000)
        u = dev.uniforms(W, H, T, box)
        _ingest(dev, u, [pts[i:i + batch] for i in range(0, n, batch)])
        ds = dev.read_stats()
        assert int(ds["dbg"]) == 0
        ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
        ref.reset(u)
        ref.add_points(u, pts, batch)
        assert_stats_equal(ds, ref.stats[0], GRANULARITY_FREE_STATS, kind)
        nodes, pers, nn = host_image_of(dev)
        got, want = oracle.dump_image(nodes, nn), ref.dump()
        assert len(got) == len(want)
        for f in GRANULARITY_FREE_FIELDS:
            assert np.array_equal(got[f], want[f]), f
        oracle.check_invariants(nodes, nn)
        assert voxel_colors_are_member(nodes, nn, pts, box) == int(nodes["numVoxelsStored"][:nn].sum())
        u["useHighQualityShading"] = 1
        dev.render(u)
        fo, _, so = _oracle_render(nodes, nn, u)
        assert np.array_equal(dev.framebuffer(W, H), fo)
        assert_stats_equal(dev.read_stats(), so, STATS_RENDER_FIELDS, kind)
    finally:
        pass                                  # (the ingest mode belongs to the octree's context: nothing process-wide to restore)


# ---- BASELINE config 4: tiles generated on the device, one global cube -------------------------------------------------------------------
def test_device_generated_tiles_are_deterministic_seamless_and_ingest_like_the_oracle(built_libs):
    """simlod_generate_terrain: any index range of the tiled-terrain stream, on the device.  Same (seed, index) -> same point whichever
    call produced it; every point inside its tile; the octree built from generated points equals the oracle's on the same points."""
    import torch
    dev = _device(ring_slots=8)
    ppt, tiles_x, ext = 700_000, 2, (600.0, 400.0, 40.0)
    n = 4 * ppt                                            # four tiles, 2 x 2
    whole = torch.empty(n * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(whole, 0, ppt, 7, tiles_x, ext)
    part = torch.empty(900_000 * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(part, 1_000_000, ppt, 7, tiles_x, ext)                  # a range that straddles the tile 1 / tile 2 boundary
    torch.cuda.synchronize()
    pts = whole.cpu().numpy().view(abi.point_dtype)
    assert np.array_equal(part.cpu().numpy(), whole[1_000_000 * 16: 1_900_000 * 16].cpu().numpy())
    other = torch.empty(1000 * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(other, 0, ppt, 8, tiles_x, ext)
    assert not np.array_equal(other.cpu().numpy(), whole[:16000].cpu().numpy()), "the seed must matter"
    tile = np.arange(n) // ppt
    assert (pts["x"] >= (tile % 2) * ext[0]).all() and (pts["x"] <= (tile % 2 + 1) * ext[0]).all()
    assert (pts["y"] >= (tile // 2) * ext[1]).all() and (pts["y"] <= (tile // 2 + 1) * ext[1]).all()
    assert (pts["z"] >= 0).all() and (pts["z"] < ext[2]).all() and float(pts["z"].std()) > 1.0
    assert ((pts["color"] >> 24) == 255).all()
    box = np.array([2 * ext[0], 2 * ext[1], ext[2]], dtype=np.float32)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    for i in range(0, n, abi.MAX_BATCH_SIZE):                # device to ring, no host round trip
        dev.upload(whole[i * 16: min(n, i + abi.MAX_BATCH_SIZE) * 16].view(-1, 16))
    dev.drain(u)
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    ref.reset(u)
    ref.add_points(u, pts)
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, "generated tiles")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "generated tiles")


# ---- row f-4: voxel colour filtering (colorfilter.cu) -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,n", [("uniform", 1_000_000), ("terrain", 3_000_000)])
def test_colorfilter_equals_the_reference_kernel_on_the_same_octree(built_libs, kind, n):
    """simlod_launch_colorfilter against colorfilter.cu itself (oracle/_ref/libref_filter.so: the reference's file compiled in place
    as host code): same octree image in, the voxels of every inner node must come out with the same positions AND the same averaged
    colours (as multisets: the order inside a node's voxel list follows the order of first hits, scheduling dependent in the reference
    too), Node.isFiltered must agree, points must be untouched."""
    if not oracle.have_ref_filter():
        pytest.skip("oracle/_ref/libref_filter.so was not built (no reference checkout where the snapshot was made)")
    pts, box = synthetic.uniform_cube(n, seed=41) if kind == "uniform" else synthetic.terrain(n, seed=6, box=(1500.0, 1000.0, 100.0), tile=125.0)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=4)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + abi.MAX_BATCH_SIZE] for i in range(0, n, abi.MAX_BATCH_SIZE)])
    nodes_a, pers_a, nn = host_image_of(dev)                      # the image as built ...
    oracle.ref_colorfilter(nodes_a, nn, u)                        # ... filtered by the reference's own kernel on the host
    dev.colorfilter(u)                                            # ... and by the HIP kernels on the device
    nodes_b, pers_b, nn_b = host_image_of(dev)
    assert nn == nn_b
    assert np.array_equal(nodes_a["isFiltered"][:nn], nodes_b["isFiltered"][:nn])
    inner = 0
    for i in range(nn):
        nv = int(nodes_a[i]["numVoxelsStored"])
        assert nv == int(nodes_b[i]["numVoxelsStored"]) and int(nodes_a[i]["numPoints"]) == int(nodes_b[i]["numPoints"])
        if nv == 0:
            continue
        va = oracle.gather_samples(int(nodes_a[i]["voxelChunks"]), nv)
        vb = oracle.gather_samples(int(nodes_b[i]["voxelChunks"]), nv)
        ka, kb = np.sort(va.view(np.dtype((np.void, 16))).reshape(-1)), np.sort(vb.view(np.dtype((np.void, 16))).reshape(-1))
        assert np.array_equal(ka, kb), f"node {i} (level {int(nodes_a[i]['level'])}): filtered voxels differ from the reference kernel's"
        inner += 1
    assert inner >= (1 if kind == "uniform" else 20)
    d = oracle.dump_image(nodes_b, nn)
    hs, hx = points_multiset_hash(pts)
    with np.errstate(over="ignore"):
        assert hs == np.uint64(d["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(d["pointsXor"]), "the filter must not touch the points"
    dev.render(u)                                                 # and the octree is still drawable
    assert int((dev.framebuffer(W, H) != abi.CLEAR_PIXEL).sum()) > 1000


# ---- per-octree contexts (include/simlod_hip.h simlod_context_*) -----------------------------------------------------------------------------
def test_two_octrees_with_their_own_contexts_build_side_by_side(built_libs):
    """Mode, node capacity, knobs, the second stream and the table registry belong to a context that a launch finds through its node
    array.  Two octrees in one process — exact with the two-stream pipeline and 263 157 nodes; coalesced on one stream with room for
    60 000 nodes — take their batches alternately, launch by launch.  Each must end as its own oracle octree (exact: every counter),
    and each frame must be the oracle's (the rasteriser reads each octree through ITS builder's chunk table)."""
    from simlod_amd.runtime import SimlodError
    pa, box_a = synthetic.terrain(3_300_000, seed=3, box=(3000.0, 2000.0, 200.0), tile=125.0)
    pb, box_b = synthetic.uniform_cube(2_100_000, seed=4)
    a = _device(ring_slots=abi.BATCH_STREAM_SIZE)
    b = _device(ring_slots=abi.BATCH_STREAM_SIZE, coalesce=True, momentary_bytes=400_000_000, max_nodes=60_000)
    b.tune("SIMLOD_OVERLAP_TAIL", 0)
    Ta = camera.lookat_transform((1.8 * box_a[0], -1.2 * box_a[1], 1.4 * max(box_a)), (0.5 * box_a[0], 0.5 * box_a[1], 0.3 * box_a[2]), W, H)
    Tb = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), W, H)
    ua, ub = a.uniforms(W, H, Ta, box_a), b.uniforms(W, H, Tb, box_b)
    a.reset(ua); b.reset(ub)
    step = 300_000
    for i in range(0, max(len(pa), len(pb)), step):
        for dev, u, pts in ((a, ua, pa), (b, ub, pb)):
            if i < len(pts):
                if dev.uploaded_host - dev.processed() >= dev.ring_slots:
                    dev.drain(u)
                dev.upload(pts[i:i + step])
                dev.construct(u)                                   # one launch each, in turn: nothing waits in between
    a.drain(ua); b.drain(ub)
    sa, sb = a.read_stats(), b.read_stats()
    assert int(sa["dbg"]) == 0 and int(sb["dbg"]) == 0
    ra = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE); ra.reset(ua); ra.add_points(ua, pa, step)
    rb = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE); rb.reset(ub); rb.add_points(ub, pb, step)
    na, keep_a, nna = host_image_of(a)                # (the node records point into the persistent copies: they must stay alive)
    nb, keep_b, nnb = host_image_of(b)
    assert_stats_equal(sa, ra.stats[0], STATS_BUILD_FIELDS, "exact octree")
    assert_dumps_equal(oracle.dump_image(na, nna), ra.dump(), "exact octree")
    assert_stats_equal(sb, rb.stats[0], GRANULARITY_FREE_STATS, "coalesced octree")
    got, want = oracle.dump_image(nb, nnb), rb.dump()
    assert len(got) == len(want)
    for f in GRANULARITY_FREE_FIELDS:
        assert np.array_equal(got[f], want[f]), f
    for dev, u, nodes, nn in ((a, ua, na, nna), (b, ub, nb, nnb)):
        for hqs in (0, 1):
            u["useHighQualityShading"] = hqs
            dev.render(u)
            assert dev is b or dev.lists_read_through_table() > 0    # (b's visible nodes at this size are inner nodes with more voxel chunks than a table row holds)
            fo, _, so = _oracle_render(nodes, nn, u)
            assert np.array_equal(dev.framebuffer(W, H), fo)
            assert_stats_equal(dev.read_stats(), so, STATS_RENDER_FIELDS, f"hqs={hqs}")
    # a knob set on one context is not seen by the other; an unknown name is refused
    b.tune("SIMLOD_RASTER_LEAF_TABLE", 0)
    a.render(ua)
    assert a.lists_read_through_table() > 0
    a.tune("SIMLOD_RASTER_LEAF_TABLE", 0)
    a.render(ua)
    assert a.lists_read_through_table() == 0
    with pytest.raises(SimlodError):
        a.tune("SIMLOD_NO_SUCH_KNOB", 1)
    a.close(); b.close()


# ---- the rasteriser reads chunk lists through the builder's chunk table ------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["exact", "coalesced"])
def test_frames_through_the_builders_chunk_table_equal_frames_by_pointer_chase(built_libs, mode):
    """render.hip r_visible takes a visible node's chunk addresses from the builder's table (leaf rows: point chunks, inner rows: voxel
    chunks) while the table's stamp says it describes the octree as it is now.  After every drain of a growing octree, after a reset and
    a rebuild with other points in the same buffers, and for an image uploaded behind the builder's back: the frame equals the oracle's
    rasteriser on the same image, and equals the frame drawn with the table switched off (SIMLOD_RASTER_LEAF_TABLE=0)."""
    from simlod_amd.runtime import lib
    box = (3000.0, 2000.0, 200.0)
    T = camera.lookat_transform((1.2 * box[0], -0.6 * box[1], 1.1 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    try:
        dev = _device(ring_slots=8, coalesce=(mode == "coalesced"), momentary_bytes=400_000_000)
        u = None

        def frames(tag, expect_table):
            nodes, pers, nn = host_image_of(dev)
            for hqs in (0, 1):
                u["useHighQualityShading"] = hqs
                fo, _, so = _oracle_render(nodes, nn, u)
                dev.render(u)
                used = dev.lists_read_through_table()
                f1, s1 = dev.framebuffer(W, H), dev.read_stats()
                dev.tune("SIMLOD_RASTER_LEAF_TABLE", 0)
                dev.render(u)
                dev.tune("SIMLOD_RASTER_LEAF_TABLE", None)
                assert dev.lists_read_through_table() == 0
                assert np.array_equal(f1, dev.framebuffer(W, H)), f"{tag} hqs={hqs}: table on != table off"
                assert np.array_equal(f1, fo), f"{tag} hqs={hqs}: {int((f1 != fo).sum())} pixels differ from the oracle"
                assert_stats_equal(s1, so, STATS_RENDER_FIELDS, tag)
                if expect_table:
                    assert used >= int(so["numVisibleNodes"]) - 1, f"{tag}: only {used} of {int(so['numVisibleNodes'])} lists came from the table"
                else:
                    assert used == 0, tag

        for seed, n in ((9, 5_000_000), (23, 3_300_000)):        # second pass: reset, other points, same buffers
            pts, _ = synthetic.terrain(n, seed=seed, box=box, tile=125.0)
            u = dev.uniforms(W, H, T, box)
            dev.reset(u)
            step = 2 * abi.MAX_BATCH_SIZE
            for i in range(0, n, step):
                for j in range(i, min(i + step, n), abi.MAX_BATCH_SIZE):
                    dev.upload(pts[j:j + abi.MAX_BATCH_SIZE])
                dev.drain(u)
                assert int(dev.read_stats()["dbg"]) == 0
                frames(f"{mode} seed {seed} after {min(i + step, n)} points", True)
        # an octree image written by the host: no builder launch describes it, the table must not be trusted
        ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
        pts, _ = synthetic.terrain(2_200_000, seed=5, box=box, tile=125.0)
        ref.reset(u)
        ref.add_points(u, pts)
        nn, used = int(ref.stats["numNodes"][0]), int(ref.stats["allocatedBytes_persistent"][0])
        nodes, pers = ref.nodes[:nn].copy(), ref.persistent[:used].copy()
        oracle.rebase_image_to(nodes, nn, pers, ref.nodes.ctypes.data, ref.persistent.ctypes.data, dev.nodes.data_ptr(), dev.persistent.data_ptr())
        dev.upload_image(nodes, pers, nn)
        frames(f"{mode} uploaded image", False)
    finally:
        pass                                  # (knobs and ingest mode belong to the octree's context: nothing process-wide to restore)


# ---- bench.py as the driver launches it for N > 1 ---------------------------------------------------------------------------------------
def test_bench_runs_under_torch_distributed_run_and_prints_one_json_line(built_libs):
    """The driver launches `bench.py --gpus N` through torch.distributed.run for N > 1: that path (RCCL process group, device-generated
    tiles, partition, routed ingest, composed frames) must run and leave exactly ONE JSON line on stdout — here with one rank, the
    only world size a 1-GPU box has; world 2 and 4 are covered on CPU (gloo) in test_distributed_cpu.py."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--points", "4000000", "--frames", "2", "--no-profile", "--persistent-gb", "8"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "M points/s"
    assert d["raster"]["plain"]["value"] > 0 and d["raster"]["hqs"]["value"] > 0 and d["partition"]["per_rank_points"] == [4000000]


@pytest.mark.parametrize("kind", ["uniform", "terrain"])
def test_many_tiny_batches_match_oracle(built_libs, kind):
    """300 batches of 3 000 points, 20 per launch: the root stays a leaf for the first 16 batches (its samples are voxelized on the
    caller's stream), then splits while the previous batch's voxel half may still be running on the side stream; every later batch
    leaves a few samples in many leaves (the wave-per-leaf path of k_voxelize).  Octree and Stats == oracle after the same batches."""
    n, bs = 900_000, 3_000
    pts, box = (synthetic.uniform_cube(n, seed=41) if kind == "uniform" else synthetic.terrain(n, seed=43, box=(900.0, 600.0, 60.0), tile=30.0))
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(ring_slots=abi.BATCH_STREAM_SIZE)
    u = dev.uniforms(W, H, T, box)
    _ingest(dev, u, [pts[i:i + bs] for i in range(0, n, bs)])
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0 and int(ds["batchletIndex"]) == n // bs
    ref = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
    ref.reset(u)
    ref.add_points(u, pts, bs)
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, kind)
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), kind)
    oracle.check_invariants(nodes, nn)


@pytest.mark.parametrize("variant", ["plain", "hqs", "hqs_boxes"])
def test_composed_frame_entry_points_call_the_reductions_in_order_and_draw_the_same_frame(built_libs, variant):
    """simlod_render_frame_composed (the C entry a non-Python host drives N GPUs with): the reductions are requested in the order and with
    the shapes include/simlod_hip.h states, and with reductions that change nothing (one rank) the frame is simlod_launch_render's.
    simlod_render_frame_rccl: the same through ncclAllReduce on a ONE-rank RCCL communicator made here (what one GPU can exercise of the
    RCCL path: library lookup, data types, in-place all-reduce on the launch stream)."""
    import torch
    pts, box = synthetic.terrain(2_000_000, seed=3)
    Wd, Hd = 640, 360
    T = camera.lookat_transform((1.1 * box[0], -0.9 * box[1], 1.2 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), Wd, Hd)
    dev = _device(max_pixels=Wd * Hd)
    u = dev.uniforms(Wd, Hd, T, box, hqs=variant.startswith("hqs"), show_bounding_box=variant.endswith("boxes"))
    _ingest(dev, u, [pts[i:i + abi.MAX_BATCH_SIZE] for i in range(0, len(pts), abi.MAX_BATCH_SIZE)])
    dev.render(u)
    want_fb, want_color = dev.framebuffer(Wd, Hd), dev.color(Wd, Hd)
    calls = []
    dev.render_buffer.fill_(0xA5)
    dev.render_composed(u, reduce=lambda plane, data, count, eb, op, stream: calls.append((plane, data - dev.render_buffer.data_ptr(), count, eb, op)) or 0)
    px = Wd * Hd
    fb_off, d_off, s_off = int(dev.L.simlod_render_framebuffer_offset()), int(dev.L.simlod_render_depth_plane_offset(Wd, Hd)), int(dev.L.simlod_render_sum_planes_offset(Wd, Hd))
    expect = {"plain": [(2, fb_off, px, 8, 0)], "hqs": [(0, d_off, px, 4, 0), (1, s_off, 4 * px, 4, 1)], "hqs_boxes": [(0, d_off, px, 4, 0), (1, s_off, 4 * px, 4, 1), (2, fb_off, px, 8, 0)]}[variant]
    assert calls == expect
    assert np.array_equal(dev.framebuffer(Wd, Hd), want_fb) and np.array_equal(dev.color(Wd, Hd), want_color)
    # a failing reduction ends the frame with its code
    from simlod_amd.runtime import SimlodError
    with pytest.raises(SimlodError):
        dev.render_composed(u, reduce=lambda *a: 7)
    # RCCL, one rank.  The communicator comes from the copy of RCCL this process already carries (PyTorch's own torch/lib/librccl.so) and the
    # library must reduce with THAT copy (a second instance loaded by bare name would be handed a foreign communicator): it is made
    # globally visible here, as a host that links RCCL has it, and the library's lookup (dlsym(RTLD_DEFAULT) first) must then report its version
    rccl = None
    for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL); break
        except OSError:
            pass
    if rccl is None:
        pytest.skip("no librccl.so on this box")
    ver = ctypes.c_int(0)
    assert rccl.ncclGetVersion(ctypes.byref(ver)) == 0 and 21000 <= ver.value < 30000, ver.value
    dev.L.simlod_rccl_version.restype = ctypes.c_int
    assert dev.L.simlod_rccl_version() == ver.value, "the library found another RCCL than the one that makes the communicator"
    uid = (ctypes.c_char * 128)()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    uid_s = UniqueId(); ctypes.memmove(ctypes.byref(uid_s), uid, 128)
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid_s, 0) == 0
    try:
        dev.render_buffer.fill_(0xA5)
        dev.render_composed(u, rccl_comm=comm.value)
        torch.cuda.synchronize()
        # one rank: every reduction (MIN of the depth plane, SUM of the colour sums, MIN of the framebuffer) must leave its plane as it was —
        # a wrong datatype or operator number (the entry passes them by value: the 2.x enum ABI) would not
        assert np.array_equal(dev.framebuffer(Wd, Hd), want_fb) and np.array_equal(dev.color(Wd, Hd), want_color)
        if variant.startswith("hqs"):
            dev._frame_size = (Wd, Hd)
            depth_after, sums_after = dev.depth_plane().cpu().numpy().copy(), dev.sum_planes().cpu().numpy().copy()
            dev.render_buffer.fill_(0xA5)
            dev.render_composed(u, reduce=lambda *a: 0)
            torch.cuda.synchronize()
            assert np.array_equal(dev.depth_plane().cpu().numpy(), depth_after) and np.array_equal(dev.sum_planes().cpu().numpy(), sums_after)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_repeated_ingests_leave_identical_counters(built_libs):
    """The race hunt of tools/stress.py in the test tier: the 36 M ingest over and over through the two-stream pipeline (kernels of one group
    on two streams, ordered by events without a system-scope fence) — in exact mode every build counter of Stats is the reference's, hence
    the same in every pass; a lost update between the streams shows up as a few thousand missing voxels once in a while."""
    n = 36_000_000
    pts, box = synthetic.terrain(n, seed=7)
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * max(box)), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(persistent_bytes=4 << 30)
    u = dev.uniforms(W, H, T, box)
    fields = ["numNodes", "numInner", "numLeaves", "numNonemptyLeaves", "numPoints", "numVoxels", "numChunksPoints", "numChunksVoxels", "batchletIndex",
              "numPointsProcessed", "numAllocatedChunks", "chunkPoolSize", "allocatedBytes_persistent", "dbg", "memCapacityReached"]
    batches = [pts[i:i + abi.MAX_BATCH_SIZE] for i in range(0, n, abi.MAX_BATCH_SIZE)]
    import torch
    rv = dev.ring.view(torch.uint8)
    for i, b in enumerate(batches):
        rv[i * abi.MAX_BATCH_SIZE * 16: i * abi.MAX_BATCH_SIZE * 16 + len(b) * 16].copy_(torch.from_numpy(b.view(np.uint8).reshape(-1)))
    sizes = torch.tensor([len(b) for b in batches], dtype=torch.int32, device=dev.device)
    first = None
    for p in range(40):
        dev.reset(u)
        dev.batch_sizes[: len(batches)] = sizes
        dev.publish(len(batches))
        dev.uploaded_host = len(batches)
        dev.drain(u)
        st = dev.read_stats()
        got = {f: int(st[f]) for f in fields}
        if first is None:
            first = got
            assert got["numPoints"] == n and got["dbg"] == 0
        assert got == first, f"pass {p}: " + str({f: (first[f], got[f]) for f in fields if got[f] != first[f]})


def test_two_coalesced_contexts_ingest_from_two_threads(built_libs):
    """Two octrees, each with its own context in coalesced mode (k_expand on every CU, meeting at a grid barrier), fed from two host threads
    at once: the k_expand launches of the device form one chain across contexts (simlod_internal.hpp expand_gate_*), so two barrier kernels
    never hold the CUs against each other; both octrees end as their oracle octrees (content fields; the coalesced mode's contract)."""
    import threading
    import torch
    sets = [synthetic.terrain(9_000_000, seed=21), synthetic.terrain(9_000_000, seed=22)]
    T = [camera.lookat_transform((1.8 * b[0], -1.2 * b[1], 1.4 * max(b)), (0.5 * b[0], 0.5 * b[1], 0.3 * b[2]), W, H) for _, b in sets]
    from simlod_amd.runtime import DeviceOctree
    devs = [DeviceOctree("cuda:0", persistent_bytes=2 << 30, momentary_bytes=700_000_000, max_pixels=W * H, coalesce=True) for _ in sets]
    us = [d.uniforms(W, H, t, b) for d, t, (_, b) in zip(devs, T, sets)]
    errors = []

    def work(k):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                for rep in range(3):
                    devs[k].reset(us[k])
                    devs[k].add_points(us[k], sets[k][0])
                torch.cuda.current_stream().synchronize()
        except Exception as e:                                  # noqa: BLE001
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=300)
    assert not errors and not any(t.is_alive() for t in threads), errors
    torch.cuda.synchronize()
    for k, (pts, box) in enumerate(sets):
        ds = devs[k].read_stats()
        assert int(ds["dbg"]) == 0 and int(ds["numPoints"]) == len(pts)
        ref = oracle.HostOctree("port", persistent_bytes=2 << 30, ring_slots=abi.BATCH_STREAM_SIZE)
        ref.reset(us[k]); ref.add_points(us[k], pts)
        assert_stats_equal(ds, ref.stats[0], GRANULARITY_FREE_STATS, f"context {k}")
        nodes, pers, nn = host_image_of(devs[k])
        da, db = oracle.dump_image(nodes, nn), ref.dump()
        for f in GRANULARITY_FREE_FIELDS:
            assert np.array_equal(da[f], db[f]), f"context {k}: {f}"
    for d in devs:
        d.close()
