"""BASELINE configs 3 and 5 at their full sizes against the CPU oracle (configs[1] at full size: test_gpu_parity.py
test_full_size_36m_properties).  Slow for unit tests — a minute or two each, most of it the serial oracle — but part of `-m gpu`:
the octree a 350 M-point stream leaves after 350 incremental batches, and the one 200 M points in a single level-6 cell force,
are compared node by node (every Stats counter incl. allocator offset and chunk pool; topology, counters, point multisets,
occupancy bitsets, voxel positions), and one HQS frame of each is compared pixel by pixel with the oracle's rasteriser on the
same image.  Reference: progressive_octree_voxels.cu:883-949 (the batch loop), main_progressive_octree.cpp:1012 (back-pressure)."""
import ctypes

import numpy as np
import pytest

import oracle
from simlod_amd import abi, camera, lasio
from util import STATS_BUILD_FIELDS, STATS_RENDER_FIELDS, assert_dumps_equal, assert_stats_equal, host_image_of

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
Wd, Hd = 1920, 1080


def _enough_memory(host_gb, device_gb):
    import psutil
    import torch
    if psutil.virtual_memory().available < host_gb * (1 << 30):
        pytest.skip(f"needs {host_gb} GB of free host memory for the oracle's copy of the octree")
    free, _ = torch.cuda.mem_get_info()
    if free < device_gb * (1 << 30):
        pytest.skip(f"needs {device_gb} GB of free device memory")


def _oracle_frame(nodes, nn, u):
    fb = np.zeros(Wd * Hd, dtype=np.uint64)
    col = np.zeros(Wd * Hd, dtype=np.uint32)
    vis = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
    stats = np.zeros(1, dtype=abi.stats_dtype)
    stats["numNodes"] = nn
    uu = np.ascontiguousarray(u).reshape(1)
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    oracle.port_lib().oracle_render(None, p(uu), p(nodes), p(stats), p(fb), p(col), p(vis), 1)
    return fb, stats[0]


def _compare(dev, u, ref, what, frames):
    """Stats + full dump of the device's octree against the oracle's, then the given frames against the oracle's rasteriser on the device-built image."""
    ds = dev.read_stats()
    assert ref.last_error() == 0
    assert_stats_equal(ds, ref.stats[0], STATS_BUILD_FIELDS, what)
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), what)
    for name, uc in frames:
        dev.render(uc)
        fb, st = dev.framebuffer(Wd, Hd), dev.read_stats()
        fo, so = _oracle_frame(nodes, nn, uc)
        assert int(st["numVisibleNodes"]) > 50, name
        assert_stats_equal(st, so, STATS_RENDER_FIELDS, f"{what}, {name}")
        assert np.array_equal(fb, fo), f"{what}, {name}: {int((fb != fo).sum())} of {Wd * Hd} pixels differ from the oracle"


def test_config3_350m_scan_ordered_las_records_streamed_in_1m_batches_equal_the_oracle(built_libs):
    """BASELINE config 3 at full size: 350 M points of the terrain stand-in in scan order (flight lines of 250 m), as LAS format-2 records
    (int32 coordinates at 1 mm, 16-bit colours, 26 bytes), streamed in 350 batches of 1 M: every batch is decoded ON THE DEVICE into the
    next ring slot (simlod_decode_las) and kernel_construct is launched whenever the ring is full and at the end (the reference's frame loop
    under back-pressure).  The oracle decodes the same records with its restatement of LasLoader.cpp:169-227 and inserts them batch by
    batch.  Everything must be equal after the 350th batch; Stats.dbg must be 0 (no split deferred)."""
    import torch
    from simlod_amd.runtime import DeviceOctree
    n, batch = 350_000_000, abi.MAX_BATCH_SIZE
    _enough_memory(host_gb=48, device_gb=40)
    box = (19000.0, 12600.0, 400.0)                        # the 36 M terrain's density (1.5 points per square metre) at 350 M points
    dev = DeviceOctree("cuda:0", persistent_bytes=48 * n, max_pixels=Wd * Hd)
    dev.momentary.fill_(0xA5); dev.render_buffer.fill_(0xA5)
    src = torch.empty(n * 16, dtype=torch.uint8, device=dev.device)
    dev.generate_terrain(src, 0, n, 7, 1, box, swath_width=250.0)
    T = camera.world_view_proj(camera.orbit_view(-0.207, -0.797, 3866.886 * box[0] / 6000.0, (box[0] / 2, box[1] / 2, 0.35 * box[2])), camera.perspective(aspect=Wd / Hd))
    u = dev.uniforms(Wd, Hd, T, box, hqs=True)
    ref = oracle.HostOctree("port", persistent_bytes=48 * n, ring_slots=abi.BATCH_STREAM_SIZE)
    uh = u.copy()
    ref.reset(uh)
    dev.reset(u)
    # the file's header, as lasio.points_to_las would write it: scale 1 mm, offset = the survey's corner; the loader translates by -min
    h = lasio.LasHeader(versionMajor=1, versionMinor=4, format=2, bytesPerPoint=26, numPoints=n, scale=(0.001, 0.001, 0.001),
                        offset=(694000.0, 3915000.0, -3.0), min=(694000.0, 3915000.0, -3.0), max=(694000.0 + box[0], 3915000.0 + box[1], -3.0 + box[2]))
    tr = tuple(-m for m in h.min)
    filler = np.random.RandomState(3).randint(0, 256, size=(batch, 26), dtype=np.uint8)   # the bytes a decoder must ignore (intensity, flags, ...)

    def drain_ref():
        while int(ref.stats["batchletIndex"][0]) < int(ref.num_uploaded[0]):
            ref.construct(uh)
    for first in range(0, n, batch):
        p = src[first * 16: (first + batch) * 16].cpu().numpy().view(abi.point_dtype)
        m = len(p)
        rec = filler[:m].copy()
        xyz = np.stack([np.rint(p["x"].astype(np.float64) / 0.001), np.rint(p["y"].astype(np.float64) / 0.001), np.rint(p["z"].astype(np.float64) / 0.001)], axis=1).astype("<i4")
        rec[:, 0:12] = xyz.view(np.uint8).reshape(m, 12)
        rgba = np.ascontiguousarray(p["color"]).view(np.uint8).reshape(m, 4)
        rec[:, 20:26] = (rgba[:, :3].astype("<u2") * 257).view(np.uint8).reshape(m, 6)    # 8-bit colours as 16-bit (LasLoader.cpp:207-215 scales them back)
        rec = rec.reshape(-1)
        if dev.uploaded_host - dev.processed_host >= dev.ring_slots:
            dev.drain(u)
        dev.upload_las(rec, h, tr)
        if int(ref.num_uploaded[0]) - int(ref.stats["batchletIndex"][0]) >= abi.BATCH_STREAM_SIZE:
            drain_ref()
        ref.upload(oracle.decode_las_port(rec, 26, 2, h.scale, lasio.decode_offset(h, tr)))
    dev.drain(u)
    drain_ref()
    del src
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0 and int(ds["numPoints"]) == n and int(ds["batchletIndex"]) == 350
    _compare(dev, u, ref, "config 3, 350 M", [("HQS frame", u)])


def _hotspot_on_device(dev, n):
    """n points uniformly inside cell (21, 40, 13) of the 64^3 grid of the unit cube, colour from the position inside the cell — generated on
    the device (the host generator of synthetic.hotspot takes minutes at 200 M); int32 [n, 4]: x, y, z as float bits, RGBA8."""
    import torch
    g = torch.Generator(device=dev.device); g.manual_seed(11)
    src = torch.empty((n, 4), dtype=torch.int32, device=dev.device)
    cell = torch.tensor([21.0, 40.0, 13.0], device=dev.device) / 64.0
    for first in range(0, n, 50_000_000):
        r = torch.rand((min(50_000_000, n - first), 3), generator=g, device=dev.device, dtype=torch.float32)
        src[first: first + len(r), :3] = (cell + r * (0.999 / 64.0)).view(torch.int32)
        c = (r * 255.0).to(torch.int32)
        src[first: first + len(r), 3] = c[:, 0] + c[:, 1] * 256 + c[:, 2] * 65536 - 16777216   # alpha 255
        del r, c
    return src


def test_config5_200m_points_in_one_level6_cell_full_size(built_libs):
    """BASELINE config 5 at full size: 200 M points uniformly inside ONE level-6 cell of the unit cube, camera on the cell so that thousands of
    samples of a node pile up on a pixel.

    This input leaves the regime in which the reference keeps every point: uniform filling makes hundreds of leaves cross the 50 000-point
    limit in the SAME batch, and the reference inserts at most 3 000 000 spilled points per batch (progressive_octree_voxels.cu:628-631) — the
    rest is dropped.  Pinned here: the oracle (the reference's algorithm, byte for byte) and the device build the same octree for the first 25
    batches; in the 26th the oracle reports the overflow and has lost points, the device has not (capacity limits never lose
    a point here).  For all 200 batches the device's octree is therefore checked by itself — no split deferred, every input point stored
    exactly once (multiset hash), every structural and accounting identity of the image — and its plain and HQS frames are compared pixel
    by pixel with the oracle's rasteriser on the same image."""
    import torch
    from simlod_amd.runtime import DeviceOctree
    from util import points_multiset_hash
    n, batch = 200_000_000, abi.MAX_BATCH_SIZE
    _enough_memory(host_gb=32, device_gb=32)
    # (4 GB of momentary memory: with the reference host's 300 MB the same ingest defers splits — Stats.dbg bit 0x2, nothing lost, the octree
    # catches up; test_deferred_splits_catch_up_to_the_oracles_octree — and its allocator history differs from an undeferred one)
    dev = DeviceOctree("cuda:0", persistent_bytes=48 * n, momentary_bytes=4_000_000_000, max_pixels=Wd * Hd)
    dev.momentary.fill_(0xA5); dev.render_buffer.fill_(0xA5)
    src = _hotspot_on_device(dev, n)
    box = np.array([1.0, 1.0, 1.0], dtype=np.float32)
    center = np.array([21, 40, 13], dtype=np.float64) / 64 + 1 / 128
    dist = (1 / 64) * (Hd / 128.0) / (2 * np.tan(np.radians(30)))
    T = camera.lookat_transform(center + np.array([0.6, -0.7, 0.4]) / np.linalg.norm([0.6, -0.7, 0.4]) * dist, center, Wd, Hd)
    u = dev.uniforms(Wd, Hd, T, box, min_node_size=8.0)
    pts = src.cpu().numpy().view(np.uint8).reshape(-1).view(abi.point_dtype)

    # 1. the first 25 batches: equal to the oracle; the 26th: the reference's algorithm overflows and drops points
    prefix = 25 * batch
    dev.reset(u)
    dev.stream(u, src.view(torch.uint8).reshape(-1)[: prefix * 16], prefix)
    ref = oracle.HostOctree("port", persistent_bytes=48 * 30 * batch, ring_slots=abi.BATCH_STREAM_SIZE)
    uh = u.copy(); uh["persistentBufferCapacity"] = 48 * 30 * batch
    ref.reset(uh)
    for i in range(0, prefix, batch):
        ref.upload(pts[i:i + batch]); ref.construct(uh)
    assert ref.last_error() == 0 and int(ref.stats["numPoints"][0]) == prefix
    ds = dev.read_stats()
    assert int(ds["dbg"]) == 0
    assert_stats_equal(ds, ref.stats[0], [f for f in STATS_BUILD_FIELDS if f != "allocatedBytes_momentary"], "config 5, first 25 batches")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "config 5, first 25 batches")
    ref.upload(pts[prefix: prefix + batch]); ref.construct(uh)
    lost = prefix + batch - int(ref.stats["numPoints"][0])
    assert ref.last_error() == 3 and lost > 0, "the 26th batch spills more than 3 000 000 points: the reference's algorithm drops the rest"
    del ref, nodes, pers

    # 2. all 200 batches on the device
    dev.reset(u)
    launches = dev.stream(u, src.view(torch.uint8).reshape(-1), n)
    del src
    s = dev.read_stats()
    assert int(s["dbg"]) == 0 and int(s["numPoints"]) == n and int(s["numPointsProcessed"]) == n and int(s["batchletIndex"]) == 200 and launches >= 10
    nodes, pers, nn = host_image_of(dev)
    inv = oracle.check_invariants(nodes, nn)
    assert inv["points"] == n and inv["voxels"] >= int(s["numVoxels"])
    assert int(s["numNodes"]) == nn == 1 + 8 * int(s["numInner"]) and int(s["numLeaves"]) == nn - int(s["numInner"])
    assert inv["point_chunks"] == int(s["numChunksPoints"]) == int(s["numAllocatedChunks"]) <= int(s["chunkPoolSize"])
    expect = 16 + inv["grids"] * abi.alloc_round(abi.GRID_BYTES) + (int(s["chunkPoolSize"]) + inv["voxel_chunks"]) * abi.alloc_round(abi.CHUNK_BYTES)
    assert int(s["allocatedBytes_persistent"]) == expect
    d = oracle.dump_image(nodes, nn)
    assert int(d["numPoints"].max()) <= abi.MAX_POINTS_PER_NODE or int(d["level"][np.argmax(d["numPoints"])]) == 20, "no leaf above the limit is left behind"
    hs, hx = points_multiset_hash(pts)
    with np.errstate(over="ignore"):
        assert hs == np.uint64(d["pointsSum"].sum()) and hx == np.bitwise_xor.reduce(d["pointsXor"]), "every input point is stored exactly once"
    del pts
    # 3. frames on that image
    uh = dev.uniforms(Wd, Hd, T, box, min_node_size=8.0, hqs=True)
    for name, uc in (("plain frame", u), ("HQS frame", uh)):
        dev.render(uc)
        fb, st = dev.framebuffer(Wd, Hd), dev.read_stats()
        fo, so = _oracle_frame(nodes, nn, uc)
        assert int(st["numVisibleNodes"]) > 50, name
        assert_stats_equal(st, so, STATS_RENDER_FIELDS, f"config 5, 200 M, {name}")
        assert np.array_equal(fb, fo), f"config 5, 200 M, {name}: {int((fb != fo).sum())} of {Wd * Hd} pixels differ from the oracle's rasteriser"
