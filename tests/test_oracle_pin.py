"""Pins the C restatement (oracle/simlod_oracle.c) against the reference ITSELF: oracle/_ref = the reference's own
reset.cu / progressive_octree_voxels.cu / render.cu compiled in place as single-thread host code (oracle/Makefile).
A serial run has one legal outcome, so the two must agree on every byte that is not an address: node records, chunk
contents at identical allocator offsets, occupancy grids, Stats, pre-EDL framebuffers."""
import os

import numpy as np
import pytest

import oracle
from cases import CASES, batches_of, case, uniforms_for
from util import STATS_BUILD_FIELDS, STATS_RENDER_FIELDS, assert_dumps_equal, assert_stats_equal

pytestmark = pytest.mark.skipif(not (oracle.have_ref() or os.path.isdir("/root/reference")),
                                reason="oracle/_ref not built and /root/reference absent")

NODE_VALUE_FIELDS = ["counter", "numPoints", "level", "X", "Y", "Z", "countIteration", "name", "numVoxels", "numVoxelsStored"]


def _run(kind, name):
    pts, box, batch, T = case(name)
    u = uniforms_for(box, T)
    o = oracle.HostOctree(kind, persistent_bytes=1 << 30, ring_slots=8)
    o.reset(u)
    for b in batches_of(name, pts, batch):
        o.upload(b)
    while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
        o.construct(u)
    return o, box, T


@pytest.mark.parametrize("name", CASES)
def test_restatement_equals_reference_build(built_libs, name):
    ref, box, T = _run("ref", name)
    port, _, _ = _run("port", name)
    assert port.last_error() == 0
    assert_stats_equal(port.stats[0], ref.stats[0], STATS_BUILD_FIELDS, name)
    n = int(ref.stats["numNodes"][0])
    for f in NODE_VALUE_FIELDS:
        assert np.array_equal(ref.nodes[f][:n], port.nodes[f][:n]), f"Node.{f}"
    # same allocation order -> same offsets inside the persistent buffer for grids and chunk lists
    for f in ("grid", "points", "voxelChunks"):
        ra = np.where(ref.nodes[f][:n] != 0, ref.nodes[f][:n] - np.uint64(ref.persistent.ctypes.data), 0)
        pa = np.where(port.nodes[f][:n] != 0, port.nodes[f][:n] - np.uint64(port.persistent.ctypes.data), 0)
        assert np.array_equal(ra, pa), f"Node.{f} offsets"
    ca = np.where(ref.nodes["children"][:n] != 0, ref.nodes["children"][:n] - np.uint64(ref.nodes.ctypes.data), 0)
    cb = np.where(port.nodes["children"][:n] != 0, port.nodes["children"][:n] - np.uint64(port.nodes.ctypes.data), 0)
    assert np.array_equal(ca, cb), "children indices"
    assert_dumps_equal(port.dump(), ref.dump(), name)       # includes every stored point, voxel position and grid bit
    for hqs in (False, True):
        u = uniforms_for(box, T, hqs=hqs)
        fa, _ = ref.render(u)
        fb, _ = port.render(u)
        assert np.array_equal(fa, fb), f"{name}: pre-EDL framebuffer differs (hqs={hqs})"
        assert_stats_equal(port.stats[0], ref.stats[0], STATS_RENDER_FIELDS, name)
        va, vb = ref.visible, port.visible
        assert len(va) == len(vb) and np.array_equal(va["name"], vb["name"]), "visible-node list order"


def test_point_size_and_color_modes_match_reference(built_libs):
    ref, box, T = _run("ref", "uniform_3x40k")
    port, _, _ = _run("port", "uniform_3x40k")
    for kw in (dict(point_size=2), dict(point_size=3)):
        u = uniforms_for(box, T, **kw)
        assert np.array_equal(ref.render(u)[0], port.render(u)[0]), kw
    for hqs in (0, 1):                      # debug lines: node boxes + frustum, rasterization.cuh:90-183
        u = uniforms_for(box, T, hqs=bool(hqs))
        u["showBoundingBox"] = 1
        fa, fb = ref.render(u)[0], port.render(u)[0]
        assert np.array_equal(fa, fb) and int(((fa & np.uint64(0xffffffff)) == np.uint64(0xff00)).sum()) > 500
    for field in ("colorByNode", "colorByLOD"):
        u = uniforms_for(box, T)
        u[field] = 1
        for hqs in (0, 1):
            u["useHighQualityShading"] = hqs
            assert np.array_equal(ref.render(u)[0], port.render(u)[0]), (field, hqs)


def test_hazard_regimes_match_reference(built_libs):
    """Where the reference is lossy the restatement must be lossy in the same way (the HIP path documents where it is not):
    20 split rounds down to level 20 with 70 000 identical points (the reference does not count after its 20th split, allocates no
    chunks for the level-20 leaf, increments its numPoints and drops the points, voxels.cu:394-412, 599-604 — the restatement
    reports that as NULL_CHUNK), and points exactly on the box faces (coordinate == boxMax quantises to 2^20, wraps into the low
    child, voxels.cu:148-179)."""
    from simlod_amd import abi, camera, synthetic
    rs = np.random.RandomState(4)
    base, box = synthetic.uniform_cube(20_000, seed=9)
    same = np.repeat(base[:1], 70_000)
    same["x"], same["y"], same["z"] = np.float32(0.3), np.float32(0.6), np.float32(0.2)
    corners = np.repeat(base[:1], 4_000)
    for k in "xyz":
        corners[k] = rs.choice(np.array([0.0, 1.0], dtype=np.float32), 4_000)
    pts = np.concatenate([base[:10_000], same, corners, base[10_000:]])
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), 256, 256)
    u = uniforms_for(box, T)
    runs = {}
    for kind in ("ref", "port"):
        o = oracle.HostOctree(kind, persistent_bytes=1 << 30, ring_slots=4)
        o.reset(u)
        o.add_points(u, pts, 50_000)
        runs[kind] = o
    ref, port = runs["ref"], runs["port"]
    assert port.last_error() == 5                                   # ORACLE_ERR_NULL_CHUNK
    assert_stats_equal(port.stats[0], ref.stats[0], STATS_BUILD_FIELDS, "hazards")
    assert int(ref.stats["numNodes"][0]) == 1 + 8 * 20 and int(ref.nodes["level"][:161].max()) == abi.MAX_DEPTH
    assert_dumps_equal(port.dump(), ref.dump(), "hazards")
    assert np.array_equal(ref.render(u)[0], port.render(u)[0])


@pytest.mark.parametrize("variant", ["odd_size_hqs", "odd_size_plain_boxes", "tiny_frame", "inside_the_cloud_hqs", "inside_the_cloud_plain",
                                     "fine_lod_hqs", "boxes_only"])
def test_render_edge_cases_match_reference(built_libs, variant):
    """The render edge cases of the GPU suite (tests/test_gpu_parity.py: frame sizes off the 16-pixel grid, camera inside the cloud,
    fine LOD threshold, lines without points), restatement against the reference's own render.cu on a reference-built octree."""
    from simlod_amd import abi, camera, synthetic
    Wd, Hd = (250, 131) if "odd_size" in variant else (40, 23) if variant == "tiny_frame" else (384, 256)
    pts, box = synthetic.uniform_cube(300_000, seed=77)
    if "inside" in variant:
        eye, target = (0.52, 0.48, 0.5), (0.9, 0.6, 0.45)
    else:
        eye, target = (1.8, -1.2, 1.4), (0.5, 0.5, 0.3)
    T = camera.lookat_transform(eye, target, Wd, Hd)
    u = abi.make_uniforms(Wd, Hd, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000, hqs="hqs" in variant)
    u["showBoundingBox"] = 1 if "boxes" in variant else 0
    u["showPoints"] = 0 if variant == "boxes_only" else 1
    if variant == "fine_lod_hqs" or "odd_size" in variant:
        u["minNodeSize"] = 8.0
    if variant == "tiny_frame":
        u["minNodeSize"] = 2.0
    frames = {}
    for kind in ("ref", "port"):
        o = oracle.HostOctree(kind, persistent_bytes=1 << 30, ring_slots=4)
        o.reset(u)
        o.add_points(u, pts, 100_000)
        frames[kind] = (o.render(u)[0], o.stats[0].copy(), o.visible["name"].copy())
    assert np.array_equal(frames["ref"][0], frames["port"][0]), variant
    assert_stats_equal(frames["port"][1], frames["ref"][1], STATS_RENDER_FIELDS, variant)
    assert np.array_equal(frames["ref"][2], frames["port"][2])
    assert int((frames["ref"][0] != abi.CLEAR_PIXEL).sum()) > 50


def test_edl_restatement_matches_the_reference_edl_block():
    """EDL (render.cu:1255-1325) pinned to the reference itself: tests/golden/edl_uniform_3x40k.npz was minted by running the
    reference's own render.cu with its EDL pass enabled, one in-tile thread per call (tests/golden/make_golden_edl.py).  The
    restatement's EDL'd RGBA8 image must agree within 1 per channel (log2 / exp come from different libms) on every full 16x16 tile.
    The last row is left out: there the reference reads the depth of pixel W*H — one past the framebuffer (render.cu:1303)."""
    import os
    from cases import H, W, batches_of, case, uniforms_for
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "edl_uniform_3x40k.npz"))
    pts, box, batch, T = case("uniform_3x40k")
    u = uniforms_for(box, T)
    o = oracle.HostOctree("port", persistent_bytes=1 << 28, ring_slots=8)
    o.reset(u)
    for b in batches_of("uniform_3x40k", pts, batch):
        o.upload(b)
    while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
        o.construct(u)
    rows = np.arange(W * H) // W
    cmp = rows < H - 1
    for mode, hqs in (("plain", False), ("hqs", True)):
        fb, col = o.render(uniforms_for(box, T, hqs=hqs), edl=True)
        want = G[f"color_{mode}"]
        d = np.abs(col.view(np.uint8).astype(np.int16) - want.view(np.uint8).astype(np.int16)).reshape(-1, 4).max(axis=1)
        assert int(d[cmp].max()) <= 1, f"{mode}: {int((d[cmp] > 1).sum())} pixels differ by more than 1 from the reference's EDL output"
        assert int((want & 0xffffff != (fb & 0xffffff).astype(np.uint32)).sum()) > 1000, "the fixture must actually be shaded"
