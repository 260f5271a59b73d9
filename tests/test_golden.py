"""The committed fixtures of tests/golden/ were produced by the reference's own sources (tests/golden/make_golden.py).
Here the CPU restatement must reproduce them — this runs everywhere, also where /root/reference does not exist."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from cases import CASES, batches_of, case, uniforms_for
from simlod_amd import abi
from util import STATS_BUILD_FIELDS, STATS_RENDER_FIELDS, assert_dumps_equal, assert_stats_equal

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"{name}.npz"))


@pytest.mark.parametrize("name", CASES)
def test_restatement_reproduces_golden(built_libs, name):
    g = load_golden(name)
    pts, box, batch, T = case(name)
    u = uniforms_for(box, T)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=8)
    o.reset(u)
    for b in batches_of(name, pts, batch):
        o.upload(b)
    while int(o.stats["batchletIndex"][0]) < int(o.num_uploaded[0]):
        o.construct(u)
    assert o.last_error() == 0
    assert_stats_equal(o.stats[0], g["build_stats"][0], STATS_BUILD_FIELDS, name)
    assert_dumps_equal(o.dump(), g["dump"], name)
    inv = oracle.check_invariants(o.nodes, int(o.stats["numNodes"][0]))
    assert inv["points"] == len(pts)
    for mode, hqs in (("plain", False), ("hqs", True)):
        fb, _ = o.render(uniforms_for(box, T, hqs=hqs))
        assert hashlib.sha256(fb.tobytes()).digest() == g[f"fb_sha256_{mode}"].tobytes(), f"{name}/{mode}: framebuffer hash"
        assert int((fb != abi.CLEAR_PIXEL).sum()) == int(g[f"fb_nonbg_{mode}"][0])
        assert_stats_equal(o.stats[0], g[f"render_stats_{mode}"][0], STATS_RENDER_FIELDS, name)


def test_config1_counters_of_the_survey(built_libs):
    """BASELINE.json configs[0] (1 M uniform points, mt19937(1234)): the numbers the reference build produced during the survey
     and that oracle/_ref reproduces here: 73 nodes, 1 766 795 voxels, 47 168 896 persistent bytes, 57 visible."""
    from simlod_amd import camera, synthetic
    pts, box = synthetic.uniform_cube(1_000_000, seed=1234)
    T = camera.lookat_transform((1.8, -1.2, 1.4), (0.5, 0.5, 0.3), 512, 512)
    u = abi.make_uniforms(512, 512, T, box, persistent_capacity=1 << 30, momentary_capacity=300_000_000)
    o = oracle.HostOctree("port", persistent_bytes=1 << 30, ring_slots=2)
    o.reset(u)
    o.add_points(u, pts)
    s = o.stats[0]
    assert (int(s["numNodes"]), int(s["numInner"]), int(s["numLeaves"])) == (73, 9, 64)
    assert (int(s["numPoints"]), int(s["numVoxels"])) == (1_000_000, 1_766_795)
    assert (int(s["numChunksPoints"]), int(s["numChunksVoxels"]), int(s["allocatedBytes_persistent"])) == (1024, 1771, 47_168_896)
    o.render(u)
    assert (int(s["numVisibleNodes"]), int(s["numVisiblePoints"]), int(s["numVisibleVoxels"])) == (57, 875_536, 120_874)
