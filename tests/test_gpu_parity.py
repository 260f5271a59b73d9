"""GPU parity: the HIP kernels (through the C ABI of libsimlod_hip.so) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

import oracle
from simlod_amd import abi, camera, synthetic

from util import (STATS_BUILD_FIELDS, STATS_RENDER_FIELDS, assert_dumps_equal, assert_stats_equal, host_image_of,
                  voxel_colors_are_member)

pytestmark = pytest.mark.gpu


def _device(**kw):
    from simlod_amd.runtime import DeviceOctree
    return DeviceOctree("cuda:0", **kw)


def _build_both(points, box, batch, *, ring_slots=4, persistent=1 << 30, W=512, H=512):
    T = camera.lookat_transform((1.8 * box[0], -1.2 * box[1], 1.4 * box[2]), (0.5 * box[0], 0.5 * box[1], 0.3 * box[2]), W, H)
    dev = _device(persistent_bytes=persistent, ring_slots=ring_slots, max_pixels=W * H)
    u = dev.uniforms(W, H, T, box)
    dev.reset(u)
    ref = oracle.HostOctree("port", persistent_bytes=persistent, ring_slots=ring_slots)
    ref.reset(u)
    for i in range(0, len(points), batch * ring_slots):
        part = points[i:i + batch * ring_slots]
        dev.add_points(u, part, batch)
        ref.add_points(u, part, batch)
    return dev, ref, u


@pytest.mark.parametrize("n,batch", [(1_000_000, 1_000_000), (300_000, 100_000), (40_000, 40_000)])
def test_construct_uniform_matches_oracle(built_libs, n, batch):
    pts, box = synthetic.uniform_cube(n, seed=1234)
    dev, ref, u = _build_both(pts, box, batch)
    ds, rs = dev.read_stats(), ref.stats[0]
    assert int(ds["dbg"]) == 0, f"device error bits {int(ds['dbg']):#x}"
    assert ref.last_error() == 0
    assert_stats_equal(ds, rs, STATS_BUILD_FIELDS, "construct")
    nodes, pers, nn = host_image_of(dev)
    assert_dumps_equal(oracle.dump_image(nodes, nn), ref.dump(), "construct")
    assert voxel_colors_are_member(nodes, nn, pts, box) == int(nodes["numVoxelsStored"][:nn].sum())


@pytest.mark.parametrize("hqs", [False, True])
def test_render_bit_exact_on_device_built_octree(built_libs, hqs):
    """Framebuffer parity on the SAME octree image: build on the GPU, render on the GPU, then hand the downloaded image to
    the oracle's rasteriser.  The pre-EDL uint64 framebuffer must be bit-identical."""
    pts, box = synthetic.uniform_cube(1_000_000, seed=1234)
    W = H = 512
    dev, ref, u = _build_both(pts, box, 1_000_000, W=W, H=H)
    u["useHighQualityShading"] = 1 if hqs else 0
    dev.render(u)
    fb_dev = dev.framebuffer(W, H)
    ds = dev.read_stats()
    nodes, pers, nn = host_image_of(dev)
    # the image's pointers refer to `nodes`/`pers`: render from those arrays directly
    import ctypes
    fb = np.zeros(W * H, dtype=np.uint64)
    col = np.zeros(W * H, dtype=np.uint32)
    vis = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
    stats = np.zeros(1, dtype=abi.stats_dtype)
    stats["numNodes"] = nn
    uu = np.ascontiguousarray(u).reshape(1)
    oracle.port_lib().oracle_render(None, ctypes.c_void_p(uu.ctypes.data), ctypes.c_void_p(nodes.ctypes.data),
                                    ctypes.c_void_p(stats.ctypes.data), ctypes.c_void_p(fb.ctypes.data),
                                    ctypes.c_void_p(col.ctypes.data), ctypes.c_void_p(vis.ctypes.data), 1)
    assert_stats_equal(ds, stats[0], STATS_RENDER_FIELDS, "render")
    diff = np.nonzero(fb_dev != fb)[0]
    assert len(diff) == 0, f"{len(diff)} pixels differ, first {diff[:5]}: dev {fb_dev[diff[:5]]} oracle {fb[diff[:5]]}"
    assert int((fb != abi.CLEAR_PIXEL).sum()) > 10_000
    # EDL'd RGBA8 output: within 1 per channel (log2/exp are not bit-portable)
    cd = dev.color(W, H).view(np.uint8).astype(np.int16)
    co = col.view(np.uint8).astype(np.int16)
    assert int(np.abs(cd - co).max()) <= 1
