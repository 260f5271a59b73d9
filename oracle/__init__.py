"""TEST INFRASTRUCTURE ONLY — CPU oracle for the SimLOD hot paths.

Two back-ends behind one interface (`HostOctree`):

* ``kind="ref"``   oracle/_ref/libref_{reset,update,render}.so — the reference's own sources
                   (modules/progressive_octree/{reset,progressive_octree_voxels,render}.cu) compiled in place as
                   single-thread host C++ through oracle/shim/ (built by `make -C oracle ref`; needs
                   /root/reference at BUILD time only, the .so files travel to the GPU box).
* ``kind="port"``  oracle/libsimlod_oracle.so — the independent C restatement (oracle/simlod_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
package simlod_amd never does.
"""
import ctypes
import os
import subprocess

import numpy as np

from simlod_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT = os.environ.get("SIMLOD_ORACLE_LIB") or os.path.join(_HERE, "libsimlod_oracle.so")   # override: a -march=native build made by bench.py
_REF = {k: os.path.join(_HERE, "_ref", f"libref_{k}.so") for k in ("reset", "update", "render")}

REF_MOMENTARY_BYTES = 408_800_192   # what kernel_construct bump-allocates; the host only gives 300 MB

dump_dtype = np.dtype([
    ("key", "<u8"), ("level", "<u4"), ("X", "<u4"), ("Y", "<u4"), ("Z", "<u4"),
    ("isLeaf", "<u4"), ("counter", "<u4"), ("numPoints", "<u4"), ("numVoxels", "<u4"), ("numVoxelsStored", "<u4"),
    ("countIteration", "<u4"), ("hasGrid", "<u4"), ("gridPopcount", "<u4"), ("pointChunks", "<u4"),
    ("voxelChunks", "<u4"), ("gridHash", "<u8"), ("pointsSum", "<u8"), ("pointsXor", "<u8"),
    ("voxelPosSum", "<u8"), ("voxelPosXor", "<u8"), ("childMask", "<u8"), ("name", "u1", 24)])
assert dump_dtype.itemsize == 136


def build(ref=True):
    """(Re)build the oracle libraries; `ref` is skipped silently when /root/reference is absent."""
    subprocess.check_call(["make", "-C", _HERE, "port"] + (["ref"] if ref else []), stdout=subprocess.DEVNULL)


def have_ref():
    return all(os.path.exists(p) for p in _REF.values())


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


_port_lib = None


def port_lib():
    global _port_lib
    if _port_lib is None:
        if not os.path.exists(_PORT):
            build(ref=False)
        lib = ctypes.CDLL(_PORT)
        lib.oracle_create.restype = ctypes.c_void_p
        lib.oracle_create.argtypes = [ctypes.c_uint32]
        lib.oracle_destroy.argtypes = [ctypes.c_void_p]
        lib.oracle_set_trunk_mask.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64]
        lib.oracle_last_error.argtypes = [ctypes.c_void_p]
        lib.oracle_last_error.restype = ctypes.c_int
        lib.oracle_reset.argtypes = [ctypes.c_void_p] * 7
        lib.oracle_construct.argtypes = [ctypes.c_void_p] * 8
        lib.oracle_render.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int]
        lib.oracle_rebase.restype = ctypes.c_int64
        lib.oracle_rebase.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.c_uint64, ctypes.c_uint64]
        lib.oracle_rebase_to.restype = ctypes.c_int64
        lib.oracle_rebase_to.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64] + [ctypes.c_uint64] * 4
        lib.oracle_check_invariants.restype = ctypes.c_int
        lib.oracle_check_invariants.argtypes = [ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_void_p] * 5
        lib.oracle_dump.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.oracle_gather.restype = ctypes.c_uint32
        lib.oracle_gather.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
        lib.ref_call_construct.argtypes = [ctypes.c_void_p] * 11
        lib.ref_call_render.argtypes = [ctypes.c_void_p] * 8
        lib.ref_call_reset.argtypes = [ctypes.c_void_p] * 8
        _port_lib = lib
    return _port_lib


class HostOctree:
    """An octree living in host memory, driven through the reference's three-kernel launch sequence
    (main_progressive_octree.cpp:333-361 reset, :364-428 update, :465-546 render)."""

    def __init__(self, kind="port", *, persistent_bytes=1 << 30, max_nodes=263_157, ring_slots=abi.BATCH_STREAM_SIZE):
        assert kind in ("port", "ref")
        self.kind = kind
        self.lib = port_lib()
        self.max_nodes = max_nodes
        self.nodes = np.zeros(max_nodes, dtype=abi.node_dtype)
        self.persistent = np.zeros(persistent_bytes, dtype=np.uint8)
        self.stats = np.zeros(1, dtype=abi.stats_dtype)
        self.num_uploaded = np.zeros(1, dtype=np.uint32)
        self.batch_sizes = np.zeros(abi.BATCH_STREAM_SIZE, dtype=np.uint32)
        self.ring_slots = ring_slots
        self.ring = np.zeros(ring_slots * abi.MAX_BATCH_SIZE, dtype=abi.point_dtype)
        self.frame_start = np.zeros(1, dtype=np.uint64)
        self.render_buffer = None
        self.ctx = None
        if kind == "port":
            self.ctx = ctypes.c_void_p(self.lib.oracle_create(max_nodes))
        else:
            if not have_ref():
                build(ref=True)
            if not have_ref():
                raise RuntimeError("oracle/_ref is not built and /root/reference is absent")
            self.ref = {k: ctypes.CDLL(p) for k, p in _REF.items()}
            self.momentary = np.zeros(REF_MOMENTARY_BYTES + 4096, dtype=np.uint8)
            self.cudaprint = np.zeros(1024 * 1001, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.oracle_destroy(self.ctx)
            self.ctx = None

    def _fn(self, lib, name):
        return ctypes.cast(getattr(self.ref[lib], name), ctypes.c_void_p)

    def set_trunk_mask(self, lo, hi):
        """EXTENSION of the restatement (the reference is single-GPU): nodes of levels 0-2 that split whatever they hold — the multi-GPU
        layer's shared upper levels (simlod_amd/distributed.trunk_mask; the CPU stand-in for DeviceOctree.set_trunk_mask)."""
        assert self.kind == "port", "the reference's own sources know nothing of ranks"
        self.lib.oracle_set_trunk_mask(self.ctx, ctypes.c_uint64(int(lo)), ctypes.c_uint64(int(hi)))

    # -- launch sequence ------------------------------------------------------------------------------
    def reset(self, uniforms):
        u = np.ascontiguousarray(uniforms).reshape(1)
        if self.kind == "port":
            self.lib.oracle_reset(self.ctx, _ptr(u), _ptr(self.persistent), _ptr(self.nodes), _ptr(self.stats),
                                  _ptr(self.num_uploaded), _ptr(self.batch_sizes))
        else:
            self.lib.ref_call_reset(self._fn("reset", "kernel"), _ptr(u), _ptr(self.persistent), _ptr(self.nodes),
                                    _ptr(self.stats), _ptr(self.cudaprint), _ptr(self.num_uploaded), _ptr(self.batch_sizes))

    def upload(self, points):
        """What the uploader thread does (main_progressive_octree.cpp:1040-1050): copy one batch into the next
        ring slot, publish its size, bump numBatchesUploaded."""
        n = len(points)
        assert n <= abi.MAX_BATCH_SIZE
        idx = int(self.num_uploaded[0])
        slot = idx % abi.BATCH_STREAM_SIZE
        assert slot < self.ring_slots
        self.ring[slot * abi.MAX_BATCH_SIZE: slot * abi.MAX_BATCH_SIZE + n] = points
        self.batch_sizes[slot] = n
        self.num_uploaded[0] = idx + 1

    def construct(self, uniforms):
        u = np.ascontiguousarray(uniforms).reshape(1)
        if self.kind == "port":
            self.lib.oracle_construct(self.ctx, _ptr(u), _ptr(self.ring), _ptr(self.persistent), _ptr(self.nodes),
                                      _ptr(self.stats), _ptr(self.num_uploaded), _ptr(self.batch_sizes))
        else:
            self.lib.ref_call_construct(self._fn("update", "kernel_construct"), _ptr(u), _ptr(self.ring),
                                        _ptr(self.momentary), _ptr(self.persistent), _ptr(self.nodes), _ptr(self.stats),
                                        _ptr(self.frame_start), _ptr(self.cudaprint), _ptr(self.num_uploaded),
                                        _ptr(self.batch_sizes))

    def add_points(self, uniforms, points, batch=abi.MAX_BATCH_SIZE):
        """Upload `points` in batches and run kernel_construct until everything is ingested."""
        for i in range(0, len(points), batch):
            self.upload(points[i:i + batch])
            if (int(self.num_uploaded[0]) - int(self.stats["batchletIndex"][0])) >= min(self.ring_slots, abi.MAX_BATCHES_PER_LAUNCH):
                self.construct(uniforms)
        while int(self.stats["batchletIndex"][0]) < int(self.num_uploaded[0]):
            before = int(self.stats["batchletIndex"][0])
            self.construct(uniforms)
            if int(self.stats["batchletIndex"][0]) == before:
                break

    def select_frame(self, k):
        """Switch between two sets of frame planes (the CPU stand-in for DeviceOctree.select_frame in distributed.render_frames_pipelined)."""
        names = ("_fb", "_color", "_visible", "_depth", "_sums")
        frames = self.__dict__.setdefault("_frames", {})
        frames[self.__dict__.get("_frame", 0)] = {n: getattr(self, n, None) for n in names}
        for n, v in frames.get(k, {}).items():
            setattr(self, n, v)
        self._frame = k

    # -- a frame in parts (oracle_render_part), the CPU stand-in for DeviceOctree in distributed.render_frame ------------------
    def render_part(self, uniforms, part, edl=False):
        import torch
        assert self.kind == "port"
        u = np.ascontiguousarray(uniforms).reshape(1)
        W, H = int(u["width"][0]), int(u["height"][0])
        if part == 0:
            self._fb = np.zeros(W * H, dtype=np.uint64)
            self._color = np.zeros(W * H, dtype=np.uint32)
            self._visible = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
            self._depth = np.zeros(W * H, dtype=np.uint32)
            self._sums = np.zeros(W * H * 4, dtype=np.uint32)
        self.lib.oracle_render_part(self.ctx, _ptr(u), _ptr(self.nodes), _ptr(self.stats), _ptr(self._fb), _ptr(self._color),
                                    _ptr(self._visible), int(bool(edl)), int(part), _ptr(self._depth), _ptr(self._sums))
        self.visible = self._visible[: int(self.stats["numVisibleNodes"][0])]

    def depth_plane(self):
        import torch
        return torch.from_numpy(self._depth.view(np.int32))

    def sum_planes(self):
        import torch
        return torch.from_numpy(self._sums.view(np.int32))

    def framebuffer_words(self):
        import torch
        return torch.from_numpy(self._fb.view(np.int64))

    def visible_records(self):
        import torch
        n = int(self.stats["numVisibleNodes"][0])
        return torch.from_numpy(self._visible.view(np.uint8).reshape(-1)), n

    def render(self, uniforms, edl=False):
        """Returns (fb uint64[H*W] pre-EDL, color uint32[H*W] as written to the surface)."""
        u = np.ascontiguousarray(uniforms).reshape(1)
        W, H = int(u["width"][0]), int(u["height"][0])
        color = np.zeros(W * H, dtype=np.uint32)
        if self.kind == "port":
            fb = np.zeros(W * H, dtype=np.uint64)
            visible = np.zeros(abi.MAX_VISIBLE_NODES, dtype=abi.node_dtype)
            self.lib.oracle_render(self.ctx, _ptr(u), _ptr(self.nodes), _ptr(self.stats), _ptr(fb), _ptr(color),
                                   _ptr(visible), int(bool(edl)))
            self.visible = visible[: int(self.stats["numVisibleNodes"][0])]
            return fb, color
        assert not edl, "oracle/_ref runs with the EDL pass disabled (SURVEY.md H4)"
        need = 15_200_000 + 64 + 32 + 16_000_000 + W * H * 8 + W * H * 20 + 4096
        if self.render_buffer is None or self.render_buffer.size < need:
            self.render_buffer = np.zeros(need, dtype=np.uint8)
        ctypes.c_int.in_dll(self.ref["render"], "simlod_shim_surface_width").value = W
        self.lib.ref_call_render(self._fn("render", "kernel_render"), _ptr(self.render_buffer), _ptr(u), _ptr(self.nodes),
                                 _ptr(color), _ptr(self.stats), _ptr(self.frame_start), _ptr(self.cudaprint))
        # momentary layout of render.cu:1108-1123: visibleNodes, 7 counters (16 B each), Lines (32 B), 1M vertices, fb
        off = 100_000 * 152 + 7 * 16 + 32 + 1_000_000 * 16
        fb = self.render_buffer[off: off + W * H * 8].view(np.uint64).copy()
        self.visible = self.render_buffer[: 100_000 * 152].view(abi.node_dtype)[: int(self.stats["numVisibleNodes"][0])].copy()
        return fb, color

    # -- inspection --------------------------------------------------------------------------------------
    def last_error(self):
        return self.lib.oracle_last_error(self.ctx) if self.kind == "port" else 0

    def dump(self):
        return dump_image(self.nodes, int(self.stats["numNodes"][0]))


def dump_image(nodes, num_nodes):
    """Canonical, order-independent description of a HOST-addressed octree image, sorted by (level,X,Y,Z)."""
    out = np.zeros(num_nodes, dtype=dump_dtype)
    port_lib().oracle_dump(_ptr(nodes), num_nodes, _ptr(out))
    return out[np.argsort(out["key"], kind="stable")]


def rebase_image(nodes, num_nodes, persistent, old_nodes_base, old_persistent_base):
    """Rewrite every pointer of an image copied from (old_nodes_base, old_persistent_base) to where the numpy
    arrays live now."""
    r = port_lib().oracle_rebase(_ptr(nodes), num_nodes, _ptr(persistent), persistent.size, old_nodes_base, old_persistent_base)
    if r < 0:
        raise ValueError("octree image holds a pointer outside its persistent buffer")
    return r


def gather_samples(head_ptr, count):
    out = np.zeros(count, dtype=abi.point_dtype)
    got = port_lib().oracle_gather(ctypes.c_void_p(int(head_ptr)), count, _ptr(out))
    return out[:got]


def rebase_image_to(nodes, num_nodes, persistent, old_nodes_base, old_persistent_base, new_nodes_base, new_persistent_base):
    """Rewrite the pointers of a host-resident image for a foreign address space (e.g. device buffers before an upload)."""
    r = port_lib().oracle_rebase_to(_ptr(nodes), num_nodes, _ptr(persistent), persistent.size, old_nodes_base, old_persistent_base,
                                    new_nodes_base, new_persistent_base)
    if r < 0:
        raise ValueError("octree image holds a pointer outside its persistent buffer")
    return r


_REF_FILTER = os.path.join(_HERE, "_ref", "libref_filter.so")


def have_ref_filter():
    return os.path.exists(_REF_FILTER)


def ref_colorfilter(nodes, num_nodes, uniforms, stats=None):
    """The reference's colour filter (colorfilter.cu `kernel`, compiled in place into oracle/_ref/libref_filter.so) on a HOST-addressed
    octree image, in place: voxel colours become averages, Node.isFiltered is set."""
    lib = ctypes.CDLL(_REF_FILTER)
    port_lib().ref_call_filter.argtypes = [ctypes.c_void_p] * 6
    u = np.ascontiguousarray(uniforms).reshape(1)
    buf = np.zeros(16 << 20, dtype=np.uint8)
    n = np.array([num_nodes], dtype=np.uint32)
    st = np.zeros(1, dtype=abi.stats_dtype) if stats is None else stats
    port_lib().ref_call_filter(ctypes.cast(lib.kernel, ctypes.c_void_p), _ptr(u), _ptr(buf), _ptr(nodes), _ptr(n), _ptr(st))


def check_invariants(nodes, num_nodes, allow_overfull=False):
    """Structural invariants of a HOST-addressed image; returns dict of totals or raises AssertionError(rule).
    allow_overfull: accept leaves above 50 000 points (splits deferred for lack of scratch space)."""
    tot = np.zeros(5, dtype=np.uint64)
    rule = port_lib().oracle_check_invariants_ex(_ptr(nodes), num_nodes, *[ctypes.c_void_p(tot.ctypes.data + 8 * k) for k in range(5)],
                                                 ctypes.c_int(1 if allow_overfull else 0))
    assert rule == 0, f"octree image violates structural rule #{rule} (oracle/oracle_support.c: oracle_check_invariants)"
    return dict(points=int(tot[0]), voxels=int(tot[1]), point_chunks=int(tot[2]), voxel_chunks=int(tot[3]), grids=int(tot[4]))


# ---- loader row: LAS point records -> Points ----------------------------------------------------------
_REF_LAS = os.path.join(_HERE, "_ref", "libref_las.so")


def have_ref_las():
    return os.path.exists(_REF_LAS)


class _RefLasHeader(ctypes.Structure):
    _fields_ = [("versionMajor", ctypes.c_int32), ("versionMinor", ctypes.c_int32), ("headerSize", ctypes.c_uint64),
                ("offsetToPointData", ctypes.c_uint64), ("format", ctypes.c_uint64), ("bytesPerPoint", ctypes.c_uint64),
                ("numPoints", ctypes.c_uint64), ("scale", ctypes.c_double * 3), ("offset", ctypes.c_double * 3),
                ("min", ctypes.c_double * 3), ("max", ctypes.c_double * 3)]


def ref_las_header(path):
    """The reference's own loadHeader (LasLoader.h:21-55) through oracle/_ref/libref_las.so, as a dict."""
    lib = ctypes.CDLL(_REF_LAS)
    h = _RefLasHeader()
    lib.ref_las_header(path.encode(), ctypes.byref(h))
    return {k: (tuple(getattr(h, k)) if k in ("scale", "offset", "min", "max") else int(getattr(h, k))) for k, _ in _RefLasHeader._fields_}


def ref_las_load(path, first, count, translation):
    """The reference's own loadLasNative (LasLoader.cpp:169-227).  Returns `count` Points pre-filled with 0xAA bytes: the
    reference never writes alpha nor, for formats without colour, r/g/b — those bytes come back as it happened to leave them."""
    lib = ctypes.CDLL(_REF_LAS)
    out = np.full(count * 16, 0xAA, dtype=np.uint8)
    t = (ctypes.c_double * 3)(*translation)
    lib.ref_las_load(path.encode(), ctypes.c_uint64(first), ctypes.c_uint64(count), _ptr(out), t)
    return out.view(abi.point_dtype)


def decode_las_port(records, bytes_per_point, fmt, scale, offset):
    """numpy restatement of the parse loop LasLoader.cpp:177-225: `records` = raw bytes, offset = header.offset + translation
    (the caller forms the sum, LasLoader.cpp:197-199).  fp64 multiply, fp64 add, round to fp32; colour channel c > 255 ? c / 256 : c
    for formats 2, 3, 5, 7.  Undefined-in-the-reference bytes: alpha = 255, colourless formats r = g = b = 0."""
    rec = np.ascontiguousarray(records, dtype=np.uint8).reshape(-1, bytes_per_point)
    n = len(rec)
    out = np.zeros(n, dtype=abi.point_dtype)
    xyz = np.ascontiguousarray(rec[:, 0:12]).view("<i4").reshape(n, 3).astype(np.float64)
    for k, name in enumerate("xyz"):
        out[name] = (xyz[:, k] * np.float64(scale[k]) + np.float64(offset[k])).astype(np.float32)
    rgb_off = 0
    if fmt == 2:
        rgb_off = 20
    elif fmt == 3:
        rgb_off = 28
    if fmt == 5:
        rgb_off = 28
    if fmt == 7:
        rgb_off = 30
    color = np.full(n, 0xff000000, dtype=np.uint32)
    if rgb_off > 0:
        c = np.ascontiguousarray(rec[:, rgb_off:rgb_off + 6]).view("<u2").reshape(n, 3).astype(np.uint32)
        c = np.where(c > 255, c // 256, c)
        color |= c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16)
    out["color"] = color
    return out
