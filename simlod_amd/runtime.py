"""Host side of the MI355X SimLOD hot paths: a headless mirror of what the reference host does around its three
kernels (modules/progressive_octree/main_progressive_octree.cpp).

    initCudaProgram()   :549-642   device buffers (nodes 40 MB, momentary 300 MB, render 200 MB, ring 50 x 16 MB, ...)
    resetCUDA()         :333-361   launch `kernel`
    uploader thread     :1040-1050 H2D copy of one batch into the ring, publish batchSizes[slot], numBatchesUploaded
    updateOctree()      :364-428   launch `kernel_construct`
    renderCUDA()        :465-546   launch `kernel_render`
    stats readback      :1201      D2H copy of Stats

Everything device-side goes through the C ABI of libsimlod_hip.so (include/simlod_hip.h); torch is used for device
memory and streams only.  There is NO CPU fallback: without the compiled library or without a GPU this module
raises.
"""
import ctypes
import os

import numpy as np
import torch

from . import abi

# SIMLOD_HIP_LIB: another build of the same library (A/B measurements of two kernel variants on one GPU box, tools/)
_LIB_PATH = os.environ.get("SIMLOD_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsimlod_hip.so")
_lib = None


class SimlodError(RuntimeError):
    pass


def lib():
    """Load libsimlod_hip.so (built by `make -C simlod_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise SimlodError(f"{_LIB_PATH} is missing: build it with `make -C simlod_amd/csrc` (no fallback path exists)")
        L = ctypes.CDLL(_LIB_PATH)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.simlod_set_node_capacity.argtypes = [u32]
        L.simlod_set_ingest_mode.argtypes = [u32]
        L.simlod_set_construct_batch_limit.argtypes = [u32]
        L.simlod_octree_image_replaced.argtypes = [vp]
        L.simlod_context_create.argtypes = [ctypes.POINTER(vp)]
        L.simlod_context_destroy.argtypes = [vp]
        L.simlod_context_attach.argtypes = [vp, vp]
        L.simlod_context_set_node_capacity.argtypes = [vp, u32]
        L.simlod_context_set_ingest_mode.argtypes = [vp, u32]
        L.simlod_context_set_construct_batch_limit.argtypes = [vp, u32]
        L.simlod_context_hint_pending_batches.argtypes = [vp, u32]
        L.simlod_upload_counter_written.argtypes = [vp, u32]
        L.simlod_context_set_knob.argtypes = [vp, ctypes.c_char_p, ctypes.c_int, ctypes.c_int]
        L.simlod_context_reload_env.argtypes = [vp]
        L.simlod_context_set_trunk_mask.argtypes = [vp, u64, u64]
        L.simlod_context_construct_buffer_min_bytes.restype = u64
        L.simlod_context_construct_buffer_min_bytes.argtypes = [vp]
        L.simlod_render_framebuffer_offset.restype = u64
        L.simlod_render_buffer_bytes.restype = u64
        L.simlod_render_buffer_bytes.argtypes = [u32, u32]
        L.simlod_construct_buffer_min_bytes.restype = u64
        L.simlod_launch_reset.argtypes = [vp] * 8
        L.simlod_launch_construct.argtypes = [vp] * 11
        L.simlod_launch_render.argtypes = [vp] * 8
        L.simlod_launch_render_part.argtypes = [ctypes.c_uint32] + [vp] * 8
        L.simlod_render_frame_composed.argtypes = [vp] * 10
        L.simlod_render_frame_rccl.argtypes = [vp] * 9
        L.simlod_render_depth_plane_offset.restype = u64
        L.simlod_render_depth_plane_offset.argtypes = [u32, u32]
        L.simlod_render_sum_planes_offset.restype = u64
        L.simlod_render_sum_planes_offset.argtypes = [u32, u32]
        L.simlod_program_create.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_char_p), ctypes.c_int,
                                            ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
        L.simlod_program_destroy.argtypes = [vp]
        L.simlod_program_kernel.restype = vp
        L.simlod_program_kernel.argtypes = [vp, ctypes.c_char_p]
        L.simlod_function_max_active_blocks.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
        L.simlod_launch_cooperative.argtypes = [vp] + [ctypes.c_uint] * 7 + [vp, ctypes.POINTER(vp)]
        L.simlod_build_info.restype = ctypes.c_char_p
        L.simlod_decode_las.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double),
                                        ctypes.POINTER(ctypes.c_double), vp, vp]
        L.simlod_launch_colorfilter.argtypes = [vp] * 6
        L.simlod_colorfilter_buffer_min_bytes.restype = u64
        L.simlod_generate_terrain.argtypes = [vp, u64, u64, u64, u32, u32, ctypes.POINTER(ctypes.c_float), vp]
        L.simlod_generate_terrain_scan.argtypes = [vp, u64, u64, u64, u32, u32, ctypes.POINTER(ctypes.c_float), ctypes.c_float, vp]
        _lib = L
    return _lib


EXPORTED_SYMBOLS = [
    "simlod_set_node_capacity", "simlod_render_framebuffer_offset", "simlod_render_buffer_bytes",
    "simlod_construct_buffer_min_bytes", "simlod_launch_reset", "simlod_launch_construct", "simlod_launch_render",
    "simlod_program_create", "simlod_program_destroy", "simlod_program_kernel", "simlod_function_max_active_blocks",
    "simlod_launch_cooperative", "simlod_build_info", "simlod_decode_las", "simlod_launch_render_part",
    "simlod_render_depth_plane_offset", "simlod_render_sum_planes_offset", "simlod_set_ingest_mode", "simlod_set_construct_batch_limit",
    "simlod_context_create", "simlod_context_destroy", "simlod_context_attach", "simlod_context_set_node_capacity", "simlod_context_set_ingest_mode",
    "simlod_context_set_construct_batch_limit", "simlod_context_set_knob", "simlod_context_reload_env", "simlod_context_construct_buffer_min_bytes",
    "simlod_octree_image_replaced", "simlod_render_frame_composed", "simlod_render_frame_rccl", "simlod_context_set_trunk_mask", "simlod_rccl_version",
    "simlod_context_hint_pending_batches", "simlod_upload_counter_written",
    "simlod_profile_enable", "simlod_profile_collect", "simlod_generate_terrain", "simlod_generate_terrain_scan", "simlod_launch_colorfilter", "simlod_colorfilter_buffer_min_bytes",
]


def _check(code, what):
    if code != 0:
        raise SimlodError(f"{what} failed with hipError {code}")


class Program:
    """CudaModularProgram look-alike (include/CudaModularProgram.h:140-264): modules -> kernels[name]."""

    def __init__(self, modules, kernels):
        L = lib()
        self._h = ctypes.c_void_p()
        m = (ctypes.c_char_p * len(modules))(*[s.encode() for s in modules])
        k = (ctypes.c_char_p * len(kernels))(*[s.encode() for s in kernels])
        _check(L.simlod_program_create(ctypes.byref(self._h), m, len(modules), k, len(kernels)), "simlod_program_create")
        self.kernels = {name: L.simlod_program_kernel(self._h, name.encode()) for name in kernels}

    def __del__(self):
        if getattr(self, "_h", None):
            lib().simlod_program_destroy(self._h)
            self._h = None


class DeviceOctree:
    """Device buffers of initCudaProgram() + the three launches, on one GPU."""

    def __init__(self, device="cuda:0", *, persistent_bytes=4 << 30, momentary_bytes=300_000_000, max_nodes=263_157,
                 ring_slots=abi.BATCH_STREAM_SIZE, max_pixels=1920 * 1080, coalesce=False, sizes_launches=True):
        if not torch.cuda.is_available():
            raise SimlodError("no GPU visible: the SimLOD hot paths only exist as gfx950 kernels")
        self.L = lib()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.max_nodes = max_nodes
        # this octree's own context (include/simlod_hip.h): node capacity, ingest granularity — exact = the reference's batch-by-batch
        # bookkeeping; coalesced = all pending batches of a launch as one (same octree content, different allocator / chunk-pool
        # counters) —, batch limit, tuning knobs (from the environment as it is NOW), second stream; launches find it by the node array
        ctx = ctypes.c_void_p()
        _check(self.L.simlod_context_create(ctypes.byref(ctx)), "simlod_context_create")
        self.ctx = ctx
        _check(self.L.simlod_context_set_node_capacity(ctx, max_nodes), "simlod_context_set_node_capacity")
        _check(self.L.simlod_context_set_ingest_mode(ctx, 1 if coalesce else 0), "simlod_context_set_ingest_mode")
        self.batch_limit = abi.MAX_BATCHES_PER_LAUNCH
        # How many batches a launch can find.  Every write of the upload counter goes past the library (publish -> simlod_upload_counter_written), as
        # shim/cuda.h does with the cuMemsetD32Async of the reference's uploader (main_progressive_octree.cpp:1047-1050): the library sizes its launches
        # by that and by what its earlier launches reported — the unchanged reference host gets the same.  sizes_launches=False: a host that tells
        # nothing (the library predicts from its launches' reports alone).  SIMLOD_HOST_HINT=1: drain() / stream() also say how many batches are
        # pending in front of every launch (simlod_context_hint_pending_batches): the two must give the same rates (bench.py reports both).
        self.notifies = sizes_launches
        self.hint_pending = sizes_launches and os.environ.get("SIMLOD_HOST_HINT", "0") == "1"
        z = dict(dtype=torch.uint8, device=self.device)
        # H11: the reference renders before any reset and relies on fresh VRAM reading as zero
        self.nodes = torch.zeros(max_nodes * 152, **z)
        _check(self.L.simlod_context_attach(ctx, self._p(self.nodes)), "simlod_context_attach")
        self.stats = torch.zeros(112, **z)
        self.persistent = torch.empty(persistent_bytes, **z)
        self.persistent[:1 << 20].zero_()
        self.momentary = torch.empty(momentary_bytes, **z)
        self.momentary[:1 << 20].zero_()
        self.ring_slots = ring_slots
        self.ring = torch.empty(ring_slots * abi.MAX_BATCH_SIZE * 16, **z)
        self.batch_sizes = torch.zeros(abi.BATCH_STREAM_SIZE, dtype=torch.int32, device=self.device)
        self.num_uploaded = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.frame_start = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.render_buffer = torch.empty(int(self.L.simlod_render_buffer_bytes(max_pixels, 1)), **z)
        self.colorbuffer = torch.zeros(max_pixels, dtype=torch.int32, device=self.device)
        self.las_stage = None
        self.persistent_bytes, self.momentary_bytes = persistent_bytes, momentary_bytes
        self.uploaded_host = 0
        self.processed_host = 0
        self.upload_stream = torch.cuda.Stream(device=self.device)

    def close(self):
        """Give the context back (waits for its second stream).  The buffers go with the object."""
        ctx, self.ctx = getattr(self, "ctx", None), None
        if ctx is not None and torch.cuda.is_available():
            torch.cuda.synchronize(self.device)
            self.L.simlod_context_attach(None, self._p(self.nodes))
            self.L.simlod_context_destroy(ctx)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tune(self, name, value=None):
        """Override one tuning knob of THIS octree's context (name as in the environment, e.g. "SIMLOD_OVERLAP_TAIL"); None: built-in default."""
        _check(self.L.simlod_context_set_knob(self.ctx, name.encode(), int(value or 0), 0 if value is None else 1), f"simlod_context_set_knob({name})")

    def reload_env(self):
        """Read the tuning knobs from the environment again (they are read once, when the context is made)."""
        _check(self.L.simlod_context_reload_env(self.ctx), "simlod_context_reload_env")

    def set_trunk_mask(self, lo, hi):
        """Multi-GPU jobs: the nodes of levels 0-2 that split whatever this rank holds under them (simlod_context_set_trunk_mask; the mask
        comes from distributed.trunk_mask).  Takes effect with the next batch ingested — flush_trunk() ingests an empty one."""
        _check(self.L.simlod_context_set_trunk_mask(self.ctx, ctypes.c_uint64(int(lo)), ctypes.c_uint64(int(hi))), "simlod_context_set_trunk_mask")

    def flush_trunk(self, uniforms):
        """Apply a trunk mask set (or widened) after the last batch: a batch of zero points goes through the ring."""
        self.upload(np.zeros(0, dtype=abi.point_dtype))
        self.drain(uniforms)

    def set_batch_limit(self, max_batches):
        self.batch_limit = max(1, min(int(max_batches), abi.MAX_BATCHES_PER_LAUNCH))
        _check(self.L.simlod_context_set_construct_batch_limit(self.ctx, max_batches), "simlod_context_set_construct_batch_limit")

    def _hint(self, pending=None):
        """Tell the library how many batches are pending for the next launch (SIMLOD_HOST_HINT=1 only; None: nothing to say)."""
        if self.hint_pending and pending is not None:
            _check(self.L.simlod_context_hint_pending_batches(self.ctx, max(0, min(int(pending), self.batch_limit))), "simlod_context_hint_pending_batches")

    # -- helpers ---------------------------------------------------------------------------------------------
    def uniforms(self, width, height, transform, box_size, **kw):
        return abi.make_uniforms(width, height, transform, box_size, persistent_capacity=self.persistent_bytes,
                                 momentary_capacity=self.momentary_bytes, **kw)

    @staticmethod
    def _u(uniforms):
        u = np.ascontiguousarray(uniforms).reshape(1)
        return u, ctypes.c_void_p(u.ctypes.data)

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _p(self, t):
        return ctypes.c_void_p(t.data_ptr())

    # -- the launch surface ------------------------------------------------------------------------------------
    def reset(self, uniforms):
        u, up = self._u(uniforms)
        _check(self.L.simlod_launch_reset(up, self._p(self.persistent), self._p(self.nodes), self._p(self.stats), None,
                                          self._p(self.num_uploaded), self._p(self.batch_sizes), self._stream()), "reset")
        self.uploaded_host = 0
        self.processed_host = 0

    def upload(self, points, *, stream=None):
        """One batch into the next ring slot (host numpy array or device tensor of 16-byte records)."""
        n = len(points)
        assert n <= abi.MAX_BATCH_SIZE
        slot = self.uploaded_host % abi.BATCH_STREAM_SIZE
        assert slot < self.ring_slots, "ring smaller than BATCH_STREAM_SIZE: consume before uploading more"
        dst = self.ring[slot * abi.MAX_BATCH_SIZE * 16: slot * abi.MAX_BATCH_SIZE * 16 + n * 16]
        if isinstance(points, torch.Tensor):
            dst.copy_(points.reshape(-1).view(torch.uint8), non_blocking=True)
        else:
            dst.copy_(torch.from_numpy(np.ascontiguousarray(points).view(np.uint8).reshape(-1)), non_blocking=True)
        self.batch_sizes[slot] = n
        self.uploaded_host += 1
        self.publish(self.uploaded_host)

    def publish(self, num_batches):
        """The upload counter: `num_batches` ring batches are complete (in stream order behind their copies and their batchSizes entries), and the
        library is told so (main_progressive_octree.cpp:1047-1050 + shim/cuda.h: cuMemsetD32Async(cptr_numBatchesUploaded, ...))."""
        self.num_uploaded.fill_(num_batches)
        if self.notifies:
            _check(self.L.simlod_upload_counter_written(self._p(self.num_uploaded), int(num_batches)), "simlod_upload_counter_written")

    def upload_las(self, records, header, translation):
        """One batch of RAW LAS point records (uint8 host array or device tensor) into the next ring slot, decoded on the
        device (simlod_decode_las); translation = -boxMin as at main_progressive_octree.cpp:868."""
        from . import lasio
        bpp = int(header.bytesPerPoint)
        raw = records if isinstance(records, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(records).reshape(-1))
        n = raw.numel() // bpp
        assert n <= abi.MAX_BATCH_SIZE
        slot = self.uploaded_host % abi.BATCH_STREAM_SIZE
        assert slot < self.ring_slots, "ring smaller than BATCH_STREAM_SIZE: consume before uploading more"
        if self.las_stage is None or self.las_stage.numel() < raw.numel():
            self.las_stage = torch.empty(max(raw.numel(), 1 << 20), dtype=torch.uint8, device=self.device)
        self.las_stage[: raw.numel()].copy_(raw.reshape(-1).view(torch.uint8), non_blocking=True)
        scale = (ctypes.c_double * 3)(*header.scale)
        offset = (ctypes.c_double * 3)(*lasio.decode_offset(header, translation))
        dst = self.ring.data_ptr() + slot * abi.MAX_BATCH_SIZE * 16
        _check(self.L.simlod_decode_las(self._p(self.las_stage), ctypes.c_uint64(n), ctypes.c_uint32(bpp), ctypes.c_uint32(int(header.format)),
                                        scale, offset, ctypes.c_void_p(dst), self._stream()), "simlod_decode_las")
        self.batch_sizes[slot] = n
        self.uploaded_host += 1
        self.publish(self.uploaded_host)

    def add_las(self, uniforms, path, translation=None, batch=abi.MAX_BATCH_SIZE):
        """Stream a LAS file through the ring: read raw bytes, decode on the device, ingest."""
        from . import lasio
        h = lasio.load_header(path)
        t = tuple(-float(v) for v in h.min) if translation is None else translation
        for first, count in lasio.batches(h, batch):
            if self.uploaded_host - self.processed_host >= self.ring_slots:
                self.drain(uniforms)                   # (sets processed_host to what it read back)
                assert self.uploaded_host - self.processed_host < self.ring_slots, "kernel_construct left the ring full"
            self.upload_las(lasio.read_records(path, h, first, count), h, t)
        self.drain(uniforms)
        return h

    def colorfilter(self, uniforms):
        """colorfilter.cu's `kernel`: average voxel colours, bottom-up (simlod_launch_colorfilter); the momentary buffer is its scratch."""
        # scratch = the render buffer, not kernel_construct's momentary buffer: the builder's recycle stack lives there and has to survive
        # (the reference, whose filter call is dead code, would run it on the momentary buffer after the last batch only)
        uu = np.array(uniforms, copy=True)
        need = int(self.L.simlod_colorfilter_buffer_min_bytes())
        if self.render_buffer.numel() < need:
            self.render_buffer = torch.empty(need, dtype=torch.uint8, device=self.device)
        uu["momentaryBufferCapacity"] = self.render_buffer.numel()
        u, up = self._u(uu)
        _check(self.L.simlod_launch_colorfilter(up, self._p(self.render_buffer), self._p(self.nodes), None, self._p(self.stats), self._stream()), "colorfilter kernel")

    def generate_terrain(self, out, first_index, points_per_tile, seed, tiles_x, tile_extent, swath_width=0.0):
        """BASELINE config 4's input made on the device: points first_index .. first_index + len(out)/16 - 1 of the tiled-terrain stream
        into the uint8 device tensor `out` (simlod_generate_terrain; with swath_width > 0: in flight lines of that width, simlod_generate_terrain_scan)."""
        n = out.numel() // 16
        ext = (ctypes.c_float * 3)(*[float(v) for v in tile_extent])
        _check(self.L.simlod_generate_terrain_scan(self._p(out), ctypes.c_uint64(n), ctypes.c_uint64(first_index), ctypes.c_uint64(points_per_tile),
                                                   ctypes.c_uint32(seed), ctypes.c_uint32(tiles_x), ext, ctypes.c_float(swath_width), self._stream()), "simlod_generate_terrain_scan")

    def construct(self, uniforms):
        u, up = self._u(uniforms)
        _check(self.L.simlod_launch_construct(up, self._p(self.ring), self._p(self.momentary), self._p(self.persistent),
                                              self._p(self.nodes), self._p(self.stats), self._p(self.frame_start), None,
                                              self._p(self.num_uploaded), self._p(self.batch_sizes), self._stream()), "kernel_construct")

    def render(self, uniforms):
        u, up = self._u(uniforms)
        W, H = int(u["width"][0]), int(u["height"][0])
        need = int(self.L.simlod_render_buffer_bytes(W, H))
        if self.render_buffer.numel() < need:
            self.render_buffer = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self.colorbuffer.numel() < W * H:
            self.colorbuffer = torch.zeros(W * H, dtype=torch.int32, device=self.device)
        _check(self.L.simlod_launch_render(self._p(self.render_buffer), up, self._p(self.nodes), self._p(self.colorbuffer),
                                           self._p(self.stats), self._p(self.frame_start), None, self._stream()), "kernel_render")
        return W, H

    def select_frame(self, k):
        """Switch between render buffers (distributed.render_frames_pipelined keeps two frames in flight: while the planes of one are being
        reduced across ranks, the next is rasterised into the other buffer).  Buffer 0 is the one the object was made with."""
        frames = self.__dict__.setdefault("_frames", {0: None})
        cur = self.__dict__.get("_frame", 0)
        frames[cur] = (self.render_buffer, self.colorbuffer, getattr(self, "_frame_size", None))
        if frames.get(k) is None:
            frames[k] = (torch.empty_like(self.render_buffer), torch.zeros_like(self.colorbuffer), None)
        self.render_buffer, self.colorbuffer, self._frame_size = frames[k]
        self._frame = k

    # -- a frame in parts, for composition across GPUs (simlod_launch_render_part; driven by distributed.render_frame) ----------
    def render_part(self, uniforms, part):
        u, up = self._u(uniforms)
        W, H = int(u["width"][0]), int(u["height"][0])
        need = int(self.L.simlod_render_buffer_bytes(W, H))
        if self.render_buffer.numel() < need:
            self.render_buffer = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self.colorbuffer.numel() < W * H:
            self.colorbuffer = torch.zeros(W * H, dtype=torch.int32, device=self.device)
        _check(self.L.simlod_launch_render_part(ctypes.c_uint32(part), self._p(self.render_buffer), up, self._p(self.nodes), self._p(self.colorbuffer),
                                                self._p(self.stats), self._p(self.frame_start), None, self._stream()), "kernel_render part")
        self._frame_size = (W, H)

    REDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p)

    def render_composed(self, uniforms, reduce=None, rccl_comm=None):
        """One frame through simlod_render_frame_composed / simlod_render_frame_rccl (include/simlod_hip.h): the four parts of kernel_render with
        a reduction of the named plane over all ranks in between — `reduce(plane, data_ptr, count, elem_bytes, op, stream) -> int`, or an
        ncclComm_t as an integer address."""
        u, up = self._u(uniforms)
        W, H = int(u["width"][0]), int(u["height"][0])
        need = int(self.L.simlod_render_buffer_bytes(W, H))
        if self.render_buffer.numel() < need:
            self.render_buffer = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self.colorbuffer.numel() < W * H:
            self.colorbuffer = torch.zeros(W * H, dtype=torch.int32, device=self.device)
        args = [self._p(self.render_buffer), up, self._p(self.nodes), self._p(self.colorbuffer), self._p(self.stats), self._p(self.frame_start), None, self._stream()]
        if rccl_comm is not None:
            _check(self.L.simlod_render_frame_rccl(*args, ctypes.c_void_p(rccl_comm)), "simlod_render_frame_rccl")
        else:
            cb = self.REDUCE_FN(lambda user, plane, data, count, eb, op, stream: int(reduce(plane, data, count, eb, op, stream))) if reduce is not None else None
            _check(self.L.simlod_render_frame_composed(*args, ctypes.cast(cb, ctypes.c_void_p) if cb is not None else None, None), "simlod_render_frame_composed")
        self._frame_size = (W, H)

    def depth_plane(self):
        """The HQS depth plane as an int32 view (positive float bits: integer MIN == float MIN)."""
        W, H = self._frame_size
        off = int(self.L.simlod_render_depth_plane_offset(W, H))
        return self.render_buffer[off: off + W * H * 4].view(torch.int32)

    def sum_planes(self):
        """{R, G, B, count} per pixel as an int32 view."""
        W, H = self._frame_size
        off = int(self.L.simlod_render_sum_planes_offset(W, H))
        return self.render_buffer[off: off + W * H * 16].view(torch.int32)

    def framebuffer_words(self):
        """depth|colour words as an int64 view (the sign bit is never set)."""
        W, H = self._frame_size
        off = int(self.L.simlod_render_framebuffer_offset())
        return self.render_buffer[off: off + W * H * 8].view(torch.int64)

    def visible_records(self):
        """(bytes of the visible-node array, number of visible nodes of the last frame as a device tensor — no host sync)."""
        off = abi.stats_dtype.fields["numVisibleNodes"][1]
        return self.render_buffer, self.stats[off: off + 4].view(torch.int32)

    def visible_records_early(self):
        """As visible_records, but the count comes from the frame's own counter (valid once part 0 of the frame has run, clamped by the
        consumer): the all-gather of the visible nodes can be issued right behind part 0 and overlap the rest of the frame."""
        off = abi.MAX_VISIBLE_NODES * abi.node_dtype.itemsize
        return self.render_buffer, self.render_buffer[off: off + 4].view(torch.int32)

    def lists_read_through_table(self):
        """How many chunk lists the last frame's r_visible read through the builder's chunk table instead of chasing `next`
        (render.hip: counter 5 of the frame counters behind the visible-node array)."""
        off = abi.MAX_VISIBLE_NODES * abi.node_dtype.itemsize + 5 * 16
        return int(self.render_buffer[off: off + 4].view(torch.int32).item())

    def samples_outside_tiles(self):
        """How many samples the last frame's first draw pass sent down the global-atomic path — outside their draw item's LDS tile, or drawn
        without one (render.hip: counter 6 of the frame counters behind the visible-node array)."""
        off = abi.MAX_VISIBLE_NODES * abi.node_dtype.itemsize + 6 * 16
        return int(self.render_buffer[off: off + 4].view(torch.int32).item())

    def samples_binned(self, width, height):
        """How many samples the last frame's first draw pass sorted into the screen bins (render.hip r_overflow: word 13 of the frame's work
        area, behind the framebuffer plane)."""
        off = int(self.L.simlod_render_framebuffer_offset()) + (width * height * 8 + 15) // 16 * 16 + 13 * 4
        return int(self.render_buffer[off: off + 4].view(torch.int32).item())

    # -- readback ------------------------------------------------------------------------------------------------
    def read_stats(self):
        return self.stats.cpu().numpy().view(abi.stats_dtype)[0].copy()

    def framebuffer(self, W, H):
        off = int(self.L.simlod_render_framebuffer_offset())
        return self.render_buffer[off: off + W * H * 8].cpu().numpy().view(np.uint64).copy()

    def color(self, W, H):
        return self.colorbuffer[: W * H].cpu().numpy().view(np.uint32).copy()

    def processed(self):
        """Stats.batchletIndex as the host sees it after the stream drained (main_progressive_octree.cpp:1201-1216)."""
        torch.cuda.current_stream().synchronize()
        self.processed_host = int(self.stats[76:80].cpu().numpy().view(np.uint32)[0])
        return self.processed_host

    def drain(self, uniforms, max_launches=1000):
        """Launch kernel_construct until every uploaded batch is ingested.  One launch takes at most 20 batches and stops
        early once it has run for 10 ms (progressive_octree_voxels.cu:883, :939-949) — the reference host simply launches
        again next frame; so does this loop.  Returns the number of launches."""
        launches = 0
        idle = 0
        before = self.processed_host              # (what the host last saw: 0 after a reset; too low only costs a launch that exits at once)
        while before < self.uploaded_host and launches < max_launches:
            # as many launches as the pending batches need at 20 per launch, enqueued back to back (the reference's frame loop does not wait
            # for a launch either before it enqueues the next frame's); then ONE look at Stats
            need = -(-(self.uploaded_host - before) // self.batch_limit)
            for k in range(need):
                self._hint(self.uploaded_host - before - k * self.batch_limit)
                self.construct(uniforms)
            self._hint()
            launches += need
            after = self.processed()
            idle, before = (idle + 1 if after == before else 0), after
            # (one round without progress is no stall: a library that sizes its launches by what earlier launches REPORTED may enqueue nothing until
            # the first report of a new burst is in — include/simlod_hip.h, launch sizing)
            if idle >= 3:
                st = self.read_stats()
                raise SimlodError(f"kernel_construct made no progress (Stats.dbg={int(st['dbg']):#x}, "
                                  f"memCapacityReached={int(st['memCapacityReached'])})")
        return launches

    def stream(self, uniforms, source, num_points, batch=abi.MAX_BATCH_SIZE):
        """Feed `num_points` 16-byte records that are RESIDENT on the device (`source`: uint8 tensor) through the ring the way the reference
        host does (main_progressive_octree.cpp:1005-1050, :364-428): an uploader fills free ring slots on its own stream — copy, then
        batchSizes[slot], then numBatchesUploaded, in stream order — but never runs more than a ring ahead of what the host has seen
        processed (back-pressure, :1012); every frame launches kernel_construct once (<= 20 batches, <= 10 ms) and reads Stats back.
        Returns the number of launches.  Arbitrarily long inputs go through 50 slots."""
        nb = (num_points + batch - 1) // batch
        if self.processed() < self.uploaded_host:      # batches uploaded earlier and not yet ingested: the back-pressure below counts from an empty ring
            self.drain(uniforms)
            assert self.processed_host == self.uploaded_host, "kernel_construct left batches pending"
        base = self.uploaded_host
        src = source.reshape(-1)
        uploaded, processed, launches, stalls = 0, 0, 0, 0

        def top_up():
            # the uploader: runs of free ring slots, each run one copy (a run ends where the ring wraps or the batches stop being full), then the
            # run's batchSizes, then the counter — in stream order on the upload stream, as main_progressive_octree.cpp:1020-1050 publishes them
            nonlocal uploaded
            if uploaded >= nb or uploaded - processed >= self.ring_slots:
                return
            with torch.cuda.stream(self.upload_stream):
                while uploaded < nb and uploaded - processed < self.ring_slots:
                    slot = (base + uploaded) % abi.BATCH_STREAM_SIZE
                    assert slot < self.ring_slots, "ring smaller than BATCH_STREAM_SIZE"
                    run = min(nb - uploaded, self.ring_slots - (uploaded - processed), abi.BATCH_STREAM_SIZE - slot, self.ring_slots - slot)
                    if batch != abi.MAX_BATCH_SIZE or (uploaded + run) * batch > num_points:
                        run = 1                   # (short batches do not lie back to back in the ring)
                    n = min(run * batch, num_points - uploaded * batch)
                    self.ring[slot * abi.MAX_BATCH_SIZE * 16: slot * abi.MAX_BATCH_SIZE * 16 + n * 16].copy_(src[uploaded * batch * 16: uploaded * batch * 16 + n * 16], non_blocking=True)
                    self.batch_sizes[slot: slot + run].fill_(min(batch, n))
                    uploaded += run
                    self.publish(base + uploaded)

        self.upload_stream.wait_stream(torch.cuda.current_stream())     # (a reset enqueued just before zeroes batchSizes: the uploader starts behind it)
        top_up()
        while processed < nb:
            self.uploaded_host = base + uploaded
            self._hint(uploaded - processed)
            self.construct(uniforms)              # takes what has been published by now (k_begin reads the counter with a device-scope load)
            launches += 1
            top_up()                              # ... and the uploader refills behind it while it runs
            before, processed = processed, self.processed() - base
            if processed != before:
                stalls = 0
                continue
            # nothing taken: either nothing had been published yet (the uploader is behind: wait for it, once) or the builder refuses
            stalls += 1
            self.upload_stream.synchronize()
            if stalls > 3:
                st = self.read_stats()
                raise SimlodError(f"kernel_construct made no progress (Stats.dbg={int(st['dbg']):#x}, memCapacityReached={int(st['memCapacityReached'])})")
        self._hint()
        self.uploaded_host = base + uploaded
        self.processed_host = self.uploaded_host
        return launches

    def add_points(self, uniforms, points, batch=abi.MAX_BATCH_SIZE):
        """Upload `points` batch by batch; ingest whenever the ring would overflow and at the end."""
        for i in range(0, len(points), batch):
            if self.uploaded_host - self.processed_host >= self.ring_slots:
                self.drain(uniforms)
                assert self.uploaded_host - self.processed_host < self.ring_slots, "kernel_construct left the ring full"
            self.upload(points[i:i + batch])
        self.drain(uniforms)

    def upload_image(self, nodes, persistent, num_nodes, stats=None):
        """Load an octree image whose pointers were already rewritten for THIS object's device buffers (self.nodes.data_ptr(),
        self.persistent.data_ptr()) — e.g. one built by another implementation.  Only what kernel_render needs is set."""
        assert num_nodes <= self.max_nodes and persistent.size <= self.persistent.numel()
        self.nodes[: num_nodes * 152].copy_(torch.from_numpy(nodes[:num_nodes].view(np.uint8).reshape(-1)))
        self.persistent[: persistent.size].copy_(torch.from_numpy(persistent))
        st = np.zeros(1, dtype=abi.stats_dtype) if stats is None else np.array(stats, dtype=abi.stats_dtype).reshape(1).copy()
        st["numNodes"] = num_nodes
        self.stats.copy_(torch.from_numpy(st.view(np.uint8).reshape(-1)))
        self.momentary[:4096].zero_()          # control block: the builder's side tables describe the previous octree
        _check(self.L.simlod_octree_image_replaced(self.nodes.data_ptr()), "simlod_octree_image_replaced")
        self.processed_host = int(st["batchletIndex"][0])

    def download_image(self):
        """(nodes, persistent, numNodes, device base addresses) — the octree image as host arrays, pointers untouched."""
        torch.cuda.synchronize(self.device)
        stats = self.read_stats()
        used = int(stats["allocatedBytes_persistent"])
        if used == 0:   # before the first construct the stats pass has not run: read the allocator header
            used = int(self.persistent[8:16].cpu().numpy().view(np.uint64)[0])
        n = int(stats["numNodes"])
        nodes = self.nodes[: n * 152].cpu().numpy().view(abi.node_dtype).copy()
        pers = self.persistent[:used].cpu().numpy().copy()
        return nodes, pers, n, self.nodes.data_ptr(), self.persistent.data_ptr()
