"""The C-ABI library loads without a GPU and exports every symbol include/simlod_hip.h declares; the Program surface
(no device work) behaves like CudaModularProgram's name lookup."""
import ctypes
import os
import re

import numpy as np
import pytest

from simlod_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "simlod_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(simlod_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_libs):
    from simlod_amd import runtime
    L = runtime.lib()
    names = declared_symbols()
    assert len(names) >= 13
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/simlod_hip.h but not exported"
    assert set(runtime.EXPORTED_SYMBOLS) <= set(names)
    assert b"gfx950" in L.simlod_build_info()


def test_layout_queries(built_libs):
    from simlod_amd import runtime
    L = runtime.lib()
    # the framebuffer sits where the reference's bump allocator puts it (render.cu:1108-1123)
    assert L.simlod_render_framebuffer_offset() == 100_000 * 152 + 7 * 16 + 32 + 16_000_000
    assert L.simlod_render_buffer_bytes(1920, 1080) <= 200_000_000          # fits the host's cptr_renderbuffer (main.cpp:555)
    assert L.simlod_construct_buffer_min_bytes() <= 300_000_000             # fits the host's 300 MB cptr_buffer (main.cpp:554)


def test_program_surface_mirrors_cuda_modular_program(built_libs):
    from simlod_amd.runtime import Program, SimlodError
    upd = Program(["./modules/progressive_octree/progressive_octree_voxels.cu", "./modules/progressive_octree/utils.cu"], ["kernel_construct"])
    assert upd.kernels["kernel_construct"]
    rst = Program(["./modules/progressive_octree/reset.cu", "./modules/progressive_octree/utils.cu"], ["kernel"])
    assert rst.kernels["kernel"]
    rnd = Program(["./modules/progressive_octree/render.cu", "./modules/progressive_octree/utils.cu"], ["kernel_render"])
    assert rnd.kernels["kernel_render"]
    with pytest.raises(SimlodError):
        Program(["./modules/progressive_octree/render.cu"], ["kernel_construct"])      # wrong module for that kernel
    with pytest.raises(SimlodError):
        Program(["./modules/progressive_octree/render.cu"], ["no_such_kernel"])


def test_null_arguments_are_rejected_without_touching_the_device(built_libs):
    from simlod_amd import runtime
    L = runtime.lib()
    assert L.simlod_launch_reset(None, None, None, None, None, None, None, None) != 0
    assert L.simlod_launch_construct(*([None] * 11)) != 0
    assert L.simlod_launch_render(*([None] * 8)) != 0


def test_numpy_mirrors_match_the_header():
    text = open(os.path.join(ROOT, "include", "simlod_abi.h")).read()
    for struct, field, off in re.findall(r"offsetof\((Simlod\w+), (\w+)\) == (\d+)", text):
        dt = {"SimlodNode": abi.node_dtype, "SimlodUniforms": abi.uniforms_dtype, "SimlodStats": abi.stats_dtype}.get(struct)
        if dt is None or field not in dt.fields:
            continue
        assert dt.fields[field][1] == int(off), (struct, field)
    assert abi.alloc_round(262144) == 262160 and abi.alloc_round(16016) == 16032


def test_runtime_refuses_to_run_without_gpu(built_libs):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from simlod_amd.runtime import DeviceOctree, SimlodError
    with pytest.raises(SimlodError):
        DeviceOctree("cuda:0")


def test_context_surface_without_a_device(built_libs):
    """simlod_context_*: contexts are host objects (nothing touches a device before the first launch); settings are validated like the
    process-wide setters, knobs are known by their environment names, a context's scratch need follows ITS node capacity."""
    from simlod_amd import runtime
    L = runtime.lib()
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    assert L.simlod_context_create(ctypes.byref(a)) == 0 and L.simlod_context_create(ctypes.byref(b)) == 0 and a.value != b.value
    assert L.simlod_context_create(None) != 0
    assert L.simlod_context_set_ingest_mode(a, 1) == 0 and L.simlod_context_set_ingest_mode(a, 2) != 0
    assert L.simlod_context_set_construct_batch_limit(a, 0) != 0 and L.simlod_context_set_construct_batch_limit(a, 1000) == 0      # (clamped to 20)
    assert L.simlod_context_set_node_capacity(a, 8) != 0 and L.simlod_context_set_node_capacity(a, (1 << 19) + 1) != 0
    assert L.simlod_context_set_node_capacity(a, 40_000) == 0 and L.simlod_context_set_node_capacity(b, 400_000) == 0
    small, large, default = (int(L.simlod_context_construct_buffer_min_bytes(c)) for c in (a, b, None))
    assert small < default < large and default == int(L.simlod_construct_buffer_min_bytes())
    for name in (b"SIMLOD_OVERLAP_TAIL", b"SIMLOD_EXPAND_WGS", b"SIMLOD_RASTER_LDS_TILES", b"SIMLOD_GROUP_BATCHES", b"SIMLOD_DEBUG_BUDGET_US"):
        assert L.simlod_context_set_knob(a, name, 1, 1) == 0 and L.simlod_context_set_knob(a, name, 0, 0) == 0
    assert L.simlod_context_set_knob(a, b"SIMLOD_NO_SUCH_KNOB", 1, 1) != 0 and L.simlod_context_set_knob(a, None, 1, 1) != 0
    os.environ["SIMLOD_EXPAND_WGS"] = "32"
    try:
        assert L.simlod_context_reload_env(a) == 0 and L.simlod_context_reload_env(None) == 0
    finally:
        del os.environ["SIMLOD_EXPAND_WGS"]
        L.simlod_context_reload_env(None)
    fake_nodes = ctypes.c_void_p(0x1000)
    assert L.simlod_context_attach(a, fake_nodes) == 0 and L.simlod_context_attach(b, fake_nodes) == 0 and L.simlod_context_attach(None, fake_nodes) == 0
    assert L.simlod_context_attach(a, None) != 0
    assert L.simlod_context_destroy(a) == 0 and L.simlod_context_destroy(b) == 0 and L.simlod_context_destroy(None) != 0
