"""The C-ABI library loads and exports every symbol include/rt_mi355x.h declares (no GPU needed)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_mi355x.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_render_surface():
    syms = declared_symbols()
    for must in ("rt_create", "rt_destroy", "rt_upload_shapes", "rt_upload_triangles", "rt_set_camera",
                 "rt_render", "rt_resolve", "rt_reset_accum", "rt_trace_debug", "rt_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol(rta):
    lib = C.CDLL(rta.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"not exported: {missing}"


def test_binding_covers_every_declared_symbol(rta):
    from ray_tracing_amd import _abi
    assert sorted(_abi.SIGNATURES) == declared_symbols()


def test_struct_sizes_match_reference_layouts(rta):
    # shape.h:5-12 (32 B), :44-54 (144 B), :103-108 (64 B); camera.h:115 (21 floats)
    assert rta.MATERIAL_DTYPE.itemsize == 32
    assert rta.SHAPE_DTYPE.itemsize == 144
    assert rta.SHAPE_DTYPE.fields["type"][1] == 64
    assert rta.SHAPE_DTYPE.fields["data"][1] == 68
    assert rta.SHAPE_DTYPE.fields["material"][1] == 112
    assert rta.TRIANGLE_DTYPE.itemsize == 64
    assert rta.TRIANGLE_DTYPE.fields["a"][1] == 16
    assert rta.TRIANGLE_DTYPE.fields["beta"][1] == 28
    assert rta.TRIANGLE_DTYPE.fields["m1"][1] == 32
    assert rta.TRIANGLE_DTYPE.fields["m2"][1] == 48
    assert rta.Camera.dataSize == 21


def test_create_rejects_bad_config(rta):
    from ray_tracing_amd import _abi
    ctx = C.c_void_p()
    cfg = _abi.Config(4, 0, 0, 0, 0, (C.c_float * 3)(1, 0, 0))      # wrong struct_size
    assert _abi.lib.rt_create(C.byref(ctx), C.byref(cfg)) == -1
    assert b"struct_size" in _abi.lib.rt_last_error(None)
    cfg = _abi.Config(C.sizeof(_abi.Config), 0, 7, 0, 0, (C.c_float * 3)(1, 0, 0))   # unknown semantics
    assert _abi.lib.rt_create(C.byref(ctx), C.byref(cfg)) == -1


def test_no_silent_cpu_fallback(rta):
    """Without a HIP device the renderer refuses to come up; it never renders on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rta.RtError) as e:
        rta.RayTracingComputeShader(rta.Camera(16, 16))
    assert e.value.code == -2      # RT_ERR_NO_DEVICE
