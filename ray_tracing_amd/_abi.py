"""ctypes binding of librt_mi355x.so (the C ABI declared in include/rt_mi355x.h).

There is no fallback: if the shared library is missing or does not export a
symbol the header declares, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librt_mi355x.so")

RT_CAMERA_FLOATS = 21

# enum rt_semantics / rt_sample_mode / rt_flags
SEM_CPU, SEM_GLSL = 0, 1
SAMPLE_JITTER, SAMPLE_GRID, SAMPLE_GRID_COMPAT = 0, 1, 2
RNG_HASH, RNG_GLSL = 0, 1
FLAG_METAL_UNNORMALIZED = 1
FLAG_BOX_SIDES = 2
FLAG_UNIT_GAUSS = 4
FLAG_LITERAL_GEOMETRY = 8
MAT_LIGHT, MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DEBUG = 0, 1, 2, 3, 100
SHAPE_CIRCLE, SHAPE_BOX, SHAPE_CUBE = 0, 1, 3

# numpy views of the reference's struct layouts (src/gpu_version/shape.h).
MATERIAL_DTYPE = np.dtype([("color", "<f4", 3), ("type", "<u4"), ("data", "<f4", 4)])
SHAPE_DTYPE = np.dtype([("transformation", "<f4", 16), ("type", "<u4"), ("data", "<f4", 11),
                        ("material", MATERIAL_DTYPE)])
TRIANGLE_DTYPE = np.dtype([("position", "<f4", 4), ("a", "<f4", 3), ("beta", "<f4"),
                           ("m1", "<f4", 4), ("m2", "<f4", 4)])
assert MATERIAL_DTYPE.itemsize == 32 and SHAPE_DTYPE.itemsize == 144 and TRIANGLE_DTYPE.itemsize == 64


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("semantics", C.c_uint32),
                ("sample_mode", C.c_uint32), ("flags", C.c_uint32), ("triangle_albedo", C.c_float * 3),
                ("rng", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "segments", "diffuse", "specular", "terminal",
                                          "shape_tests", "sphere_tests", "box_tests", "triangle_tests",
                                          "triangle_tests_done", "bound_tests_done", "wave_iterations", "work_units",
                                          "box_tests_done", "box_bound_tests_done", "sphere_tests_done", "primary_steps", "tree_entries", "tree_entering_lanes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


NOALLOC_MAX_TILES = 65536                # RT_NOALLOC_MAX_TILES

_vp, _u32, _u64, _i32, _int = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_int
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/rt_mi355x.h declares.
SIGNATURES = {
    "rt_create": (_int, [C.POINTER(_vp), C.POINTER(Config)]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_set_stream": (_int, [_vp, _vp]),
    "rt_upload_shapes": (_int, [_vp, _vp, _u32]),
    "rt_upload_triangles": (_int, [_vp, _vp, _u32]),
    "rt_upload_triangle_colors": (_int, [_vp, _vp, _u32]),
    "rt_set_camera": (_int, [_vp, _fp]),
    "rt_set_row_shard": (_int, [_vp, _u32, _u32, _u32]),
    "rt_bind_accum": (_int, [_vp, _vp, _u32, _u32]),
    "rt_commit_scene": (_int, [_vp]),
    "rt_render": (_int, [_vp, _u32, _u32, _u32, _u32, _u64, _u32, _u32]),
    "rt_render_noalloc": (_int, [_vp, _vp, _u32, _u32, _u32, _u32, _u64, _u32, _u32]),
    "rt_render_stats": (_int, [_vp, _u32, _u32, _u32, _u32, _u64, _u32, _u32, C.POINTER(Stats)]),
    "rt_render_timeline": (_int, [_vp, _u32, _u32, _u32, _u32, _u64, _u32, _u32, C.POINTER(Stats), _vp, _u32]),
    "rt_accum_samples": (_int, [_vp, C.POINTER(_u32)]),
    "rt_set_accum_samples": (_int, [_vp, _u32]),
    "rt_reset_accum": (_int, [_vp]),
    "rt_discard_accum": (_int, [_vp]),
    "rt_render_resolved": (_int, [_vp, _u32, _u32, _u32, _u32, _u64, _u32, _u32, _vp, _int, _int]),
    "rt_resolve": (_int, [_vp, _vp, _int, _int]),
    "rt_resolve_device": (_int, [_vp, _vp, _int, _int]),
    "rt_convert": (_int, [_vp, _u32]),
    "rt_read_image": (_int, [_vp, _vp]),
    "rt_trace_debug": (_int, [_vp, _u32, _u32, _u32, _u64, _u32, _u32, _vp, _vp]),
    "rt_camera_init": (None, [_fp, _u32, _u32, C.c_float, _fp]),
    "rt_camera_rotate": (None, [_fp, _u32, _u32, C.c_float, C.c_float, C.c_float]),
    "rt_camera_move": (None, [_fp, _u32, _u32, C.c_float, _int]),
    "rt_make_circle": (None, [_vp, _fp, C.c_float, _vp]),
    "rt_make_box": (None, [_vp, _fp, _fp, _vp]),
    "rt_make_cube": (None, [_vp, _fp, _vp]),
    "rt_cube_rotate": (None, [_vp, C.c_float]),
    "rt_make_triangle": (None, [_vp, _fp, _fp, _fp]),
    "rt_mesh_load": (_int, [C.POINTER(_vp), C.c_char_p]),
    "rt_mesh_free": (None, [_vp]),
    "rt_mesh_vertex_count": (_u32, [_vp]),
    "rt_mesh_triangle_count": (_u32, [_vp]),
    "rt_mesh_normalize": (None, [_vp]),
    "rt_mesh_resize": (None, [_vp]),
    "rt_mesh_transform": (None, [_vp, _fp, _fp]),
    "rt_mesh_copy_vertices": (None, [_vp, _vp, _vp]),
    "rt_mesh_copy_indices": (None, [_vp, _vp]),
    "rt_mesh_to_triangles": (_int, [_vp, _vp]),
    "rt_mesh_triangle_colors": (_int, [_vp, _vp]),
    "rt_mesh_create": (_int, [C.POINTER(_vp)]),
    "rt_mesh_clear": (None, [_vp]),
    "rt_mesh_add_vertex": (_u32, [_vp, _fp, _fp]),
    "rt_mesh_add_face": (_u32, [_vp, _u32, _u32, _u32]),
    "rt_mesh_set_color": (None, [_vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    "rt_mesh_write": (_int, [_vp, C.c_char_p]),
    "rt_mesh_join": (_int, [C.POINTER(_vp), _vp, _vp, _fp]),
    "rt_mesh_sphere": (_int, [C.POINTER(_vp), _fp, C.c_float, _fp]),
    "rt_device_info": (_int, [_int, C.c_char_p, C.c_size_t, C.POINTER(_int), C.POINTER(_int)]),
}


def _load():
    # librt_mi355x.so names its HIP runtime "libamdhip64.so"; PyTorch bundles one under that name.
    # Importing torch first makes both use the SAME runtime, so torch streams, events and device
    # tensors can be handed to the library.  Without torch the ROCm install's runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C ray_tracing_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def f3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


def fptr(arr):
    return arr.ctypes.data_as(_fp)
