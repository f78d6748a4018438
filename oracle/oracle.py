"""ctypes binding of the CPU oracle (oracle/_build/librt_oracle.so).

Test infrastructure: imported only by tests/, the smoke check and the
cpu_baseline leg of bench.py — never by the renderer.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "librt_oracle.so")

SEM_CPU, SEM_GLSL = 0, 1
SAMPLE_JITTER, SAMPLE_GRID, SAMPLE_GRID_COMPAT = 0, 1, 2
COLOR_AS_WRITTEN, COLOR_FORWARD = 0, 1
UNIT_DIRECT, UNIT_GAUSS = 0, 1
RNG_HASH, RNG_GLSL = 0, 1
FLAG_METAL_UNNORMALIZED, FLAG_BOX_SIDES, FLAG_UNIT_GAUSS = 1, 2, 4
FLAG_LITERAL_GEOMETRY = 8          # the renderer's flag of that value: the oracle ignores it and takes `arith` instead
ARITH_RECIPE, ARITH_LITERAL, ARITH_LITERAL_GEOMETRY = 0, 1, 2


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("first_sample", C.c_uint32),
                ("max_depth", C.c_uint32), ("grid_n", C.c_uint32), ("seed", C.c_uint64),
                ("semantics", C.c_uint32), ("sample_mode", C.c_uint32), ("flags", C.c_uint32),
                ("triangle_albedo", C.c_float * 3), ("color_order", C.c_uint32), ("unit_sampler", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("band_rows", C.c_uint32),
                ("threads", C.c_int32), ("rng", C.c_uint32), ("triangle_colors", C.c_void_p), ("arith", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("samples", "segments", "diffuse", "specular", "terminal",
                                          "shape_tests", "sphere_tests", "box_tests", "triangle_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, fp = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float)
    lib.orc_render.restype = C.c_int
    lib.orc_render.argtypes = [vp, u32, vp, u32, fp, C.POINTER(Params), vp, C.POINTER(Stats)]
    lib.orc_trace.restype = C.c_int
    lib.orc_trace.argtypes = [vp, u32, vp, u32, fp, C.POINTER(Params), u32, vp, vp]
    lib.orc_resolve.restype = None
    lib.orc_resolve.argtypes = [vp, u32, u32, u32, u32, C.c_int, C.c_int, vp]
    lib.orc_convert.restype = None
    lib.orc_convert.argtypes = [vp, vp, u32, u32, u32, u32, u32]
    lib.orc_set_arith.restype = None; lib.orc_set_arith.argtypes = [C.c_int]
    lib.orc_hash32.restype = u32; lib.orc_hash32.argtypes = [u32]
    lib.orc_log_draw.restype = C.c_float; lib.orc_log_draw.argtypes = [C.c_float]
    lib.orc_path_key.restype = u32; lib.orc_path_key.argtypes = [u64, u32, u32]
    lib.orc_draw.restype = C.c_float; lib.orc_draw.argtypes = [u32, u32]
    lib.orc_glsl_random.restype = C.c_float; lib.orc_glsl_random.argtypes = [fp]
    lib.orc_glsl_seed.restype = None; lib.orc_glsl_seed.argtypes = [u64, u32, u32, u32, u32, u32, fp]
    lib.orc_sincos2pi.restype = None; lib.orc_sincos2pi.argtypes = [C.c_float, fp, fp]
    lib.orc_unit_vector.restype = None; lib.orc_unit_vector.argtypes = [C.c_float, C.c_float, fp]
    lib.orc_hemisphere_glsl.restype = None; lib.orc_hemisphere_glsl.argtypes = [fp, C.c_float, C.c_float, fp]
    lib.orc_glass_direction.restype = None; lib.orc_glass_direction.argtypes = [fp, fp, C.c_float, C.c_float, fp]
    lib.orc_hit_sphere.restype = C.c_float; lib.orc_hit_sphere.argtypes = [fp, C.c_float, fp, fp]
    lib.orc_hit_box.restype = C.c_float; lib.orc_hit_box.argtypes = [fp, C.c_float, fp, fp]
    lib.orc_hit_triangle.restype = C.c_float; lib.orc_hit_triangle.argtypes = [vp, fp, fp]
    lib.orc_box_normal.restype = None; lib.orc_box_normal.argtypes = [fp, C.c_float, fp, fp]
    lib.orc_hit_box3.restype = C.c_float; lib.orc_hit_box3.argtypes = [fp, fp, fp, fp]
    lib.orc_box_normal3.restype = None; lib.orc_box_normal3.argtypes = [fp, fp, fp, fp]
    lib.orc_hit_cube.restype = C.c_int; lib.orc_hit_cube.argtypes = [fp, fp, fp, fp, fp]
    lib.orc_cube_normal.restype = None; lib.orc_cube_normal.argtypes = [fp, fp, fp]
    lib.orc_camera_ray.restype = None; lib.orc_camera_ray.argtypes = [fp, C.POINTER(Params), u32, u32, u32, fp, fp]
    lib.orc_camera_init.restype = None; lib.orc_camera_init.argtypes = [fp, u32, u32, C.c_float, fp]
    lib.orc_make_triangle.restype = None; lib.orc_make_triangle.argtypes = [vp, fp, fp, fp]
    lib.orc_mesh_triangles.restype = C.c_int
    lib.orc_mesh_triangles.argtypes = [C.c_char_p, C.c_int, vp, u32, C.POINTER(u32)]
    return lib


lib = _load()


def _f(*v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def params(width, height, spp, max_depth, seed=1, semantics=SEM_CPU, sample_mode=SAMPLE_JITTER, flags=0,
           first_sample=0, grid_n=0, triangle_albedo=(1.0, 0.0, 0.0), color_order=COLOR_FORWARD,
           unit_sampler=UNIT_DIRECT, shard=(0, 1, 16), threads=0, rng=RNG_HASH, triangle_colors=None, arith=ARITH_RECIPE):
    """triangle_colors: None, or float32 [n_triangles, 3] (kept alive by the returned object)."""
    p = Params(width, height, spp, first_sample, max_depth, grid_n, seed, semantics, sample_mode, flags,
               (C.c_float * 3)(*triangle_albedo), color_order, unit_sampler, shard[0], shard[1], shard[2], threads, rng, None, arith)
    if triangle_colors is not None and len(triangle_colors):
        p._colors = np.ascontiguousarray(triangle_colors, dtype=np.float32).reshape(-1, 3)
        p.triangle_colors = p._colors.ctypes.data
    return p


def render(shapes, triangles, camera, p, accum=None):
    """-> (accum uint32[h,w,4], stats dict).  `accum` is added into when given."""
    if accum is None:
        accum = np.zeros((p.height, p.width, 4), dtype=np.uint32)
    cam = np.ascontiguousarray(camera, dtype=np.float32)
    st = Stats()
    rc = lib.orc_render(shapes.ctypes.data if len(shapes) else None, len(shapes),
                        triangles.ctypes.data if len(triangles) else None, len(triangles),
                        _fp(cam), C.byref(p), accum.ctypes.data, C.byref(st))
    if rc != 0:
        raise ValueError("orc_render: bad parameters")
    return accum, st.as_dict()


def trace(shapes, triangles, camera, p, sample):
    hit = np.zeros((p.height, p.width, p.max_depth), dtype=np.int32)
    mat = np.zeros((p.height, p.width, p.max_depth), dtype=np.uint32)
    cam = np.ascontiguousarray(camera, dtype=np.float32)
    rc = lib.orc_trace(shapes.ctypes.data if len(shapes) else None, len(shapes),
                       triangles.ctypes.data if len(triangles) else None, len(triangles),
                       _fp(cam), C.byref(p), sample, hit.ctypes.data, mat.ctypes.data)
    if rc != 0:
        raise ValueError("orc_trace: bad parameters")
    return hit, mat


def resolve(accum, samples, semantics, gamma_sqrt=False, flip_rows=True):
    h, w, _ = accum.shape
    out = np.zeros((h, w, 4), dtype=np.float32)
    lib.orc_resolve(accum.ctypes.data, w, h, samples, semantics, int(gamma_sqrt), int(flip_rows), out.ctypes.data)
    return out


def convert(accum, image, samples_in_accum, semantics, n_samples):
    h, w, _ = accum.shape
    lib.orc_convert(accum.ctypes.data, image.ctypes.data, w, h, samples_in_accum, semantics, n_samples)


class arithmetic:
    """with arithmetic(ARITH_LITERAL): ... — the leaf routines below evaluate in that arithmetic on this thread."""
    def __init__(self, arith):
        self.arith = arith

    def __enter__(self):
        lib.orc_set_arith(self.arith)

    def __exit__(self, *exc):
        lib.orc_set_arith(ARITH_RECIPE)


def sincos2pi(u):
    c, s = C.c_float(), C.c_float()
    lib.orc_sincos2pi(u, C.byref(c), C.byref(s))
    return c.value, s.value


def unit_vector(u1, u2):
    o = _f(0, 0, 0)
    lib.orc_unit_vector(u1, u2, o)
    return np.array(o[:], dtype=np.float32)


def hemisphere_glsl(n, u1, u2):
    o = _f(0, 0, 0)
    lib.orc_hemisphere_glsl(_f(*n), u1, u2, o)
    return np.array(o[:], dtype=np.float32)


def glass_direction(n, d, ior, u1):
    o = _f(0, 0, 0)
    lib.orc_glass_direction(_f(*n), _f(*d), ior, u1, o)
    return np.array(o[:], dtype=np.float32)


def hit_sphere(center, radius, o, d):
    return lib.orc_hit_sphere(_f(*center), radius, _f(*o), _f(*d))


def hit_box(center, scale, o, d):
    return lib.orc_hit_box(_f(*center), scale, _f(*o), _f(*d))


def hit_box3(center, sides, o, d):
    return lib.orc_hit_box3(_f(*center), _f(*sides), _f(*o), _f(*d))


def box_normal3(center, sides, p):
    o = _f(0, 0, 0)
    lib.orc_box_normal3(_f(*center), _f(*sides), _f(*p), o)
    return np.array(o[:], dtype=np.float32)


def hit_triangle(tri_rec, o, d):
    return lib.orc_hit_triangle(tri_rec.ctypes.data, _f(*o), _f(*d))


def hit_cube(m16, o, d):
    """-> (t, point in the cube's frame, normal) or None"""
    t = C.c_float(); ip = _f(0, 0, 0); n = _f(0, 0, 0)
    m = _f(*[float(x) for x in m16])
    if not lib.orc_hit_cube(m, _f(*o), _f(*d), C.byref(t), ip):
        return None
    lib.orc_cube_normal(m, ip, n)
    return t.value, np.array(ip[:], dtype=np.float32), np.array(n[:], dtype=np.float32)


def box_normal(center, scale, p):
    o = _f(0, 0, 0)
    lib.orc_box_normal(_f(*center), scale, _f(*p), o)
    return np.array(o[:], dtype=np.float32)


def camera_ray(camera, p, gx, gy, sample):
    o, d = _f(0, 0, 0), _f(0, 0, 0)
    cam = np.ascontiguousarray(camera, dtype=np.float32)
    lib.orc_camera_ray(_fp(cam), C.byref(p), gx, gy, sample, o, d)
    return np.array(o[:], dtype=np.float32), np.array(d[:], dtype=np.float32)


def camera_init(width, height, focal=1.0, position=(1, 1, 6)):
    data = np.zeros(21, dtype=np.float32)
    lib.orc_camera_init(_fp(data), width, height, focal, _f(*position))
    return data


TRIANGLE_DTYPE = np.dtype([("position", "<f4", 4), ("a", "<f4", 3), ("beta", "<f4"),
                           ("m1", "<f4", 4), ("m2", "<f4", 4)])


def make_triangle(p1, p2, p3):
    t = np.zeros((), dtype=TRIANGLE_DTYPE)
    lib.orc_make_triangle(t.ctypes.data, _f(*p1), _f(*p2), _f(*p3))
    return t


def mesh_triangles(path, main_cpp_pipeline=True, max_tris=8192):
    out = np.zeros(max_tris, dtype=TRIANGLE_DTYPE)
    nv = C.c_uint32()
    n = lib.orc_mesh_triangles(str(path).encode(), int(main_cpp_pipeline), out.ctypes.data, max_tris, C.byref(nv))
    if n < 0:
        raise IOError(f"orc_mesh_triangles({path}) -> {n}")
    return out[:n].copy(), nv.value


def log_draw(u):
    """ln(u) of a draw u = k * 2^-24 by the oracle's fixed recipe."""
    return float(lib.orc_log_draw(u))
