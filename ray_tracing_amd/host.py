"""Host-side mirror of the reference's scene and render-call classes.

Same names and argument meaning as the reference's C++ headers
(src/gpu_version/shape.h, camera.h, ray_tracing_compute_shader.h,
src/SimpleMesh.h), so that code and tests read like src/main.cpp:88-151.  All
arithmetic and all rendering happen in librt_mi355x.so; this module only
marshals.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (MATERIAL_DTYPE, SHAPE_DTYPE, TRIANGLE_DTYPE, RtError, lib,
                   SEM_CPU, SEM_GLSL, SAMPLE_JITTER)


# ---------------------------------------------------------------- materials
# src/gpu_version/shape.h:5-42
class Material:
    def __init__(self, type_, color, data=(0.0, 0.0, 0.0, 0.0)):
        self.rec = np.zeros((), dtype=MATERIAL_DTYPE)
        self.rec["color"] = np.asarray(color, dtype=np.float32)
        self.rec["type"] = type_
        self.rec["data"] = np.asarray(data, dtype=np.float32)

    @property
    def type(self):
        return int(self.rec["type"])

    @property
    def color(self):
        return self.rec["color"]


class Light(Material):                      # shape.h:14-19
    def __init__(self, color):
        super().__init__(_abi.MAT_LIGHT, color)


class Lambertian(Material):                 # shape.h:21-26
    def __init__(self, color):
        super().__init__(_abi.MAT_LAMBERTIAN, color)


class Metal(Material):                      # shape.h:28-34; fuzz: src/cpu_version/scene.h:71
    def __init__(self, color, refraction=1.0, fuzz=0.0):
        super().__init__(_abi.MAT_METAL, color, (refraction, fuzz, 0.0, 0.0))


class Dielectric(Material):                 # extension: no counterpart in the reference
    def __init__(self, color, ior=1.5):
        super().__init__(_abi.MAT_DIELECTRIC, color, (ior, 0.0, 0.0, 0.0))


class DebugMaterial(Material):              # shape.h:37-42
    def __init__(self, color):
        super().__init__(_abi.MAT_DEBUG, color)


# ------------------------------------------------------------------- shapes
# src/gpu_version/shape.h:44-70
class Shape:
    def __init__(self):
        self.rec = np.zeros((), dtype=SHAPE_DTYPE)

    @property
    def transformation(self):
        """4x4, element [r][c] (stored column-major like Eigen's ColMajor)."""
        return self.rec["transformation"].reshape(4, 4).T


class Circle(Shape):                        # shape.h:56-62
    def __init__(self, translation, radius, material):
        super().__init__()
        lib.rt_make_circle(self.rec.ctypes.data, _abi.f3(translation), float(radius), material.rec.ctypes.data)


class Box(Shape):                           # shape.h:64-70
    def __init__(self, translation, sides, material):
        super().__init__()
        lib.rt_make_box(self.rec.ctypes.data, _abi.f3(translation), _abi.f3(sides), material.rec.ctypes.data)


class Cube(Shape):                          # the CPU tracer's Cube, src/cpu_version/scene.h:142-217
    def __init__(self, center, material):
        super().__init__()
        lib.rt_make_cube(self.rec.ctypes.data, _abi.f3(center), material.rec.ctypes.data)

    def rotate(self, angle):                # scene.h:206-208
        lib.rt_cube_rotate(self.rec.ctypes.data, float(angle))


class SimpleTriangle:                       # shape.h:81-109
    def __init__(self, p1, p2, p3):
        self.rec = np.zeros((), dtype=TRIANGLE_DTYPE)
        lib.rt_make_triangle(self.rec.ctypes.data, _abi.f3(p1), _abi.f3(p2), _abi.f3(p3))


def pack(records, dtype):
    """list of objects with .rec (or a ready structured array) -> contiguous array."""
    if isinstance(records, np.ndarray):
        assert records.dtype == dtype
        return np.ascontiguousarray(records)
    out = np.zeros(len(records), dtype=dtype)
    for i, r in enumerate(records):
        out[i] = r.rec
    return out


# ------------------------------------------------------------------- camera
class Camera:
    """src/gpu_version/camera.h:8-122.  `position` defaults to the GPU host's
    (1,1,6) (camera.h:17); the CPU tracer uses (1,0,6) (rayTracer.h:37)."""

    FORWARD, BACKWARD, RIGHT, LEFT = 0, 1, 2, 3       # camera.h:10-12
    dataSize = _abi.RT_CAMERA_FLOATS

    def __init__(self, width, height, focalLength=1.0, position=(1.0, 1.0, 6.0)):
        self.width, self.height, self.focalLength = int(width), int(height), float(focalLength)
        self.data = np.zeros(self.dataSize, dtype=np.float32)
        lib.rt_camera_init(_abi.fptr(self.data), self.width, self.height, self.focalLength, _abi.f3(position))

    def getCameraPosition(self):
        return self.data[0:3]

    def getCameraDirection(self):
        return self.data[4:7]

    def getUpperLeft(self):
        return self.data[8:11]

    def getUpDirection(self):
        return self.data[12:15]

    def getRightDirection(self):
        return self.data[16:19]

    def getWorldStep(self):
        return float(self.data[19])

    def rotate(self, horizontal, vertical):             # camera.h:85-92
        lib.rt_camera_rotate(_abi.fptr(self.data), self.width, self.height, self.focalLength,
                             float(horizontal), float(vertical))

    def move(self, direction):                          # camera.h:94-113
        lib.rt_camera_move(_abi.fptr(self.data), self.width, self.height, self.focalLength, int(direction))


# --------------------------------------------------------------------- mesh
class SimpleMesh:
    """src/SimpleMesh.h:22-323."""

    def __init__(self):                                 # SimpleMesh.h:42
        self._h = C.c_void_p()
        rc = lib.rt_mesh_create(C.byref(self._h))
        if rc != 0:
            raise RtError(rc, "rt_mesh_create")

    @classmethod
    def _adopt(cls, handle):
        m = cls.__new__(cls)
        m._h = handle
        return m

    def clear(self):                                    # SimpleMesh.h:47-50
        lib.rt_mesh_clear(self._h)

    def addVertex(self, position, color=(0.0, 0.0, 0.0, 0.0)):   # SimpleMesh.h:93-97
        p = np.ascontiguousarray(position, dtype=np.float32)
        c = np.ascontiguousarray(color, dtype=np.float32)
        assert p.size == 4 and c.size == 4
        return lib.rt_mesh_add_vertex(self._h, _abi.fptr(p), _abi.fptr(c))

    def addFace(self, idx0, idx1, idx2):                # SimpleMesh.h:99-104
        return lib.rt_mesh_add_face(self._h, int(idx0), int(idx1), int(idx2))

    def setColor(self, r, g, b, a):                     # SimpleMesh.h:87-91
        lib.rt_mesh_set_color(self._h, float(r), float(g), float(b), float(a))

    def writeMesh(self, filename):                      # SimpleMesh.h:217-245
        return lib.rt_mesh_write(self._h, str(filename).encode()) == 0

    @classmethod
    def joinMeshes(cls, mesh1, mesh2, pose1to2=None):   # SimpleMesh.h:251-288; pose1to2: 4x4, element [r][c]
        h = C.c_void_p()
        pose = None
        if pose1to2 is not None:
            pose = np.ascontiguousarray(np.asarray(pose1to2, dtype=np.float32).reshape(4, 4).T)   # column-major
        rc = lib.rt_mesh_join(C.byref(h), mesh1._h, mesh2._h, _abi.fptr(pose) if pose is not None else None)
        if rc != 0:
            raise RtError(rc, "rt_mesh_join")
        return cls._adopt(h)

    @classmethod
    def sphere(cls, center, scale=1.0, color=(0.0, 0.0, 1.0, 1.0)):   # SimpleMesh.h:293-317
        h = C.c_void_p()
        c = np.ascontiguousarray(color, dtype=np.float32)
        rc = lib.rt_mesh_sphere(C.byref(h), _abi.f3(center), float(scale), _abi.fptr(c))
        if rc != 0:
            raise RtError(rc, "rt_mesh_sphere")
        return cls._adopt(h)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:      # lib is None while the interpreter shuts down
            lib.rt_mesh_free(self._h)
            self._h = C.c_void_p()

    def loadMesh(self, filename):                       # SimpleMesh.h:145-215
        loaded = C.c_void_p()
        if lib.rt_mesh_load(C.byref(loaded), str(filename).encode()) != 0:
            self.clear()                                # :147-148 clear first
            return False
        lib.rt_mesh_free(self._h)
        self._h = loaded
        return True

    def normalize(self):                                # SimpleMesh.h:60-68
        lib.rt_mesh_normalize(self._h)

    def resize(self):                                   # SimpleMesh.h:79-85
        lib.rt_mesh_resize(self._h)

    def transform(self, rotation, translation):         # SimpleMesh.h:122-126
        r = np.ascontiguousarray(np.asarray(rotation, dtype=np.float32).reshape(3, 3))
        lib.rt_mesh_transform(self._h, _abi.fptr(r), _abi.f3(translation))

    def getVertices(self):
        """(positions[n,4], colors[n,4])"""
        n = lib.rt_mesh_vertex_count(self._h)
        pos = np.zeros((n, 4), dtype=np.float32)
        col = np.zeros((n, 4), dtype=np.float32)
        lib.rt_mesh_copy_vertices(self._h, pos.ctypes.data, col.ctypes.data)
        return pos, col

    def getTriangles(self):
        n = lib.rt_mesh_triangle_count(self._h)
        idx = np.zeros((n, 3), dtype=np.uint32)
        lib.rt_mesh_copy_indices(self._h, idx.ctypes.data)
        return idx

    def toSimpleTriangles(self):
        """The loop of src/main.cpp:113-122 -> structured array of SimpleTriangle."""
        n = lib.rt_mesh_triangle_count(self._h)
        out = np.zeros(n, dtype=TRIANGLE_DTYPE)
        if lib.rt_mesh_to_triangles(self._h, out.ctypes.data) != 0:
            raise IndexError("SimpleMesh: a face names a vertex the mesh does not have")
        return out

    def triangleColors(self):
        """rgb per face: the mean of its vertex colours (what a COFF file carries,
        SimpleMesh.h:170-183; the reference's shader ignores them, RayTracing.glsl:275)."""
        n = lib.rt_mesh_triangle_count(self._h)
        out = np.zeros((n, 3), dtype=np.float32)
        if lib.rt_mesh_triangle_colors(self._h, out.ctypes.data) != 0:
            raise IndexError("SimpleMesh: a face names a vertex the mesh does not have")
        return out


# ------------------------------------------------------ render-call surface
class RayTracingComputeShader:
    """src/gpu_version/ray_tracing_compute_shader.h:9-67 on top of the HIP core.

    `shapes` / `triangles` are public lists filled by the caller and uploaded
    by updateShapeBuffer(), exactly as in src/main.cpp:95-124.  compute(x,y,z)
    is the dispatch of main.cpp:145; the wall-clock `time` uniform of
    main.cpp:139-141 becomes the explicit `seed`.
    """

    def __init__(self, camera, device=0, semantics=SEM_GLSL, sample_mode=SAMPLE_JITTER, flags=0,
                 triangle_albedo=(1.0, 0.0, 0.0), samples_per_frame=100, max_depth=10, seed=1, rng=0):
        self.camera = camera
        self.shapes = []
        self.triangles = []
        self.triangle_colors = None                     # None: every triangle has triangle_albedo; else rgb per triangle
        self.semantics = semantics
        self.samples_per_frame = samples_per_frame      # RayTracing.glsl:2,6 (10 x 10)
        self.max_depth = max_depth                      # RayTracing.glsl:300
        self.seed = seed
        self.frame = 0
        self.grid_n = 0
        cfg = _abi.Config(C.sizeof(_abi.Config), device, semantics, sample_mode, flags,
                          (C.c_float * 3)(*triangle_albedo), rng)
        self._ctx = C.c_void_p()
        rc = lib.rt_create(C.byref(self._ctx), C.byref(cfg))
        if rc != 0:
            raise RtError(rc, lib.rt_last_error(None).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        if lib is not None:               # at interpreter shutdown the module's globals may already be gone
            self.close()

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, lib.rt_last_error(self._ctx).decode())

    # -- reference surface ------------------------------------------------
    def updateCameraBuffer(self):                        # :34-37
        self._check(lib.rt_set_camera(self._ctx, _abi.fptr(self.camera.data)))

    def updateShapeBuffer(self):                         # :44-57
        s = pack(self.shapes, SHAPE_DTYPE)
        t = pack(self.triangles, TRIANGLE_DTYPE)
        self._check(lib.rt_upload_shapes(self._ctx, s.ctypes.data if len(s) else None, len(s)))
        self._check(lib.rt_upload_triangles(self._ctx, t.ctypes.data if len(t) else None, len(t)))
        if self.triangle_colors is not None and len(self.triangle_colors):
            c = np.ascontiguousarray(self.triangle_colors, dtype=np.float32).reshape(-1, 3)
            self._check(lib.rt_upload_triangle_colors(self._ctx, c.ctypes.data, len(c)))
        self._check(lib.rt_commit_scene(self._ctx))      # one build of the device copy for all three uploads

    def compute(self, x, y, z=1, spp=None, first_sample=None):   # :39-42
        """One dispatch over x * y pixels: samples_per_frame new samples per pixel."""
        assert z == 1
        spp = self.samples_per_frame if spp is None else spp
        first = self.frame * self.samples_per_frame if first_sample is None else first_sample
        self._check(lib.rt_render(self._ctx, x, y, spp, self.max_depth, self.seed, first, self.grid_n))
        self.frame += 1

    def convert(self, n_samples):                        # main.cpp:147-151 + PixelConversion.glsl
        self._check(lib.rt_convert(self._ctx, n_samples))

    def readImage(self):
        w, h = self.camera.width, self.camera.height
        out = np.zeros((h, w, 4), dtype=np.float32)
        self._check(lib.rt_read_image(self._ctx, out.ctypes.data))
        return out

    # -- extras of the C ABI ---------------------------------------------
    def setStream(self, stream_ptr):
        self._check(lib.rt_set_stream(self._ctx, C.c_void_p(stream_ptr)))

    def setRowShard(self, index, count, band_rows=16):
        self._check(lib.rt_set_row_shard(self._ctx, index, count, band_rows))

    def bindAccum(self, device_ptr, width, height):
        self._check(lib.rt_bind_accum(self._ctx, C.c_void_p(device_ptr) if device_ptr else None, width, height))

    def render(self, width, height, spp, max_depth, seed, first_sample=0, grid_n=0, stats=False):
        if stats:
            st = _abi.Stats()
            self._check(lib.rt_render_stats(self._ctx, width, height, spp, max_depth, seed, first_sample, grid_n,
                                            C.byref(st)))
            return st.as_dict()
        self._check(lib.rt_render(self._ctx, width, height, spp, max_depth, seed, first_sample, grid_n))
        return None

    def renderNoAlloc(self, accum_device_ptr, width, height, spp, max_depth, seed, first_sample=0, grid_n=0):
        """rt_render for callers that cannot allocate: adds into caller-owned device memory, uses the library's static scratch."""
        self._check(lib.rt_render_noalloc(self._ctx, C.c_void_p(accum_device_ptr), width, height, spp, max_depth, seed,
                                          first_sample, grid_n))

    def resetAccum(self):
        self._check(lib.rt_reset_accum(self._ctx))
        self.frame = 0

    def discardAccum(self):
        """rt_reset_accum for a caller that renders next: the next render call overwrites instead of adding (no clear)."""
        self._check(lib.rt_discard_accum(self._ctx))
        self.frame = 0

    def renderResolved(self, width, height, spp, max_depth, seed, device_ptr, first_sample=0, grid_n=0, gamma_sqrt=False, flip_rows=True):
        """rt_render + rt_resolve_device as one call (the resolve rides on the launch that ends the frame where it can)."""
        self._check(lib.rt_render_resolved(self._ctx, width, height, spp, max_depth, seed, first_sample, grid_n,
                                           C.c_void_p(device_ptr), int(gamma_sqrt), int(flip_rows)))

    def setAccumSamples(self, n):
        self._check(lib.rt_set_accum_samples(self._ctx, n))

    def accumSamples(self):
        n = C.c_uint32()
        self._check(lib.rt_accum_samples(self._ctx, C.byref(n)))
        return n.value

    def resolve(self, width, height, gamma_sqrt=False, flip_rows=True):
        out = np.zeros((height, width, 4), dtype=np.float32)
        self._check(lib.rt_resolve(self._ctx, out.ctypes.data, int(gamma_sqrt), int(flip_rows)))
        return out

    def resolveDevice(self, device_ptr, gamma_sqrt=False, flip_rows=True):
        self._check(lib.rt_resolve_device(self._ctx, C.c_void_p(device_ptr), int(gamma_sqrt), int(flip_rows)))

    def traceDebug(self, width, height, max_depth, seed, sample, grid_n=0):
        hit = np.zeros((height, width, max_depth), dtype=np.int32)
        mat = np.zeros((height, width, max_depth), dtype=np.uint32)
        self._check(lib.rt_trace_debug(self._ctx, width, height, max_depth, seed, sample, grid_n,
                                       hit.ctypes.data, mat.ctypes.data))
        return hit, mat


def device_info(device=0):
    name = C.create_string_buffer(256)
    cu, mhz = C.c_int(), C.c_int()
    rc = lib.rt_device_info(device, name, 256, C.byref(cu), C.byref(mhz))
    if rc != 0:
        raise RtError(rc, "no HIP device")
    return {"name": name.value.decode(), "compute_units": cu.value, "clock_mhz": mhz.value}
