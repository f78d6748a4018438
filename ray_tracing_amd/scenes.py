"""The benchmark configurations as concrete scenes.

Each builder returns a `SceneConfig`: shapes / triangles in the reference's
struct layouts, the 21-float camera block, and the render parameters.  Scene
contents are the reference's own hard-coded lists (cited per builder).
"""
import math
import os
from dataclasses import dataclass, field

import numpy as np

from . import _abi, host
from ._abi import SEM_CPU, SEM_GLSL, SHAPE_DTYPE, TRIANGLE_DTYPE

MESH_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "meshes")


@dataclass
class SceneConfig:
    name: str
    shapes: np.ndarray
    triangles: np.ndarray
    camera: np.ndarray
    width: int
    height: int
    spp: int
    max_depth: int
    semantics: int
    seed: int = 1
    sample_mode: int = _abi.SAMPLE_JITTER
    flags: int = 0
    triangle_albedo: tuple = (1.0, 0.0, 0.0)
    rng: int = 0
    triangle_colors: object = None       # None, or float32 [n_triangles, 3]
    notes: dict = field(default_factory=dict)


def _shapes(objs):
    return host.pack(objs, SHAPE_DTYPE)


def cpu_sphere_scene(with_metal=False, extra_lights=False):
    """The sphere part of the CPU tracer's scene, src/cpu_version/rayTracer.h:48-58."""
    L, M, Li = host.Lambertian, host.Metal, host.Light
    objs = [
        host.Circle((0, 0, -4), 0.5, L((0.8, 0.8, 0))),           # :48
        host.Circle((1.5, 0, -3), 0.5, L((0.5, 0, 0))),            # :49
        host.Circle((0, -94, -40), 100, L((0.3, 0.9, 0.7))),       # :50
        host.Circle((1, 2, -2), 0.5, Li((5, 5, 5))),               # :54
    ]
    if with_metal:
        objs.append(host.Circle((4.5, 1, -3), 1, M((0.8, 0.5, 1), fuzz=0.08)))   # :58
    if extra_lights:
        # Not in the reference: three more LightSource spheres for the
        # multi-light configuration, positions and intensities fixed here.
        objs += [
            host.Circle((-2.5, 1.5, -3), 0.4, Li((4, 2, 1))),
            host.Circle((3, 3, -6), 0.6, Li((1, 3, 5))),
            host.Circle((-0.5, 0.2, -1.5), 0.2, Li((6, 6, 2))),
        ]
    return _shapes(objs)


def cpu_tracer_scene(cube_turns=3):
    """The whole scene of the CPU tracer, src/cpu_version/rayTracer.h:48-61: four spheres, the fuzzy
    metal sphere and the Lambertian cube, the cube turned `cube_turns` times by rotate(0.1) as
    render() does once per frame (rayTracer.h:82)."""
    L = host.Lambertian
    cube = host.Cube((-1.5, 0.5, -4), L((0.5, 0.5, 1)))                                  # :61
    for _ in range(cube_turns):
        cube.rotate(0.1)
    return np.concatenate([cpu_sphere_scene(with_metal=True), _shapes([cube])])


def main_cpp_shapes():
    """The nine shapes of src/main.cpp:95-106."""
    L, M, Li = host.Lambertian, host.Metal, host.Light
    return _shapes([
        host.Box((-1, 12, -2), (10, 1, 1), Li((1.5, 1.5, 1.5))),          # :95
        host.Circle((1, 3, -10), 1, L((0, 1, 0))),                        # :98
        host.Circle((-1.5, 1.5, -4), 0.5, M((1, 1, 1))),                  # :99
        host.Circle((0, 1, -4), 0.5, L((0.8, 0.8, 0))),                   # :100
        host.Circle((1.5, 1, -3), 1.1, L((1, 0, 0))),                     # :102
        host.Box((0, -50, 0), (100, 1, 1), L((0.3, 0.9, 0.7))),           # :103
        host.Box((3, 2, -4.5), (1, 2, 1), L((0, 1, 1))),                  # :104
        host.Circle((-2, 1, -1), 0.5, L((0.5, 1, 1))),                    # :105
        host.Box((-2, 1, -6), (0.5, 0.5, 1), L((1, 0, 0.5))),             # :106
    ])


def main_cpp_mesh(filename="Bunny-LowPoly.off"):
    """The mesh block of src/main.cpp:108-122: load, normalize, resize,
    transform(0.5 * Rx(-pi/2), (0.5, 0.7, -2)), one SimpleTriangle per face."""
    mesh = host.SimpleMesh()
    if not mesh.loadMesh(os.path.join(MESH_DIR, filename)):
        raise FileNotFoundError(filename)
    mesh.normalize()
    mesh.resize()
    ang = np.float32(-math.pi / 2)                       # AngleAxisf takes a float angle
    c, s = np.float32(math.cos(float(ang))), np.float32(math.sin(float(ang)))
    half = np.float32(0.5)
    rot = np.array([[half * 1, half * 0, half * 0],
                    [half * 0, half * c, half * -s],
                    [half * 0, half * s, half * c]], dtype=np.float32)
    mesh.transform(rot, (0.5, 0.7, -2))
    return mesh.toSimpleTriangles()


def _camera(width, height, position):
    return host.Camera(width, height, 1.0, position).data.copy()


def config1(width=512, height=512, spp=16, max_depth=4):
    """cpu_version sphere scene (diffuse + emissive), 512x512, 16 spp, depth 4."""
    return SceneConfig("c1_cpu_spheres", cpu_sphere_scene(), np.zeros(0, TRIANGLE_DTYPE),
                       _camera(width, height, (1, 0, 6)), width, height, spp, max_depth, SEM_CPU)


def config2(width=1920, height=1080, spp=100, max_depth=8):
    """The same sphere scene at 1920x1080, 100 spp, depth 8 (the headline workload)."""
    c = config1(width, height, spp, max_depth)
    c.name = "c2_cpu_spheres_1080p"
    return c


def config3(width=1920, height=1080, spp=256, max_depth=10, mesh="Bunny-LowPoly.off"):
    """src/main.cpp's nine shapes + the SimpleMesh bunny, shader semantics."""
    return SceneConfig("c3_mesh_" + os.path.splitext(mesh)[0].lower(), main_cpp_shapes(), main_cpp_mesh(mesh),
                       _camera(width, height, (1, 1, 6)), width, height, spp, max_depth, SEM_GLSL)


def config4(width=3840, height=2160, spp=1024, max_depth=11):
    """Multi-light 'virtual material' scene: the CPU tracer's material classes
    (src/cpu_version/scene.h:39-109) incl. the fuzzy metal sphere of rayTracer.h:58."""
    return SceneConfig("c4_multi_light_4k", cpu_sphere_scene(with_metal=True, extra_lights=True),
                       np.zeros(0, TRIANGLE_DTYPE), _camera(width, height, (1, 0, 6)),
                       width, height, spp, max_depth, SEM_CPU)


def _config5_objects():
    rng = np.random.RandomState(7)
    L, M, Li, G = host.Lambertian, host.Metal, host.Light, host.Dielectric
    objs = [host.Circle((0, -1000.5, -6), 1000, L((0.5, 0.5, 0.5)))]
    k = 0
    for iz in range(8):
        for ix in range(8):
            jx, jz = rng.uniform(-0.25, 0.25, 2)
            pos = (float(-3.5 + ix + jx + 1), 0.0, float(-2.0 - iz + jz))
            col = tuple(float(v) for v in rng.uniform(0.3, 1.0, 3))
            kind = k % 6
            if kind in (0, 3):
                m = L(col)
            elif kind == 1:
                m = M(col, fuzz=0.0)
            elif kind == 2:
                m = M(col, fuzz=0.3)
            elif kind == 4:
                m = G((0.95, 0.95, 0.95), ior=1.5)
            else:
                m = Li(tuple(3.0 * c for c in col))
            objs.append(host.Circle(pos, 0.4, m))
            k += 1
    return objs


def config5(width=2048, height=2048, spp=64, max_depth=16):
    """Divergence stress (not in the reference): 64 spheres on a jittered 8x8
    grid, materials round-robin diffuse / mirror / fuzzy metal / glass / light."""
    return SceneConfig("c5_divergence", _shapes(_config5_objects()), np.zeros(0, TRIANGLE_DTYPE),
                       _camera(width, height, (1, 1.5, 6)), width, height, spp, max_depth, SEM_CPU)


def config5_enclosed(width=2048, height=2048, spp=64, max_depth=16):
    """Config 5's spheres inside a room, so that a path ends only on one of the eleven light spheres or at its
    sixteenth segment: long paths, lanes of a wave at very different depths.  The walls are the shader's Boxes
    (cubes of side 200 seen from outside, RayTracing.glsl:138-156: a Box is hit on the faces of the octant its
    origin lies in, so a room is five cubes around it, not one cube entered from inside): left and back diffuse,
    right a mirror, front (behind the camera) and ceiling diffuse; the floor is config 5's ground sphere."""
    L, M = host.Lambertian, host.Metal
    s = 200.0
    walls = [
        host.Box((-5.0 - s / 2, 0, 0), (s, s, s), L((0.75, 0.25, 0.25))),      # left,    x = -5
        host.Box((7.0 + s / 2, 0, 0), (s, s, s), M((0.9, 0.9, 0.9))),          # right,   x =  7: mirror
        host.Box((0, 0, -12.0 - s / 2), (s, s, s), L((0.25, 0.25, 0.75))),     # back,    z = -12
        host.Box((0, 0, 8.0 + s / 2), (s, s, s), L((0.7, 0.7, 0.7))),          # front,   z =  8
        host.Box((0, 5.0 + s / 2, 0), (s, s, s), L((0.7, 0.7, 0.7))),          # ceiling, y =  5
    ]
    return SceneConfig("c5_enclosed", _shapes(_config5_objects() + walls), np.zeros(0, TRIANGLE_DTYPE),
                       _camera(width, height, (1, 1.5, 6)), width, height, spp, max_depth, SEM_CPU)


def config_cpu_tracer(width=512, height=512, spp=25, max_depth=11, cube_turns=3):
    """rayTracer.h as shipped: its full scene, 5 x 5 samples on the grid of renderPixel<5> as written
    (x offset accumulating), 10 + 1 segments (rayTracer.h:78,113)."""
    return SceneConfig("cpu_tracer_as_shipped", cpu_tracer_scene(cube_turns), np.zeros(0, TRIANGLE_DTYPE),
                       _camera(width, height, (1, 0, 6)), width, height, spp, max_depth, SEM_CPU,
                       sample_mode=_abi.SAMPLE_GRID_COMPAT)


def colored_mesh():
    """A 20-face icosahedron (SimpleMesh::sphere, SimpleMesh.h:293-317) rebuilt vertex by vertex with a
    colour per vertex, as a COFF file would carry (SimpleMesh.h:170-183) -> (SimpleTriangles, rgb per face)."""
    src = host.SimpleMesh.sphere((0.5, 1.2, -2.5), 1.3)
    pos, _ = src.getVertices()
    mesh = host.SimpleMesh()
    for i, p in enumerate(pos):
        mesh.addVertex(p, ((i % 3) / 2.0, ((i + 1) % 4) / 3.0, ((i * 5) % 7) / 6.0, 1.0))
    for a, b, c in src.getTriangles():
        mesh.addFace(a, b, c)
    return mesh.toSimpleTriangles(), mesh.triangleColors()


def config_colored_mesh(width=320, height=200, spp=16, max_depth=6, semantics=SEM_GLSL):
    """src/main.cpp's shapes with a mesh whose faces have their own albedo instead of the shader's
    hard-coded red (RayTracing.glsl:275)."""
    tris, colors = colored_mesh()
    return SceneConfig("colored_mesh", main_cpp_shapes(), tris, _camera(width, height, (1, 1, 6)),
                       width, height, spp, max_depth, semantics, triangle_colors=colors)


def config_box_sides(width=320, height=200, spp=16, max_depth=6, semantics=SEM_GLSL):
    """Boxes with three different sides (Box::Box stores them, shape.h:67; the shader reads one,
    RayTracing.glsl:139): rendered with RT_FLAG_BOX_SIDES."""
    L, M, Li = host.Lambertian, host.Metal, host.Light
    shapes = _shapes([
        host.Box((-1, 9, -2), (10, 1, 6), Li((1.5, 1.5, 1.5))),
        host.Box((0, -50, 0), (100, 100, 100), L((0.3, 0.9, 0.7))),
        host.Box((3, 1, -4.5), (1, 2, 0.5), L((0, 1, 1))),
        host.Box((-2, 0.25, -3), (2.5, 0.5, 1), L((1, 0, 0.5))),
        host.Box((0.5, 1.5, -7), (6, 3, 0.2), M((0.9, 0.9, 0.9))),
        host.Circle((0.5, 0.6, -2.5), 0.6, L((0.8, 0.8, 0))),
    ])
    return SceneConfig("box_sides", shapes, np.zeros(0, TRIANGLE_DTYPE), _camera(width, height, (1, 1, 6)),
                       width, height, spp, max_depth, semantics, flags=_abi.FLAG_BOX_SIDES)


def config_sphere_normals(which="far", radius=None, width=64, height=64):
    """One sphere with the debug material (shape.h:37-42: the normal is the colour), CPU rules, one sample per pixel
    on the pixel's corner, seen through a long lens so that it fills the frame.  "far": radius 0.5 at 16 100 from
    the coordinate origin, where the hit point is off the sphere by up to 1e-3 of the radius (the rounding of its
    coordinates) and the radius is no first guess of |point - centre|; "near": radius 0.5 at distance 4 (`radius`
    overrides it: a negative one must give the same frame, Sphere::hit squares it, scene.h:118)."""
    far = which == "far"
    r = radius if radius is not None else 0.5
    dist = 16100.0 if far else 4.0
    focal = 16000.0 if far else 3.0
    cam = host.Camera(width, height, focal, (0.0, 0.0, 0.0)).data.copy()
    shapes = _shapes([host.Circle((0.03, -0.02, -dist), r, host.DebugMaterial((1, 1, 1)))])
    return SceneConfig("sphere_normals_" + which, shapes, np.zeros(0, TRIANGLE_DTYPE), cam, width, height, 1, 2, SEM_CPU,
                       sample_mode=_abi.SAMPLE_GRID)


CONFIGS = {"sphere_normals": config_sphere_normals, "colored_mesh": config_colored_mesh, "box_sides": config_box_sides, "cpu_tracer": config_cpu_tracer, "c1": config1, "c2": config2, "c3": config3, "c4": config4, "c5": config5, "c5e": config5_enclosed}


def make_renderer(cfg, device=0):
    """A RayTracingComputeShader loaded with `cfg`'s scene and camera."""
    cam = host.Camera(cfg.width, cfg.height)
    cam.data[:] = cfg.camera
    rt = host.RayTracingComputeShader(cam, device=device, semantics=cfg.semantics, sample_mode=cfg.sample_mode,
                                      flags=cfg.flags, triangle_albedo=cfg.triangle_albedo,
                                      samples_per_frame=cfg.spp, max_depth=cfg.max_depth, seed=cfg.seed, rng=cfg.rng)
    rt.shapes = cfg.shapes
    rt.triangles = cfg.triangles
    rt.triangle_colors = cfg.triangle_colors
    rt.updateShapeBuffer()
    rt.updateCameraBuffer()
    return rt
