"""What updateShapeBuffer() costs on the host: the uploads (which only record the arrays) and the one build of the
device copy — repacking, the tree over the triangles, the copy to the device — for meshes of 292, 4 968 and 20 000
triangles.   usage: python tools/upload_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rta
from ray_tracing_amd import _abi
lib = _abi.lib


def big_mesh(n):
    """n triangles on a bumpy sheet (random but fixed)."""
    rng = np.random.default_rng(5)
    side = int(np.ceil(np.sqrt(n / 2))) + 1
    x, z = np.meshgrid(np.linspace(-2, 2, side), np.linspace(-6, -2, side))
    y = 0.3 * np.sin(3 * x) * np.cos(2 * z) + 0.02 * rng.standard_normal(x.shape)
    p = np.stack([x, y, z], -1).astype(np.float32)
    tris = np.zeros(n, _abi.TRIANGLE_DTYPE)
    k = 0
    for i in range(side - 1):
        for j in range(side - 1):
            for a, b, c in ((p[i, j], p[i + 1, j], p[i, j + 1]), (p[i + 1, j], p[i + 1, j + 1], p[i, j + 1])):
                if k < n:
                    lib.rt_make_triangle(tris[k:k + 1].ctypes.data, _abi.f3(a), _abi.f3(b), _abi.f3(c))
                    k += 1
    return tris


cases = [("Bunny-LowPoly.off, 292", rta.scenes.main_cpp_mesh("Bunny-LowPoly.off")), ("bunny.off, 4 968", rta.scenes.main_cpp_mesh("bunny.off")),
         ("synthetic sheet, 20 000", big_mesh(20000))]
cfg = rta.scenes.config3(width=64, height=64, spp=1)
for name, tris in cases:
    cfg.triangles = tris
    rt = rta.scenes.make_renderer(cfg)                     # first build: pays the context's allocations too
    t_up, t_commit = [], []
    for rep in range(5):
        t0 = time.perf_counter()
        rt._check(lib.rt_upload_shapes(rt._ctx, cfg.shapes.ctypes.data, len(cfg.shapes)))
        rt._check(lib.rt_upload_triangles(rt._ctx, tris.ctypes.data, len(tris)))
        t1 = time.perf_counter()
        rt._check(lib.rt_commit_scene(rt._ctx))
        t2 = time.perf_counter()
        t_up.append(t1 - t0); t_commit.append(t2 - t1)
    print(f"{name:28s} triangles: uploads {min(t_up) * 1e3:7.3f} ms, build + copy {min(t_commit) * 1e3:8.2f} ms", flush=True)
    rt.close()
