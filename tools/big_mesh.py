"""Frame time of the mesh scene with a rippled sheet of n x n x 2 triangles in place of the bunny, at 1080p, 8 samples
per pixel: the tree with as many levels as the mesh needs against the tree cut off at four (RT_TREE_LEVELS=4), and the
cost of the upload.   usage: python tools/big_mesh.py [n ...]      (default 100 316: 20 000 and 199 712 triangles; BIG_SPP=n: n samples per pixel)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ray_tracing_amd as rta
from ray_tracing_amd import _abi


def sheet(n):
    u = np.linspace(-3.0, 4.0, n + 1, dtype=np.float32)
    v = np.linspace(0.05, 4.0, n + 1, dtype=np.float32)
    x, y = np.meshgrid(u, v)
    z = (-7.0 + 0.3 * np.sin(3.0 * x) * np.cos(2.0 * y)).astype(np.float32)
    pts = np.stack([x, y, z], axis=-1)
    out = np.zeros(2 * n * n, dtype=_abi.TRIANGLE_DTYPE)
    k = 0
    for j in range(n):
        for i in range(n):
            a, b, c, d = pts[j, i], pts[j, i + 1], pts[j + 1, i], pts[j + 1, i + 1]
            for tri in ((a, b, c), (b, d, c)):
                _abi.lib.rt_make_triangle(out[k:k + 1].ctypes.data, _abi.fptr(np.ascontiguousarray(tri[0])), _abi.fptr(np.ascontiguousarray(tri[1])),
                                          _abi.fptr(np.ascontiguousarray(tri[2])))
                k += 1
    return out


for n in [int(a) for a in sys.argv[1:]] or [100, 316]:
    cfg = rta.scenes.config3(spp=int(os.environ.get("BIG_SPP", "8")))
    cfg.triangles = sheet(n)
    for levels in ("8", "4"):
        os.environ["RT_TREE_LEVELS"] = levels
        t0 = time.perf_counter()
        rt = rta.scenes.make_renderer(cfg)
        rt._check(_abi.lib.rt_commit_scene(rt._ctx))
        upload = time.perf_counter() - t0
        ms = []
        for frame in range(6):
            rt.discardAccum()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        rt.discardAccum()
        st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
        rt.close()
        print(f"{len(cfg.triangles):7d} triangles, at most {levels} tree levels: upload {upload * 1e3:7.1f} ms, frame {min(ms[2:]):7.2f} ms (1080p, {cfg.spp} spp; the first two frames {ms[0]:.2f} and {ms[1]:.2f}), "
              f"bound tests per segment {st['bound_tests_done'] / st['segments']:.1f}, triangle tests per segment {st['triangle_tests_done'] / st['segments']:.2f}", flush=True)
