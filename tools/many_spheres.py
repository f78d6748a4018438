"""Frame time of a scene of n spheres (no boxes, no mesh) at 1080p, 32 samples per pixel, depth 6, with the camera rays'
candidate lists and without (RT_NO_TILE_LISTS=1).   usage: python tools/many_spheres.py [n ...]   (default 100 300 900)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import ray_tracing_amd as rta
from ray_tracing_amd import _abi, host

for n in [int(a) for a in sys.argv[1:]] or [100, 300, 900]:
    rng = np.random.default_rng(n)
    objs = [host.Circle((1, 60, -4), 20.0, host.Light((3, 3, 3))), host.Circle((0, -1000, -20), 995.0, host.Lambertian((0.5, 0.5, 0.5)))]
    for _ in range(n - 2):
        objs.append(host.Circle(tuple(float(x) for x in rng.uniform(-12, 12, 3) * (1, 0.4, 1) + (0, 0, -20)), float(rng.uniform(0.2, 1.0)),
                                host.Lambertian(tuple(float(x) for x in rng.uniform(0.2, 1, 3))) if rng.random() < 0.7 else host.Metal((0.9, 0.9, 0.9))))
    w, h, spp = 1920, 1080, int(os.environ.get("SPP", "32"))
    for sem in (0, 1):
        cfg = rta.scenes.SceneConfig("spheres", host.pack(objs, _abi.SHAPE_DTYPE), np.zeros(0, dtype=_abi.TRIANGLE_DTYPE),
                                     host.Camera(w, h, 1.0, (1, 1, 6)).data.copy(), w, h, spp, 6, sem, seed=4)
        out = {}
        for lists in (True, False):
            if lists:
                os.environ.pop("RT_NO_TILE_LISTS", None)
            else:
                os.environ["RT_NO_TILE_LISTS"] = "1"
            rt = rta.scenes.make_renderer(cfg)
            ms = []
            for frame in range(6):
                rt.discardAccum()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                rt.render(w, h, spp, cfg.max_depth, cfg.seed)
                torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
            rt.discardAccum()
            st = rt.render(w, h, spp, cfg.max_depth, cfg.seed, stats=True)
            rt.close()
            out[lists] = (min(ms[2:]), st["sphere_tests_done"] / st["segments"], st["primary_steps"] / max(1, st["wave_iterations"]))
        print(f"{n:5d} spheres, {'CPU tracer' if sem == 0 else 'shader'} rules, 1080p, {spp} spp: {out[True][0]:8.2f} ms with lists ({out[True][1]:.0f} sphere tests per segment, "
              f"{100 * out[True][2]:.0f} % primary steps), {out[False][0]:8.2f} ms without ({out[False][1]:.0f})", flush=True)
