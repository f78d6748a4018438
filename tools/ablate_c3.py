"""Where the time of the mesh scene goes: the frame time of the scene with its triangles, boxes or both left out.
usage: python tools/ablate_c3.py   (times include the copy of the resolved frame to host memory, about 2.1 ms)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
import ray_tracing_amd as rta
base = rta.scenes.config3()
def run(name, shapes, tris):
    cfg = rta.scenes.config3(); cfg.shapes = shapes; cfg.triangles = tris
    rt = rta.scenes.make_renderer(cfg)
    ts = []
    for rep in range(5):
        rt.resetAccum()
        rt.accumSamples()
        t0 = time.perf_counter()
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        rt.resolve(cfg.width, cfg.height)       # waits
        ts.append(time.perf_counter() - t0)
    st = rt.render(cfg.width, cfg.height, 8, cfg.max_depth, cfg.seed, stats=True)
    wi = max(1, st["wave_iterations"])
    print(f"{name:28s} {min(ts)*1e3:7.2f} ms (incl. ~2.1 ms copy)   segments/sample {st['segments']/st['samples']:.3f}  lane use {st['segments']/(64*wi):.3f}  "
          f"per wave iteration (lane tests): box bounds {st['box_bound_tests_done']/wi:.1f}, box tests {st['box_tests_done']/wi:.1f}, "
          f"tree bounds {st['bound_tests_done']/wi:.1f}, triangle tests {st['triangle_tests_done']/wi:.1f}", flush=True)
    rt.close()
is_box = base.shapes["type"] == 1
run("full", base.shapes, base.triangles)
run("no triangles", base.shapes, base.triangles[:0])
run("no boxes", base.shapes[~is_box], base.triangles)
run("spheres only", base.shapes[~is_box], base.triangles[:0])
run("floor box + spheres", np.concatenate([base.shapes[~is_box], base.shapes[5:6]]), base.triangles[:0])
run("floor box + spheres + tris", np.concatenate([base.shapes[~is_box], base.shapes[5:6]]), base.triangles)
