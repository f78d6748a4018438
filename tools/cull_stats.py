"""Triangle and bound tests per ray segment that the kernels execute on the two meshes, beside the exhaustive scan's.
usage: python tools/cull_stats.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rta
for mesh in ("Bunny-LowPoly.off","bunny.off"):
    cfg = rta.scenes.config3(width=480, height=270, spp=16, max_depth=10, mesh=mesh)
    rt = rta.scenes.make_renderer(cfg)
    st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
    seg = st["segments"]
    print(mesh, "segments", seg, "tri tests/segment brute", st["triangle_tests"]/seg, "done", st["triangle_tests_done"]/seg, "node tests/segment", st["bound_tests_done"]/seg)
    rt.close()
