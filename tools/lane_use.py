"""Segments per sample and the share of lanes at work per wave iteration, per configuration.
usage: python tools/lane_use.py [c1 c2 ...]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rta
for name in sys.argv[1:] or ["c2", "c3", "c5"]:
    cfg = rta.scenes.CONFIGS[name]()
    if name == "c3": cfg.spp = 32
    rt = rta.scenes.make_renderer(cfg)
    st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
    print(name, "segments/sample %.3f" % (st["segments"] / st["samples"]), "lane use %.3f" % (st["segments"] / (64.0 * st["wave_iterations"])),
          "iterations per wave %.1f" % (st["wave_iterations"] / ((cfg.width + 7) // 8 * ((cfg.height + 7) // 8))))
    rt.close()
