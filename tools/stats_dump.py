"""Work counters of one frame of a configuration: python tools/stats_dump.py [config] [mesh]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rta
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
kw = {"mesh": sys.argv[2]} if len(sys.argv) > 2 else {}
cfg = rta.scenes.CONFIGS[name](**kw)
rt = rta.scenes.make_renderer(cfg)
rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
rt.resetAccum()
st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
rt.close()
wi, ps = st["wave_iterations"], st["primary_steps"]
print(name, kw, {k: v for k, v in st.items()})
print(f"  steps {wi}, primary {ps} ({ps / wi:.3f}); per other step: bound tests {st['bound_tests_done'] / max(1, wi - ps):.1f}, "
      f"triangle tests {st['triangle_tests_done'] / max(1, wi - ps):.1f}, box bounds {st['box_bound_tests_done'] / max(1, wi - ps):.1f}, "
      f"box tests {st['box_tests_done'] / max(1, wi - ps):.1f}, sphere tests {st['sphere_tests_done'] / max(1, wi):.1f} per step; "
      f"the tree was entered in {st['tree_entries'] / max(1, wi - ps):.3f} of the other steps, by {st['tree_entering_lanes'] / max(1, st['tree_entries']):.1f} lanes on average")
