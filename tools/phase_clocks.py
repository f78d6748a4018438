"""Where a wave's time goes, phase by phase (needs the library built with -DRT_PHASE_CLOCKS=1: the timeline record
of a wave then holds the shader clocks / 16 its loop spent in the sphere scan, boxes and cubes, the triangle tree,
and the rest).   usage: python tools/phase_clocks.py [config] [spp]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rta
from ray_tracing_amd import _abi
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
cfg = rta.scenes.CONFIGS[name]()
if len(sys.argv) > 2: cfg.spp = int(sys.argv[2])
rt = rta.scenes.make_renderer(cfg)
n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
for rep in range(3):
    rt.resetAccum()
    tl = np.zeros((n_tiles * 8, 4), dtype=np.int32)
    st = _abi.Stats()
    rt._check(_abi.lib.rt_render_timeline(rt._ctx, cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, 0, 0, C.byref(st), tl.ctypes.data, n_tiles * 8))
t = tl.view(np.uint32).astype(np.int64)
t[:, 2] &= 0x0fffffff
tot = t.sum(axis=0) * 16
names = ["sphere scan", "boxes / cubes", "triangle tree", "surface, draws, new direction"]
if os.environ.get("PHASE_SPLIT") == "3":        # -DRT_PHASE_CLOCKS=3
    names = ["everything but the triangle tree", "walk by the whole wave, queueing", "passes over queued nodes", "passes over queued leaves"]
if os.environ.get("PHASE_SPLIT") == "4":        # -DRT_PHASE_CLOCKS=4
    names = ["primary steps", "other steps: closest hit", "other steps: surface, draws, direction", "(number of primary steps x 16)"]
if os.environ.get("PHASE_SPLIT") == "2":        # the library was built with -DRT_PHASE_CLOCKS=2
    names = ["closest hit (all of it)", "record, normal", "accumulate, draws, scatter / restart", "new direction"]
wi = max(1, st.wave_iterations)
print(f"{name}: {st.wave_iterations} wave iterations, lane use {st.segments / (64 * wi):.3f}")
for n, v in zip(names, tot):
    print(f"  {n:32s} {100.0 * v / tot.sum():5.1f} %   {v / wi:8.0f} clocks per wave iteration")
print(f"  {'all':32s} {tot.sum() / wi:8.0f} clocks per wave iteration")
if os.environ.get("PHASE_SPLIT") == "4":
    n_p = tot[3] / 256.0
    print(f"  primary steps: {n_p:.0f} = {n_p / wi:.3f} of the steps, {tot[0] / max(1.0, n_p):.0f} clocks each; other steps: {(tot[1] + tot[2]) / max(1.0, wi - n_p):.0f} clocks each "
          f"({tot[1] / max(1.0, wi - n_p):.0f} closest hit)")
rt.close()
