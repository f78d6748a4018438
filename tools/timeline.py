"""Diagnostic: when did each wave of a launch run?  Prints occupancy over time and per-XCD finish times."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rta
from ray_tracing_amd import _abi
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = rta.scenes.CONFIGS[name](**({"mesh": os.environ["TL_MESH"]} if os.environ.get("TL_MESH") else {}))
if len(sys.argv) > 2: cfg.spp = int(sys.argv[2])
rt = rta.scenes.make_renderer(cfg)
n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
reps = int(os.environ.get('TL_REPS', '3'))
for rep in range(reps):
    rt.resetAccum()
    tl = np.zeros((n_tiles * 8, 4), dtype=np.int32)
    st = _abi.Stats()
    rt._check(_abi.lib.rt_render_timeline(rt._ctx, cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, 0, 0, (C.byref(st) if os.environ.get('TL_STATS') else None), tl.ctypes.data, n_tiles * 8))
t = tl.view(np.uint32)
if os.environ.get('TL_DUMP'): np.save(os.environ['TL_DUMP'], t)
start, end, hw, it = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2], t[:, 3]
live = it > 0
t0 = start[live].min(); start = (start - t0) * 0.01; end = (end - t0) * 0.01          # microseconds
total = end[live].max()
print(f"{name}: {live.sum()} waves, span {total:.1f} us, mean wave {np.mean(end[live]-start[live]):.1f} us, max wave {np.max(end[live]-start[live]):.1f} us")
edges = np.linspace(0, total, 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    print(f"  t={mid:8.1f} us  resident waves {int(((start <= mid) & (end > mid) & live).sum()):6d}")
xcc = hw >> 28
for k in range(8):
    m = live & (xcc == k)
    if m.any():
        print(f"  XCC {k}: {int(m.sum()):5d} waves, first start {start[m].min():7.1f}, last start {start[m].max():7.1f}, last end {end[m].max():7.1f} us, sum of wave times {(end[m]-start[m]).sum():9.0f} us, iterations {int(it[m].sum())}")
print("  resident waves per XCC over time:")
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    print(f"  t={mid:8.1f} us ", " ".join(f"{int(((start <= mid) & (end > mid) & live & (xcc == k)).sum()):5d}" for k in range(8)))
blk = np.arange(len(hw)) // 4
print("  XCC of the first 16 workgroups:", [int(xcc[4*b]) for b in range(16)])
busy = (end[live] - start[live]).sum()
print(f"  sum of wave times {busy:.0f} us -> mean resident {busy/total:.0f} waves of {256*4*8} slots")
print(f"  last start at {start[live].max():.1f} us; waves ending in the last 10% of the span: {int((end[live] > 0.9*total).sum())}")
rt.close()
idx = np.argsort(-end * live)[:12]
cyc = (hw & 0x0fffffff).astype(np.int64) * 16
print("  latest-ending waves: (queue position, start, end, iterations, shader clocks, MHz)")
for i in idx:
    print(f"    q={i:6d} start {start[i]:8.1f} end {end[i]:8.1f} iterations {it[i]:5d} clocks {cyc[i]:9d} -> {cyc[i]/max(end[i]-start[i],1e-9):7.0f} MHz, {cyc[i]/max(it[i],1):6.0f} clocks/iteration")
mhz = cyc[live] / np.maximum(end[live] - start[live], 1e-9)
print(f"  shader clock seen by waves: median {np.median(mhz):.0f} MHz, 5%..95% {np.percentile(mhz,5):.0f}..{np.percentile(mhz,95):.0f}")
cpi = cyc[live] / np.maximum(it[live], 1)
print(f"  clocks per iteration: median {np.median(cpi):.0f}, 5%..95% {np.percentile(cpi,5):.0f}..{np.percentile(cpi,95):.0f}")
qs = np.arange(len(it))
for a in range(0, len(it), len(it) // 10):
    seg = it[a:a + len(it) // 10]
    print(f"  queue positions {a:6d}..: mean iterations {seg.mean():7.1f} max {seg.max():5d}")
for a in range(0, len(it), len(it) // 10):
    sl = slice(a, a + len(it) // 10)
    print(f"  queue positions {a:6d}..: clocks/iteration median {np.median(cyc[sl]/np.maximum(it[sl],1)):7.0f}, wave time median {np.median(end[sl]-start[sl]):7.1f} us, start median {np.median(start[sl]):7.1f}")
