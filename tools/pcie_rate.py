"""Frame rate of the headline workload when every frame is also resolved into HOST memory (rt_resolve: the
copy over PCIe included), beside the device-resident rate bench.py reports.  usage: python tools/pcie_rate.py [config]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_amd as rta

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = rta.scenes.CONFIGS[name]()
rt = rta.scenes.make_renderer(cfg)
gamma = cfg.semantics == rta.SEM_CPU


def frame():
    rt.resetAccum()
    rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    return rt.resolve(cfg.width, cfg.height, gamma_sqrt=gamma)          # blocks until the frame is in host memory


for _ in range(3):
    frame()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    img = frame()
dt = (time.perf_counter() - t0) / n
samples = cfg.width * cfg.height * cfg.spp
print(f"{name}: {dt * 1e3:.2f} ms per frame with the resolved frame copied to host memory "
      f"({img.nbytes / 1e6:.1f} MB) -> {samples / dt / 1e6:.0f} Msamples/s")
rt.close()
