"""One full-size frame of a configuration on the GPU and on the CPU restatement, compared word for word.
usage: python tools/compare_full_frame.py <config> [spp] [mesh file for c3]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ray_tracing_amd as rta
from oracle import oracle as orc

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kw = {"mesh": sys.argv[3]} if len(sys.argv) > 3 else {}
cfg = rta.scenes.CONFIGS[name](**kw)
rt = rta.scenes.make_renderer(cfg)
import torch
acc = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
rt.bindAccum(acc.data_ptr(), cfg.width, cfg.height)
rt.render(cfg.width, cfg.height, spp, cfg.max_depth, cfg.seed)
torch.cuda.synchronize()
got = acc.cpu().numpy().view("uint32")
rt.bindAccum(None, 0, 0)
rt.close()
p = orc.params(cfg.width, cfg.height, spp, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode,
               flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, threads=min(16, os.cpu_count() or 1))
t0 = time.perf_counter()
want, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
print(f"{cfg.name}: {cfg.width}x{cfg.height}, {spp} spp, {len(cfg.triangles)} triangles: CPU {time.perf_counter() - t0:.1f} s; "
      f"differing words: {int((got != want).sum())} of {got.size}", flush=True)
