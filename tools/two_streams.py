"""Frames on one stream against frames alternating between two streams (two contexts, two accumulators): how much of a
launch's ramp and tail another frame can fill.   usage: python tools/two_streams.py [config] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rta
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = rta.scenes.CONFIGS[name]()
for n_streams in (1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    rts, accs, outs = [], [], []
    for s in streams:
        rt = rta.scenes.make_renderer(cfg)
        rt.setStream(s.cuda_stream)
        acc = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
        rt.bindAccum(acc.data_ptr(), cfg.width, cfg.height)
        rts.append(rt); accs.append(acc); outs.append(torch.empty(cfg.height, cfg.width, 4, dtype=torch.float32, device="cuda"))
    def frame(k):
        rt = rts[k % n_streams]
        rt.resetAccum()
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        rt.resolveDevice(outs[k % n_streams].data_ptr(), gamma_sqrt=(cfg.semantics == 0), flip_rows=True)
    for k in range(6): frame(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(frames): frame(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {n_streams} stream(s): {dt / frames * 1e3:.4f} ms per frame", flush=True)
    for rt in rts:
        rt.bindAccum(None, 0, 0); rt.close()
