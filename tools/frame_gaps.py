"""Where the time of a short timed region goes: per frame, the host time at which it was queued and the GPU time
(events on the frame's stream) at which its clear started and its resolve ended, after a device-wide wait.
usage: python tools/frame_gaps.py [config] [frames] [streams] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_amd as rta
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n_streams = int(sys.argv[3]) if len(sys.argv) > 3 else 2
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 5
cfg = rta.scenes.CONFIGS[name]()
streams = [torch.cuda.Stream() for _ in range(n_streams)]
rts, accs, outs = [], [], []
for s in streams:
    rt = rta.scenes.make_renderer(cfg)
    rt.setStream(s.cuda_stream)
    acc = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
    rt.bindAccum(acc.data_ptr(), cfg.width, cfg.height)
    rts.append(rt); accs.append(acc); outs.append(torch.empty(cfg.height, cfg.width, 4, dtype=torch.float32, device="cuda"))
def frame(k, ev=None):
    rt = rts[k % n_streams]
    if ev: ev[0].record(streams[k % n_streams])
    rt.resetAccum()
    rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    rt.resolveDevice(outs[k % n_streams].data_ptr(), gamma_sqrt=(cfg.semantics == 0), flip_rows=True)
    if ev: ev[1].record(streams[k % n_streams])
for rep in range(3):
    for k in range(warmup): frame(k)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(frames)]
    base = torch.cuda.Event(enable_timing=True)
    host = []
    t0 = time.perf_counter()
    base.record(streams[0])
    for k in range(frames):
        frame(k, evs[k]); host.append((time.perf_counter() - t0) * 1e3)
    t_queued = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) * 1e3
    print(f"rep {rep}: {frames} frames, {n_streams} stream(s): total {total:.3f} ms ({total / frames:.4f} per frame), all queued after {t_queued:.3f} ms")
    for k in range(frames):
        same = base.elapsed_time(evs[k][0]) if k % n_streams == 0 else float("nan")
        print(f"  frame {k:2d}: queued at {host[k]:7.3f} ms host | starts {same:7.3f} ms after the base (stream 0 only) | lasts {evs[k][0].elapsed_time(evs[k][1]):.4f} ms")
