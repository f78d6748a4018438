#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path tracer on the 1080p sphere scene.

    python bench.py --gpus N --steps K --warmup W

A step is one full frame of the workload: the accumulator starts from zero (rt_discard_accum: the launch overwrites it,
nothing is cleared), `spp` samples per pixel are traced, the ranks' accumulators are exchanged (RCCL) when N > 1, and the
frame is resolved on rank 0 (on one GPU by the launch itself, rt_render_resolved).  Consecutive frames use two accumulators in turn: on one
GPU each has its own stream, so that one launch's ramp and tail are filled by the next frame's; with N > 1 one
frame's exchange runs beside the next frame's trace.  The launch duration quoted under "roofline" is measured
separately with one frame in flight.  The scene and camera are resident in HBM
before the timed region.  With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank
traces its own `spp` samples of every pixel (disjoint sample ranges -> weak scaling: the frame gains
N x spp samples per pixel), or, with --shard rows, its own interleaved bands of rows.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic flop counts (FMA = 2, transcendental = 1), from the reference's own expressions:
# hitSphere RayTracing.glsl:118-132 ~20, hitBox :138-156 ~58, hitTriangle :179-194 ~32,
# camera ray :293-298 ~25 + two draws of random() :44-51 (~11 each), a diffuse bounce
# (advance, normal, randomOnHalfUnitShere :64-80, throughput) ~90, a mirror bounce ~30,
# a terminal event (light / miss) ~3.
F_SPHERE, F_BOX, F_TRI = 20, 58, 32
F_BOUND = 20                          # one ray / bounding-sphere test of the triangle tree (3 sub, 2 dots, R*R, 2 fma, limit mul, add)
F_BOX_BOUND = 21                      # one slab test of a box's bounds (3 sub, 6 mul, 6 add/sub, 2 max, 2 min, fma, add)
F_CAMERA, F_DIFFUSE, F_SPECULAR, F_TERMINAL = 25 + 2 * 11, 90, 30, 3
PEAK_FP32_VECTOR_TFLOPS = 157.3       # 256 CU x 4 SIMD x 64 flop/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0
PEAK_CLOCK_HZ, N_SIMD = 2.4e9, 1024


def algorithmic_flops(st, executed=True, spheres_executed=False):
    """Flops of one launch.  executed=True prices the triangle and box tests the kernel really ran (it culls them by
    their bounds, and pays for the bound tests); executed=False prices the reference's exhaustive scan of every
    triangle and box.  Spheres are priced as the reference's algorithm tests them — every sphere for every segment
    (Scene::hit, scene.h:232-255; getClosestShapeIndex, RayTracing.glsl:206-231) — unless spheres_executed, which counts
    the tests run: camera rays test only the spheres their tile can see."""
    tri = st["triangle_tests_done"] * F_TRI + st["bound_tests_done"] * F_BOUND if executed else st["triangle_tests"] * F_TRI
    box = st["box_tests_done"] * F_BOX + st["box_bound_tests_done"] * F_BOX_BOUND if executed else st["box_tests"] * F_BOX
    sph = st["sphere_tests_done"] if spheres_executed else st["sphere_tests"]
    return (st["samples"] * F_CAMERA + sph * F_SPHERE + box + tri +
            st["diffuse"] * F_DIFFUSE + st["specular"] * F_SPECULAR + st["terminal"] * F_TERMINAL)


def host_threads():
    """Threads this process may use: its CPU affinity, capped by OMP_NUM_THREADS when the environment sets one (a
    GPU box gives a one-GPU job a share of the machine, not all of it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        n = min(n, int(os.environ["OMP_NUM_THREADS"]))
    except (KeyError, ValueError):
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_sha():
    """Identifies the kernel sources a profile was taken with."""
    import hashlib
    h = hashlib.sha256()
    for f in ("rt_kernels.hip", "rt_device.h"):
        h.update(open(os.path.join(ROOT, "ray_tracing_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def live_counters(argv_config, timeout_s=90):
    """Hardware counters of the dominant kernel, measured now: this same script, as a child of rocprofv3, once per
    --pmc pass (the passes the MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE apart, the SQ counters in a pass of
    their own), each a few frames with one frame in flight and no one-shot frame.  Called before this process
    touches the GPU.  -> per-launch averages over the render kernel's dispatches, or None when anything goes wrong
    (no rocprofv3, a pass that fails or runs into its time limit): the line then falls back on the filed profile."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3")
    if rocprof is None:
        return None
    passes = [["FETCH_SIZE", "GRBM_GUI_ACTIVE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_LDS_IDX_ACTIVE"]]
    out = {}
    tmp = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for i, counters in enumerate(passes):
            d = os.path.join(tmp, f"p{i}")
            cmd = [rocprof, "--pmc", *counters, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--counters-child", "--steps", "3", "--warmup", "1", "--streams", "1", "--no-cpu-baseline", *argv_config]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            vals = {}
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"]
                    if re.search(r"trace_kernel<\d+, \d+, 0>", name):                     # MODE 0: the render kernel
                        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            for k, v in vals.items():
                v.sort()
                med = v[len(v) // 2]
                keep = [x for x in v if med == 0 or abs(x - med) <= 0.25 * abs(med)]
                out[k] = sum(keep) / len(keep)
        return out or None
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(cfg, target_seconds=12.0, gpu_render=None):
    """The CPU oracle (a restatement of the reference's tracer; the reference itself cannot be built
    here) timed on this host on a bounded sample of the same workload.  The frame it produces is
    also compared, word for word, with the GPU's frame of the same samples (gpu_render(spp) ->
    accumulator words): the PSNR of the headline metric."""
    from oracle import oracle as orc
    threads = host_threads()
    # the arithmetic the kernels repeat bit for bit: the recipes, or under RT_FLAG_LITERAL_GEOMETRY the literal geometry
    arith = orc.ARITH_LITERAL_GEOMETRY if cfg.flags & 8 else orc.ARITH_RECIPE
    p = orc.params(cfg.width, cfg.height, 1, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics,
                   sample_mode=cfg.sample_mode, flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, threads=threads, arith=arith)
    t0 = time.perf_counter()
    orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    t1 = time.perf_counter() - t0
    spp = int(max(1, min(cfg.spp, target_seconds / max(t1, 1e-3))))
    frames = int(max(1, min(64, target_seconds / max(t1 * spp, 1e-3))))
    p.spp = spp
    t0 = time.perf_counter()
    for _ in range(frames):
        acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    dt = time.perf_counter() - t0
    n = cfg.width * cfg.height * spp * frames
    out = {"value": round(n / dt / 1e6, 3), "unit": "Msamples/s", "cores": threads, "kind": "port",
           "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
           "flags": "gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -mfma -mavx2 -fopenmp (oracle/Makefile; not -march=native: the library is built on another machine than it runs on)",
           "sample": f"{frames} frame(s) of {cfg.width}x{cfg.height}, {spp} of {cfg.spp} spp, depth {cfg.max_depth}, "
                     f"same scene and seed; {dt:.1f} s on {threads} threads"}
    # The reference's own loop, std::for_each(par_unseq) (rayTracer.h:81), runs on one thread without TBB: one thread too.
    p1 = orc.params(cfg.width, cfg.height, max(1, spp * frames // (4 * threads)), cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics,
                    sample_mode=cfg.sample_mode, flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, threads=1, arith=arith)
    t0 = time.perf_counter()
    orc.render(cfg.shapes, cfg.triangles, cfg.camera, p1)
    dt1 = time.perf_counter() - t0
    out["one_thread"] = {"value": round(cfg.width * cfg.height * p1.spp / dt1 / 1e6, 3), "unit": "Msamples/s",
                         "sample": f"1 frame of {cfg.width}x{cfg.height}, {p1.spp} spp; {dt1:.1f} s"}
    if gpu_render is not None:
        import numpy as np
        got = gpu_render(spp)
        differing = int((got != acc).sum())
        a = orc.resolve(acc, spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))[..., :3].astype(np.float64)
        b = orc.resolve(got, spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))[..., :3].astype(np.float64)
        mse = float(((a - b) ** 2).mean())
        out["image_vs_cpu"] = {"frame": f"{cfg.width}x{cfg.height}, {spp} spp", "differing_words": differing,
                               "psnr_db": None if mse == 0.0 else round(10.0 * float(np.log10(max(a.max(), 1e-30) ** 2 / mse)), 2),
                               "note": "psnr_db null = the two frames are bit-identical (infinite PSNR)"}
        # The same frame once more in the reference's literal arithmetic (no FMA, IEEE divide, v / sqrt(v.v), libm
        # sin / cos): the GPU's frame against the source text rather than against the recipes it shares with the oracle.
        p.arith = orc.ARITH_LITERAL
        lit, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
        c = orc.resolve(lit, spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))[..., :3].astype(np.float64)
        mse = float(((c - b) ** 2).mean())
        out["image_vs_cpu"]["literal_arithmetic"] = {
            "psnr_db": None if mse == 0.0 else round(10.0 * float(np.log10(max(c.max(), 1e-30) ** 2 / mse)), 2),
            "max_abs": round(float(np.abs(c - b).max()), 6),
            "pixels_differing_percent": round(100.0 * float((np.abs(c - b).max(axis=-1) > 0).mean()), 4)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "c5e"])
    ap.add_argument("--shard", default="samples", choices=["samples", "rows"])
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per rank")
    ap.add_argument("--width", type=int, default=0, help="override the frame width (experiments)")
    ap.add_argument("--height", type=int, default=0, help="override the frame height (experiments)")
    ap.add_argument("--semantics", type=int, default=-1, choices=[-1, 0, 1], help="override the rule set (experiments): 0 CPU tracer, 1 shader")
    ap.add_argument("--flags", type=int, default=0, help="rt_flags to add to the configuration's (experiments): 8 = RT_FLAG_LITERAL_GEOMETRY")
    ap.add_argument("--mesh", default="", help="c3 only: the mesh file under ray_tracing_amd/data (default Bunny-LowPoly.off; bunny.off has 4 968 triangles)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="one GPU: frames in flight, each on its own context and stream (1: one after the other)")
    ap.add_argument("--settle-ms", type=float, default=250.0, help="untimed frames before the warm-up steps, for this long, so that the device has reached its sustained clock (0: none)")
    ap.add_argument("--no-live-counters", action="store_true", help="do not run the rocprofv3 --pmc passes; roofline.traffic then comes from profiles/traffic.json")
    ap.add_argument("--counters-child", action="store_true", help=argparse.SUPPRESS)   # this run is one of those passes
    args = ap.parse_args()

    # One GPU: the dominant kernel's counters, measured in this run by three short rocprofv3 passes over this script
    # (before this process touches the GPU).
    live = None
    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ)        # someone else's rocprofv3 is already watching this run
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.counters_child and not args.no_live_counters and not under_profiler:
        cfg_argv = ["--config", args.config] + (["--mesh", args.mesh] if args.mesh else []) + (["--spp", str(args.spp)] if args.spp else []) + \
                   (["--width", str(args.width), "--height", str(args.height)] if args.width and args.height else []) + \
                   (["--semantics", str(args.semantics)] if args.semantics >= 0 else []) + (["--flags", str(args.flags)] if args.flags else [])
        live = live_counters(cfg_argv)

    import torch
    import torch.distributed as dist
    import ray_tracing_amd as rta
    from ray_tracing_amd import parallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N > 1 path on a box with one GPU: RT_BENCH_BACKEND=gloo RT_BENCH_DEVICE=0 puts all
    # ranks on one device and exchanges through gloo.  Never set by the driver; the numbers mean nothing.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if "RT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["RT_BENCH_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))

    verbose = "RT_BENCH_VERBOSE" in os.environ

    def say(*a):
        if verbose:
            print(f"[rank {rank} {time.strftime('%H:%M:%S')}]", *a, file=sys.stderr, flush=True)

    say("process group up")
    overrides = {"width": args.width, "height": args.height} if args.width and args.height else {}
    if args.mesh:
        if args.config != "c3":
            raise SystemExit("--mesh applies to --config c3 only")
        overrides["mesh"] = args.mesh
    cfg = rta.scenes.CONFIGS[args.config](**overrides)
    if args.spp:
        cfg.spp = args.spp
    if args.semantics >= 0:
        cfg.semantics = args.semantics
    cfg.flags |= args.flags
    # Frames in flight.  One GPU: consecutive frames alternate between --streams contexts, each with its own
    # accumulator and HIP stream, so that the ramp and tail of one frame's launch are filled by the next frame's
    # (a progressive renderer queues its frames the same way).  Several GPUs: two accumulators on one stream, so that
    # the framebuffer exchange of frame k runs beside the trace of frame k+1.
    n_buf = max(1, args.streams) if world == 1 else 2
    streams = [torch.cuda.Stream() for _ in range(n_buf)] if world == 1 else [torch.cuda.current_stream()] * n_buf

    def measure(mode):
        """-> timings and counters of args.steps frames split over the ranks by `mode`."""
        shard = parallel.plan(mode, rank, world, cfg.spp)
        rts, accums, outs = [], [], []
        for i in range(n_buf):
            rt = rta.scenes.make_renderer(cfg, device=local_rank)
            rt.setStream(streams[i].cuda_stream)
            parallel.apply(rt, shard)
            accum = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
            rt.bindAccum(accum.data_ptr(), cfg.width, cfg.height)
            rts.append(rt); accums.append(accum)
            outs.append(torch.empty(cfg.height, cfg.width, 4, dtype=torch.float32, device="cuda"))
        torch.cuda.synchronize()

        gamma = cfg.semantics == rta.SEM_CPU

        def trace(k, ev=None):
            rt = rts[k % n_buf]
            rt.discardAccum()                     # the frame starts from zero: the launch overwrites the accumulator, nothing is cleared
            if ev:
                ev[0].record(streams[k % n_buf])
            if world == 1:
                # one GPU: the frame's resolve rides on the launch that ends it (rt_render_resolved)
                rt.renderResolved(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, outs[k % n_buf].data_ptr(),
                                  first_sample=shard.first_sample, gamma_sqrt=gamma, flip_rows=True)
            else:
                rt.render(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, first_sample=shard.first_sample)
            if ev:
                ev[1].record(streams[k % n_buf])
            return k, parallel.exchange_framebuffer(accums[k % n_buf], cfg.semantics, shard, async_op=True)

        def finish(pending):
            k, work = pending
            if work is not None:
                work.wait()                       # the current stream waits for the collective
            if rank == 0 and world > 1:
                rt = rts[k % n_buf]
                rt.setAccumSamples(parallel.total_samples(shard))
                rt.resolveDevice(outs[k % n_buf].data_ptr(), gamma_sqrt=gamma, flip_rows=True)

        def run(n, events=None, one_context=False):
            pending = None
            for k in range(n):
                nxt = trace(0 if one_context else k, events[k] if events else None)
                say(mode, "frame", k, "queued")
                if pending is not None:
                    finish(pending)               # frame k-1: after frame k's trace is queued
                    say(mode, "frame", k - 1, "finished")
                if world == 1:
                    finish(nxt); nxt = None
                pending = nxt
            if pending is not None:
                finish(pending)

        # Work counters of one launch (same inputs every step -> same counts); not timed.
        rts[0].resetAccum()
        stats = rts[0].render(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, first_sample=shard.first_sample, stats=True)
        say(mode, "stats launch done")
        # A one-shot call: a fresh context (no plan learned from an earlier frame), one render of the whole workload,
        # waited for.  The process has run the kernel before (the stats launch), so its code is loaded.
        cold_ms = None
        if world == 1 and not args.counters_child:
            rt_cold = rta.scenes.make_renderer(cfg, device=local_rank)
            rt_cold.setStream(streams[0].cuda_stream)
            accums[0].zero_()
            rt_cold.bindAccum(accums[0].data_ptr(), cfg.width, cfg.height)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            rt_cold.render(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, first_sample=shard.first_sample)
            torch.cuda.synchronize()
            cold_ms = (time.perf_counter() - tc) * 1e3
            rt_cold.bindAccum(None, 0, 0)
            rt_cold.close()
        # Device clocks.  After the idle stretch above (contexts created and freed) the GPU runs below its sustained
        # clock and takes some 20 ms of load to reach it (tools/frame_gaps.py: frames of C2 last 1.20 ms, then 1.15,
        # ... 1.03 from the twentieth on); K timed steps of a millisecond each would measure that ramp, not the
        # renderer.  So the same frames run untimed for --settle-ms first (at most 2 000 of them) — every rank the same number, the count
        # derived from the slowest rank's time for the first few — and the line says how many ("clock_settle_frames").
        # The same K steps once before that, as a run that does not wait for the clocks measures them ("value_unsettled").
        unsettled_ms = None
        if not args.counters_child and args.settle_ms > 0:
            run(args.warmup)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tu = time.perf_counter()
            run(args.steps)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            tu = torch.tensor([time.perf_counter() - tu], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(tu, op=dist.ReduceOp.MAX)
            unsettled_ms = float(tu.item()) / args.steps * 1e3
        settle_frames = 0
        if not args.counters_child and args.settle_ms > 0:
            torch.cuda.synchronize()
            ts = time.perf_counter()
            run(4)
            torch.cuda.synchronize()
            per = torch.tensor([(time.perf_counter() - ts) / 4.0], dtype=torch.float64, device="cuda")
            if world > 1:
                dist.all_reduce(per, op=dist.ReduceOp.MAX)
            settle_frames = int(min(2000, max(0, round(args.settle_ms * 1e-3 / max(float(per.item()), 1e-6)) - 4)))
            run(settle_frames)
            settle_frames += 4
        run(args.warmup)
        say(mode, "warmup done")
        if world > 1:
            dist.barrier()
        say(mode, "barrier passed")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        say(mode, "timed steps done")
        # The dominant kernel's launch duration, from HIP events on the stream it is launched on: the same K frames
        # once more, one frame in flight (with two in flight a launch shares the machine and its own duration says
        # nothing), outside the timed region.
        events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        run(args.steps, events, one_context=(world == 1))
        torch.cuda.synchronize()
        kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, args.steps)
        # What a frame costs the host to queue, and how long the exchange takes: the same frames once more, one at a
        # time with the device idle before each (nothing overlaps, so the parts can be told apart).  exchange_ms runs
        # from the end of this rank's trace to the end of the collective on this rank: it includes waiting for the
        # slowest rank's trace.
        host_ms, exchange_ms = [], []
        for k in range(min(args.steps, 8)):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            th = time.perf_counter()
            rt = rts[k % n_buf]
            rt.discardAccum()
            if world == 1:
                rt.renderResolved(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, outs[k % n_buf].data_ptr(),
                                  first_sample=shard.first_sample, gamma_sqrt=gamma, flip_rows=True)
            else:
                rt.render(cfg.width, cfg.height, shard.spp, cfg.max_depth, cfg.seed, first_sample=shard.first_sample)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(streams[k % n_buf])
            work = parallel.exchange_framebuffer(accums[k % n_buf], cfg.semantics, shard, async_op=True)
            finish((k, work))
            e1.record(streams[k % n_buf])
            host_ms.append((time.perf_counter() - th) * 1e3)
            torch.cuda.synchronize()
            exchange_ms.append(e0.elapsed_time(e1))
        host_ms_per_frame = sum(host_ms) / len(host_ms)
        exchange_ms = sum(exchange_ms) / len(exchange_ms) if world > 1 else None

        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        samples = torch.tensor([stats["samples"]], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(samples)
        m = {"shard": shard, "stats": stats, "elapsed": float(t.item()), "samples_per_step": int(samples.item()),
             "kernel_ms": kernel_ms, "cold_ms": cold_ms, "rts": rts, "accums": accums, "settle_frames": settle_frames,
             "unsettled_ms": unsettled_ms, "host_ms_per_frame": host_ms_per_frame, "exchange_ms": exchange_ms}
        return m

    def release(m):
        for rt in m["rts"]:
            rt.bindAccum(None, 0, 0)
            rt.close()

    # With several GPUs both splits are measured: the headline line is --shard's, the other one rides along.
    other = None
    if world > 1:
        other_mode = "rows" if args.shard == "samples" else "samples"
        mo = measure(other_mode)
        other = {"shard": other_mode, "scaling": "weak" if other_mode == "samples" else "strong",
                 "value": round(mo["samples_per_step"] / (mo["elapsed"] / args.steps) / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(mo["elapsed"] / args.steps * 1e3, 4), "kernel_ms": round(mo["kernel_ms"], 4),
                 "exchange_ms": None if mo["exchange_ms"] is None else round(mo["exchange_ms"], 4),
                 "host_ms_per_frame": round(mo["host_ms_per_frame"], 4),
                 "exchange_bytes_per_rank": parallel.exchange_bytes(other_mode, cfg.width, cfg.height, world)}
        release(mo)
    m = measure(args.shard)
    shard, stats, elapsed, samples_per_step, kernel_ms, cold_ms = m["shard"], m["stats"], m["elapsed"], m["samples_per_step"], m["kernel_ms"], m["cold_ms"]
    rts, accums = m["rts"], m["accums"]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = samples_per_step / (elapsed / args.steps) / 1e6
        flops = algorithmic_flops(stats)
        executed_flops = algorithmic_flops(stats, spheres_executed=True)
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        # Counters of the same launch from the committed rocprofv3 passes (profiles/traffic.json): HBM bytes, and
        # the vector instructions issued, which price the launch in issue slots of the vector pipe.
        traffic, issue, traffic_note, bound = None, None, None, "valu"
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(f"{args.config}_{cfg.width}x{cfg.height}_{shard.spp}spp", {}) if not args.mesh else {}
                traffic = tj.get("hbm_bytes_per_launch")
                bound = tj.get("bound", "valu")               # "lds" where the profile shows the LDS busier than the vector pipe
                if traffic is not None:
                    traffic_note = {"source": "replayed from " + str(tj.get("source", "profiles/")) + " (rocprofv3 --pmc passes of this "
                                              "command; not measured in this run)",
                                    "profiled_kernel_sha": tj.get("kernel_sha"), "kernel_sha": kernel_source_sha(),
                                    "traffic_stale": tj.get("kernel_sha") != kernel_source_sha()}
                if tj.get("valu_instructions_per_launch"):
                    n_inst = tj["valu_instructions_per_launch"]
                    cyc = kernel_ms * 1e-3 * PEAK_CLOCK_HZ * N_SIMD / n_inst
                    issue = {"valu_instructions_per_launch": n_inst, "cycles_per_instruction_per_simd": round(cyc, 3),
                             "best_cycles_per_instruction": 2.0, "issue_slots_used": round(2.0 / cyc, 4),
                             "lanes_enabled_percent": tj.get("valu_lane_utilisation_percent"),
                             "mean_waves_per_simd": tj.get("mean_waves_per_simd"), "lds_busy_percent": tj.get("lds_busy_percent"),
                             "source": tj.get("source"),
                             "note": "1024 SIMDs at 2.4 GHz; a wave64 fp32 instruction issues in 2 cycles at best, compares / selects / "
                                     "conversions / integer multiplies in 4, sqrt / rcp in 8"}
            except Exception:
                traffic, issue = None, None
        if live and "FETCH_SIZE" in live and "WRITE_SIZE" in live:
            # measured in this run: (2 * FETCH_SIZE + WRITE_SIZE) KB — on gfx950 FETCH_SIZE tallies the 128-byte requests
            # of 16-byte-per-lane reads at 64 bytes
            traffic = round((2.0 * live["FETCH_SIZE"] + live["WRITE_SIZE"]) * 1024.0)
            traffic_note = {"source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command, 3 frames each, "
                                      "(2 * FETCH_SIZE + WRITE_SIZE) * 1024", "fetch_size_kb": round(live["FETCH_SIZE"], 1),
                            "write_size_kb": round(live["WRITE_SIZE"], 1), "kernel_sha": kernel_source_sha(), "traffic_stale": False}
        if live and live.get("SQ_INSTS_VALU"):
            n_inst = live["SQ_INSTS_VALU"]
            cyc = kernel_ms * 1e-3 * PEAK_CLOCK_HZ * N_SIMD / n_inst
            lds_busy = None
            if live.get("SQ_LDS_IDX_ACTIVE") and live.get("GRBM_GUI_ACTIVE"):
                lds_busy = round(100.0 * live["SQ_LDS_IDX_ACTIVE"] / (live["GRBM_GUI_ACTIVE"] / 8.0 * 256.0), 1)
            issue = {"valu_instructions_per_launch": round(n_inst), "cycles_per_instruction_per_simd": round(cyc, 3),
                     "best_cycles_per_instruction": 2.0, "issue_slots_used": round(2.0 / cyc, 4),
                     "lanes_enabled_percent": round(100.0 * live["SQ_THREAD_CYCLES_VALU"] / (64.0 * live["SQ_ACTIVE_INST_VALU"]), 1)
                     if live.get("SQ_ACTIVE_INST_VALU") and live.get("SQ_THREAD_CYCLES_VALU") else None,
                     "lds_busy_percent": lds_busy, "source": "measured in this run (rocprofv3 --pmc pass of the SQ counters)",
                     "note": "1024 SIMDs at 2.4 GHz; a wave64 fp32 instruction issues in 2 cycles at best, compares / selects / "
                             "conversions / integer multiplies in 4, sqrt / rcp in 8"}
            if lds_busy is not None:
                bound = "lds" if lds_busy / 100.0 > 2.0 / cyc else "valu"
        info = rta.device_info(local_rank)
        result = {
            "metric": "Msamples/sec at 1080p", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            # the same K steps timed before the device's clocks had settled (no --settle-ms): what a short cold run sees
            "value_unsettled": None if m["unsettled_ms"] is None else round(samples_per_step / (m["unsettled_ms"] * 1e-3) / 1e6, 2),
            "ms_per_step_unsettled": None if m["unsettled_ms"] is None else round(m["unsettled_ms"], 4),
            # host time to queue one frame (discard, render, exchange, resolve; the device idle, nothing waited for)
            "host_ms_per_frame": round(m["host_ms_per_frame"], 4),
            # N > 1: from the end of this rank's trace to the end of the framebuffer exchange on rank 0 (HIP events)
            "exchange_ms": None if m["exchange_ms"] is None else round(m["exchange_ms"], 4),
            "higher_is_better": True, "scaling": "weak" if args.shard == "samples" else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg.name}: {cfg.width}x{cfg.height}, {shard.spp} spp per GPU, depth {cfg.max_depth}, "
                                   f"{len(cfg.shapes)} shapes + {len(cfg.triangles)} triangles, "
                                   f"{'cpu_version' if cfg.semantics == rta.SEM_CPU else 'RayTracing.glsl'} semantics, seed {cfg.seed}",
                       "shard": args.shard, "spp_total": parallel.total_samples(shard), "device": info["name"],
                       "frames_in_flight": n_buf,
                       "exchange_bytes_per_rank": parallel.exchange_bytes(args.shard, cfg.width, cfg.height, world)},
            "mrays_per_s": round(stats["segments"] * world / (elapsed / args.steps) / 1e6, 2),
            "segments_per_sample": round(stats["segments"] / max(1, stats["samples"]), 4),
            # share of the lane-iterations of the bounce loop that traced a segment (the rest: lanes whose pixel had
            # no samples left while their wave went on)
            "lane_use": round(stats["segments"] / max(1, 64 * stats["wave_iterations"]), 4),
            # steps of a wave that traced camera rays only, from their tile's candidate list (instances with boxes or a mesh)
            "primary_steps_share": round(stats["primary_steps"] / max(1, stats["wave_iterations"]), 4),
            "cold_first_frame_ms": None if cold_ms is None else round(cold_ms, 4),
            # untimed frames run before the warm-up steps so that the device is at its sustained clock (--settle-ms)
            "clock_settle_frames": m["settle_frames"],
            "roofline": {"bound": bound, "achieved": round(achieved, 3), "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                         "kernel": "trace_kernel", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_flops_per_launch": flops,
                         "executed_flops_per_launch": executed_flops,
                         "achieved_executed": round(executed_flops / (kernel_ms * 1e-3) / 1e12, 3),
                         "frac_executed": round(executed_flops / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_VECTOR_TFLOPS, 4),
                         "exhaustive_scan_flops_per_launch": algorithmic_flops(stats, executed=False),
                         "algorithmic_hbm_bytes_per_launch": cfg.width * cfg.height * 32,
                         "valu_issue": issue,
                         "note": "fp32 vector (VALU) bound: no dense contraction, no MFMA; HBM traffic is the framebuffer only. "
                                 "achieved is the algorithmic work of the launch (measured segments and events; every sphere for every "
                                 "segment, as the reference's closest-hit loop scans them; triangle, box and bound tests as executed: "
                                 "those are culled by a tree the reference does not have) over the kernel's time. achieved_executed / "
                                 "frac_executed count the sphere tests actually run (camera rays test only the spheres their tile can "
                                 "see), exhaustive_scan_flops_per_launch the reference's scan of every primitive"},
        }
        if other is not None:
            result["other_split"] = other
        if world == 1 and not args.no_cpu_baseline:
            def gpu_render(spp):
                rt = rts[0]
                rt.discardAccum()
                rt.render(cfg.width, cfg.height, spp, cfg.max_depth, cfg.seed)
                torch.cuda.synchronize()
                return accums[0].cpu().numpy().view("uint32")
            # The CPU legs are single library calls of up to a minute each and print nothing; a harness that takes a
            # silent run for a hung one gets a line on stderr every half minute.
            import threading
            done = threading.Event()

            def heartbeat():
                while not done.wait(30.0):
                    print("[bench] CPU baseline running", file=sys.stderr, flush=True)
            threading.Thread(target=heartbeat, daemon=True).start()
            try:
                result["cpu_baseline"] = cpu_baseline(cfg, gpu_render=gpu_render)
            finally:
                done.set()
        print(json.dumps(result), flush=True)

    release(m)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
