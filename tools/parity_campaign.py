"""A randomised parity campaign: many small random scenes, every one rendered by the HIP path (through the C ABI)
and by the CPU oracle and compared word for word — work counters, every accumulated image word, and the hit and
material ids of one followed sample.  Wider than tests/test_gpu_parity.py::test_random_scenes: both rule sets, all
sample modes, both draw streams, the flags, cubes, dielectrics, debug and unknown materials, boxes with their own
sides, triangle colours, meshes from none to a few thousand triangles (staged in LDS or read from global memory),
cameras inside and outside the scene, ragged frame sizes, depths 1 .. 9.

    python tools/parity_campaign.py [--calls [--big] | --large-mesh] [n_scenes] [first_seed]       # needs an MI355X; prints one line per failure

With --calls every scene (64 .. 100 samples per pixel) goes through every way of calling the renderer instead (see
run_calls) and the accumulators are compared.

The oracle is the checker here, never the thing measured."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import ray_tracing_amd as rta
from ray_tracing_amd import _abi, host
from oracle import oracle as orc

COUNTERS = ("samples", "segments", "diffuse", "specular", "terminal",
            "shape_tests", "sphere_tests", "box_tests", "triangle_tests")


def random_config(seed):
    rng = np.random.default_rng(77000 + seed)
    sem = int(rng.integers(0, 2))
    flags = 0
    if sem == 0 and rng.random() < 0.25:
        flags |= _abi.FLAG_METAL_UNNORMALIZED
    if rng.random() < 0.3:
        flags |= _abi.FLAG_BOX_SIDES
    rng_mode = _abi.RNG_GLSL if rng.random() < 0.2 else _abi.RNG_HASH
    if sem == 0 and rng_mode == _abi.RNG_HASH and rng.random() < 0.2:
        flags |= _abi.FLAG_UNIT_GAUSS
    mode = int(rng.choice([_abi.SAMPLE_JITTER, _abi.SAMPLE_JITTER, _abi.SAMPLE_GRID, _abi.SAMPLE_GRID_COMPAT]))
    grid_n = int(rng.integers(1, 4)) if mode != _abi.SAMPLE_JITTER else 0
    spp = grid_n * grid_n if grid_n else int(rng.integers(1, 7))

    def colour(lo=0.2, hi=1.0):
        return tuple(float(x) for x in rng.uniform(lo, hi, 3))

    def material():
        kind = int(rng.integers(0, 10))
        if kind == 0:
            return host.Light(colour(1.0, 4.0))
        if kind in (1, 2):
            return host.Metal(colour(), fuzz=float(rng.choice([0.0, 0.0, 0.15, 0.6]))) if sem == 0 else host.Metal(colour())
        if kind == 3:
            return host.Dielectric(colour(0.7, 1.0), ior=float(rng.uniform(1.1, 2.0)))
        if kind == 4 and rng.random() < 0.5:
            return host.DebugMaterial(colour())
        return host.Lambertian(colour())

    spread = float(rng.choice([1.0, 1.0, 3.0, 30.0]))

    def point():
        return (float(rng.uniform(-3, 5) * spread), float(rng.uniform(0, 4) * spread), float(rng.uniform(-9, -1) * spread))

    objs = []
    if rng.random() < 0.7:
        objs.append(host.Box((0, -50, 0), (100, 1, 1), host.Lambertian(colour())))
    if rng.random() < 0.8:
        objs.append(host.Circle((1, 7 * spread, -4), 2.5 * spread, host.Light((3, 3, 3))))
    n_sph = int(rng.choice([0, 1, 3, 4, 5, 8, 13, 40]))
    n_box = int(rng.choice([0, 0, 1, 3, 7, 20]))
    if seed >= CROWDED_FROM:                                  # scenes near what the LDS stage holds (64 KiB)
        n_sph = int(rng.choice([0, 150, 400, 600, 1200, 1500]))
        n_box = int(rng.choice([0, 60, 150, 300, 900]))
        room = 160 * 1024 - 36 * 1024 - 90 * n_sph                # a workgroup's LDS less the waves' queues and a staged mesh
        n_box = max(0, min(n_box, room // 120))
    n_cube = int(rng.choice([0, 0, 1, 2])) if sem == 0 else 0
    for _ in range(n_sph):
        objs.append(host.Circle(point(), float(rng.uniform(0.2, 1.0) * spread), material()))
    for _ in range(n_box):
        sides = tuple(float(x) for x in rng.uniform(0.3, 1.5, 3) * spread) if rng.random() < 0.5 else (float(rng.uniform(0.3, 1.5) * spread),) * 3
        objs.append(host.Box(point(), sides, material()))
    for _ in range(n_cube):
        c = host.Cube(point(), material())
        c.rotate(float(rng.uniform(-3, 3)))
        objs.append(c)
    rng.shuffle(objs)
    shapes = host.pack(objs, _abi.SHAPE_DTYPE) if objs else np.zeros(0, dtype=_abi.SHAPE_DTYPE)
    if len(shapes) and rng.random() < 0.1:
        shapes = shapes.copy()
        shapes["material"]["type"][int(rng.integers(0, len(shapes)))] = 55          # a material type the shader does not know

    n_tris = int(rng.choice([0, 0, 1, 3, 5, 37, 150, 292, 700, 2500]))
    tris = []
    size = float(rng.choice([0.1, 0.6, 2.0])) * spread
    for _ in range(n_tris):
        a = np.array(point())
        e1, e2 = rng.uniform(-size, size, 3), rng.uniform(-size, size, 3)
        if rng.random() < 0.02:
            e2 = e1 * float(rng.uniform(-2, 2))                                      # a degenerate one
        tris.append(host.SimpleTriangle(tuple(a), tuple(a + e1), tuple(a + e2)))
    triangles = host.pack(tris, _abi.TRIANGLE_DTYPE) if tris else np.zeros(0, dtype=_abi.TRIANGLE_DTYPE)
    tcol = rng.uniform(0.1, 1.0, (n_tris, 3)).astype(np.float32) if n_tris and rng.random() < 0.3 else None

    w, h = int(rng.integers(1, 120)), int(rng.integers(1, 70))
    if seed >= CROWDED_FROM:
        w, h, spp = int(rng.integers(1, 40)), int(rng.integers(1, 24)), (spp if grid_n else int(rng.integers(1, 3)))
    cam_pos = (1.0, 1.0, 6.0) if rng.random() < 0.6 else tuple(float(x) for x in np.array(point()) * rng.choice([1.0, 0.2]))
    cam = host.Camera(w, h, float(rng.choice([1.0, 0.5, 2.0])), cam_pos)
    if rng.random() < 0.3:
        cam.rotate(float(rng.uniform(-0.8, 0.8)), float(rng.uniform(-0.4, 0.4)))
    cfg = rta.scenes.SceneConfig(f"campaign{seed}", shapes, triangles, cam.data.copy(), w, h, spp, int(rng.integers(1, 10)), sem,
                                 seed=int(rng.integers(0, 2 ** 40)), sample_mode=mode, flags=flags,
                                 triangle_albedo=colour(), rng=rng_mode, triangle_colors=tcol)
    stage = bool(n_tris and n_tris <= 300 and rng.random() < 0.4)
    if seed >= EXTENDED_FROM:
        extend(cfg, np.random.default_rng(55000 + seed))
    return cfg, grid_n, stage


EXTENDED_FROM = 100000      # seeds from here on also get what extend() adds (earlier seeds keep their scenes)
CROWDED_FROM = 400000       # and from here on hundreds of spheres and boxes on small frames


def extend(cfg, r):
    """Ties and edges: copies of shapes and triangles with another colour somewhere else in the arrays (equal
    distances: the first in upload order must win), shapes that touch or contain the camera, 64-bit seeds, a first
    sample far into the stream, an occasional long path."""
    if len(cfg.shapes) and r.random() < 0.6:
        for _ in range(int(r.integers(1, 4))):
            dup = cfg.shapes[int(r.integers(0, len(cfg.shapes)))].copy()
            dup["material"]["color"] = r.uniform(0.1, 1.0, 3).astype(np.float32)
            if r.random() < 0.3:
                dup["material"]["type"] = int(r.choice([0, 1, 2]))
            cfg.shapes = np.insert(cfg.shapes, int(r.integers(0, len(cfg.shapes) + 1)), dup)
    if len(cfg.triangles) and r.random() < 0.5:
        for _ in range(int(r.integers(1, 6))):
            at, src = int(r.integers(0, len(cfg.triangles) + 1)), int(r.integers(0, len(cfg.triangles)))
            cfg.triangles = np.insert(cfg.triangles, at, cfg.triangles[src].copy())
            if cfg.triangle_colors is not None:
                cfg.triangle_colors = np.insert(cfg.triangle_colors, at, r.uniform(0.1, 1.0, 3).astype(np.float32), axis=0)
    if r.random() < 0.15:                                     # a sphere around the camera, or one it sits on
        pos = cfg.camera[0:3].astype(np.float64)
        radius = float(r.choice([0.5, 3.0, 40.0]))
        centre = pos if r.random() < 0.5 else pos + np.array([radius, 0, 0])
        extra = host.pack([host.Circle(tuple(centre), radius, host.Lambertian((0.6, 0.6, 0.6)))], _abi.SHAPE_DTYPE)
        cfg.shapes = np.concatenate([cfg.shapes, extra]) if len(cfg.shapes) else extra
    cfg.seed = int(r.integers(0, 2 ** 64, dtype=np.uint64))
    if cfg.sample_mode == _abi.SAMPLE_JITTER and r.random() < 0.5:
        cfg.notes["first_sample"] = int(r.choice([1, 99, 2 ** 24 - 3, 2 ** 31 + 5, 2 ** 32 - 1 - cfg.spp]))
    if r.random() < 0.05:
        cfg.max_depth = int(r.choice([33, 64]))


def run_one(seed):
    cfg, grid_n, stage = random_config(seed)
    if stage:
        os.environ["RT_MESH_IN_LDS"] = "1"
    else:
        os.environ.pop("RT_MESH_IN_LDS", None)
    rt = rta.scenes.make_renderer(cfg)
    problems = []
    try:
        first = cfg.notes.get("first_sample", 0)
        stats = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, first_sample=first, grid_n=grid_n, stats=True)
        img = rt.resolve(cfg.width, cfg.height, gamma_sqrt=(cfg.semantics == 0))
        p = orc.params(cfg.width, cfg.height, cfg.spp, cfg.max_depth, seed=cfg.seed, semantics=cfg.semantics,
                       sample_mode=cfg.sample_mode, flags=cfg.flags, triangle_albedo=cfg.triangle_albedo, rng=cfg.rng,
                       triangle_colors=cfg.triangle_colors, grid_n=grid_n, first_sample=first)
        acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
        ref = orc.resolve(acc, cfg.spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))
        bad = [k for k in COUNTERS if stats[k] != ostats[k]]
        if bad:
            problems.append("counters " + ", ".join(f"{k} {stats[k]} != {ostats[k]}" for k in bad))
        diff = img.view(np.uint32) != ref.view(np.uint32)
        if diff.any():
            problems.append(f"{int(diff.sum())} of {diff.size} image words")
        s = first + cfg.spp - 1
        hit, mat = rt.traceDebug(cfg.width, cfg.height, cfg.max_depth, cfg.seed, s, grid_n=grid_n)
        ohit, omat = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, s)
        if not np.array_equal(hit, ohit):
            problems.append(f"{int((hit != ohit).sum())} hit ids of sample {s}")
        if not np.array_equal(mat, omat):
            problems.append(f"{int((mat != omat).sum())} material ids of sample {s}")
        # the plain render (no counters; the pooled instance under the shader's rules) gives the same words
        rt.resetAccum()
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, first_sample=first, grid_n=grid_n)
        img2 = rt.resolve(cfg.width, cfg.height, gamma_sqrt=(cfg.semantics == 0))
        if not np.array_equal(img2.view(np.uint32), ref.view(np.uint32)):
            problems.append(f"{int((img2.view(np.uint32) != ref.view(np.uint32)).sum())} image words of the plain render")
    finally:
        rt.close()
    return cfg, stats, problems


def run_calls(seed):
    """The same scene through every way of calling the renderer, each compared with the oracle's accumulator word for
    word: a fresh context's one-shot call of 64+ samples (probe, plan, slices under the shader's rules), the same call
    again (now by the plan of the first), the samples in three calls, the rows in shards of random band height, the
    call that allocates nothing; then three frames through the running-mean converter against the oracle's."""
    import torch
    cfg, _, stage = random_config(seed)
    rng = np.random.default_rng(991000 + seed)
    cfg.sample_mode = _abi.SAMPLE_JITTER
    cfg.spp = int(rng.integers(64, 101))
    cfg.max_depth = min(cfg.max_depth, 6)
    cfg.width, cfg.height = int(rng.integers(1, 73)), int(rng.integers(1, 49))
    if BIG_FRAMES:                                            # hundreds of tiles: plans with many entries and slices
        cfg.width, cfg.height, cfg.spp = int(rng.integers(100, 401)), int(rng.integers(60, 301)), int(rng.integers(64, 81))
        cfg.max_depth = min(cfg.max_depth, 4)
        if len(cfg.triangles) > 150:
            cfg.triangles = cfg.triangles[:150]
            cfg.triangle_colors = None if cfg.triangle_colors is None else cfg.triangle_colors[:150]
    cam = host.Camera(cfg.width, cfg.height, 1.0, tuple(float(x) for x in cfg.camera[0:3]))
    cfg.camera = cam.data.copy()
    if len(cfg.triangles) > 300:
        cfg.triangles = cfg.triangles[:300]
        cfg.triangle_colors = None if cfg.triangle_colors is None else cfg.triangle_colors[:300]
    if stage:
        os.environ["RT_MESH_IN_LDS"] = "1"
    else:
        os.environ.pop("RT_MESH_IN_LDS", None)
    w, h, spp, depth = cfg.width, cfg.height, cfg.spp, cfg.max_depth
    p = orc.params(w, h, spp, depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode, flags=cfg.flags,
                   triangle_albedo=cfg.triangle_albedo, rng=cfg.rng, triangle_colors=cfg.triangle_colors)
    acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    ref = np.ascontiguousarray(acc).view(np.uint32).reshape(h, w, 4)[..., :3]
    problems = []
    buf = torch.zeros(h, w, 4, dtype=torch.int32, device="cuda")

    def check(what):
        torch.cuda.synchronize()
        got = buf.cpu().numpy().view(np.uint32)[..., :3]
        if not np.array_equal(got, ref):
            problems.append(f"{what}: {int((got != ref).sum())} of {ref.size} accumulator words")
        buf.zero_()
        torch.cuda.synchronize()

    rt = rta.scenes.make_renderer(cfg)
    try:
        rt.bindAccum(buf.data_ptr(), w, h)
        rt.render(w, h, spp, depth, cfg.seed)
        check("one-shot call of a fresh context")
        rt.render(w, h, spp, depth, cfg.seed)
        check("the same call by the first one's plan")
        a = int(rng.integers(1, spp - 1)); b = int(rng.integers(a + 1, spp))
        for first, n in ((0, a), (a, b - a), (b, spp - b)):
            rt.render(w, h, n, depth, cfg.seed, first_sample=first)
        check(f"three calls ({a}, {b - a}, {spp - b} samples)")
        shards, band = int(rng.integers(2, 5)), int(rng.choice([1, 3, 8, 16, 40]))
        for g in range(shards):
            rt.setRowShard(g, shards, band)
            rt.render(w, h, spp, depth, cfg.seed)
        rt.setRowShard(0, 1, 16)
        check(f"{shards} row shards of {band}-row bands")
        rt.renderNoAlloc(buf.data_ptr(), w, h, spp, depth, cfg.seed)
        check("the call that allocates nothing")
        # a frame that starts from a discarded accumulator and ends with its resolve (rt_discard_accum, rt_render_resolved)
        out = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        want = orc.resolve(acc, spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))
        for rep in range(2):
            buf.fill_(987654321)
            rt.discardAccum()
            rt.renderResolved(w, h, spp, depth, cfg.seed, out.data_ptr(), gamma_sqrt=(cfg.semantics == 0))
            torch.cuda.synchronize()
            if not np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)):
                problems.append(f"discard + render_resolved, call {rep}: the resolved frame differs")
            check(f"discard + render_resolved, call {rep}")
        rt.bindAccum(None, 0, 0)
        # main.cpp:135-151: three frames through the running-mean converter, against the oracle's PixelConversion
        k = int(rng.integers(1, 6))
        rt.samples_per_frame = k
        image = np.zeros((h, w, 4), np.float32)
        for n in (1, 2, 3):
            rt.updateCameraBuffer()
            rt.compute(w, h, 1)
            rt.convert(n)
            pk = orc.params(w, h, k, depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode, flags=cfg.flags,
                            triangle_albedo=cfg.triangle_albedo, rng=cfg.rng, triangle_colors=cfg.triangle_colors, first_sample=(n - 1) * k)
            acc_k, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pk)
            orc.convert(acc_k, image, k, cfg.semantics, n)
        got = rt.readImage()
        if not np.array_equal(got.view(np.uint32), image.view(np.uint32)):
            problems.append(f"running-mean converter, 3 frames of {k}: {int((got.view(np.uint32) != image.view(np.uint32)).sum())} image words")
    finally:
        rt.close()
        os.environ.pop("RT_MESH_IN_LDS", None)
    return cfg, ostats, problems


def run_large_mesh(seed):
    """A one-shot render of a fresh context under the shader's rules, 64 .. 80 samples per pixel, of main.cpp's shapes
    and 801 .. 4 968 faces of bunny.off seen from a random eye: the call whose probe is planned from the camera rays'
    candidate lists (and whose rest runs by the plan made from that probe), then the same call again by the plan the
    first one left, accumulators and counters against the oracle's."""
    import torch
    rng = np.random.default_rng(555000 + seed)
    cfg = rta.scenes.config3(width=int(rng.integers(48, 129)), height=int(rng.integers(32, 81)), spp=int(rng.integers(64, 81)),
                             max_depth=int(rng.integers(2, 6)), mesh="bunny.off")
    keep = int(rng.integers(801, len(cfg.triangles) + 1))
    start = int(rng.integers(0, len(cfg.triangles) - keep + 1))
    cfg.triangles = cfg.triangles[start:start + keep]
    eye = (float(rng.uniform(-3, 4)), float(rng.uniform(0.3, 3)), float(rng.uniform(2, 8)))
    cam = host.Camera(cfg.width, cfg.height, float(rng.choice([0.7, 1.0, 1.6])), eye)
    cfg.camera = cam.data.copy()
    cfg.seed = int(rng.integers(1, 1 << 31))
    w, h, spp, depth = cfg.width, cfg.height, cfg.spp, cfg.max_depth
    p = orc.params(w, h, spp, depth, seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode, flags=cfg.flags,
                   triangle_albedo=cfg.triangle_albedo, rng=cfg.rng, triangle_colors=cfg.triangle_colors)
    acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    ref = np.ascontiguousarray(acc).view(np.uint32).reshape(h, w, 4)[..., :3]
    problems = []
    buf = torch.zeros(h, w, 4, dtype=torch.int32, device="cuda")
    rt = rta.scenes.make_renderer(cfg)
    try:
        rt.bindAccum(buf.data_ptr(), w, h)
        for what in ("one-shot call of a fresh context", "the same call by the first one's plan"):
            st = rt.render(w, h, spp, depth, cfg.seed, stats=True)
            torch.cuda.synchronize()
            got = buf.cpu().numpy().view(np.uint32)[..., :3]
            if not np.array_equal(got, ref):
                problems.append(f"{what}: {int((got != ref).sum())} of {ref.size} accumulator words")
            bad = [k for k in COUNTERS if st[k] != ostats[k]]
            if bad:
                problems.append(f"{what}: counters {bad}")
            buf.zero_()
            torch.cuda.synchronize()
        rt.bindAccum(None, 0, 0)
    finally:
        rt.close()
    return cfg, ostats, problems


BIG_FRAMES = False


def main():
    global BIG_FRAMES
    calls = "--calls" in sys.argv
    if calls:
        sys.argv.remove("--calls")
    if "--big" in sys.argv:                                   # with --calls: frames of 100x60 .. 400x300 pixels
        sys.argv.remove("--big")
        BIG_FRAMES = True
    large = "--large-mesh" in sys.argv
    if large:
        sys.argv.remove("--large-mesh")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    failed = 0
    tot = dict(samples=0, segments=0, scenes=0, with_tris=0, with_boxes=0, sem0=0)
    for seed in range(first, first + n):
        cfg, stats, problems = run_large_mesh(seed) if large else run_calls(seed) if calls else run_one(seed)
        tot["scenes"] += 1
        tot["samples"] += stats["samples"]; tot["segments"] += stats["segments"]
        tot["with_tris"] += 1 if len(cfg.triangles) else 0
        tot["with_boxes"] += 1 if (cfg.shapes["type"] == 1).any() else 0
        tot["sem0"] += 1 if cfg.semantics == 0 else 0
        if problems:
            failed += 1
            print(f"FAIL seed {seed}: {cfg.width}x{cfg.height} spp {cfg.spp} depth {cfg.max_depth} sem {cfg.semantics} mode {cfg.sample_mode} "
                  f"flags {cfg.flags} rng {cfg.rng} shapes {len(cfg.shapes)} tris {len(cfg.triangles)}: " + "; ".join(problems), flush=True)
        if (seed - first + 1) % 25 == 0:
            print(f"... {seed - first + 1} scenes, {failed} failed, {time.time() - t0:.0f} s", flush=True)
    print(("one-shot renders of a large mesh: " if large else "every way of calling the renderer: " if calls else "") + f"{tot['scenes']} random scenes (seeds {first} .. {first + n - 1}; {tot['sem0']} under the CPU tracer's rules, {tot['with_tris']} with "
          f"triangles, {tot['with_boxes']} with boxes), {tot['samples']} samples, {tot['segments']} segments: {failed} differ from the oracle")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
