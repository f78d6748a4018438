"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same scenes and seeds.

Bar: bit-exact.  The kernels use only IEEE-754 binary32 +,-,*,/,sqrt and explicit fma in a fixed order,
so hit indices, material ids, work counters AND every accumulated pixel word must equal the oracle's.
(Against the reference's own product order — the recursion of rayTracer.h:121 — colours agree to
rounding: rtol 2e-6, checked on the oracle side in test_oracle_kat.py.)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ("samples", "segments", "diffuse", "specular", "terminal",
            "shape_tests", "sphere_tests", "box_tests", "triangle_tests")


def oracle_params(orc, cfg, **kw):
    args = dict(seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode, flags=cfg.flags,
                triangle_albedo=cfg.triangle_albedo, rng=cfg.rng, triangle_colors=cfg.triangle_colors)
    if cfg.flags & 8:                    # RT_FLAG_LITERAL_GEOMETRY: the oracle's arithmetic of that name
        args["arith"] = orc.ARITH_LITERAL_GEOMETRY
    args.update(kw)
    return orc.params(cfg.width, cfg.height, cfg.spp, cfg.max_depth, **args)


def check_config(rta, orc, cfg, trace_samples=(0,), grid_n=0):
    rt = rta.scenes.make_renderer(cfg)
    try:
        stats = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, grid_n=grid_n, stats=True)
        img = rt.resolve(cfg.width, cfg.height, gamma_sqrt=(cfg.semantics == 0))
        p = oracle_params(orc, cfg, grid_n=grid_n)
        acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
        ref = orc.resolve(acc, cfg.spp, cfg.semantics, gamma_sqrt=(cfg.semantics == 0))
        assert {k: stats[k] for k in COUNTERS} == {k: ostats[k] for k in COUNTERS}
        diff = img.view(np.uint32) != ref.view(np.uint32)
        assert not diff.any(), f"{int(diff.sum())} of {diff.size} image words differ from the oracle"
        assert np.nanmax(ref[..., :3]) > 0                  # (the root of a debug normal's negative component is a NaN)
        for s in trace_samples:
            hit, mat = rt.traceDebug(cfg.width, cfg.height, cfg.max_depth, cfg.seed, s, grid_n=grid_n)
            ohit, omat = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, s)
            np.testing.assert_array_equal(hit, ohit)
            np.testing.assert_array_equal(mat, omat)
        # the plain render (no counters) gives the same words
        rt.resetAccum()
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, grid_n=grid_n)
        img2 = rt.resolve(cfg.width, cfg.height, gamma_sqrt=(cfg.semantics == 0))
        np.testing.assert_array_equal(img2.view(np.uint32), img.view(np.uint32))
    finally:
        rt.close()
    return stats


class _HipFrames:
    """`ours` of literal_study.study(): hit indices and accumulator words of the HIP renderer."""

    def __init__(self, rta, cfg, grid_n):
        self.rt, self.cfg, self.grid_n = rta.scenes.make_renderer(cfg), cfg, grid_n

    def trace(self, sample):
        c = self.cfg
        return self.rt.traceDebug(c.width, c.height, c.max_depth, c.seed, sample, grid_n=self.grid_n)[0]

    def render(self):
        import torch
        c = self.cfg
        acc = torch.zeros(c.height, c.width, 4, dtype=torch.int32, device="cuda")
        self.rt.bindAccum(acc.data_ptr(), c.width, c.height)
        self.rt.render(c.width, c.height, c.spp, c.max_depth, c.seed, grid_n=self.grid_n)
        torch.cuda.synchronize()
        self.rt.bindAccum(None, 0, 0)
        return acc.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5", "c5e", "cpu_tracer"])
def test_hip_frame_within_the_literal_bound(rta, orc, name):
    """The HIP frame and its followed hit indices against the oracle in the reference's LITERAL arithmetic (no FMA,
    IEEE divide, v / sqrt(v.v), libm sine and cosine, Sphere::hit's distance in double) — not against the recipes the
    kernels share with the oracle: first hits differing, segments hitting the same primitive, PSNR and max-abs of the
    resolved frame must be no worse than the frozen floor of tests/golden/literal_bounds_frozen.json, and equal to
    the committed report of the recipe arithmetic (the kernels repeat it bit for bit)."""
    import json
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from literal_study import cases, no_worse_than, study
    cfg, grid_n = cases(rta.scenes)[name]
    ours = _HipFrames(rta, cfg, grid_n)
    try:
        r = study(orc, cfg, grid_n, ours=ours)
    finally:
        ours.rt.close()
    worse = no_worse_than(r, json.load(open(os.path.join(here, "golden", "literal_bounds_frozen.json")))["configs"][name])
    assert not worse, f"{name}: {worse}"
    want = json.load(open(os.path.join(here, "golden", "literal_vs_recipe.json")))["configs"][name]
    for k in ("first_hits_differing", "segment_hit_agreement_percent", "image_psnr_db", "image_max_abs"):
        assert r[k] == want[k], (k, r[k], want[k])


@pytest.mark.parametrize("which,radius", [("far", None), ("near", None), ("near", -0.5)])
def test_sphere_normals_far_from_the_origin_and_with_a_negative_radius(rta, orc, which, radius):
    """The debug material shows the normal: a sphere whose hit points are off it by 1e-3 of the radius (the kernel's
    one-step normal falls back on the full normalisation there) and a sphere with a negative radius (the stored
    first guess is 1/|radius|)."""
    cfg = rta.scenes.config_sphere_normals(which, radius)
    check_config(rta, orc, cfg, grid_n=1)
    rt = rta.scenes.make_renderer(cfg)
    rt.render(cfg.width, cfg.height, 1, cfg.max_depth, cfg.seed, grid_n=1)
    n = rt.resolve(cfg.width, cfg.height)[..., :3].astype(np.float64)
    rt.close()
    length = np.sqrt((n ** 2).sum(-1))
    assert (length > 0).sum() > 1000 and np.abs(length[length > 0] - 1.0).max() < 4e-7


@pytest.mark.parametrize("name", ["c1", "c2", "c4", "c5", "cpu_tracer"])
def test_literal_geometry_mode(rta, orc, name):
    """RT_FLAG_LITERAL_GEOMETRY: Sphere::hit, normalAt, the Ray constructor and reflect as scene.h:14-23,115-132 and
    utils.h:62-64 are written (no fused operation, v / sqrt(v.v), the hit distance in double).  Bit for bit against
    the oracle's arithmetic of the same name — counters, image words, followed hit and material ids — and the FIRST
    hits of every pixel equal those of the fully literal arithmetic (which differs only in its samplers' sine and
    cosine, i.e. from the first scatter on): 0 differing, where the default recipes differ on silhouettes."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from literal_study import cases
    cfg, grid_n = cases(rta.scenes)[name]
    cfg.flags |= rta.FLAG_LITERAL_GEOMETRY
    check_config(rta, orc, cfg, trace_samples=(0, 1, 2, 3), grid_n=grid_n)
    rt = rta.scenes.make_renderer(cfg)
    p_lit = oracle_params(orc, cfg, grid_n=grid_n, arith=orc.ARITH_LITERAL)
    differing = later = later_same = 0
    for s in range(4):
        hit, mat = rt.traceDebug(cfg.width, cfg.height, cfg.max_depth, cfg.seed, s, grid_n=grid_n)
        lhit, lmat = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p_lit, s)
        differing += int((hit[..., 0] != lhit[..., 0]).sum()) + int((mat[..., 0] != lmat[..., 0]).sum())
        live = (hit != -2) | (lhit != -2)
        later += int(live.sum()); later_same += int(((hit == lhit) & live).sum())
    rt.close()
    assert differing == 0
    assert later_same >= 0.995 * later          # later segments: only the samplers' last bits separate the two


def test_literal_geometry_flag_is_refused_where_it_does_not_apply(rta):
    cfg = rta.scenes.config3(width=32, height=32, spp=1, max_depth=2)
    cfg.semantics = 0
    cfg.flags |= rta.FLAG_LITERAL_GEOMETRY
    with pytest.raises(rta.RtError):
        rta.scenes.make_renderer(cfg)           # boxes and triangles
    cfg = rta.scenes.config1(width=32, height=32)
    cfg.flags |= rta.FLAG_LITERAL_GEOMETRY
    cfg.semantics = 1
    with pytest.raises(rta.RtError):
        rta.scenes.make_renderer(cfg)           # the shader's rules


def test_config1_cpu_sphere_scene(rta, orc):
    """The CPU-runnable configuration at full size: 512x512, 16 spp, depth 4."""
    st = check_config(rta, orc, rta.scenes.config1(), trace_samples=(0, 7, 15))
    assert st["samples"] == 512 * 512 * 16


def test_config2_scene_reduced(rta, orc):
    """Config 2's scene and depth at 320x180 (the oracle finishes in seconds), ragged tile edges."""
    check_config(rta, orc, rta.scenes.config2(width=323, height=181, spp=25, max_depth=8), trace_samples=(0, 24))


def test_config3_mesh_scene(rta, orc):
    """main.cpp's shapes + the 292-triangle bunny, shader semantics (boxes, mirror, quantised sums)."""
    check_config(rta, orc, rta.scenes.config3(width=160, height=90, spp=8, max_depth=10), trace_samples=(0, 5))


def test_config3_big_bunny(rta, orc):
    """The 4 968-triangle COFF bunny."""
    check_config(rta, orc, rta.scenes.config3(width=64, height=36, spp=2, max_depth=6, mesh="bunny.off"))


def test_config4_multi_light_metal(rta, orc):
    """Fuzzy metal (scene.h:73-81) + several emitters under the CPU tracer's rules."""
    check_config(rta, orc, rta.scenes.config4(width=192, height=108, spp=16, max_depth=11), trace_samples=(0, 9))


def test_config5_divergence_scene(rta, orc):
    check_config(rta, orc, rta.scenes.config5(width=128, height=128, spp=8, max_depth=16), trace_samples=(3,))


def test_config5_enclosed_room(rta, orc):
    """Config 5's spheres inside five wall boxes: paths of ten and more segments, CPU rules with boxes, metal walls."""
    st = check_config(rta, orc, rta.scenes.config5_enclosed(width=96, height=96, spp=6, max_depth=16), trace_samples=(2,))
    assert st["segments"] >= 5 * st["samples"]


def test_cpu_tracer_scene_as_shipped(rta, orc):
    """rayTracer.h:48-61 in full — spheres, fuzzy metal and the rotated Cube (scene.h:142-217) — on the 5 x 5 grid
    of renderPixel<5> as written, 11 segments."""
    st = check_config(rta, orc, rta.scenes.config_cpu_tracer(width=160, height=160), trace_samples=(0, 12, 24), grid_n=5)
    hit, _ = None, None
    cfg = rta.scenes.config_cpu_tracer(width=160, height=160)
    rt = rta.scenes.make_renderer(cfg)
    hit, mat = rt.traceDebug(160, 160, 11, cfg.seed, 0, grid_n=5)
    rt.close()
    assert (hit == 5).sum() > 50          # the cube (shape 5) is hit by some rays; its exact surface test passes only sometimes


@pytest.mark.parametrize("name,kw", [("c3", dict(width=160, height=90, spp=8, max_depth=10)),
                                     ("c4", dict(width=128, height=72, spp=8, max_depth=11)),
                                     ("cpu_tracer", dict(width=96, height=96, spp=16, max_depth=11))])
def test_shader_random_stream(rta, orc, name, kw):
    """RT_RNG_GLSL: the draws come from random() of RayTracing.glsl:44-51 with the seed of :9, in program order."""
    cfg = rta.scenes.CONFIGS[name](**kw)
    cfg.rng = rta.RNG_GLSL
    cfg.seed = 4321
    check_config(rta, orc, cfg, trace_samples=(0, 7), grid_n=4 if name == "cpu_tracer" else 0)
    # ... and it is a different stream from the counter hash
    rt = rta.scenes.make_renderer(cfg); rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    a = rt.resolve(cfg.width, cfg.height); rt.close()
    cfg.rng = rta.RNG_HASH
    rt = rta.scenes.make_renderer(cfg); rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    b = rt.resolve(cfg.width, cfg.height); rt.close()
    assert (a != b).any() and abs(a[..., :3].mean() - b[..., :3].mean()) < 0.25 * b[..., :3].mean() + 1e-3


@pytest.mark.parametrize("name", ["c4", "cpu_tracer"])
def test_unit_vector_from_normalised_gaussians(rta, orc, name):
    """RT_FLAG_UNIT_GAUSS: randomUnitVector() as src/utils.h:45 writes it — three normal deviates, normalised — for the
    Lambertian and the fuzzy metal, four draws per event; bit for bit against the oracle's recipe of it."""
    from ray_tracing_amd import _abi
    if name == "c4":
        cfg, grid_n = rta.scenes.config4(width=160, height=90, spp=12, max_depth=11), 0
    else:
        cfg, grid_n = rta.scenes.config_cpu_tracer(width=96, height=96), 5
    cfg.flags |= _abi.FLAG_UNIT_GAUSS
    st = check_config(rta, orc, cfg, trace_samples=(0, 3), grid_n=grid_n)
    assert st["diffuse"] > 0 and st["specular"] > 0
    # the flag is refused where the shader's rules or its generator are asked for
    cam = rta.Camera(8, 8)
    for kw in (dict(semantics=rta.SEM_GLSL), dict(semantics=rta.SEM_CPU, rng=1)):
        with pytest.raises(rta.RtError):
            rta.RayTracingComputeShader(cam, flags=_abi.FLAG_UNIT_GAUSS, **kw)


def test_metal_unnormalized_flag(rta, orc):
    cfg = rta.scenes.config4(width=96, height=54, spp=8, max_depth=11)
    cfg.flags = rta.FLAG_METAL_UNNORMALIZED
    check_config(rta, orc, cfg)


@pytest.mark.parametrize("mode", ["grid", "grid_compat"])
def test_grid_sample_modes(rta, orc, mode):
    """renderPixel<N> of rayTracer.h:89-108 (N = 5 as in rayTracer.h:78)."""
    cfg = rta.scenes.config1(width=96, height=64, spp=25, max_depth=11)
    cfg.sample_mode = rta.SAMPLE_GRID if mode == "grid" else rta.SAMPLE_GRID_COMPAT
    check_config(rta, orc, cfg, trace_samples=(0, 13, 24), grid_n=5)


def test_cpu_semantics_with_triangles_and_boxes(rta, orc):
    """CPU-tracer shading rules over the shader's box and triangle geometry (mixed scene)."""
    cfg = rta.scenes.config3(width=96, height=54, spp=4, max_depth=8)
    cfg.semantics = rta.SEM_CPU
    cfg.triangle_albedo = (0.9, 0.4, 0.1)
    check_config(rta, orc, cfg)


def test_glsl_semantics_on_sphere_scene_with_debug_material(rta, orc):
    cfg = rta.scenes.config1(width=80, height=60, spp=4, max_depth=10)
    cfg.semantics = rta.SEM_GLSL
    cfg.shapes = cfg.shapes.copy()
    cfg.shapes["material"]["type"][1] = 100           # DebugMaterial: normal as colour (shape.h:37-42)
    cfg.shapes["material"]["type"][0] = 55            # unknown type: returns the throughput (glsl:253)
    check_config(rta, orc, cfg, trace_samples=(1,))


def test_empty_scene_and_single_pixel(rta, orc):
    cfg = rta.scenes.config1(width=1, height=1, spp=3, max_depth=2)
    rt = rta.scenes.make_renderer(cfg)
    rt.render(1, 1, 3, 2, 1)
    acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
    np.testing.assert_array_equal(rt.resolve(1, 1).view(np.uint32), orc.resolve(acc, 3, 0).view(np.uint32))
    rt.shapes = np.zeros(0, rta.SHAPE_DTYPE)
    rt.updateShapeBuffer()
    rt.resetAccum()
    st = rt.render(1, 1, 3, 2, 1, stats=True)
    assert st["segments"] == 3 and not rt.resolve(1, 1)[..., :3].any()
    rt.close()


def test_progressive_chunks_equal_one_shot(rta, orc):
    """Sums continue across render calls in sample order: 4+4+8 samples == 16 at once, bit for bit."""
    cfg = rta.scenes.config1(width=128, height=72, spp=16, max_depth=6)
    rt = rta.scenes.make_renderer(cfg)
    rt.render(128, 72, 16, 6, 5)
    one = rt.resolve(128, 72)
    rt.resetAccum()
    for first, n in ((0, 4), (4, 4), (8, 8)):
        rt.render(128, 72, n, 6, 5, first_sample=first)
    assert rt.accumSamples() == 16
    np.testing.assert_array_equal(rt.resolve(128, 72).view(np.uint32), one.view(np.uint32))
    rt.close()


def test_row_shards_reassemble_bit_exact(rta, orc):
    """What each GPU of a node renders (rows with (y//16) % G == g): the shards sum to the whole frame."""
    import torch
    cfg = rta.scenes.config2(width=200, height=150, spp=6, max_depth=8)
    rt = rta.scenes.make_renderer(cfg)
    rt.render(200, 150, 6, 8, 1)
    whole = rt.resolve(200, 150)
    total = torch.zeros(150, 200, 4, dtype=torch.float32, device="cuda")
    for g in range(3):
        part = torch.zeros(150, 200, 4, dtype=torch.int32, device="cuda")
        rt.bindAccum(part.data_ptr(), 200, 150)
        rt.setRowShard(g, 3, 16)
        rt.render(200, 150, 6, 8, 1)
        torch.cuda.synchronize()
        rows = torch.tensor([(y // 16) % 3 == g for y in range(150)], device="cuda")
        assert not part[~rows].any()
        total += part.view(torch.float32)
    rt.bindAccum(None, 0, 0)
    rt.close()
    ref = (total[..., :3].cpu().numpy() / np.float32(6))[::-1]      # IEEE division on the host
    np.testing.assert_array_equal(ref.view(np.uint32), whole[..., :3].copy().view(np.uint32))


@pytest.mark.parametrize("name,kw", [("c2", dict(width=323, height=181, spp=25, max_depth=8)),
                                     ("c3", dict(width=211, height=123, spp=80, max_depth=6)),      # probe, plan, slices
                                     ("c3", dict(width=160, height=90, spp=8, max_depth=10)),
                                     ("c5e", dict(width=96, height=96, spp=6, max_depth=16))])
def test_discard_and_render_resolved_equal_reset_render_resolve(rta, name, kw):
    """rt_discard_accum + rt_render_resolved leave the words of rt_reset_accum + rt_render + rt_resolve_device, in the
    accumulator and in the resolved frame — frame after frame on one context (first without a plan, then with one,
    with and without sliced tiles, while the plan is kept and when it is renewed)."""
    import torch
    cfg = rta.scenes.CONFIGS[name](**kw)
    gamma = cfg.semantics == 0
    ref_rt = rta.scenes.make_renderer(cfg)
    ref_rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    ref_img = ref_rt.resolve(cfg.width, cfg.height, gamma_sqrt=gamma)
    ref_rt.close()
    rt = rta.scenes.make_renderer(cfg)
    acc = torch.full((cfg.height, cfg.width, 4), 12345, dtype=torch.int32, device="cuda")       # garbage: must not be read
    out = torch.empty(cfg.height, cfg.width, 4, dtype=torch.float32, device="cuda")
    rt.bindAccum(acc.data_ptr(), cfg.width, cfg.height)
    first = None
    for frame in range(36):                                   # beyond the 32 calls a plan is kept for
        if frame not in (0, 1, 2, 3, 31, 32, 33, 35):
            rt.discardAccum(); rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
            continue
        out.fill_(-1.0)
        rt.discardAccum()
        rt.renderResolved(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, out.data_ptr(), gamma_sqrt=gamma)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref_img.view(np.uint32), err_msg=f"frame {frame}")
        words = acc.cpu().numpy()
        if first is None:
            first = words
        np.testing.assert_array_equal(words[..., :3], first[..., :3], err_msg=f"frame {frame}")
        assert rt.accumSamples() == cfg.spp
    # adding to a frame: the resolve counts all samples
    rt.renderResolved(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed + 1, out.data_ptr(), gamma_sqrt=gamma)
    torch.cuda.synchronize()
    two = out.cpu().numpy()
    rt.bindAccum(None, 0, 0)
    rt.close()
    rt = rta.scenes.make_renderer(cfg)
    rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
    rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed + 1)
    np.testing.assert_array_equal(two.view(np.uint32), rt.resolve(cfg.width, cfg.height, gamma_sqrt=gamma).view(np.uint32))
    rt.close()


def test_converter_running_mean(rta, orc):
    """main.cpp:135-151 for three frames against the oracle's PixelConversion restatement."""
    cfg = rta.scenes.config3(width=96, height=54, spp=10, max_depth=10)
    rt = rta.scenes.make_renderer(cfg)
    image = np.zeros((54, 96, 4), np.float32)
    for n in (1, 2, 3):
        rt.updateCameraBuffer()                       # main.cpp:143
        rt.compute(96, 54, 1)                         # main.cpp:145
        rt.convert(n)                                 # main.cpp:147-149
        p = oracle_params(orc, cfg, first_sample=(n - 1) * 10)
        acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
        orc.convert(acc, image, 10, cfg.semantics, n)
    np.testing.assert_array_equal(rt.readImage().view(np.uint32), image.view(np.uint32))
    rt.close()


def test_first_frame_of_a_fresh_context_equals_the_oracle(rta, orc):
    """A one-shot render (no earlier frame to learn a plan from) of 64+ samples per pixel splits itself into a
    probe of its first 8 samples, a plan and the rest; a short one runs in the built-in order.  Same words either way."""
    for sem in (0, 1):
        cfg = rta.scenes.config3(width=211, height=123, spp=80, max_depth=6)
        cfg.semantics = sem
        acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
        ref = orc.resolve(acc, cfg.spp, cfg.semantics)
        n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
        rt = rta.scenes.make_renderer(cfg)
        st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
        np.testing.assert_array_equal(rt.resolve(cfg.width, cfg.height).view(np.uint32), ref.view(np.uint32))
        assert {k: st[k] for k in COUNTERS} == {k: ostats[k] for k in COUNTERS}
        assert st["work_units"] >= 2 * n_tiles
        rt.close()
        rt = rta.scenes.make_renderer(cfg)                 # 40 + 40: two short calls, no probe
        rt.render(cfg.width, cfg.height, 40, cfg.max_depth, cfg.seed)
        st = rt.render(cfg.width, cfg.height, 40, cfg.max_depth, cfg.seed, first_sample=40, stats=True)
        np.testing.assert_array_equal(rt.resolve(cfg.width, cfg.height).view(np.uint32), ref.view(np.uint32))
        rt.close()


def test_planned_probe_of_a_large_mesh(rta, orc, monkeypatch):
    """A fresh context's one-shot render of a scene with a mesh of more than 800 triangles under the shader's rules plans
    its probe from the camera rays' candidate lists (tiles with many candidate leaves are cut into slices of a sample
    each): the same words as the oracle's, as with that plan switched off, and as with a garbage accumulator that is
    discarded first (the slices' atomics then need the clear an overwriting launch would not)."""
    import torch
    cfg = rta.scenes.config3(width=112, height=64, spp=64, max_depth=4, mesh="bunny.off")
    assert len(cfg.triangles) > 800
    acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
    ref = orc.resolve(acc, cfg.spp, cfg.semantics)
    n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
    rt = rta.scenes.make_renderer(cfg)
    st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
    np.testing.assert_array_equal(rt.resolve(cfg.width, cfg.height).view(np.uint32), ref.view(np.uint32))
    assert {k: st[k] for k in COUNTERS} == {k: ostats[k] for k in COUNTERS}
    assert st["work_units"] > 2 * n_tiles                 # the probe's own slices on top of probe + rest
    rt.close()
    monkeypatch.setenv("RT_PROBE_PRIOR", "0")
    rt = rta.scenes.make_renderer(cfg)
    st0 = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
    np.testing.assert_array_equal(rt.resolve(cfg.width, cfg.height).view(np.uint32), ref.view(np.uint32))
    assert st0["work_units"] < st["work_units"]
    rt.close()
    monkeypatch.delenv("RT_PROBE_PRIOR")
    rt = rta.scenes.make_renderer(cfg)
    buf = torch.full((cfg.height, cfg.width, 4), 987654321, dtype=torch.int32, device="cuda")
    out = torch.empty(cfg.height, cfg.width, 4, dtype=torch.float32, device="cuda")
    rt.bindAccum(buf.data_ptr(), cfg.width, cfg.height)
    rt.discardAccum()
    rt.renderResolved(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, out.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    rt.bindAccum(None, 0, 0)
    rt.close()


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5", "c5e"])
def test_full_size_properties(rta, name, monkeypatch):
    """Configurations 2-5 (and 5 enclosed in its room of wall boxes) at their full sizes: properties that need no oracle run (counters, idempotence, linearity in
    the sample split, seed independence of the mean; for the mesh scene also: the frame does not depend on the launch
    plan — tile order, slices of the sample range — or on where the mesh lives)."""
    cfg = rta.scenes.CONFIGS[name]()
    w, h, spp = cfg.width, cfg.height, cfg.spp
    rt = rta.scenes.make_renderer(cfg)
    st = rt.render(w, h, spp, cfg.max_depth, cfg.seed, stats=True)
    img = rt.resolve(w, h)
    assert st["samples"] == w * h * spp and st["terminal"] == st["samples"]
    assert st["samples"] <= st["segments"] <= cfg.max_depth * st["samples"]
    assert st["segments"] - st["terminal"] <= st["diffuse"] + st["specular"] <= st["segments"]     # every segment that went on scattered
    assert st["shape_tests"] == st["segments"] * len(cfg.shapes) and st["triangle_tests"] == st["segments"] * len(cfg.triangles)
    assert st["triangle_tests_done"] <= st["triangle_tests"] and st["box_tests_done"] <= st["box_tests"]
    assert np.isfinite(img).all() and img[..., :3].min() >= 0 and img[..., :3].max() <= 10.0
    words = img.view(np.uint32)
    # idempotence: the same call again (now by the plan made from the first frame) gives the same words
    rt.resetAccum()
    rt.render(w, h, spp, cfg.max_depth, cfg.seed)
    np.testing.assert_array_equal(rt.resolve(w, h).view(np.uint32), words)
    # linearity in the sample split: 60 % + 40 % of the samples == all of them
    first = spp * 6 // 10
    rt.resetAccum()
    rt.render(w, h, first, cfg.max_depth, cfg.seed, first_sample=0)
    rt.render(w, h, spp - first, cfg.max_depth, cfg.seed, first_sample=first)
    np.testing.assert_array_equal(rt.resolve(w, h).view(np.uint32), words)
    # a different seed gives a different image with the same mean to Monte-Carlo accuracy
    rt.resetAccum()
    rt.render(w, h, spp, cfg.max_depth, cfg.seed + 1)
    img2 = rt.resolve(w, h)
    assert (img2 != img).any() and abs(img2[..., :3].mean() - img[..., :3].mean()) < 0.01 * img[..., :3].mean()
    rt.close()
    if name == "c3":
        # fresh contexts: no plan at all (built-in tile order, no slices), and the mesh staged in LDS
        for env in ({"RT_NO_TILE_ORDER": "1"}, {"RT_MESH_IN_LDS": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            rt = rta.scenes.make_renderer(cfg)
            rt.render(w, h, spp, cfg.max_depth, cfg.seed)
            np.testing.assert_array_equal(rt.resolve(w, h).view(np.uint32), words)
            rt.close()
            for k in env:
                monkeypatch.delenv(k)


def test_argument_errors(rta):
    cfg = rta.scenes.config1(width=32, height=32, spp=1, max_depth=2)
    rt = rta.scenes.make_renderer(cfg)
    with pytest.raises(rta.RtError):
        rt.render(0, 32, 1, 2, 1)
    with pytest.raises(rta.RtError):
        rt.render(32, 32, 1, 0, 1)
    with pytest.raises(rta.RtError):
        rt.resolve(32, 32)                 # nothing rendered yet
    with pytest.raises(rta.RtError):
        rt.setRowShard(3, 3)
    rt.close()
    cam = rta.Camera(8, 8)
    rt2 = rta.RayTracingComputeShader(cam)
    with pytest.raises(rta.RtError):
        rt2.render(8, 8, 1, 2, 1)          # camera never set
    rt2.close()


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0            # little-endian
        data = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return data[::-1]                             # PFM stores rows bottom to top


def test_headless_cpp_driver_matches_python_host(rta, tmp_path):
    """examples/headless_main (rt_host.hpp, the frame loop of main.cpp:135-151) == the same frames through host.py."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "headless_main")
    assert os.path.exists(exe), "examples/headless_main is not built (run __graft_entry__.build())"
    out = str(tmp_path / "frame.pfm")
    mesh = os.path.join(rta.scenes.MESH_DIR, "Bunny-LowPoly.off")
    subprocess.run([exe, "--mesh", mesh, "--frames", "2", "--width", "96", "--height", "54", "--out", out],
                   check=True, timeout=120)
    got = read_pfm(out)
    cfg = rta.scenes.config3(width=96, height=54, spp=100, max_depth=10)
    rt = rta.scenes.make_renderer(cfg)
    for n in (1, 2):
        rt.updateCameraBuffer(); rt.compute(96, 54, 1); rt.convert(n)
    want = rt.readImage()[::-1, :, :3]           # the library's image has the bottom of the view in row 0; a PFM reader returns the top first
    rt.close()
    np.testing.assert_array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    # orientation as a PFM reader sees it: the lit floor at the bottom of the picture, the black sky at the top
    assert got.shape == (54, 96, 3) and got[-8:].mean() > 0.05 and got[:8].mean() < 0.01


@pytest.mark.parametrize("sem", [0, 1])
def test_triangle_colours(rta, orc, sem):
    """One albedo per triangle (vertex colours of a COFF mesh) in place of RayTracing.glsl:275's red, both rule sets."""
    cfg = rta.scenes.config_colored_mesh(width=203, height=131, spp=8, semantics=sem)
    check_config(rta, orc, cfg, trace_samples=(0, 5))
    # uploading triangles again returns to the single colour
    rt = rta.scenes.make_renderer(cfg)
    try:
        rt.triangle_colors = None
        rt.updateShapeBuffer()
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        img = rt.resolve(cfg.width, cfg.height)
        acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg, triangle_colors=None))
        np.testing.assert_array_equal(img.view(np.uint32), orc.resolve(acc, cfg.spp, cfg.semantics).view(np.uint32))
        with pytest.raises(rta.RtError):
            rt.triangle_colors = np.zeros((3, 3), np.float32)        # wrong count
            rt.updateShapeBuffer()
    finally:
        rt.close()


@pytest.mark.parametrize("sem", [0, 1])
def test_box_with_its_own_sides(rta, orc, sem):
    """RT_FLAG_BOX_SIDES: a Box has the three sides shape.h:67 stores; without the flag only the first (glsl:139)."""
    cfg = rta.scenes.config_box_sides(width=203, height=131, spp=8, semantics=sem)
    check_config(rta, orc, cfg, trace_samples=(0, 5))
    cfg.flags = 0
    check_config(rta, orc, cfg, trace_samples=(0,))


def test_sliced_sample_ranges(rta, orc):
    """Under the shader's rules a tile that cost much more than the others in the previous frame is traced by
    several workgroups, each a slice of the sample range, added with integer atomics: same words, same counters."""
    cfg = rta.scenes.config3(width=323, height=181, spp=96, max_depth=6)
    rt = rta.scenes.make_renderer(cfg)
    try:
        acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
        ref = orc.resolve(acc, cfg.spp, cfg.semantics)
        n_tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
        units = []
        for frame in range(3):
            rt.resetAccum()
            st = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
            img = rt.resolve(cfg.width, cfg.height)
            np.testing.assert_array_equal(img.view(np.uint32), ref.view(np.uint32))
            assert {k: st[k] for k in COUNTERS} == {k: ostats[k] for k in COUNTERS}
            units.append(st["work_units"])
        # the first frame has no plan: it probes with its first 8 samples (one workgroup per tile), then runs the
        # rest by the plan made from them; later frames run by the plan of the frame before
        assert units[0] > 2 * n_tiles and units[1] > n_tiles and units[2] > n_tiles, units
        # two half frames on top of each other (the second one sliced too) = one frame
        rt.resetAccum()
        rt.render(cfg.width, cfg.height, 48, cfg.max_depth, cfg.seed, first_sample=0)
        rt.render(cfg.width, cfg.height, 48, cfg.max_depth, cfg.seed, first_sample=48)
        np.testing.assert_array_equal(rt.resolve(cfg.width, cfg.height).view(np.uint32), ref.view(np.uint32))
        # the CPU tracer's float sums are never sliced: their order of addition is part of the result
    finally:
        rt.close()
    cfg0 = rta.scenes.config3(width=323, height=181, spp=96, max_depth=6)
    cfg0.semantics = 0
    rt = rta.scenes.make_renderer(cfg0)
    try:
        for frame in range(2):
            rt.resetAccum()
            st = rt.render(cfg0.width, cfg0.height, cfg0.spp, cfg0.max_depth, cfg0.seed, stats=True)
            assert st["work_units"] == (2 * n_tiles if frame == 0 else n_tiles)      # probe + rest, then one launch
    finally:
        rt.close()


def _axis_ray_scene(rta, sphere_z, semantics):
    """A 1x1 frame whose one grid sample leaves (0, 0, -7) along exactly (0, 0, -1) (the normalisation of
    (0, 0, -7) is exact), and one emissive unit sphere centred on the ray at z = sphere_z."""
    from ray_tracing_amd import host, _abi
    cam = np.zeros(21, dtype=np.float32)
    cam[4:7] = (0, 0, -1); cam[8:11] = (0, 0, -7); cam[12:15] = (0, 1, 0); cam[16:19] = (1, 0, 0); cam[19] = 1.0 / 64
    shapes = host.pack([host.Circle((0, 0, sphere_z), 1.0, host.Light((1, 1, 1)))], _abi.SHAPE_DTYPE)
    return rta.scenes.SceneConfig("far_edge", shapes, np.zeros(0, dtype=_abi.TRIANGLE_DTYPE), cam, 1, 1, 1, 2, semantics,
                                  sample_mode=_abi.SAMPLE_GRID)


def test_far_edge_of_the_window_is_exclusive(rta, orc):
    """RayTracing.glsl:212 accepts distance < 1000 only.  With the sphere at z = -1008 the hit distance is exactly
    1000.0f (oc.d = -1001, disc = 1): a miss; one float nearer it is a hit."""
    far = _axis_ray_scene(rta, -1008.0, 1)
    near = _axis_ray_scene(rta, float(np.nextafter(np.float32(-1008.0), np.float32(0))), 1)
    assert orc.hit_sphere((0.0, 0.0, -1008.0), 1.0, (0.0, 0.0, -7.0), (0.0, 0.0, -1.0)) == 1000.0
    for cfg, want in ((far, -1), (near, 0)):
        rt = rta.scenes.make_renderer(cfg)
        try:
            hit, _ = rt.traceDebug(1, 1, 2, cfg.seed, 0, grid_n=1)
            ohit, _ = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg, grid_n=1), 0)
            assert int(ohit.ravel()[0]) == want
            np.testing.assert_array_equal(hit, ohit)
        finally:
            rt.close()


@pytest.mark.parametrize("n_tris", [1, 5, 101, 290, 800, 801, 1303])
def test_triangle_count_not_a_multiple_of_the_leaf(rta, orc, n_tris):
    """Both meshes of the reference happen to hold a multiple of four triangles (292, 4 968); the tree's
    leaves hold two triangles for a mesh of up to 800 and four beyond, so other counts exercise its
    padding, on both sides of that border."""
    cfg = rta.scenes.config3(width=96, height=54, spp=4, max_depth=6, mesh="Bunny-LowPoly.off" if n_tris <= 292 else "bunny.off")
    cfg.triangles = cfg.triangles[:n_tris].copy()
    check_config(rta, orc, cfg)


def test_ties_keep_the_first_in_upload_order(rta, orc):
    """Coincident primitives: the closest-hit scans replace the best only by a strictly smaller distance
    (scene.h:237, RayTracing.glsl:212,226), so of equal distances the first uploaded wins — here every sphere,
    box and triangle of the mesh scene exists twice, the copy in another colour, and the triangles (which the
    library reorders for its tree) also in reversed order."""
    cfg = rta.scenes.config3(width=96, height=54, spp=4, max_depth=6)
    twin = cfg.shapes.copy()
    twin["material"]["color"] = 1.0 - twin["material"]["color"]
    cfg.shapes = np.concatenate([cfg.shapes, twin])
    cfg.triangles = np.concatenate([cfg.triangles, cfg.triangles[::-1]])
    cfg.triangle_colors = np.concatenate([np.tile(np.float32([[0.9, 0.2, 0.1]]), (292, 1)), np.tile(np.float32([[0.1, 0.2, 0.9]]), (292, 1))])
    st = check_config(rta, orc, cfg)
    hit, _ = (lambda rt: (rt.traceDebug(cfg.width, cfg.height, cfg.max_depth, cfg.seed, 0), rt.close())[0])(rta.scenes.make_renderer(cfg))
    first = hit[..., 0]
    assert (first < 9).sum() > 0 and ((first >= 18) & (first < 18 + 292)).sum() > 0      # shapes and triangles were hit ...
    assert ((first >= 9) & (first < 18)).sum() == 0 and (first >= 18 + 292).sum() == 0   # ... and never a copy
    assert st["segments"] > 0


def test_degenerate_triangles_do_not_break_the_tree(rta, orc):
    """Zero-area triangles (their plane and dual basis are NaN or infinite, shape.h:93-101) can never be hit, and the
    bounds over them must not hide their neighbours."""
    from ray_tracing_amd import host
    cfg = rta.scenes.config3(width=96, height=54, spp=4, max_depth=6)
    bad = host.pack([host.SimpleTriangle((0, 1, -2), (0, 1, -2), (1, 1, -2)), host.SimpleTriangle((1, 1, 1), (2, 2, 2), (3, 3, 3))], cfg.triangles.dtype)
    cfg.triangles = np.concatenate([cfg.triangles[:50], bad[:1], cfg.triangles[50:200], bad[1:], cfg.triangles[200:]])
    check_config(rta, orc, cfg)


def _wavy_sheet(n):
    """2 * n * n triangles: a rippled sheet standing behind the spheres of the mesh scene."""
    from ray_tracing_amd import host, _abi
    u = np.linspace(-3.0, 4.0, n + 1, dtype=np.float32)
    v = np.linspace(0.05, 4.0, n + 1, dtype=np.float32)
    x, y = np.meshgrid(u, v)
    z = (-7.0 + 0.3 * np.sin(3.0 * x) * np.cos(2.0 * y)).astype(np.float32)
    pts = np.stack([x, y, z], axis=-1)
    tris = []
    for j in range(n):
        for i in range(n):
            a, b, c, d = pts[j, i], pts[j, i + 1], pts[j + 1, i], pts[j + 1, i + 1]
            tris.append(host.SimpleTriangle(a, b, c))
            tris.append(host.SimpleTriangle(b, d, c))
    return host.pack(tris, _abi.TRIANGLE_DTYPE)


def test_mesh_larger_than_the_lds_stage(rta, orc):
    """20 000 triangles: neither the mesh (1.3 MB) nor its tree (91 KB) fits the 64 KB a workgroup may stage, so
    both are read from global memory; the result is the exhaustive scan's all the same."""
    cfg = rta.scenes.config3(width=64, height=36, spp=2, max_depth=4)
    cfg.triangles = _wavy_sheet(100)
    assert len(cfg.triangles) == 20000
    st = check_config(rta, orc, cfg)
    assert st["triangle_tests"] == st["segments"] * 20000


def test_mesh_of_two_hundred_thousand_triangles(rta, orc, monkeypatch):
    """199 712 triangles: with a top level of at most 32 nodes the tree has five levels (49 928 leaves, 6 241, 781, 98
    and 13 nodes) and the wave goes down through the one above the walk's four together; with a top level of one node,
    six.  The exhaustive scan's result under both, and the same frame from the tree as shipped (four levels here: 98
    top nodes scanned one after the other)."""
    cfg = rta.scenes.config3(width=48, height=27, spp=2, max_depth=4)
    cfg.triangles = _wavy_sheet(316)
    assert len(cfg.triangles) == 199712
    monkeypatch.setenv("RT_TREE_TOP", "32")
    st = check_config(rta, orc, cfg)
    assert st["triangle_tests"] == st["segments"] * 199712
    frames = []
    for top in ("32", "1", None):
        if top is None:
            monkeypatch.delenv("RT_TREE_TOP")
        else:
            monkeypatch.setenv("RT_TREE_TOP", top)
        rt = rta.scenes.make_renderer(cfg)
        rt.render(cfg.width, cfg.height, 40, cfg.max_depth, cfg.seed)       # 40 samples: with the camera rays' lists
        frames.append(rt.resolve(cfg.width, cfg.height).view(np.uint32).copy())
        rt.close()
    np.testing.assert_array_equal(frames[0], frames[1])
    np.testing.assert_array_equal(frames[0], frames[2])


def test_mesh_scene_without_culling(rta, orc):
    """The as-written fuzzy metal leaves directions unnormalised (scene.h:78), which the bounds cannot assume: with
    that flag every lane takes every leaf and every box, through the same queues."""
    from ray_tracing_amd import host
    cfg = rta.scenes.config3(width=64, height=36, spp=4, max_depth=6)
    cfg.semantics = rta.SEM_CPU
    cfg.flags = rta.FLAG_METAL_UNNORMALIZED
    cfg.shapes = np.concatenate([cfg.shapes, host.pack([host.Circle((0.5, 0.6, -1.0), 0.4, host.Metal((0.9, 0.9, 0.9), fuzz=0.3))], cfg.shapes.dtype)])
    check_config(rta, orc, cfg)


def test_bounds_are_conservative_for_far_grazing_rays(rta, orc):
    """A long lens 300 units away: |origin - centre|^2 is 10^7 times a leaf's R^2, so whether a bounding ball is
    touched is decided by the slack term of the bound test.  Culled and exhaustive walks (flag 1 switches the bounds
    off; under the shader's rules it does nothing else) must give the same hits, and both the oracle's."""
    from ray_tracing_amd import host
    cfg = rta.scenes.config3(width=96, height=64, spp=6, max_depth=6)
    cfg.camera = host.Camera(cfg.width, cfg.height, 60.0, (0.5, 0.9, 298.0)).data.copy()
    frames, hits = [], []
    for flags in (0, rta.FLAG_METAL_UNNORMALIZED):
        cfg.flags = flags
        st = check_config(rta, orc, cfg, trace_samples=(0, 4))
        rt = rta.scenes.make_renderer(cfg)
        rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        frames.append(rt.resolve(cfg.width, cfg.height).view(np.uint32).copy())
        hits.append(rt.traceDebug(cfg.width, cfg.height, cfg.max_depth, cfg.seed, 1)[0])
        rt.close()
        assert st["triangle_tests_done"] > 0
    np.testing.assert_array_equal(frames[0], frames[1])
    np.testing.assert_array_equal(hits[0], hits[1])
    assert (hits[0][..., 0] >= len(cfg.shapes)).mean() > 0.02          # the mesh is in view


def test_many_boxes_fill_more_than_one_pass(rta, orc):
    """Forty boxes in view: the (lane, box) pairs of one ray segment exceed the 64 a pass takes."""
    from ray_tracing_amd import host
    rng = np.random.default_rng(5)
    boxes = [host.Box((float(x), float(y), float(z)), (float(s), float(s), float(s)), host.Lambertian(tuple(rng.uniform(0.2, 1.0, 3))))
             for x, y, z, s in zip(rng.uniform(-3, 5, 40), rng.uniform(0, 4, 40), rng.uniform(-8, -2, 40), rng.uniform(0.4, 1.6, 40))]
    cfg = rta.scenes.config3(width=96, height=54, spp=4, max_depth=6)
    cfg.shapes = np.concatenate([cfg.shapes, host.pack(boxes, cfg.shapes.dtype)])
    cfg.triangles = cfg.triangles[:0]
    st = check_config(rta, orc, cfg)
    assert st["box_tests"] == st["segments"] * 44 and st["box_tests_done"] < st["box_tests"] // 2


def test_small_mesh_staged_in_lds(rta, orc, monkeypatch):
    """The 292-triangle mesh and its tree copied into LDS by every workgroup (RT_MESH_IN_LDS, read at upload) instead of
    read from global memory as by default: only its level-1 nodes are queued there."""
    monkeypatch.setenv("RT_MESH_IN_LDS", "1")
    check_config(rta, orc, rta.scenes.config3(width=96, height=54, spp=4, max_depth=8))


@pytest.mark.parametrize("mesh, staged", [("Bunny-LowPoly.off", False), ("Bunny-LowPoly.off", True), ("bunny.off", False)])
def test_every_leaf_size_gives_the_same_frame(rta, orc, monkeypatch, mesh, staged):
    """The tree's leaves hold two triangles for a small mesh and four for a large one (chosen at upload); RT_LEAF sets
    the size for measurements.  The closest hit is a minimum over the same tests whatever the tree's shape: leaves of
    one, two and four triangles give the same words, counters included, on both meshes, staged in LDS or not, and
    the small mesh's equal the oracle's."""
    cfg = rta.scenes.config3(width=160, height=96, spp=8, max_depth=8, mesh=mesh)
    if staged:
        monkeypatch.setenv("RT_MESH_IN_LDS", "1")
    frames = {}
    for leaf in ("1", "2", "4", None):
        if leaf is None:
            monkeypatch.delenv("RT_LEAF")
        else:
            monkeypatch.setenv("RT_LEAF", leaf)
        rt = rta.scenes.make_renderer(cfg)
        try:
            stats = rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed, stats=True)
            frames[leaf] = (rt.resolve(cfg.width, cfg.height).view(np.uint32).copy(), {k: stats[k] for k in COUNTERS},
                            stats["triangle_tests_done"])
        finally:
            rt.close()
    for leaf in ("1", "2", "4"):
        np.testing.assert_array_equal(frames[leaf][0], frames[None][0])
        assert frames[leaf][1] == frames[None][1]
    assert frames[None][2] == frames["2" if mesh == "Bunny-LowPoly.off" else "4"][2]      # the default is the size chosen by mesh size
    assert frames["1"][2] < frames["2"][2] < frames["4"][2]                               # smaller leaves, fewer triangle tests
    if mesh == "Bunny-LowPoly.off":
        acc, ostats = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
        np.testing.assert_array_equal(frames[None][0], orc.resolve(acc, cfg.spp, cfg.semantics).view(np.uint32))


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes(rta, orc, seed):
    """Random mixtures of spheres, boxes and loose triangles with random materials, under both rule sets: the queued
    tests (pairs from many primitives, partly filled passes, waves with few live lanes) against the exhaustive scan."""
    from ray_tracing_amd import host, _abi
    rng = np.random.default_rng(1000 + seed)
    sem = seed % 2

    def material():
        kind = rng.integers(0, 5)
        col = tuple(float(x) for x in rng.uniform(0.2, 1.0, 3))
        if kind == 0:
            return host.Light(tuple(float(x) for x in rng.uniform(1.0, 4.0, 3)))
        if kind == 1:
            return host.Metal(col, fuzz=float(rng.choice([0.0, 0.2]))) if sem == 0 else host.Metal(col)
        return host.Lambertian(col)

    def point():
        return (float(rng.uniform(-3, 5)), float(rng.uniform(0, 4)), float(rng.uniform(-9, -1)))

    objs = [host.Box((0, -50, 0), (100, 1, 1), host.Lambertian((0.5, 0.7, 0.6))), host.Circle((1, 7, -4), 2.5, host.Light((3, 3, 3)))]
    objs += [host.Circle(point(), float(rng.uniform(0.2, 1.0)), material()) for _ in range(rng.integers(0, 13))]
    objs += [host.Box(point(), (float(rng.uniform(0.3, 1.5)),) * 3, material()) for _ in range(rng.integers(0, 13))]
    tris = []
    for _ in range(int(rng.integers(0, 300)) if seed < 10 else int(rng.integers(1500, 3000))):      # the last two: too many to stage
        a = np.array(point()); e1 = rng.uniform(-0.6, 0.6, 3); e2 = rng.uniform(-0.6, 0.6, 3)
        tris.append(host.SimpleTriangle(tuple(a), tuple(a + e1), tuple(a + e2)))
    w, h = int(rng.integers(20, 100)), int(rng.integers(10, 60))
    cfg = rta.scenes.SceneConfig(f"random{seed}", host.pack(objs, _abi.SHAPE_DTYPE),
                                 host.pack(tris, _abi.TRIANGLE_DTYPE) if tris else np.zeros(0, dtype=_abi.TRIANGLE_DTYPE),
                                 host.Camera(w, h, 1.0, (1, 1, 6)).data.copy(), w, h, 3, 5, sem, seed=seed + 1)
    check_config(rta, orc, cfg, trace_samples=(0, 2))


def test_two_contexts_on_two_streams(rta, orc):
    """Two contexts, each on a stream of its own, with frames queued alternately and nothing waited for in between (how
    bench.py keeps two frames in flight): the launches overlap on the GPU and neither frame may change."""
    import torch
    cfgs = [rta.scenes.config3(width=320, height=200, spp=70, max_depth=8), rta.scenes.config2(width=400, height=240, spp=40, max_depth=8)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    rts, bufs, refs = [], [], []
    for cfg, st in zip(cfgs, streams):
        rt = rta.scenes.make_renderer(cfg)
        rt.setStream(st.cuda_stream)
        buf = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
        rt.bindAccum(buf.data_ptr(), cfg.width, cfg.height)
        rts.append(rt); bufs.append(buf)
        acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, oracle_params(orc, cfg))
        refs.append(np.ascontiguousarray(acc).view(np.uint32)[..., :3])
    torch.cuda.synchronize()
    try:
        for frame in range(4):                         # frame 0 probes and plans, the later ones run by the plan
            for cfg, rt in zip(cfgs, rts):
                rt.resetAccum()
                rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed)
        torch.cuda.synchronize()
        for cfg, buf, ref in zip(cfgs, bufs, refs):
            got = buf.cpu().numpy().view(np.uint32)[..., :3]
            assert np.array_equal(got, ref), f"{cfg.name}: {int((got != ref).sum())} accumulator words differ"
    finally:
        for rt in rts:
            rt.bindAccum(None, 0, 0)
            rt.close()


@pytest.mark.parametrize("sem", [0, 1])
def test_many_hundred_shapes(rta, orc, sem):
    """The shapes are staged in LDS whatever their number: 1 200 spheres and 250 boxes take 140 KiB of a workgroup's
    160; twice as many are refused with a message, not launched."""
    from ray_tracing_amd import host, _abi
    rng = np.random.default_rng(5 + sem)

    def many(n_sph, n_box):
        objs = [host.Circle((1, 60, -4), 20.0, host.Light((3, 3, 3)))]
        for _ in range(n_sph):
            objs.append(host.Circle(tuple(float(x) for x in rng.uniform(-20, 20, 3) + (0, 0, -30)), float(rng.uniform(0.3, 1.5)),
                                    host.Lambertian(tuple(float(x) for x in rng.uniform(0.2, 1, 3))) if rng.random() < 0.7 else host.Metal((0.9, 0.9, 0.9))))
        for _ in range(n_box):
            objs.append(host.Box(tuple(float(x) for x in rng.uniform(-20, 20, 3) + (0, 0, -30)), (float(rng.uniform(0.5, 2.0)),) * 3,
                                 host.Lambertian(tuple(float(x) for x in rng.uniform(0.2, 1, 3)))))
        return host.pack(objs, _abi.SHAPE_DTYPE)

    w, h = 40, 24
    cfg = rta.scenes.SceneConfig("crowd", many(1200, 250), np.zeros(0, dtype=_abi.TRIANGLE_DTYPE),
                                 host.Camera(w, h, 1.0, (1, 1, 6)).data.copy(), w, h, 2, 5, sem, seed=9)
    check_config(rta, orc, cfg)
    cfg.shapes = many(2400, 500)
    with pytest.raises(rta.RtError, match="LDS a workgroup can have"):
        rta.scenes.make_renderer(cfg)


@pytest.mark.parametrize("sem,n_sph", [(0, 129), (1, 500), (0, 960), (1, 961)])
def test_candidate_lists_of_many_spheres(rta, orc, sem, n_sph, monkeypatch):
    """A scene of spheres only has no leaves to list, so the sphere masks of a tile's candidate list run on through the
    leaves' words: up to 960 spheres get primary steps (961 trace without), the same words as the oracle's either way."""
    from ray_tracing_amd import host, _abi
    rng = np.random.default_rng(77 + n_sph)
    objs = [host.Circle((1, 60, -4), 20.0, host.Light((3, 3, 3)))]
    for _ in range(n_sph - 1):
        objs.append(host.Circle(tuple(float(x) for x in rng.uniform(-12, 12, 3) + (0, 0, -20)), float(rng.uniform(0.2, 1.2)),
                                host.Lambertian(tuple(float(x) for x in rng.uniform(0.2, 1, 3))) if rng.random() < 0.7 else host.Metal((0.9, 0.9, 0.9))))
    w, h = 72, 40
    cfg = rta.scenes.SceneConfig("spheres", host.pack(objs, _abi.SHAPE_DTYPE), np.zeros(0, dtype=_abi.TRIANGLE_DTYPE),
                                 host.Camera(w, h, 1.0, (1, 1, 6)).data.copy(), w, h, 3, 5, sem, seed=4)
    monkeypatch.setenv("RT_LISTS_MIN_SPP", "1")
    st = check_config(rta, orc, cfg)
    assert (st["primary_steps"] > 0) == (n_sph <= 960)
    assert st["sphere_tests_done"] < st["sphere_tests"] or n_sph > 960


@pytest.mark.parametrize("seed", [241, 847, 949, 3, 22, 47])
def test_campaign_scenes(rta, orc, seed):
    """Scenes of the randomised campaign (tools/parity_campaign.py: all modes, flags, materials and shape kinds mixed).
    241, 847 and 949 are the three of its first thousand that once differed: fuzzy-metal directions left unnormalised
    (RT_FLAG_METAL_UNNORMALIZED) in a wave with fewer than 24 live lanes, whose queued tree nodes were still culled by
    bounds that assume a unit direction."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("parity_campaign", os.path.join(os.path.dirname(__file__), "..", "tools", "parity_campaign.py"))
    campaign = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(campaign)
    try:
        cfg, stats, problems = campaign.run_one(seed)
    finally:
        os.environ.pop("RT_MESH_IN_LDS", None)
    assert not problems, "; ".join(problems)
    assert stats["samples"] == cfg.width * cfg.height * cfg.spp


def test_rccl_exchanges_on_one_rank(rta):
    """The RCCL branch of both exchanges on hardware: a one-rank `nccl` group (a node is not ours to use), the collective
    forced although one rank has nothing to exchange; integer and float views, the async handle, two accumulators in turn
    as bench.py uses them, and the band gather of the `rows` split."""
    import socket
    import torch
    import torch.distributed as dist
    from ray_tracing_amd import parallel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for sem in (rta.SEM_GLSL, rta.SEM_CPU):
            cfg = rta.scenes.config2(width=320, height=176, spp=4, max_depth=4)
            cfg.semantics = sem
            shard = parallel.plan("samples", 0, 1, cfg.spp)
            rts, accums = [], []
            for _ in range(2):
                rt = rta.scenes.make_renderer(cfg)
                rt.setStream(torch.cuda.current_stream().cuda_stream)
                acc = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
                rt.bindAccum(acc.data_ptr(), cfg.width, cfg.height)
                rts.append(rt); accums.append(acc)
            pending, frames = None, []
            for k in range(4):                                   # frame k's exchange is waited for after frame k+1's trace is queued
                rt = rts[k % 2]
                rt.resetAccum()
                rt.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed + k)
                work = parallel.reduce_framebuffer(accums[k % 2], sem, async_op=True, always=True)
                assert work is not None
                if pending is not None:
                    pending[1].wait()
                    frames.append(accums[pending[0] % 2].cpu().numpy().copy())
                pending = (k, work)
            pending[1].wait()
            frames.append(accums[pending[0] % 2].cpu().numpy().copy())
            # a sum over one rank changes nothing: every frame equals the same render without an exchange
            ref = rta.scenes.make_renderer(cfg)
            racc = torch.zeros(cfg.height, cfg.width, 4, dtype=torch.int32, device="cuda")
            ref.setStream(torch.cuda.current_stream().cuda_stream)
            ref.bindAccum(racc.data_ptr(), cfg.width, cfg.height)
            for k in range(4):
                ref.resetAccum()
                ref.render(cfg.width, cfg.height, cfg.spp, cfg.max_depth, cfg.seed + k)
                torch.cuda.synchronize()
                np.testing.assert_array_equal(frames[k], racc.cpu().numpy())
            ref.bindAccum(None, 0, 0)
            ref.close()
            # the rows split's gather: the one rank owns every band
            rshard = parallel.plan("rows", 0, 1, cfg.spp)
            before = accums[0].clone()
            handle = parallel.gather_bands(accums[0], rshard, async_op=True, always=True)
            handle.wait()
            torch.cuda.synchronize()
            assert torch.equal(accums[0], before)
            for rt in rts:
                rt.bindAccum(None, 0, 0)
                rt.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("split", ["rows", "samples"])
def test_cpp_multi_gpu_driver_on_one_gpu(rta, tmp_path, split):
    """examples/multi_gpu (C++ host, RCCL through its C API: ncclCommInitAll, grouped send / recv of the row bands or one
    ncclReduce) with a one-GPU communicator == the same frame from one context."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "multi_gpu")
    assert os.path.exists(exe), "examples/multi_gpu is not built (run __graft_entry__.build())"
    out = str(tmp_path / "frame.pfm")
    r = subprocess.run([exe, "--gpus", "1", "--split", split, "--width", "96", "--height", "54", "--spp", "8", "--frames", "2", "--out", out],
                       check=True, timeout=120, capture_output=True, text=True)
    assert "1 GPU(s)" in r.stdout
    got = read_pfm(out)
    cfg = rta.scenes.config3(width=96, height=54, spp=8, max_depth=10)
    cfg.triangles = cfg.triangles[:0]
    rt = rta.scenes.make_renderer(cfg)
    rt.render(96, 54, 8, 10, 2)                       # the driver's second frame has seed 2
    want = rt.resolve(96, 54)[::-1, :, :3]
    rt.close()
    np.testing.assert_array_equal(got.view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
