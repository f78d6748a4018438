"""How far the bit-exact recipes are from the reference's literal arithmetic (helper of test_literal_arithmetic.py
and tools/literal_report.py).

The oracle evaluates a scene twice: in the arithmetic the HIP kernels repeat bit for bit (ORC_ARITH_RECIPE) and in
the one a compiler without contraction gives the reference's source text (ORC_ARITH_LITERAL: no FMA, IEEE divide,
v / sqrt(v.v), libm sinf / cosf, Sphere::hit's distance in double) — same scene, seed and draws.  In both tracers a
path's colour is a product of material constants, so the two frames differ exactly where a path's sequence of hits
differs.
"""
import numpy as np


def oracle_params(orc, cfg, **kw):
    args = dict(seed=cfg.seed, semantics=cfg.semantics, sample_mode=cfg.sample_mode, flags=cfg.flags,
                triangle_albedo=cfg.triangle_albedo, rng=cfg.rng, triangle_colors=cfg.triangle_colors)
    args.update(kw)
    return orc.params(cfg.width, cfg.height, cfg.spp, cfg.max_depth, **args)


def cases(scenes, full=False):
    """name -> (config, grid_n): every benchmark configuration and the CPU tracer's scene as shipped, at sizes the
    oracle renders in a second or two; full=True adds configuration 1 at its full size."""
    out = {
        "c1": (scenes.config1(width=128, height=128), 0),
        "c2": (scenes.config2(width=160, height=90, spp=25), 0),
        "c3": (scenes.config3(width=160, height=90, spp=8), 0),
        "c4": (scenes.config4(width=192, height=108, spp=16), 0),
        "c5": (scenes.config5(width=128, height=128, spp=8), 0),
        "c5e": (scenes.config5_enclosed(width=128, height=128, spp=8), 0),
        "cpu_tracer": (scenes.config_cpu_tracer(width=160, height=160), 5),
    }
    if full:
        out["c1_full"] = (scenes.config1(), 0)
    return out


def _psnr(a, b):
    mse = float(((a - b) ** 2).mean())
    return None if mse == 0.0 else round(10.0 * float(np.log10(float(a.max()) ** 2 / mse)), 2)


def _grazes(orc, cfg, p, x, y, sample, index):
    """Does the camera ray of (x, y, sample) pass a sphere's silhouette within 1e-4 of its radius?  There the
    discriminant of Sphere::hit is zero to rounding and its sign belongs to the arithmetic."""
    if index < 0 or index >= len(cfg.shapes) or int(cfg.shapes[index]["type"]) != 0:
        return False
    o, d = orc.camera_ray(cfg.camera, p, int(x), int(y), int(sample))
    c = cfg.shapes[index]["transformation"][12:15].astype(np.float64)
    r = float(cfg.shapes[index]["transformation"][0])
    oc = o.astype(np.float64) - c
    dd = d.astype(np.float64) / np.linalg.norm(d.astype(np.float64))
    miss = np.sqrt(max(0.0, float(oc @ oc) - float(oc @ dd) ** 2))
    return abs(miss / r - 1.0) < 1e-4


class RecipeOracle:
    """`ours` of study(): the oracle in the arithmetic the HIP kernels repeat bit for bit."""

    def __init__(self, orc, cfg, grid_n=0):
        self.orc, self.cfg, self.p = orc, cfg, oracle_params(orc, cfg, grid_n=grid_n)

    def trace(self, sample):
        return self.orc.trace(self.cfg.shapes, self.cfg.triangles, self.cfg.camera, self.p, sample)[0]

    def render(self):
        return self.orc.render(self.cfg.shapes, self.cfg.triangles, self.cfg.camera, self.p)[0]


def study(orc, cfg, grid_n=0, trace_samples=4, ours=None):
    """-> dict of agreement figures for one scene between `ours` — an object with trace(sample) -> hit indices
    [h, w, depth] and render() -> accumulator words; default: the oracle's recipe arithmetic, and the GPU tests pass
    the HIP renderer — and the oracle in the reference's literal arithmetic."""
    with_sampler_study = ours is None
    pr = oracle_params(orc, cfg, grid_n=grid_n)
    if ours is None:
        ours = RecipeOracle(orc, cfg, grid_n)
    pl = oracle_params(orc, cfg, grid_n=grid_n, arith=orc.ARITH_LITERAL)
    cube_ids = {i for i in range(len(cfg.shapes)) if int(cfg.shapes[i]["type"]) == 3}
    first_n = first_diff = seg_n = seg_same = 0
    unexplained, classes = [], {"cube_exact_surface_test": 0, "sphere_silhouette": 0}
    for s in range(min(trace_samples, cfg.spp)):
        hr = ours.trace(s)
        hl, _ = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, pl, s)
        d = hr[..., 0] != hl[..., 0]
        first_n += d.size
        first_diff += int(d.sum())
        for y, x in zip(*np.nonzero(d)):
            a, b = int(hr[y, x, 0]), int(hl[y, x, 0])
            if a in cube_ids or b in cube_ids:
                classes["cube_exact_surface_test"] += 1       # Cube::hit accepts a point only if its norm == 0.5 exactly (scene.h:193)
            elif _grazes(orc, cfg, pr, x, y, s, a) or _grazes(orc, cfg, pr, x, y, s, b):
                classes["sphere_silhouette"] += 1
            else:
                unexplained.append((int(x), int(y), s, a, b))
        live = (hr != -2) | (hl != -2)
        seg_n += int(live.sum())
        seg_same += int(((hr == hl) & live).sum())
    gamma = cfg.semantics == 0
    ar = ours.render()
    al, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pl)
    a = orc.resolve(ar, cfg.spp, cfg.semantics, gamma_sqrt=gamma)[..., :3].astype(np.float64)
    b = orc.resolve(al, cfg.spp, cfg.semantics, gamma_sqrt=gamma)[..., :3].astype(np.float64)
    out = {
        "frame": f"{cfg.width}x{cfg.height}, {cfg.spp} spp, depth {cfg.max_depth}",
        "first_hits_compared": first_n, "first_hits_differing": first_diff,
        "first_hit_agreement_percent": round(100.0 * (first_n - first_diff) / first_n, 4),
        "first_hit_differences_by_cause": classes, "first_hit_differences_unexplained": unexplained,
        "segment_hit_agreement_percent": round(100.0 * seg_same / seg_n, 4),
        "image_max_abs": round(float(np.abs(a - b).max()), 6),
        "image_pixels_differing_percent": round(100.0 * float((np.abs(a - b).max(axis=-1) > 0).mean()), 4),
        "image_psnr_db": _psnr(a, b),                  # None: the two frames are identical
        "image_mean_relative_difference": round(abs(float(a.mean()) - float(b.mean())) / float(a.mean()), 6),
    }
    if cfg.semantics == 0 and with_sampler_study:
        # utils.h:45 as written — three normal deviates, normalised — maps the draws to directions differently, so
        # frames agree only statistically; the same frame with another seed gives the noise floor to read it against.
        pg = oracle_params(orc, cfg, grid_n=grid_n, arith=orc.ARITH_LITERAL, unit_sampler=orc.UNIT_GAUSS)
        ag, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pg)
        c = orc.resolve(ag, cfg.spp, cfg.semantics, gamma_sqrt=gamma)[..., :3].astype(np.float64)
        p2 = oracle_params(orc, cfg, grid_n=grid_n)
        p2.seed = cfg.seed + 1
        a2, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p2)
        e = orc.resolve(a2, cfg.spp, cfg.semantics, gamma_sqrt=gamma)[..., :3].astype(np.float64)
        out["normalised_gaussian_sampler"] = {
            "image_psnr_db": _psnr(a, c), "image_mean_relative_difference": round(abs(float(a.mean()) - float(c.mean())) / float(a.mean()), 6),
            "noise_floor_other_seed_psnr_db": _psnr(a, e),
            "noise_floor_other_seed_mean_relative_difference": round(abs(float(a.mean()) - float(e.mean())) / float(a.mean()), 6)}
    return out


def no_worse_than(figures, frozen):
    """-> list of complaints: where `figures` (a study() result) are further from the literal arithmetic than the
    frozen bounds of tests/golden/literal_bounds_frozen.json allow."""
    bad = []
    if figures["first_hit_differences_unexplained"]:
        bad.append(f"first hits differ away from a silhouette or the Cube: {figures['first_hit_differences_unexplained'][:3]}")
    if figures["first_hits_differing"] > frozen["first_hits_differing_max"]:
        bad.append(f"first hits differing {figures['first_hits_differing']} > {frozen['first_hits_differing_max']}")
    if figures["segment_hit_agreement_percent"] < frozen["segment_hit_agreement_percent_min"] - 1e-4:
        bad.append(f"segments agreeing {figures['segment_hit_agreement_percent']} % < {frozen['segment_hit_agreement_percent_min']} %")
    psnr, floor = figures["image_psnr_db"], frozen["image_psnr_db_min"]          # None: identical frames
    if floor is None and psnr is not None:
        bad.append(f"frames were identical, PSNR now {psnr} dB")
    if floor is not None and psnr is not None and psnr < floor - 0.01:
        bad.append(f"PSNR {psnr} dB < {floor} dB")
    if figures["image_max_abs"] > frozen["image_max_abs_max"] + 1e-6:
        bad.append(f"max-abs {figures['image_max_abs']} > {frozen['image_max_abs_max']}")
    return bad
