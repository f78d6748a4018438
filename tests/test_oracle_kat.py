"""Known-answer tests that pin the CPU oracle: closed forms for every routine it restates.
The reference holds no golden vectors, so these (plus tests/golden) are what the oracle stands on."""
import math

import numpy as np
import pytest


def lowbias32(x):
    x &= 0xffffffff
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xffffffff
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xffffffff
    x ^= x >> 16
    return x


def test_hash_and_draws(orc):
    for x in (0, 1, 2, 0xdeadbeef, 0xffffffff, 12345):
        assert orc.lib.orc_hash32(x) == lowbias32(x)
    key = orc.lib.orc_path_key(1, 77, 3)
    sk = lowbias32((1 & 0xffffffff) ^ lowbias32(0 ^ 0x85ebca6b))
    assert key == (lowbias32(sk ^ 77) + 3 * 0x85ebca6b) & 0xffffffff
    for i in range(16):
        u = orc.lib.orc_draw(key, i)
        x = lowbias32((key + (i // 2) * 0x9e3779b9) & 0xffffffff)       # one hashed word per pair of draws
        if i & 1:
            x = (x * 0xadb4a92d) & 0xffffffff
        assert u == (x >> 8) / 2 ** 24
        assert 0.0 <= u < 1.0
    us = np.array([orc.lib.orc_draw(lowbias32(k), i) for k in range(200) for i in range(50)])
    assert abs(us.mean() - 0.5) < 0.01 and abs(us.var() - 1 / 12) < 0.005
    # The two draws of a pair share a hashed word: they must still fill the unit square evenly
    # (a 16 x 16 histogram of 65 536 pairs: chi-square with 255 degrees of freedom) and be uncorrelated.
    pairs = np.array([[orc.lib.orc_draw(lowbias32(k) + s * 0x85ebca6b, 2 * j), orc.lib.orc_draw(lowbias32(k) + s * 0x85ebca6b, 2 * j + 1)]
                      for k in range(16) for s in range(256) for j in range(16)])
    hist = np.histogram2d(pairs[:, 0], pairs[:, 1], bins=16, range=[[0, 1], [0, 1]])[0]
    expected = len(pairs) / 256.0
    assert ((hist - expected) ** 2 / expected).sum() < 340.0            # p ~ 3e-4 above this
    assert abs(np.corrcoef(pairs[:, 0], pairs[:, 1])[0, 1]) < 0.02
    # streams of neighbouring pixels / samples are different
    assert len({orc.lib.orc_path_key(1, p, s) for p in range(64) for s in range(64)}) == 4096


def test_shader_random_stream(orc):
    """random() of RayTracing.glsl:44-51 and its seed (:9), restated."""
    import ctypes as C
    f32 = np.float32
    s = (C.c_float * 3)()
    # sample 237 of pixel (12, 5) in a 64 x 36 frame: frame 2, local invocation (7, 3)
    orc.lib.orc_glsl_seed(1000, 64, 36, 12, 5, 237, s)
    time = f32(f32(1002) * f32(0.0001))
    want = [f32(time + f32(f32(127) * f32(f32(1) / f32(64)))), f32(time + f32(f32(53) * f32(f32(1) / f32(36)))), time]
    assert [f32(v) for v in s] == want
    us = []
    state = np.array(want, dtype=np.float64)
    for n in range(200):
        u = orc.lib.orc_glsl_random(s)
        us.append(u)
        assert 0.0 <= u <= 1.0
        if n < 5:       # against the formula in double precision; the float sine of an angle in the thousands agrees to ~1e-3 turns
            state += [1.78, 3.1, 50.78]
            x = state @ [12.9898, 78.233, 51.112]
            assert abs(float(s[0]) - state[0]) < 1e-5 and abs(float(s[2]) - state[2]) < 1e-4
    us = np.array(us)
    # fract(sin * 43758.5) keeps only ~8 fractional bits in binary32 — a property of the shader's generator itself
    assert 0.35 < us.mean() < 0.65 and len(np.unique(us)) > 100
    # different invocations give different streams
    a, b = (C.c_float * 3)(), (C.c_float * 3)()
    orc.lib.orc_glsl_seed(1000, 64, 36, 12, 5, 0, a); orc.lib.orc_glsl_seed(1000, 64, 36, 12, 5, 1, b)
    assert orc.lib.orc_glsl_random(a) != orc.lib.orc_glsl_random(b)


def test_sincos2pi(orc):
    # half-angle form: exact values are met to a few 1e-7, not bit for bit
    for u, want in ((0.0, (1.0, 0.0)), (0.25, (0.0, 1.0)), (0.5, (-1.0, 0.0)), (0.75, (0.0, -1.0)), (0.125, (2 ** -0.5, 2 ** -0.5))):
        c, s = orc.sincos2pi(u)
        assert abs(c - want[0]) < 6e-7 and abs(s - want[1]) < 6e-7
    us = (np.arange(0, 1 << 24, 4099, dtype=np.int64) / 2.0 ** 24).astype(np.float32)
    err = 0.0
    for u in us:
        c, s = orc.sincos2pi(float(u))
        err = max(err, abs(c - math.cos(2 * math.pi * float(u))), abs(s - math.sin(2 * math.pi * float(u))))
        assert abs(c * c + s * s - 1) < 1.5e-6
    assert err < 6e-7


def test_logarithm_recipe_of_the_gaussian_sampler(orc):
    """orc_log_draw(k * 2^-24) against math.log: 6e-7 absolute over draws from the smallest to the largest."""
    import math
    ks = list(range(1, 4097)) + [2 ** j for j in range(12, 24)] + [2 ** j + 1 for j in range(12, 24)] + [2 ** 24 - 1, 2 ** 24 - 2, 11863283, 11863284]
    worst = max(abs(orc.log_draw(k * 2.0 ** -24) - math.log(k * 2.0 ** -24)) for k in ks)
    assert worst < 1e-6
    assert orc.log_draw(0.5) == np.float32(-math.log(2.0))          # m = 1 exactly: e * ln 2


def test_unit_vector_is_uniform_on_the_sphere(orc):
    rng = np.random.RandomState(1)
    u = (rng.randint(0, 1 << 24, (4000, 2)) / 2.0 ** 24).astype(np.float32)
    v = np.array([orc.unit_vector(float(a), float(b)) for a, b in u], dtype=np.float64)
    assert np.abs(np.linalg.norm(v, axis=1) - 1).max() < 3e-7
    assert np.abs(v.mean(0)).max() < 0.04                    # E[v] = 0
    np.testing.assert_allclose((v ** 2).mean(0), [1 / 3] * 3, atol=0.02)   # E[v_i^2] = 1/3
    np.testing.assert_array_equal(orc.unit_vector(0.0, 0.0), [0, 0, 1])   # z = 1 - 2*u1
    assert orc.unit_vector(0.5, 0.0)[2] == 0.0


def test_hemisphere_glsl(orc):
    rng = np.random.RandomState(2)
    for n in ([0, 1, 0], [0, 0, 1], [0, 0, -1], [1, 0, 0], [0.6, 0.0, 0.8], [0.001, 0.002, 0.999997]):
        n = np.array(n, dtype=np.float32); n /= np.float32(np.linalg.norm(n))
        for _ in range(100):
            u1, u2 = (rng.randint(0, 1 << 24, 2) / 2.0 ** 24).astype(np.float32)
            d = orc.hemisphere_glsl(n, float(u1), float(u2)).astype(np.float64)
            assert abs(np.linalg.norm(d) - 1) < 2e-6
            assert abs(d @ n - u2) < 2e-6                 # RayTracing.glsl:78: component along n is r = u2
    # n = +-z takes the fallback tangent of RayTracing.glsl:72-73
    d = orc.hemisphere_glsl([0, 0, 1], 0.0, 0.0)
    np.testing.assert_allclose(d, [0, -1, 0], atol=6e-7)


def test_hit_sphere_closed_forms(orc):
    inf = float("inf")
    assert orc.hit_sphere((0, 0, 0), 1.0, (0, 0, 5), (0, 0, -1)) == 4.0        # head-on, near root
    assert orc.hit_sphere((0, 0, 0), 1.0, (0, 0, 5), (0, 0, 1)) == -6.0        # behind: -b - sqrt(disc) < 0
    assert orc.hit_sphere((0, 0, 0), 1.0, (0, 2, 5), (0, 0, -1)) == inf        # miss
    assert orc.hit_sphere((0, 0, 0), 1.0, (0, 1, 5), (0, 0, -1)) == 5.0        # tangent: disc = 0
    assert orc.hit_sphere((0, 0, 0), 1.0, (0, 0, 0), (0, 0, -1)) == -1.0       # from inside: near root is behind
    assert orc.hit_sphere((1, 2, -2), 0.5, (1, 2, 6), (0, 0, -1)) == 7.5
    d = np.array([3, 4, 12], dtype=np.float64) / 13
    t = orc.hit_sphere((3, 4, 12), 2.0, (0, 0, 0), tuple(d))
    assert abs(t - 11.0) < 1e-5


def test_hit_box_closed_forms(orc):
    # a / b of the shader routines is a * rcp_nr(b) (relative error <= 1.2e-7): finite answers to 1e-6
    inf = float("inf")
    assert abs(orc.hit_box((0, 0, 0), 2.0, (0.25, 0.5, 5), (0, 0, -1)) - 4.0) < 1e-6
    assert abs(orc.hit_box((0, 0, 0), 2.0, (5, 0.25, 0.5), (-1, 0, 0)) - 4.0) < 1e-6
    assert abs(orc.hit_box((0, 0, 0), 2.0, (0.25, -5, 0.5), (0, 1, 0)) - 4.0) < 1e-6
    assert orc.hit_box((0, 0, 0), 2.0, (3, 0.5, 5), (0, 0, -1)) == inf          # passes beside
    assert abs(orc.hit_box((1, 1, 1), 1.0, (1.25, 1.1, 4), (0, 0, -1)) - 2.5) < 1e-6
    # sign(0) = 0 -> 0/0 -> NaN -> rejected (RayTracing.glsl:144-145): an origin component exactly 0
    # still hits through the other axes
    assert abs(orc.hit_box((0, 0, 0), 2.0, (0, 0.5, 5), (0, 0, -1)) - 4.0) < 1e-6
    np.testing.assert_allclose(orc.box_normal((0, 0, 0), 2.0, (0.3, 0.2, 1.0)), [0, 0, 1], atol=2e-7)
    np.testing.assert_allclose(orc.box_normal((0, 0, 0), 2.0, (-1.0, 0.2, 0.1)), [-1, 0, 0], atol=2e-7)
    np.testing.assert_allclose(orc.box_normal((0, 0, 0), 2.0, (1.0, 1.0, 0.1)), [2 ** -0.5, 2 ** -0.5, 0], atol=2e-7)
    # only transformation[0][0] scales the box (RayTracing.glsl:139): Box({..},{10,1,1}) is a 10-cube
    assert abs(orc.hit_box((-1, 12, -2), 10.0, (-1, 0, -2.5), (0, 1, 0)) - 7.0) < 1e-5


def test_hit_box_with_its_own_sides(orc):
    """ORC_FLAG_BOX_SIDES: the three sides Box::Box stores (shape.h:67), closed forms per face."""
    inf = float("inf")
    c, sides = (1.0, 2.0, 3.0), (4.0, 2.0, 1.0)               # x in [-1,3], y in [1,3], z in [2.5,3.5]
    assert abs(orc.hit_box3(c, sides, (1.5, 2.25, 10), (0, 0, -1)) - 6.5) < 1e-5
    assert abs(orc.hit_box3(c, sides, (10, 2.25, 3.1), (-1, 0, 0)) - 7.0) < 1e-5
    assert abs(orc.hit_box3(c, sides, (2.5, -5, 3.1), (0, 1, 0)) - 6.0) < 1e-5
    assert orc.hit_box3(c, sides, (2.5, 3.5, 10), (0, 0, -1)) == inf             # above the box: y = 3.5 > 3
    assert abs(orc.hit_box((1, 2, 3), 4.0, (2.5, 3.5, 10), (0, 0, -1)) - 5.0) < 1e-5   # the shader's 4-cube is hit there
    # equal sides give the shader's function bit for bit
    for o, d in (((0.3, 0.2, 5.0), (0.0, 0.0, -1.0)), ((4.0, 0.1, 0.2), (-0.8, 0.0, 0.6))):
        assert orc.hit_box3((0, 0, 0), (2.0, 2.0, 2.0), o, d) == orc.hit_box((0, 0, 0), 2.0, o, d)
    np.testing.assert_allclose(orc.box_normal3(c, sides, (1.5, 2.25, 3.5)), [0, 0, 1], atol=2e-7)
    np.testing.assert_allclose(orc.box_normal3(c, sides, (-1.0, 2.25, 3.1)), [-1, 0, 0], atol=2e-7)
    np.testing.assert_allclose(orc.box_normal3(c, sides, (2.5, 1.0, 3.1)), [0, -1, 0], atol=2e-7)


def test_triangle_colours_replace_the_single_albedo(orc, rta):
    """With equal colours per triangle the frame equals the one rendered with that triangle_albedo; with
    different ones it differs."""
    cfg = rta.scenes.config_colored_mesh(width=64, height=40, spp=4)
    n = len(cfg.triangles)
    base = dict(seed=cfg.seed, semantics=cfg.semantics, triangle_albedo=(0.9, 0.4, 0.1))
    a, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(64, 40, 4, 6, **base))
    same = np.tile(np.float32([0.9, 0.4, 0.1]), (n, 1))
    b, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(64, 40, 4, 6, triangle_colors=same, **base))
    np.testing.assert_array_equal(a, b)
    c, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(64, 40, 4, 6, triangle_colors=cfg.triangle_colors, **base))
    assert (a != c).any()


def test_hit_cube_closed_forms(rta, orc):
    """Cube::hit of the CPU tracer (scene.h:185-204): unit cube, exact == 0.5 surface test, + axis normal."""
    ident = np.eye(4, dtype=np.float32).T.reshape(-1)
    t, ip, n = orc.hit_cube(ident, (0, 0, 5), (0, 0, -1))
    assert t == 4.5 and tuple(ip) == (0, 0, 0.5) and tuple(n) == (0, 0, 1)
    t, ip, n = orc.hit_cube(ident, (0, 0, -5), (0, 0, 1))
    assert t == 4.5 and tuple(ip) == (0, 0, -0.5) and tuple(n) == (0, 0, 1)      # sign from an absolute value: always +
    t, ip, n = orc.hit_cube(ident, (0.25, 3, 0.125), (0, -1, 0))
    assert t == 2.5 and tuple(n) == (0, 1, 0)
    assert orc.hit_cube(ident, (2, 0, 5), (0, 0, -1)) is None                     # passes beside
    c = rta.Cube((1, 2, 3), rta.Lambertian((1, 1, 1)))
    m = c.rec["transformation"]
    t, ip, n = orc.hit_cube(m, (1, 2, 8), (0, 0, -1))
    assert t == 4.5 and tuple(n) == (0, 0, 1)
    # rotate(0.5): Rx(pi/2) * Ry(pi/2); the matrix stays orthonormal and the normal is a row of it
    c.rotate(0.5)
    R = c.transformation[:3, :3].astype(np.float64)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    hits = 0
    rng = np.random.RandomState(4)
    for _ in range(400):
        o = rng.uniform(-0.3, 0.3, 3) + np.array([1, 2, 8]); d = np.array([1, 2, 3]) - o + rng.uniform(-0.2, 0.2, 3); d /= np.linalg.norm(d)
        r = orc.hit_cube(c.rec["transformation"], tuple(o.astype(np.float32)), tuple(d.astype(np.float32)))
        if r is not None:
            hits += 1
            assert abs(np.abs(r[1]).max() - 0.5) == 0 and any(np.allclose(r[2], R[i], atol=1e-6) for i in range(3))
    assert hits > 20            # the exact == 0.5 test of scene.h:193 passes only now and then


def test_hit_triangle_closed_forms(orc):
    inf = float("inf")
    t = orc.make_triangle((0, 0, 0), (1, 0, 0), (0, 1, 0))
    assert abs(orc.hit_triangle(t, (0.25, 0.25, 3), (0, 0, -1)) - 3.0) < 1e-6
    assert abs(orc.hit_triangle(t, (0.25, 0.25, -3), (0, 0, 1)) - 3.0) < 1e-6             # two-sided
    assert orc.hit_triangle(t, (0.75, 0.75, 3), (0, 0, -1)) == inf             # x + y >= 1
    assert orc.hit_triangle(t, (-0.1, 0.25, 3), (0, 0, -1)) == inf
    assert orc.hit_triangle(t, (0.25, 0.25, 3), (0, 0, 1)) == inf              # k < 0
    assert orc.hit_triangle(t, (0.25, 0.25, 3), (1, 0, 0)) == inf              # parallel: k = inf or nan
    rng = np.random.RandomState(5)
    for _ in range(100):
        p = rng.uniform(-1, 1, (3, 3)).astype(np.float32)
        tri = orc.make_triangle(*p)
        bary = rng.dirichlet([1, 1, 1])
        target = (bary[:, None] * p.astype(np.float64)).sum(0)
        o = target + rng.uniform(-1, 1, 3) + np.array([0, 0, 4])
        d = target - o; dist = np.linalg.norm(d); d /= dist
        got = orc.hit_triangle(tri, tuple(o.astype(np.float32)), tuple(d.astype(np.float32)))
        if bary.min() > 0.02:
            assert abs(got - dist) < 1e-3 * dist


def test_camera_rays(rta, orc):
    cam = rta.Camera(64, 36, 1.0, (1, 0, 6)).data
    step = float(cam[19])
    ul = cam[8:11].astype(np.float64)
    # jitter (RayTracing.glsl:296): origin inside the pixel, on the image plane, unit direction
    p = orc.params(64, 36, 4, 4, seed=9)
    for gx, gy, s in [(0, 0, 0), (63, 35, 3), (10, 20, 1)]:
        o, d = orc.camera_ray(cam, p, gx, gy, s)
        assert ul[0] + step * gx <= o[0] < ul[0] + step * (gx + 1) + 1e-6
        assert ul[1] - step * (gy + 1) - 1e-6 < o[1] <= ul[1] - step * gy
        assert o[2] == cam[10]
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 2e-7
        w = o.astype(np.float64) - cam[0:3]
        np.testing.assert_allclose(d, w / np.linalg.norm(w), atol=2e-7)
    # regular grid: offsets (j, -i) * step/(N+1)
    p = orc.params(64, 36, 16, 4, sample_mode=orc.SAMPLE_GRID)
    o00, _ = orc.camera_ray(cam, p, 5, 7, 0)
    np.testing.assert_allclose(o00, [ul[0] + 5 * step, ul[1] - 7 * step, ul[2]], atol=1e-6)
    for s in range(16):
        i, j = s // 4, s % 4
        o, _ = orc.camera_ray(cam, p, 5, 7, s)
        np.testing.assert_allclose(o - o00, [j * step / 5, -i * step / 5, 0], atol=1e-7)
    # rayTracer.h:98-99 as written: the x offset accumulates 0, 1, 3, 6 (x step/(N+1))
    p = orc.params(64, 36, 16, 4, sample_mode=orc.SAMPLE_GRID_COMPAT)
    for s in range(16):
        i, j = s // 4, s % 4
        o, _ = orc.camera_ray(cam, p, 5, 7, s)
        np.testing.assert_allclose(o - o00, [j * (j + 1) / 2 * step / 5, -i * step / 5, 0], atol=1e-7)


def one_light_scene(rta):
    return rta.pack([rta.Circle((1, 0, -2), 1.0, rta.Light((5, 4, 3)))], rta.SHAPE_DTYPE)


@pytest.mark.parametrize("sem", [0, 1])
def test_direct_light_image(rta, orc, sem):
    """A lone emitter: pixels whose ray hits it get its colour, all others are black."""
    shapes = one_light_scene(rta)
    cam = rta.Camera(64, 64, 1.0, (1, 0, 6)).data
    p = orc.params(64, 64, 4, 3, semantics=sem)
    acc, st = orc.render(shapes, np.zeros(0, rta.TRIANGLE_DTYPE), cam, p)
    img = orc.resolve(acc, 4, sem, flip_rows=False)
    hit, mat = orc.trace(shapes, np.zeros(0, rta.TRIANGLE_DTYPE), cam, p, 0)
    lit = hit[:, :, 0] == 0
    # projected disc: tan(angle) = 1/sqrt(8^2-1); image plane at distance 1, pixel = 1/64
    r_pix = 64 / math.sqrt(63)
    assert abs(lit.sum() - math.pi * r_pix ** 2) < 0.08 * math.pi * r_pix ** 2
    inner = np.zeros_like(lit); yy, xx = np.mgrid[0:64, 0:64]
    inner[(xx - 31.5) ** 2 + (yy - 31.5) ** 2 < (r_pix - 1.5) ** 2] = True
    if sem == 0:
        np.testing.assert_array_equal(img[inner][:, :3], np.tile(np.float32([5, 4, 3]), (inner.sum(), 1)))
    else:       # RayTracing.glsl:301: uint(255*c) per sample, / (255*spp)
        np.testing.assert_allclose(img[inner][:, :3], np.tile([5, 4, 3], (inner.sum(), 1)), rtol=1e-6)
    ever = np.zeros_like(lit)
    for smp in range(4):
        ever |= orc.trace(shapes, np.zeros(0, rta.TRIANGLE_DTYPE), cam, p, smp)[0][:, :, 0] == 0
    assert np.all(img[~ever][:, :3] == 0)
    assert np.all(mat[lit][:, 0] == 0) and np.all(hit[:, :, 1] == -2)        # path ends at the light
    assert st["samples"] == 64 * 64 * 4 and st["terminal"] == st["samples"] and st["segments"] == st["samples"]


def test_depth_rules(rta, orc):
    """max_depth counts ray segments: 1 -> only directly visible emitters (rayTracer.h:113-115, glsl:263)."""
    cfg = rta.scenes.config1(48, 48, 4, 1)
    p = orc.params(48, 48, 4, 1)
    acc, st = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
    assert st["segments"] == st["samples"]
    ever = np.zeros((48, 48), bool)
    for smp in range(4):
        ever |= orc.trace(cfg.shapes, cfg.triangles, cfg.camera, p, smp)[0][:, :, 0] == 3
    img = acc.view(np.float32)
    assert ever.any() and np.all(img[~ever][:, :3] == 0) and np.all(img[ever][:, :3] > 0)
    p8 = orc.params(48, 48, 4, 8)
    _, st8 = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p8)
    assert st8["segments"] > st["segments"] and st8["segments"] <= 8 * st8["samples"]
    assert st8["diffuse"] + st8["terminal"] >= st8["segments"] - st8["samples"]


def test_recursive_and_forward_products_agree(rta, orc):
    """rayTracer.h:121 multiplies on the way out of the recursion; the forward running product differs
    only by rounding, and the paths (hit ids) are the same."""
    cfg = rta.scenes.config4(40, 24, 16, 11)
    pa = orc.params(40, 24, 16, 11, color_order=orc.COLOR_AS_WRITTEN)
    pf = orc.params(40, 24, 16, 11, color_order=orc.COLOR_FORWARD)
    a, sa = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pa)
    f, sf = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pf)
    assert sa == sf
    np.testing.assert_allclose(a.view(np.float32), f.view(np.float32), rtol=2e-6, atol=1e-7)
    ha, ma = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, pa, 5)
    hf, mf = orc.trace(cfg.shapes, cfg.triangles, cfg.camera, pf, 5)
    np.testing.assert_array_equal(ha, hf); np.testing.assert_array_equal(ma, mf)


def test_direct_sphere_sampler_matches_normalised_gaussians(rta, orc):
    """utils.h:45 normalises three normal deviates; the oracle samples the same distribution directly.
    Two renders of the same scene must agree to within Monte-Carlo noise."""
    cfg = rta.scenes.config1(32, 32, 1024, 6)
    pd = orc.params(32, 32, 1024, 6, seed=11, unit_sampler=orc.UNIT_DIRECT)
    pg = orc.params(32, 32, 1024, 6, seed=12, unit_sampler=orc.UNIT_GAUSS)
    pd2 = orc.params(32, 32, 1024, 6, seed=13, unit_sampler=orc.UNIT_DIRECT)
    d = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pd)[0].view(np.float32)[..., :3] / 1024
    g = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pg)[0].view(np.float32)[..., :3] / 1024
    d2 = orc.render(cfg.shapes, cfg.triangles, cfg.camera, pd2)[0].view(np.float32)[..., :3] / 1024
    noise = np.abs(d - d2).mean()                    # two seeds of the same sampler
    assert np.abs(d - g).mean() < 1.5 * noise + 1e-4
    assert abs(d.mean() - g.mean()) < 0.05 * d.mean()


def test_glsl_quantisation_and_resolve(rta, orc):
    """Light (1.5,1.5,1.5) -> uint(382) per sample (RayTracing.glsl:301), resolve / (255*spp) (PixelConversion.glsl:17)."""
    shapes = rta.pack([rta.Box((1, 1, -3), (2, 2, 2), rta.Light((1.5, 1.5, 1.5)))], rta.SHAPE_DTYPE)
    cam = rta.Camera(16, 16).data
    p = orc.params(16, 16, 7, 10, semantics=1)
    acc, _ = orc.render(shapes, np.zeros(0, rta.TRIANGLE_DTYPE), cam, p)
    assert acc[8, 8, 0] == 7 * 382 and acc[8, 8, 3] == 0
    img = orc.resolve(acc, 7, 1)
    assert img[16 - 1 - 8, 8, 0] == np.float32(np.float32(7 * 382) / np.float32(255 * 7)) and img[0, 0, 3] == 1.0


def test_resolve_flip_gamma_and_running_mean(orc):
    rng = np.random.RandomState(0)
    acc = rng.uniform(0, 50, (6, 5, 4)).astype(np.float32)
    img = orc.resolve(acc.view(np.uint32), 10, 0, gamma_sqrt=True, flip_rows=True)
    np.testing.assert_array_equal(img[::-1, :, :3], np.sqrt(acc[..., :3] / np.float32(10)))    # rayTracer.h:78,105
    assert np.all(img[..., 3] == 1)
    # PixelConversion.glsl:21-23 over three frames == mean of the frames
    image = np.zeros((6, 5, 4), np.float32)
    frames = [rng.randint(0, 255 * 100, (6, 5, 4)).astype(np.uint32) for _ in range(3)]
    for n, f in enumerate(frames, 1):
        a = f.copy()
        orc.convert(a, image, 100, 1, n)
        assert not a.any()                                                                    # :25-27
    want = sum(f[::-1, :, :3].astype(np.float64) / 25500 for f in frames) / 3
    np.testing.assert_allclose(image[..., :3], want, rtol=1e-5)
    np.testing.assert_allclose(image[..., 3], 1.0, rtol=1e-6)


def test_row_shards_partition_the_frame(rta, orc):
    cfg = rta.scenes.config1(40, 50, 4, 4)
    full, sfull = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(40, 50, 4, 4))
    total = np.zeros_like(full); segs = 0
    for r in range(3):
        part, st = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(40, 50, 4, 4, shard=(r, 3, 16)))
        rows = [y for y in range(50) if (y // 16) % 3 == r]
        other = [y for y in range(50) if (y // 16) % 3 != r]
        assert not part[other].any()
        np.testing.assert_array_equal(part[rows], full[rows])
        total += part; segs += st["segments"]
    np.testing.assert_array_equal(total, full)
    assert segs == sfull["segments"]


def test_progressive_chunks_equal_one_shot(rta, orc):
    """Sums continue across calls in sample order, so 16 = 4+4+8 samples bit for bit."""
    cfg = rta.scenes.config1(24, 24, 16, 5)
    one, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(24, 24, 16, 5))
    acc = None
    for first, n in ((0, 4), (4, 4), (8, 8)):
        acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, orc.params(24, 24, n, 5, first_sample=first), accum=acc)
    np.testing.assert_array_equal(acc, one)


def test_dielectric_obeys_snell_and_schlick(orc):
    """The dielectric (material type 3, defined by this build: the reference has no refraction) at a surface with
    normal +z: refracted rays obey sin(t) = sin(i) / ior and stay in the plane of incidence; the mirrored ray is
    reflect(d, n); the reflectance is Schlick's, r0 = ((1 - ior) / (1 + ior))^2 at normal incidence, 1 at grazing."""
    n = (0.0, 0.0, 1.0)
    for ior in (1.5, 1.33, 2.4):
        for deg in (0.0, 10.0, 35.0, 60.0, 80.0):
            si, ci = np.sin(np.radians(deg)), np.cos(np.radians(deg))
            d = (float(si), 0.0, float(-ci))
            refr = orc.glass_direction(n, d, ior, 0.999999)            # u1 above any reflectance below grazing
            refr = refr / np.linalg.norm(refr)
            assert abs(refr[1]) < 1e-7 and refr[2] < 0
            assert abs(refr[0] - si / ior) < 2e-6                       # Snell
            mirr = orc.glass_direction(n, d, ior, 0.0)
            np.testing.assert_allclose(mirr, [si, 0.0, ci], atol=2e-7)  # reflect(d, n), utils.h:62-64
    r0 = ((1 - 1.5) / (1 + 1.5)) ** 2                                   # 0.04
    d0 = (0.0, 0.0, -1.0)
    assert orc.glass_direction(n, d0, 1.5, r0 * 0.99)[2] > 0            # below r0: mirrored
    assert orc.glass_direction(n, d0, 1.5, r0 * 1.01)[2] < 0            # above: refracted
    graze = (float(np.sin(np.radians(89.9))), 0.0, float(-np.cos(np.radians(89.9))))
    assert orc.glass_direction(n, graze, 1.5, 0.98)[2] > 0              # reflectance -> 1 at grazing incidence


def test_floor_under_a_ceiling_light_has_the_closed_form_radiance(rta, orc):
    """A diffuse floor under an emitter that fills all of its sky but a rim at the horizon: one bounce reaches the
    light or nothing, so a floor pixel converges to attenuation * Le * P(bounce reaches the light).
      Shader rules (RayTracing.glsl:64-80,241): the bounce's component along the normal is uniform in [0,1); the
        ceiling (bottom face of a 1000-box, 10 above the floor, half-width 500) is reached when it exceeds
        10/sqrt(10^2+500^2)-ish = 0.02 -> P = 0.98; attenuation = albedo.
      CPU rules (scene.h:51-58, utils.h:45-50): cosine-weighted; a sphere of radius 1000 whose lowest point is 10
        above the floor fills the cone sin(alpha) = 1000/1010 -> P = sin^2(alpha) = 0.9803; attenuation = 0.6 * albedo."""
    L, Li = rta.Lambertian, rta.Light
    no_tris = np.zeros(0, rta.TRIANGLE_DTYPE)
    # shader rules: boxes.  albedo 0.5, Le 2 -> every lit sample is exactly 1.0 (255 after quantisation)
    shapes = rta.pack([rta.Box((0, -50, 0), (100, 1, 1), L((0.5, 0.5, 0.5))), rta.Box((0, 510, 0), (1000, 1, 1), Li((2, 2, 2)))], rta.SHAPE_DTYPE)
    cam = rta.Camera(64, 64, 1.0, (0, 1, 0)).data
    p = orc.params(64, 64, 64, 3, semantics=1)
    acc, _ = orc.render(shapes, no_tris, cam, p)
    img = orc.resolve(acc, 64, 1, flip_rows=False)
    first = orc.trace(shapes, no_tris, cam, p, 0)[0][:, :, 0]
    floor = (first == 0) & (np.arange(64)[:, None] >= 44)         # steep rays: every sample of the pixel lands near the camera
    assert floor.sum() > 1000
    assert abs(img[floor][:, 0].mean() - 0.98) < 0.004
    # CPU rules: spheres.  0.6 * 0.5 * Le 2 = 0.6 per lit sample
    shapes = rta.pack([rta.Circle((0, -100, 0), 100.0, L((0.5, 0.5, 0.5))), rta.Circle((0, 1010, 0), 1000.0, Li((2, 2, 2)))], rta.SHAPE_DTYPE)
    p = orc.params(64, 64, 64, 3, semantics=0)
    acc, _ = orc.render(shapes, no_tris, cam, p)
    img = orc.resolve(acc, 64, 0, flip_rows=False)
    first = orc.trace(shapes, no_tris, cam, p, 0)[0][:, :, 0]
    near = (first == 0) & (np.arange(64)[:, None] >= 48)          # steep rays: floor points close to the sphere's top
    assert near.sum() > 500
    assert abs(img[near][:, 0].mean() - 0.6 * (1000.0 / 1010.0) ** 2) < 0.006



def test_tangent_fallback_threshold(orc):
    """randomOnHalfUnitShere switches its tangent when length((-n.y, n.x)) < 0.01 (RayTracing.glsl:72).  The kernel compares
    the squared length with the first float whose correctly rounded root reaches 0.01f instead of taking the root."""
    t = np.float32(0.01)
    first = np.uint32(0x38d1b716).view(np.float32)
    assert np.sqrt(first, dtype=np.float32) >= t and np.sqrt(np.nextafter(first, np.float32(0)), dtype=np.float32) < t
    # both sides of the switch through the oracle: a normal with |n.xy| just below and just above 0.01
    for nxy, fallback in ((0.0099, True), (0.0101, False)):
        n = np.array([nxy, 0.0, math.sqrt(1.0 - nxy * nxy)], dtype=np.float32)
        v = orc.hemisphere_glsl(tuple(float(x) for x in n), 0.0, 0.0)        # angle 0, r 0: the first tangent itself
        want = np.array([0.0, -n[2], n[1]]) / math.hypot(n[2], n[1]) if fallback else np.array([-n[1], n[0], 0.0]) / math.hypot(n[0], n[1])
        np.testing.assert_allclose(np.asarray(v, dtype=np.float64), want, atol=2e-6)


def test_sphere_normals_are_unit_and_ignore_the_radius_sign(rta, orc):
    """normalAt (scene.h:130-132) is (point - centre).normalized(): unit length wherever the sphere is, and the same
    for a radius of either sign (Sphere::hit squares it, scene.h:118).  The recipe's one Newton step from 1/|radius|
    must fall back on the full normalisation where the hit point is off the sphere by more than it can absorb."""
    frames = {}
    for which, radius in (("far", None), ("near", None), ("near", -0.5)):
        cfg = rta.scenes.config_sphere_normals(which, radius)
        for arith in (orc.ARITH_RECIPE, orc.ARITH_LITERAL):
            p = orc.params(cfg.width, cfg.height, 1, cfg.max_depth, seed=1, semantics=cfg.semantics, sample_mode=cfg.sample_mode,
                           grid_n=1, arith=arith)
            acc, _ = orc.render(cfg.shapes, cfg.triangles, cfg.camera, p)
            n = acc.view(np.float32)[..., :3].astype(np.float64)          # debug material, one sample: the sum is the normal
            length = np.sqrt((n ** 2).sum(-1))
            hit = length > 0
            assert hit.sum() > 1000
            assert np.abs(length[hit] - 1.0).max() < 4e-7, (which, radius, arith)
            assert n[..., 2][hit].min() >= 0.0 and n[..., 2][hit].mean() > 0.5   # towards the camera: outward
            frames[(which, radius, arith)] = acc
    for arith in (orc.ARITH_RECIPE, orc.ARITH_LITERAL):
        np.testing.assert_array_equal(frames[("near", None, arith)], frames[("near", -0.5, arith)])
