"""A second evaluation of the reference's formulas, independent of the oracle's C: numpy binary32 scalars (every
operation rounded once, as IEEE-754 prescribes and as the shader's source text reads), written line by line from
RayTracing.glsl:64-80 (randomOnHalfUnitShere), :118-132 (hitSphere), :138-156 (hitBox), :158-163 (boxNormalAt),
:179-194 (hitTriangle) and from SimpleTriangle's constructor (src/gpu_version/shape.h:83-101).  The oracle in its
literal arithmetic (ORC_ARITH_LITERAL: no fused operations, IEEE division and square root) must return the same
bits on random inputs; only the sampler, whose sine and cosine come from two different libraries, is compared to
2e-7.  This does not pin the oracle to the reference's outputs (the reference holds none): it guards the oracle's
restatement against slips of transcription — a sign, an operand order, a missed case — by a restatement in another
language that shares no code with it."""
import numpy as np
import pytest

f = np.float32
INF = f(np.inf)


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def sign(x):
    return f(1) if x > 0 else (f(-1) if x < 0 else f(0))


def normalize(v):
    return v / np.sqrt(dot(v, v))


def cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f)


def hit_sphere(c, r, o, d):                      # RayTracing.glsl:118-132
    oc = o - c
    b = dot(d, oc)
    cc = dot(oc, oc) - r * r
    disc = b * b - cc
    if disc < 0:
        return INF
    return -b - np.sqrt(disc)


def hit_box(c, s, o, d):                         # RayTracing.glsl:138-156
    inv = f(1) / s
    o = inv * (o - c)
    d = inv * d
    sg = np.array([sign(x) for x in o], dtype=f)
    with np.errstate(all="ignore"):
        t = (f(0.5) - o * sg) / (sg * d)
    best = INF
    for k in range(3):
        with np.errstate(all="ignore"):
            p = o + d * t[k]
        n = max(abs(p[0]), max(abs(p[1]), abs(p[2])))
        if n < f(0.5001) and n > f(0.4999) and best > t[k]:
            best = t[k]
    return best


def box_normal(c, s, p):                         # RayTracing.glsl:158-163
    q = (p - c) / s
    step = np.array([f(0) if abs(x) < f(0.499) else f(1) for x in q], dtype=f)
    with np.errstate(all="ignore"):
        return normalize(step * np.array([sign(x) for x in q], dtype=f))


def hit_triangle(t, o, d):                       # RayTracing.glsl:179-194
    with np.errstate(all="ignore"):
        k = (t["beta"] - dot(t["a"], o)) / dot(t["a"], d)
    if k < 0:
        return INF
    p = k * d + o - t["position"][:3]
    x, y = dot(p, t["m1"][:3]), dot(p, t["m2"][:3])
    if x > 0 and y > 0 and x + y < 1:
        return k
    return INF


def hemisphere(n, u1, u2):                       # RayTracing.glsl:64-80
    pi = f(3.14159265358979)
    angle = u1 * f(2) * pi
    r = u2
    dist = np.sqrt(f(1) - r * r)
    x, y = f(np.cos(angle)) * dist, f(np.sin(angle)) * dist
    b1 = np.array([-n[1], n[0], f(0)], dtype=f)
    if np.sqrt(b1[0] * b1[0] + b1[1] * b1[1]) < f(0.01):
        b1 = normalize(np.array([f(0), -n[2], n[1]], dtype=f))
    else:
        b1 = normalize(b1)
    b2 = cross(n, b1)
    return b1 * x + b2 * y + n * r


def simple_triangle(p1, p2, p3):                 # shape.h:83-101
    v1, v2 = p2 - p1, p3 - p1
    a = normalize(cross(v1, v2))
    beta = dot(a, p1)
    m1 = v1 - (dot(v1, v2) / dot(v2, v2)) * v2
    m1 = m1 / dot(m1, v1)
    m2 = v2 - (dot(v1, v2) / dot(v1, v1)) * v1
    m2 = m2 / dot(m2, v2)
    return a, beta, m1, m2


def same_bits(a, b):
    a, b = np.asarray(a, dtype=f), np.asarray(b, dtype=f)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32)) or (np.isnan(a).all() and np.isnan(b).all())


@pytest.fixture(scope="module")
def literal(orc):
    with orc.arithmetic(orc.ARITH_LITERAL):
        yield orc


def vec(rng, lo=-3.0, hi=3.0):
    return rng.uniform(lo, hi, 3).astype(f)


def test_intersections_bit_for_bit(literal):
    orc = literal
    rng = np.random.default_rng(3)
    hits = [0, 0, 0]
    for _ in range(6000):
        c, o, d = vec(rng), vec(rng), vec(rng)
        d = (d / np.linalg.norm(d)).astype(f)
        r, s = f(rng.uniform(0.2, 2.0)), f(rng.uniform(0.3, 3.0))
        a = hit_sphere(c, r, o, d)
        assert same_bits(a, orc.hit_sphere(c, float(r), o, d))
        hits[0] += bool(np.isfinite(a))
        a = hit_box(c, s, o, d)
        assert same_bits(a, orc.hit_box(c, float(s), o, d))
        hits[1] += bool(np.isfinite(a))
        t = orc.make_triangle(vec(rng), vec(rng), vec(rng))
        a = hit_triangle(t, o, d)
        assert same_bits(a, orc.hit_triangle(t, o, d))
        hits[2] += bool(np.isfinite(a))
    assert min(hits) > 150                       # the comparison saw crossings, not only misses


def test_axis_parallel_rays_and_origins_on_a_plane(literal):
    """sign(0) = 0 gives 0/0 (RayTracing.glsl:144-145) and a direction component of 0 an infinity: the cases a
    restatement is most likely to get wrong."""
    orc = literal
    rng = np.random.default_rng(5)
    for _ in range(2000):
        c, s = vec(rng), f(rng.uniform(0.3, 3.0))
        o, d = vec(rng), vec(rng)
        k = int(rng.integers(0, 3))
        if rng.random() < 0.5:
            d[k] = f(0)
        if rng.random() < 0.5:
            o[int(rng.integers(0, 3))] = c[int(rng.integers(0, 3))]
        if rng.random() < 0.3:
            o[k] = c[k]                          # origin on the box's centre plane along k
        n = np.linalg.norm(d)
        if n == 0:
            continue
        d = (d / n).astype(f)
        assert same_bits(hit_box(c, s, o, d), orc.hit_box(c, float(s), o, d))


def test_box_normal_and_triangle_constructor_bit_for_bit(literal):
    orc = literal
    rng = np.random.default_rng(4)
    for _ in range(3000):
        c, s = vec(rng), f(rng.uniform(0.3, 3.0))
        p = c + (rng.uniform(-0.5, 0.5, 3) * s).astype(f)
        k = int(rng.integers(0, 3))
        p[k] = c[k] + f(rng.choice([-0.5, 0.5])) * s                  # on a face; sometimes on an edge too
        if rng.random() < 0.2:
            p[(k + 1) % 3] = c[(k + 1) % 3] + f(0.4995) * s
        assert same_bits(box_normal(c, s, p.astype(f)), orc.box_normal(c, float(s), p.astype(f)))
        p1, p2, p3 = vec(rng), vec(rng), vec(rng)
        a, beta, m1, m2 = simple_triangle(p1, p2, p3)
        t = orc.make_triangle(p1, p2, p3)
        assert same_bits(a, t["a"]) and same_bits(beta, t["beta"]) and same_bits(m1, t["m1"][:3]) and same_bits(m2, t["m2"][:3])
        assert same_bits(p1, t["position"][:3])


def test_hemisphere_sampler(literal):
    orc = literal
    rng = np.random.default_rng(6)
    worst = 0.0
    for i in range(3000):
        n = normalize(vec(rng))
        if i % 10 == 0:
            n = np.array([f(0.003), f(-0.004), f(1)], dtype=f)         # the branch of RayTracing.glsl:72-73
            n = normalize(n)
        u1, u2 = f(rng.integers(0, 1 << 24)) / f(1 << 24), f(rng.integers(0, 1 << 24)) / f(1 << 24)
        worst = max(worst, float(np.abs(hemisphere(n, u1, u2) - orc.hemisphere_glsl(n, float(u1), float(u2))).max()))
    assert worst <= 2e-7


def cube_hit(R, c, o, d):                        # Cube::hit, src/cpu_version/scene.h:185-204
    oc = o - c
    ot = np.array([dot(R[i], oc) for i in range(3)], dtype=f)
    dt = np.array([dot(R[i], d) for i in range(3)], dtype=f)
    sg = np.array([sign(x) for x in ot], dtype=f)
    with np.errstate(all="ignore"):
        t = (f(0.5) - ot * sg) / (sg * dt)
        for k in range(3):
            p = ot + dt * t[k]
            if max(abs(p[0]), abs(p[1]), abs(p[2])) == f(0.5):       # :193: exactly on the surface, first axis that is
                return t[k], p
    return None


def test_cube_of_the_cpu_tracer_bit_for_bit(literal, rta):
    """The Cube accepts a point only when its infinity norm is exactly 0.5, so every rounding is part of the result:
    which rays hit, at what distance, and the point in the cube's frame that its normal is taken from."""
    from ray_tracing_amd import host
    orc = literal
    rng = np.random.default_rng(1)
    hits = 0
    for i in range(3000):
        cube = host.Cube(tuple(rng.uniform(-2, 2, 3)), host.Lambertian((0.5, 0.5, 0.5)))
        if i % 3:
            cube.rotate(float(rng.uniform(-1, 1)))                    # scene.h:206-208
        m = cube.rec["transformation"].reshape(4, 4).T
        rot, c = m[:3, :3].astype(f), m[:3, 3].astype(f)
        o = rng.uniform(-4, 4, 3).astype(f)
        d = c + rng.uniform(-0.4, 0.4, 3).astype(f) - o
        d = (d / np.linalg.norm(d)).astype(f)
        a, b = cube_hit(rot, c, o, d), orc.hit_cube(cube.rec["transformation"], o, d)
        assert (a is None) == (b is None)
        if a is not None:
            hits += 1
            assert same_bits(a[0], b[0]) and same_bits(a[1], b[1])
    assert hits > 1000
