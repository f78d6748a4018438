/*
 * rt_oracle.c — CPU restatement of the reference path tracer (TEST ORACLE).
 *
 * Test infrastructure only; see rt_oracle.h.  PARITY UNPINNED BY THE
 * REFERENCE (it holds no golden vectors and cannot be built or seeded here).
 *
 * Written as scalar C that keeps the reference's structure (a scene loop, a
 * scatter per material, a recursive ray_color for the CPU tracer).  Build with
 * -ffp-contract=off: every fused multiply-add in the arithmetic below is an
 * explicit fmaf(), everything else is a separately rounded IEEE-754 binary32
 * operation, so a second implementation can match it bit for bit.  No libm
 * function other than sqrtf/fmaf/fabsf (all exactly rounded) feeds geometry;
 * the ORC_UNIT_GAUSS sampler (src/utils.h:45 as written: three normal deviates, normalised) has a recipe
 * of its own for the logarithm (orc_log_draw) and uses libm only in the literal arithmetic.
 *
 * Arithmetic conventions used throughout (ORC_ARITH_RECIPE; ORC_ARITH_LITERAL, further down, drops them):
 *   dot(a,b)      = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
 *   madd(t,d,o)   = fmaf(t,d,o) per component               (o + t*d)
 *   normalize(v)  = v * rsqrt_nr(dot(v,v))      (rsqrt_nr: fixed Newton recipe, below)
 */
#include "rt_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

/* Two arithmetics.  ORC_ARITH_RECIPE (default) is the one the HIP kernels repeat bit for bit: explicit fused
 * multiply-adds, a * rcp_nr(b) for the shader's divisions, v * rsqrt_nr(v.v) for normalize, a polynomial for
 * sin/cos.  ORC_ARITH_LITERAL evaluates the reference's source text the way a plain C++ / GLSL compiler without
 * contraction does: every product and sum rounded on its own (no FMA), IEEE a / b
 * (RayTracing.glsl:145,161,181), v / sqrt(v.v) for every normalize (scene.h:15,131; glsl:73,75,83,135,162,297),
 * libm sinf / cosf (glsl:48,69-70), and Sphere::hit's rec.t = -b - sqrt(discriminant) in double (HitRecord::t
 * is a double and the unqualified sqrt of scene.h:123 is ::sqrt(double)).  It exists to measure how far the
 * recipes are from the text; nothing is compared bit for bit with it.  The flag belongs to the calling thread. */
static __thread int t_lit __attribute__((tls_model("initial-exec"))) = 0;
/* ORC_ARITH_LITERAL_GEOMETRY: the literal arithmetic for everything but the samplers, which keep their recipe (the
 * polynomial sine and cosine, the fused z and r of the sphere point) — what the HIP kernels compute under
 * RT_FLAG_LITERAL_GEOMETRY, bit for bit. */
static __thread int t_poly __attribute__((tls_model("initial-exec"))) = 0;
static inline void set_arith(uint32_t arith) { t_lit = arith != ORC_ARITH_RECIPE; t_poly = arith == ORC_ARITH_LITERAL_GEOMETRY; }
void orc_set_arith(int arith) { set_arith((uint32_t)arith); }
/* a*b + c: fused in the recipe arithmetic, two roundings in the literal one */
static inline float lfma(float a, float b, float c) { return t_lit ? a * b + c : fmaf(a, b, c); }

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return lfma(a.z, b.z, lfma(a.y, b.y, a.x * b.x)); }
static inline v3 madd(float t, v3 d, v3 o) { return V(lfma(t, d.x, o.x), lfma(t, d.y, o.y), lfma(t, d.z, o.z)); }
/* 1/sqrt(q) by a fixed recipe: bit-trick first guess and three Newton steps
 * y <- y * (1.5 - (0.5 q y) y).  Relative error <= 1.3e-7 (about one ulp) for
 * normal q > 0; q = 0 gives a finite value (so normalize(0) = 0), negative q
 * is not used.  Every step is an IEEE operation, so it is the same function on
 * every machine — which `1.0f / sqrtf(q)` is too, at three times the cost on
 * the GPU.  It stands in for Eigen's normalized() (scene.h:15,131) and GLSL's
 * normalize() (RayTracing.glsl:73,75,83,135,162,297), whose own precision is
 * implementation-defined. */
static inline float rsqrt_nr(float q)
{
    uint32_t bits; memcpy(&bits, &q, 4);
    bits = 0x5f3759dfu - (bits >> 1);
    float y; memcpy(&y, &bits, 4);
    const float h = 0.5f * q;
    for (int k = 0; k < 3; ++k) {
        float t = h * y;
        float e = fmaf(-t, y, 1.5f);
        y = y * e;
    }
    return y;
}
static inline v3 normalize(v3 v)
{
    if (t_lit) { float len = sqrtf(dot(v, v)); return V(v.x / len, v.y / len, v.z / len); }     /* Eigen normalized(), GLSL normalize() */
    return vscale(v, rsqrt_nr(dot(v, v)));
}
/* 1/x by a fixed recipe: first guess by subtracting the bit pattern from a
 * constant, then three Newton steps y <- y * (2 - x y), each an IEEE
 * operation; relative error <= 2e-7 for normal x of either sign, x = 0 gives
 * +-inf.  a / b in the routines taken from the shader (hitTriangle, hitBox,
 * boxNormalAt) is a * rcp_nr(b): GLSL specifies its own division only to
 * 2.5 ulp, and a correctly rounded divide costs the GPU five times as much. */
static inline float rcp_nr(float x)
{
    uint32_t bits; memcpy(&bits, &x, 4);
    bits = 0x7ef311c7u - bits;
    float y; memcpy(&y, &bits, 4);
    for (int k = 0; k < 3; ++k) {
        float e = fmaf(-x, y, 2.0f);
        y = y * e;
    }
    return y;
}
static inline float div_nr(float a, float b) { return t_lit ? a / b : a * rcp_nr(b); }
static inline v3 cross(v3 a, v3 b)
{
    return V(lfma(a.y, b.z, -(a.z * b.y)), lfma(a.z, b.x, -(a.x * b.z)), lfma(a.x, b.y, -(a.y * b.x)));
}
/* GLSL sign(): -1, 0 or 1 (NaN -> 0). */
static inline float fsign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
/* GLSL max(x,y) = (x < y) ? y : x. */
static inline float gmax(float x, float y) { return (x < y) ? y : x; }

/* --------------------------------------------------------------------- RNG
 * The reference's streams cannot be reproduced (utils.h:8-9: mt19937 seeded
 * from random_device and shared by all threads; RayTracing.glsl:8-9,44-51:
 * a sin-hash of wall-clock time).  What is kept is their structure: one
 * stream per (seed, pixel, sample), indexed by draw number.  The path has the key
 * hash32(seed_key ^ pixel) + sample * 0x85EBCA6B; its draws come in pairs (every event of a path
 * needs two) from one hashed word per pair: x = hash32(key + (i / 2) * 0x9E3779B9) gives draw i,
 * i even, from its top 24 bits and draw i + 1 from the top 24 bits of x * 0xADB4A92D.  The two
 * steps differ so that no (sample, pair) meets another one's hash input: s * 0x85EBCA6B +
 * p * 0x9E3779B9 repeats only for sample numbers 17 million apart (pairs up to 128 apart). */
uint32_t orc_hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

static uint32_t seed_key(uint64_t seed)
{
    return orc_hash32((uint32_t)seed ^ orc_hash32((uint32_t)(seed >> 32) ^ 0x85ebca6bu));
}

uint32_t orc_path_key(uint64_t seed, uint32_t pixel_index, uint32_t sample)
{
    return orc_hash32(seed_key(seed) ^ pixel_index) + sample * 0x85ebca6bu;
}

/* Uniform in [0,1) with 24 bits, like random() of RayTracing.glsl:44-51. */
float orc_draw(uint32_t key, uint32_t index)
{
    uint32_t x = orc_hash32(key + (index >> 1) * 0x9e3779b9u);
    if (index & 1u) x *= 0xadb4a92du;
    return (float)(x >> 8) * 0x1p-24f;
}

/* random() of RayTracing.glsl:44-51, the shader's own stream: the state is the vec3 `seed`
 * (RayTracing.glsl:9: time + global invocation id / number of work groups), every draw adds
 * (1.78, 3.1, 50.78) to it and returns fract(sin(dot(seed, (12.9898, 78.233, 51.112))) *
 * 43758.5453123).  The sine is taken of the angle reduced to one turn, by the same polynomial
 * as everywhere else here; a GL driver's own sine of arguments in the thousands is no better
 * defined.  Draws are consumed in program order, so the state travels with the path. */
float orc_glsl_random(float seed[3])
{
    seed[0] = seed[0] + 1.78f; seed[1] = seed[1] + 3.1f; seed[2] = seed[2] + 50.78f;      /* :47 */
    float x = lfma(seed[2], 51.112f, lfma(seed[1], 78.233f, seed[0] * 12.9898f));          /* dot  :48 */
    if (t_lit) { float v = sinf(x) * 43758.5453123f; return v - floorf(v); }               /* :48 with libm's sine */
    float t = x * 0x1.45f306p-3f;                                                          /* x / (2 pi) */
    float u = t - floorf(t);
    float c, sn;
    orc_sincos2pi(u, &c, &sn);
    float v = sn * 43758.5453123f;
    return v - floorf(v);                                                                  /* fract */
}

/* The state of one path's stream: the counter key of the default generator, or the vec3
 * seed of the shader's. */
typedef struct { uint32_t key; uint32_t glsl; float s[3]; } pathrng;

static inline float rdraw(pathrng* g, uint32_t index)
{
    return g->glsl ? orc_glsl_random(g->s) : orc_draw(g->key, index);
}

/* RayTracing.glsl:9 for sample `sample` of pixel (gx, gy): the frame is sample / 100, the
 * local invocation (sample % 10, sample / 10 % 10) of the pixel's 10 x 10 work group
 * (RayTracing.glsl:2,6); `time` (main.cpp:139-141: milliseconds / 10000) is (seed + frame)
 * / 10000. */
void orc_glsl_seed(uint64_t seed, uint32_t width, uint32_t height, uint32_t gx, uint32_t gy, uint32_t sample, float out[3])
{
    uint32_t frame = sample / 100u, lx = sample % 10u, ly = (sample / 10u) % 10u;
    float time = (float)(uint32_t)(seed + frame) * 0.0001f;
    float inv_w = 1.0f / (float)width, inv_h = 1.0f / (float)height;
    out[0] = time + (float)(10u * gx + lx) * inv_w;
    out[1] = time + (float)(10u * gy + ly) * inv_h;
    out[2] = time;
}

/* Draw numbering inside one path: 0,1 = sub-pixel jitter (RayTracing.glsl:296,
 * first draw goes with `right`, second with `up`); bounce b owns draws
 * 2+4b .. 5+4b. */
#define DRAW_JITTER 0u
#define DRAW_BOUNCE(b) (2u + 4u * (uint32_t)(b))

/* cos(2 pi u), sin(2 pi u) for u in [0,1) from fmaf/mul only, so that it is
 * the same function on every machine, and without a quadrant case split: the
 * half angle h = pi*u - pi/2 lies in [-pi/2, pi/2), where one odd and one even
 * Taylor polynomial give sin h and cos h; with theta = 2h = 2 pi u - pi,
 * cos(2 pi u) = -cos(theta) = sin^2 h - cos^2 h and sin(2 pi u) = -sin(theta)
 * = -2 sin h cos h.  Absolute error <= 6e-7.  Stands in for the cos/sin of
 * RayTracing.glsl:69-70 (a GL driver's own approximations) and for the angular
 * part of utils.h:45. */
void orc_sincos2pi(float u, float* c_out, float* s_out)
{
    if (t_lit && !t_poly) {                                /* angle = random() * 2 * pi; cos(angle), sin(angle)   glsl:66,69-70 */
        float angle = (u * 2.0f) * 3.14159265358979323846f;
        *c_out = cosf(angle); *s_out = sinf(angle);
        return;
    }
    float x = fmaf(u, 0x1.921fb6p+1f, -0x1.921fb6p+0f);   /* pi*u - pi/2 */
    float x2 = x * x;
    float ps = -0x1.ae6456p-26f;
    ps = fmaf(ps, x2, 0x1.71de3ap-19f);
    ps = fmaf(ps, x2, -0x1.a01a02p-13f);
    ps = fmaf(ps, x2, 0x1.111112p-7f);
    ps = fmaf(ps, x2, -0x1.555556p-3f);
    ps = fmaf(ps, x2, 1.0f);
    float s1 = x * ps;
    float pc = 0x1.1eed8ep-29f;
    pc = fmaf(pc, x2, -0x1.27e4fcp-22f);
    pc = fmaf(pc, x2, 0x1.a01a02p-16f);
    pc = fmaf(pc, x2, -0x1.6c16c2p-10f);
    pc = fmaf(pc, x2, 0x1.555556p-5f);
    pc = fmaf(pc, x2, -0.5f);
    float c1 = fmaf(pc, x2, 1.0f);
    *c_out = fmaf(s1, s1, -(c1 * c1));
    *s_out = (s1 * -2.0f) * c1;
}

/* randomUnitVector() of utils.h:37-46: a normalised vector of three normal
 * deviates, i.e. a point uniform on the unit sphere.  Sampled directly:
 * z uniform in (-1,1], azimuth uniform. */
void orc_unit_vector(float u1, float u2, float out[3])
{
    const int fused = !t_lit || t_poly;                    /* the sampler's own recipe */
    float z = fused ? fmaf(-2.0f, u1, 1.0f) : -2.0f * u1 + 1.0f;
    float r = sqrtf(fused ? fmaf(-z, z, 1.0f) : -z * z + 1.0f);
    float c, s;
    orc_sincos2pi(u2, &c, &s);
    out[0] = r * c; out[1] = r * s; out[2] = z;
}

/* utils.h:45 literally: three N(0,1) deviates, normalised.  The deviates come
 * from Box-Muller on four draws, the construction the reference itself writes
 * in RayTracing.glsl:53-62.  libm; statistical cross-check only. */
static v3 unit_vector_gauss(pathrng* g, uint32_t base)
{
    float u1 = rdraw(g, base), u2 = rdraw(g, base + 1);
    float u3 = rdraw(g, base + 2), u4 = rdraw(g, base + 3);
    if (u1 < 0x1p-24f) u1 = 0x1p-24f;
    if (u3 < 0x1p-24f) u3 = 0x1p-24f;
    if (!t_lit) {
        /* the recipe the HIP kernel repeats (RT_FLAG_UNIT_GAUSS): logarithm by orc_log_draw, sine and cosine
         * by the shared polynomial, the exact square root; |.| because the logarithm of a draw next to 1 may
         * come out as +1e-8 */
        float r1 = sqrtf(fabsf(-2.0f * orc_log_draw(u1))), r2 = sqrtf(fabsf(-2.0f * orc_log_draw(u3)));
        float c1, s1, c2, s2;
        orc_sincos2pi(u2, &c1, &s1);
        orc_sincos2pi(u4, &c2, &s2);
        return normalize(V(r1 * c1, r1 * s1, r2 * c2));
    }
    float r1 = sqrtf(-2.0f * logf(u1)), r2 = sqrtf(-2.0f * logf(u3));
    float a1 = 6.283185307179586f * u2, a2 = 6.283185307179586f * u4;
    v3 n = V(r1 * cosf(a1), r1 * sinf(a1), r2 * cosf(a2));
    return normalize(n);
}

/* ln(u) for a draw u = k * 2^-24, k = 1 .. 2^24 - 1, by a fixed recipe of IEEE operations: u = m * 2^e with m
 * folded into [sqrt(1/2), sqrt(2)), ln m = f + f^2 q(f) for f = m - 1 with a degree-7 polynomial q (absolute
 * error 4e-8 on the interval), plus e * ln 2. */
float orc_log_draw(float u)
{
    float raw = u * 0x1p24f;                                  /* the integer the draw is made of; exact */
    uint32_t b; memcpy(&b, &raw, 4);
    int e = (int)(b >> 23) - 127;
    uint32_t mb = (b & 0x007fffffu) | 0x3f800000u;
    float m; memcpy(&m, &mb, 4);
    if (m > 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float q = 0x1.70cfe8p-4f;
    q = fmaf(q, f, -0x1.23ffe8p-3f);
    q = fmaf(q, f, 0x1.2f3c7ep-3f);
    q = fmaf(q, f, -0x1.5374fcp-3f);
    q = fmaf(q, f, 0x1.990c98p-3f);
    q = fmaf(q, f, -0x1.000438p-2f);
    q = fmaf(q, f, 0x1.555630p-2f);
    q = fmaf(q, f, -0x1.fffffcp-2f);
    float lm = fmaf(f * f, q, f);
    return fmaf((float)(e - 24), 0x1.62e430p-1f, lm);
}

/* randomOnHalfUnitShere, RayTracing.glsl:64-80. */
void orc_hemisphere_glsl(const float nn[3], float u1, float u2, float out[3])
{
    v3 n = V(nn[0], nn[1], nn[2]);
    float c, s;
    orc_sincos2pi(u1, &c, &s);                         /* angle = random()*2*pi        :66 */
    float r = u2;                                      /* r = random()                 :67 */
    float dist = sqrtf(lfma(-r, r, 1.0f));             /* sqrt(1 - r*r)                :68 */
    float x = c * dist, y = s * dist;                  /*                              :69-70 */
    v3 b1 = V(-n.y, n.x, 0.0f);                        /*                              :71 */
    float len_xy = sqrtf(lfma(b1.y, b1.y, b1.x * b1.x));
    if (len_xy < 0.01f) b1 = normalize(V(0.0f, -n.z, n.y));   /*                      :72-73 */
    else b1 = normalize(b1);                           /*                              :75 */
    v3 b2 = cross(n, b1);                              /*                              :77 */
    /* b1*x + b2*y + normal*r                                                          :78 */
    out[0] = lfma(n.x, r, lfma(b2.x, y, b1.x * x));
    out[1] = lfma(n.y, r, lfma(b2.y, y, b1.y * x));
    out[2] = lfma(n.z, r, lfma(b2.z, y, b1.z * x));
}

/* ------------------------------------------------------------ intersection */

/* Sphere::hit, scene.h:115-128 == hitSphere, RayTracing.glsl:118-132.
 * Near root only; needs a unit direction.  Returns +inf when there is no
 * real root. */
static double hit_sphere_t(const float cc[3], float radius, const float oo[3], const float dd[3], int t_in_double)
{
    v3 oc = vsub(V(oo[0], oo[1], oo[2]), V(cc[0], cc[1], cc[2]));   /* scene.h:116 */
    v3 d = V(dd[0], dd[1], dd[2]);
    float b = dot(oc, d);                                            /* :117 */
    float r2 = radius * radius;
    float c, disc;
    if (t_lit) {
        c = dot(oc, oc) - r2;                                        /* :118 */
        disc = b * b - c;                                            /* :119 */
    } else {
        c = fmaf(oc.z, oc.z, fmaf(oc.y, oc.y, fmaf(oc.x, oc.x, -r2)));
        disc = fmaf(b, b, -c);
    }
    if (disc < 0.0f) return INFINITY;                                /* :120 */
    if (t_lit && t_in_double) return -(double)b - sqrt((double)disc);   /* :123: rec.t is a double, sqrt is ::sqrt(double) */
    return -b - sqrtf(disc);                                         /* :123 / glsl:130 */
}
float orc_hit_sphere(const float cc[3], float radius, const float oo[3], const float dd[3])
{
    return (float)hit_sphere_t(cc, radius, oo, dd, 0);
}

/* hitBox, RayTracing.glsl:138-156, with a side per axis.  The shader reads one side,
 * transformation[0][0], for all three axes (:139), although Box::Box stores three
 * (src/gpu_version/shape.h:67); orc_hit_box below is that case. */
float orc_hit_box3(const float cc[3], const float sides[3], const float oo[3], const float dd[3])
{
    float o[3], d[3], sg[3], t[3];
    for (int k = 0; k < 3; ++k) {
        float inv = 1.0f / sides[k];                                 /* (1/scale)   :141 */
        o[k] = inv * (oo[k] - cc[k]);                                /* :141 */
        d[k] = inv * dd[k];                                          /* :142 */
        sg[k] = fsign(o[k]);                                         /* :144 */
        t[k] = div_nr(0.5f - o[k] * sg[k], sg[k] * d[k]);            /* :145 */
    }
    float best = INFINITY;                                           /* :146 */
    for (int k = 0; k < 3; ++k) {                                    /* :147 */
        float px = fabsf(lfma(d[0], t[k], o[0]));                    /* :148-149 */
        float py = fabsf(lfma(d[1], t[k], o[1]));
        float pz = fabsf(lfma(d[2], t[k], o[2]));
        float inf_norm = gmax(px, gmax(py, pz));                     /* :150 */
        if (inf_norm < 0.5001f && inf_norm > 0.4999f && best > t[k]) /* :151 */
            best = t[k];
    }
    return best;
}
float orc_hit_box(const float cc[3], float scale, const float oo[3], const float dd[3])
{
    const float sides[3] = { scale, scale, scale };
    return orc_hit_box3(cc, sides, oo, dd);
}

/* boxNormalAt, RayTracing.glsl:158-163, with a side per axis (see orc_hit_box3). */
void orc_box_normal3(const float cc[3], const float sides[3], const float pp[3], float nn[3])
{
    float q[3];
    for (int k = 0; k < 3; ++k) {
        float p = div_nr(pp[k] - cc[k], sides[k]);                   /* :161 */
        float st = (fabsf(p) < 0.499f) ? 0.0f : 1.0f;                /* step(0.499,|p|) :162 */
        q[k] = st * fsign(p);
    }
    v3 n = normalize(V(q[0], q[1], q[2]));
    nn[0] = n.x; nn[1] = n.y; nn[2] = n.z;
}
void orc_box_normal(const float cc[3], float scale, const float pp[3], float nn[3])
{
    const float sides[3] = { scale, scale, scale };
    orc_box_normal3(cc, sides, pp, nn);
}

/* Cube::hit, src/cpu_version/scene.h:185-204: the CPU tracer's unit cube.  m is the
 * shape's 4x4 transformation (column-major): rotation in the upper-left 3x3, centre in
 * column 3.  As written there: the three slab distances are tried in axis order and the
 * FIRST whose point lies exactly on the cube's surface (infinity norm == 0.5) is the hit,
 * whether or not it is the nearest; divisions are IEEE.  The exact comparison makes the
 * rounding of `originTransformed + directionTransformed * t` part of the result, so this
 * routine never fuses: every product and sum is rounded on its own, in Eigen's order, as a
 * C++ compiler without contraction evaluates :186-192 (with a fused multiply-add the point
 * lands exactly on 0.5 far less often and most of the cube disappears).  Returns 1 and
 * fills t and the point in the cube's frame, else 0. */
int orc_hit_cube(const float m[16], const float oo[3], const float dd[3], float* t_out, float in_point[3])
{
    float oc[3] = { oo[0] - m[12], oo[1] - m[13], oo[2] - m[14] };
    float o[3], d[3], t[3];
    for (int i = 0; i < 3; ++i) {                                      /* rotation * (orig - center), rotation * dir   :186-187 */
        o[i] = (m[i] * oc[0] + m[4 + i] * oc[1]) + m[8 + i] * oc[2];
        d[i] = (m[i] * dd[0] + m[4 + i] * dd[1]) + m[8 + i] * dd[2];
    }
    for (int i = 0; i < 3; ++i) {
        float sg = fsign(o[i]);                                        /* cwiseSign                                    :188 */
        t[i] = (0.5f - o[i] * sg) / (sg * d[i]);                       /*                                              :190 */
    }
    for (int k = 0; k < 3; ++k) {                                      /*                                              :191 */
        float px = o[0] + d[0] * t[k], py = o[1] + d[1] * t[k], pz = o[2] + d[2] * t[k];               /* :192 */
        if (gmax(fabsf(px), gmax(fabsf(py), fabsf(pz))) == 0.5f) {     /*                                              :193 */
            *t_out = t[k];
            in_point[0] = px; in_point[1] = py; in_point[2] = pz;
            return 1;
        }
    }
    return 0;
}

/* rotation^T * normalAt(inPoint), scene.h:150-164,196.  normalAt takes the sign from the
 * maximum of the ABSOLUTE values, so it is always +1: the normal is the positive axis of
 * the face's direction, on either side of the cube (as written).  rotation^T * e_i is row
 * i of the rotation. */
void orc_cube_normal(const float m[16], const float in_point[3], float nn[3])
{
    float ax = fabsf(in_point[0]), ay = fabsf(in_point[1]), az = fabsf(in_point[2]);
    int i = 0; float best = ax;                                        /* maxCoeff(&index): first maximum */
    if (ay > best) { best = ay; i = 1; }
    if (az > best) { best = az; i = 2; }
    nn[0] = m[i]; nn[1] = m[4 + i]; nn[2] = m[8 + i];
}

/* hitTriangle, RayTracing.glsl:179-194. */
float orc_hit_triangle(const orc_triangle* tr, const float oo[3], const float dd[3])
{
    v3 a = V(tr->a[0], tr->a[1], tr->a[2]);
    v3 o = V(oo[0], oo[1], oo[2]), d = V(dd[0], dd[1], dd[2]);
    float k = div_nr(tr->beta - dot(a, o), dot(a, d));               /* :181 */
    if (k < 0.0f) return INFINITY;                                   /* :182-184 */
    v3 q = vsub(madd(k, d, o), V(tr->position[0], tr->position[1], tr->position[2])); /* :186 */
    float x = dot(q, V(tr->m1[0], tr->m1[1], tr->m1[2]));            /* :187 */
    float y = dot(q, V(tr->m2[0], tr->m2[1], tr->m2[2]));            /* :188 */
    if (x > 0.0f && y > 0.0f && x + y < 1.0f) return k;              /* :190-192 */
    return INFINITY;
}

/* ------------------------------------------------------------------ scene */

typedef struct {
    const orc_shape* shapes; uint32_t n_shapes;
    const orc_triangle* tris; uint32_t n_tris;
    const float* cam;
    const orc_params* p;
    uint32_t seed_k;
    orc_stats st;                 /* per-thread */
    int32_t* trace_hit; uint32_t* trace_mat;   /* per-path slices or NULL */
} tracer;

typedef struct { v3 o, d; } ray;

static inline v3 shape_center(const orc_shape* s) { return V(s->m[12], s->m[13], s->m[14]); }

/* The sides of a Box: the shader's single side transformation[0][0] (RayTracing.glsl:139,159),
 * or — ORC_FLAG_BOX_SIDES — the three sides Box::Box stores on the diagonal (shape.h:67). */
static inline void box_sides(const orc_params* p, const orc_shape* s, float sides[3])
{
    const int own = (p->flags & ORC_FLAG_BOX_SIDES) != 0;
    sides[0] = s->m[0]; sides[1] = own ? s->m[5] : s->m[0]; sides[2] = own ? s->m[10] : s->m[0];
}

/* Distances travel as doubles: the CPU tracer's HitRecord::t is one (scene.h:34).  In the recipe arithmetic and
 * under the shader's rules every value is a float widened, which compares exactly as the float does. */
static double hit_shape(tracer* T, const orc_shape* s, const ray* r)
{
    float c[3] = { s->m[12], s->m[13], s->m[14] };
    float o[3] = { r->o.x, r->o.y, r->o.z }, d[3] = { r->d.x, r->d.y, r->d.z };
    if (s->type == 0) { T->st.sphere_tests++; return hit_sphere_t(c, s->m[0], o, d, T->p->semantics == ORC_SEM_CPU); }
    if (s->type == 1) {
        float sides[3]; box_sides(T->p, s, sides);
        T->st.box_tests++;
        return orc_hit_box3(c, sides, o, d);
    }
    if (s->type == 3 && T->p->semantics == ORC_SEM_CPU) {             /* the CPU tracer's Cube */
        float t, ip[3];
        return orc_hit_cube(s->m, o, d, &t, ip) ? t : INFINITY;
    }
    return -1.0f;                                                     /* RayTracing.glsl:176 */
}

/* The acceptance window of a hit distance.  CPU: scene.h:237 with the
 * arguments of rayTracer.h:117 — `t_min < temp.t` compares the double 0.001
 * with a widened float, which holds exactly when t >= 0.001f.  GLSL:
 * RayTracing.glsl:212,226. */
static inline int in_window(const orc_params* p, double t)
{
    if (p->semantics == ORC_SEM_CPU) return (t_lit ? 0.001 < t : t >= (double)0.001f) && t < (double)FLT_MAX;
    return t < 1000.0 && t > (double)0.0001f;
}

/* Scene::hit (scene.h:232-255) / getClosestShapeIndex + getClosestTriangleIndex
 * (RayTracing.glsl:206-231): linear scan, strict <, first index wins ties;
 * a triangle replaces a shape only when strictly closer (RayTracing.glsl:271). */
static int closest_hit(tracer* T, const ray* r, float* t_out)
{
    double best = INFINITY; int index = -1;
    for (uint32_t i = 0; i < T->n_shapes; ++i) {
        double t = hit_shape(T, &T->shapes[i], r);
        T->st.shape_tests++;
        if (in_window(T->p, t) && t < best) { best = t; index = (int)i; }
    }
    double best_tri = INFINITY; int index_tri = -1;
    float o[3] = { r->o.x, r->o.y, r->o.z }, d[3] = { r->d.x, r->d.y, r->d.z };
    for (uint32_t i = 0; i < T->n_tris; ++i) {
        float t = orc_hit_triangle(&T->tris[i], o, d);
        T->st.triangle_tests++;
        if (in_window(T->p, t) && t < best_tri) { best_tri = t; index_tri = (int)i; }
    }
    if (index < 0 && index_tri < 0) return -1;                        /* RayTracing.glsl:267 */
    if (best_tri < best) { *t_out = (float)best_tri; return (int)T->n_shapes + index_tri; }  /* :271 */
    *t_out = (float)best;                                             /* Ray::at(double) multiplies a float vector: narrowed there */
    return index;
}

static v3 shape_normal(const orc_params* prm, const orc_shape* s, v3 p)
{
    if (s->type == 0) {                                               /* scene.h:130-132, glsl:134-136 */
        v3 v = vsub(p, shape_center(s));
        if (t_lit) return normalize(v);
        /* A fuzzy metal's direction left as written (scene.h:78) is not unit, and Sphere::hit, which assumes it is
         * (scene.h:115-123), then returns a point off the sphere: |v| is not the radius there. */
        if (prm->semantics == ORC_SEM_CPU && (prm->flags & ORC_FLAG_METAL_UNNORMALIZED)) return normalize(v);
        /* The recipe: |v| is the radius to the rounding of the hit distance, so 1/|radius| is a first guess of
         * 1/sqrt(v.v) that one Newton step y <- y (1.5 - (0.5 q y) y) takes to rounding level.  The step's factor
         * e = 1 - d (d: the guess's relative error) tells when that fails — a tiny sphere far away, whose hit
         * point is off the sphere by more than 2^-12 of the radius: the full normalisation then. */
        float inv_r = 1.0f / fabsf(s->m[0]);
        float q = dot(v, v);
        float t = (0.5f * q) * inv_r;
        float e = fmaf(-t, inv_r, 1.5f);
        if (!(fabsf(e - 1.0f) <= 0x1p-12f)) return normalize(v);
        return vscale(v, inv_r * e);
    }
    if (s->type == 1) {
        float c[3] = { s->m[12], s->m[13], s->m[14] }, pp[3] = { p.x, p.y, p.z }, n[3], sides[3];
        box_sides(prm, s, sides);
        orc_box_normal3(c, sides, pp, n);
        return V(n[0], n[1], n[2]);
    }
    return V(1.0f, 1.0f, 1.0f);                                       /* glsl:203 */
}

static v3 unit_vector(tracer* T, pathrng* g, uint32_t base)
{
    if (T->p->unit_sampler == ORC_UNIT_GAUSS || (T->p->flags & ORC_FLAG_UNIT_GAUSS)) return unit_vector_gauss(g, base);
    float u[3];
    float u1 = rdraw(g, base), u2 = rdraw(g, base + 1);
    orc_unit_vector(u1, u2, u);
    return V(u[0], u[1], u[2]);
}

/* Dielectric (material type 3; no counterpart in the reference, whose Metal::data[0] =
 * "refraction" is never read): the ray is mirrored with Schlick's probability, refracted
 * otherwise.  The sphere test knows only the near root, so a refracted ray leaves the
 * sphere without a second event.  ior = material.data[0]. */
static v3 glass_direction(v3 n, v3 d, float ior, float u1)
{
    float eta = 1.0f / ior;
    float tt = (1.0f - ior) / (1.0f + ior);
    float r0 = tt * tt;
    float cosi = -dot(d, n);
    float x = 1.0f - cosi, x2 = x * x;
    float refl = fmaf(1.0f - r0, (x2 * x2) * x, r0);
    v3 perp = vscale(madd(cosi, n, d), eta);
    float par = -sqrtf(fabsf(1.0f - dot(perp, perp)));
    v3 refr = madd(par, n, perp);
    v3 mirr = madd(-(2.0f * dot(d, n)), n, d);
    return (u1 < refl) ? mirr : refr;
}

/* Albedo of triangle i: one colour for all (RayTracing.glsl:275 hard-codes red), or the
 * caller's colour per triangle. */
static inline v3 triangle_color(const tracer* T, int i)
{
    const float* c = T->p->triangle_colors ? T->p->triangle_colors + 3 * (size_t)i : T->p->triangle_albedo;
    return V(c[0], c[1], c[2]);
}

void orc_glass_direction(const float n[3], const float d[3], float ior, float u1, float out[3])
{
    v3 r = glass_direction(V(n[0], n[1], n[2]), V(d[0], d[1], d[2]), ior, u1);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* Outcome of the surface interaction at one hit. */
enum { EV_SCATTER = 0, EV_EMIT = 1 };

/* One hit under the CPU tracer's rules: Material::scatter / emit,
 * scene.h:39-109.  On EV_SCATTER `att` is the attenuation and *r the new ray;
 * on EV_EMIT `att` is what the path returns at this hit. */
static int interact_cpu(tracer* T, int index, float t, ray* r, pathrng* key, int bounce, v3* att, uint32_t* mat_id)
{
    v3 p = madd(t, r->d, r->o);                                        /* Ray::at, scene.h:21-23 */
    v3 n; v3 albedo; uint32_t type; float fuzz = 0.0f;
    if (index >= (int)T->n_shapes) {
        const orc_triangle* tr = &T->tris[index - (int)T->n_shapes];
        v3 a = V(tr->a[0], tr->a[1], tr->a[2]);
        n = vscale(a, fsign(-dot(r->d, a)));                           /* glsl:165-167 */
        albedo = triangle_color(T, index - (int)T->n_shapes);
        type = 1;
    } else {
        const orc_shape* s = &T->shapes[index];
        if (s->type == 3) {
            float o[3] = { r->o.x, r->o.y, r->o.z }, d[3] = { r->d.x, r->d.y, r->d.z }, tt, ip[3], nn[3];
            orc_hit_cube(s->m, o, d, &tt, ip);
            orc_cube_normal(s->m, ip, nn);
            n = V(nn[0], nn[1], nn[2]);
        } else n = shape_normal(T->p, s, p);
        albedo = V(s->mat.color[0], s->mat.color[1], s->mat.color[2]);
        type = s->mat.type; fuzz = s->mat.data[1];
    }
    *mat_id = type;
    if (type == 1) {                                                   /* Lambertian::scatter, scene.h:51-58 */
        v3 u = unit_vector(T, key, DRAW_BOUNCE(bounce));
        v3 dir = madd(1.0001f, n, u);                                  /* utils.h:48-50 */
        r->o = p; r->d = normalize(dir);                               /* Ray ctor, scene.h:14-16 */
        *att = vscale(albedo, 0.6f);                                   /* scene.h:56 */
        T->st.diffuse++;
        return EV_SCATTER;
    }
    if (type == 2) {                                                   /* Metal::scatter, scene.h:73-81 */
        float k = 2.0f * dot(n, r->d);                                 /* utils.h:62-64 */
        v3 dir = madd(-k, n, r->d);
        if (fuzz != 0.0f) dir = madd(fuzz, unit_vector(T, key, DRAW_BOUNCE(bounce)), dir);  /* scene.h:78 */
        T->st.specular++;
        if (!(dot(dir, n) > 0.0f)) { *att = V(0, 0, 0); return EV_EMIT; }  /* scene.h:80,83-85 */
        r->o = p;
        r->d = (T->p->flags & ORC_FLAG_METAL_UNNORMALIZED) ? dir : normalize(dir);
        *att = albedo;                                                 /* scene.h:79 */
        return EV_SCATTER;
    }
    if (type == 3) {
        float u1 = rdraw(key, DRAW_BOUNCE(bounce)), u2 = rdraw(key, DRAW_BOUNCE(bounce) + 1); (void)u2;
        float ior = (index >= (int)T->n_shapes) ? 1.0f : T->shapes[index].mat.data[0];
        v3 dir = glass_direction(n, r->d, ior, u1);
        T->st.specular++;
        r->o = p; r->d = normalize(dir);
        *att = albedo;
        return EV_SCATTER;
    }
    if (type == 0) { *att = albedo; return EV_EMIT; }                  /* LightSource, scene.h:97-105 */
    if (type == 100) { *att = n; return EV_EMIT; }                     /* DebugMaterial, glsl:248-252 */
    *att = V(0, 0, 0);
    return EV_EMIT;
}

/* ray_color, rayTracer.h:110-132, as written: recursive; the products are
 * formed on the way back out. */
static v3 ray_color_cpu_recursive(tracer* T, ray r, int depth, pathrng* key, int bounce)
{
    if (depth < 0) { T->st.terminal++; return V(0, 0, 0); }            /* :113-115 */
    float t; T->st.segments++;
    int index = closest_hit(T, &r, &t);                                /* :117 */
    if (T->trace_hit) T->trace_hit[bounce] = index;
    if (index < 0) { T->st.terminal++; return V(0, 0, 0); }            /* :131 */
    v3 att; uint32_t mat;
    int ev = interact_cpu(T, index, t, &r, key, bounce, &att, &mat);
    if (T->trace_mat) T->trace_mat[bounce] = mat;
    if (ev == EV_SCATTER)
        return vmul(att, ray_color_cpu_recursive(T, r, depth - 1, key, bounce + 1)); /* :121 */
    T->st.terminal++;
    return att;                                                        /* :124 */
}

/* The same paths with the products formed camera-outwards. */
static v3 ray_color_cpu_forward(tracer* T, ray r, int max_depth, pathrng* key)
{
    v3 thr = V(1, 1, 1);
    for (int bounce = 0; bounce < max_depth; ++bounce) {
        float t; T->st.segments++;
        int index = closest_hit(T, &r, &t);
        if (T->trace_hit) T->trace_hit[bounce] = index;
        if (index < 0) { T->st.terminal++; return V(0, 0, 0); }
        v3 att; uint32_t mat;
        int ev = interact_cpu(T, index, t, &r, key, bounce, &att, &mat);
        if (T->trace_mat) T->trace_mat[bounce] = mat;
        if (ev == EV_EMIT) {
            T->st.terminal++;
            return vmul(thr, att);
        }
        thr = vmul(thr, att);
    }
    T->st.terminal++;
    return V(0, 0, 0);
}

/* ray_color + scatter, RayTracing.glsl:233-288. */
static v3 ray_color_glsl(tracer* T, ray r, int max_depth, pathrng* key)
{
    v3 result = V(1, 1, 1);                                            /* :258 */
    for (int bounce = 0; bounce < max_depth; ++bounce) {               /* :263 */
        float t; T->st.segments++;
        int index = closest_hit(T, &r, &t);                            /* :264-265 */
        if (T->trace_hit) T->trace_hit[bounce] = index;
        if (index < 0) { T->st.terminal++; return V(0, 0, 0); }        /* :267-269 */
        if (index >= (int)T->n_shapes) {                               /* :271-277 */
            const orc_triangle* tr = &T->tris[index - (int)T->n_shapes];
            v3 a = V(tr->a[0], tr->a[1], tr->a[2]);
            v3 n = vscale(a, fsign(-dot(r.d, a)));                     /* :272 */
            r.o = madd(t, r.d, r.o);                                   /* :273 */
            float u1 = rdraw(key, DRAW_BOUNCE(bounce)), u2 = rdraw(key, DRAW_BOUNCE(bounce) + 1);   /* :66-67, in that order */
            float nn[3] = { n.x, n.y, n.z }, out[3];
            orc_hemisphere_glsl(nn, u1, u2, out);                      /* :274 */
            r.d = V(out[0], out[1], out[2]);
            result = vmul(result, triangle_color(T, index - (int)T->n_shapes));   /* :275 */
            if (T->trace_mat) T->trace_mat[bounce] = 1;
            T->st.diffuse++;
            continue;
        }
        const orc_shape* s = &T->shapes[index];
        v3 color = V(s->mat.color[0], s->mat.color[1], s->mat.color[2]);
        uint32_t type = s->mat.type;
        if (T->trace_mat) T->trace_mat[bounce] = type;
        if (type == 0) { T->st.terminal++; return vmul(result, color); }  /* :278-280 (and :235-237) */
        if (type == 1) {                                               /* :238-242 */
            r.o = madd(t, r.d, r.o);
            v3 n = shape_normal(T->p, s, r.o);
            float u1 = rdraw(key, DRAW_BOUNCE(bounce)), u2 = rdraw(key, DRAW_BOUNCE(bounce) + 1);
            float nn[3] = { n.x, n.y, n.z }, out[3];
            orc_hemisphere_glsl(nn, u1, u2, out);
            r.d = V(out[0], out[1], out[2]);
            result = vmul(result, color);
            T->st.diffuse++;
            continue;
        }
        if (type == 2) {                                               /* :243-247, reflect :82-84 */
            r.o = madd(t, r.d, r.o);
            v3 n = shape_normal(T->p, s, r.o);
            float k = 2.0f * dot(r.d, n);
            r.d = normalize(madd(-k, n, r.d));
            result = vmul(result, color);
            T->st.specular++;
            continue;
        }
        if (type == 3) {                                               /* dielectric extension */
            r.o = madd(t, r.d, r.o);
            v3 n = shape_normal(T->p, s, r.o);
            float u1 = rdraw(key, DRAW_BOUNCE(bounce)), u2 = rdraw(key, DRAW_BOUNCE(bounce) + 1); (void)u2;
            r.d = normalize(glass_direction(n, r.d, s->mat.data[0], u1));
            result = vmul(result, color);
            T->st.specular++;
            continue;
        }
        T->st.terminal++;
        if (type == 100) {                                             /* :248-252 */
            r.o = madd(t, r.d, r.o);
            return shape_normal(T->p, s, r.o);
        }
        return result;                                                 /* :253, :282-284 */
    }
    T->st.terminal++;
    return V(0, 0, 0);                                                 /* :287 */
}

/* -------------------------------------------------------------- camera ray */

static inline v3 cam3(const float* cam, int off) { return V(cam[off], cam[off + 1], cam[off + 2]); }

/* rayTracer.h:77 / RayTracing.glsl:293: upperLeft + step*gx*right - step*gy*up. */
static v3 pixel_corner(const float* cam, uint32_t gx, uint32_t gy)
{
    float step = cam[19];
    v3 ul = cam3(cam, 8), up = cam3(cam, 12), right = cam3(cam, 16);
    float sx = step * (float)gx, sy = step * (float)gy;
    return madd(-sy, up, madd(sx, right, ul));
}

static uint32_t isqrt_u32(uint32_t n) { uint32_t r = (uint32_t)sqrt((double)n); while (r * r > n) --r; while ((r + 1) * (r + 1) <= n) ++r; return r; }

static ray camera_ray(const float* cam, const orc_params* p, uint32_t seed_k, uint32_t gx, uint32_t gy, uint32_t sample, pathrng* g)
{
    float step = cam[19];
    v3 pos = cam3(cam, 0), up = cam3(cam, 12), right = cam3(cam, 16);
    v3 pc = pixel_corner(cam, gx, gy);
    g->key = orc_hash32(seed_k ^ (gy * p->width + gx)) + sample * 0x85ebca6bu;      /* orc_path_key */
    g->glsl = p->rng == ORC_RNG_GLSL;
    if (g->glsl) orc_glsl_seed(p->seed, p->width, p->height, gx, gy, sample, g->s);
    v3 origin;
    if (p->sample_mode == ORC_SAMPLE_JITTER) {
        /* RayTracing.glsl:296 */
        float u1 = rdraw(g, DRAW_JITTER), u2 = rdraw(g, DRAW_JITTER + 1);     /* first draw goes with `right` */
        float s1 = step * u1, s2 = step * u2;
        origin = madd(-s2, up, madd(s1, right, pc));
    } else {
        /* renderPixel<N>, rayTracer.h:92-101 */
        uint32_t N = p->grid_n;
        uint32_t i = (sample / N) % N, j = sample % N;
        float step_size = step / (float)(N + 1);                       /* :92 */
        v3 ups = vscale(up, step_size), rs = vscale(right, step_size);
        origin = vsub(pc, vscale(ups, (float)i));                      /* :97 */
        if (p->sample_mode == ORC_SAMPLE_GRID_COMPAT) {
            for (uint32_t jj = 0; jj <= j; ++jj) {                     /* :98-99, offset accumulates */
                v3 inc = vscale(rs, (float)jj);
                origin = V(origin.x + inc.x, origin.y + inc.y, origin.z + inc.z);
            }
        } else {
            v3 inc = vscale(rs, (float)j);
            origin = V(origin.x + inc.x, origin.y + inc.y, origin.z + inc.z);
        }
    }
    ray r; r.o = origin; r.d = normalize(vsub(origin, pos));           /* rayTracer.h:100, glsl:297 */
    return r;
}

void orc_camera_ray(const float camera[21], const orc_params* p, uint32_t gx, uint32_t gy, uint32_t sample, float origin[3], float dir[3])
{
    pathrng key;
    orc_params q = *p;
    if (q.sample_mode != ORC_SAMPLE_JITTER && q.grid_n == 0) q.grid_n = isqrt_u32(q.first_sample + q.spp);
    const int lit_before = t_lit, poly_before = t_poly;
    set_arith(q.arith);
    ray r = camera_ray(camera, &q, seed_key(p->seed), gx, gy, sample, &key);
    t_lit = lit_before; t_poly = poly_before;
    origin[0] = r.o.x; origin[1] = r.o.y; origin[2] = r.o.z;
    dir[0] = r.d.x; dir[1] = r.d.y; dir[2] = r.d.z;
}

static v3 path_color(tracer* T, uint32_t gx, uint32_t gy, uint32_t sample)
{
    pathrng key;
    ray r = camera_ray(T->cam, T->p, T->seed_k, gx, gy, sample, &key);
    T->st.samples++;
    if (T->p->semantics == ORC_SEM_GLSL) return ray_color_glsl(T, r, (int)T->p->max_depth, &key);
    if (T->p->color_order == ORC_COLOR_FORWARD) return ray_color_cpu_forward(T, r, (int)T->p->max_depth, &key);
    return ray_color_cpu_recursive(T, r, (int)T->p->max_depth - 1, &key, 0);
}

/* uvec4(255 * color), RayTracing.glsl:301.  GLSL leaves negative and
 * out-of-range conversions undefined; here they saturate and NaN gives 0. */
static inline uint32_t quantise(float c)
{
    float v = 255.0f * c;
    if (!(v > 0.0f)) return 0u;
    if (v >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)v;
}

static int row_in_shard(const orc_params* p, uint32_t y)
{
    if (p->shard_count <= 1) return 1;
    uint32_t band = p->band_rows ? p->band_rows : 1;
    return (y / band) % p->shard_count == p->shard_index;
}

static void stats_add(orc_stats* a, const orc_stats* b)
{
    a->samples += b->samples; a->segments += b->segments; a->diffuse += b->diffuse;
    a->specular += b->specular; a->terminal += b->terminal; a->shape_tests += b->shape_tests;
    a->sphere_tests += b->sphere_tests; a->box_tests += b->box_tests; a->triangle_tests += b->triangle_tests;
}

static int fix_params(const orc_params* in, orc_params* q)
{
    *q = *in;
    if (q->width == 0 || q->height == 0 || q->max_depth == 0) return -1;
    if (q->sample_mode != ORC_SAMPLE_JITTER) {
        if (q->grid_n == 0) q->grid_n = isqrt_u32(q->first_sample + q->spp);
        if (q->grid_n == 0) return -1;
    }
    return 0;
}

int orc_render(const orc_shape* shapes, uint32_t n_shapes, const orc_triangle* tris, uint32_t n_tris,
               const float camera[21], const orc_params* pin, uint32_t* accum, orc_stats* stats)
{
    orc_params q;
    if (fix_params(pin, &q)) return -1;
    orc_stats total; memset(&total, 0, sizeof total);
    int threads = q.threads;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#else
    threads = 1;
#endif
    (void)threads;
#pragma omp parallel num_threads(threads)
    {
        tracer T; memset(&T, 0, sizeof T);
        T.shapes = shapes; T.n_shapes = n_shapes; T.tris = tris; T.n_tris = n_tris;
        T.cam = camera; T.p = &q; T.seed_k = seed_key(q.seed);
        const int lit_before = t_lit, poly_before = t_poly;            /* the flags are per thread: set them in every worker */
        set_arith(q.arith);
#pragma omp for schedule(dynamic, 1)
        for (int64_t y = 0; y < (int64_t)q.height; ++y) {
            if (!row_in_shard(&q, (uint32_t)y)) continue;
            for (uint32_t x = 0; x < q.width; ++x) {
                uint32_t* px = accum + 4u * ((size_t)y * q.width + x);
                if (q.semantics == ORC_SEM_GLSL) {
                    uint32_t r = px[0], g = px[1], b = px[2];
                    for (uint32_t s = 0; s < q.spp; ++s) {
                        v3 c = path_color(&T, x, (uint32_t)y, q.first_sample + s);
                        r += quantise(c.x); g += quantise(c.y); b += quantise(c.z);   /* glsl:301-307 */
                    }
                    px[0] = r; px[1] = g; px[2] = b;
                } else {
                    float f[3]; memcpy(f, px, sizeof f);
                    for (uint32_t s = 0; s < q.spp; ++s) {                            /* rayTracer.h:96-103 */
                        v3 c = path_color(&T, x, (uint32_t)y, q.first_sample + s);
                        f[0] += c.x; f[1] += c.y; f[2] += c.z;
                    }
                    memcpy(px, f, sizeof f);
                }
            }
        }
        t_lit = lit_before; t_poly = poly_before;
#pragma omp critical
        stats_add(&total, &T.st);
    }
    if (stats) *stats = total;
    return 0;
}

int orc_trace(const orc_shape* shapes, uint32_t n_shapes, const orc_triangle* tris, uint32_t n_tris,
              const float camera[21], const orc_params* pin, uint32_t sample,
              int32_t* hit_index, uint32_t* material_id)
{
    orc_params q;
    if (fix_params(pin, &q)) return -1;
    tracer T; memset(&T, 0, sizeof T);
    T.shapes = shapes; T.n_shapes = n_shapes; T.tris = tris; T.n_tris = n_tris;
    T.cam = camera; T.p = &q; T.seed_k = seed_key(q.seed);
    const int lit_before = t_lit, poly_before = t_poly;
    set_arith(q.arith);
    for (uint32_t y = 0; y < q.height; ++y)
        for (uint32_t x = 0; x < q.width; ++x) {
            size_t base = ((size_t)y * q.width + x) * q.max_depth;
            for (uint32_t b = 0; b < q.max_depth; ++b) { hit_index[base + b] = -2; material_id[base + b] = 0xffffffffu; }
            T.trace_hit = hit_index + base; T.trace_mat = material_id + base;
            (void)path_color(&T, x, y, sample);
        }
    t_lit = lit_before; t_poly = poly_before;
    return 0;
}

/* rayTracer.h:105 (sum / totalRays, cwiseSqrt, row flip :78) and
 * PixelConversion.glsl:17-20 (uint sum / (255 * samples)). */
void orc_resolve(const uint32_t* accum, uint32_t width, uint32_t height, uint32_t samples,
                 uint32_t semantics, int gamma_sqrt, int flip_rows, float* rgba)
{
    for (uint32_t y = 0; y < height; ++y) {
        uint32_t oy = flip_rows ? height - 1 - y : y;
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t* px = accum + 4u * ((size_t)y * width + x);
            float* o = rgba + 4u * ((size_t)oy * width + x);
            for (int k = 0; k < 3; ++k) {
                float v;
                if (semantics == ORC_SEM_GLSL) v = (float)px[k] / (255.0f * (float)samples);
                else { float f; memcpy(&f, &px[k], 4); v = f / (float)samples; }
                o[k] = gamma_sqrt ? sqrtf(v) : v;
            }
            o[3] = 1.0f;
        }
    }
}

/* PixelConversion.glsl:12-28: running mean over frames, then clear. */
void orc_convert(uint32_t* accum, float* image, uint32_t width, uint32_t height,
                 uint32_t samples_in_accum, uint32_t semantics, uint32_t n_samples)
{
    float nf = (float)n_samples;
    float wa = (nf - 1.0f) / nf, wb = 1.0f / nf;                       /* :22 */
    for (uint32_t y = 0; y < height; ++y) {
        uint32_t oy = height - 1 - y;
        for (uint32_t x = 0; x < width; ++x) {
            uint32_t* px = accum + 4u * ((size_t)y * width + x);
            float* o = image + 4u * ((size_t)oy * width + x);
            float cur[4];
            for (int k = 0; k < 3; ++k) {
                if (semantics == ORC_SEM_GLSL) cur[k] = (float)px[k] / (255.0f * (float)samples_in_accum);
                else { float f; memcpy(&f, &px[k], 4); cur[k] = f / (float)samples_in_accum; }
            }
            cur[3] = 1.0f;                                             /* :20 */
            for (int k = 0; k < 4; ++k)
                o[k] = (nf != 1.0f) ? fmaf(wa, o[k], cur[k] * wb) : cur[k];   /* :21-24 */
            px[0] = px[1] = px[2] = 0; px[3] = 0;                      /* :25-27 */
        }
    }
}

/* ------------------------------------------------- host-side constructors */

/* Camera::Camera, src/gpu_version/camera.h:14-28 (rayTracer.h:37-46 is the
 * same with position (1,0,6)). */
void orc_camera_init(float data[21], uint32_t width, uint32_t height, float focal, const float pos[3])
{
    memset(data, 0, 21 * sizeof(float));
    data[0] = pos[0]; data[1] = pos[1]; data[2] = pos[2];              /* :17 */
    data[4] = 0; data[5] = 0; data[6] = -1;                            /* :18 */
    data[12] = 0; data[13] = 1; data[14] = 0;                          /* :20 */
    data[16] = 1; data[17] = 0; data[18] = 0;                          /* :21 */
    unsigned int wh = width * height;
    float world_w = (float)(width * (1 / sqrt((double)wh)));           /* :23 */
    float world_h = (float)(height * (1 / sqrt((double)wh)));          /* :24 */
    data[19] = world_w / (float)width;                                 /* :25 */
    data[8] = -world_w / 2 + pos[0];                                   /* :26 */
    data[9] = world_h / 2 + pos[1];
    data[10] = -focal + pos[2];
}

/* SimpleTriangle::SimpleTriangle, shape.h:83-102.  Host arithmetic as Eigen
 * evaluates it: no fused operations, sums left to right. */
static inline float hdot(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void orc_make_triangle(orc_triangle* out, const float p1[3], const float p2[3], const float p3[3])
{
    memset(out, 0, sizeof *out);
    float v1[3], v2[3], n[3];
    for (int k = 0; k < 3; ++k) { out->position[k] = p1[k]; v1[k] = p2[k] - p1[k]; v2[k] = p3[k] - p1[k]; }  /* :84-87 */
    n[0] = v1[1] * v2[2] - v1[2] * v2[1];
    n[1] = v1[2] * v2[0] - v1[0] * v2[2];
    n[2] = v1[0] * v2[1] - v1[1] * v2[0];
    float len = sqrtf(hdot(n, n));
    for (int k = 0; k < 3; ++k) out->a[k] = n[k] / len;                /* :88 */
    out->beta = hdot(out->a, p1);                                      /* :89 */
    float f1 = hdot(v1, v2) / hdot(v2, v2), m1[3];
    for (int k = 0; k < 3; ++k) m1[k] = v1[k] - f1 * v2[k];            /* :93 */
    float d1 = hdot(m1, v1);
    for (int k = 0; k < 3; ++k) out->m1[k] = m1[k] / d1;               /* :94 */
    float f2 = hdot(v1, v2) / hdot(v1, v1), m2[3];
    for (int k = 0; k < 3; ++k) m2[k] = v2[k] - f2 * v1[k];            /* :96 */
    float d2 = hdot(m2, v2);
    for (int k = 0; k < 3; ++k) out->m2[k] = m2[k] / d2;               /* :97 */
}

/* SimpleMesh::loadMesh (SimpleMesh.h:145-215), normalize (:60-77), resize
 * (:79-85), transform (:122-126) and the triangle loop of main.cpp:113-122. */
int orc_mesh_triangles(const char* path, int pipeline, orc_triangle* out, uint32_t max_tris, uint32_t* n_vertices)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char tag[8] = { 0 };
    unsigned nv = 0, np = 0, ne = 0;
    if (fscanf(f, "%4s %u %u %u", tag, &nv, &np, &ne) != 4) { fclose(f); return -2; }
    int coff = strcmp(tag, "COFF") == 0;
    if (!coff && strcmp(tag, "OFF") != 0) { fclose(f); return -3; }
    float* pos = (float*)malloc(sizeof(float) * 3 * (nv ? nv : 1));
    for (unsigned i = 0; i < nv; ++i) {
        if (fscanf(f, "%f %f %f", &pos[3 * i], &pos[3 * i + 1], &pos[3 * i + 2]) != 3) { free(pos); fclose(f); return -4; }
        if (coff) { float c[4]; if (fscanf(f, "%f %f %f %f", &c[0], &c[1], &c[2], &c[3]) != 4) { free(pos); fclose(f); return -4; } }
    }
    if (pipeline) {
        float c[3] = { 0, 0, 0 };
        for (unsigned i = 0; i < nv; ++i) for (int k = 0; k < 3; ++k) c[k] += pos[3 * i + k];     /* :52-58 */
        for (int k = 0; k < 3; ++k) c[k] = c[k] / (float)nv;
        float radius = 0;
        for (unsigned i = 0; i < nv; ++i) {                                                         /* :70-77 */
            float d[3] = { pos[3 * i] - c[0], pos[3 * i + 1] - c[1], pos[3 * i + 2] - c[2] };
            float nrm = sqrtf(hdot(d, d));
            if (nrm > radius) radius = nrm;
        }
        const float ang = (float)(-3.14159265358979323846 / 2);                                    /* main.cpp:112 */
        const float cs = cosf(ang), sn = sinf(ang);
        /* 0.5 * AngleAxisf(ang, UnitX).toRotationMatrix() */
        const float R[9] = { 0.5f * 1.0f, 0.5f * 0.0f, 0.5f * 0.0f,
                             0.5f * 0.0f, 0.5f * cs,   0.5f * -sn,
                             0.5f * 0.0f, 0.5f * sn,   0.5f * cs };
        const float tr[3] = { 0.5f, 0.7f, -2.0f };
        for (unsigned i = 0; i < nv; ++i) {
            float v[3];
            for (int k = 0; k < 3; ++k) v[k] = (pos[3 * i + k] - c[k]) / radius;                    /* :66 */
            for (int k = 0; k < 3; ++k) v[k] = v[k] * 2.0f;                                         /* :81-84 */
            for (int k = 0; k < 3; ++k)
                pos[3 * i + k] = ((R[3 * k] * v[0] + R[3 * k + 1] * v[1]) + R[3 * k + 2] * v[2]) + tr[k];   /* :124 */
        }
    }
    uint32_t count = 0;
    for (unsigned i = 0; i < np; ++i) {
        unsigned k, a, b, c;
        if (fscanf(f, "%u %u %u %u", &k, &a, &b, &c) != 4 || k != 3 || a >= nv || b >= nv || c >= nv) { free(pos); fclose(f); return -5; }
        if (count < max_tris) orc_make_triangle(&out[count], &pos[3 * a], &pos[3 * b], &pos[3 * c]);
        ++count;
    }
    if (n_vertices) *n_vertices = nv;
    free(pos); fclose(f);
    return (int)count;
}
