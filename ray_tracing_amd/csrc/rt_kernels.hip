// rt_kernels.hip — gfx950 path-tracing kernels.
//
// Replaces src/shaders/RayTracing.glsl (one workgroup of 10x10 invocations per
// pixel, three contended atomicAdds per sample) and src/shaders/
// PixelConversion.glsl of the reference.  Here a lane owns a pixel and walks
// that pixel's samples one after the other, so a pixel's sum is formed in
// registers in sample order and written once: no atomics, and the result does
// not depend on scheduling.  A wave covers an 8x8 pixel tile, so that the 64
// camera rays of an iteration are coherent and later bounces share a small
// neighbourhood of the scene.  When a lane's path ends it starts the pixel's
// next sample in the same loop iteration in which its neighbours trace a
// bounce: lanes never wait for the longest path of a sample.
//
// Scene data: every workgroup first copies the scene blob (rt_device.h) into
// LDS.  Spheres are tested by the whole wave at once: every lane tests the same
// sphere at the same time, reading it from LDS with a wave-uniform address
// (identical addresses broadcast), which puts the primitive in VECTOR registers:
// on this chip a vector instruction with a scalar-register operand issues at
// half the rate of one whose operands are all vector registers, so the scene is
// kept out of the scalar file on purpose.  Boxes and triangles are different:
// a given box or leaf of the mesh concerns only the few lanes whose rays come
// near it, so those lanes queue (lane, primitive) pairs in LDS and the wave
// then tests the queued pairs side by side, one or a few lanes per pair — see
// the tree walk in trace_kernel.  The record of the primitive a lane finally
// hit is gathered per lane, also from LDS.
//
// Arithmetic: binary32 throughout, compiled with -ffp-contract=off.  Every
// fused multiply-add is an explicit __builtin_fmaf; sqrt is correctly rounded
// (v_rsq_f32 plus one exact correction step, sqrt_rn below); reciprocal square root and reciprocal are
// fixed Newton recipes; sin/cos a fixed polynomial.  No OCML function feeds
// geometry.  With that, every value computed here can be reproduced bit for
// bit by scalar IEEE-754 code on a CPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "../../include/rt_mi355x.h"
#include "rt_device.h"

namespace rtd {

#ifndef RT_COST_CLOCK
#define RT_COST_CLOCK 1
#endif
// Waves per SIMD the instances with queues are built for (their register budget: 512 / this, in steps of 8).
#ifndef RT_QUEUE_WAVES
#define RT_QUEUE_WAVES 7
#endif
// A box that at least NUM / DEN of the live lanes touch is tested by the whole wave on the spot instead of through the queue.
// (Measured on the mesh scene, four boxes of which one is the floor: never on the spot 14.51 ms, from 3/4 of the lanes 14.39, from 1/2 14.34.)
#ifndef RT_BOX_DIRECT_NUM
#define RT_BOX_DIRECT_NUM 1u
#define RT_BOX_DIRECT_DEN 2u
#endif
// Measurement build (-DRT_PHASE_CLOCKS=1): the timeline record of a wave holds, instead of its start and end, the
// shader clocks / 16 its loop spent in the sphere scan, the boxes and cubes, the triangle tree, and the rest
// (surface, draws, new direction); tools/phase_clocks.py sums them.
#ifndef RT_PHASE_CLOCKS
#define RT_PHASE_CLOCKS 0
#endif
// RT_PHASE_CLOCKS=2 splits the last of these instead: everything up to the closest hit / the surface's record and
// normal / accumulation, draws and the scatter or restart / the new direction.
#define RT_PHASE_MARK_(k) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); phase_clk[k] += now_ - phase_t; phase_t = now_; }
// RT_PHASE_CLOCKS=3: the triangle tree alone: everything else / the walk by the whole wave and the queueing / the
// passes over queued nodes / the passes over queued leaves.
// RT_PHASE_CLOCKS=4: primary steps (all of them) / the closest hit of the other steps / their rest; the fourth word
// counts the primary steps (x 16).
#define RT_PHASE_MARK(k)
#define RT_PHASE_MARK2(k)
#define RT_PHASE_MARK3(k)
#define RT_PHASE_MARK4(k)
#if RT_PHASE_CLOCKS == 1
#undef RT_PHASE_MARK
#define RT_PHASE_MARK(k) RT_PHASE_MARK_(k)
#elif RT_PHASE_CLOCKS == 2
#undef RT_PHASE_MARK2
#define RT_PHASE_MARK2(k) RT_PHASE_MARK_(k)
#elif RT_PHASE_CLOCKS == 3
#undef RT_PHASE_MARK3
#define RT_PHASE_MARK3(k) RT_PHASE_MARK_(k)
#elif RT_PHASE_CLOCKS == 4
#undef RT_PHASE_MARK4
#define RT_PHASE_MARK4(k) RT_PHASE_MARK_(k)
#elif RT_PHASE_CLOCKS == 5
// the four phases of RT_PHASE_CLOCKS=1, over the steps that are not primary steps
#undef RT_PHASE_MARK
#define RT_PHASE_MARK(k) { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); if (!pstep) phase_clk[k] += now_ - phase_t; phase_t = now_; }
#endif
// Primary steps (see trace_kernel): -DRT_PSTEP=0 builds without them.
#ifndef RT_PSTEP
#define RT_PSTEP 1
#endif

struct V3 { float x, y, z; };
constexpr uint32_t WORK_NONE = 0xffffffffu;      // a plan entry without work

#define RT_DEV static __device__ __forceinline__

// the first enabled lane of the wave (counts one per wave whatever the lanes taking part)
RT_DEV bool lane_is_first() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (uint32_t)__builtin_ctzll(__ballot(1)); }
RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// o + t*d
RT_DEV V3 madd(float t, V3 d, V3 o) { return v3(fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)); }
RT_DEV V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }
// 1/sqrt(q): bit-trick first guess + three Newton steps, every step an IEEE
// operation (relative error <= 1.3e-7).  A third of the issue slots of a
// correctly rounded sqrt followed by a correctly rounded divide, and — unlike
// v_rsq_f32 — reproducible bit for bit on a CPU.
RT_DEV float rsqrt_nr(float q)
{
    float y = __uint_as_float(0x5f3759dfu - (__float_as_uint(q) >> 1));
    const float h = 0.5f * q;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = h * y;
        float e = fma_(-t, y, 1.5f);
        y = y * e;
    }
    return y;
}
RT_DEV V3 normalize(V3 v) { return v * rsqrt_nr(dot(v, v)); }
// normalize(p - centre) at a point of a sphere (scene.h:130-132, RayTracing.glsl:134-136): the length is the radius
// to the rounding of the hit distance (relative d, 1e-7 .. 1e-5), so 1/|radius| — stored with the shape — is a first
// guess that ONE Newton step y <- y * (1.5 - (0.5 q y) y) takes to 1.5 d^2 of 1/sqrt(q): four instructions for the
// thirteen of the bit-trick guess and its three steps.  The step's own factor e = 1 - d tells how good the guess
// was: where it is further than SPHERE_NORMAL_GUESS from 1 (a tiny sphere far away, whose hit point is off the
// sphere by more than that share of the radius) the lane takes the full normalisation instead — a branch no lane of
// an ordinary scene takes.
constexpr float SPHERE_NORMAL_GUESS = 0x1p-12f;
RT_DEV V3 sphere_normal(V3 v, float inv_radius)
{
    const float q = dot(v, v);
    const float t = (0.5f * q) * inv_radius;
    const float e = fma_(-t, inv_radius, 1.5f);
    V3 n = v * (inv_radius * e);
    const bool off = !(__builtin_fabsf(e - 1.0f) <= SPHERE_NORMAL_GUESS);
    if (__any(off)) {
        const float full = rsqrt_nr(q);
        n = v3(off ? v.x * full : n.x, off ? v.y * full : n.y, off ? v.z * full : n.z);
    }
    return n;
}
// 1/x the same way: y <- y * (2 - x*y), three times (relative error <= 1.2e-7).
// a / b of the shader's routines (RayTracing.glsl:145,161,181) is a * rcp_nr(b):
// GLSL specifies its division to 2.5 ulp only, and the correctly rounded
// divide costs about 42 issue cycles here against 19.
RT_DEV float rcp_nr(float x)
{
    float y = __uint_as_float(0x7ef311c7u - __float_as_uint(x));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float e = fma_(-x, y, 2.0f);
        y = y * e;
    }
    return y;
}
// v_mul_legacy_f32: a product in which 0 times anything — an infinity too — is 0.  The compiler
// does not see into the asm, so the wait state a vector instruction needs before it reads the
// result of a transcendental one (v_rsq_f32 below) is written out: without it some lanes read
// the register before the result is there.
RT_DEV float mul_legacy_after_trans(float a, float b)
{
    float r;
    asm("s_nop 0\n\tv_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
RT_DEV float mul_legacy(float a, float b)
{
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// Correctly rounded sqrt from ONE transcendental instruction: y = v_rsq_f32(x) (within 1 ulp),
// g = x*y, and one correction step s = g + (x - g*g) * y/2 whose residual is exact (fma).
// tools/sqrt_exact.hip runs this sequence over every binary32 input on the chip: the result is
// the IEEE square root for all x in [2^-96, FLT_MAX] and for x = 0 (the two legacy products
// carry 0 * inf through as 0).  For arguments in (0, 2^-96), where the residual goes
// subnormal, the result is only a tiny number of the right order; every caller below
// tolerates that (see them).  Negative x gives NaN, +inf gives 0 (no caller can pass it).
// Six instructions and 19 issue cycles against nine and 34 for v_sqrt_f32 with a residual test
// against each neighbour, and 53 cycles for the compiler's sqrtf.
RT_DEV float sqrt_rn(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = mul_legacy_after_trans(x, y);
    const float h = 0.5f * y;
    const float r = fma_(-g, g, x);
    return g + mul_legacy(r, h);
}
RT_DEV V3 cross(V3 a, V3 b)
{
    return v3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
// The reference's expressions as a C++ compiler without contraction evaluates them (RT_FLAG_LITERAL_GEOMETRY): every
// product and sum rounded on its own, v / sqrt(v.v) with the IEEE root and quotient for normalized().
RT_DEV float dot_plain(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV V3 madd_plain(float t, V3 d, V3 o) { return v3(t * d.x + o.x, t * d.y + o.y, t * d.z + o.z); }
RT_DEV V3 normalize_plain(V3 v)
{
    const float len = __builtin_sqrtf(dot_plain(v, v));
    return v3(v.x / len, v.y / len, v.z / len);
}
template <bool LIT> RT_DEV float dot_as(V3 a, V3 b) { return LIT ? dot_plain(a, b) : dot(a, b); }
template <bool LIT> RT_DEV V3 madd_as(float t, V3 d, V3 o) { return LIT ? madd_plain(t, d, o) : madd(t, d, o); }
// GLSL sign(): -1, 0, 1; NaN gives 0.
RT_DEV float sign_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
// GLSL max(): (x < y) ? y : x — not v_max_f32, whose NaN rule differs.
RT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }

// ------------------------------------------------------------------- RNG
// Counter-based, one stream per path: the path (seed, pixel, sample) has the key
// hash32(seed_key ^ pixel) + sample * 0x85EBCA6B, and its draws come in pairs — every event of a
// path needs two — from ONE hashed word per pair: x = hash32(key + pair * 0x9E3779B9) gives
// draw 2*pair from its top 24 bits and draw 2*pair + 1 from the top 24 bits of x * 0xADB4A92D (as a
// pair the two lie on the lattice of that multiplier modulo 2^32, whose shortest vector is 0.98 of the best
// possible: the 2^32 points of a pair are spread evenly over the unit square).
// Pair 0: sub-pixel jitter.  Bounce b: pair 1 + 2b (pair 2 + 2b is left to samplers that need four
// draws).  It stands in for random() of RayTracing.glsl:44-51 (a hash of time, invocation id and
// draw count) and for the mt19937 of src/utils.h:8-12.  The two steps differ so that no (sample,
// pair) meets another one's hash input: s * 0x85EBCA6B + p * 0x9E3779B9 repeats only for sample
// numbers 17 million apart (pairs up to 128 apart).
RT_DEV uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
constexpr uint32_t SAMPLE_STEP = 0x85ebca6bu;
RT_DEV uint32_t path_key(uint32_t pixel_hash, uint32_t sample) { return pixel_hash + sample * SAMPLE_STEP; }
RT_DEV void draw_pair(uint32_t key, uint32_t pair, float& a, float& b)
{
    const uint32_t x = hash32(key + pair * 0x9e3779b9u);
    const uint32_t y = x * 0xadb4a92du;
    a = (float)(x >> 8);
    b = (float)(y >> 8);
}
// A draw travels as the integer it is made of, 0 .. 2^24 - 1 as a float ("raw"); the value in [0, 1) is raw * 2^-24.
// Scaling by a power of two is exact, so every consumer folds the factor into a constant it multiplies by anyway
// (fma(-2, u, 1) = fma(-2^-23, raw, 1) to the last bit) and the two multiplications per pair are never issued.
constexpr float RAW = 0x1p-24f;
RT_DEV uint32_t bounce_pair(uint32_t bounce) { return 1u + 2u * bounce; }

RT_DEV void sincos2pi(float u, float& c_out, float& s_out);
RT_DEV void sincos2pi_raw(float a, float& c_out, float& s_out);

// random() of RayTracing.glsl:44-51, the shader's own stream, for contexts created with
// RT_RNG_GLSL: the state is the vec3 `seed` of RayTracing.glsl:9; every draw adds (1.78, 3.1,
// 50.78) to it and returns fract(sin(dot(seed, (12.9898, 78.233, 51.112))) * 43758.5453123).
// The sine is taken of the angle reduced to one turn, with the polynomial used everywhere
// here.  Draws are consumed in program order, so the state travels with the path.
RT_DEV float glsl_random(V3& s)
{
    s = v3(s.x + 1.78f, s.y + 3.1f, s.z + 50.78f);
    const float x = fma_(s.z, 51.112f, fma_(s.y, 78.233f, s.x * 12.9898f));
    const float t = x * 0x1.45f306p-3f;                  // x / (2 pi)
    const float u = t - __builtin_floorf(t);
    float c, sn;
    sincos2pi(u, c, sn);
    const float v = sn * 43758.5453123f;
    return v - __builtin_floorf(v);
}

// cos(2 pi u), sin(2 pi u), u in [0,1), without a quadrant case split: the half
// angle h = pi*u - pi/2 lies in [-pi/2, pi/2), where one odd and one even Taylor
// polynomial (Horner form) give sin h and cos h; then cos(2 pi u) = sin^2 h -
// cos^2 h and sin(2 pi u) = -2 sin h cos h.
RT_DEV void sincos_half(float x, float& c_out, float& s_out);
RT_DEV void sincos2pi(float u, float& c_out, float& s_out) { sincos_half(fma_(u, 0x1.921fb6p+1f, -0x1.921fb6p+0f), c_out, s_out); }
// the same for a raw draw: u * pi = raw * (pi * 2^-24), both products exact to the same real number
RT_DEV void sincos2pi_raw(float a, float& c_out, float& s_out) { sincos_half(fma_(a, 0x1.921fb6p+1f * RAW, -0x1.921fb6p+0f), c_out, s_out); }
// x = the half angle pi*u - pi/2
RT_DEV void sincos_half(float x, float& c_out, float& s_out)
{
    float x2 = x * x;
    float ps = -0x1.ae6456p-26f;
    ps = fma_(ps, x2, 0x1.71de3ap-19f);
    ps = fma_(ps, x2, -0x1.a01a02p-13f);
    ps = fma_(ps, x2, 0x1.111112p-7f);
    ps = fma_(ps, x2, -0x1.555556p-3f);
    ps = fma_(ps, x2, 1.0f);
    float s1 = x * ps;
    float pc = 0x1.1eed8ep-29f;
    pc = fma_(pc, x2, -0x1.27e4fcp-22f);
    pc = fma_(pc, x2, 0x1.a01a02p-16f);
    pc = fma_(pc, x2, -0x1.6c16c2p-10f);
    pc = fma_(pc, x2, 0x1.555556p-5f);
    pc = fma_(pc, x2, -0.5f);
    float c1 = fma_(pc, x2, 1.0f);
    c_out = fma_(s1, s1, -(c1 * c1));
    s_out = (s1 * -2.0f) * c1;
}

// A point uniform on the unit sphere: what randomUnitVector() of
// src/utils.h:37-46 (three normal deviates, normalised) is distributed as.  Raw draws.
RT_DEV V3 unit_vector(float a1, float a2)
{
    float z = fma_(-2.0f * RAW, a1, 1.0f);
    float r = sqrt_rn(fma_(-z, z, 1.0f));            // 1 - z*z is 0 or >= 2^-23
    float c, s;
    sincos2pi_raw(a2, c, s);
    return v3(r * c, r * s, z);
}

// ln(raw * 2^-24) for a raw draw 1 .. 2^24 - 1: raw = m * 2^e with m folded into [sqrt(1/2), sqrt(2)),
// ln m = f + f^2 q(f), f = m - 1, q of degree 7 (4e-8 absolute), plus (e - 24) ln 2.  IEEE operations in a fixed
// order: the oracle's orc_log_draw to the bit.
RT_DEV float log_raw(float raw)
{
    const uint32_t b = __float_as_uint(raw);
    int e = (int)(b >> 23) - 127;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f800000u);
    const bool fold = m > 0x1.6a09e6p+0f;
    m = fold ? m * 0.5f : m;
    e += fold ? 1 : 0;
    const float f = m - 1.0f;
    float q = 0x1.70cfe8p-4f;
    q = fma_(q, f, -0x1.23ffe8p-3f);
    q = fma_(q, f, 0x1.2f3c7ep-3f);
    q = fma_(q, f, -0x1.5374fcp-3f);
    q = fma_(q, f, 0x1.990c98p-3f);
    q = fma_(q, f, -0x1.000438p-2f);
    q = fma_(q, f, 0x1.555630p-2f);
    q = fma_(q, f, -0x1.fffffcp-2f);
    const float lm = fma_(f * f, q, f);
    return fma_((float)(e - 24), 0x1.62e430p-1f, lm);
}
// randomUnitVector() as written, src/utils.h:45: three normal deviates (Box-Muller on four raw draws; the fourth
// deviate is dropped), normalised.
RT_DEV V3 unit_vector_gauss(float a1, float a2, float a3, float a4)
{
    a1 = (a1 < 1.0f) ? 1.0f : a1;                     // a zero draw counts as the smallest one
    a3 = (a3 < 1.0f) ? 1.0f : a3;
    const float r1 = sqrt_rn(__builtin_fabsf(-2.0f * log_raw(a1))), r2 = sqrt_rn(__builtin_fabsf(-2.0f * log_raw(a3)));
    float c1, s1, c2, s2;
    sincos2pi_raw(a2, c1, s1);
    sincos2pi_raw(a4, c2, s2);
    return normalize(v3(r1 * c1, r1 * s1, r2 * c2));
}

// randomOnHalfUnitShere, RayTracing.glsl:64-80.
RT_DEV V3 hemisphere_glsl(V3 n, float a1, float a2)            // raw draws
{
    float c, s;
    sincos2pi_raw(a1, c, s);
    float r = a2 * RAW;
    float dist = sqrt_rn(fma_(-r, r, 1.0f));         // 1 - r*r >= 2^-24
    float x = c * dist, y = s * dist;
    float q = fma_(n.x, n.x, n.y * n.y);          // |(-n.y, n.x, 0)|^2
    // length(b1.xy) < 0.01 (:72): the correctly rounded square root is monotonic, so the comparison holds exactly for
    // q below 0x38d1b716 (9.999999e-05), the first float whose root reaches 0.01f — no root needed.
    V3 b1;
    if (q < __uint_as_float(0x38d1b716u)) b1 = normalize(v3(0.0f, -n.z, n.y));
    else { float inv = rsqrt_nr(q); b1 = v3(-n.y * inv, n.x * inv, 0.0f * inv); }
    V3 b2 = cross(n, b1);
    return v3(fma_(n.x, r, fma_(b2.x, y, b1.x * x)),
              fma_(n.y, r, fma_(b2.y, y, b1.y * x)),
              fma_(n.z, r, fma_(b2.z, y, b1.z * x)));
}

// ---------------------------------------------------------- intersection
// A hit distance t lies in the acceptance window and is closer than the best
// so far exactly when  bits(t) - TMIN  <  bits(best) - TMIN  as unsigned
// integers, TMIN being the bit pattern of the window's first accepted value:
// positive floats order like their bit patterns, and anything below the window,
// negative, infinite or NaN wraps to a number above every possible right-hand
// side.  One integer compare stands for the window's two tests and the `<` of
// the closest-hit loop (scene.h:237 with rayTracer.h:117; RayTracing.glsl:212).
//   CPU tracer: 0.001 < t (double against widened float: true exactly from
//               0.001f on) and t < FLT_MAX.
//   Shader:     0.0001 < t < 1000.
// `far` itself is outside: the running best starts one float below it.
template <int SEM> struct Window;
template <> struct Window<0> { static constexpr uint32_t tmin = 0x3a83126fu, far = 0x7f7fffffu; };   // 0.001f, FLT_MAX
template <> struct Window<1> { static constexpr uint32_t tmin = 0x38d1b718u, far = 0x447a0000u; };   // next above 0.0001f, 1000.0f

// The closest hit is the minimum of (distance, index) pairs in dictionary order: both
// reference tracers scan in index order and replace the best only by a strictly smaller
// distance (scene.h:236-252, RayTracing.glsl:212,226), and a triangle, whose index counts
// on from the shapes', replaces a shape only when strictly closer (RayTracing.glsl:271).
// Written as a minimum it no longer depends on the order of the tests, which frees the
// scan order: spheres before boxes, triangles along a space-filling curve.
struct Best { uint32_t key, id; };
RT_DEV void take_if_closer(Best& best, uint32_t key, uint32_t id, bool also = true)
{
    const unsigned long long cand = ((unsigned long long)key << 32) | id;
    const unsigned long long cur = ((unsigned long long)best.key << 32) | best.id;
    const bool take = also && cand < cur;
    best.key = take ? key : best.key;
    best.id = take ? id : best.id;
}

// Sphere::hit (src/cpu_version/scene.h:115-128) = hitSphere
// (RayTracing.glsl:118-132).  s = centre.xyz, radius^2.
// Spheres are tested in upload order, so inside their scan the reference's own rule suffices: a
// later sphere replaces the best only by a strictly smaller key (scene.h:237, glsl:212) — one
// 32-bit compare; the index joins the comparison only against boxes, cubes and triangles.
template <int SEM>
RT_DEV void test_sphere(float4 s, uint32_t shape_id, V3 o, V3 d, Best& best)
{
    V3 oc = v3(o.x - s.x, o.y - s.y, o.z - s.z);
    float b = dot(oc, d);
    float c = fma_(oc.z, oc.z, fma_(oc.y, oc.y, fma_(oc.x, oc.x, -s.w)));   // oc.oc - r*r
    float disc = fma_(b, b, -c);
    // The root -b - sqrt(disc) can only lie in the window (t > 0) when b < 0 and disc >= 0.
    // When that holds for no lane of the wave the square root is skipped.  Where it holds
    // for some lanes only, the others produce a NaN or a non-positive t, which the key
    // test rejects.  (A disc below 2^-96 makes the root irrelevant: t is then -b to the
    // last bit, or lies below the window.)
    if (!__any(b < 0.0f && disc >= 0.0f)) return;
    float t = -b - sqrt_rn(disc);
    const uint32_t key = __float_as_uint(t) - Window<SEM>::tmin;
    const bool take = key < best.key;
    best.key = take ? key : best.key;
    best.id = take ? shape_id : best.id;
}

// hitBox, RayTracing.glsl:138-156.  bx = centre.xyz, 1/side along x; iy, iz = 1/side along
// y and z (the shader has one side for all three, :139).  The same values as the shader's expressions, from fewer
// instructions (compares and selects issue at half the rate of the arithmetic here):
//   * t = (0.5 - ob*sign(ob)) / (sign(ob)*db), :144-145: ob*sign(ob) is |ob|, and sign(ob)*db is db with ob's sign bit
//     flipped in — or 0 where ob is 0 (or -0), which keeps that axis' "no candidate" (t = NaN);
//   * max(|p.x|, |p.y|, |p.z|), :150: the hardware's three-way maximum differs from GLSL's max only when an operand is a
//     NaN, and a NaN coordinate comes either with all three NaN or with an infinite one: rejected both ways;
//   * 0.4999 < m < 0.5001, :151: one unsigned compare of bit patterns (m is non-negative or a NaN).
RT_DEV float box_slab(float obk, float dbk)
{
    const uint32_t flip = __float_as_uint(obk) & 0x80000000u;
    float den = __uint_as_float(__float_as_uint(dbk) ^ flip);
    den = (obk == 0.0f) ? 0.0f : den;
    return (0.5f - __builtin_fabsf(obk)) * rcp_nr(den);
}
RT_DEV float hit_box(float4 bx, float iy, float iz, V3 o, V3 d)
{
    const float ix = bx.w;
    V3 ob = v3(ix * (o.x - bx.x), iy * (o.y - bx.y), iz * (o.z - bx.z));
    V3 db = v3(ix * d.x, iy * d.y, iz * d.z);
    float t[3];
    t[0] = box_slab(ob.x, db.x);
    t[1] = box_slab(ob.y, db.y);
    t[2] = box_slab(ob.z, db.z);
    constexpr uint32_t LO = __builtin_bit_cast(uint32_t, 0.4999f), HI = __builtin_bit_cast(uint32_t, 0.5001f);
    static_assert(LO < HI, "window");
    float best = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float px = fma_(db.x, t[k], ob.x), py = fma_(db.y, t[k], ob.y), pz = fma_(db.z, t[k], ob.z);
        const float inf_norm = __builtin_fmaxf(__builtin_fabsf(px), __builtin_fmaxf(__builtin_fabsf(py), __builtin_fabsf(pz)));
        const bool on_face = __float_as_uint(inf_norm) - (LO + 1u) < HI - (LO + 1u);
        if (on_face && best > t[k]) best = t[k];
    }
    return best;
}

// boxNormalAt, RayTracing.glsl:158-163.  c = centre, rx, ry, rz = rcp_nr(side) per axis.
RT_DEV V3 box_normal(V3 c, float rx, float ry, float rz, V3 p)
{
    float qx = (p.x - c.x) * rx, qy = (p.y - c.y) * ry, qz = (p.z - c.z) * rz;
    // step(0.499, |q|) * sign(q): 0 below 0.499 — where q = 0 lies too — and otherwise 1 with q's sign (q is a finite
    // number here: the point of a hit)
    V3 n = v3((__builtin_fabsf(qx) < 0.499f) ? 0.0f : __builtin_copysignf(1.0f, qx),
              (__builtin_fabsf(qy) < 0.499f) ? 0.0f : __builtin_copysignf(1.0f, qy),
              (__builtin_fabsf(qz) < 0.499f) ? 0.0f : __builtin_copysignf(1.0f, qz));
    return normalize(n);
}

// Cube::hit, src/cpu_version/scene.h:185-204, on the rows {R(i,*), c_i} of a cube.  As
// written there: the three slab distances are tried in axis order and the first whose point
// lies exactly on the surface (infinity norm == 0.5) is the hit, nearest or not; IEEE
// divisions.  The exact comparison makes the rounding of origin + direction * t part of the
// result, so nothing here is fused: every product and sum is rounded on its own, in Eigen's
// order, as a C++ compiler without contraction evaluates those lines (the file is compiled
// with -ffp-contract=off; with a fused multiply-add the point lands exactly on 0.5 far less
// often and most of the cube disappears).  Returns the distance (+inf without a hit) and the
// point in the cube's frame.
RT_DEV float dot_unfused(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_DEV float hit_cube(float4 r0, float4 r1, float4 r2, V3 o, V3 d, V3& in_point)
{
    const V3 oc = v3(o.x - r0.w, o.y - r1.w, o.z - r2.w);
    const V3 ob = v3(dot_unfused(xyz(r0), oc), dot_unfused(xyz(r1), oc), dot_unfused(xyz(r2), oc));
    const V3 db = v3(dot_unfused(xyz(r0), d), dot_unfused(xyz(r1), d), dot_unfused(xyz(r2), d));
    const V3 sg = v3(sign_(ob.x), sign_(ob.y), sign_(ob.z));
    float t[3];
    t[0] = (0.5f - ob.x * sg.x) / (sg.x * db.x);
    t[1] = (0.5f - ob.y * sg.y) / (sg.y * db.y);
    t[2] = (0.5f - ob.z * sg.z) / (sg.z * db.z);
    float hit_t = __builtin_inff();
    bool found = false;
    in_point = v3(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float px = ob.x + db.x * t[k], py = ob.y + db.y * t[k], pz = ob.z + db.z * t[k];
        const bool on = gmax(__builtin_fabsf(px), gmax(__builtin_fabsf(py), __builtin_fabsf(pz))) == 0.5f;
        const bool first = on && !found;
        hit_t = first ? t[k] : hit_t;
        in_point = v3(first ? px : in_point.x, first ? py : in_point.y, first ? pz : in_point.z);
        found = found || on;
    }
    return hit_t;
}

// hitTriangle, RayTracing.glsl:179-194, on the four words of a SimpleTriangle
// (w0.w carries the triangle's index in the uploaded array).  The early return
// for k < 0 (:182-184) is left to the window test, which rejects every negative k.
// -> does the ray cross the triangle; `key` = its window key (any value outside the window loses later), `id` = its hit index.
template <int SEM>
RT_DEV bool triangle_candidate(float4 w0, float4 w1, float4 w2, float4 w3, uint32_t n_shapes, V3 o, V3 d, uint32_t& key, uint32_t& id)
{
    V3 a = xyz(w1);
    float k = (w1.w - dot(a, o)) * rcp_nr(dot(a, d));
    V3 p = madd(k, d, o);
    V3 q = v3(p.x - w0.x, p.y - w0.y, p.z - w0.z);
    float x = dot(q, xyz(w2));
    float y = dot(q, xyz(w3));
    key = __float_as_uint(k) - Window<SEM>::tmin;
    id = n_shapes + __float_as_uint(w0.w);
    return x > 0.0f && y > 0.0f && (x + y) < 1.0f;
}
template <int SEM>
RT_DEV void test_triangle(float4 w0, float4 w1, float4 w2, float4 w3, uint32_t n_shapes, V3 o, V3 d, Best& best)
{
    uint32_t key, id;
    const bool inside = triangle_candidate<SEM>(w0, w1, w2, w3, n_shapes, o, d, key, id);
    take_if_closer(best, key, id, inside);
}

// Can the ray touch the bounding sphere nb = {centre.xyz, R}?  It does when its line passes
// within R of the centre, (oc.d)^2 + R^2 >= |oc|^2 for a unit direction, and the sphere is not
// behind the origin.  A direction whose length is off by e shifts (oc.d)^2 by 2 e |oc|^2 — the shader's diffuse
// direction is used as sampled, unit to about 1e-6 — and the rounding of the two dot products adds about 1e-6 of
// |oc|^2 more: the term 4e-6 * |oc|^2 on the right covers both, R's own slack covers the triangle test's.  A NaN
// centre never hits.
template <int SEM>
RT_DEV bool bound_hit(float4 nb, V3 o, V3 d, uint32_t best_key)
{
    V3 oc = v3(o.x - nb.x, o.y - nb.y, o.z - nb.z);
    float b = dot(oc, d);
    float oc2 = dot(oc, oc);
    float rhs = fma_(4e-6f, oc2, nb.w * nb.w);
    // Nothing inside the sphere is nearer along the ray than -(b + R): with a hit already nearer than that
    // (by more than the rounding of b), the sphere cannot hold the closest one.
    float lim = __uint_as_float(best_key + Window<SEM>::tmin) * -1.00001f;
    return fma_(b, b, rhs) >= oc2 && (b < 0.0f || oc2 <= rhs) && (b + nb.w) >= lim;
}

// Can the ray touch the axis-aligned box {centre c, half extents h}?  Slab test with the reciprocal direction: along
// each axis the ray is inside the slab for t in [tc - hw, tc + hw], tc = (c - o) / d, hw = h / |d|; it touches the box
// when the latest entry is not after the earliest exit (a relative 1e-5 and an absolute 1e-6 cover the rounding of both,
// on top of the slack in h), the exit is not behind the origin and the entry not beyond the closest hit so far.  Written
// with negated comparisons: a NaN (0 * infinity: origin on a slab's plane and no movement along that axis) lets the ray
// through.
template <int SEM>
RT_DEV bool box_bound_hit(float4 c, float4 h, V3 o, V3 inv_d, uint32_t best_key)
{
    const float tcx = (c.x - o.x) * inv_d.x, tcy = (c.y - o.y) * inv_d.y, tcz = (c.z - o.z) * inv_d.z;
    const float hwx = h.x * __builtin_fabsf(inv_d.x), hwy = h.y * __builtin_fabsf(inv_d.y), hwz = h.z * __builtin_fabsf(inv_d.z);
    const float t_in = __builtin_fmaxf(__builtin_fmaxf(tcx - hwx, tcy - hwy), tcz - hwz);
    const float t_out = __builtin_fminf(__builtin_fminf(tcx + hwx, tcy + hwy), tcz + hwz);
    const float lim = __uint_as_float(best_key + Window<SEM>::tmin) * 1.00001f;
    return !(t_in > fma_(__builtin_fabsf(t_out), 1e-5f, t_out + 1e-6f)) && !(t_out < 0.0f) && !(t_in > lim);
}

// uvec4(255 * color), RayTracing.glsl:301; negative and NaN give 0, large
// values saturate (v_cvt_u32_f32 does exactly that).
RT_DEV uint32_t quantise(float c)
{
    float v = 255.0f * c;
    // the instruction itself: C leaves the conversion of an out-of-range value undefined, so spelled in C the
    // saturation costs two compares and two selects per channel on top of it
    uint32_t q;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(v));
    return q;
}

// One pixel's mean: rayTracer.h:105 (sum / totalRays, optional cwiseSqrt) and PixelConversion.glsl:17-20 (uint sum /
// (255 * samples)); alpha = 1.
template <int SEM>
RT_DEV float4 resolve_pixel(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t samples, bool gamma_sqrt)
{
    float4 v;
    if (SEM == 0) {
        const float ns = (float)samples;
        v.x = __uint_as_float(a0) / ns; v.y = __uint_as_float(a1) / ns; v.z = __uint_as_float(a2) / ns;
    } else {
        const float den = 255.0f * (float)samples;
        v.x = (float)a0 / den; v.y = (float)a1 / den; v.z = (float)a2 / den;
    }
    if (gamma_sqrt) { v.x = __builtin_sqrtf(v.x); v.y = __builtin_sqrtf(v.y); v.z = __builtin_sqrtf(v.z); }
    v.w = 1.0f;
    return v;
}

// -------------------------------------------------------------- the tracer
// SEM 0: rules of src/cpu_version (rayTracer.h:110-132, scene.h:39-109).
// SEM 1: rules of RayTracing.glsl:233-288.
// GEO bits 0-1: 0 spheres and cubes only (the instance has no traversal and no queue code), 1 boxes and / or
// triangles staged in LDS, 2 triangles and their tree read from the blob in global memory; bit 2: the scene holds cubes of the CPU tracer, dielectric
// materials or a colour per triangle, or the context asks for the reference's normalised-Gaussian sphere sampler (kept out of the other instances, whose registers and issue slots they
// would cost: 12 % on the four-sphere scene); bit 3: draws come from the
// shader's own random() instead of the counter hash; bit 4 (CPU rules, spheres and cubes only): the geometry in the
// reference's literal arithmetic, see LIT below; bit 5 (counter hash, recipe arithmetic): camera rays are traced in
// steps of their own from their tile's candidate list, see PSTEP below; bit 6 (a mesh in global memory, counter hash): the
// mesh's tree has levels above the walk's four, see DEEP_TREE below.  MODE 0: render.  MODE 1: render and count work.  MODE 2:
// follow one sample, record hit and material ids, leave the accumulator alone.
//
// One loop iteration = one ray segment of every live lane: closest hit, then
// the surface event.  Lanes whose path ended start their next sample in the
// same iteration, and the two expensive steps every continuing lane needs —
// two random draws and the normalisation of the new direction — are issued
// once for the whole wave, whether a lane is bouncing or starting a sample.
template <int SEM, int GEO, int MODE>
__global__ __launch_bounds__((GEO & 3) != 0 ? 64 * MAX_GROUP_WAVES : BLOCK)
// Spheres only: 64 registers, eight waves per SIMD.  With boxes or a mesh the staged scene and the wave's queues
// leave room for six workgroups per CU at most, so those instances may use the registers of six waves per SIMD.
// Sphere scenes with dielectrics, cubes or triangle colours need a few registers more than 64: seven waves, no spills.
__attribute__((amdgpu_waves_per_eu((GEO & 3) != 0 ? RT_QUEUE_WAVES : ((GEO & 4) != 0 ? 7 : 8), 8)))
void trace_kernel(const TraceParams P,
                  const float4* __restrict__ blob,
                  uint32_t* __restrict__ accum,
                  unsigned long long* __restrict__ stats,
                  int32_t* __restrict__ trace_hit,
                  uint32_t* __restrict__ trace_mat,
                  const uint32_t* __restrict__ tile_order,
                  uint32_t* __restrict__ tile_cost,
                  const uint32_t* __restrict__ tile_lists)
{
    constexpr bool STATS = MODE == 1, TRACE = MODE == 2;
    constexpr bool COST_CLOCK = RT_COST_CLOCK != 0;        // a tile's cost: shader clocks of its slowest wave, not loop iterations
    // The 64 x (samples) paths of a wave's 8x8 tile are one pool that its lanes draw from (shader rules, whose integer
    // sums take any order): see "pool" below.
    constexpr bool POOL = RT_POOL != 0 && SEM == 1 && !TRACE && (GEO & 3) != 0;
    // RT_FLAG_LITERAL_GEOMETRY: Sphere::hit, normalAt, the Ray constructor and reflect as scene.h:14-16,115-132 and
    // utils.h:62-64 are written — no fused operation, normalized() as v / sqrt(v.v) with the IEEE root and quotient,
    // the hit distance in double (HitRecord::t, scene.h:34; the unqualified sqrt of scene.h:123 is the double one) and
    // compared as a double (scene.h:237), narrowed where Ray::at multiplies a float vector by it.  The samplers keep
    // their polynomial (the CPU tracer's stream cannot be reproduced anyway).
    constexpr bool LIT = (GEO & 16) != 0;
    static_assert(!LIT || (SEM == 0 && (GEO & 3) == 0 && (GEO & 8) == 0), "literal geometry: CPU rules, spheres and cubes, counter hash");
    constexpr bool HAS_QUEUES = (GEO & 3) != 0, TRI_GLOBAL = (GEO & 3) == 2, SPREAD_BOXES = HAS_QUEUES, HAS_CUBE = SEM == 0 && (GEO & 4) != 0, HAS_GLASS = (GEO & 4) != 0, HAS_TCOL = (GEO & 4) != 0, RNG_GLSL = (GEO & 8) != 0,
                   HAS_GAUSS = SEM == 0 && (GEO & 4) != 0 && (GEO & 8) == 0;
    // Primary steps (below): instances with queues, whose closest hit is long.  Not with the shader's own stream, whose
    // pooled restart reads its pixel's coordinates from the pixel's home lane, which a primary step may have left out.
    // An instance of its own (GEO bit 5), launched when the scene has lists: the others keep their code as it was.
    constexpr bool PSTEP = (GEO & 32) != 0;
    // A mesh (in global memory) whose tree has levels above the walk's four: an instance of its own as well (GEO bit 6).
    constexpr bool DEEP_TREE = (GEO & 64) != 0;
    // The candidate spheres of a tile are bits of up to four words.  The sphere-only instances are built twice: for
    // scenes of up to 32 spheres (one word, kept in a register: the loop every small scene runs, and the headline's) and
    // for larger ones (GEO bit 7); their loop is short enough for the difference to show (3 % on the four-sphere scene).
    constexpr bool MANY_SPH = (GEO & 128) != 0 || (GEO & 3) != 0;
    static_assert((GEO & 128) == 0 || (PSTEP && (GEO & 3) == 0), "more than 32 candidate spheres: a sphere-only instance with primary steps");
    static_assert(!DEEP_TREE || ((GEO & 3) == 2 && (GEO & 8) == 0), "upper tree levels: a mesh in global memory, counter hash");
    static_assert(!PSTEP || (!RNG_GLSL && (GEO & 16) == 0), "primary steps: counter hash, recipe arithmetic");
    extern __shared__ float4 lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t n_samples = 0, n_segments = 0, n_diffuse = 0, n_specular = 0, n_terminal = 0, n_tri = 0, n_node = 0, n_iterations = 0, n_box_test = 0, n_box_bound = 0,
             n_sph_test = 0, n_psteps = 0, n_tree_entries = 0, n_tree_lanes = 0;

    // One 8x8-pixel tile per wave, four waves (a 16x16 tile) per workgroup; workgroups start in
    // blockIdx order.  `work` is the plan made from the previous frame of this view (tile_plan_kernel):
    // one entry per workgroup = tile | slice << 24 | (slices - 1) << 28, or WORK_NONE.  Tiles that cost
    // more than twice the mean come first, so that the launch does not end waiting for them; without a
    // plan the bottom rows of the frame (ground, objects) go first and the top rows (sky) last.  Under
    // the shader's rules, whose sums are integers, a very expensive tile is cut into several slices of
    // the sample range, each its own workgroup, added to the frame with atomics: otherwise the launch
    // lasts as long as its slowest wave, which walks all samples of its 64 pixels one after another.
    // Any order and any slicing give the same image.  (Handing tiles out dynamically through a
    // device-wide counter to a resident set of waves was tried: it evens out XCDs that run at
    // different speeds in the first launches of a process, but costs 10 registers and was 5 % slower
    // in steady state.)
    constexpr bool CAN_SLICE = SEM == 1 && !TRACE;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    // A workgroup is the four waves of one entry, or (P.group_waves = 1) a single wave, four consecutive
    // workgroups sharing an entry.
    const uint32_t q = blockIdx.x * P.group_waves + (tid >> 6);
    const uint32_t unit = (q >> 2) < P.plan_entries ? (tile_order ? tile_order[q >> 2] : (n_tiles - 1u - (q >> 2))) : WORK_NONE;
    // Workgroups of one, two or four waves lie inside one entry; those of three or of more than four may hold
    // entries with and without work, and leave only when none of their waves has any (every wave that stays takes
    // part in the staging loop and the barrier below).
    const bool one_entry = P.group_waves <= 4u && P.group_waves != 3u;
    if (one_entry ? unit == WORK_NONE : __syncthreads_count(unit != WORK_NONE) == 0) return;
    for (uint32_t i = tid; i < P.scene.lds_words; i += blockDim.x) lds[i] = blob[i];
    __syncthreads();
    if (unit == WORK_NONE) return;                       // uniform over the wave; no barrier follows
    {
    const uint32_t sub = q & 3u;
    const uint32_t tile = unit & 0xffffffu;
    const uint32_t slice = CAN_SLICE ? (unit >> 24) & 15u : 0u, n_slices = CAN_SLICE ? (unit >> 28) + 1u : 1u;
    const uint32_t tile_y = tile / P.tiles_x, tile_x = tile - tile_y * P.tiles_x;
    const uint32_t gx = tile_x * TILE_W + (sub & 1u) * 8u + (lane & 7u);
    const uint32_t row = tile_y * TILE_H + (sub >> 1) * 8u + (lane >> 3);   // ordinal among owned rows
    const uint32_t band = row / P.band_rows;
    const uint32_t gy = (band * P.shard_count + P.shard_index) * P.band_rows + (row - band * P.band_rows);
    const bool valid = gx < P.width && row < P.owned_rows && gy < P.height;

    const uint32_t pixel = gy * P.width + gx;
    const float step = P.cam[19];
    const V3 cam_pos = v3(P.cam[0], P.cam[1], P.cam[2]);
    const V3 cam_up = v3(P.cam[12], P.cam[13], P.cam[14]);
    const V3 cam_right = v3(P.cam[16], P.cam[17], P.cam[18]);
    // rayTracer.h:77 / glsl:293: upperLeft + step*gx*right - step*gy*up
    const V3 pc = madd_as<LIT>(-(step * (float)gy), cam_up, madd_as<LIT>(step * (float)gx, cam_right, v3(P.cam[8], P.cam[9], P.cam[10])));
    const uint32_t h1 = hash32(P.seed_key ^ pixel);

    // Running sums of this pixel: float for SEM 0, uint for SEM 1.
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    uint32_t* const acc_ptr = accum + 4u * (size_t)pixel;
    if (!TRACE && valid && n_slices == 1u && P.overwrite == 0u) {
        uint4 v = *reinterpret_cast<const uint4*>(acc_ptr);
        acc0 = v.x; acc1 = v.y; acc2 = v.z;
    }

    // This workgroup's part of the sample range: all of it, or slice `slice` of `n_slices` nearly equal ones.
    uint32_t sample = TRACE ? P.trace_sample : P.first_sample + (CAN_SLICE ? P.spp * slice / n_slices : 0u);
    const uint32_t sample_end = TRACE ? P.trace_sample + 1u : P.first_sample + (CAN_SLICE ? P.spp * (slice + 1u) / n_slices : P.spp);
    bool alive = valid && sample < sample_end;

    // Origin of the camera ray of `sample`: RayTracing.glsl:296 (jitter, from
    // two draws) or renderPixel<N>, src/cpu_version/rayTracer.h:92-101 (grid).
    V3 pcur = pc;                      // of the pixel this lane is tracing (POOL: not always its own)
    uint32_t cgx = gx, cgy = gy;
    // The jitter's vectors in vector registers (an operand from a scalar register halves the issue rate of the six
    // fused multiply-adds of every new camera ray), and the pixel step times 2^-24 for raw draws.
    V3 right_v = cam_right, up_v = cam_up;
    float kstep = step * RAW;
    // Only where registers are to spare: the instances with queues or cubes / dielectrics are at their budget.
    if (GEO == 0) asm volatile("" : "+v"(right_v.x), "+v"(right_v.y), "+v"(right_v.z), "+v"(up_v.x), "+v"(up_v.y), "+v"(up_v.z), "+v"(kstep));
    auto camera_origin = [&](uint32_t smp, float a1, float a2) -> V3 {       // raw draws
        if (P.sample_mode == 0u) {
            float s1 = kstep * a1, s2 = kstep * a2;                          // step * u, RayTracing.glsl:296
            return madd_as<LIT>(-s2, up_v, madd_as<LIT>(s1, right_v, pcur));
        }
        const uint32_t N = P.grid_n;
        const uint32_t i = (smp / N) % N, j = smp % N;
        float step_size = step / (float)(N + 1u);
        asm volatile("" : "+v"(step_size));                                  // computed where it is used: the grid's six step vectors would otherwise sit in registers through every launch
        const V3 ups = cam_up * step_size, rs = cam_right * step_size;
        V3 org = pcur - ups * (float)i;
        if (P.sample_mode == 2u) {
            for (uint32_t jj = 0; jj <= j; ++jj) {
                V3 inc = rs * (float)jj;
                org = v3(org.x + inc.x, org.y + inc.y, org.z + inc.z);
            }
        } else {
            V3 inc = rs * (float)j;
            org = v3(org.x + inc.x, org.y + inc.y, org.z + inc.z);
        }
        return org;
    };

    // RayTracing.glsl:9 for sample `smp` of this pixel: frame = smp / 100, local invocation
    // (smp % 10, smp / 10 % 10) of the pixel's 10 x 10 work group; `time` = (seed + frame) / 10000.
    auto glsl_seed = [&](uint32_t smp) -> V3 {
        const uint32_t frame = smp / 100u, lx = smp % 10u, ly = (smp / 10u) % 10u;
        const float time = (float)(P.glsl_seed32 + frame) * 0.0001f;
        return v3(time + (float)(10u * cgx + lx) * P.inv_w, time + (float)(10u * cgy + ly) * P.inv_h, time);
    };
    // The two draws that start sample `smp`: sub-pixel jitter, first draw with `right`.
    auto start_draws = [&](uint32_t smp, uint32_t k, V3& state, float& a, float& b) {
        if (RNG_GLSL) {
            state = glsl_seed(smp);
            a = b = 0.0f;
            if (P.sample_mode == 0u) { a = glsl_random(state) * (1.0f / RAW); b = glsl_random(state) * (1.0f / RAW); }   // the grid modes draw nothing here
        }
        else draw_pair(k, 0u, a, b);
    };
    V3 gs = v3(0, 0, 0);               // the shader stream's state (RNG_GLSL)
    uint32_t key = 0, bounce = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1), thr = v3(1, 1, 1);
    if (alive) {
        key = path_key(h1, sample);
        float j1, j2;
        start_draws(sample, key, gs, j1, j2);
        o = camera_origin(sample, j1, j2);
        d = LIT ? normalize_plain(o - cam_pos) : normalize(o - cam_pos);
        if (STATS) ++n_samples;
    }

    const SceneLayout& L = P.scene;
    const float4* const sph = lds + L.sph_off;
    const float4* const sid = lds + L.sid_off;
    const float4* const box = lds + L.box_off;
    const float4* const rec = lds + L.rec_off;
    const float4* const cub = lds + L.cub_off;
    const float4* const tri = TRI_GLOBAL ? blob + L.tri_off : lds + L.tri_off;
    const uint32_t* const slot_of = reinterpret_cast<const uint32_t*>(blob + L.inv_off);        // read once per triangle hit: global memory
    // The tree over a staged mesh is staged with it; that of a mesh in global memory is read from there too
    // (measured on the 4 968-triangle mesh: where its tree lives makes no difference to the frame time).
    const float4* const nodes = TRI_GLOBAL ? blob : lds;
    const float4* const node0 = nodes + L.node_off[0];
    const float4* const node1 = nodes + L.node_off[1];
    const float4* const node2 = nodes + L.node_off[2];
    const float4* const node3 = nodes + L.node_off[3];
    const uint32_t n_chunks = L.n_sph >> 2, n_sph = L.n_sph, n_box = L.n_box, n_shapes = L.n_shapes;     // whole chunks of four spheres
    // Behind the staged scene every wave has a queue of (lane, leaf) or (lane, box) pairs waiting for their tests,
    // one slot per lane for the closest hit those tests found, and a stack of tree nodes (rt_device.h).
    uint32_t* const queue = reinterpret_cast<uint32_t*>(lds + L.lds_words + (tid >> 6) * (TRI_GLOBAL ? WAVE_LDS_WORDS_GLOBAL : WAVE_LDS_WORDS));
    unsigned long long* const found = reinterpret_cast<unsigned long long*>(queue + QUEUE_ENTRIES);
    uint32_t* const node_stack = queue + QUEUE_ENTRIES + 128u;              // (lane, level, node) triples
    constexpr unsigned long long NOTHING = ~0ull;
    if (HAS_QUEUES) found[lane] = NOTHING;
    // The bounds assume a unit direction; the as-written fuzzy metal leaves it unnormalised.
    const bool cull = (P.flags & 1u) == 0u;
    constexpr uint32_t NONE = 0xffffffffu;

    uint32_t iterations = 0;           // of this lane
    const unsigned long long t_start = (!TRACE && trace_hit) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long c_start = (STATS || COST_CLOCK) ? __builtin_amdgcn_s_memtime() : 0ull;
    // Pool.  A lane that walks the samples of its own pixel one after the other is done when ITS pixel is, and waits
    // for the wave's costliest pixel: with sample ranges cut into slices of 16 a third of the lane-iterations on the
    // mesh scene were spent waiting.  Under the shader's rules a pixel's sum is an integer sum and takes its terms in
    // any order, so the wave's paths — item k*64 + p = sample k of the pixel of lane p — are handed out in item order
    // to whichever lane is free: a lane whose path ends takes the next item (a count over the ended lanes, no
    // atomics), reads that pixel's ray constants (its corner on the image plane, the hash of its index) from a
    // table the wave keeps in LDS, and adds the path's colour to the pixel's sum there.  The home lane writes the
    // sums out at the end.  Waves of tiles that stick out of the frame keep the own-pixel order.
    const uint32_t sample_begin = sample;
    const uint32_t n_items = 64u * (sample_end - sample_begin);
    const bool pool_on = POOL && __ballot(valid) == ~0ull;
    uint32_t item = lane, next_item = 64u, cur_p = lane;
    uint32_t* const pool_acc = node_stack + (TRI_GLOBAL ? NODE_STACK_GLOBAL : NODE_STACK_STAGED);   // [3][64]
    float4* const pool_pix = reinterpret_cast<float4*>(pool_acc + 192u);                             // [64]: {pc.xyz, h1}
    if (POOL) {
        pool_acc[lane] = 0u; pool_acc[64u + lane] = 0u; pool_acc[128u + lane] = 0u;
        pool_pix[lane] = make_float4(pc.x, pc.y, pc.z, __uint_as_float(h1));
    }
    // Primary steps.  59 % of the mesh scene's ray segments are camera rays, and what a camera ray of this wave's 8x8
    // tile can hit is known before the frame starts: the few primitives that touch the pyramid from the eye through
    // the tile (tile_lists_kernel).  A camera ray that takes part in an ordinary iteration pays for the whole scene —
    // every sphere, every box's bounds, the tree walk its bouncing neighbours cause.  So when all live lanes of the
    // wave are about to trace a camera ray, or at least PSTEP_MIN of them, the wave takes a step for those lanes
    // alone (the others wait one step): they test the tile's candidates directly, no bounds, no queues.  The
    // candidates include every primitive whose test could accept such a ray, the closest hit is the same minimum
    // of (distance, index): the same words.  A step with fewer camera rays is an ordinary iteration for everyone.
    uint32_t pl_box = 0u, pl_n = TLIST_NONE, pl_sph0 = 0u;            // (the first 32 spheres' bits are kept at hand: most scenes have no more)
    const uint32_t* pl_sph = nullptr;                                 // TLIST_SPH_WORDS words, more for a scene of many spheres without a mesh
    const uint32_t* pl_leaves = nullptr;
    if (PSTEP && tile_lists) {
        const uint32_t wt = __builtin_amdgcn_readfirstlane((tile_y * 2u + (sub >> 1)) * (2u * P.tiles_x) + tile_x * 2u + (sub & 1u));
        const uint32_t* const rec_ = tile_lists + (size_t)wt * TLIST_WORDS;
        pl_n = rec_[0]; pl_box = rec_[1]; pl_sph = rec_ + 2; pl_leaves = rec_ + TLIST_LEAF0;
        pl_sph0 = rec_[2];
    }
    const uint32_t pl_sph_words = (L.n_sph + 31u) >> 5;               // mask words in use
    const bool have_list = PSTEP && pl_n != TLIST_NONE;
#if RT_PHASE_CLOCKS
    uint32_t phase_clk[4] = { 0u, 0u, 0u, 0u }, phase_t = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
    while (alive) {
        ++iterations;
        bool pstep = false;                                           // wave-uniform
        bool ended_now = false;                                       // POOL: this lane's path ended in this step
        if (PSTEP && have_list) {
            const uint32_t n_all = (uint32_t)__popcll(__ballot(1)), n_first = (uint32_t)__popcll(__ballot(bounce == 0u));
            pstep = n_first == n_all || n_first >= PSTEP_MIN;
        }
        if (!pstep || bounce == 0u) {
        if (STATS && pstep && lane_is_first()) ++n_psteps;
        // ---- closest hit: Scene::hit (scene.h:232-255) /
        //      getClosestShapeIndex + getClosestTriangleIndex (glsl:206-231).
        //      One running (key, id) serves shapes and triangles: shapes come first
        //      and every replacement needs a strictly smaller distance, so a
        //      triangle wins only when strictly closer than the closest shape
        //      (glsl:271) and the first of equal distances stays.
        // "Nothing yet" = (last accepted distance, NONE): a hit at that very distance still wins by its index,
        // anything from `far` on loses (t < FLT_MAX, scene.h:237; distance < 1000, glsl:212).
        // The spheres' scan compares keys only and starts one key further out: `far` itself still loses.
        Best best = { Window<SEM>::far - Window<SEM>::tmin, NONE };
        float lit_t = 0.0f;                                           // LIT: the closest hit's distance, narrowed (Ray::at)
        if (LIT) {
            // Scene::hit, scene.h:232-255, with Sphere::hit as written (scene.h:115-128): distances are doubles and are
            // compared as doubles; the window is 0.001 < t < FLT_MAX with both limits widened (rayTracer.h:117).  The
            // first shape of equal distances stays (strict <): the minimum of (t, index).
            double best_t = __builtin_inf();
            auto consider = [&](double t, uint32_t id) {
                const bool take = 0.001 < t && t < (double)__builtin_bit_cast(float, Window<0>::far) && (t < best_t || (t == best_t && id < best.id));
                best_t = take ? t : best_t;
                best.id = take ? id : best.id;
            };
            for (uint32_t i = 0; i < n_sph; ++i) {
                const float4 sp = sph[i];
                const V3 oc = v3(o.x - sp.x, o.y - sp.y, o.z - sp.z);                     // :116
                const float b = dot_plain(oc, d);                                         // :117
                const float c = dot_plain(oc, oc) - sp.w;                                 // :118 (radius * radius formed at upload)
                const float disc = b * b - c;                                             // :119
                if (!__any(disc >= 0.0f)) continue;                                       // :120
                const double t = -(double)b - __builtin_sqrt((double)disc);               // :123
                consider(disc >= 0.0f ? t : __builtin_inf(), reinterpret_cast<const uint32_t*>(sid)[i]);
            }
            if (HAS_CUBE) {
                for (uint32_t i = 0; i < L.n_cube; ++i) {
                    V3 ip;
                    const float t = hit_cube(cub[4u * i], cub[4u * i + 1u], cub[4u * i + 2u], o, d, ip);
                    consider((double)t, __float_as_uint(cub[4u * i + 3u].x));
                }
            }
            lit_t = best.id == NONE ? 0.0f : (float)best_t;
            if (STATS) n_sph_test += n_sph;
        }
        if (PSTEP && pstep) {
            // the tile's candidate spheres, in upload order (a later one replaces the best only by a strictly smaller key)
            if (!MANY_SPH) {
                for (uint32_t m = pl_sph0; m != 0u; m &= m - 1u) {
                    const uint32_t i = (uint32_t)__builtin_ctz(m);
                    test_sphere<SEM>(sph[i], reinterpret_cast<const uint32_t*>(sid)[i], o, d, best);
                    if (STATS) ++n_sph_test;
                }
            } else {
                for (uint32_t w = 0; w < pl_sph_words; ++w) {
                    for (uint32_t m = w == 0u ? pl_sph0 : pl_sph[w]; m != 0u; m &= m - 1u) {
                        const uint32_t i = 32u * w + (uint32_t)__builtin_ctz(m);
                        test_sphere<SEM>(sph[i], reinterpret_cast<const uint32_t*>(sid)[i], o, d, best);
                        if (STATS) ++n_sph_test;
                    }
                }
            }
        }
        for (uint32_t c = 0; !LIT && !(PSTEP && pstep) && c < n_chunks; ++c) {
            const float4 s0 = sph[4u * c], s1 = sph[4u * c + 1u], s2 = sph[4u * c + 2u], s3 = sph[4u * c + 3u];
            const float4 ids = sid[c];
            test_sphere<SEM>(s0, __float_as_uint(ids.x), o, d, best);
            test_sphere<SEM>(s1, __float_as_uint(ids.y), o, d, best);
            test_sphere<SEM>(s2, __float_as_uint(ids.z), o, d, best);
            test_sphere<SEM>(s3, __float_as_uint(ids.w), o, d, best);
        }
        // the one to three spheres behind the last whole chunk, one at a time: the padding of their chunk is not tested
        for (uint32_t i = 4u * n_chunks; !LIT && !(PSTEP && pstep) && i < n_sph; ++i)
            test_sphere<SEM>(sph[i], reinterpret_cast<const uint32_t*>(sid)[i], o, d, best);
        if (STATS && !LIT && !(PSTEP && pstep)) n_sph_test += n_sph;
        if (!LIT && (HAS_QUEUES || HAS_CUBE)) best.key = (best.id == NONE) ? Window<SEM>::far - 1u - Window<SEM>::tmin : best.key;
        RT_PHASE_MARK(0)
        // The lanes still tracing, and this lane's place among them: they share the queued tests.
        uint32_t n_live = 64u, lane_rank = lane;
        if (HAS_QUEUES) {
            const unsigned long long live = __ballot(1);
            n_live = (uint32_t)__popcll(live);
            lane_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live, 0u));
        }
        const bool spread = HAS_QUEUES && n_live >= 16u;             // fewer live lanes: every test as a wave
        // Boxes.  The full test is long (three reciprocals, three candidate points); a lane needs it only for a box
        // whose bounds its ray touches (a slab test).  Those (lane, box) pairs are queued and tested one lane per pair.
        if (HAS_QUEUES && PSTEP && pstep) {
            // the tile's candidate boxes: the exact test by every lane of the step, no bounds first
            for (uint32_t m = pl_box; m != 0u; m &= m - 1u) {
                const uint32_t i = (uint32_t)__builtin_ctz(m);
                const float4 bx = box[3u * i], bx1 = box[3u * i + 1u];
                take_if_closer(best, __float_as_uint(hit_box(bx, bx1.y, bx1.z, o, d)) - Window<SEM>::tmin, __float_as_uint(bx1.x));
                if (STATS) ++n_box_test;
            }
        } else if (SPREAD_BOXES && spread && cull) {
            uint32_t bq_len = 0u;                                 // wave-uniform
            // 1 / d for the slab test of the bounds: hardware reciprocals (within an ulp; a zero component gives an
            // infinity) — the bounds carry the slack for that, and decide nothing but who runs the exact test.
            const V3 inv_d = v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
            auto box_pass = [&](uint32_t count) {
                const bool work = lane_rank < count;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t ent = work ? queue[lane_rank] : (lane << 24);
                const uint32_t src = ent >> 24, i = ent & 0xffffffu;
                const int from = (int)(src << 2);
                const V3 ro = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.z))));
                const V3 rd = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.z))));
                if (work) {
                    const float4 bx = box[3u * i], bx1 = box[3u * i + 1u];
                    const uint32_t ckey = __float_as_uint(hit_box(bx, bx1.y, bx1.z, ro, rd)) - Window<SEM>::tmin;
                    if (ckey < Window<SEM>::far - Window<SEM>::tmin)
                        atomicMin(&found[src], ((unsigned long long)ckey << 32) | __float_as_uint(bx1.x));
                    if (STATS) ++n_box_test;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            };
            for (uint32_t i = 0; i < n_box; ++i) {
                float4 bc = box[3u * i], bh = box[3u * i + 2u];
                asm volatile("" : "+v"(bc.w), "+v"(bh.w));                 // whole 16-byte LDS reads: a 12-byte one costs twice the LDS cycles
                const bool touch = box_bound_hit<SEM>(bc, bh, o, inv_d, best.key);
                if (STATS) ++n_box_bound;
                const unsigned long long touching = __ballot(touch);
                if (touching == 0ull) continue;
                const uint32_t n_touching = (uint32_t)__popcll(touching);
                if (RT_BOX_DIRECT_DEN * n_touching >= RT_BOX_DIRECT_NUM * n_live) {
                    // Nearly every lane wants this box (a floor): the wave tests it on the spot, without the queue.
                    const float4 bx1 = box[3u * i + 1u];
                    take_if_closer(best, __float_as_uint(hit_box(bc, bx1.y, bx1.z, o, d)) - Window<SEM>::tmin, __float_as_uint(bx1.x));
                    if (STATS) ++n_box_test;
                    continue;
                }
                if (bq_len + n_touching > n_live) { box_pass(bq_len); bq_len = 0u; }      // a pass takes one pair per live lane
                const uint32_t at = bq_len + __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
                if (touch) queue[at] = (lane << 24) | i;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                bq_len += n_touching;
            }
            if (bq_len != 0u) box_pass(bq_len);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const unsigned long long f = found[lane];
            if (f != NOTHING) {
                take_if_closer(best, (uint32_t)(f >> 32), (uint32_t)f);
                found[lane] = NOTHING;
            }
        } else if (HAS_QUEUES) {                                  // a scene with boxes never runs the instance without queues
            for (uint32_t i = 0; i < n_box; ++i) {
                const float4 bx = box[3u * i];
                const float4 bx1 = box[3u * i + 1u];
                take_if_closer(best, __float_as_uint(hit_box(bx, bx1.y, bx1.z, o, d)) - Window<SEM>::tmin, __float_as_uint(bx1.x));
            }
            if (STATS) n_box_test += n_box;
        }
        if (!LIT && HAS_CUBE) {
            for (uint32_t i = 0; i < L.n_cube; ++i) {
                V3 ip;
                const float t = hit_cube(cub[4u * i], cub[4u * i + 1u], cub[4u * i + 2u], o, d, ip);
                take_if_closer(best, __float_as_uint(t) - Window<SEM>::tmin, __float_as_uint(cub[4u * i + 3u].x));
            }
        }
        RT_PHASE_MARK(1)
        RT_PHASE_MARK3(0)
        // Triangles: walk the four-level bounding-sphere tree with the whole wave; a subtree is entered when the
        // ray of any live lane can touch its sphere.  At a leaf only the lanes whose rays touch it have any business
        // with its triangles — typically a few of the 64, since bounce rays share little — so they do not test
        // them on the spot (which would occupy the whole wave four times per leaf) but queue (lane, leaf).  Whenever
        // the queue holds a pass worth of pairs, the wave tests them side by side: four lanes per pair, one per
        // triangle, each fetching the ray of the pair's lane across the wave.  A crossing goes to the slot of the
        // ray's lane by an atomic minimum on (key, id) — the same order-independent minimum as everywhere else —
        // and every lane folds its slot into its closest hit when the walk is over.
        if (HAS_QUEUES && PSTEP && pstep) {
            // the tile's candidate leaves: their triangles by every lane of the step
            const uint32_t leaf_shift = L.leaf_shift, leaf_size = 1u << leaf_shift;
            for (uint32_t k = 0; k < pl_n; ++k) {
                const uint32_t first = pl_leaves[k] << leaf_shift;
                for (uint32_t j = 0; j < leaf_size; ++j) {
                    const float4* const t4 = tri + 4u * (first + j);
                    float4 w0 = t4[0], w1 = t4[1], w2 = t4[2], w3 = t4[3];
                    if (!TRI_GLOBAL) asm volatile("" : "+v"(w2.w), "+v"(w3.w));          // whole 16-byte LDS reads
                    test_triangle<SEM>(w0, w1, w2, w3, n_shapes, o, d, best);
                }
                if (STATS) n_tri += leaf_size;
            }
        } else if (HAS_QUEUES) {
            const uint32_t leaf_shift = L.leaf_shift, leaf_size = 1u << leaf_shift;     // triangles per leaf: chosen per mesh
            const uint32_t per_pass = n_live >> leaf_shift;       // pairs one pass takes: one lane per triangle of the leaf
            uint32_t q_head = 0u, q_len = 0u;                     // wave-uniform
            auto pass = [&](uint32_t count) {
                RT_PHASE_MARK3(1)
                const uint32_t e = lane_rank >> leaf_shift, k = lane_rank & (leaf_size - 1u);
                const bool work = e < count;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t ent = work ? queue[(q_head + e) & (QUEUE_ENTRIES - 1u)] : (lane << 24);
                const uint32_t src = ent >> 24, leaf = ent & 0xffffffu;
                // the pair's ray: a read across the wave needs its source lane enabled, so every lane takes part
                const int from = (int)(src << 2);
                const V3 ro = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.z))));
                const V3 rd = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.z))));
                if (work) {
                    const float4* const t4 = tri + 4u * ((leaf << leaf_shift) + k);
                    float4 w0 = t4[0], w1 = t4[1], w2 = t4[2], w3 = t4[3];
                    if (!TRI_GLOBAL) asm volatile("" : "+v"(w0.w), "+v"(w2.w), "+v"(w3.w));    // whole 16-byte LDS reads (see the direct path)
                    uint32_t ckey, cid;
                    const bool inside = triangle_candidate<SEM>(w0, w1, w2, w3, n_shapes, ro, rd, ckey, cid);
                    if (inside && ckey < Window<SEM>::far - Window<SEM>::tmin)
                        atomicMin(&found[src], ((unsigned long long)ckey << 32) | cid);
                    if (STATS) ++n_tri;
                }
                RT_PHASE_MARK3(3)
            };
            // Pairs (lane, leaf) go to the triangle queue; full passes are run at once.
            auto push_leaves = [&](bool touch, uint32_t src, uint32_t leaf) {
                const unsigned long long touching = __ballot(touch);
                const uint32_t at = q_head + q_len +
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
                if (touch) queue[at & (QUEUE_ENTRIES - 1u)] = (src << 24) | leaf;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                q_len += (uint32_t)__popcll(touching);
                while (q_len >= per_pass) { pass(per_pass); q_head += per_pass; q_len -= per_pass; }
            };
            // The same further up: a node that few lanes touch is put on the wave's stack as (lane, level, node), and a
            // pass gives NODE_LPT lanes to each triple on top of the stack, NODE_CPL children each: they test the
            // child's bound against the triple's ray.  A touched leaf goes to the triangle queue, a touched inner node back onto the
            // stack, where the next pass finds it first — depth first, which bounds the stack (see rt_device.h).
            // A staged mesh is small: only its level-1 nodes are worth queueing.
            constexpr bool DEEP = TRI_GLOBAL;
            const uint32_t node_pass = n_live / NODE_LPT;         // triples one pass takes: NODE_LPT lanes each
            uint32_t ns_len = 0u;                                  // wave-uniform
            auto push_nodes = [&](bool touch, uint32_t src, uint32_t level, uint32_t node) {
                const unsigned long long touching = __ballot(touch);
                const uint32_t at = ns_len +
                    __builtin_amdgcn_mbcnt_hi((uint32_t)(touching >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)touching, 0u));
                if (touch) node_stack[at] = (src << 26) | (level << 24) | node;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                ns_len += (uint32_t)__popcll(touching);
            };
            auto bound_pass = [&](uint32_t count) {
                RT_PHASE_MARK3(1)
                const uint32_t e = lane_rank / NODE_LPT, c = (lane_rank % NODE_LPT) * NODE_CPL;
                const bool work = e < count;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t ent = work ? node_stack[ns_len - count + e] : ((lane << 26) | (1u << 24));
                ns_len -= count;                                   // popped: what the pass pushes may take their place
                const uint32_t src = ent >> 26, level = DEEP ? (ent >> 24) & 3u : 1u, child0 = TREE_FAN * (ent & 0xffffffu) + c;
                const int from = (int)(src << 2);
                const V3 ro = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(o.z))));
                const V3 rd = v3(__uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.x))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.y))),
                                 __uint_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_uint(d.z))));
                const uint32_t rkey = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)best.key);
                // the children live one level down
                const uint32_t n_children = level == 1u ? L.node_n[0] : (level == 2u ? L.node_n[1] : L.node_n[2]);
                const uint32_t children_at = level == 1u ? L.node_off[0] : (level == 2u ? L.node_off[1] : L.node_off[2]);
#pragma unroll
                for (uint32_t cc = 0; cc < NODE_CPL; ++cc) {
                    const uint32_t child = child0 + cc;
                    const bool valid = work && child < n_children;
                    const bool touch = valid && (!cull || bound_hit<SEM>(nodes[children_at + (valid ? child : 0u)], ro, rd, rkey));
                    if (STATS && valid) ++n_node;
                    if (DEEP) push_nodes(touch && level != 1u, src, level - 1u, child);
                    RT_PHASE_MARK3(2)
                    push_leaves(touch && level == 1u, src, child);
                }
            };
            // -> the node was put on the stack (few lanes touch it) and full passes were run
            auto queue_node = [&](bool touch, uint32_t n_touching, uint32_t level, uint32_t node) -> bool {
                if (!spread || n_touching >= SPREAD_BELOW) return false;
                push_nodes(touch, lane, level, node);
                while (ns_len >= node_pass) bound_pass(node_pass);
                return true;
            };
            // The level-3 nodes to walk: all of them, one after the other — or, for a large mesh, which has levels above
            // the third (rt_device.h), those that the wave reaches going down through these levels together, depth first: a
            // node is entered when the ray of any live lane can touch it, and its children wait on a small stack the wave
            // shares.
            const uint32_t top = DEEP_TREE ? L.tree_levels - 1u : 3u;
            uint32_t* const upper = queue + (WAVE_LDS_WORDS_GLOBAL * 4u - UPPER_STACK_ENTRIES);
            uint32_t top_next = 0u, un = 0u;                       // wave-uniform
            for (;;) {
                uint32_t i3 = 0u;
                if (top == 3u) {
                    if (top_next >= L.node_n[3]) break;
                    i3 = top_next++;
                } else {
                    bool got = false;
                    for (;;) {
                        if (un == 0u) {
                            if (top_next >= L.node_n[top]) break;
                            if (lane_is_first()) upper[0] = (top << 24) | top_next;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            ++top_next; un = 1u;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const uint32_t ent = __builtin_amdgcn_readfirstlane(upper[un - 1u]);
                        --un;
                        const uint32_t lvl = ent >> 24, node = ent & 0xffffffu;
                        if (lvl == 3u) { i3 = node; got = true; break; }
                        if (STATS) ++n_node;
                        const bool touch_u = !cull || bound_hit<SEM>(nodes[L.node_off[lvl] + node], o, d, best.key);
                        if (__ballot(touch_u) == 0ull) continue;
                        const uint32_t c0 = TREE_FAN * node, c1 = min(c0 + TREE_FAN, L.node_n[lvl - 1u]);
                        if (lane_is_first())
                            for (uint32_t c = c0; c < c1; ++c) upper[un + (c1 - 1u - c)] = ((lvl - 1u) << 24) | c;   // the lowest child on top
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        un += c1 - c0;
                    }
                    if (!got) break;
                }
                if (STATS) ++n_node;
                const bool touch3 = !cull || bound_hit<SEM>(node3[i3], o, d, best.key);
                const unsigned long long touching3 = __ballot(touch3);
                if (touching3 == 0ull) continue;
                if (STATS) { if (lane_is_first()) ++n_tree_entries; n_tree_lanes += touch3 ? 1u : 0u; }
                if (DEEP && queue_node(touch3, (uint32_t)__popcll(touching3), 3u, i3)) continue;
                const uint32_t e2 = min(TREE_FAN * i3 + TREE_FAN, L.node_n[2]);
                for (uint32_t i2 = TREE_FAN * i3; i2 < e2; ++i2) {
                    if (STATS) ++n_node;
                    const bool touch2 = !cull || bound_hit<SEM>(node2[i2], o, d, best.key);
                    const unsigned long long touching2 = __ballot(touch2);
                    if (touching2 == 0ull) continue;
                    if (DEEP && queue_node(touch2, (uint32_t)__popcll(touching2), 2u, i2)) continue;
                    const uint32_t e1 = min(TREE_FAN * i2 + TREE_FAN, L.node_n[1]);
                    for (uint32_t i1 = TREE_FAN * i2; i1 < e1; ++i1) {
                        if (STATS) ++n_node;
                        const bool touch1 = !cull || bound_hit<SEM>(node1[i1], o, d, best.key);
                        const unsigned long long touching1 = __ballot(touch1);
                        if (touching1 == 0ull) continue;
                        if (queue_node(touch1, (uint32_t)__popcll(touching1), 1u, i1)) continue;
                        const uint32_t e0 = min(TREE_FAN * i1 + TREE_FAN, L.node_n[0]);
                        for (uint32_t i0 = TREE_FAN * i1; i0 < e0; ++i0) {
                            if (STATS) ++n_node;
                            const bool touch = !cull || bound_hit<SEM>(node0[i0], o, d, best.key);
                            const unsigned long long touching = __ballot(touch);
                            if (touching == 0ull) continue;
                            if (spread) { push_leaves(touch, lane, i0); continue; }
                            for (uint32_t k = 0; k < leaf_size; ++k) {
                                const uint32_t ti = (i0 << leaf_shift) + k;
                                float4 w0 = tri[4u * ti], w1 = tri[4u * ti + 1u], w2 = tri[4u * ti + 2u], w3 = tri[4u * ti + 3u];
                                // Only xyz of two of the words is used, which would let the compiler
                                // fetch 12 bytes each — and a 12-byte LDS read costs twice the LDS
                                // cycles of a 16-byte one.
                                if (!TRI_GLOBAL) asm volatile("" : "+v"(w2.w), "+v"(w3.w));
                                test_triangle<SEM>(w0, w1, w2, w3, n_shapes, o, d, best);
                            }
                            if (STATS) n_tri += leaf_size;
                        }
                    }
                }
            }
            if (spread) {
                while (ns_len != 0u) bound_pass(min(ns_len, node_pass));
                while (q_len != 0u) { const uint32_t c = min(q_len, per_pass); pass(c); q_head += c; q_len -= c; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long f = found[lane];
                if (f != NOTHING) {
                    take_if_closer(best, (uint32_t)(f >> 32), (uint32_t)f);
                    found[lane] = NOTHING;
                }
            }
        }
        RT_PHASE_MARK(2)
        RT_PHASE_MARK3(1)
        RT_PHASE_MARK2(0)
        RT_PHASE_MARK4(pstep ? 0 : 1)
        const uint32_t best_key = best.key, best_id = best.id;
        if (STATS) { ++n_segments; }

        // ---- the surface
        // A shape's record carries what the event needs in the form the event uses it (rt_device.h): its class
        // (0 ends the path and returns throughput x colour x `emit`; 1 Lambertian, 2 metal, 3 dielectric, 4 debug)
        // and `emit` = 1 for a light (LightSource::emit / glsl:278-280) and for a material the shader does not know
        // (whose colour is stored as 1: glsl:253,282-284 returns the throughput), 0 otherwise — so that the colour
        // a path returns is a product, not a chain of cases.  A miss reads the record behind the shapes' (class 0,
        // emit 0).  Colours are taken to be finite (0 x colour = 0).
        const float t_hit = LIT ? lit_t : __uint_as_float(best_key + Window<SEM>::tmin);
        const V3 p = madd_as<LIT>(t_hit, d, o);                       // Ray::at / glsl:239
        const bool is_tri = HAS_QUEUES && (best_id - n_shapes) < L.n_tris;       // a miss (NONE) wraps far beyond
        const uint32_t rid = min(best_id, n_shapes);
        const float4 r0 = rec[4u * rid], r1 = rec[4u * rid + 1u];
        float4 r2 = rec[4u * rid + 2u];
        asm volatile("" : "+v"(r2.w));                                // one 16-byte LDS read
        const uint32_t tag = __float_as_uint(r1.w);
        uint32_t cls = tag >> 24;
        const uint32_t stype = (tag >> 16) & 0xffu;
        V3 n, att = xyz(r1);
        float emit = r2.y, fuzz = r2.x, glass_eta = 0.0f, glass_r0 = 0.0f;
        // An instance without boxes, triangles and cubes is launched only for scenes in which every shape that can
        // be hit is a sphere.
        if ((!HAS_QUEUES && !HAS_CUBE) || stype == 0u) {
            const V3 pc_ = v3(p.x - r0.x, p.y - r0.y, p.z - r0.z);
            if (LIT) n = normalize_plain(pc_);                        // scene.h:130-132 as written
            else {
            n = sphere_normal(pc_, r2.z);                             // scene.h:130-132
            // A fuzzy metal's direction left as written is not unit, and the sphere test, which assumes it is, then
            // returns a point off the sphere: the radius is no first guess of that length.  (A branch on the launch's
            // flags; the empty statement keeps it one.)
            if (SEM == 0 && !cull) { asm volatile(""); n = normalize(pc_); }
            }
        }
        else if (HAS_QUEUES && stype == 1u) { const float4 r3 = rec[4u * rid + 3u]; n = box_normal(xyz(r0), r2.z, r3.z, r3.w, p); }
        else if (HAS_CUBE && stype == 3u) {
            // rotation^T * normalAt(inPoint), scene.h:150-164,196: the row of the rotation that
            // belongs to the largest |coordinate| of the point in the cube's frame (first of
            // equals); the sign there is taken from an absolute value, so it is always +.
            const uint32_t ci = 4u * __float_as_uint(r2.w);
            const float4 c0 = cub[ci], c1 = cub[ci + 1u], c2 = cub[ci + 2u];
            V3 ip;
            (void)hit_cube(c0, c1, c2, o, d, ip);
            const float ax = __builtin_fabsf(ip.x), ay = __builtin_fabsf(ip.y), az = __builtin_fabsf(ip.z);
            const bool y_gt = ay > ax;
            const float m1 = y_gt ? ay : ax;
            const bool z_gt = az > m1;
            n = z_gt ? xyz(c2) : (y_gt ? xyz(c1) : xyz(c0));
        }
        else n = v3(1.0f, 1.0f, 1.0f);                                // glsl:203
        if (HAS_GLASS && cls == 3u) { const float4 r3 = rec[4u * rid + 3u]; glass_eta = r3.x; glass_r0 = r3.y; }
        if (is_tri) {
            V3 a = xyz(tri[4u * slot_of[best_id - n_shapes] + 1u]);
            n = a * sign_(-dot(d, a));                                // glsl:165-167,272
            att = v3(P.tri_albedo[0], P.tri_albedo[1], P.tri_albedo[2]);
            if (HAS_TCOL && L.tcol_off != 0u) att = xyz(blob[L.tcol_off + (best_id - n_shapes)]);
            cls = 1u; emit = 0.0f; fuzz = 0.0f;
        }
        if (TRACE) {
            size_t at = (size_t)pixel * P.max_depth + bounce;
            trace_hit[at] = (int32_t)best_id;
            trace_mat[at] = best_id == NONE ? NONE : (is_tri ? 1u : (tag & 0xffffu));
        }

        RT_PHASE_MARK2(1)
        const bool last = bounce + 1u >= P.max_depth;                 // rayTracer.h:113-115 / glsl:263,287
        bool raw_dir = false;                                         // new direction is used as it is
        V3 v = v3(0, 0, 1);                                           // new direction before normalisation
        const bool diffuse = cls == 1u, metal = cls == 2u, glass = HAS_GLASS && cls == 3u;
        if (STATS) { n_diffuse += diffuse ? 1u : 0u; n_specular += (metal || glass) ? 1u : 0u; }
        const V3 ta = thr * att;                                      // a light returns it, a bounce carries it on
        V3 color = ta * emit;                                         // what the path returns if it ends here
        if (L.has_debug != 0u) {
            // a branch on the scene, not a select on every lane of every scene: the empty statement keeps the compiler
            // from turning the block into three multiplications and three selects that always run
            asm volatile("");
            if (cls == 4u) color = (SEM == 0) ? thr * n : n;          // glsl:248-252
        }
        bool ended = !(cls - 1u < (HAS_GLASS ? 3u : 2u)) || last;

        // Two draws per lane: the bounce's pair, or — for a lane that starts a
        // sample — the jitter pair of the new sample.
        uint32_t new_item = 0u;
        if (POOL) {
            const unsigned long long em = __ballot(ended);
            new_item = pool_on ? next_item + __builtin_amdgcn_mbcnt_hi((uint32_t)(em >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)em, 0u)) : item + 64u;
            ended_now = ended;                                       // next_item moves on below, in every lane of the wave
        }
        if (ended) {
            if (STATS) ++n_terminal;
            if (SEM == 0) {
                acc0 = __float_as_uint(__uint_as_float(acc0) + color.x);   // rayTracer.h:101
                acc1 = __float_as_uint(__uint_as_float(acc1) + color.y);
                acc2 = __float_as_uint(__uint_as_float(acc2) + color.z);
            } else if (POOL) {
                // glsl:301-307.  Most paths end in the dark (a miss, the last segment): they add nothing.
                const uint32_t q0 = quantise(color.x), q1 = quantise(color.y), q2 = quantise(color.z);
                if ((q0 | q1 | q2) != 0u) {
                    atomicAdd(&pool_acc[cur_p], q0);
                    atomicAdd(&pool_acc[64u + cur_p], q1);
                    atomicAdd(&pool_acc[128u + cur_p], q2);
                }
            } else {
                acc0 += quantise(color.x);                                 // glsl:301-307
                acc1 += quantise(color.y);
                acc2 += quantise(color.z);
            }
            if (POOL) {
                item = new_item;
                cur_p = item & 63u;
                sample = sample_begin + (item >> 6);
                alive = item < n_items;
                const float4 pp = pool_pix[cur_p];                         // the constants of the pixel the lane goes on with
                pcur = xyz(pp);
                key = __float_as_uint(pp.w) + sample * SAMPLE_STEP;
            } else {
                ++sample;
                alive = sample < sample_end;
                key += SAMPLE_STEP;                                        // path_key(h1, sample)
            }
            if (STATS && alive) ++n_samples;
        }
        if (POOL && RNG_GLSL) {
            // the shader's seed needs the pixel's coordinates: from the pixel's home lane (every lane reads: a read
            // across the wave cannot sit in a branch; lanes in mid-path read their own pixel's again)
            const int from = (int)(cur_p << 2);
            const uint32_t ngx = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)gx), ngy = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)gy);
            cgx = ended ? ngx : cgx; cgy = ended ? ngy : cgy;
        }
        float u1, u2;                                                 // raw draws
        if (RNG_GLSL) {
            // The shader draws only where its program does: two values for a new camera ray,
            // two for a diffuse bounce (angle, then r: glsl:66-67), two for a fuzzy reflection
            // of the CPU rules, none for a mirror.
            if (ended) gs = glsl_seed(sample);
            V3 g = gs;
            u1 = glsl_random(g) * (1.0f / RAW); u2 = glsl_random(g) * (1.0f / RAW);      // raw, like the hash's draws
            const bool uses = (ended && P.sample_mode == 0u) || (!ended && (diffuse || glass || (SEM == 0 && metal && fuzz != 0.0f)));
            gs = v3(uses ? g.x : gs.x, uses ? g.y : gs.y, uses ? g.z : gs.z);
        } else {
            draw_pair(key, ended ? 0u : bounce_pair(bounce), u1, u2);
        }

        // randomUnitVector(): the direct sphere point, or — RT_FLAG_UNIT_GAUSS — the reference's normalised Gaussians,
        // whose two further draws are the event's second pair (called after ++bounce: the event's first pair is
        // 2*bounce - 1).
        auto sphere_point = [&](float a1, float a2) -> V3 {
            if (HAS_GAUSS && (P.flags & 4u) != 0u) {
                float a3, a4;
                draw_pair(key, 2u * bounce, a3, a4);
                return unit_vector_gauss(a1, a2, a3, a4);
            }
            return unit_vector(a1, a2);
        };
        if (ended) {
            o = camera_origin(sample, u1, u2);
            v = o - cam_pos;                                          // rayTracer.h:100 / glsl:297
            thr = v3(1, 1, 1);
            bounce = 0;
        } else {
            ++bounce;
            o = p;
            thr = ta;                                                 // scene.h:56 (0.6*albedo, formed at upload), :79; glsl:241,246,275
            if (diffuse) {
                if (SEM == 0) {
                    v = madd_as<LIT>(1.0001f, n, sphere_point(u1, u2));   // Lambertian::scatter, scene.h:51-58; utils.h:48-50
                } else {
                    v = hemisphere_glsl(n, u1, u2);                   // glsl:240,274
                    raw_dir = true;
                }
            } else if (glass) {
                // Dielectric (no counterpart in the reference): at the surface the ray is mirrored with
                // Schlick's probability and refracted otherwise; the sphere test knows only the near
                // root, so a refracted ray leaves the sphere without a second event.
                const float cosi = -dot_as<LIT>(d, n);
                const float x = 1.0f - cosi, x2 = x * x;
                const float refl = fma_(1.0f - glass_r0, (x2 * x2) * x, glass_r0);
                const V3 perp = madd_as<LIT>(cosi, n, d) * glass_eta;
                const float one_less = 1.0f - dot_as<LIT>(perp, perp);
                const float par = -(LIT ? __builtin_sqrtf(__builtin_fabsf(one_less)) : sqrt_rn(__builtin_fabsf(one_less)));
                const V3 refr = madd_as<LIT>(par, n, perp);
                const V3 mirr = madd_as<LIT>(-(2.0f * dot_as<LIT>(d, n)), n, d);
                const bool mirror = u1 * RAW < refl;
                v = v3(mirror ? mirr.x : refr.x, mirror ? mirr.y : refr.y, mirror ? mirr.z : refr.z);
            } else if (SEM == 0) {
                // Metal::scatter, scene.h:73-81; reflect, utils.h:62-64
                v = madd_as<LIT>(-(2.0f * dot_as<LIT>(n, d)), n, d);
                if (fuzz != 0.0f) v = madd_as<LIT>(fuzz, sphere_point(u1, u2), v);
                raw_dir = (P.flags & 1u) != 0u;
                if (!(dot_as<LIT>(v, n) > 0.0f)) {
                    // absorbed: emit() = 0 (scene.h:80,83-85).  The path ends here after all;
                    // this rare case starts the next sample with draws of its own.
                    if (STATS) ++n_terminal;
                    if (SEM == 0) {
                        acc0 = __float_as_uint(__uint_as_float(acc0) + 0.0f);
                        acc1 = __float_as_uint(__uint_as_float(acc1) + 0.0f);
                        acc2 = __float_as_uint(__uint_as_float(acc2) + 0.0f);
                    }
                    ++sample;
                    alive = sample < sample_end;
                    key += SAMPLE_STEP;
                    if (STATS && alive) ++n_samples;
                    float j1, j2;
                    start_draws(sample, key, gs, j1, j2);
                    o = camera_origin(sample, j1, j2);
                    v = o - cam_pos;
                    thr = v3(1, 1, 1);
                    bounce = 0;
                    raw_dir = false;
                }
            } else {
                v = madd(-(2.0f * dot(d, n)), n, d);                  // glsl:243-247; reflect :82-84
            }
        }
        RT_PHASE_MARK2(2)
        // normalize(v), or v as it is: one select on the factor instead of three on the components (v * 1 = v)
        if (LIT) {
            const V3 unit = normalize_plain(v);                       // the Ray constructor, scene.h:14-16
            d = v3(raw_dir ? v.x : unit.x, raw_dir ? v.y : unit.y, raw_dir ? v.z : unit.z);
        } else {
            const float inv_len = raw_dir ? 1.0f : rsqrt_nr(dot(v, v));
            d = v * inv_len;
        }
        RT_PHASE_MARK(3)
        RT_PHASE_MARK2(3)
        }   // lanes of this step
        RT_PHASE_MARK4(pstep ? 0 : 2)
#if RT_PHASE_CLOCKS == 4
        if (pstep) phase_clk[3] += 16u;
#endif
        // The pool's next item is one number for the wave: every lane keeps it, also those a primary step left out.
        if (POOL) next_item += (uint32_t)__popcll(__ballot(ended_now));
    }

    if (!TRACE && valid) {
        if (POOL) { acc0 += pool_acc[lane]; acc1 += pool_acc[64u + lane]; acc2 += pool_acc[128u + lane]; }
        if (CAN_SLICE && n_slices != 1u) {
            atomicAdd(acc_ptr, acc0); atomicAdd(acc_ptr + 1, acc1); atomicAdd(acc_ptr + 2, acc2);   // integer sums: any order
        } else {
            uint4 v; v.x = acc0; v.y = acc1; v.z = acc2; v.w = 0u;
            *reinterpret_cast<uint4*>(acc_ptr) = v;
            if (P.resolve_out != nullptr) {
                // the frame ends with this launch: its resolve here, from the sums still in registers
                const uint32_t oy = P.resolve_flip ? P.height - 1u - gy : gy;
                *reinterpret_cast<float4*>(P.resolve_out + 4u * ((size_t)oy * P.width + gx)) =
                    resolve_pixel<SEM>(acc0, acc1, acc2, P.resolve_samples, P.resolve_gamma != 0u);
            }
        }
    }
    // What this tile cost, for the next frame's launch order: the longest of its four waves' loops.
    if (!TRACE && tile_cost && lane == 0u)
        atomicMax(&tile_cost[tile], (COST_CLOCK ? (uint32_t)((__builtin_amdgcn_s_memtime() - c_start) >> 8) + 1u : iterations) * n_slices);
    if (!TRACE && trace_hit && lane == 0u) {
        // Diagnostic timeline: start and end of this wave tile's loop (100 MHz ticks), where it ran, how long its loop was.
        int32_t* rec4 = trace_hit + 4u * q;
        rec4[0] = (int32_t)(uint32_t)t_start;
        rec4[1] = (int32_t)(uint32_t)__builtin_amdgcn_s_memrealtime();
        rec4[2] = (int32_t)((((uint32_t)(__builtin_amdgcn_s_memtime() - c_start)) >> 4) |                       // shader clocks / 16
                            (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 28));                      // XCC_ID
        rec4[3] = (int32_t)iterations;
#if RT_PHASE_CLOCKS
        rec4[0] = (int32_t)(phase_clk[0] >> 4); rec4[1] = (int32_t)(phase_clk[1] >> 4);
        rec4[2] = (int32_t)(phase_clk[2] >> 4) | (rec4[2] & (int32_t)0xf0000000); rec4[3] = (int32_t)(phase_clk[3] >> 4);
#endif
    }
    if (STATS) {
        // The wave ran as long as its longest lane: the maximum of the lanes' own counts.
        uint32_t longest = iterations;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) longest = max(longest, (uint32_t)__shfl_down(longest, off, 64));
        n_iterations += longest;
    }
    }

    if (STATS) {
        // Sum over the wave with cross-lane shuffles, one atomic per wave and counter.
        uint32_t c[ST_COUNT] = { n_samples, n_segments, n_diffuse, n_specular, n_terminal, n_tri, n_node, (lane == 0u) ? n_iterations : 0u,
                                 (lane == 0u && (q & 3u) == 0u) ? 1u : 0u, n_box_test, n_box_bound, n_sph_test, n_psteps, n_tree_entries, n_tree_lanes };
#pragma unroll
        for (int k = 0; k < ST_COUNT; ++k) {
            unsigned long long v = c[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&stats[k], v);
        }
    }
}

// Turns the per-tile costs of the frame just traced into the plan of the next one — one entry per
// workgroup, see trace_kernel — and clears the costs.  n tiles, n + extra entries.
//   * A tile whose cost exceeds `share` of what a wave slot would carry if all work were spread
//     evenly (sum of costs / 2048: four waves per tile, 8192 slots) is cut into up to max_slices
//     slices; the threshold is raised until the additional entries fit into `extra`.  These go first.
//   * Tiles that cost more than twice the mean but are not cut (max_slices = 1: float sums, whose
//     order of addition is part of the result) go next; when tiles are being cut, so do the uncut
//     ones above early_share of the cutting threshold.
//   * All other tiles keep the built-in order (bottom rows first), which interleaves long and short
//     tiles on every CU — measured faster than a full longest-first sort, whose tail of short tiles
//     is all start-up latency.
// One workgroup.
__global__ __launch_bounds__(1024)
void tile_plan_kernel(uint32_t* __restrict__ cost, uint32_t* __restrict__ work, uint32_t n, uint32_t extra, uint32_t max_slices, float share, float early_share)
{
    __shared__ unsigned long long scan[1024];
    __shared__ unsigned long long total;
    __shared__ uint32_t n_extra;
    const uint32_t tid = threadIdx.x;
    if (tid == 0u) total = 0ull;
    __syncthreads();
    unsigned long long sum = 0ull;
    for (uint32_t i = tid; i < n; i += 1024u) sum += cost[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((tid & 63u) == 0u) atomicAdd(&total, sum);
    __syncthreads();
    uint32_t limit = (uint32_t)min(0xffffffffull, 2ull * total / n);            // twice the mean
    // Position j of the built-in order is tile n-1-j; thread t owns positions [t*chunk, (t+1)*chunk).
    const uint32_t chunk = (n + 1023u) / 1024u;
    const uint32_t j0 = min(n, tid * chunk), j1 = min(n, j0 + chunk);

    float target = fmaxf(1.0f, (float)total * (share / 2048.0f));
    auto slices_of = [&](uint32_t c) -> uint32_t {
        const uint32_t k = (uint32_t)fminf(ceilf((float)c / target), (float)max_slices);
        return max(k, 1u);
    };
    if (max_slices > 1u && extra > 0u) {
        for (int round = 0; round < 24; ++round) {
            __syncthreads();
            if (tid == 0u) n_extra = 0u;
            __syncthreads();
            uint32_t mine = 0u;
            for (uint32_t j = j0; j < j1; ++j) mine += slices_of(cost[n - 1u - j]) - 1u;
            if (mine) atomicAdd(&n_extra, mine);
            __syncthreads();
            if (n_extra <= extra) break;                 // the same value in every thread
            target *= 1.5f;
        }
        __syncthreads();
        if (n_extra > extra) max_slices = 1u;
    } else {
        max_slices = 1u;
    }

    // With cutting on, the longest pieces of work left are the tiles just under the cutting threshold; those above
    // `early_share` of it follow the slices directly, or they would start late and end the launch alone.
    if (max_slices > 1u && n_extra > 0u) limit = min(limit, (uint32_t)fminf(target * early_share, 4294967040.0f));

    // Entries per class, packed 21 bits each: cut tiles' slices, uncut expensive tiles, the rest.
    unsigned long long mine = 0ull;
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t c = cost[n - 1u - j];
        const uint32_t k = max_slices > 1u ? slices_of(c) : 1u;
        mine += k > 1u ? (unsigned long long)k : (c > limit ? 1ull << 21 : 1ull << 42);
    }
    scan[tid] = mine;
    __syncthreads();
    for (uint32_t off = 1u; off < 1024u; off <<= 1) {
        const unsigned long long v = tid >= off ? scan[tid - off] : 0ull;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    const unsigned long long all = scan[1023], before = scan[tid] - mine;
    const uint32_t n0 = (uint32_t)(all & 0x1fffffu), n1 = (uint32_t)((all >> 21) & 0x1fffffu), n2 = (uint32_t)(all >> 42);
    uint32_t at0 = (uint32_t)(before & 0x1fffffu), at1 = n0 + (uint32_t)((before >> 21) & 0x1fffffu), at2 = n0 + n1 + (uint32_t)(before >> 42);
    for (uint32_t j = j0; j < j1; ++j) {
        const uint32_t tile = n - 1u - j, c = cost[tile];
        const uint32_t k = max_slices > 1u ? slices_of(c) : 1u;
        if (k > 1u) { for (uint32_t sl = 0u; sl < k; ++sl) work[at0++] = tile | (sl << 24) | ((k - 1u) << 28); }
        else if (c > limit) work[at1++] = tile;
        else work[at2++] = tile;
    }
    for (uint32_t i = n0 + n1 + n2 + tid; i < n + extra; i += 1024u) work[i] = WORK_NONE;
    __syncthreads();
    for (uint32_t i = tid; i < n; i += 1024u) cost[i] = 0u;
}

// The candidate lists of the camera rays (rt_device.h, TLIST_*): one thread per 8x8-pixel wave tile of the launch.
// A camera ray starts on the image plane inside its pixel's square (RayTracing.glsl:293-296, rayTracer.h:92-101) and
// points away from the eye (glsl:297, rayTracer.h:100), so every camera ray of a tile lies in the pyramid from the eye
// through the tile's rectangle on the image plane.  A primitive is a candidate when its bound reaches into that
// pyramid: for each of the four side planes, the bound's support point along the plane's inward normal is not
// outside by more than `slack`.  The slack covers what the tracer's own arithmetic can move a ray by (the rounding of
// the origin on the plane, a few units of 2^-24 of the camera's coordinates, seen from the eye at the distance of the
// plane) and, for the spheres, what the rounding of Sphere::hit's discriminant can still accept (about 2^-22 d^2 / r
// beside a sphere of radius r at distance d); the tree's bounds and the boxes' bounds carry the slack of their own
// tests already.  Computed in double: the lists must never miss, and the kernel runs once per camera.
__global__ __launch_bounds__(64)
void tile_lists_kernel(const TraceParams P, const float4* __restrict__ blob, uint32_t* __restrict__ lists)
{
    const uint32_t wt = blockIdx.x * 64u + threadIdx.x;
    const uint32_t wtx_n = 2u * P.tiles_x, n_wt = wtx_n * 2u * P.tiles_y;
    if (wt >= n_wt) return;
    uint32_t* const rec = lists + (size_t)wt * TLIST_WORDS;
    const uint32_t wty = wt / wtx_n, wtx = wt - wty * wtx_n;
    // image rows of the tile's eight owned rows (a row shard's bands need not be multiples of eight)
    uint32_t gy_lo = 0xffffffffu, gy_hi = 0u;
    for (uint32_t r = 0; r < 8u; ++r) {
        const uint32_t row = wty * 8u + r;
        if (row >= P.owned_rows) break;
        const uint32_t band = row / P.band_rows;
        const uint32_t gy = (band * P.shard_count + P.shard_index) * P.band_rows + (row - band * P.band_rows);
        if (gy >= P.height) continue;
        gy_lo = min(gy_lo, gy); gy_hi = max(gy_hi, gy);
    }
    const uint32_t gx_lo = wtx * 8u;
    rec[0] = TLIST_NONE; rec[1] = 0u;
    for (uint32_t w = 0; w < TLIST_SPH_WORDS; ++w) rec[2u + w] = 0u;
    if (gy_lo > gy_hi || gx_lo >= P.width) return;                 // no pixel: no wave reads this record
    const SceneLayout& L = P.scene;
    if (L.n_sph > tlist_max_spheres(L.n_tris != 0u) || L.n_box > TLIST_MAX_BOXES) return;   // beyond the masks: no lists for this scene

    // the tile's rectangle in pixel units, a little wider than its pixels; the grid as written lets the x offset
    // accumulate beyond the pixel (rayTracer.h:98-99): by less than grid_n / 2 pixels
    const double margin = 1.0 / 64.0;
    const double x0 = (double)gx_lo - margin, x1 = (double)min(gx_lo + 8u, P.width) + margin + (P.sample_mode == 2u ? 0.5 * (double)P.grid_n + 1.0 : 0.0);
    const double y0 = (double)gy_lo - margin, y1 = (double)gy_hi + 1.0 + margin;
    const double step = P.cam[19];
    const double E[3] = { P.cam[0], P.cam[1], P.cam[2] }, UL[3] = { P.cam[8], P.cam[9], P.cam[10] };
    const double up[3] = { P.cam[12], P.cam[13], P.cam[14] }, right[3] = { P.cam[16], P.cam[17], P.cam[18] };
    double a[4][3];                                                // eye -> corners, around the rectangle
    const double cx[4] = { x0, x1, x1, x0 }, cy[4] = { y0, y0, y1, y1 };
    double f_min = 1e300, mag = 1.0;
    for (int k = 0; k < 4; ++k) {
        double l2 = 0.0;
        for (int c = 0; c < 3; ++c) {
            const double pnt = UL[c] + step * cx[k] * right[c] - step * cy[k] * up[c];
            a[k][c] = pnt - E[c];
            l2 += a[k][c] * a[k][c];
            mag = fmax(mag, fmax(fabs(pnt), fabs(E[c])));
        }
        f_min = fmin(f_min, sqrt(l2));
    }
    const double mid[3] = { 0.25 * (a[0][0] + a[1][0] + a[2][0] + a[3][0]), 0.25 * (a[0][1] + a[1][1] + a[2][1] + a[3][1]),
                            0.25 * (a[0][2] + a[1][2] + a[2][2] + a[3][2]) };
    double nrm[4][3];
    bool planes_ok = true;
    for (int k = 0; k < 4; ++k) {
        const double* u = a[k]; const double* v = a[(k + 1) & 3];
        double n[3] = { u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0] };
        const double len = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        const double side = n[0] * mid[0] + n[1] * mid[1] + n[2] * mid[2];
        if (!(len > 0.0) || !(fabs(side) > 0.0) || !(len < 1e300)) planes_ok = false;
        const double sc = (side < 0.0 ? -1.0 : 1.0) / len;
        for (int c = 0; c < 3; ++c) nrm[k][c] = n[c] * sc;         // unit, pointing into the pyramid
    }
    if (!planes_ok || !(f_min > 0.0)) return;                      // a degenerate camera: no lists
    // what the tracer's rounding can turn a ray by, as an angle seen from the eye
    const double angle = 1e-5 + 64.0 * 0x1p-24 * mag / f_min;
    // -> does the ball {c, R} (or, with h, the box c +- h) reach into the pyramid?
    auto reaches = [&](const double c[3], double R, const double* h, double extra) -> bool {
        const double rel[3] = { c[0] - E[0], c[1] - E[1], c[2] - E[2] };
        // the farthest point of the bound from the eye: a ray turned by `angle` is off by angle * distance there
        const double dist = sqrt(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) + R + (h ? sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2]) : 0.0);
        const double slack = angle * dist + 1e-6 * mag + extra;
        bool in = true;
        for (int k = 0; k < 4; ++k) {
            double s = nrm[k][0] * rel[0] + nrm[k][1] * rel[1] + nrm[k][2] * rel[2] + R;
            if (h) s += fabs(nrm[k][0]) * h[0] + fabs(nrm[k][1]) * h[1] + fabs(nrm[k][2]) * h[2];
            in = in && (s >= -slack);                              // a NaN (a never-hit padding node) is out
        }
        return in;
    };
    // (a scene without a mesh may have more spheres than TLIST_SPH_WORDS words hold: their bits go on into the words
    // of the leaves it does not have)
    const uint32_t sph_words = max(TLIST_SPH_WORDS, (L.n_sph + 31u) >> 5);
    uint32_t box_mask = 0u;
    for (uint32_t w = 0; w < sph_words; ++w) rec[2u + w] = 0u;
    for (uint32_t i = 0; i < L.n_sph; ++i) {
        const float4 sp = blob[L.sph_off + i];
        const double c[3] = { sp.x, sp.y, sp.z }, R = sqrt((double)sp.w);
        const double d2 = (c[0] - E[0]) * (c[0] - E[0]) + (c[1] - E[1]) * (c[1] - E[1]) + (c[2] - E[2]) * (c[2] - E[2]);
        const double far2 = d2 + f_min * f_min + R * R;
        // a sphere of radius zero or with a NaN in it: keep it a candidate (its test decides)
        const bool sane = R > 0.0 && far2 < 1e300;
        if (!sane || reaches(c, R, nullptr, 0x1p-20 * far2 / R)) rec[2u + (i >> 5)] |= 1u << (i & 31u);
    }
    for (uint32_t i = 0; i < L.n_box; ++i) {
        const float4 bc = blob[L.box_off + 3u * i], bh = blob[L.box_off + 3u * i + 2u];
        const double c[3] = { bc.x, bc.y, bc.z }, h[3] = { bh.x, bh.y, bh.z };
        const bool sane = h[0] == h[0] && h[1] == h[1] && h[2] == h[2] && c[0] == c[0] && c[1] == c[1] && c[2] == c[2];
        if (!sane || reaches(c, 0.0, h, 0.0)) box_mask |= 1u << i;
    }
    uint32_t n_leaf = 0u;
    bool overflow = false;
    auto node_reaches = [&](uint32_t level, uint32_t i) -> bool {
        const float4 nb = blob[L.node_off[level] + i];
        const double c[3] = { nb.x, nb.y, nb.z };
        if (!(nb.w < 3e38f)) return nb.w == nb.w;                  // an unbounded node (a degenerate triangle) is always entered
        return reaches(c, (double)nb.w, nullptr, 0.0);
    };
    if (L.n_tris != 0u) {
        // depth first from every node of the top level; at most TREE_FAN - 1 siblings wait per level
        uint32_t stack[MAX_TREE_LEVELS * TREE_FAN];
        const uint32_t top = L.tree_levels - 1u;
        for (uint32_t it = 0; it < L.node_n[top] && !overflow; ++it) {
            uint32_t n = 0u;
            stack[n++] = (top << 24) | it;
            while (n != 0u && !overflow) {
                const uint32_t ent = stack[--n], lvl = ent >> 24, node = ent & 0xffffffu;
                if (!node_reaches(lvl, node)) continue;
                if (lvl == 0u) {
                    if (n_leaf < TLIST_MAX_LEAVES) rec[TLIST_LEAF0 + n_leaf++] = node; else overflow = true;
                    continue;
                }
                const uint32_t c0 = TREE_FAN * node, c1 = min(c0 + TREE_FAN, L.node_n[lvl - 1u]);
                for (uint32_t c = c1; c-- > c0;) stack[n++] = ((lvl - 1u) << 24) | c;      // the lowest child on top
            }
        }
    }
    rec[1] = box_mask;
    rec[0] = overflow ? TLIST_NONE : n_leaf;
}

// What a 16x16 tile is likely to cost before anything has been traced, from the candidate lists of its four wave tiles:
// 1 + leaf_weight x the largest number of candidate leaves among them (a list that overflowed counts as a full one).
// A probe launch is planned from it (rt_api.cpp, trace_frame): without any plan the probe of a scene with a large mesh
// lasts as long as the lone waves on the mesh's tiles at its end, whose passes each wait for memory with nothing to
// hide it behind.  (Spreading the assumed cost to the tiles around, halved per tile of distance up to three — the floor
// beside a mesh is expensive too, its rays bounce into it — was measured and made the cold frames slower.)
__global__ void tile_prior_kernel(const TraceParams P, const uint32_t* __restrict__ lists, uint32_t* __restrict__ cost, uint32_t leaf_weight)
{
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= P.tiles_x * P.tiles_y) return;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    uint32_t most = 0u;
    for (uint32_t sub = 0; sub < 4u; ++sub) {
        const uint32_t wtx = 2u * tx + (sub & 1u), wty = 2u * ty + (sub >> 1);
        if (wtx * 8u >= P.width || wty * 8u >= P.owned_rows) continue;       // no pixel there
        const uint32_t n = lists[(size_t)(wty * 2u * P.tiles_x + wtx) * TLIST_WORDS];
        most = max(most, n == TLIST_NONE ? TLIST_MAX_LEAVES + 6u : n);
    }
    cost[tile] = 1u + leaf_weight * most;
}

// TRACE launches first mark every entry "path already over".
__global__ void trace_fill_kernel(int32_t* hit, uint32_t* mat, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { hit[i] = -2; mat[i] = 0xffffffffu; }
}

// One frame's mean: rayTracer.h:105 (sum / totalRays, optional cwiseSqrt) and
// PixelConversion.glsl:17-20 (uint sum / (255 * samples)); alpha = 1.
template <int SEM>
__global__ void resolve_kernel(const uint32_t* __restrict__ accum, float* __restrict__ out,
                               uint32_t width, uint32_t height, uint32_t samples,
                               int gamma_sqrt, int flip_rows)
{
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= width || y >= height) return;
    uint4 a = *reinterpret_cast<const uint4*>(accum + 4u * ((size_t)y * width + x));
    const float4 v = resolve_pixel<SEM>(a.x, a.y, a.z, samples, gamma_sqrt != 0);
    uint32_t oy = flip_rows ? height - 1u - y : y;
    *reinterpret_cast<float4*>(out + 4u * ((size_t)oy * width + x)) = v;
}

// PixelConversion.glsl:12-28: fold the frame into the running mean, clear.
template <int SEM>
__global__ void convert_kernel(uint32_t* __restrict__ accum, float* __restrict__ image,
                               uint32_t width, uint32_t height, uint32_t samples, uint32_t n_samples)
{
    uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t y = blockIdx.y;
    if (x >= width || y >= height) return;
    uint4* ap = reinterpret_cast<uint4*>(accum + 4u * ((size_t)y * width + x));
    uint4 a = *ap;
    float4 cur;
    if (SEM == 0) {
        float ns = (float)samples;
        cur.x = __uint_as_float(a.x) / ns; cur.y = __uint_as_float(a.y) / ns; cur.z = __uint_as_float(a.z) / ns;
    } else {
        float den = 255.0f * (float)samples;
        cur.x = (float)a.x / den; cur.y = (float)a.y / den; cur.z = (float)a.z / den;
    }
    cur.w = 1.0f;
    float4* ip = reinterpret_cast<float4*>(image + 4u * ((size_t)(height - 1u - y) * width + x));
    float nf = (float)n_samples;
    if (nf != 1.0f) {
        float wa = (nf - 1.0f) / nf, wb = 1.0f / nf;
        float4 prev = *ip;
        cur.x = fma_(wa, prev.x, cur.x * wb); cur.y = fma_(wa, prev.y, cur.y * wb);
        cur.z = fma_(wa, prev.z, cur.z * wb); cur.w = fma_(wa, prev.w, cur.w * wb);
    }
    *ip = cur;
    *ap = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------ static scratch
// Per-tile costs (RT_NOALLOC_MAX_TILES words) and the launch plan (twice as many) of rt_render_noalloc, the render
// call that allocates nothing.  In host code the array's name is only a handle for hipGetSymbolAddress: each GPU
// holds its own copy at its own address, looked up once per device (with that device current) and kept; kernels,
// memsets and copies get that pointer, never the name.
constexpr uint32_t SCRATCH_WORDS = 3u * RT_NOALLOC_MAX_TILES;
static __device__ __align__(256) uint32_t scratch[SCRATCH_WORDS];

constexpr int SCRATCH_MAX_DEVICES = 64;
static uint32_t* g_scratch_ptr[SCRATCH_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// -> this device's address of `scratch`.  `device` must be the current device.
hipError_t scratch_for_device(int device, uint32_t** out)
{
    *out = nullptr;
    if (device < 0 || device >= SCRATCH_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<uint32_t*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// ------------------------------------------------------------ host launchers

template <int SEM, int GEO, int MODE>
static hipError_t launch_one(dim3 grid, size_t lds_bytes, hipStream_t stream, const TraceParams& P, const float4* blob,
                             uint32_t* accum, unsigned long long* st, int32_t* th, uint32_t* tm,
                             const uint32_t* tile_order, uint32_t* tile_cost, const uint32_t* tile_lists)
{
    hipLaunchKernelGGL((trace_kernel<SEM, GEO, MODE>), grid, dim3(64u * P.group_waves), lds_bytes, stream, P, blob, accum, st, th, tm,
                       tile_order, tile_cost, tile_lists);
    return hipGetLastError();
}

hipError_t launch_tile_lists(hipStream_t stream, const TraceParams& P, const void* blob, uint32_t* lists)
{
    const uint32_t n_wt = 4u * P.tiles_x * P.tiles_y;
    if (n_wt == 0u) return hipSuccess;
    hipLaunchKernelGGL(tile_lists_kernel, dim3((n_wt + 63u) / 64u), dim3(64), 0, stream, P, static_cast<const float4*>(blob), lists);
    return hipGetLastError();
}

hipError_t launch_tile_prior(hipStream_t stream, const TraceParams& P, const uint32_t* lists, uint32_t* cost, uint32_t leaf_weight)
{
    const uint32_t n = P.tiles_x * P.tiles_y;
    if (n == 0u) return hipSuccess;
    hipLaunchKernelGGL(tile_prior_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, P, lists, cost, leaf_weight);
    return hipGetLastError();
}
hipError_t launch_tile_plan(hipStream_t stream, uint32_t* cost, uint32_t* work, uint32_t n, uint32_t extra, uint32_t max_slices, float share, float early_share)
{
    hipLaunchKernelGGL(tile_plan_kernel, dim3(1), dim3(1024), 0, stream, cost, work, n, extra, max_slices, share, early_share);
    return hipGetLastError();
}

hipError_t launch_trace(int sem, int mode, hipStream_t stream, const TraceParams& Pin, const void* blob,
                        uint32_t* accum, unsigned long long* st, int32_t* th, uint32_t* tm,
                        const uint32_t* tile_order, uint32_t* tile_cost, uint32_t n_workgroups, const uint32_t* tile_lists)
{
    const float4* b = static_cast<const float4*>(blob);
    const int tri_mode = Pin.scene.n_tris == 0 ? (Pin.scene.n_box != 0 ? 1 : 0) : (Pin.scene.tri_in_lds ? 1 : 2);
    TraceParams P = Pin;
    if (tri_mode == 0 && P.group_waves > 4u) P.group_waves = 4u;          // instances without queues are built for 256 threads
    P.plan_entries = n_workgroups;
    const dim3 grid((4u * n_workgroups + P.group_waves - 1u) / P.group_waves);
    const size_t lds_bytes = ((size_t)P.scene.lds_words + (tri_mode ? lds_wave_words(P.group_waves, tri_mode == 2) : 0u)) * 16u;
    const int geo = tri_mode | (((sem == 0 && P.scene.n_cube != 0) || P.scene.has_glass || P.scene.tcol_off != 0 || (P.flags & 4u) != 0u) ? 4 : 0) |
                    (P.rng_mode ? 8 : 0) | ((P.flags & RT_FLAG_LITERAL_GEOMETRY) != 0u ? 16 : 0) |
                    ((RT_PSTEP != 0 && tile_lists != nullptr && P.rng_mode == 0 && (P.flags & RT_FLAG_LITERAL_GEOMETRY) == 0u && mode != 2) ? 32 : 0);
    if ((geo & 32) == 0) tile_lists = nullptr;
    const int deep = ((tri_mode == 2 && P.scene.tree_levels > TREE_LEVELS) ? 64 : 0) | (((geo & 32) != 0 && tri_mode == 0 && P.scene.n_sph > 32u) ? 128 : 0);          // (never with the shader's own stream: rt_api.cpp keeps such a tree at four levels)
#define RT_CASE(S, G, M) if (sem == S && (geo | deep) == G && mode == M) return launch_one<S, G, M>(grid, lds_bytes, stream, P, b, accum, st, th, tm, tile_order, tile_cost, tile_lists)
#define RT_CASES(S, G) RT_CASE(S, G, 0); RT_CASE(S, G, 1); RT_CASE(S, G, 2)
#define RT_GEOS(S) RT_CASES(S, 0); RT_CASES(S, 1); RT_CASES(S, 2); RT_CASES(S, 4); RT_CASES(S, 5); RT_CASES(S, 6); \
                   RT_CASES(S, 8); RT_CASES(S, 9); RT_CASES(S, 10); RT_CASES(S, 12); RT_CASES(S, 13); RT_CASES(S, 14)
    RT_GEOS(0); RT_GEOS(1);
    RT_CASES(0, 16); RT_CASES(0, 20);                      // literal geometry: CPU rules, spheres and cubes, counter hash
#define RT_PCASES(S, G) RT_CASE(S, G, 0); RT_CASE(S, G, 1)
    RT_PCASES(0, 32); RT_PCASES(0, 33); RT_PCASES(0, 34); RT_PCASES(0, 36); RT_PCASES(0, 37); RT_PCASES(0, 38);   // primary steps: counter hash
    RT_PCASES(1, 32); RT_PCASES(1, 33); RT_PCASES(1, 34); RT_PCASES(1, 36); RT_PCASES(1, 37); RT_PCASES(1, 38);
    RT_CASES(0, 66); RT_CASES(0, 70); RT_PCASES(0, 98); RT_PCASES(0, 102);                 // ... and a tree of more than four levels
    RT_CASES(1, 66); RT_CASES(1, 70); RT_PCASES(1, 98); RT_PCASES(1, 102);
    RT_PCASES(0, 160); RT_PCASES(0, 164); RT_PCASES(1, 160); RT_PCASES(1, 164);          // ... and sphere-only scenes of more than 32 spheres
#undef RT_PCASES
#undef RT_GEOS
#undef RT_CASES
#undef RT_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_trace_fill(hipStream_t stream, int32_t* hit, uint32_t* mat, size_t n)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(trace_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, hit, mat, n);
    return hipGetLastError();
}

hipError_t launch_resolve(int sem, hipStream_t stream, const uint32_t* accum, float* out, uint32_t w, uint32_t h,
                          uint32_t samples, int gamma_sqrt, int flip_rows)
{
    dim3 grid((w + 255) / 256, h), block(256);
    if (sem == 0) hipLaunchKernelGGL(resolve_kernel<0>, grid, block, 0, stream, accum, out, w, h, samples, gamma_sqrt, flip_rows);
    else hipLaunchKernelGGL(resolve_kernel<1>, grid, block, 0, stream, accum, out, w, h, samples, gamma_sqrt, flip_rows);
    return hipGetLastError();
}

hipError_t launch_convert(int sem, hipStream_t stream, uint32_t* accum, float* image, uint32_t w, uint32_t h,
                          uint32_t samples, uint32_t n_samples)
{
    dim3 grid((w + 255) / 256, h), block(256);
    if (sem == 0) hipLaunchKernelGGL(convert_kernel<0>, grid, block, 0, stream, accum, image, w, h, samples, n_samples);
    else hipLaunchKernelGGL(convert_kernel<1>, grid, block, 0, stream, accum, image, w, h, samples, n_samples);
    return hipGetLastError();
}

}  // namespace rtd
