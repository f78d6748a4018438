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
// The scene is read-only and tiny next to the caches; every lane tests the
// same primitive at the same time, so the closest-hit loops address it with a
// wave-uniform index and the compiler fetches it through the scalar data path
// into SGPRs (s_load_dwordx4/x8/x16).  Nothing is staged in VGPRs or LDS for
// the loops; only the record of the primitive a lane finally hit is fetched
// per lane.
//
// Arithmetic: binary32 throughout, compiled with -ffp-contract=off.  Every
// fused multiply-add is an explicit __builtin_fmaf; division and sqrtf are the
// correctly rounded forms (hipcc default).  No OCML sin/cos/log: the samplers
// use the polynomial below.  With that, every value computed here can be
// reproduced bit for bit by scalar IEEE-754 code on a CPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_device.h"

namespace rtd {

struct V3 { float x, y, z; };

#define RT_DEV static __device__ __forceinline__

RT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_DEV float dot(V3 a, V3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// o + t*d
RT_DEV V3 madd(float t, V3 d, V3 o) { return v3(fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)); }
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
// Correctly rounded sqrt: v_sqrt_f32 (within 1 ulp) plus a residual test against each
// neighbour — the sequence hipcc emits for sqrtf, without its range scaling.  Arguments in
// (0, 2^-96), where the residuals would go subnormal, take the compiler's general sequence
// when GUARD is set; callers whose argument cannot be that small pass GUARD = false.
template <bool GUARD>
RT_DEV float sqrt_rn(float x)
{
    if (GUARD && __builtin_expect(__any((__float_as_uint(x) - 1u) < 0x0f7fffffu), 0)) return __builtin_sqrtf(x);
    float s = __builtin_amdgcn_sqrtf(x);
    float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    float rm = fma_(-sm, s, x), rp = fma_(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
}
RT_DEV V3 cross(V3 a, V3 b)
{
    return v3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
// GLSL sign(): -1, 0, 1; NaN gives 0.
RT_DEV float sign_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
// GLSL max(): (x < y) ? y : x — not v_max_f32, whose NaN rule differs.
RT_DEV float gmax(float x, float y) { return (x < y) ? y : x; }

// ------------------------------------------------------------------- RNG
// Counter-based: draw i of the path (seed, pixel, sample) is
// hash32(key + i * 0x9E3779B9), key = hash32(hash32(seed_key ^ pixel) ^ sample).
// It stands in for random() of RayTracing.glsl:44-51 (a hash of time,
// invocation id and draw count) and for the mt19937 of src/utils.h:8-12.
RT_DEV uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
RT_DEV float draw(uint32_t key, uint32_t index)
{
    return (float)(hash32(key + index * 0x9e3779b9u) >> 8) * 0x1p-24f;
}
// Draws 0,1: sub-pixel jitter.  Bounce b: draws 2+4b .. 5+4b.
RT_DEV uint32_t bounce_draw(uint32_t bounce) { return 2u + 4u * bounce; }

// cos(2 pi u), sin(2 pi u), u in [0,1): quadrant split (exact) and Taylor
// polynomials on [0, pi/2) in Horner form.
RT_DEV void sincos2pi(float u, float& c_out, float& s_out)
{
    float t = u * 4.0f;
    int q = (int)t;
    float f = t - (float)q;
    float x = f * 0x1.921fb6p+0f;
    float x2 = x * x;
    float ps = -0x1.ae6456p-26f;
    ps = fma_(ps, x2, 0x1.71de3ap-19f);
    ps = fma_(ps, x2, -0x1.a01a02p-13f);
    ps = fma_(ps, x2, 0x1.111112p-7f);
    ps = fma_(ps, x2, -0x1.555556p-3f);
    ps = fma_(ps, x2, 1.0f);
    float s = x * ps;
    float pc = 0x1.1eed8ep-29f;
    pc = fma_(pc, x2, -0x1.27e4fcp-22f);
    pc = fma_(pc, x2, 0x1.a01a02p-16f);
    pc = fma_(pc, x2, -0x1.6c16c2p-10f);
    pc = fma_(pc, x2, 0x1.555556p-5f);
    pc = fma_(pc, x2, -0.5f);
    float c = fma_(pc, x2, 1.0f);
    // q: 0 (c,s)  1 (-s,c)  2 (-c,-s)  3 (s,-c)
    float a = (q & 1) ? s : c;
    float b = (q & 1) ? c : s;
    c_out = ((q + 1) & 2) ? -a : a;
    s_out = (q & 2) ? -b : b;
}

// A point uniform on the unit sphere: what randomUnitVector() of
// src/utils.h:37-46 (three normal deviates, normalised) is distributed as.
RT_DEV V3 unit_vector(float u1, float u2)
{
    float z = fma_(-2.0f, u1, 1.0f);
    float r = sqrt_rn<false>(fma_(-z, z, 1.0f));     // 1 - z*z is 0 or >= 2^-23
    float c, s;
    sincos2pi(u2, c, s);
    return v3(r * c, r * s, z);
}

// randomOnHalfUnitShere, RayTracing.glsl:64-80.
RT_DEV V3 hemisphere_glsl(V3 n, float u1, float u2)
{
    float c, s;
    sincos2pi(u1, c, s);
    float r = u2;
    float dist = sqrt_rn<false>(fma_(-r, r, 1.0f));  // 1 - r*r >= 2^-24
    float x = c * dist, y = s * dist;
    float q = fma_(n.x, n.x, n.y * n.y);          // |(-n.y, n.x, 0)|^2
    float len_xy = sqrt_rn<true>(q);
    V3 b1;
    if (len_xy < 0.01f) b1 = normalize(v3(0.0f, -n.z, n.y));
    else { float inv = rsqrt_nr(q); b1 = v3(-n.y * inv, n.x * inv, 0.0f * inv); }
    V3 b2 = cross(n, b1);
    return v3(fma_(n.x, r, fma_(b2.x, y, b1.x * x)),
              fma_(n.y, r, fma_(b2.y, y, b1.y * x)),
              fma_(n.z, r, fma_(b2.z, y, b1.z * x)));
}

// ---------------------------------------------------------- intersection
// Sphere::hit (src/cpu_version/scene.h:115-128) = hitSphere
// (RayTracing.glsl:118-132).  r2 = radius*radius, formed at upload.
RT_DEV float hit_sphere(float cx, float cy, float cz, float r2, V3 o, V3 d)
{
    V3 oc = v3(o.x - cx, o.y - cy, o.z - cz);
    float b = dot(oc, d);
    float c = fma_(oc.z, oc.z, fma_(oc.y, oc.y, fma_(oc.x, oc.x, -r2)));   // oc.oc - r*r
    float disc = fma_(b, b, -c);
    // The root -b - sqrt(disc) can only lie in the window (t > 0) when b < 0 and disc >= 0.
    // When that holds for no lane of the wave the square root is skipped; the NaN returned
    // instead fails every comparison, exactly as the skipped value would have.
    if (!__any(b < 0.0f && disc >= 0.0f)) return __builtin_nanf("");
    return -b - sqrt_rn<true>(disc);       // disc < 0 gives NaN, which every later comparison rejects
}

// hitBox, RayTracing.glsl:138-156.
RT_DEV float hit_box(float cx, float cy, float cz, float inv, V3 o, V3 d)
{
    V3 ob = v3(inv * (o.x - cx), inv * (o.y - cy), inv * (o.z - cz));
    V3 db = v3(inv * d.x, inv * d.y, inv * d.z);
    V3 sg = v3(sign_(ob.x), sign_(ob.y), sign_(ob.z));
    float t[3];
    t[0] = (0.5f - ob.x * sg.x) / (sg.x * db.x);
    t[1] = (0.5f - ob.y * sg.y) / (sg.y * db.y);
    t[2] = (0.5f - ob.z * sg.z) / (sg.z * db.z);
    float best = __builtin_inff();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float px = __builtin_fabsf(fma_(db.x, t[k], ob.x));
        float py = __builtin_fabsf(fma_(db.y, t[k], ob.y));
        float pz = __builtin_fabsf(fma_(db.z, t[k], ob.z));
        float inf_norm = gmax(px, gmax(py, pz));
        if (inf_norm < 0.5001f && inf_norm > 0.4999f && best > t[k]) best = t[k];
    }
    return best;
}

// boxNormalAt, RayTracing.glsl:158-163.
RT_DEV V3 box_normal(float cx, float cy, float cz, float scale, V3 p)
{
    float qx = (p.x - cx) / scale, qy = (p.y - cy) / scale, qz = (p.z - cz) / scale;
    V3 n = v3(((__builtin_fabsf(qx) < 0.499f) ? 0.0f : 1.0f) * sign_(qx),
              ((__builtin_fabsf(qy) < 0.499f) ? 0.0f : 1.0f) * sign_(qy),
              ((__builtin_fabsf(qz) < 0.499f) ? 0.0f : 1.0f) * sign_(qz));
    return normalize(n);
}

// hitTriangle, RayTracing.glsl:179-194.
RT_DEV float hit_triangle(const DevTriangle& tr, V3 o, V3 d)
{
    V3 a = v3(tr.ax, tr.ay, tr.az);
    float k = (tr.beta - dot(a, o)) / dot(a, d);
    V3 p = madd(k, d, o);
    V3 q = v3(p.x - tr.px, p.y - tr.py, p.z - tr.pz);
    float x = dot(q, v3(tr.m1x, tr.m1y, tr.m1z));
    float y = dot(q, v3(tr.m2x, tr.m2y, tr.m2z));
    bool hit = !(k < 0.0f) && x > 0.0f && y > 0.0f && (x + y) < 1.0f;
    return hit ? k : __builtin_inff();
}

// Acceptance window of a hit distance.  CPU tracer: scene.h:237 with the
// arguments of rayTracer.h:117 (t_min is the double 0.001 compared with a
// widened float, true exactly when t >= 0.001f).  Shader: RayTracing.glsl:212.
template <int SEM>
RT_DEV bool in_window_near(float t)
{
    return (SEM == 0) ? (t >= 0.001f) : (t > 0.0001f);
}

// uvec4(255 * color), RayTracing.glsl:301; negative and NaN give 0, large
// values saturate (v_cvt_u32_f32 does exactly that).
RT_DEV uint32_t quantise(float c)
{
    float v = 255.0f * c;
    return (v > 0.0f) ? ((v >= 4294967296.0f) ? 0xffffffffu : (uint32_t)v) : 0u;
}

// -------------------------------------------------------------- the tracer
// SEM 0: rules of src/cpu_version (rayTracer.h:110-132, scene.h:39-109).
// SEM 1: rules of RayTracing.glsl:233-288.
// STATS: count work.  TRACE: follow one sample, record hit and material ids,
// leave the accumulator alone.
//
// One loop iteration = one ray segment of every live lane: closest hit, then
// the surface event.  Lanes whose path ended start their next sample in the
// same iteration, and the two expensive steps every continuing lane needs —
// two random draws and the normalisation of the new direction — are issued
// once for the whole wave, whether a lane is bouncing or starting a sample.
template <int SEM, bool STATS, bool TRACE>
__global__ __launch_bounds__(BLOCK)
void trace_kernel(const TraceParams P,
                  const DevShape* __restrict__ shapes,
                  const DevTriangle* __restrict__ tris,
                  uint32_t* __restrict__ accum,
                  unsigned long long* __restrict__ stats,
                  int32_t* __restrict__ trace_hit,
                  uint32_t* __restrict__ trace_mat)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    const uint32_t gx = blockIdx.x * TILE_W + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t row = (gridDim.y - 1u - blockIdx.y) * TILE_H + (wave >> 1) * 8u + (lane >> 3);   // ordinal among owned rows
    const uint32_t band = row / P.band_rows;
    const uint32_t gy = (band * P.shard_count + P.shard_index) * P.band_rows + (row - band * P.band_rows);
    const bool valid = gx < P.width && row < P.owned_rows && gy < P.height;

    const uint32_t pixel = gy * P.width + gx;
    const float step = P.cam[19];
    const V3 cam_pos = v3(P.cam[0], P.cam[1], P.cam[2]);
    const V3 cam_up = v3(P.cam[12], P.cam[13], P.cam[14]);
    const V3 cam_right = v3(P.cam[16], P.cam[17], P.cam[18]);
    // rayTracer.h:77 / glsl:293: upperLeft + step*gx*right - step*gy*up
    const V3 pc = madd(-(step * (float)gy), cam_up, madd(step * (float)gx, cam_right, v3(P.cam[8], P.cam[9], P.cam[10])));
    const uint32_t h1 = hash32(P.seed_key ^ pixel);

    // Running sums of this pixel: float for SEM 0, uint for SEM 1.
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
    uint32_t* const acc_ptr = accum + 4u * (size_t)pixel;
    if (!TRACE && valid) {
        uint4 v = *reinterpret_cast<const uint4*>(acc_ptr);
        acc0 = v.x; acc1 = v.y; acc2 = v.z;
    }

    uint32_t n_samples = 0, n_segments = 0, n_diffuse = 0, n_specular = 0, n_terminal = 0;

    uint32_t sample = TRACE ? P.trace_sample : P.first_sample;
    const uint32_t sample_end = TRACE ? P.trace_sample + 1u : P.first_sample + P.spp;
    bool alive = valid && sample < sample_end;

    // Origin and un-normalised direction of the camera ray of `sample`:
    // RayTracing.glsl:296-297 (jitter, from two draws) or renderPixel<N>,
    // src/cpu_version/rayTracer.h:92-101 (grid).
    auto camera_origin = [&](uint32_t smp, float u1, float u2) -> V3 {
        if (P.sample_mode == 0u) {
            float s1 = step * u1, s2 = step * u2;
            return madd(-s2, cam_up, madd(s1, cam_right, pc));
        }
        const uint32_t N = P.grid_n;
        const uint32_t i = (smp / N) % N, j = smp % N;
        const float step_size = step / (float)(N + 1u);
        const V3 ups = cam_up * step_size, rs = cam_right * step_size;
        V3 org = pc - ups * (float)i;
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

    uint32_t key = 0, bounce = 0;
    V3 o = v3(0, 0, 0), d = v3(0, 0, 1), thr = v3(1, 1, 1);
    if (alive) {
        key = hash32(h1 ^ sample);
        o = camera_origin(sample, draw(key, 0u), draw(key, 1u));
        d = normalize(o - cam_pos);
        if (STATS) ++n_samples;
    }

    const uint32_t n_shapes = P.n_shapes, n_tris = P.n_tris;
    const float t_far = (SEM == 0) ? 3.402823466e+38f : 1000.0f;   // scene.h:237 / glsl:212

    while (alive) {
        // ---- closest hit: Scene::hit (scene.h:232-255) /
        //      getClosestShapeIndex + getClosestTriangleIndex (glsl:206-231).
        //      `best` starts at the far end of the window, so `t < best`
        //      also is the window's upper test; a NaN t fails both tests.
        float best = t_far;
        int index = -1;
        for (uint32_t i = 0; i < n_shapes; ++i) {
            const DevShape& s = shapes[i];
            float t;
            if (s.type == 0u) t = hit_sphere(s.cx, s.cy, s.cz, s.p0, o, d);
            else if (s.type == 1u) t = hit_box(s.cx, s.cy, s.cz, s.p1, o, d);
            else t = -1.0f;
            if (in_window_near<SEM>(t) && t < best) { best = t; index = (int)i; }
        }
        float best_tri = t_far;
        int index_tri = -1;
        for (uint32_t i = 0; i < n_tris; ++i) {
            float t = hit_triangle(tris[i], o, d);
            if (in_window_near<SEM>(t) && t < best_tri) { best_tri = t; index_tri = (int)i; }
        }
        if (STATS) ++n_segments;

        // ---- the surface
        uint32_t type = 0xffffffffu;           // nothing hit
        int hit_id = -1;
        V3 n = v3(0, 0, 0), att = v3(0, 0, 0);
        float fuzz = 0.0f;
        const bool is_tri = index_tri >= 0 && best_tri < best;        // glsl:271
        const V3 p = madd(is_tri ? best_tri : best, d, o);            // Ray::at / glsl:239
        if (is_tri) {
            const DevTriangle& tr = tris[index_tri];
            V3 a = v3(tr.ax, tr.ay, tr.az);
            n = a * sign_(-dot(d, a));                                // glsl:165-167,272
            att = v3(P.tri_albedo[0], P.tri_albedo[1], P.tri_albedo[2]);
            type = 1u;
            hit_id = (int)n_shapes + index_tri;
        } else if (index >= 0) {
            const DevShape s = shapes[index];
            if (s.type == 0u) n = normalize(v3(p.x - s.cx, p.y - s.cy, p.z - s.cz));     // scene.h:130-132
            else if (s.type == 1u) n = box_normal(s.cx, s.cy, s.cz, s.p0, p);
            else n = v3(1.0f, 1.0f, 1.0f);                            // glsl:203
            att = v3(s.r, s.g, s.b);
            type = s.mat_type;
            fuzz = s.fuzz;
            hit_id = index;
        }
        if (TRACE) {
            size_t at = (size_t)pixel * P.max_depth + bounce;
            trace_hit[at] = hit_id;
            trace_mat[at] = type;
        }

        const bool last = bounce + 1u >= P.max_depth;                 // rayTracer.h:113-115 / glsl:263,287
        bool ended = true;
        bool raw_dir = false;                                         // new direction is used as it is
        V3 color = v3(0, 0, 0);                                       // what the path returns if it ends here
        V3 v = v3(0, 0, 1);                                           // new direction before normalisation
        const bool diffuse = type == 1u, metal = type == 2u;
        if (STATS) { n_diffuse += diffuse ? 1u : 0u; n_specular += metal ? 1u : 0u; }
        if (type == 0u) color = thr * att;                            // LightSource::emit / glsl:278-280
        else if (type == 100u) color = (SEM == 0) ? thr * n : n;      // glsl:248-252
        else if (!diffuse && !metal && type != 0xffffffffu) color = (SEM == 0) ? v3(0, 0, 0) : thr;   // glsl:253,282-284
        else if ((diffuse || metal) && !last) ended = false;

        // Two draws per lane: the bounce's pair, or — for a lane that starts a
        // sample — the jitter pair of the new sample.
        if (ended) {
            if (STATS) ++n_terminal;
            if (SEM == 0) {
                acc0 = __float_as_uint(__uint_as_float(acc0) + color.x);   // rayTracer.h:101
                acc1 = __float_as_uint(__uint_as_float(acc1) + color.y);
                acc2 = __float_as_uint(__uint_as_float(acc2) + color.z);
            } else {
                acc0 += quantise(color.x);                                 // glsl:301-307
                acc1 += quantise(color.y);
                acc2 += quantise(color.z);
            }
            ++sample;
            alive = sample < sample_end;
            key = hash32(h1 ^ sample);
            if (STATS && alive) ++n_samples;
        }
        const uint32_t di = ended ? 0u : bounce_draw(bounce);
        const float u1 = draw(key, di), u2 = draw(key, di + 1u);

        if (ended) {
            o = camera_origin(sample, u1, u2);
            v = o - cam_pos;                                          // rayTracer.h:100 / glsl:297
            thr = v3(1, 1, 1);
            bounce = 0;
        } else {
            ++bounce;
            o = p;
            thr = thr * att;                                          // scene.h:56 (0.6*albedo, formed at upload), :79; glsl:241,246,275
            if (diffuse) {
                if (SEM == 0) {
                    v = madd(1.0001f, n, unit_vector(u1, u2));        // Lambertian::scatter, scene.h:51-58; utils.h:48-50
                } else {
                    v = hemisphere_glsl(n, u1, u2);                   // glsl:240,274
                    raw_dir = true;
                }
            } else if (SEM == 0) {
                // Metal::scatter, scene.h:73-81; reflect, utils.h:62-64
                v = madd(-(2.0f * dot(n, d)), n, d);
                if (fuzz != 0.0f) v = madd(fuzz, unit_vector(u1, u2), v);
                raw_dir = (P.flags & 1u) != 0u;
                if (!(dot(v, n) > 0.0f)) {
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
                    key = hash32(h1 ^ sample);
                    if (STATS && alive) ++n_samples;
                    o = camera_origin(sample, draw(key, 0u), draw(key, 1u));
                    v = o - cam_pos;
                    thr = v3(1, 1, 1);
                    bounce = 0;
                    raw_dir = false;
                }
            } else {
                v = madd(-(2.0f * dot(d, n)), n, d);                  // glsl:243-247; reflect :82-84
            }
        }
        const V3 nd = normalize(v);
        d = raw_dir ? v : nd;
    }

    if (!TRACE && valid) {
        uint4 v; v.x = acc0; v.y = acc1; v.z = acc2; v.w = 0u;
        *reinterpret_cast<uint4*>(acc_ptr) = v;
    }

    if (STATS) {
        // Sum over the wave with cross-lane shuffles, one atomic per wave and counter.
        uint32_t c[ST_COUNT] = { n_samples, n_segments, n_diffuse, n_specular, n_terminal };
#pragma unroll
        for (int k = 0; k < ST_COUNT; ++k) {
            unsigned long long v = c[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0 && v) atomicAdd(&stats[k], v);
        }
    }
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
    float4 v;
    if (SEM == 0) {
        float ns = (float)samples;
        v.x = __uint_as_float(a.x) / ns; v.y = __uint_as_float(a.y) / ns; v.z = __uint_as_float(a.z) / ns;
    } else {
        float den = 255.0f * (float)samples;
        v.x = (float)a.x / den; v.y = (float)a.y / den; v.z = (float)a.z / den;
    }
    if (gamma_sqrt) { v.x = __builtin_sqrtf(v.x); v.y = __builtin_sqrtf(v.y); v.z = __builtin_sqrtf(v.z); }
    v.w = 1.0f;
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

// ------------------------------------------------------------ host launchers

hipError_t launch_trace(int sem, bool stats, bool trace, dim3 grid, hipStream_t stream,
                        const TraceParams& P, const DevShape* shapes, const DevTriangle* tris,
                        uint32_t* accum, unsigned long long* st, int32_t* th, uint32_t* tm)
{
    dim3 block(BLOCK);
#define RT_LAUNCH(S, ST, TR) \
    hipLaunchKernelGGL((trace_kernel<S, ST, TR>), grid, block, 0, stream, P, shapes, tris, accum, st, th, tm)
    if (trace) { if (sem == 0) RT_LAUNCH(0, false, true); else RT_LAUNCH(1, false, true); }
    else if (stats) { if (sem == 0) RT_LAUNCH(0, true, false); else RT_LAUNCH(1, true, false); }
    else { if (sem == 0) RT_LAUNCH(0, false, false); else RT_LAUNCH(1, false, false); }
#undef RT_LAUNCH
    return hipGetLastError();
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
