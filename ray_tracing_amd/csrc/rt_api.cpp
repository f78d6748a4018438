// rt_api.cpp — the C ABI of include/rt_mi355x.h on top of the HIP kernels.
//
// Owns the device copy of the scene (the reference's 144-byte Shapes and 64-byte
// SimpleTriangles repacked into one blob of 16-byte words, rt_device.h), the
// accumulator, the running-mean image and the statistics block.  Every HIP return code is checked and turned into an
// rt_status plus a message; nothing throws across the boundary.  There is no
// CPU fallback: without a HIP device rt_create fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <limits>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

namespace rtd {
hipError_t launch_trace(int sem, int mode, hipStream_t stream, const TraceParams& P, const void* blob,
                        uint32_t* accum, unsigned long long* st, int32_t* th, uint32_t* tm,
                        const uint32_t* tile_order, uint32_t* tile_cost, uint32_t n_workgroups, const uint32_t* tile_lists);
hipError_t launch_tile_lists(hipStream_t stream, const TraceParams& P, const void* blob, uint32_t* lists);
hipError_t launch_tile_prior(hipStream_t stream, const TraceParams& P, const uint32_t* lists, uint32_t* cost, uint32_t leaf_weight);
hipError_t launch_tile_plan(hipStream_t stream, uint32_t* cost, uint32_t* work, uint32_t n, uint32_t extra, uint32_t max_slices, float share, float early_share);
hipError_t launch_trace_fill(hipStream_t stream, int32_t* hit, uint32_t* mat, size_t n);
hipError_t launch_resolve(int sem, hipStream_t stream, const uint32_t* accum, float* out, uint32_t w, uint32_t h,
                          uint32_t samples, int gamma_sqrt, int flip_rows);
hipError_t launch_convert(int sem, hipStream_t stream, uint32_t* accum, float* image, uint32_t w, uint32_t h,
                          uint32_t samples, uint32_t n_samples);
hipError_t scratch_for_device(int device, uint32_t** out);
}  // namespace rtd

static_assert(sizeof(rt_material) == 32, "Material layout (shape.h:5-12)");
static_assert(sizeof(rt_shape) == 144, "Shape layout (shape.h:44-54)");
static_assert(offsetof(rt_shape, type) == 64 && offsetof(rt_shape, data) == 68 && offsetof(rt_shape, material) == 112,
              "Shape field offsets (RayTracing.glsl:92-97)");
static_assert(sizeof(rt_triangle) == 64 && offsetof(rt_triangle, a) == 16 && offsetof(rt_triangle, beta) == 28 &&
              offsetof(rt_triangle, m1) == 32 && offsetof(rt_triangle, m2) == 48,
              "SimpleTriangle layout (shape.h:103-108, RayTracing.glsl:10-16)");

struct rt_ctx {
    rt_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    std::vector<rt_shape> shapes;        // host copies of what was uploaded
    std::vector<rt_triangle> tris;
    std::vector<float> tri_colors;       // empty, or rgb per triangle
    bool blob_dirty = true;              // the device blob does not reflect them yet
    void* d_blob = nullptr;              size_t blob_cap = 0;     // bytes
    rtd::SceneLayout layout{};
    uint32_t n_spheres = 0, n_boxes = 0;
    float cam[RT_CAMERA_FLOATS] = {};
    bool have_camera = false;

    uint32_t* d_accum = nullptr;         // owned accumulator
    uint32_t* d_accum_ext = nullptr;     // caller-owned accumulator, if bound
    uint32_t width = 0, height = 0;      // dimensions of the accumulator in use
    uint32_t accum_samples = 0;
    float* d_image = nullptr;            // running mean (rt_convert)
    uint32_t image_w = 0, image_h = 0;

    uint32_t shard_index = 0, shard_count = 1, band_rows = 16;

    unsigned long long* d_stats = nullptr;

    // Plan of the next launch (which workgroup traces which tile, or which slice of a tile's
    // samples), made from what the tiles cost in the previous frame of the same tiling.
    uint32_t* d_tile_cost = nullptr;     // per tile: shader clocks of its slowest wave
    uint32_t* d_tile_order = nullptr;    // the plan: plan_units entries
    uint32_t tile_cap = 0;               // tiles allocated for (the plan holds 2 * tile_cap entries)
    uint32_t order_tiles_x = 0, order_tiles_y = 0;   // tiling the plan is valid for (0 = none yet)
    uint32_t plan_units = 0;             // workgroups of the planned launch
    uint32_t plan_max_slices = 1;        // the most slices a tile of that plan may have been cut into
    // What the plan was made from.  While a call renders the same view (scene, camera, samples, depth), the plan is
    // kept and no costs are collected; every REPLAN_EVERY-th such call renews it.
    struct PlanView { float cam[RT_CAMERA_FLOATS]; uint32_t scene_version, spp, max_depth, shard_index, shard_count, band_rows; };
    PlanView plan_view{};
    uint32_t plan_age = 0;               // calls that reused the plan since it was made
    bool discard_pending = false;        // rt_discard_accum: the next render call overwrites the accumulator

    // Candidate lists of the camera rays, one record per 8x8-pixel wave tile (rt_device.h): made by the first render
    // of a (scene, camera, frame size, row shard) and kept until one of them changes.
    struct ListKey {
        float cam[RT_CAMERA_FLOATS];
        uint32_t width, height, sample_mode, grid_n, shard_index, shard_count, band_rows, owned_rows, tiles_x, tiles_y, scene_version;
    };
    uint32_t* d_tile_lists = nullptr;
    uint32_t lists_cap = 0;              // wave tiles allocated for
    bool lists_valid = false;
    ListKey lists_key{};
    uint32_t scene_version = 0;          // counts the builds of the device copy of the scene
    // Whether the lists pay.  The second call in a row of a view that has lists runs the counting instance of the
    // kernel (the one rt_render_stats uses: the same words, a tenth slower) and its counters come back to pinned host
    // memory; a later call finds them there, and a view of which fewer than LISTS_MIN_SHARE of the waves' steps were
    // primary steps (long paths: an enclosed scene) goes on without lists.  A camera that moves every frame is never
    // measured and keeps its lists.
    unsigned long long* h_stats_copy = nullptr;      // pinned, ST_COUNT words
    hipEvent_t stats_copied = nullptr;               // recorded behind the copy
    uint32_t view_streak = 0;                        // calls in a row that found the lists made
    bool lists_measured = false, measure_pending = false, lists_useless = false;

    int compute_units = 0;
    uint32_t lds_limit = rtd::LDS_WORKGROUP_MAX_BYTES;   // what a workgroup can have of a CU's LDS on this device (hipDeviceProp_t::sharedMemPerBlock, at most 160 KiB)

    // This device's copy of the library's static scratch (rt_render_noalloc), looked up once in rt_create.
    uint32_t* d_scratch = nullptr;
    // Staging for entry points that return data to host memory (rt_resolve, rt_trace_debug): grown on demand, kept.
    void* d_stage = nullptr;             size_t stage_cap = 0;
    // Switches for measurements, read from the environment once, when the context is created.
    uint32_t group_waves = 4;           // waves per workgroup (RT_GROUP_WAVES: 1 .. 8, for measurements)
    bool env_no_tile_order = false, env_no_slices = false, env_mesh_in_lds = false, env_no_probe = false, env_no_tile_lists = false;
    uint32_t env_leaf = 0;               // RT_LEAF: triangles per tree leaf for every mesh (measurements); 0: chosen by mesh size
    uint32_t lists_min_spp = 0u;         // RT_LISTS_MIN_SPP: launches of fewer samples per pixel trace without the camera rays' lists (measurements, tests); 0: lists_min_spp_of()
    uint32_t env_tree_levels = 0;        // RT_TREE_LEVELS: at most this many tree levels, 4 .. 8 (measurements); 0: as many as the mesh needs
    uint32_t tree_top_max = rtd::TREE_TOP_MAX;   // RT_TREE_TOP: a tree gets levels until its top one has at most this many nodes (measurements, tests)
    float slice_share = 0.25f, slice_early = 0.5f;       // RT_SLICE_SHARE, RT_SLICE_EARLY (measurements)
    uint32_t slice_min_spp = 16u, slice_max = 16u;       // RT_SLICE_MIN_SPP, RT_SLICE_MAX
    uint32_t probe_spp = 0u;                             // RT_PROBE_SPP (0: the built-in PROBE_SPP)
    uint32_t probe_prior_weight = 3u;                    // RT_PROBE_PRIOR: what a candidate leaf adds to a tile's assumed cost when the probe is planned (0: the probe has no plan)
};

static thread_local std::string g_create_error;

static int fail(rt_ctx* ctx, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define RT_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,       \
                        "%s failed: %s", #call, hipGetErrorString(e_));                           \
    } while (0)

#define RT_REQUIRE(ctx, cond, msg)                                                                \
    do { if (!(cond)) return fail(ctx, RT_ERR_INVALID_ARG, "%s", msg); } while (0)

static uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
static uint32_t seed_key(uint64_t seed)
{
    return hash32((uint32_t)seed ^ hash32((uint32_t)(seed >> 32) ^ 0x85ebca6bu));
}

// The Newton reciprocal of the kernels (rt_kernels.hip, rcp_nr), on the host;
// this file is compiled with -ffp-contract=off, so it is the same function.
static float rcp_nr(float x)
{
    uint32_t bits; memcpy(&bits, &x, 4);
    bits = 0x7ef311c7u - bits;
    float y; memcpy(&y, &bits, 4);
    for (int k = 0; k < 3; ++k) {
        float e = std::fmaf(-x, y, 2.0f);
        y = y * e;
    }
    return y;
}

struct Word { float f[4]; };
static inline uint32_t& as_u32(float& f) { return reinterpret_cast<uint32_t&>(f); }


// ---- bounding-sphere tree over the triangles ---------------------------------
// A SimpleTriangle (shape.h:81-109) keeps one vertex, the plane and the dual basis;
// the other two vertices follow from m1.v1 = 1, m1.v2 = 0, m2.v2 = 1, m2.v1 = 0,
// a.v1 = a.v2 = 0:  v1 = (m2 x a) / det,  v2 = (a x m1) / det,  det = m1 . (m2 x a).
struct TriGeom { double v[3][3]; double c[3]; bool ok; };

static void build_tree(const std::vector<rt_triangle>& tris, const rtd::SceneLayout& L, const uint32_t* node_pad,
                       uint32_t n_tri_pad, std::vector<Word>& blob)
{
    const uint32_t n = (uint32_t)tris.size();
    const uint32_t leaf = 1u << L.leaf_shift;               // triangles per leaf of this mesh's tree
    std::vector<TriGeom> g(n);
    for (uint32_t i = 0; i < n; ++i) {
        const rt_triangle& t = tris[i];
        const double m1[3] = { t.m1[0], t.m1[1], t.m1[2] }, m2[3] = { t.m2[0], t.m2[1], t.m2[2] }, a[3] = { t.a[0], t.a[1], t.a[2] };
        const double m2xa[3] = { m2[1] * a[2] - m2[2] * a[1], m2[2] * a[0] - m2[0] * a[2], m2[0] * a[1] - m2[1] * a[0] };
        const double axm1[3] = { a[1] * m1[2] - a[2] * m1[1], a[2] * m1[0] - a[0] * m1[2], a[0] * m1[1] - a[1] * m1[0] };
        const double det = m1[0] * m2xa[0] + m1[1] * m2xa[1] + m1[2] * m2xa[2];
        TriGeom& q = g[i];
        q.ok = true;
        for (int k = 0; k < 3; ++k) {
            q.v[0][k] = t.position[k];
            q.v[1][k] = t.position[k] + m2xa[k] / det;
            q.v[2][k] = t.position[k] + axm1[k] / det;
            q.c[k] = (q.v[0][k] + q.v[1][k] + q.v[2][k]) / 3.0;
            if (!std::isfinite(q.v[1][k]) || !std::isfinite(q.v[2][k]) || !std::isfinite(q.v[0][k])) q.ok = false;
        }
    }
    // Order the triangles by recursive median splits along the longest axis of their centroids'
    // extent (degenerate ones last).  Split points fall on multiples of the largest tree block
    // (leaf * TREE_FAN^k) that is smaller than the range, so every node of the implicit
    // tree bounds one spatially compact run.  Measured against a Morton sort: the summed squared
    // leaf radius (the chance of a ray entering a leaf) falls by a quarter on the 292-triangle
    // bunny and by more than half on the 4 968-triangle one.
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::stable_partition(order.begin(), order.end(), [&](uint32_t x) { return g[x].ok; });
    uint32_t n_ok = 0;
    while (n_ok < n && g[order[n_ok]].ok) ++n_ok;
    struct Range { uint32_t lo, hi; };
    std::vector<Range> todo;
    todo.push_back({ 0u, n_ok });
    while (!todo.empty()) {
        const Range r = todo.back(); todo.pop_back();
        const uint32_t len = r.hi - r.lo;
        if (len <= leaf) continue;
        uint32_t block = leaf;
        while ((uint64_t)block * rtd::TREE_FAN < len) block *= rtd::TREE_FAN;
        // the multiple of `block` nearest to the middle, strictly inside the range
        uint32_t h = ((len / 2u + block / 2u) / block) * block;
        if (h == 0u) h = block;
        if (h >= len) h = len - (len > block ? ((len - 1u) % block) + 1u : len / 2u);
        if (h == 0u || h >= len) h = len / 2u;
        double blo[3] = { 1e300, 1e300, 1e300 }, bhi[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = r.lo; i < r.hi; ++i) for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], g[order[i]].c[k]); bhi[k] = std::max(bhi[k], g[order[i]].c[k]); }
        // Split along the axis whose two halves are the most compact: the sum, over the halves, of triangle count
        // times the squared half diagonal of the box around the half's vertices (what a ray's chance of touching
        // the half's bound grows with).  Ranges of more than 4 096 triangles take the longest axis of the
        // centroids' extent without trying all three.
        int ax = 0;
        if (bhi[1] - blo[1] > bhi[ax] - blo[ax]) ax = 1;
        if (bhi[2] - blo[2] > bhi[ax] - blo[ax]) ax = 2;
        auto sort_along = [&](int a) {
            std::stable_sort(order.begin() + r.lo, order.begin() + r.hi, [&](uint32_t x, uint32_t y) { return g[x].c[a] < g[y].c[a]; });
        };
        if (len <= 4096u) {
            double best_cost = 1e300;
            int best_ax = ax;
            for (int a = 0; a < 3; ++a) {
                sort_along(a);
                double cost = 0;
                for (int half = 0; half < 2; ++half) {
                    const uint32_t b0 = half ? r.lo + h : r.lo, b1 = half ? r.hi : r.lo + h;
                    double lo3[3] = { 1e300, 1e300, 1e300 }, hi3[3] = { -1e300, -1e300, -1e300 };
                    for (uint32_t i = b0; i < b1; ++i)
                        for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) { lo3[k] = std::min(lo3[k], g[order[i]].v[v][k]); hi3[k] = std::max(hi3[k], g[order[i]].v[v][k]); }
                    double d2 = 0;
                    for (int k = 0; k < 3; ++k) d2 += 0.25 * (hi3[k] - lo3[k]) * (hi3[k] - lo3[k]);
                    cost += d2 * (double)(b1 - b0);
                }
                if (cost < best_cost) { best_cost = cost; best_ax = a; }
            }
            ax = best_ax;
        }
        sort_along(ax);
        todo.push_back({ r.lo, r.lo + h });
        todo.push_back({ r.lo + h, r.hi });
    }

    const float nan = std::nanf("");
    // Sorted triangles, w0.w = index in the uploaded array; padding never hits (NaN plane).
    for (uint32_t s_ = 0; s_ < n_tri_pad; ++s_) {
        Word* w = &blob[L.tri_off + 4u * s_];
        if (s_ < n) {
            memcpy(w, &tris[order[s_]], sizeof(rt_triangle));
            as_u32(w[0].f[3]) = order[s_];
            as_u32(blob[L.inv_off + order[s_] / 4u].f[order[s_] % 4u]) = s_;
        } else {
            for (int k = 0; k < 4; ++k) w[k].f[0] = w[k].f[1] = w[k].f[2] = w[k].f[3] = nan;
            as_u32(w[0].f[3]) = 0xffffffffu;
        }
    }
    // Level l node i covers the sorted triangles [span*i, span*(i+1)), span = LEAF * FAN^l.
    uint32_t span = leaf;
    for (uint32_t l = 0; l < L.tree_levels; ++l, span *= rtd::TREE_FAN) {
        for (uint32_t i = 0; i < node_pad[l]; ++i) {
            Word& w = blob[L.node_off[l] + i];
            w.f[0] = w.f[1] = w.f[2] = nan; w.f[3] = 0.0f;
            if (i >= L.node_n[l]) continue;
            const uint64_t first = (uint64_t)span * i, last = std::min<uint64_t>(first + span, n);
            if (first >= last) continue;
            double blo[3] = { 1e300, 1e300, 1e300 }, bhi[3] = { -1e300, -1e300, -1e300 };
            bool all_ok = true;
            for (uint64_t s_ = first; s_ < last; ++s_) {
                const TriGeom& q = g[order[s_]];
                if (!q.ok) { all_ok = false; break; }
                for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) { blo[k] = std::min(blo[k], q.v[v][k]); bhi[k] = std::max(bhi[k], q.v[v][k]); }
            }
            if (!all_ok) {       // a degenerate triangle has no recoverable extent: never cull its node
                w.f[0] = w.f[1] = w.f[2] = 0.0f; w.f[3] = std::numeric_limits<float>::infinity();
                continue;
            }
            // Centre: that of the bounding box, moved towards the farthest vertex a shrinking step at a time
            // (Badoiu and Clarkson), which converges on the centre of the smallest enclosing ball; the better of
            // the two is kept.  Any centre is valid: the radius below is measured from the one chosen.
            double c[3] = { 0.5 * (blo[0] + bhi[0]), 0.5 * (blo[1] + bhi[1]), 0.5 * (blo[2] + bhi[2]) };
            {
                auto radius2 = [&](const double* q_) {
                    double r2 = 0;
                    for (uint64_t s_ = first; s_ < last; ++s_)
                        for (int v = 0; v < 3; ++v) {
                            double dd = 0;
                            for (int k = 0; k < 3; ++k) { const double e = g[order[s_]].v[v][k] - q_[k]; dd += e * e; }
                            r2 = std::max(r2, dd);
                        }
                    return r2;
                };
                double best_c[3] = { c[0], c[1], c[2] }, best_r2 = radius2(c), m[3] = { c[0], c[1], c[2] };
                const int steps = (last - first) <= 64 ? 256 : 64;
                for (int it = 1; it <= steps; ++it) {
                    double far2 = -1; const double* far_p = nullptr;
                    for (uint64_t s_ = first; s_ < last; ++s_)
                        for (int v = 0; v < 3; ++v) {
                            double dd = 0;
                            for (int k = 0; k < 3; ++k) { const double e = g[order[s_]].v[v][k] - m[k]; dd += e * e; }
                            if (dd > far2) { far2 = dd; far_p = g[order[s_]].v[v]; }
                        }
                    if (far2 < best_r2) { best_r2 = far2; for (int k = 0; k < 3; ++k) best_c[k] = m[k]; }
                    for (int k = 0; k < 3; ++k) m[k] += (far_p[k] - m[k]) / (double)(it + 1);
                }
                for (int k = 0; k < 3; ++k) c[k] = best_c[k];
            }
            const float cf[3] = { (float)c[0], (float)c[1], (float)c[2] };
            double r2 = 0;
            for (uint64_t s_ = first; s_ < last; ++s_) {
                const TriGeom& q = g[order[s_]];
                for (int v = 0; v < 3; ++v) {
                    double dd = 0;
                    for (int k = 0; k < 3; ++k) { const double e = q.v[v][k] - (double)cf[k]; dd += e * e; }
                    r2 = std::max(r2, dd);
                }
            }
            const double R = 1.01 * std::sqrt(r2) + 1e-4;
            w.f[0] = cf[0]; w.f[1] = cf[1]; w.f[2] = cf[2];
            w.f[3] = std::nextafter((float)R, std::numeric_limits<float>::infinity());
        }
    }
}

// Where everything goes in the blob (layout: rt_device.h) for the arrays uploaded so far; fails when the part that
// must be staged does not fit.  Cheap: run at every upload, so that a scene that cannot be rendered is refused there.
struct BlobPlan { rtd::SceneLayout L; uint32_t node_pad[rtd::MAX_TREE_LEVELS]; uint32_t n_tri_pad, n_sph, n_box, n_cube; bool colors; };
static int plan_blob(rt_ctx* ctx, BlobPlan& B)
{
    const uint32_t n_shapes = (uint32_t)ctx->shapes.size(), n_tris = (uint32_t)ctx->tris.size();
    uint32_t n_sph = 0, n_box = 0, n_cube = 0;
    for (const rt_shape& s : ctx->shapes) {
        if (s.type == RT_SHAPE_CIRCLE) ++n_sph; else if (s.type == RT_SHAPE_BOX) ++n_box;
        else if (s.type == RT_SHAPE_CUBE && ctx->cfg.semantics == RT_SEM_CPU) ++n_cube;
    }
    rtd::SceneLayout L{};
    L.n_shapes = n_shapes; L.n_sph_pad = (n_sph + 3u) & ~3u; L.n_sph = n_sph; L.n_box = n_box; L.n_cube = n_cube; L.n_tris = n_tris;
    L.rec_off = 0;
    L.sph_off = L.rec_off + 4u * (n_shapes + 1u);          // one record more: what a miss reads
    L.sid_off = L.sph_off + L.n_sph_pad;
    L.box_off = L.sid_off + L.n_sph_pad / 4u;
    // Bounding-sphere tree over the triangles (rt_device.h).
    // Leaves of a small mesh hold half as many triangles (rt_device.h).
    uint32_t leaf = n_tris <= rtd::SMALL_MESH_TRIS ? rtd::TREE_LEAF_SMALL : rtd::TREE_LEAF;
    if (ctx->env_leaf && n_tris / ctx->env_leaf < (1u << 24)) leaf = ctx->env_leaf;      // a leaf's number must fit 24 bits
    for (L.leaf_shift = 0; (1u << (L.leaf_shift + 1u)) <= leaf; ++L.leaf_shift) {}
    leaf = 1u << L.leaf_shift;
    const uint32_t n_tri_pad = (n_tris + leaf - 1u) / leaf * leaf;
    {
        uint32_t n = n_tri_pad / leaf;
        // Four levels at least (the walk's); more while the top one has too many nodes to scan (rt_device.h).
        L.tree_levels = 0;
        for (uint32_t l = 0; l < rtd::MAX_TREE_LEVELS; ++l) {
            L.node_n[l] = n;
            B.node_pad[l] = (n + rtd::TREE_FAN - 1u) / rtd::TREE_FAN * rtd::TREE_FAN;
            L.tree_levels = l + 1u;
            if (l + 1u >= rtd::TREE_LEVELS && n <= ctx->tree_top_max) break;
            n = (n + rtd::TREE_FAN - 1u) / rtd::TREE_FAN;
        }
        if (ctx->env_tree_levels) L.tree_levels = std::min(L.tree_levels, std::max(rtd::TREE_LEVELS, ctx->env_tree_levels));   // measurements
        if (ctx->cfg.rng != RT_RNG_HASH) L.tree_levels = rtd::TREE_LEVELS;      // the kernels of the shader's own stream walk four levels (the top one then is as long as it is)
        // The top level is scanned linearly, so it is not grouped any further.
    }
    L.cub_off = L.box_off + 3u * n_box;
    uint32_t off = L.cub_off + 4u * n_cube;
    for (int l = (int)L.tree_levels - 1; l >= 0; --l) { L.node_off[l] = off; off += B.node_pad[l]; }
    L.tri_off = off;
    const uint32_t staged_words = L.tri_off + 4u * n_tri_pad;        // what a workgroup stages when the mesh fits
    // The place of each uploaded triangle in the sorted list is looked up once per hit: it stays in global memory.
    L.inv_off = staged_words;
    L.total_words = L.inv_off + (n_tris + 3u) / 4u;
    if ((ctx->cfg.flags & RT_FLAG_LITERAL_GEOMETRY) && (n_box != 0 || n_tris != 0))
        return fail(ctx, RT_ERR_INVALID_ARG, "scene: RT_FLAG_LITERAL_GEOMETRY renders spheres and Cubes only (%u boxes, %u triangles uploaded)", n_box, n_tris);
    B.colors = !ctx->tri_colors.empty() && ctx->tri_colors.size() == 3u * (size_t)n_tris;
    const uint32_t wave_words = (n_tris || n_box) ? rtd::lds_wave_words(ctx->group_waves, false) : 0u;     // queues and slots of the spread tests
    if ((size_t)(L.node_off[L.tree_levels - 1u] + (n_tris ? rtd::lds_wave_words(ctx->group_waves, true) : wave_words)) * 16u > ctx->lds_limit)
        return fail(ctx, RT_ERR_INVALID_ARG, "scene: %u shapes need more than the %u bytes of LDS a workgroup can have on this device", n_shapes, ctx->lds_limit);
    if ((size_t)(staged_words + wave_words) * 16u <= std::min(rtd::LDS_MAX_BYTES, ctx->lds_limit) && ctx->env_mesh_in_lds && L.tree_levels == rtd::TREE_LEVELS) { L.lds_words = staged_words; L.tri_in_lds = 1; }
    else { L.lds_words = L.node_off[L.tree_levels - 1u]; L.tri_in_lds = 0; }     // the mesh and its tree stay in global memory
    // Per-triangle albedo: read only at a hit, so it stays in global memory behind everything else.
    if (B.colors) { L.tcol_off = L.total_words; L.total_words += n_tris; }
    B.L = L; B.n_tri_pad = n_tri_pad; B.n_sph = n_sph; B.n_box = n_box; B.n_cube = n_cube;
    return RT_OK;
}

// Repack the uploaded arrays into the blob the kernels read and copy it to the device.  Runs once per changed
// scene, at the first render (or rt_commit_scene) after the uploads: updateShapeBuffer() of the reference uploads
// shapes and triangles one after the other (ray_tracing_compute_shader.h:44-57), and the tree over the triangles
// is the expensive part.
static int build_blob_impl(rt_ctx* ctx)
{
    RT_HIP(ctx, hipSetDevice(ctx->device));
    BlobPlan B;
    int prc = plan_blob(ctx, B);
    if (prc) return prc;
    rtd::SceneLayout L = B.L;
    const uint32_t n_shapes = L.n_shapes, n_tris = L.n_tris, n_sph = B.n_sph, n_box = B.n_box, n_tri_pad = B.n_tri_pad;
    const uint32_t* node_pad = B.node_pad;
    const bool colors = B.colors;

    std::vector<Word> blob(L.total_words ? L.total_words : 1u);
    memset(blob.data(), 0, blob.size() * sizeof(Word));
    const float nan = std::nanf("");
    for (uint32_t i = 0; i < L.n_sph_pad; ++i) { Word& w = blob[L.sph_off + i]; w.f[0] = w.f[1] = w.f[2] = nan; w.f[3] = 0.0f; }
    for (uint32_t i = 0; i < L.n_sph_pad; ++i) as_u32(blob[L.sid_off + i / 4u].f[i % 4u]) = 0xffffffffu;
    uint32_t is = 0, ib = 0, ic = 0;
    for (uint32_t i = 0; i < n_shapes; ++i) {
        const rt_shape& s = ctx->shapes[i];
        const float cx = s.transformation[12], cy = s.transformation[13], cz = s.transformation[14];
        const float scale = s.transformation[0];          // radius (shape.h:60) or side (RayTracing.glsl:139)
        Word& r0 = blob[L.rec_off + 4u * i]; Word& r1 = blob[L.rec_off + 4u * i + 1u]; Word& r2 = blob[L.rec_off + 4u * i + 2u];
        Word& r3 = blob[L.rec_off + 4u * i + 3u];
        r0.f[0] = cx; r0.f[1] = cy; r0.f[2] = cz; r0.f[3] = scale;
        float r = s.material.color[0], g = s.material.color[1], b = s.material.color[2];
        if (ctx->cfg.semantics == RT_SEM_CPU && s.material.type == RT_MAT_LAMBERTIAN) {
            // Lambertian::scatter attenuates by 0.6 * albedo (scene.h:56): formed once here.
            r = r * 0.6f; g = g * 0.6f; b = b * 0.6f;
        }
        // What the event at this surface is (rt_device.h): class and emission factor.
        uint32_t cls = 0u; float emit = 0.0f;
        switch (s.material.type) {
        case RT_MAT_LIGHT: emit = 1.0f; break;
        case RT_MAT_LAMBERTIAN: cls = 1u; break;
        case RT_MAT_METAL: cls = 2u; break;
        case RT_MAT_DIELECTRIC: cls = 3u; break;
        case RT_MAT_DEBUG: cls = 4u; L.has_debug = 1; break;
        default:                                            // glsl:253,282-284 returns the throughput; the CPU tracer has no such material: black
            if (ctx->cfg.semantics == RT_SEM_GLSL) { emit = 1.0f; r = g = b = 1.0f; }
            break;
        }
        r1.f[0] = r; r1.f[1] = g; r1.f[2] = b;
        as_u32(r1.f[3]) = (s.material.type & 0xffffu) | ((s.type & 0xffu) << 16) | (cls << 24);
        r2.f[0] = cls == 2u && ctx->cfg.semantics == RT_SEM_CPU ? s.material.data[1] : 0.0f;   // fuzz: the CPU tracer's metal only (scene.h:71,78)
        r2.f[1] = emit;
        r2.f[2] = s.type == RT_SHAPE_CIRCLE ? 1.0f / std::fabs(scale) : rcp_nr(scale);      // sphere: first guess of its normal's 1 / length (the test squares the radius, so its sign means nothing: scene.h:118,131); box: boxNormalAt's 1 / side
        // A Box's sides along y and z: the shader's single side, or its own (RT_FLAG_BOX_SIDES).
        const bool own_sides = (ctx->cfg.flags & RT_FLAG_BOX_SIDES) != 0u;
        const float side_y = own_sides ? s.transformation[5] : scale, side_z = own_sides ? s.transformation[10] : scale;
        if (s.type == RT_SHAPE_BOX) { r3.f[2] = rcp_nr(side_y); r3.f[3] = rcp_nr(side_z); }
        if (s.material.type == RT_MAT_DIELECTRIC) {
            L.has_glass = 1;
            const float ior = s.material.data[0];
            const float t = (1.0f - ior) / (1.0f + ior);
            r3.f[0] = 1.0f / ior; r3.f[1] = t * t;
        }
        if (s.type == RT_SHAPE_CIRCLE) {
            Word& w = blob[L.sph_off + is];
            w.f[0] = cx; w.f[1] = cy; w.f[2] = cz; w.f[3] = scale * scale;
            as_u32(blob[L.sid_off + is / 4u].f[is % 4u]) = i;
            ++is;
        } else if (s.type == RT_SHAPE_BOX) {
            Word& w = blob[L.box_off + 3u * ib]; Word& w1 = blob[L.box_off + 3u * ib + 1u]; Word& w2 = blob[L.box_off + 3u * ib + 2u];
            w.f[0] = cx; w.f[1] = cy; w.f[2] = cz; w.f[3] = 1.0f / scale;
            as_u32(w1.f[0]) = i;
            w1.f[1] = 1.0f / side_y; w1.f[2] = 1.0f / side_z;
            // Half extents of a box around everything the test can accept (points up to 0.5001 of a side from the
            // centre along every axis, RayTracing.glsl:151), with slack for the rounding of the kernels' slab test.
            const float sides[3] = { scale, side_y, side_z };
            for (int k = 0; k < 3; ++k)
                w2.f[k] = std::nextafter((float)(0.5001 * 1.0001 * std::fabs((double)sides[k]) + 1e-4), std::numeric_limits<float>::infinity());
            ++ib;
        } else if (s.type == RT_SHAPE_CUBE && ctx->cfg.semantics == RT_SEM_CPU) {
            Word* w = &blob[L.cub_off + 4u * ic];
            for (int r = 0; r < 3; ++r) {
                w[r].f[0] = s.transformation[r]; w[r].f[1] = s.transformation[4 + r]; w[r].f[2] = s.transformation[8 + r];
                w[r].f[3] = s.transformation[12 + r];
            }
            as_u32(w[3].f[0]) = i;
            as_u32(r2.f[3]) = ic;
            ++ic;
        }
    }
    static_assert(sizeof(rt_triangle) == 4 * sizeof(Word), "a SimpleTriangle is four 16-byte words");
    if (n_tris) build_tree(ctx->tris, L, node_pad, n_tri_pad, blob);
    if (colors) {
        // Triangles scatter like Lambertians; under the CPU rules that is 0.6 * albedo (scene.h:56).
        const float k = ctx->cfg.semantics == RT_SEM_CPU ? 0.6f : 1.0f;
        for (uint32_t i = 0; i < n_tris; ++i)
            for (int c = 0; c < 3; ++c) blob[L.tcol_off + i].f[c] = ctx->tri_colors[3u * i + c] * k;
    }

    // The kernels may still be reading the old copy.
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t bytes = blob.size() * sizeof(Word);
    if (bytes > ctx->blob_cap) {
        RT_HIP(ctx, hipFree(ctx->d_blob));
        ctx->d_blob = nullptr; ctx->blob_cap = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_blob, bytes));
        ctx->blob_cap = bytes;
    }
    RT_HIP(ctx, hipMemcpy(ctx->d_blob, blob.data(), bytes, hipMemcpyHostToDevice));
    ctx->layout = L;
    ctx->n_spheres = n_sph; ctx->n_boxes = n_box;
    ctx->blob_dirty = false;
    ++ctx->scene_version;
    return RT_OK;
}

static int build_blob(rt_ctx* ctx)
{
    try { return build_blob_impl(ctx); }
    catch (...) { return fail(ctx, RT_ERR_OUT_OF_MEMORY, "scene upload: host allocation failed"); }
}

// An upload changed the host arrays: check that the scene can still be laid out; the blob itself is rebuilt later.
static int scene_changed(rt_ctx* ctx)
{
    ctx->blob_dirty = true;
    BlobPlan B;
    return plan_blob(ctx, B);
}

extern "C" {

int rt_create(rt_ctx** out, const rt_config* cfg)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: out is NULL");
    *out = nullptr;
    rt_config c{};
    c.struct_size = sizeof(rt_config);
    c.triangle_albedo[0] = 1.0f;           // RayTracing.glsl:275
    if (cfg) {
        if (cfg->struct_size != sizeof(rt_config))
            return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: struct_size %u, expected %zu", cfg->struct_size, sizeof(rt_config));
        c = *cfg;
    }
    if (c.semantics > RT_SEM_GLSL) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: unknown semantics %u", c.semantics);
    if (c.rng > RT_RNG_GLSL) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: unknown rng %u", c.rng);
    if ((c.flags & RT_FLAG_UNIT_GAUSS) && (c.semantics != RT_SEM_CPU || c.rng != RT_RNG_HASH))
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: RT_FLAG_UNIT_GAUSS needs RT_SEM_CPU and RT_RNG_HASH");
    if ((c.flags & RT_FLAG_LITERAL_GEOMETRY) && (c.semantics != RT_SEM_CPU || c.rng != RT_RNG_HASH || (c.flags & RT_FLAG_UNIT_GAUSS)))
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: RT_FLAG_LITERAL_GEOMETRY needs RT_SEM_CPU and RT_RNG_HASH, without RT_FLAG_UNIT_GAUSS");
    if (c.sample_mode > RT_SAMPLE_GRID_COMPAT) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: unknown sample_mode %u", c.sample_mode);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "rt_create: no HIP device (%s)", e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (c.device < 0 || c.device >= count)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: device %d out of range [0,%d)", c.device, count);
    e = hipSetDevice(c.device);
    if (e != hipSuccess) return fail(nullptr, RT_ERR_HIP, "hipSetDevice(%d): %s", c.device, hipGetErrorString(e));
    rt_ctx* ctx = new (std::nothrow) rt_ctx();
    if (!ctx) return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "rt_create: host allocation failed");
    ctx->cfg = c;
    ctx->device = c.device;
    e = hipMalloc(&ctx->d_stats, sizeof(unsigned long long) * rtd::ST_COUNT);
    if (e != hipSuccess) { delete ctx; return fail(nullptr, RT_ERR_OUT_OF_MEMORY, "hipMalloc(stats): %s", hipGetErrorString(e)); }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, c.device);
    if (e != hipSuccess) { (void)hipFree(ctx->d_stats); delete ctx; return fail(nullptr, RT_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e)); }
    ctx->compute_units = prop.multiProcessorCount;
    if (prop.sharedMemPerBlock > 0) ctx->lds_limit = (uint32_t)std::min<size_t>(prop.sharedMemPerBlock, rtd::LDS_WORKGROUP_MAX_BYTES);
    // This device's address of the library's static scratch (the device is current here); rt_render_noalloc uses it.
    e = rtd::scratch_for_device(c.device, &ctx->d_scratch);
    if (e != hipSuccess) { (void)hipFree(ctx->d_stats); delete ctx; return fail(nullptr, RT_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)); }
    auto env_on = [](const char* name) { const char* v = getenv(name); return v && *v && strcmp(v, "0") != 0; };
    if (const char* gw = getenv("RT_GROUP_WAVES")) { const int n = atoi(gw); if (n >= 1 && n <= (int)rtd::MAX_GROUP_WAVES) ctx->group_waves = (uint32_t)n; }
    ctx->env_no_tile_order = env_on("RT_NO_TILE_ORDER");
    ctx->env_no_slices = env_on("RT_NO_SLICES");
    ctx->env_mesh_in_lds = env_on("RT_MESH_IN_LDS");
    if (const char* v = getenv("RT_LEAF")) { const int n = atoi(v); if (n >= 1 && n <= (int)rtd::TREE_LEAF && (n & (n - 1)) == 0) ctx->env_leaf = (uint32_t)n; }
    ctx->env_no_probe = env_on("RT_NO_PROBE");
    if (const char* v = getenv("RT_LISTS_MIN_SPP")) { const int n = atoi(v); if (n >= 1) ctx->lists_min_spp = (uint32_t)n; }
    if (const char* v = getenv("RT_PROBE_PRIOR")) { const int n = atoi(v); if (n >= 0 && n <= 1000) ctx->probe_prior_weight = (uint32_t)n; }
    if (const char* v = getenv("RT_TREE_TOP")) { const int n = atoi(v); if (n >= 1 && n <= 4096) ctx->tree_top_max = (uint32_t)n; }
    if (const char* v = getenv("RT_TREE_LEVELS")) { const int n = atoi(v); if (n >= (int)rtd::TREE_LEVELS && n <= (int)rtd::MAX_TREE_LEVELS) ctx->env_tree_levels = (uint32_t)n; }
    ctx->env_no_tile_lists = env_on("RT_NO_TILE_LISTS");
    if (const char* v = getenv("RT_SLICE_SHARE")) { const float f = (float)atof(v); if (f > 0.0f) ctx->slice_share = f; }
    if (const char* v = getenv("RT_SLICE_EARLY")) { const float f = (float)atof(v); if (f > 0.0f) ctx->slice_early = f; }
    if (const char* v = getenv("RT_SLICE_MIN_SPP")) { const int n = atoi(v); if (n >= 1) ctx->slice_min_spp = (uint32_t)n; }
    if (const char* v = getenv("RT_PROBE_SPP")) { const int n = atoi(v); if (n >= 1 && n <= 32) ctx->probe_spp = (uint32_t)n; }
    if (const char* v = getenv("RT_SLICE_MAX")) { const int n = atoi(v); if (n >= 1 && n <= 16) ctx->slice_max = (uint32_t)n; }
    *out = ctx;
    return RT_OK;
}

void rt_destroy(rt_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->d_blob);
    (void)hipFree(ctx->d_accum);
    (void)hipFree(ctx->d_image);
    (void)hipFree(ctx->d_stats);
    (void)hipFree(ctx->d_tile_cost);
    (void)hipFree(ctx->d_tile_order);
    (void)hipFree(ctx->d_tile_lists);
    if (ctx->h_stats_copy) (void)hipHostFree(ctx->h_stats_copy);
    if (ctx->stats_copied) (void)hipEventDestroy(ctx->stats_copied);
    (void)hipFree(ctx->d_stage);
    delete ctx;
}

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_set_stream(rt_ctx* ctx, void* hip_stream)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (s != ctx->stream) {
        // what the old stream is still writing (the launch plan, the camera rays' candidate lists) the new one will read
        RT_HIP(ctx, hipSetDevice(ctx->device));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->stream = s;
    return RT_OK;
}

int rt_upload_shapes(rt_ctx* ctx, const rt_shape* shapes, uint32_t n)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, n == 0 || shapes, "rt_upload_shapes: shapes is NULL");
    try { ctx->shapes.assign(shapes, shapes + n); } catch (...) { return fail(ctx, RT_ERR_OUT_OF_MEMORY, "rt_upload_shapes: host allocation failed"); }
    return scene_changed(ctx);
}

int rt_upload_triangles(rt_ctx* ctx, const rt_triangle* tris, uint32_t n)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, n == 0 || tris, "rt_upload_triangles: triangles is NULL");
    RT_REQUIRE(ctx, n <= (1u << 26) - 4u, "rt_upload_triangles: more than 2^26 triangles (a leaf's number must fit 24 bits)");
    try { ctx->tris.assign(tris, tris + n); } catch (...) { return fail(ctx, RT_ERR_OUT_OF_MEMORY, "rt_upload_triangles: host allocation failed"); }
    ctx->tri_colors.clear();
    return scene_changed(ctx);
}

int rt_upload_triangle_colors(rt_ctx* ctx, const float* rgb, uint32_t n)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, n == 0 || rgb, "rt_upload_triangle_colors: rgb is NULL");
    if (n != 0 && n != ctx->tris.size())
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_triangle_colors: %u colours for %zu uploaded triangles", n, ctx->tris.size());
    try { ctx->tri_colors.assign(rgb, rgb + 3u * (size_t)n); } catch (...) { return fail(ctx, RT_ERR_OUT_OF_MEMORY, "rt_upload_triangle_colors: host allocation failed"); }
    return scene_changed(ctx);
}

int rt_commit_scene(rt_ctx* ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    return ctx->blob_dirty ? build_blob(ctx) : RT_OK;
}

int rt_set_camera(rt_ctx* ctx, const float data[RT_CAMERA_FLOATS])
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, data, "rt_set_camera: data is NULL");
    memcpy(ctx->cam, data, sizeof ctx->cam);
    ctx->have_camera = true;
    return RT_OK;
}

int rt_set_row_shard(rt_ctx* ctx, uint32_t shard_index, uint32_t shard_count, uint32_t band_rows)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, shard_count >= 1 && shard_index < shard_count, "rt_set_row_shard: need shard_index < shard_count");
    RT_REQUIRE(ctx, band_rows >= 1, "rt_set_row_shard: band_rows must be >= 1");
    ctx->shard_index = shard_index; ctx->shard_count = shard_count; ctx->band_rows = band_rows;
    return RT_OK;
}

int rt_bind_accum(rt_ctx* ctx, void* device_words, uint32_t width, uint32_t height)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!device_words) { ctx->d_accum_ext = nullptr; if (!ctx->d_accum) { ctx->width = ctx->height = 0; } ctx->accum_samples = 0; return RT_OK; }
    RT_REQUIRE(ctx, width > 0 && height > 0, "rt_bind_accum: empty image");
    RT_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(device_words) & 15u) == 0, "rt_bind_accum: buffer must be 16-byte aligned");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_accum) { RT_HIP(ctx, hipFree(ctx->d_accum)); ctx->d_accum = nullptr; }
    ctx->d_accum_ext = static_cast<uint32_t*>(device_words);
    ctx->width = width; ctx->height = height;
    ctx->accum_samples = 0;
    ctx->discard_pending = false;        // the new buffer is used as it is
    return RT_OK;
}

static uint32_t* accum_ptr(rt_ctx* ctx) { return ctx->d_accum_ext ? ctx->d_accum_ext : ctx->d_accum; }

// Makes sure an accumulator of width x height exists (zeroed when created).
static int ensure_accum(rt_ctx* ctx, uint32_t width, uint32_t height)
{
    if (ctx->d_accum_ext) {
        if (ctx->width != width || ctx->height != height)
            return fail(ctx, RT_ERR_INVALID_ARG, "render size %ux%u differs from the bound accumulator %ux%u", width, height, ctx->width, ctx->height);
        return RT_OK;
    }
    if (ctx->d_accum && ctx->width == width && ctx->height == height) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_accum) { RT_HIP(ctx, hipFree(ctx->d_accum)); ctx->d_accum = nullptr; }
    const size_t bytes = (size_t)width * height * 16u;
    RT_HIP(ctx, hipMalloc(&ctx->d_accum, bytes));
    RT_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, bytes, ctx->stream));
    ctx->width = width; ctx->height = height; ctx->accum_samples = 0;
    return RT_OK;
}

static uint32_t isqrt_u32(uint32_t n)
{
    uint32_t r = (uint32_t)std::sqrt((double)n);
    while ((uint64_t)r * r > n) --r;
    while ((uint64_t)(r + 1) * (r + 1) <= n) ++r;
    return r;
}

static uint32_t owned_rows(const rt_ctx* ctx, uint32_t height)
{
    if (ctx->shard_count <= 1) return height;
    const uint32_t band = ctx->band_rows;
    const uint32_t n_bands = (height + band - 1) / band;
    uint32_t rows = 0;
    for (uint32_t b = ctx->shard_index; b < n_bands; b += ctx->shard_count) {
        uint32_t y0 = b * band, y1 = y0 + band; if (y1 > height) y1 = height;
        rows += y1 - y0;
    }
    return rows;
}

static int fill_params(rt_ctx* ctx, rtd::TraceParams& P, uint32_t width, uint32_t height, uint32_t spp,
                       uint32_t max_depth, uint64_t seed, uint32_t first_sample, uint32_t grid_n)
{
    RT_REQUIRE(ctx, width > 0 && height > 0, "render: empty image");
    RT_REQUIRE(ctx, (uint64_t)width * height <= 0x0fffffffu, "render: image too large (pixel index must fit 28 bits)");
    RT_REQUIRE(ctx, max_depth >= 1 && max_depth <= 1024, "render: max_depth must be in [1,1024]");
    RT_REQUIRE(ctx, (uint64_t)first_sample + spp <= 0xffffffffu, "render: sample range overflows");
    RT_REQUIRE(ctx, ctx->have_camera, "render: rt_set_camera has not been called");
    memset(&P, 0, sizeof P);
    memcpy(P.cam, ctx->cam, sizeof P.cam);
    for (int k = 0; k < 3; ++k)     // triangles are Lambertian (glsl:275); CPU rules attenuate by 0.6 * albedo (scene.h:56)
        P.tri_albedo[k] = ctx->cfg.semantics == RT_SEM_CPU ? ctx->cfg.triangle_albedo[k] * 0.6f : ctx->cfg.triangle_albedo[k];
    P.width = width; P.height = height;
    P.scene = ctx->layout;
    P.spp = spp; P.first_sample = first_sample; P.max_depth = max_depth;
    P.seed_key = seed_key(seed);
    P.sample_mode = ctx->cfg.sample_mode; P.flags = ctx->cfg.flags;
    if (P.sample_mode != RT_SAMPLE_JITTER) {
        if (grid_n == 0) {
            grid_n = isqrt_u32(first_sample + spp);
            RT_REQUIRE(ctx, grid_n * grid_n == first_sample + spp, "render: grid sample mode needs first_sample + spp to be a square (or pass grid_n)");
        }
        RT_REQUIRE(ctx, grid_n >= 1, "render: grid_n must be >= 1");
    }
    P.grid_n = grid_n ? grid_n : 1u;
    if (ctx->shard_count <= 1) { P.shard_index = 0; P.shard_count = 1; P.band_rows = height; }
    else { P.shard_index = ctx->shard_index; P.shard_count = ctx->shard_count; P.band_rows = ctx->band_rows; }
    P.owned_rows = owned_rows(ctx, height);
    P.rng_mode = ctx->cfg.rng;
    P.glsl_seed32 = (uint32_t)seed;
    P.inv_w = 1.0f / (float)width; P.inv_h = 1.0f / (float)height;
    P.tiles_x = (P.width + rtd::TILE_W - 1) / rtd::TILE_W;
    P.tiles_y = (P.owned_rows + rtd::TILE_H - 1) / rtd::TILE_H;
    // Four waves per workgroup share one staged copy of the scene.  Single-wave workgroups (RT_GROUP_WAVES=1, for
    // measurements) free their slot the moment the wave ends: no faster on the four-sphere scene (1.34 against
    // 1.36 ms), 2.2 times slower on the mesh scene, where the staged scene then limits the waves per CU.
    P.group_waves = ctx->group_waves;
    // tile_plan_kernel counts the entries of its three classes in fields of 21 bits, and a sliced frame has up to
    // twice as many entries as tiles.
    RT_REQUIRE(ctx, (uint64_t)P.tiles_x * P.tiles_y < (1u << 20), "render: more than 2^20 tiles of 16x16 pixels");
    return RT_OK;
}

// Makes room for the per-tile cost and order arrays of a tiles_x x tiles_y launch.
static int ensure_tiles(rt_ctx* ctx, const rtd::TraceParams& P)
{
    const uint32_t n = P.tiles_x * P.tiles_y;
    if (n > ctx->tile_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        RT_HIP(ctx, hipFree(ctx->d_tile_cost)); ctx->d_tile_cost = nullptr;
        RT_HIP(ctx, hipFree(ctx->d_tile_order)); ctx->d_tile_order = nullptr;
        ctx->tile_cap = 0; ctx->order_tiles_x = ctx->order_tiles_y = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_tile_cost, sizeof(uint32_t) * n));
        RT_HIP(ctx, hipMalloc(&ctx->d_tile_order, sizeof(uint32_t) * 2u * (size_t)n));
        ctx->tile_cap = n;
    }
    return RT_OK;
}

// How many slices a tile's sample range may be cut into.  Slices are added with integer atomics, which only the
// shader's sums allow (a float sum depends on the order of its terms); each keeps at least 16 samples.
static uint32_t slices_allowed(const rt_ctx* ctx, uint32_t spp)
{
    if (ctx->cfg.semantics != RT_SEM_GLSL || spp >= (1u << 24) || ctx->env_no_slices) return 1u;
    return std::min(ctx->slice_max, std::max(1u, spp / ctx->slice_min_spp));
}

// Turns the tile costs in `cost` into the plan of a launch of `spp` samples per pixel -> its number of workgroups.
static int make_plan(rt_ctx* ctx, uint32_t* cost, uint32_t* order, uint32_t n_tiles, uint32_t spp, uint32_t* plan_units)
{
    const uint32_t max_slices = slices_allowed(ctx, spp);
    const uint32_t extra = max_slices > 1u ? n_tiles : 0u;
    // A tile is cut when it costs more than a quarter of an even share (measured: 0.5 -> 48 ms, 0.35 -> 45,
    // 0.25 -> 41, 0.125 -> 41 on the 292-triangle mesh scene; no effect on scenes without expensive tiles).
    const float share = ctx->slice_share;
    // Uncut tiles above half the cutting threshold are as long as the slices and follow them directly (measured on the
    // same scene: 41.0 ms without, 39.1-40.3 ms with; putting them before the slices instead: the same).
    const float early_share = ctx->slice_early;
    RT_HIP(ctx, rtd::launch_tile_plan(ctx->stream, cost, order, n_tiles, extra, max_slices, share, early_share));
    *plan_units = n_tiles + extra;
    return RT_OK;
}

// Primary steps trade steps for work: a wave takes more, cheaper steps.  That pays while the machine is kept full; a
// launch of a few samples per pixel on a small frame is over before it is (measured on 512 x 512, 16 samples, four
// spheres: 58 us without the lists, 77 us with them), so such a launch traces without them.  Where the waves outnumber
// the machine's 8 192 slots at least twice, a scene of spheres profits from 16 samples per pixel on (1080p, four
// spheres: 0.176 -> 0.167 ms; 8 samples: equal) and one of many spheres from 8 on (2048 x 2048, 65 spheres, 8 samples:
// 1.04 -> 0.74 ms); a scene with boxes or a mesh from 32 on (the mesh scene at 8 and 16 samples: 2 % slower with them).
static uint32_t lists_min_spp_of(const rt_ctx* ctx, const rtd::TraceParams& P)
{
    if (ctx->lists_min_spp) return ctx->lists_min_spp;
    const bool spheres_only = P.scene.n_box == 0 && P.scene.n_tris == 0;
    if (!spheres_only || 4u * P.tiles_x * P.tiles_y < 16384u) return 32u;
    return P.scene.n_sph >= 8u ? 8u : 16u;
}
// ... and while a fair share of a wave's steps are primary steps: where paths are long (an enclosed scene: one camera
// ray for fourteen segments) the steps' bookkeeping costs more than the few primary steps save (measured: 1.2 %).
constexpr double LISTS_MIN_SHARE = 0.10;
// The candidate lists of the camera rays for this launch's tiling (nullptr: the scene has no use for them, or they are
// switched off): made when the scene, the camera, the frame size or the row shard changed since they were last made.
static int ensure_lists(rt_ctx* ctx, const rtd::TraceParams& P, const uint32_t** out)
{
    *out = nullptr;
    // (the masks of a list hold 128 spheres — 960 when there is no mesh — and 32 boxes: a scene beyond them has no lists)
    if (ctx->env_no_tile_lists || P.rng_mode != 0 || (P.flags & RT_FLAG_LITERAL_GEOMETRY) != 0u ||
        P.scene.n_sph > rtd::tlist_max_spheres(P.scene.n_tris != 0u) || P.scene.n_box > rtd::TLIST_MAX_BOXES ||
        (P.scene.n_sph < 2u && P.scene.n_box == 0 && P.scene.n_tris == 0)) return RT_OK;
    rt_ctx::ListKey key{};
    memcpy(key.cam, P.cam, sizeof key.cam);
    key.width = P.width; key.height = P.height; key.sample_mode = P.sample_mode; key.grid_n = P.grid_n;
    key.shard_index = P.shard_index; key.shard_count = P.shard_count; key.band_rows = P.band_rows; key.owned_rows = P.owned_rows;
    key.tiles_x = P.tiles_x; key.tiles_y = P.tiles_y; key.scene_version = ctx->scene_version;
    const uint32_t n_wt = 4u * P.tiles_x * P.tiles_y;
    if (n_wt > ctx->lists_cap) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        RT_HIP(ctx, hipFree(ctx->d_tile_lists)); ctx->d_tile_lists = nullptr; ctx->lists_cap = 0; ctx->lists_valid = false;
        RT_HIP(ctx, hipMalloc(&ctx->d_tile_lists, sizeof(uint32_t) * rtd::TLIST_WORDS * (size_t)n_wt));
        ctx->lists_cap = n_wt;
    }
    // the counters of the measuring call, once they have come back (this only steers speed: any choice gives the same words)
    if (ctx->measure_pending && hipEventQuery(ctx->stats_copied) == hipSuccess) {
        const unsigned long long steps = ctx->h_stats_copy[rtd::ST_WAVE_ITERS], psteps = ctx->h_stats_copy[rtd::ST_PRIMARY_STEPS];
        if (steps != 0) ctx->lists_useless = (double)psteps < LISTS_MIN_SHARE * (double)steps;
        ctx->measure_pending = false;
    }
    if (!ctx->lists_valid || memcmp(&key, &ctx->lists_key, sizeof key) != 0) {
        RT_HIP(ctx, rtd::launch_tile_lists(ctx->stream, P, ctx->d_blob, ctx->d_tile_lists));
        ctx->lists_key = key; ctx->lists_valid = true;
        ctx->lists_useless = false; ctx->lists_measured = false; ctx->measure_pending = false;   // a new view: to be found out again
        ctx->view_streak = 0;
    } else {
        ++ctx->view_streak;
    }
    if (!ctx->lists_useless) *out = ctx->d_tile_lists;
    return RT_OK;
}

// A render call of at least this many samples per pixel that has no plan yet traces its first PROBE_SPP samples in
// the built-in order, makes the plan from what the tiles cost there, and traces the rest by that plan.
constexpr uint32_t PROBE_MIN_SPP = 64u, PROBE_SPP = 8u;      // measured: 16 -> 8 takes the cold frame of the mesh scene from 17.0 to 16.4 ms, of the 4 968-triangle mesh from 31.5 to 26.7

// Traces samples [P.first_sample, P.first_sample + P.spp) into `accum`: one launch by the plan in `order` (have_plan),
// or — without a plan — a probe launch, the plan made from it, and the rest; without a plan and with few samples,
// one launch in the built-in order.  `cost` (n_tiles words, zero on entry) collects what the tiles cost; with
// plan_after the plan of the next frame is made from it and left in `order` (and `cost` is zero again).
// *plan_units: in, the workgroups of the plan in `order`; out, those of the plan left there.
static int trace_frame(rt_ctx* ctx, rtd::TraceParams P, uint32_t* accum, int mode, uint32_t* cost, uint32_t* order,
                       bool have_plan, bool allow_probe, bool plan_after, uint32_t* plan_units, int32_t* d_timeline, const uint32_t* lists)
{
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    const uint32_t spp_all = P.spp;
    if (!have_plan && allow_probe && P.spp >= PROBE_MIN_SPP) {
        rtd::TraceParams probe = P;
        const uint32_t probe_spp = ctx->probe_spp ? ctx->probe_spp : PROBE_SPP;
        probe.spp = probe_spp;
        probe.resolve_out = nullptr;                       // the frame ends with the launch after this one
        // Under the shader's rules the probe has a plan of its own when the view has candidate lists of a large mesh: the
        // tiles with many candidate leaves are cut into slices of a sample or two (tile_prior_kernel).  Any plan gives the
        // same words.  (Measured: the cold frame of the 4 968-triangle mesh 26.5 -> 21.8 ms; with the 292-triangle mesh,
        // whose probe ends on tiles the lists say nothing about, 15.1 -> 15.6: hence the border.)
        const uint32_t* probe_order = nullptr;
        uint32_t probe_units = n_tiles;
        const uint32_t probe_slices = std::min(probe_spp, ctx->slice_max);
        if (lists && P.scene.n_tris > rtd::SMALL_MESH_TRIS && ctx->probe_prior_weight != 0u && probe_slices > 1u &&
            ctx->cfg.semantics == RT_SEM_GLSL && !ctx->env_no_slices) {
            RT_HIP(ctx, rtd::launch_tile_prior(ctx->stream, P, lists, cost, ctx->probe_prior_weight));
            RT_HIP(ctx, rtd::launch_tile_plan(ctx->stream, cost, order, n_tiles, n_tiles, probe_slices, ctx->slice_share, ctx->slice_early));   // leaves the costs at zero
            probe_order = order; probe_units = 2u * n_tiles;
            if (probe.overwrite) {
                // the slices' sums meet through atomics: they need the zeros an overwriting launch would not
                RT_HIP(ctx, hipMemsetAsync(accum, 0, (size_t)P.width * P.height * 16u, ctx->stream));
                probe.overwrite = 0u;
            }
        }
        P.overwrite = 0u;                                  // the launch after the probe adds to what the probe stores
        RT_HIP(ctx, rtd::launch_trace((int)ctx->cfg.semantics, mode, ctx->stream, probe, ctx->d_blob, accum, ctx->d_stats, nullptr, nullptr,
                                      probe_order, cost, probe_units, probe_spp >= lists_min_spp_of(ctx, P) ? lists : nullptr));
        P.first_sample += probe_spp; P.spp -= probe_spp;
        int rc = make_plan(ctx, cost, order, n_tiles, P.spp, plan_units);
        if (rc) return rc;
        have_plan = true;
        // The plan just made may cut tiles into slices (only the device knows): their sums go through atomics, which
        // the resolve in the launch cannot follow.  The caller resolves afterwards (see render_impl).
        if (*plan_units != n_tiles) P.resolve_out = nullptr;
    }
    RT_HIP(ctx, rtd::launch_trace((int)ctx->cfg.semantics, mode, ctx->stream, P, ctx->d_blob, accum, ctx->d_stats, d_timeline, nullptr,
                                  have_plan ? order : nullptr, cost, have_plan ? *plan_units : n_tiles, P.spp >= lists_min_spp_of(ctx, P) ? lists : nullptr));
    if (plan_after) return make_plan(ctx, cost, order, n_tiles, spp_all, plan_units);
    return RT_OK;
}

constexpr uint32_t REPLAN_EVERY = 32u;

struct ResolveInto { float* rgba_device; int gamma_sqrt, flip_rows; };

static int render_impl(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
                       uint64_t seed, uint32_t first_sample, uint32_t grid_n, rt_stats* stats,
                       int32_t* timeline_host = nullptr, uint32_t timeline_waves = 0, const ResolveInto* resolve = nullptr)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = RT_OK;
    if (ctx->blob_dirty) { rc = build_blob(ctx); if (rc) return rc; }
    rtd::TraceParams P;
    rc = fill_params(ctx, P, width, height, spp, max_depth, seed, first_sample, grid_n);
    if (rc) return rc;
    rc = ensure_accum(ctx, width, height);
    if (rc) return rc;
    if (stats) RT_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * rtd::ST_COUNT, ctx->stream));
    bool resolved_in_launch = false;
    bool discard = ctx->discard_pending;                 // the accumulator's contents are dropped (rt_discard_accum)
    if (discard) ctx->accum_samples = 0;
    if (discard && !(spp > 0 && P.owned_rows > 0)) {
        RT_HIP(ctx, hipMemsetAsync(accum_ptr(ctx), 0, (size_t)ctx->width * ctx->height * 16u, ctx->stream));
        ctx->discard_pending = false;
    }
    if (spp > 0 && P.owned_rows > 0) {
        rc = ensure_tiles(ctx, P);
        if (rc) return rc;
        const uint32_t n_tiles = P.tiles_x * P.tiles_y;
        // The plan made from the last frame holds while the tiling stays the same; a new tiling starts from clean
        // costs.  A plan that cuts tiles into slices of the sample range (it then has more entries than tiles) is
        // dropped for a call of 2^24 samples or more, which the slices' arithmetic (spp * slice) does not hold.
        // So is a plan whose slices this call's samples cannot fill (a plan of 16 slices made for 256 samples,
        // then a preview of 4): most of its workgroups would start, stage the scene and find nothing to do.
        const bool have_plan = ctx->order_tiles_x == P.tiles_x && ctx->order_tiles_y == P.tiles_y && !ctx->env_no_tile_order &&
                               (ctx->plan_units == n_tiles || (spp < (1u << 24) && slices_allowed(ctx, spp) >= ctx->plan_max_slices));
        if (!have_plan) RT_HIP(ctx, hipMemsetAsync(ctx->d_tile_cost, 0, sizeof(uint32_t) * n_tiles, ctx->stream));
        uint32_t plan_units = have_plan ? ctx->plan_units : n_tiles;
        // The same view as the plan was made from: the plan stays and this call collects no costs (a sphere scene's
        // plan never changes; a plan of a mesh scene only orders and cuts the tiles, any plan gives the same words).
        rt_ctx::PlanView view{};
        memcpy(view.cam, P.cam, sizeof view.cam);
        view.scene_version = ctx->scene_version; view.spp = spp; view.max_depth = max_depth;
        view.shard_index = P.shard_index; view.shard_count = P.shard_count; view.band_rows = P.band_rows;
        const bool replan = !have_plan || memcmp(&view, &ctx->plan_view, sizeof view) != 0 || ctx->plan_age + 1u >= REPLAN_EVERY ||
                            stats != nullptr || timeline_host != nullptr;
        // A launch whose plan cuts no tile (every pixel's sum is formed by one lane) can start from zero instead of
        // from the accumulator, and can end with the frame's resolve.
        const bool first_sliced = have_plan && plan_units != n_tiles;
        if (discard) {
            if (first_sliced) RT_HIP(ctx, hipMemsetAsync(accum_ptr(ctx), 0, (size_t)ctx->width * ctx->height * 16u, ctx->stream));
            else P.overwrite = 1u;
            ctx->discard_pending = false;
        }
        if (resolve && !first_sliced && !timeline_host) {
            P.resolve_out = resolve->rgba_device; P.resolve_samples = ctx->accum_samples + spp;
            P.resolve_gamma = resolve->gamma_sqrt ? 1u : 0u; P.resolve_flip = resolve->flip_rows ? 1u : 0u;
        }
        int32_t* d_timeline = nullptr;
        if (timeline_host) {
            if (timeline_waves < 4u * plan_units) return fail(ctx, RT_ERR_INVALID_ARG, "timeline buffer holds %u waves, the launch has %u", timeline_waves, 4u * plan_units);
            RT_HIP(ctx, hipMalloc(&d_timeline, sizeof(int32_t) * 16u * plan_units));
            const hipError_t me = hipMemsetAsync(d_timeline, 0, sizeof(int32_t) * 16u * plan_units, ctx->stream);
            if (me != hipSuccess) { (void)hipFree(d_timeline); RT_HIP(ctx, me); }
        }
        const uint32_t launched = plan_units;
        const uint32_t* lists = nullptr;
        if (spp >= lists_min_spp_of(ctx, P)) {              // a call of fewer samples would not use them: it does not make them either
            rc = ensure_lists(ctx, P, &lists);
            if (rc) { if (d_timeline) (void)hipFree(d_timeline); return rc; }
        }
        // the view's second call in a row: count its steps (see rt_ctx::h_stats_copy)
        const bool measure = lists != nullptr && !stats && !timeline_host && ctx->view_streak >= 1u && !ctx->lists_measured && spp >= lists_min_spp_of(ctx, P);
        if (measure) {
            if (!ctx->h_stats_copy) RT_HIP(ctx, hipHostMalloc(&ctx->h_stats_copy, sizeof(unsigned long long) * rtd::ST_COUNT));
            if (!ctx->stats_copied) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->stats_copied, hipEventDisableTiming));
            RT_HIP(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * rtd::ST_COUNT, ctx->stream));
        }
        const bool will_probe = !have_plan && !timeline_host && !ctx->env_no_probe && !ctx->env_no_tile_order && spp >= PROBE_MIN_SPP;
        rc = trace_frame(ctx, P, accum_ptr(ctx), (stats || measure) ? 1 : 0, replan ? ctx->d_tile_cost : nullptr, ctx->d_tile_order, have_plan,
                         !timeline_host && !ctx->env_no_probe && !ctx->env_no_tile_order, replan, &plan_units, d_timeline, lists);
        // resolved by the launch itself unless a probe's plan may have cut tiles (trace_frame then dropped the request)
        resolved_in_launch = P.resolve_out != nullptr && !(will_probe && slices_allowed(ctx, spp - (ctx->probe_spp ? ctx->probe_spp : PROBE_SPP)) > 1u);
        if (replan) { ctx->plan_view = view; ctx->plan_age = 0; } else ++ctx->plan_age;
        if (measure && rc == RT_OK) {
            ctx->lists_measured = true;
            if (hipMemcpyAsync(ctx->h_stats_copy, ctx->d_stats, sizeof(unsigned long long) * rtd::ST_COUNT, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                hipEventRecord(ctx->stats_copied, ctx->stream) == hipSuccess)
                ctx->measure_pending = true;
        }
        if (rc) { if (d_timeline) (void)hipFree(d_timeline); return rc; }
        hipError_t le = hipSuccess;
        if (d_timeline) {
            le = hipMemcpyAsync(timeline_host, d_timeline, sizeof(int32_t) * 16u * launched, hipMemcpyDeviceToHost, ctx->stream);
            if (le == hipSuccess) le = hipStreamSynchronize(ctx->stream);
            (void)hipFree(d_timeline);
        }
        RT_HIP(ctx, le);
        ctx->order_tiles_x = P.tiles_x; ctx->order_tiles_y = P.tiles_y;
        ctx->plan_units = plan_units;
        ctx->plan_max_slices = plan_units == n_tiles ? 1u : slices_allowed(ctx, spp);
    }
    ctx->accum_samples += spp;
    if (resolve && !resolved_in_launch) {
        if (ctx->accum_samples == 0) return fail(ctx, RT_ERR_STATE, "resolve: nothing has been rendered");
        RT_HIP(ctx, rtd::launch_resolve((int)ctx->cfg.semantics, ctx->stream, accum_ptr(ctx), resolve->rgba_device, ctx->width, ctx->height,
                                        ctx->accum_samples, resolve->gamma_sqrt, resolve->flip_rows));
    }
    if (stats) {
        unsigned long long h[rtd::ST_COUNT];
        RT_HIP(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memset(stats, 0, sizeof *stats);
        stats->samples = h[rtd::ST_SAMPLES]; stats->segments = h[rtd::ST_SEGMENTS];
        stats->diffuse = h[rtd::ST_DIFFUSE]; stats->specular = h[rtd::ST_SPECULAR];
        stats->terminal = h[rtd::ST_TERMINAL];
        // Every segment tests every primitive (the reference's closest-hit
        // loops are exhaustive), so the test counts follow from the segments.
        stats->shape_tests = stats->segments * ctx->shapes.size();
        stats->sphere_tests = stats->segments * ctx->n_spheres;
        stats->box_tests = stats->segments * ctx->n_boxes;
        stats->triangle_tests = stats->segments * ctx->tris.size();
        stats->triangle_tests_done = h[rtd::ST_TRI_TESTS];
        stats->bound_tests_done = h[rtd::ST_NODE_TESTS];
        stats->wave_iterations = h[rtd::ST_WAVE_ITERS];
        stats->work_units = h[rtd::ST_UNITS];
        stats->box_tests_done = h[rtd::ST_BOX_TESTS];
        stats->box_bound_tests_done = h[rtd::ST_BOX_BOUNDS];
        stats->sphere_tests_done = h[rtd::ST_SPHERE_TESTS];
        stats->primary_steps = h[rtd::ST_PRIMARY_STEPS];
        stats->tree_entries = h[rtd::ST_TREE_ENTRIES];
        stats->tree_entering_lanes = h[rtd::ST_TREE_ENTERING_LANES];
    }
    return RT_OK;
}

int rt_render_noalloc(rt_ctx* ctx, void* accum_device, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
                      uint64_t seed, uint32_t first_sample, uint32_t grid_n)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, accum_device, "rt_render_noalloc: accumulator is NULL");
    RT_REQUIRE(ctx, (reinterpret_cast<uintptr_t>(accum_device) & 15u) == 0, "rt_render_noalloc: accumulator must be 16-byte aligned");
    if (ctx->blob_dirty) return fail(ctx, RT_ERR_STATE, "rt_render_noalloc: the scene has changed since it was last built; call rt_commit_scene first");
    if (!ctx->d_scratch) return fail(ctx, RT_ERR_STATE, "rt_render_noalloc: no scratch on this device");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    rtd::TraceParams P;
    int rc = fill_params(ctx, P, width, height, spp, max_depth, seed, first_sample, grid_n);
    if (rc) return rc;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    if (n_tiles > RT_NOALLOC_MAX_TILES)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_noalloc: %u tiles of 16x16 pixels, the static scratch holds %u (use rt_render)",
                    n_tiles, (unsigned)RT_NOALLOC_MAX_TILES);
    if (spp == 0 || P.owned_rows == 0) return RT_OK;
    // The scratch keeps what the last call on this device left: the costs start at zero, the plan is written whole
    // before it is read.
    uint32_t* const cost = ctx->d_scratch;
    uint32_t* const order = ctx->d_scratch + RT_NOALLOC_MAX_TILES;
    RT_HIP(ctx, hipMemsetAsync(cost, 0, sizeof(uint32_t) * n_tiles, ctx->stream));
    uint32_t plan_units = n_tiles;
    // The candidate lists of the camera rays live in memory rt_render allocates: this call takes those an earlier
    // rt_render of the same view left, or traces without them (the same words, more slowly on scenes with boxes or a mesh).
    const uint32_t* lists = nullptr;
    if (ctx->lists_valid && !ctx->lists_useless && !ctx->env_no_tile_lists && P.rng_mode == 0) {
        rt_ctx::ListKey key{};
        memcpy(key.cam, P.cam, sizeof key.cam);
        key.width = P.width; key.height = P.height; key.sample_mode = P.sample_mode; key.grid_n = P.grid_n;
        key.shard_index = P.shard_index; key.shard_count = P.shard_count; key.band_rows = P.band_rows; key.owned_rows = P.owned_rows;
        key.tiles_x = P.tiles_x; key.tiles_y = P.tiles_y; key.scene_version = ctx->scene_version;
        if (memcmp(&key, &ctx->lists_key, sizeof key) == 0) lists = ctx->d_tile_lists;
    }
    return trace_frame(ctx, P, static_cast<uint32_t*>(accum_device), 0, cost, order, false, true, false, &plan_units, nullptr, lists);
}

int rt_render(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
              uint64_t seed, uint32_t first_sample, uint32_t grid_n)
{
    return render_impl(ctx, width, height, spp, max_depth, seed, first_sample, grid_n, nullptr);
}

int rt_render_resolved(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
                       uint64_t seed, uint32_t first_sample, uint32_t grid_n, float* rgba_device, int gamma_sqrt, int flip_rows)
{
    if (ctx && !rgba_device) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_resolved: output is NULL");
    const ResolveInto r = { rgba_device, gamma_sqrt, flip_rows };
    return render_impl(ctx, width, height, spp, max_depth, seed, first_sample, grid_n, nullptr, nullptr, 0, &r);
}

int rt_discard_accum(rt_ctx* ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->discard_pending = true;
    ctx->accum_samples = 0;
    return RT_OK;
}

int rt_render_stats(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
                    uint64_t seed, uint32_t first_sample, uint32_t grid_n, rt_stats* out)
{
    if (ctx && !out) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_stats: out is NULL");
    return render_impl(ctx, width, height, spp, max_depth, seed, first_sample, grid_n, out);
}

int rt_render_timeline(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp, uint32_t max_depth,
                       uint64_t seed, uint32_t first_sample, uint32_t grid_n, rt_stats* out,
                       int32_t* timeline_host, uint32_t timeline_waves)
{
    if (ctx && !timeline_host) return fail(ctx, RT_ERR_INVALID_ARG, "rt_render_timeline: output is NULL");
    return render_impl(ctx, width, height, spp, max_depth, seed, first_sample, grid_n, out, timeline_host, timeline_waves);
}

int rt_accum_samples(const rt_ctx* ctx, uint32_t* out)
{
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    *out = ctx->accum_samples;
    return RT_OK;
}

int rt_set_accum_samples(rt_ctx* ctx, uint32_t samples)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    ctx->accum_samples = samples;
    return RT_OK;
}

int rt_reset_accum(rt_ctx* ctx)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (accum_ptr(ctx))
        RT_HIP(ctx, hipMemsetAsync(accum_ptr(ctx), 0, (size_t)ctx->width * ctx->height * 16u, ctx->stream));
    ctx->accum_samples = 0;
    ctx->discard_pending = false;
    return RT_OK;
}

int rt_resolve_device(rt_ctx* ctx, float* rgba_device, int gamma_sqrt, int flip_rows)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, rgba_device, "rt_resolve_device: output is NULL");
    if (!accum_ptr(ctx) || ctx->accum_samples == 0) return fail(ctx, RT_ERR_STATE, "resolve: nothing has been rendered");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, rtd::launch_resolve((int)ctx->cfg.semantics, ctx->stream, accum_ptr(ctx), rgba_device, ctx->width, ctx->height,
                                    ctx->accum_samples, gamma_sqrt, flip_rows));
    return RT_OK;
}

// Device staging of at least `bytes` for the entry points that hand results to host memory; kept between calls.
static int ensure_stage(rt_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->stage_cap) return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_stage) { RT_HIP(ctx, hipFree(ctx->d_stage)); ctx->d_stage = nullptr; ctx->stage_cap = 0; }
    RT_HIP(ctx, hipMalloc(&ctx->d_stage, bytes));
    ctx->stage_cap = bytes;
    return RT_OK;
}

int rt_resolve(rt_ctx* ctx, float* rgba_host, int gamma_sqrt, int flip_rows)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, rgba_host, "rt_resolve: output is NULL");
    if (!accum_ptr(ctx) || ctx->accum_samples == 0) return fail(ctx, RT_ERR_STATE, "resolve: nothing has been rendered");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->width * ctx->height * 16u;
    int rc = ensure_stage(ctx, bytes);
    if (rc) return rc;
    float* const tmp = static_cast<float*>(ctx->d_stage);
    rc = rt_resolve_device(ctx, tmp, gamma_sqrt, flip_rows);
    if (rc == RT_OK) {
        hipError_t e = hipMemcpyAsync(rgba_host, tmp, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, "resolve copy-back failed: %s", hipGetErrorString(e));
    }
    return rc;
}

int rt_convert(rt_ctx* ctx, uint32_t n_samples)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, n_samples >= 1, "rt_convert: n_samples starts at 1 (main.cpp:42)");
    if (!accum_ptr(ctx) || ctx->accum_samples == 0) return fail(ctx, RT_ERR_STATE, "convert: nothing has been rendered");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_image || ctx->image_w != ctx->width || ctx->image_h != ctx->height) {
        RT_REQUIRE(ctx, n_samples == 1 || !ctx->d_image, "rt_convert: image size changed; restart with n_samples = 1");
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_image) { RT_HIP(ctx, hipFree(ctx->d_image)); ctx->d_image = nullptr; }
        const size_t bytes = (size_t)ctx->width * ctx->height * 16u;
        RT_HIP(ctx, hipMalloc(&ctx->d_image, bytes));
        RT_HIP(ctx, hipMemsetAsync(ctx->d_image, 0, bytes, ctx->stream));
        ctx->image_w = ctx->width; ctx->image_h = ctx->height;
    }
    RT_HIP(ctx, rtd::launch_convert((int)ctx->cfg.semantics, ctx->stream, accum_ptr(ctx), ctx->d_image, ctx->width, ctx->height,
                                    ctx->accum_samples, n_samples));
    ctx->accum_samples = 0;
    return RT_OK;
}

int rt_read_image(rt_ctx* ctx, float* rgba_host)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, rgba_host, "rt_read_image: output is NULL");
    if (!ctx->d_image) return fail(ctx, RT_ERR_STATE, "rt_read_image: rt_convert has not run");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipMemcpyAsync(rgba_host, ctx->d_image, (size_t)ctx->image_w * ctx->image_h * 16u, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace_debug(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t max_depth, uint64_t seed,
                   uint32_t sample, uint32_t grid_n, int32_t* hit_index_host, uint32_t* material_id_host)
{
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_REQUIRE(ctx, hit_index_host && material_id_host, "rt_trace_debug: output is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = RT_OK;
    if (ctx->blob_dirty) { rc = build_blob(ctx); if (rc) return rc; }
    rtd::TraceParams P;
    // The trace ignores sharding: it follows every pixel.
    const uint32_t si = ctx->shard_index, sc = ctx->shard_count;
    ctx->shard_index = 0; ctx->shard_count = 1;
    rc = fill_params(ctx, P, width, height, 1, max_depth, seed, sample, grid_n ? grid_n : (ctx->cfg.sample_mode ? 0u : 1u));
    ctx->shard_index = si; ctx->shard_count = sc;
    if (rc) return rc;
    P.trace_sample = sample;
    const size_t n = (size_t)width * height * max_depth;
    rc = ensure_stage(ctx, n * (sizeof(int32_t) + sizeof(uint32_t)));
    if (rc) return rc;
    int32_t* const d_hit = static_cast<int32_t*>(ctx->d_stage);
    uint32_t* const d_mat = reinterpret_cast<uint32_t*>(d_hit + n);
    RT_HIP(ctx, rtd::launch_trace_fill(ctx->stream, d_hit, d_mat, n));
    // accum is not touched by a TRACE launch; pass any valid pointer.
    RT_HIP(ctx, rtd::launch_trace((int)ctx->cfg.semantics, 2, ctx->stream, P, ctx->d_blob, reinterpret_cast<uint32_t*>(d_hit), ctx->d_stats,
                                  d_hit, d_mat, nullptr, nullptr, P.tiles_x * P.tiles_y, nullptr));
    RT_HIP(ctx, hipMemcpyAsync(hit_index_host, d_hit, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(material_id_host, d_mat, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_device_info(int device, char* name, size_t name_len, int* compute_units, int* clock_mhz)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RT_ERR_NO_DEVICE;
    if (name && name_len) { snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    return RT_OK;
}

}  // extern "C"
