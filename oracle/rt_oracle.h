/*
 * rt_oracle.h — CPU restatement of the reference path tracer (TEST ORACLE).
 *
 * This is test infrastructure, not product code: only tests/, the smoke check
 * and the cpu_baseline leg of bench.py may build, load or call it.  The
 * renderer itself never does.
 *
 * PARITY UNPINNED BY THE REFERENCE: jaafer06/ray_tracing ships no tests, no
 * golden vectors and no seedable RNG (src/utils.h:8-9 seeds mt19937 from
 * random_device; src/main.cpp:139-141 seeds the shader from the wall clock),
 * and neither of its tracers can be built here (Eigen / GL are not vendored).
 * The oracle is therefore pinned by (i) closed-form known-answer tests of each
 * routine, (ii) the reference's own constants (camera maths, struct layouts,
 * its two .off meshes) and (iii) statistical agreement with a literal
 * Box-Muller restatement of utils.h:37-46.
 *
 * Every routine in rt_oracle.c cites the reference lines it follows.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Byte layouts of the reference's structs (src/gpu_version/shape.h:5-12,
 * 44-54, 81-109), declared here independently of the renderer's header. */
typedef struct { float color[3]; uint32_t type; float data[4]; } orc_material;          /*  32 B */
typedef struct { float m[16]; uint32_t type; float data[11]; orc_material mat; } orc_shape; /* 144 B */
typedef struct { float position[4]; float a[3]; float beta; float m1[4]; float m2[4]; } orc_triangle; /* 64 B */

enum { ORC_SEM_CPU = 0, ORC_SEM_GLSL = 1 };
enum { ORC_SAMPLE_JITTER = 0, ORC_SAMPLE_GRID = 1, ORC_SAMPLE_GRID_COMPAT = 2 };
/* BOX_SIDES: a Box has the three sides Box::Box stores (shape.h:67) instead of the one the shader reads. */
/* UNIT_GAUSS: randomUnitVector as written (src/utils.h:45); the same as unit_sampler = ORC_UNIT_GAUSS. */
enum { ORC_FLAG_METAL_UNNORMALIZED = 1u, ORC_FLAG_BOX_SIDES = 2u, ORC_FLAG_UNIT_GAUSS = 4u };
/* Which stream feeds the draws: the counter hash (one value per (seed, pixel, sample, draw
 * index)), or the shader's own random() (RayTracing.glsl:8-9,44-51), consumed in program order. */
enum { ORC_RNG_HASH = 0, ORC_RNG_GLSL = 1 };
/* Order in which the colour of a path is multiplied together (RT_SEM_CPU):
 * AS_WRITTEN = the recursion of rayTracer.h:121, innermost product first;
 * FORWARD    = running throughput from the camera outwards (the order an
 *              iterative tracer such as RayTracing.glsl:256-288 uses). */
enum { ORC_COLOR_AS_WRITTEN = 0, ORC_COLOR_FORWARD = 1 };
/* How randomUnitVector() (utils.h:37-46) is sampled:
 * DIRECT = uniform on the sphere from two draws (z, phi), libm-free;
 * GAUSS  = three Box-Muller normal deviates, normalised — utils.h:45 taken
 *          literally (uses libm; only for the statistical cross-check). */
enum { ORC_UNIT_DIRECT = 0, ORC_UNIT_GAUSS = 1 };

/* Which arithmetic evaluates the reference's expressions (see rt_oracle.c):
 * RECIPE  = the fixed recipes the HIP kernels repeat bit for bit (explicit FMAs, Newton reciprocal and reciprocal
 *           root, polynomial sin/cos);
 * LITERAL = the source text as a compiler without contraction evaluates it: no FMA, IEEE divide, v / sqrt(v.v),
 *           libm sinf/cosf, Sphere::hit's distance in double.  Used only to state how far RECIPE is from the text.
 * LITERAL_GEOMETRY = LITERAL for the geometry (intersections, normals, normalisations, reflections, hit distances in
 *           double), RECIPE for the samplers (polynomial sin/cos, the fused sphere point): what the HIP kernels compute
 *           under RT_FLAG_LITERAL_GEOMETRY, bit for bit.  Its hit indices equal LITERAL's up to the first scatter
 *           (first hits: always), since only the sampled directions differ. */
enum { ORC_ARITH_RECIPE = 0, ORC_ARITH_LITERAL = 1, ORC_ARITH_LITERAL_GEOMETRY = 2 };

typedef struct {
    uint32_t width, height;
    uint32_t spp, first_sample, max_depth, grid_n;
    uint64_t seed;
    uint32_t semantics, sample_mode, flags;
    float    triangle_albedo[3];
    uint32_t color_order;
    uint32_t unit_sampler;
    uint32_t shard_index, shard_count, band_rows;   /* rows with (y/band)%count==index */
    int32_t  threads;                               /* <=0: all cores                  */
    uint32_t rng;                                   /* ORC_RNG_HASH / ORC_RNG_GLSL     */
    const float* triangle_colors;                   /* NULL, or rgb per triangle (replaces triangle_albedo) */
    uint32_t arith;                                 /* ORC_ARITH_RECIPE / ORC_ARITH_LITERAL / ORC_ARITH_LITERAL_GEOMETRY */
} orc_params;

typedef struct {
    uint64_t samples, segments, diffuse, specular, terminal;
    uint64_t shape_tests, sphere_tests, box_tests, triangle_tests;
} orc_stats;

/* Adds samples [first_sample, first_sample+spp) of every (sharded) pixel into
 * accum (width*height*4 32-bit words: float sums for SEM_CPU, uint sums of
 * the 0-255 quantised samples for SEM_GLSL; word 3 unused).  Rows are in
 * camera order (row 0 = top); the flip happens in orc_resolve. */
int orc_render(const orc_shape* shapes, uint32_t n_shapes,
               const orc_triangle* tris, uint32_t n_tris,
               const float camera[21], const orc_params* p,
               uint32_t* accum, orc_stats* stats);

/* Per pixel and bounce of one sample: primitive index hit (shape index or
 * n_shapes + triangle index; -1 miss; -2 path already over) and material type
 * (0xFFFFFFFF = none). */
int orc_trace(const orc_shape* shapes, uint32_t n_shapes,
              const orc_triangle* tris, uint32_t n_tris,
              const float camera[21], const orc_params* p, uint32_t sample,
              int32_t* hit_index, uint32_t* material_id);

void orc_resolve(const uint32_t* accum, uint32_t width, uint32_t height,
                 uint32_t samples, uint32_t semantics, int gamma_sqrt,
                 int flip_rows, float* rgba);
void orc_convert(uint32_t* accum, float* image_rgba, uint32_t width, uint32_t height,
                 uint32_t samples_in_accum, uint32_t semantics, uint32_t n_samples);

/* Pieces exposed for known-answer tests.  They use the calling thread's arithmetic, ORC_ARITH_RECIPE unless
 * orc_set_arith(ORC_ARITH_LITERAL) was called on it. */
void     orc_set_arith(int arith);
uint32_t orc_hash32(uint32_t x);
float    orc_log_draw(float u);
uint32_t orc_path_key(uint64_t seed, uint32_t pixel_index, uint32_t sample);
float    orc_draw(uint32_t key, uint32_t index);
float    orc_glsl_random(float seed[3]);
void     orc_glsl_seed(uint64_t seed, uint32_t width, uint32_t height, uint32_t gx, uint32_t gy, uint32_t sample, float out[3]);
void     orc_sincos2pi(float u, float* c, float* s);
void     orc_unit_vector(float u1, float u2, float out[3]);
void     orc_hemisphere_glsl(const float n[3], float u1, float u2, float out[3]);
/* The dielectric's new direction (not normalised): mirrored when u1 < Schlick's reflectance, else refracted. */
void     orc_glass_direction(const float n[3], const float d[3], float ior, float u1, float out[3]);
float    orc_hit_sphere(const float center[3], float radius, const float o[3], const float d[3]);
float    orc_hit_box(const float center[3], float scale, const float o[3], const float d[3]);
float    orc_hit_box3(const float center[3], const float sides[3], const float o[3], const float d[3]);
void     orc_box_normal3(const float center[3], const float sides[3], const float p[3], float n[3]);
float    orc_hit_triangle(const orc_triangle* t, const float o[3], const float d[3]);
void     orc_box_normal(const float center[3], float scale, const float p[3], float n[3]);
int      orc_hit_cube(const float m[16], const float o[3], const float d[3], float* t, float in_point[3]);
void     orc_cube_normal(const float m[16], const float in_point[3], float n[3]);
void     orc_camera_ray(const float camera[21], const orc_params* p, uint32_t gx, uint32_t gy,
                        uint32_t sample, float origin[3], float dir[3]);

/* Host-side constructors restated (camera.h:14-28, shape.h:83-102,
 * SimpleMesh.h:60-85,122-126,145-215). */
void orc_camera_init(float data[21], uint32_t width, uint32_t height, float focal, const float pos[3]);
void orc_make_triangle(orc_triangle* out, const float p1[3], const float p2[3], const float p3[3]);
/* Loads an OFF/COFF file, applies normalize -> resize -> transform(0.5*Rx(-pi/2),
 * (0.5,0.7,-2)) (src/main.cpp:109-112) when `main_cpp_pipeline` is set, and
 * writes up to max_tris triangles; returns the triangle count or <0. */
int  orc_mesh_triangles(const char* off_path, int main_cpp_pipeline, orc_triangle* out, uint32_t max_tris,
                        uint32_t* n_vertices);

#ifdef __cplusplus
}
#endif
#endif
