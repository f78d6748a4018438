/*
 * rt_mi355x.h — C ABI of the MI355X (gfx950) path-tracing core.
 *
 * This library replaces the OpenGL dispatch of the reference renderer
 * (jaafer06/ray_tracing): the class RayTracingComputeShader
 * (src/gpu_version/ray_tracing_compute_shader.h:9-67), the two compute
 * dispatches of the frame loop (src/main.cpp:139-151) and the two shaders
 * behind them (src/shaders/RayTracing.glsl, src/shaders/PixelConversion.glsl).
 *
 * Scene data crosses the boundary as raw bytes in the reference's own struct
 * layouts, so a host that keeps using the reference's shape.h / camera.h
 * structs can hand its std::vector<Shape>::data() straight in:
 *
 *   rt_material  32 B  = Material        (src/gpu_version/shape.h:5-12)
 *   rt_shape    144 B  = Shape           (src/gpu_version/shape.h:44-54,
 *                                         std430 twin RayTracing.glsl:86-97)
 *   rt_triangle  64 B  = SimpleTriangle  (src/gpu_version/shape.h:81-109,
 *                                         std430 twin RayTracing.glsl:10-16)
 *   camera      21 f32 = Camera::data    (src/gpu_version/camera.h:115-117,
 *                                         std430 twin RayTracing.glsl:28-36)
 *
 * Conventions: every entry point returns RT_OK (0) or a negative rt_status;
 * nothing throws across the boundary; rt_last_error() gives a message for
 * the last failure on that context.  A context is bound to one HIP device and
 * must be driven by one host thread at a time.  All device work is queued on
 * the context's stream (rt_set_stream); entry points that return data to host
 * memory synchronise that stream before returning, the others do not.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ types */

/* Material: src/gpu_version/shape.h:5-12.  type: 0 Light, 1 Lambertian,
 * 2 Metal, 100 Debug (shape.h:14-42); 3 Dielectric is an extension with no
 * counterpart in the reference (mirror with Schlick's probability, else refract at
 * the surface; the sphere test has only the near root, so there is no exit event).  data[0] is what the reference's Metal ctor
 * stores ("refraction", never read by its shader; the Dielectric extension
 * reads it as the index of refraction); data[1] is the metal fuzz radius of
 * the CPU tracer (src/cpu_version/scene.h:71,78), 0 = perfect mirror. */
typedef struct rt_material {
    float    color[3];
    uint32_t type;
    float    data[4];
} rt_material;

/* Shape: src/gpu_version/shape.h:44-54.  transformation is column-major;
 * column 3 holds the translation, element [0][0] the radius (Circle,
 * shape.h:56-62) or the cube side (Box, shape.h:64-70 as read by
 * RayTracing.glsl:139).  type: 0 Circle, 1 Box. */
typedef struct rt_shape {
    float       transformation[16];
    uint32_t    type;
    float       data[11];
    rt_material material;
} rt_shape;

/* SimpleTriangle: src/gpu_version/shape.h:81-109. */
typedef struct rt_triangle {
    float position[4];   /* first vertex, w unused               */
    float a[3];          /* unit plane normal                    */
    float beta;          /* a . position                         */
    float m1[4];         /* dual basis of (v1, v2), w unused     */
    float m2[4];
} rt_triangle;

#define RT_CAMERA_FLOATS 21u   /* src/gpu_version/camera.h:115 */

enum rt_material_type { RT_MAT_LIGHT = 0, RT_MAT_LAMBERTIAN = 1, RT_MAT_METAL = 2,
                        RT_MAT_DIELECTRIC = 3, RT_MAT_DEBUG = 100 };
/* 0 and 1 are the shader's shapes (shape.h:56-70).  RT_SHAPE_CUBE is the CPU tracer's Cube
 * (src/cpu_version/scene.h:142-217): a unit cube, its rotation matrix in the upper-left 3x3 of
 * `transformation`, its centre in column 3; hit only under RT_SEM_CPU (the shader's hit() returns
 * -1 for a type it does not know, RayTracing.glsl:176).  The shader's own type 2 (Triangle,
 * shape.h:72-79) is never hit there either. */
enum rt_shape_type    { RT_SHAPE_CIRCLE = 0, RT_SHAPE_BOX = 1, RT_SHAPE_CUBE = 3 };

/* Which of the reference's two tracers the context reproduces. */
enum rt_semantics {
    RT_SEM_CPU  = 0,  /* src/cpu_version: rayTracer.h:110-132, scene.h:39-255, utils.h:37-64 */
    RT_SEM_GLSL = 1   /* src/shaders/RayTracing.glsl:206-308 + PixelConversion.glsl          */
};

/* Where inside its pixel a sample's camera ray starts. */
enum rt_sample_mode {
    RT_SAMPLE_JITTER      = 0, /* two uniform draws per sample (RayTracing.glsl:296)            */
    RT_SAMPLE_GRID        = 1, /* regular N x N grid, spp = N*N, step worldStep/(N+1)           */
    RT_SAMPLE_GRID_COMPAT = 2  /* rayTracer.h:92-101 as written: the x offset accumulates       */
};

enum rt_flags {
    /* scene.h:78 leaves the fuzzed metal direction un-normalised although
     * Sphere::hit assumes a unit direction; by default it is normalised.
     * With this flag it is left as the reference leaves it. */
    RT_FLAG_METAL_UNNORMALIZED = 1u << 0,
    /* Box::Box stores three sides on the diagonal of `transformation` (shape.h:67), but the
     * shader reads only transformation[0][0] for all axes (RayTracing.glsl:139,159), so every
     * Box is a cube there — the default here too.  With this flag a Box has its three sides. */
    RT_FLAG_BOX_SIDES          = 1u << 1,
    /* randomUnitVector() as the CPU tracer writes it (src/utils.h:45): three normal deviates, normalised
     * (Box-Muller on four draws per event), instead of the direct sphere point that has the same
     * distribution and is the default.  RT_SEM_CPU with RT_RNG_HASH only. */
    RT_FLAG_UNIT_GAUSS         = 1u << 2,
    /* The CPU tracer's geometry in the arithmetic its source text has when a C++ compiler evaluates it without
     * contraction: Sphere::hit, normalAt, the Ray constructor, Ray::at and reflect (src/cpu_version/scene.h:14-23,
     * 115-132; src/utils.h:62-64) with every product and sum rounded on its own, normalized() as v / sqrt(v.v)
     * with the IEEE root and quotient, and the hit distance computed, compared and windowed as the double that
     * HitRecord::t is (scene.h:34,123,237).  By default those lines are evaluated with fused multiply-adds and
     * Newton reciprocal roots, which moves a hit on a silhouette now and then.  The samplers keep their
     * polynomial either way.  RT_SEM_CPU with RT_RNG_HASH, spheres and Cubes only (a scene with Boxes or
     * triangles is refused), not together with RT_FLAG_UNIT_GAUSS. */
    RT_FLAG_LITERAL_GEOMETRY   = 1u << 3
};

/* Which stream feeds the random draws. */
enum rt_rng {
    /* Counter based: draw i of the path (seed, pixel, sample) is a hash of those four numbers.
     * Stands in for both reference generators, neither of which can be replayed (src/utils.h:8-9:
     * an unseeded mt19937 shared by racing threads; RayTracing.glsl:8-9: wall-clock time). */
    RT_RNG_HASH = 0,
    /* The shader's own random() (RayTracing.glsl:44-51) with its seed of RayTracing.glsl:9:
     * sample s of a pixel is invocation (s % 10, s / 10 % 10) of the pixel's 10 x 10 work group
     * in frame s / 100, `time` = (seed + frame) / 10000 (main.cpp:139-141 passes milliseconds /
     * 10000), and draws are consumed in the order the shader's program makes them. */
    RT_RNG_GLSL = 1
};

enum rt_status {
    RT_OK                 =  0,
    RT_ERR_INVALID_ARG    = -1,
    RT_ERR_NO_DEVICE      = -2,
    RT_ERR_HIP            = -3,
    RT_ERR_OUT_OF_MEMORY  = -4,
    RT_ERR_STATE          = -5,
    RT_ERR_IO             = -6
};

typedef struct rt_config {
    uint32_t struct_size;      /* = sizeof(rt_config)                            */
    int32_t  device;           /* HIP device ordinal                             */
    uint32_t semantics;        /* enum rt_semantics                              */
    uint32_t sample_mode;      /* enum rt_sample_mode                            */
    uint32_t flags;            /* enum rt_flags                                  */
    float    triangle_albedo[3]; /* RayTracing.glsl:275 hard-codes (1,0,0)       */
    uint32_t rng;              /* enum rt_rng                                    */
} rt_config;

/* Work counters of the last rt_render call that asked for them. */
typedef struct rt_stats {
    uint64_t samples;          /* camera paths started                           */
    uint64_t segments;         /* ray segments traced (closest-hit queries)      */
    uint64_t diffuse;          /* Lambertian / triangle scatter events           */
    uint64_t specular;         /* metal + dielectric scatter events              */
    uint64_t terminal;         /* paths ended by light, miss, absorb or depth    */
    uint64_t shape_tests;      /* ray-sphere + ray-box tests                     */
    uint64_t sphere_tests;
    uint64_t box_tests;        /* ... of these, ray-box: what the reference's scan does, segments x boxes */
    uint64_t triangle_tests;   /* what the reference's exhaustive scan does: segments x triangles */
    uint64_t triangle_tests_done;  /* ray-triangle tests this library executed (it culls by bounding spheres) */
    uint64_t bound_tests_done;     /* ray-bounding-sphere tests it executed to do so */
    uint64_t wave_iterations;      /* loop iterations summed over waves: segments / (64 * this) = lane use */
    uint64_t work_units;           /* workgroups that traced: the tiles, plus one per extra slice of a tile's sample range */
    uint64_t box_tests_done;       /* ray-box tests this library executed (it culls by the boxes' bounds) */
    uint64_t box_bound_tests_done; /* slab tests of those bounds it executed to do so */
    uint64_t sphere_tests_done;    /* ray-sphere tests this library executed (camera rays test only the spheres their tile can see) */
    uint64_t primary_steps;        /* of wave_iterations: steps that traced camera rays only, from their tile's candidate list */
    uint64_t tree_entries;         /* times a wave entered a top node of the mesh's tree (some lane's ray touched its bound) */
    uint64_t tree_entering_lanes;  /* ... and the lanes whose rays did: tree_entering_lanes / tree_entries lanes share a walk */
} rt_stats;

typedef struct rt_ctx rt_ctx;

/* ------------------------------------------------------- context lifetime */

/* Replaces RayTracingComputeShader::RayTracingComputeShader
 * (ray_tracing_compute_shader.h:12-32: creates the camera, shape and
 * triangle buffers) and LoadShaders() (src/gpu_version/loadShader.h:61-103;
 * the HIP kernels are linked into this library). */
int  rt_create(rt_ctx** out, const rt_config* cfg);
void rt_destroy(rt_ctx* ctx);

/* Text of the last error on ctx (or of the last failed rt_create if ctx is
 * NULL).  The reference has no error convention (loadShader.h:27-31 prints
 * and continues); this replaces it. */
const char* rt_last_error(const rt_ctx* ctx);

/* Queue all later work of ctx on this hipStream_t (NULL = the default
 * stream).  No counterpart in the reference (one GL context, one queue). */
int rt_set_stream(rt_ctx* ctx, void* hip_stream);

/* ------------------------------------------------------------ scene upload */

/* Replace RayTracingComputeShader::updateShapeBuffer
 * (ray_tracing_compute_shader.h:44-57): glBufferData of shapes.data() with
 * sizeof(Shape) stride + glUniform1ui(shapeCount), and the same for the
 * SimpleTriangle vector + triangleCount.  The caller keeps its arrays; the
 * library copies and repacks them before returning.  n = 0 clears. */
int rt_upload_shapes(rt_ctx* ctx, const rt_shape* shapes, uint32_t n);
int rt_upload_triangles(rt_ctx* ctx, const rt_triangle* triangles, uint32_t n);

/* One albedo per uploaded triangle (n x rgb), in place of the single colour the shader
 * hard-codes (RayTracing.glsl:275; rt_config.triangle_albedo) — e.g. the vertex colours a
 * COFF file carries (SimpleMesh.h:170-183), which the reference parses and then ignores;
 * see rt_mesh_triangle_colors.  n must equal the number of uploaded triangles; n = 0 returns
 * to the single colour, and so does every rt_upload_triangles. */
int rt_upload_triangle_colors(rt_ctx* ctx, const float* rgb, uint32_t n);

/* The uploads above only record the arrays (and refuse a scene that cannot be laid out); the
 * device copy — with the tree over the triangles, the expensive part — is built once, by the
 * first render call after them, or here.  updateShapeBuffer() of the reference uploads shapes and
 * triangles one after the other (ray_tracing_compute_shader.h:44-57): one build for both. */
int rt_commit_scene(rt_ctx* ctx);

/* Replaces RayTracingComputeShader::updateCameraBuffer
 * (ray_tracing_compute_shader.h:34-37): 21 floats, Camera::data. */
int rt_set_camera(rt_ctx* ctx, const float data[RT_CAMERA_FLOATS]);

/* ----------------------------------------------------------------- render */

/* Restrict rendering to the image rows y with (y / band_rows) % shard_count
 * == shard_index; the other rows of the accumulator are left untouched.
 * (0, 1, any) renders every row.  No counterpart in the reference (single
 * GPU); it is how one frame is split over the GPUs of a node. */
int rt_set_row_shard(rt_ctx* ctx, uint32_t shard_index, uint32_t shard_count,
                     uint32_t band_rows);

/* Use caller-owned device memory (width*height*4 32-bit words, 16-byte
 * aligned) as the accumulator instead of the context's own, so that a
 * collective can run on it directly.  NULL goes back to the internal one.
 * The buffer is used as it is: zero it (or call rt_reset_accum) to start. */
int rt_bind_accum(rt_ctx* ctx, void* device_words, uint32_t width, uint32_t height);

/* Replaces glUniform1f(time) + RayTracingComputeShader::compute(width,
 * height, 1) (src/main.cpp:139-145; ray_tracing_compute_shader.h:39-42) and,
 * for RT_SEM_CPU, Camera::render() of the CPU tracer
 * (src/cpu_version/rayTracer.h:71-87).  Traces samples [first_sample,
 * first_sample + spp) of every pixel and adds them to the accumulator.
 * max_depth = most ray segments per path (RayTracing.glsl:263,300 uses 10;
 * rayTracer.h:78,113 traces steps + 1 = 11).  seed replaces the wall-clock
 * `time` uniform.  For the grid sample modes grid_n * grid_n samples make one
 * pixel; pass 0 to derive grid_n from first_sample + spp (must be a square).
 * Asynchronous on the context's stream.
 * Workgroups are handed tiles by a plan made from what the tiles cost in the previous call of the
 * same frame size; a call of 64 or more samples per pixel that has no plan yet makes one from its
 * own first 8 samples.  Any plan gives the same words. */
int rt_render(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp,
              uint32_t max_depth, uint64_t seed, uint32_t first_sample,
              uint32_t grid_n);

/* rt_render followed by rt_resolve_device, as one call: the same words in the accumulator and in `rgba_device`
 * (width*height RGBA float32 on the context's device).  Where every pixel's sum is formed by one lane of one launch —
 * always under RT_SEM_CPU, and under RT_SEM_GLSL while the launch plan cuts no tile's sample range — the launch that
 * ends the frame writes the resolved pixel from the sums it holds in registers, and no resolve kernel runs: the
 * converter dispatch of src/main.cpp:147-149 folded into the compute dispatch of :145.  Asynchronous. */
int rt_render_resolved(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp,
                       uint32_t max_depth, uint64_t seed, uint32_t first_sample,
                       uint32_t grid_n, float* rgba_device, int gamma_sqrt, int flip_rows);

/* rt_render for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid it).  No counterpart in the reference.  It adds the same
 * samples to `accum_device` (caller-owned, width*height*4 32-bit words, 16-byte aligned, on the
 * context's device) as rt_render adds to its accumulator — bit for bit — and calls no hipMalloc /
 * hipFree: the per-tile costs and the launch plan that rt_render keeps in memory of its own live
 * in a static device array of the library instead (768 KiB, held by every GPU the library has been
 * used on until the process ends).  Limits: at most RT_NOALLOC_MAX_TILES tiles of 16x16 pixels
 * (more: RT_ERR_INVALID_ARG); the scene must have been built (rt_commit_scene or an earlier
 * render; else RT_ERR_STATE).  All calls on one GPU share the one array: calls that overlap there
 * (two streams, two contexts, two threads) race on it and may trace tiles twice or not at all.
 * The context's own accumulator, sample count and plan are left alone.  Asynchronous on the
 * context's stream. */
#define RT_NOALLOC_MAX_TILES 65536u
int rt_render_noalloc(rt_ctx* ctx, void* accum_device, uint32_t width, uint32_t height,
                      uint32_t spp, uint32_t max_depth, uint64_t seed, uint32_t first_sample,
                      uint32_t grid_n);

/* The same render, and also fills `out` with the work counters (synchronises
 * the stream).  The image produced is bit-identical to rt_render's. */
int rt_render_stats(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp,
                    uint32_t max_depth, uint64_t seed, uint32_t first_sample,
                    uint32_t grid_n, rt_stats* out);

/* Diagnostic: a render that also records, per wave of the launch (four per workgroup
 * of the launch plan, in launch order), {start, end} of its loop in 100 MHz ticks, its
 * shader clocks / 16 with the XCC id in the top four bits, and the loop's length.
 * timeline_host: 4 int32 per wave, timeline_waves = capacity in waves (8 per 16x16
 * tile always suffice).  out = NULL: the plain render kernel, whose timing the work
 * counters do not disturb; else also the counters of rt_render_stats.  No counterpart
 * in the reference. */
int rt_render_timeline(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t spp,
                       uint32_t max_depth, uint64_t seed, uint32_t first_sample,
                       uint32_t grid_n, rt_stats* out, int32_t* timeline_host,
                       uint32_t timeline_waves);

/* Number of samples per pixel added to the accumulator since the last reset. */
int rt_accum_samples(const rt_ctx* ctx, uint32_t* out);

/* Declare that the accumulator now holds `samples` samples per pixel: for a
 * caller that has summed other GPUs' accumulators into this one.  No
 * counterpart in the reference. */
int rt_set_accum_samples(rt_ctx* ctx, uint32_t samples);

/* main.cpp:179,194 (number_of_samples = 1): drop everything accumulated. */
int rt_reset_accum(rt_ctx* ctx);

/* The same for a caller that renders next: nothing is cleared now; the next render call starts its sums from zero
 * and stores them instead of reading the accumulator and adding (where its launch plan cuts a tile's sample range,
 * whose slices meet in the accumulator through atomics, it clears first after all).  The frame that call leaves is
 * the one rt_reset_accum + that call leave — in the rows the context renders: with a row shard set
 * (rt_set_row_shard) the other rows keep what they held.  No counterpart in the reference, whose converter clears
 * the pixel buffer it has just read (PixelConversion.glsl:25-27). */
int rt_discard_accum(rt_ctx* ctx);

/* ---------------------------------------------------------------- resolve */

/* Replaces the converter dispatch (src/main.cpp:147-149,
 * PixelConversion.glsl:12-20) for one frame, and rayTracer.h:105 for
 * RT_SEM_CPU: out = accumulated / samples as width*height RGBA float32
 * (alpha 1).  gamma_sqrt applies rayTracer.h:105's cwiseSqrt; flip_rows
 * stores image row y at row height-1-y, the orientation both reference
 * tracers write (rayTracer.h:78, RayTracing.glsl:303).  Synchronises. */
int rt_resolve(rt_ctx* ctx, float* rgba_host, int gamma_sqrt, int flip_rows);

/* The same into device memory, asynchronous on the context's stream. */
int rt_resolve_device(rt_ctx* ctx, float* rgba_device, int gamma_sqrt, int flip_rows);

/* The progressive form of the converter (PixelConversion.glsl:12-28 with
 * src/main.cpp:147-151): folds the accumulator into the context's running
 * mean image with weight 1/n_samples, then zeroes the accumulator.
 * n_samples = 1 restarts the mean (main.cpp:179,194). */
int rt_convert(rt_ctx* ctx, uint32_t n_samples);

/* Copy the running-mean image (the RGBA32F texture of main.cpp:72-79) to
 * host memory, in the reference's row order.  Synchronises. */
int rt_read_image(rt_ctx* ctx, float* rgba_host);

/* ------------------------------------------------------------------ debug */

/* For exactness checks (the reference's only aid of this kind is
 * DebugMaterial, shape.h:37-42): re-traces sample `sample` of every pixel
 * without touching the accumulator and writes, per pixel and bounce b <
 * max_depth, the index of the primitive hit (shape index, or n_shapes +
 * triangle index; -1 = miss, -2 = path had already ended) and its material
 * type (0xFFFFFFFF where there is none).  Arrays are host memory of
 * width*height*max_depth entries, pixel-major, rows unflipped. */
int rt_trace_debug(rt_ctx* ctx, uint32_t width, uint32_t height, uint32_t max_depth,
                   uint64_t seed, uint32_t sample, uint32_t grid_n,
                   int32_t* hit_index_host, uint32_t* material_id_host);

/* ------------------------------------------- host-side scene construction */
/* Plain-C forms of the reference's host constructors, so that a host without
 * the reference's Eigen-based headers builds byte-identical structs. */

/* Camera::Camera (src/gpu_version/camera.h:14-28); position (1,1,6) there,
 * (1,0,6) in the CPU tracer (rayTracer.h:37). */
void rt_camera_init(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height,
                    float focal_length, const float position[3]);
/* Camera::rotate / Camera::move (camera.h:85-113).  direction: 0 forward,
 * 1 backward, 2 right, 3 left. */
void rt_camera_rotate(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height,
                      float focal_length, float horizontal, float vertical);
void rt_camera_move(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height,
                    float focal_length, int direction);

/* Circle / Box constructors (shape.h:56-70). */
void rt_make_circle(rt_shape* out, const float translation[3], float radius,
                    const rt_material* material);
void rt_make_box(rt_shape* out, const float translation[3], const float sides[3],
                 const rt_material* material);
/* Cube::Cube and Cube::rotate of the CPU tracer (src/cpu_version/scene.h:144-148, :206-208):
 * rotation *= Rx(angle*pi) * Ry(angle*pi). */
void rt_make_cube(rt_shape* out, const float center[3], const rt_material* material);
void rt_cube_rotate(rt_shape* cube, float angle);
/* SimpleTriangle::SimpleTriangle (shape.h:83-102). */
void rt_make_triangle(rt_triangle* out, const float p1[3], const float p2[3],
                      const float p3[3]);

/* SimpleMesh (src/SimpleMesh.h): loadMesh :145-215, normalize :60-68,
 * resize :79-85, transform :122-126, and the mesh -> SimpleTriangle loop of
 * src/main.cpp:113-122. */
typedef struct rt_mesh rt_mesh;
int      rt_mesh_load(rt_mesh** out, const char* off_path);
void     rt_mesh_free(rt_mesh* mesh);
uint32_t rt_mesh_vertex_count(const rt_mesh* mesh);
uint32_t rt_mesh_triangle_count(const rt_mesh* mesh);
void     rt_mesh_normalize(rt_mesh* mesh);
void     rt_mesh_resize(rt_mesh* mesh);
void     rt_mesh_transform(rt_mesh* mesh, const float rotation_rowmajor[9],
                           const float translation[3]);
/* positions: n*4 floats (xyzw); colors: n*4 floats; indices: n*3 uint32. */
void     rt_mesh_copy_vertices(const rt_mesh* mesh, float* positions_xyzw, float* colors_rgba);
void     rt_mesh_copy_indices(const rt_mesh* mesh, uint32_t* indices);
/* Writes rt_mesh_triangle_count() SimpleTriangles.  RT_ERR_INVALID_ARG (nothing written) when a face names a vertex
 * the mesh does not have: rt_mesh_add_face takes any index, as SimpleMesh::addFace does, so a host may add faces
 * before their vertices — the indices are checked here, where they are followed. */
int      rt_mesh_to_triangles(const rt_mesh* mesh, rt_triangle* out);
/* Writes rt_mesh_triangle_count() x rgb: the mean of each face's three vertex colours
 * (Vertex::color, SimpleMesh.h:27), for rt_upload_triangle_colors.  No counterpart in the
 * reference, whose shader colours every triangle red.  RT_ERR_INVALID_ARG as for rt_mesh_to_triangles. */
int      rt_mesh_triangle_colors(const rt_mesh* mesh, float* rgb);
/* The rest of SimpleMesh: an empty mesh (:42), clear :47-50, addVertex :93-97, addFace
 * :99-104 (both return the new element's index, 0xffffffff when memory runs out; addFace does not look at the
 * indices), setColor :87-91, writeMesh :217-245 (a COFF
 * file; colours are written as int(colour), as there), joinMeshes :251-288 (pose1to2: 4x4
 * column-major applied to mesh1's positions, NULL = identity), sphere :293-317 (the
 * 12-vertex, 20-face icosahedron). */
int      rt_mesh_create(rt_mesh** out);
void     rt_mesh_clear(rt_mesh* mesh);
uint32_t rt_mesh_add_vertex(rt_mesh* mesh, const float position_xyzw[4], const float color_rgba[4]);
uint32_t rt_mesh_add_face(rt_mesh* mesh, uint32_t idx0, uint32_t idx1, uint32_t idx2);
void     rt_mesh_set_color(rt_mesh* mesh, float r, float g, float b, float a);
int      rt_mesh_write(const rt_mesh* mesh, const char* off_path);
int      rt_mesh_join(rt_mesh** out, const rt_mesh* mesh1, const rt_mesh* mesh2, const float pose1to2_colmajor[16]);
int      rt_mesh_sphere(rt_mesh** out, const float center[3], float scale, const float color_rgba[4]);

/* Library / device facts for reports. */
int rt_device_info(int device, char* name, size_t name_len, int* compute_units,
                   int* clock_mhz);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
