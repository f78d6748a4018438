// rt_device.h — device-side data layout shared by the kernels and the host API.
#pragma once
#include <stdint.h>

namespace rtd {

// The scene as the kernels read it: ONE contiguous array of 16-byte words
// ("blob") in device memory, built at upload time from the reference's
// 144-byte Shape (src/gpu_version/shape.h:44-54) and 64-byte SimpleTriangle
// (shape.h:103-108) arrays.  Every workgroup copies the first `lds_words`
// words into LDS before tracing and addresses them there: the whole blob, or
// — when a mesh makes it larger than the LDS budget — the shapes only, the
// mesh and its tree then being read from the blob in global memory with
// wave-uniform addresses.
//
//   rec  [4 * (n_shapes + 1)] per shape, by the shape's index in the uploaded array, and one more for "nothing hit":
//                           word 0: centre.xyz, scale (Circle: radius, Box: side)
//                           word 1: colour.rgb (x 0.6 for Lambertian under CPU rules; 1 for a material the shader
//                                   does not know), tag = material type | shape type << 16 | class << 24
//                                   class: 0 the path ends here and returns throughput x colour x emit,
//                                          1 Lambertian, 2 metal, 3 dielectric, 4 debug (returns the normal)
//                           word 2: fuzz, emit (1: light, or unknown material under the shader's rules; else 0),
//                                   1 / scale (Circle: IEEE quotient; Box: rcp_nr), place in cub (uint32; cubes only)
//                           word 3: dielectric: 1/ior, Schlick's r0 = ((1-ior)/(1+ior))^2; box: 1/side y, 1/side z
//                         the record behind the shapes' is all zero: class 0, emit 0, colour 0
//   sph  [n_sph_pad]      spheres in upload order: centre.xyz, radius^2; padded to a
//                         multiple of 4 with NaN centres (never hit)
//   sid  [n_sph_pad / 4]  the spheres' shape indices, four per word
//   box  [3 * n_box]      boxes in upload order: {centre.xyz, 1/side x}, {shape index, 1/side y, 1/side z, 0},
//                         {half extents of the bounding box that holds everything the test accepts, 0}
//                         (the three sides are equal unless the context asks for a Box's own sides)
//   cub  [4 * n_cube]     the CPU tracer's cubes in upload order: rows of the rotation with the
//                         centre, {R(i,0), R(i,1), R(i,2), c_i} for i = 0..2, then {shape index, 0, 0, 0}
//   node [n3p+n2p+n1p+n0p] bounding spheres over the triangles, four levels (below)
//   tri  [4 * n_tri_pad]  SimpleTriangles ordered by recursive median splits and padded to a
//                         multiple of the leaf size with never-hit ones: {position.xyz, index in
//                         the uploaded array}, {a.xyz, beta}, {m1}, {m2}
//   inv  [n_tris / 4]     never staged in LDS: for each uploaded triangle, its place in the sorted list (uint32)
//   tcol [n_tris]         optional, never staged in LDS: albedo.rgb of each uploaded triangle (x 0.6 under CPU rules)
//
// Bounding-sphere tree: level 0 node i (a leaf) bounds the sorted triangles [leaf*i, leaf*(i+1)), leaf =
// 1 << SceneLayout::leaf_shift — chosen per mesh at upload: TREE_LEAF, or half of it for a mesh of at most
// SMALL_MESH_TRIS triangles (measured: leaves of two are 3 % faster than leaves of four on a 292-triangle mesh
// and 3-4 % slower on a 4 968-triangle one; on its first 600 / 1 000 / 1 500 / 2 200 / 3 200 faces: 1 % faster, 0.7 / 1.7 / 3.4 / 5.9 % slower); a node i of level l+1 bounds the nodes [TREE_FAN*i, TREE_FAN*(i+1)) of
// level l.  Every level is padded to a multiple of TREE_FAN with never-hit nodes (NaN
// centre).  A node is {centre.xyz, R}, R = 1.01 * radius + 1e-4 (radius: of a small enclosing ball of the node's vertices): the slack covers the
// rounding of the triangle test itself, so that a ray the triangle test could accept is
// never culled (see bound_hit in rt_kernels.hip).
#ifndef RT_TREE_LEAF
#define RT_TREE_LEAF 4
#endif
#ifndef RT_TREE_FAN
#define RT_TREE_FAN 8
#endif
#ifndef RT_TREE_TOP_MAX
#define RT_TREE_TOP_MAX 128        // a tree gets levels until its top one has at most this many nodes (RT_TREE_TOP=n in the environment: another limit, 1 for deep trees over small meshes in tests)
#endif
#ifndef RT_MAX_TREE_LEVELS
#define RT_MAX_TREE_LEVELS 8
#endif
// The walk handles the four levels 0 .. 3 with its queues and stacks.  A mesh whose level 3 would have more than
// TREE_TOP_MAX nodes (some 260 000 triangles and more) gets further levels on top, up to MAX_TREE_LEVELS in all, until the
// top one is that short: the wave descends through those together, depth first, and enters the level-3 nodes that any
// of its rays can touch (SceneLayout::tree_levels; 4 for every smaller mesh, whose upper levels may hold a single node).
// (Measured with a sheet of 20 000 triangles, ten level-3 nodes: a fifth level costs 8 % — scanning ten nodes is
// cheaper than the shared stack —; with 199 712 triangles, 98 level-3 nodes, a fifth level costs 2 % at 8 samples per
// pixel and 7 % at 32; with a million triangles, 489 level-3 nodes, one more level (62 nodes on top) or two (8) save a
// third of the frame, two 2-3 % more than one.)
constexpr uint32_t TREE_LEAF = RT_TREE_LEAF, TREE_FAN = RT_TREE_FAN, TREE_LEVELS = 4, MAX_TREE_LEVELS = RT_MAX_TREE_LEVELS, TREE_TOP_MAX = RT_TREE_TOP_MAX;
constexpr uint32_t SMALL_MESH_TRIS = 800u;
constexpr uint32_t TREE_LEAF_SMALL = TREE_LEAF >= 2u ? TREE_LEAF / 2u : TREE_LEAF;
static_assert((TREE_LEAF & (TREE_LEAF - 1u)) == 0u, "a leaf holds a power of two of triangles (lanes of a pass find theirs by shift and mask)");
struct SceneLayout {
    uint32_t n_shapes, n_sph_pad, n_box, n_cube, n_tris;
    uint32_t rec_off, sph_off, sid_off, box_off, cub_off, inv_off, tri_off;   // in 16-byte words
    uint32_t node_off[MAX_TREE_LEVELS];                     // level 0 .. tree_levels - 1
    uint32_t node_n[MAX_TREE_LEVELS];                       // real (unpadded) node counts
    uint32_t tree_levels;                                   // TREE_LEVELS .. MAX_TREE_LEVELS
    uint32_t total_words;                                   // whole blob
    uint32_t lds_words;                                     // prefix staged in LDS
    uint32_t tri_in_lds;                                    // 1 if the tree and tri are part of that prefix
    uint32_t has_glass;                                     // 1 if any shape has the dielectric material
    uint32_t tcol_off;                                      // 0: all triangles share TraceParams::tri_albedo
    uint32_t has_debug;                                     // 1 if any shape has the debug material
    uint32_t n_sph;                                         // spheres uploaded (n_sph_pad less the padding)
    uint32_t leaf_shift;                                    // a leaf of this scene's tree holds 1 << leaf_shift triangles (<= TREE_LEAF)
};

// Everything one trace launch needs besides the blob; passed by value in the
// kernel-argument segment.
struct TraceParams {
    float    cam[21];          // Camera::data (src/gpu_version/camera.h:117)
    float    tri_albedo[3];
    uint32_t width, height;
    uint32_t spp, first_sample, max_depth, grid_n;
    uint32_t seed_key;         // hash of the 64-bit seed
    uint32_t sample_mode, flags;
    uint32_t shard_index, shard_count, band_rows;
    uint32_t owned_rows;       // rows of this shard
    uint32_t trace_sample;     // TRACE launches: the one sample to follow
    uint32_t tiles_x, tiles_y; // 16x16-pixel tiles covering width x owned_rows
    uint32_t group_waves;      // waves per workgroup, 1 .. MAX_GROUP_WAVES; wave q of the launch traces quarter q & 3 of plan entry q >> 2
    uint32_t plan_entries;     // entries (16x16 tiles, or slices of one) in this launch's plan
    uint32_t rng_mode;         // 0 counter hash, 1 the shader's random() (RayTracing.glsl:44-51)
    uint32_t glsl_seed32;      // low word of the seed: the shader's `time` is (this + frame) / 10000
    float    inv_w, inv_h;     // 1 / width, 1 / height (RayTracing.glsl:9 divides the invocation id by them)
    uint32_t overwrite;        // 1: the accumulator's contents are dropped — sums start at zero and are stored, not added (a launch without sliced tiles)
    // The frame's resolve as the launch's last step (a launch without sliced tiles that ends the frame): every lane
    // writes its pixel's mean as resolve_kernel would.  resolve_out = nullptr: no resolve.
    uint32_t resolve_samples, resolve_gamma, resolve_flip;
    float*   resolve_out;
    SceneLayout scene;
};

// Slots of the statistics block (uint64 each).
enum { ST_SAMPLES = 0, ST_SEGMENTS, ST_DIFFUSE, ST_SPECULAR, ST_TERMINAL, ST_TRI_TESTS, ST_NODE_TESTS, ST_WAVE_ITERS, ST_UNITS, ST_BOX_TESTS, ST_BOX_BOUNDS,
       ST_SPHERE_TESTS, ST_PRIMARY_STEPS, ST_TREE_ENTRIES, ST_TREE_ENTERING_LANES, ST_COUNT };

// Candidate lists of the camera rays, one record per 8x8-pixel wave tile (tile_lists_kernel makes them once per scene,
// camera and frame size; trace_kernel's "primary steps" read them): every primitive that a camera ray through the
// tile could be accepted by — conservatively: whatever touches the pyramid from the eye through the tile.
//   word 0: number of tree leaves listed at the record's end, or TLIST_NONE: no list for this tile (more leaves than
//           fit) — its waves then trace camera rays like any others
//   word 1: boxes, bit b = box b in upload order among the boxes (scenes of up to 32)
//   word 2..5: spheres, bit s of the 128 = sphere s in upload order among the spheres
//   word 6..: the leaves
// A scene with more spheres or boxes than the masks hold has no lists.
#ifndef RT_TLIST_SPH_WORDS
#define RT_TLIST_SPH_WORDS 4
#endif
constexpr uint32_t TLIST_WORDS = 32u, TLIST_SPH_WORDS = RT_TLIST_SPH_WORDS, TLIST_LEAF0 = 2u + TLIST_SPH_WORDS, TLIST_MAX_LEAVES = TLIST_WORDS - TLIST_LEAF0,
                   TLIST_NONE = 0xffffffffu, TLIST_MAX_SPHERES = 32u * TLIST_SPH_WORDS, TLIST_MAX_BOXES = 32u,
                   TLIST_MAX_SPHERES_NO_MESH = 32u * (TLIST_WORDS - 2u);
// A scene without a mesh has no leaves to list: its sphere masks run on through the leaves' words (960 spheres).
constexpr uint32_t tlist_max_spheres(bool has_mesh) { return has_mesh ? TLIST_MAX_SPHERES : TLIST_MAX_SPHERES_NO_MESH; }
// A wave takes a primary step — only its lanes that are about to trace a camera ray take part, and they test the
// tile's candidates instead of walking the scene — when all its live lanes are such, or at least this many.
#ifndef RT_PSTEP_MIN
#define RT_PSTEP_MIN 24
#endif
constexpr uint32_t PSTEP_MIN = RT_PSTEP_MIN;

constexpr int TILE_W = 16;     // pixels per workgroup in x (two 8-wide wave tiles)
constexpr int TILE_H = 16;     // and in y
constexpr int BLOCK = 256;
constexpr uint32_t MAX_GROUP_WAVES = 8;   // instances with queues may be launched with up to this many waves per workgroup

// LDS a workgroup may use for the staged scene.  8 workgroups of 4 waves fill a
// CU's 32 wave slots; 160 KiB / 8 = 20 KiB each keeps that occupancy.  Scenes
// up to LDS_MAX_BYTES are still staged (fewer workgroups per CU); beyond it the
// triangles and their tree stay in global memory.
constexpr uint32_t LDS_FULL_OCCUPANCY_BYTES = 20u * 1024u;
constexpr uint32_t LDS_MAX_BYTES = 64u * 1024u;
// The shapes are always staged: a scene may take what one workgroup can have of a CU's LDS (one workgroup per CU then;
// about 1 800 spheres).  The runtime takes such a launch as it is (hipDeviceProp_t::sharedMemPerBlock is 160 KiB here).
constexpr uint32_t LDS_WORKGROUP_MAX_BYTES = 160u * 1024u;
// With boxes or triangles in the scene every wave of a workgroup owns, behind the staged scene, a queue of
// (lane, leaf) pairs (QUEUE_ENTRIES x 4 bytes), one 8-byte slot per lane for the closest hit found, and a stack of
// (lane, level, node) triples.  A node is put there only while fewer than SPREAD_BELOW lanes touch it (more: the
// wave tests the node's children itself), and stack and queue are worked off whenever they hold a full pass (at
// most 8 and 16 pairs).  That bounds them: a staged mesh queues only level-1 nodes, 7 left over + 23 new;  for a
// mesh in global memory, whose nodes of all levels are queued, a pass takes 8 triples off the top and may put 64
// children there, which the next pass takes first: 30 + 2 x 56 + 64 at most (three levels above the leaves).
// (Measured on the 292-triangle mesh scene: thresholds of 8, 16, 24 and 40 lanes give 14.9, 14.7, 14.7 and 14.9 ms,
// on the 4 968-triangle one 18.9, 18.3, 18.3 and 18.7.)
#ifndef RT_SPREAD_BELOW
#define RT_SPREAD_BELOW 24
#endif
constexpr uint32_t QUEUE_ENTRIES = 128u, SPREAD_BELOW = RT_SPREAD_BELOW;
#ifndef RT_NODE_STACK_STAGED
#define RT_NODE_STACK_STAGED 32u
#endif
// A pass over queued nodes gives TREE_FAN / NODE_CPL lanes to each triple, every lane testing NODE_CPL children: fewer,
// fuller passes (a pass costs some 60 instructions around its tests).
#ifndef RT_NODE_CPL
#define RT_NODE_CPL 1
#endif
constexpr uint32_t NODE_CPL = RT_NODE_CPL, NODE_LPT = TREE_FAN / NODE_CPL;
static_assert(NODE_CPL >= 1u && NODE_LPT * NODE_CPL == TREE_FAN, "children per lane must divide the fan-out");
constexpr uint32_t NODE_STACK_STAGED = RT_NODE_STACK_STAGED, NODE_STACK_GLOBAL = 256u * NODE_CPL;
// What is left below a full pass, plus what one step can add, must fit (64 lanes; a pass takes 64 / TREE_LEAF pairs
// or 64 / TREE_FAN triples and can put 64 new ones there; three levels above the leaves).
static_assert(TREE_LEAF >= 1u && TREE_LEAF <= 64u && TREE_FAN >= 2u && TREE_FAN <= 64u && SPREAD_BELOW <= 64u, "tree shape");
static_assert((64u / TREE_LEAF_SMALL - 1u) + 64u <= QUEUE_ENTRIES, "the leaf queue overflows");
static_assert((64u / NODE_LPT - 1u) + (SPREAD_BELOW - 1u) <= NODE_STACK_STAGED, "the node stack of a staged mesh overflows");
static_assert((64u / NODE_LPT - 1u) + (SPREAD_BELOW - 1u) + (TREE_LEVELS - 2u) * (64u * NODE_CPL - 64u / NODE_LPT) + 64u * NODE_CPL <= NODE_STACK_GLOBAL,
              "the node stack of a mesh in global memory overflows");
static_assert(TREE_LEVELS == 4u, "stack entries carry two bits of level, the walk has four nested loops");
// Behind them, while the wave's lanes share its paths: the sums of the wave's 64 pixels, three 32-bit words each,
// and a table of the pixels' ray constants, 16 bytes each.
#ifndef RT_POOL
#define RT_POOL 1
#endif
constexpr uint32_t POOL_SUM_BYTES = RT_POOL ? 3u * 64u * 4u + 64u * 16u : 0u;
constexpr uint32_t WAVE_LDS_WORDS = (QUEUE_ENTRIES * 4u + 64u * 8u + NODE_STACK_STAGED * 4u + POOL_SUM_BYTES) / 16u;
// ... and, for a mesh in global memory, the stack of (level, node) the wave shares while it descends through the levels
// above the third together: depth first, at most TREE_FAN - 1 siblings waiting per level and the node in hand.
constexpr uint32_t UPPER_STACK_ENTRIES = 32u;
static_assert((MAX_TREE_LEVELS - TREE_LEVELS) * (TREE_FAN - 1u) + 1u <= UPPER_STACK_ENTRIES, "the shared stack of the upper levels overflows");
constexpr uint32_t WAVE_LDS_WORDS_GLOBAL = (QUEUE_ENTRIES * 4u + 64u * 8u + NODE_STACK_GLOBAL * 4u + POOL_SUM_BYTES + UPPER_STACK_ENTRIES * 4u) / 16u;
// The 16-byte words a workgroup of `group_waves` waves needs behind the staged scene.
constexpr uint32_t lds_wave_words(uint32_t group_waves, bool mesh_in_global) { return group_waves * (mesh_in_global ? WAVE_LDS_WORDS_GLOBAL : WAVE_LDS_WORDS); }

}  // namespace rtd
