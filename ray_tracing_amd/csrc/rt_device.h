// rt_device.h — device-side data layout shared by the kernels and the host API.
#pragma once
#include <stdint.h>

namespace rtd {

// One shape as the kernels read it: 48 bytes, three 16-byte words, repacked
// from the 144-byte Shape of the reference (src/gpu_version/shape.h:44-54) at
// upload time.  The closest-hit loop reads word 0 and 1 with wave-uniform
// (scalar) loads; after the loop each lane reads the record it hit.
struct alignas(16) DevShape {
    float    cx, cy, cz;   // translation: transformation column 3
    float    p0;           // Circle: radius*radius.  Box: side (transformation[0][0])
    float    p1;           // Box: 1.0f / side
    uint32_t type;         // 0 Circle, 1 Box
    uint32_t mat_type;     // 0 Light, 1 Lambertian, 2 Metal, 3 Dielectric, 100 Debug
    float    fuzz;         // material.data[1]
    float    r, g, b;      // material.color
    float    ior;          // material.data[0]
};
static_assert(sizeof(DevShape) == 48, "DevShape must be three 16-byte words");

// A triangle keeps the reference's 64-byte SimpleTriangle layout
// (src/gpu_version/shape.h:103-108): position.xyz _, a.xyz beta, m1.xyz _, m2.xyz _.
struct alignas(16) DevTriangle {
    float px, py, pz, pw;
    float ax, ay, az, beta;
    float m1x, m1y, m1z, m1w;
    float m2x, m2y, m2z, m2w;
};
static_assert(sizeof(DevTriangle) == 64, "DevTriangle must match SimpleTriangle");

// Everything one trace launch needs besides the scene arrays; passed by value
// in the kernel-argument segment, so every field is read with scalar loads.
struct TraceParams {
    float    cam[21];          // Camera::data (src/gpu_version/camera.h:117)
    float    tri_albedo[3];
    uint32_t width, height;
    uint32_t n_shapes, n_tris;
    uint32_t spp, first_sample, max_depth, grid_n;
    uint32_t seed_key;         // hash of the 64-bit seed
    uint32_t sample_mode, flags;
    uint32_t shard_index, shard_count, band_rows;
    uint32_t owned_rows;       // rows of this shard
    uint32_t trace_sample;     // TRACE launches: the one sample to follow
};

// Slots of the statistics block (uint64 each).
enum { ST_SAMPLES = 0, ST_SEGMENTS, ST_DIFFUSE, ST_SPECULAR, ST_TERMINAL, ST_COUNT };

constexpr int TILE_W = 8;      // pixels per workgroup in x (two 8-wide wave tiles)
constexpr int TILE_H = 8;      // and in y
constexpr int BLOCK = 64;

}  // namespace rtd
