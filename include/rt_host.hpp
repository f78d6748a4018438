// rt_host.hpp — header-only C++17 mirror of the reference's host classes on
// top of the C ABI (rt_mi355x.h).  Same class and method names and the same
// argument meaning as src/gpu_version/shape.h, camera.h,
// ray_tracing_compute_shader.h and src/SimpleMesh.h of jaafer06/ray_tracing,
// without their Eigen / OpenGL dependencies, so that a program written like
// src/main.cpp:88-151 drives the HIP core.  All arithmetic is done by the
// library; these classes only hold the reference's byte layouts.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_mi355x.h"

namespace rt {

using Vector3f = std::array<float, 3>;

// ---- materials: src/gpu_version/shape.h:5-42 --------------------------------
class Material : public rt_material {
protected:
    explicit Material(unsigned int t) { std::memset(static_cast<rt_material*>(this), 0, sizeof(rt_material)); type = t; }
    void setColor(const Vector3f& c) { color[0] = c[0]; color[1] = c[1]; color[2] = c[2]; }
};
class Light : public Material {              // shape.h:14-19
public:
    explicit Light(const Vector3f& color) : Material(RT_MAT_LIGHT) { setColor(color); }
};
class Lambertian : public Material {         // shape.h:21-26
public:
    explicit Lambertian(const Vector3f& color) : Material(RT_MAT_LAMBERTIAN) { setColor(color); }
};
class Metal : public Material {              // shape.h:28-34; fuzz = the CPU tracer's `fuzzy`, scene.h:71
public:
    explicit Metal(const Vector3f& color, float refraction = 1, float fuzz = 0) : Material(RT_MAT_METAL)
    {
        setColor(color);
        data[0] = refraction;
        data[1] = fuzz;
    }
};
class Dielectric : public Material {         // extension, no counterpart in the reference: data[0] = index of refraction
public:
    explicit Dielectric(const Vector3f& color, float ior = 1.5f) : Material(RT_MAT_DIELECTRIC) { setColor(color); data[0] = ior; }
};
class DebugMaterial : public Material {      // shape.h:37-42
public:
    explicit DebugMaterial(const Vector3f& color) : Material(RT_MAT_DEBUG) { setColor(color); }
};
static_assert(sizeof(Material) == 32, "Material must keep the layout of shape.h:5-12");

// ---- shapes: shape.h:44-70 -----------------------------------------------------
class Shape : public rt_shape {
protected:
    Shape() = default;
};
class Circle : public Shape {                // shape.h:56-62
public:
    Circle(const Vector3f& translation, float radius, const Material& material)
    {
        rt_make_circle(this, translation.data(), radius, &material);
    }
};
class Box : public Shape {                   // shape.h:64-70
public:
    Box(const Vector3f& translation, const Vector3f& sides, const Material& material)
    {
        rt_make_box(this, translation.data(), sides.data(), &material);
    }
};
class Cube : public Shape {                  // the CPU tracer's Cube, src/cpu_version/scene.h:142-217
public:
    Cube(const Vector3f& center, const Material& material) { rt_make_cube(this, center.data(), &material); }
    void rotate(float angle) { rt_cube_rotate(this, angle); }   // scene.h:206-208
};
static_assert(sizeof(Shape) == 144, "Shape must keep the layout of shape.h:44-54");

class SimpleTriangle : public rt_triangle {  // shape.h:81-109
public:
    SimpleTriangle(const Vector3f& p1, const Vector3f& p2, const Vector3f& p3)
    {
        rt_make_triangle(this, p1.data(), p2.data(), p3.data());
    }
};
static_assert(sizeof(SimpleTriangle) == 64, "SimpleTriangle must keep the layout of shape.h:103-108");

// ---- camera: src/gpu_version/camera.h:8-122 --------------------------------
class Camera {
public:
    enum class Direction { FORWARD, BACKWARD, RIGHT, LEFT };    // camera.h:10-12

    Camera(unsigned int width, unsigned int height, float focalLength = 1, const Vector3f& position = { 1, 1, 6 })
        : width(width), height(height), focalLength(focalLength)
    {
        rt_camera_init(data, width, height, focalLength, position.data());
    }
    Vector3f getCameraPosition() const { return { data[0], data[1], data[2] }; }
    Vector3f getCameraDirection() const { return { data[4], data[5], data[6] }; }
    Vector3f getUpperLeft() const { return { data[8], data[9], data[10] }; }
    Vector3f getUpDirection() const { return { data[12], data[13], data[14] }; }
    Vector3f getRightDirection() const { return { data[16], data[17], data[18] }; }
    float getWorldStep() const { return data[19]; }
    void rotate(float horizontal, float vertical) { rt_camera_rotate(data, width, height, focalLength, horizontal, vertical); }
    void move(Direction direction) { rt_camera_move(data, width, height, focalLength, static_cast<int>(direction)); }

    static constexpr unsigned int dataSize = RT_CAMERA_FLOATS;  // camera.h:115
    unsigned int width, height;
    float focalLength;
    float data[RT_CAMERA_FLOATS];
};

// ---- mesh: src/SimpleMesh.h:22-323 ----------------------------------------
struct Vertex { float position[4]; float color[4]; };           // SimpleMesh.h:24-30 (normal unused)
struct Triangle { unsigned int idx0, idx1, idx2; };             // SimpleMesh.h:32-41

class SimpleMesh {
public:
    SimpleMesh() { if (rt_mesh_create(&mesh) != RT_OK) throw std::bad_alloc(); }           // SimpleMesh.h:42
    SimpleMesh(const SimpleMesh&) = delete;
    SimpleMesh& operator=(const SimpleMesh&) = delete;
    SimpleMesh(SimpleMesh&& other) noexcept : mesh(other.mesh), vertices(std::move(other.vertices)), triangles(std::move(other.triangles)) { other.mesh = nullptr; }
    ~SimpleMesh() { rt_mesh_free(mesh); }

    void clear() { rt_mesh_clear(mesh); refresh(); }            // SimpleMesh.h:47-50
    unsigned int addVertex(const Vertex& v) { unsigned int id = rt_mesh_add_vertex(mesh, v.position, v.color); refresh(); return id; }   // :93-97
    unsigned int addFace(unsigned int idx0, unsigned int idx1, unsigned int idx2)             // :99-104
    {
        unsigned int id = rt_mesh_add_face(mesh, idx0, idx1, idx2);
        refresh();
        return id;
    }
    void setColor(float r, float g, float b, float a) { rt_mesh_set_color(mesh, r, g, b, a); refresh(); }   // :87-91
    bool writeMesh(const std::string& filename) const { return rt_mesh_write(mesh, filename.c_str()) == RT_OK; }   // :217-245
    // pose1to2: 4x4 column-major (Eigen's default), applied to mesh1's positions.  SimpleMesh.h:251-288
    static SimpleMesh joinMeshes(const SimpleMesh& mesh1, const SimpleMesh& mesh2, const float* pose1to2 = nullptr)
    {
        rt_mesh* joined = nullptr;
        if (rt_mesh_join(&joined, mesh1.mesh, mesh2.mesh, pose1to2) != RT_OK) throw std::bad_alloc();
        return SimpleMesh(joined);
    }
    static SimpleMesh sphere(const Vector3f& center, float scale = 1.f, const std::array<float, 4>& color = { 0, 0, 1, 1 })   // :293-317
    {
        rt_mesh* m = nullptr;
        if (rt_mesh_sphere(&m, center.data(), scale, color.data()) != RT_OK) throw std::bad_alloc();
        return SimpleMesh(m);
    }
    // rgb per face, the mean of its vertex colours: what rt_upload_triangle_colors takes.
    std::vector<float> triangleColors() const
    {
        std::vector<float> rgb(3 * triangles.size());
        if (rt_mesh_triangle_colors(mesh, rgb.data()) != RT_OK) throw std::out_of_range("SimpleMesh: a face names a vertex the mesh does not have");
        return rgb;
    }

    bool loadMesh(const std::string& filename)                  // SimpleMesh.h:145-215
    {
        rt_mesh* loaded = nullptr;
        if (rt_mesh_load(&loaded, filename.c_str()) != RT_OK) { clear(); return false; }   // :147-148 clear first
        rt_mesh_free(mesh);
        mesh = loaded;
        refresh();
        return true;
    }
    void normalize() { rt_mesh_normalize(mesh); refresh(); }    // SimpleMesh.h:60-68
    void resize() { rt_mesh_resize(mesh); refresh(); }          // SimpleMesh.h:79-85
    // rotation: 3x3 row-major; SimpleMesh.h:122-126
    void transform(const std::array<float, 9>& rotation, const Vector3f& translation)
    {
        rt_mesh_transform(mesh, rotation.data(), translation.data());
        refresh();
    }
    const std::vector<Vertex>& getVertices() const { return vertices; }
    const std::vector<Triangle>& getTriangles() const { return triangles; }
    // The loop of src/main.cpp:113-122.
    std::vector<SimpleTriangle> toSimpleTriangles() const
    {
        std::vector<SimpleTriangle> out;
        out.reserve(triangles.size());
        for (const Triangle& t : triangles) {
            // addFace takes any index (SimpleMesh.h:99-104): checked here, where it is followed
            if (t.idx0 >= vertices.size() || t.idx1 >= vertices.size() || t.idx2 >= vertices.size())
                throw std::out_of_range("SimpleMesh: a face names a vertex the mesh does not have");
            const float* a = vertices[t.idx0].position; const float* b = vertices[t.idx1].position; const float* c = vertices[t.idx2].position;
            out.emplace_back(Vector3f{ a[0], a[1], a[2] }, Vector3f{ b[0], b[1], b[2] }, Vector3f{ c[0], c[1], c[2] });
        }
        return out;
    }

private:
    explicit SimpleMesh(rt_mesh* adopted) : mesh(adopted) { refresh(); }
    void refresh()
    {
        const uint32_t nv = rt_mesh_vertex_count(mesh), nt = rt_mesh_triangle_count(mesh);
        std::vector<float> pos(4 * (size_t)nv), col(4 * (size_t)nv);
        std::vector<uint32_t> idx(3 * (size_t)nt);
        rt_mesh_copy_vertices(mesh, pos.data(), col.data());
        rt_mesh_copy_indices(mesh, idx.data());
        vertices.resize(nv);
        triangles.resize(nt);
        for (uint32_t i = 0; i < nv; ++i) {
            std::memcpy(vertices[i].position, &pos[4 * i], 16);
            std::memcpy(vertices[i].color, &col[4 * i], 16);
        }
        for (uint32_t i = 0; i < nt; ++i) triangles[i] = { idx[3 * i], idx[3 * i + 1], idx[3 * i + 2] };
    }
    rt_mesh* mesh = nullptr;
    std::vector<Vertex> vertices;
    std::vector<Triangle> triangles;
};

// ---- render-call surface: src/gpu_version/ray_tracing_compute_shader.h:9-67 --
class RayTracingComputeShader {
public:
    explicit RayTracingComputeShader(Camera& camera, int device = 0, rt_semantics semantics = RT_SEM_GLSL,
                                     rt_sample_mode sampleMode = RT_SAMPLE_JITTER, unsigned int flags = 0, rt_rng rng = RT_RNG_HASH)
        : camera(camera)
    {
        rt_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.device = device;
        cfg.semantics = semantics;
        cfg.sample_mode = sampleMode;
        cfg.flags = flags;
        cfg.rng = rng;
        cfg.triangle_albedo[0] = 1.0f;                          // RayTracing.glsl:275
        if (rt_create(&ctx, &cfg) != RT_OK) throw std::runtime_error(rt_last_error(nullptr));
    }
    RayTracingComputeShader(const RayTracingComputeShader&) = delete;
    RayTracingComputeShader& operator=(const RayTracingComputeShader&) = delete;
    ~RayTracingComputeShader() { rt_destroy(ctx); }

    void updateCameraBuffer() { check(rt_set_camera(ctx, camera.data)); }                       // :34-37
    void compute(unsigned int x, unsigned int y, unsigned int z = 1)                             // :39-42
    {
        (void)z;
        check(rt_render(ctx, x, y, samplesPerFrame, maxDepth, seed, frame * samplesPerFrame, 0));
        ++frame;
    }
    void updateShapeBuffer()                                                                     // :44-57
    {
        check(rt_upload_shapes(ctx, shapes.data(), (uint32_t)shapes.size()));
        check(rt_upload_triangles(ctx, triangles.data(), (uint32_t)triangles.size()));
        if (!triangleColors.empty()) check(rt_upload_triangle_colors(ctx, triangleColors.data(), (uint32_t)(triangleColors.size() / 3)));
    }
    // The converter dispatch of src/main.cpp:147-149 (PixelConversion.glsl:12-28).
    void convert(unsigned int n_samples) { check(rt_convert(ctx, n_samples)); }
    // The RGBA32F texture of src/main.cpp:72-79, width*height*4 floats.
    void readImage(float* rgba) { check(rt_read_image(ctx, rgba)); }
    // One frame's mean without the running-mean image (rayTracer.h:105 when gammaSqrt).
    void resolve(float* rgba, bool gammaSqrt = false, bool flipRows = true) { check(rt_resolve(ctx, rgba, gammaSqrt, flipRows)); }
    void restart() { check(rt_reset_accum(ctx)); frame = 0; }                                    // main.cpp:179,194
    rt_ctx* handle() { return ctx; }

    std::vector<Shape> shapes;                  // :59
    std::vector<SimpleTriangle> triangles;      // :60
    std::vector<float> triangleColors;          // empty, or rgb per triangle (in place of RayTracing.glsl:275's red)
    unsigned int samplesPerFrame = 100;         // RayTracing.glsl:2,6
    unsigned int maxDepth = 10;                 // RayTracing.glsl:300
    uint64_t seed = 1;                          // stands in for the `time` uniform, main.cpp:139-141
    unsigned int frame = 0;

private:
    void check(int rc) { if (rc != RT_OK) throw std::runtime_error(rt_last_error(ctx)); }
    Camera& camera;
    rt_ctx* ctx = nullptr;
};

}  // namespace rt
