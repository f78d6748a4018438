// rt_host.cpp — host-side scene construction behind the C ABI: the reference's
// Camera, Circle/Box, SimpleTriangle and SimpleMesh constructors in plain C++,
// without Eigen, producing byte-identical structs.
//
// Host arithmetic follows the expressions of the reference as Eigen evaluates
// them (no fused operations, three-term sums left to right); this file is
// compiled with -ffp-contract=off.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace {

struct Vec3 { float x, y, z; };
inline Vec3 operator+(Vec3 a, Vec3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline Vec3 operator*(Vec3 a, float s) { return { a.x * s, a.y * s, a.z * s }; }
inline Vec3 operator/(Vec3 a, float s) { return { a.x / s, a.y / s, a.z / s }; }
inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline Vec3 load3(const float* p) { return { p[0], p[1], p[2] }; }
inline void store3(float* p, Vec3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

struct Mat3 { float m[3][3]; };   // row-major
inline Vec3 mul(const Mat3& r, Vec3 v)
{
    return { (r.m[0][0] * v.x + r.m[0][1] * v.y) + r.m[0][2] * v.z,
             (r.m[1][0] * v.x + r.m[1][1] * v.y) + r.m[1][2] * v.z,
             (r.m[2][0] * v.x + r.m[2][1] * v.y) + r.m[2][2] * v.z };
}
inline Mat3 mul(const Mat3& a, const Mat3& b)
{
    Mat3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = (a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j]) + a.m[i][2] * b.m[2][j];
    return r;
}
// Rotation by `angle` about the unit `axis` (Rodrigues), the matrix
// Eigen::AngleAxisf(angle, axis).toRotationMatrix() stands for in camera.h:86-87.
Mat3 angle_axis(float angle, Vec3 axis)
{
    const float s = std::sin(angle), c = std::cos(angle);
    const Vec3 sa = axis * s, ca = axis * (1.0f - c);
    Mat3 r;
    float t;
    t = ca.x * axis.y; r.m[0][1] = t - sa.z; r.m[1][0] = t + sa.z;
    t = ca.x * axis.z; r.m[0][2] = t + sa.y; r.m[2][0] = t - sa.y;
    t = ca.y * axis.z; r.m[1][2] = t - sa.x; r.m[2][1] = t + sa.x;
    r.m[0][0] = ca.x * axis.x + c; r.m[1][1] = ca.y * axis.y + c; r.m[2][2] = ca.z * axis.z + c;
    return r;
}

// worldWidth / worldHeight of camera.h:23-24: unsigned product, double sqrt,
// narrowed to float on assignment.
void world_extent(uint32_t width, uint32_t height, float& ww, float& wh)
{
    const unsigned int area = width * height;
    ww = (float)(width * (1 / std::sqrt((double)area)));
    wh = (float)(height * (1 / std::sqrt((double)area)));
}

// setUpperLeft(...) of camera.h:91,112.
void update_upper_left(float* data, uint32_t width, uint32_t height, float focal)
{
    float ww, wh; world_extent(width, height, ww, wh);
    const Vec3 right = load3(data + 16), up = load3(data + 12), dir = load3(data + 4), pos = load3(data);
    store3(data + 8, ((right * (-ww / 2)) + (up * (wh / 2))) + dir * focal + pos);
}

}  // namespace

struct rt_mesh {
    struct Vertex { float position[4]; float color[4]; float normal[3]; };   // SimpleMesh.h:24-30
    std::vector<Vertex> vertices;
    std::vector<uint32_t> indices;     // three per face (SimpleMesh.h:32-41)
};

extern "C" {

// Camera::Camera, src/gpu_version/camera.h:14-28.
void rt_camera_init(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height, float focal, const float position[3])
{
    memset(data, 0, sizeof(float) * RT_CAMERA_FLOATS);                    // :15
    store3(data + 0, load3(position));                                    // :17
    store3(data + 4, { 0, 0, -1 });                                       // :18
    store3(data + 12, { 0, 1, 0 });                                       // :20
    store3(data + 16, { 1, 0, 0 });                                       // :21
    float ww, wh; world_extent(width, height, ww, wh);                    // :23-24
    data[19] = ww / (float)width;                                         // :25
    store3(data + 8, Vec3{ -ww / 2, wh / 2, -focal } + load3(position));  // :26
}

// Camera::rotate, camera.h:85-92.
void rt_camera_rotate(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height, float focal, float horizontal, float vertical)
{
    Mat3 rot = angle_axis(horizontal, load3(data + 12));
    rot = mul(angle_axis(vertical, mul(rot, load3(data + 16))), rot);
    store3(data + 12, mul(rot, load3(data + 12)));
    store3(data + 16, mul(rot, load3(data + 16)));
    store3(data + 4, mul(rot, load3(data + 4)));
    update_upper_left(data, width, height, focal);
}

// Camera::move, camera.h:94-113.
void rt_camera_move(float data[RT_CAMERA_FLOATS], uint32_t width, uint32_t height, float focal, int direction)
{
    const Vec3 pos = load3(data), dir = load3(data + 4), right = load3(data + 16);
    switch (direction) {
    case 0: store3(data, pos + dir * 0.2f); break;
    case 1: store3(data, pos - dir * 0.2f); break;
    case 2: store3(data, pos + right * 0.2f); break;
    case 3: store3(data, pos - right * 0.2f); break;
    default: break;
    }
    update_upper_left(data, width, height, focal);
}

static void shape_base(rt_shape* out, uint32_t type, const rt_material* material)
{
    memset(out, 0, sizeof *out);
    out->transformation[0] = out->transformation[5] = out->transformation[10] = out->transformation[15] = 1.0f;  // shape.h:47
    out->type = type;
    if (material) out->material = *material;
}

// Circle::Circle, shape.h:58-61.
void rt_make_circle(rt_shape* out, const float translation[3], float radius, const rt_material* material)
{
    shape_base(out, RT_SHAPE_CIRCLE, material);
    out->transformation[12] = translation[0]; out->transformation[13] = translation[1]; out->transformation[14] = translation[2];
    out->transformation[0] = radius;
}

// Box::Box, shape.h:66-69.
void rt_make_box(rt_shape* out, const float translation[3], const float sides[3], const rt_material* material)
{
    shape_base(out, RT_SHAPE_BOX, material);
    out->transformation[0] = sides[0]; out->transformation[5] = sides[1]; out->transformation[10] = sides[2];
    out->transformation[12] = translation[0]; out->transformation[13] = translation[1]; out->transformation[14] = translation[2];
}

// Cube::Cube, src/cpu_version/scene.h:144-148: identity rotation.
void rt_make_cube(rt_shape* out, const float center[3], const rt_material* material)
{
    shape_base(out, RT_SHAPE_CUBE, material);
    out->transformation[12] = center[0]; out->transformation[13] = center[1]; out->transformation[14] = center[2];
}

// Cube::rotate, scene.h:206-208: rotation *= (AngleAxisf(angle*pi, UnitX) * AngleAxisf(angle*pi, UnitY)).toRotationMatrix().
void rt_cube_rotate(rt_shape* cube, float angle)
{
    if (!cube) return;
    const float a = angle * 3.14159265358979323846f;          // `pi` is a float-converted constant there
    const Mat3 step = mul(angle_axis(a, { 1, 0, 0 }), angle_axis(a, { 0, 1, 0 }));
    Mat3 r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = cube->transformation[4 * j + i];   // column-major
    r = mul(r, step);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cube->transformation[4 * j + i] = r.m[i][j];
}

// SimpleTriangle::SimpleTriangle, shape.h:83-102.
void rt_make_triangle(rt_triangle* out, const float p1[3], const float p2[3], const float p3[3])
{
    memset(out, 0, sizeof *out);
    const Vec3 a1 = load3(p1);
    store3(out->position, a1);                                   // :84
    const Vec3 v1 = load3(p2) - a1, v2 = load3(p3) - a1;         // :86-87
    const Vec3 c = cross(v1, v2);
    const Vec3 a = c / std::sqrt(dot(c, c));                     // :88
    store3(out->a, a);
    out->beta = dot(a, a1);                                      // :89
    Vec3 m1 = v1 - v2 * (dot(v1, v2) / dot(v2, v2));             // :93
    m1 = m1 / dot(m1, v1);                                       // :94
    Vec3 m2 = v2 - v1 * (dot(v1, v2) / dot(v1, v1));             // :96
    m2 = m2 / dot(m2, v2);                                       // :97
    store3(out->m1, m1); store3(out->m2, m2);
}

// SimpleMesh::loadMesh, src/SimpleMesh.h:145-215.
int rt_mesh_load(rt_mesh** out, const char* path)
{
    if (!out || !path) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    FILE* f = fopen(path, "r");
    if (!f) return RT_ERR_IO;                                    // :151-154
    char tag[8] = { 0 };
    unsigned nv = 0, np = 0, ne = 0;
    if (fscanf(f, "%4s", tag) != 1 || fscanf(f, "%u %u %u", &nv, &np, &ne) != 3) { fclose(f); return RT_ERR_IO; }
    const bool coff = strcmp(tag, "COFF") == 0;
    if (!coff && strcmp(tag, "OFF") != 0) { fclose(f); return RT_ERR_IO; }   // :198-201
    // Every vertex and every face takes at least six characters of the file ("0 0 0\n", "3 0 0 0\n"): a header that
    // promises more than the file can hold is rejected before anything is reserved for it.
    {
        const long here = ftell(f);
        long size = -1;
        if (here >= 0 && fseek(f, 0, SEEK_END) == 0) { size = ftell(f); (void)fseek(f, here, SEEK_SET); }
        if (size < 0 || ((uint64_t)nv + (uint64_t)np) * 6u > (uint64_t)size) { fclose(f); return RT_ERR_IO; }
    }
    rt_mesh* mesh = new (std::nothrow) rt_mesh();
    if (!mesh) { fclose(f); return RT_ERR_OUT_OF_MEMORY; }
    try {
    mesh->vertices.reserve(nv); mesh->indices.reserve((size_t)np * 3);
    for (unsigned i = 0; i < nv; ++i) {
        rt_mesh::Vertex v{};
        if (fscanf(f, "%f %f %f", &v.position[0], &v.position[1], &v.position[2]) != 3) { delete mesh; fclose(f); return RT_ERR_IO; }
        v.position[3] = 1.0f;                                    // :175,190
        if (coff) {
            float c[4];
            if (fscanf(f, "%f %f %f %f", &c[0], &c[1], &c[2], &c[3]) != 4) { delete mesh; fclose(f); return RT_ERR_IO; }
            for (int k = 0; k < 4; ++k) v.color[k] = c[k] / 255;  // :179
        } else {
            v.color[3] = 255;                                    // :191-194
        }
        mesh->vertices.push_back(v);
    }
    for (unsigned i = 0; i < np; ++i) {
        unsigned k = 0, a = 0, b = 0, c = 0;
        if (fscanf(f, "%u", &k) != 1 || k != 3 ||                // :207 (triangles only)
            fscanf(f, "%u %u %u", &a, &b, &c) != 3 || a >= nv || b >= nv || c >= nv) { delete mesh; fclose(f); return RT_ERR_IO; }
        mesh->indices.push_back(a); mesh->indices.push_back(b); mesh->indices.push_back(c);
    }
    } catch (...) { delete mesh; fclose(f); return RT_ERR_OUT_OF_MEMORY; }
    fclose(f);
    *out = mesh;
    return RT_OK;
}

void rt_mesh_free(rt_mesh* mesh) { delete mesh; }
uint32_t rt_mesh_vertex_count(const rt_mesh* mesh) { return mesh ? (uint32_t)mesh->vertices.size() : 0; }
uint32_t rt_mesh_triangle_count(const rt_mesh* mesh) { return mesh ? (uint32_t)(mesh->indices.size() / 3) : 0; }

// get_mesh_center / get_mesh_center_and_radius, SimpleMesh.h:52-58,70-77.
static void center_and_radius(const rt_mesh* mesh, Vec3& center, float& radius)
{
    center = { 0, 0, 0 };
    for (const auto& v : mesh->vertices) center = center + load3(v.position);
    center = center / (float)mesh->vertices.size();
    radius = 0;
    for (const auto& v : mesh->vertices) {
        const Vec3 d = load3(v.position) - center;
        radius = std::max(std::sqrt(dot(d, d)), radius);
    }
}

// SimpleMesh::normalize, SimpleMesh.h:60-68.
void rt_mesh_normalize(rt_mesh* mesh)
{
    if (!mesh || mesh->vertices.empty()) return;
    Vec3 c; float r; center_and_radius(mesh, c, r);
    for (auto& v : mesh->vertices) store3(v.position, (load3(v.position) - c) / r);
}

// SimpleMesh::resize, SimpleMesh.h:79-85: scales all four position components.
void rt_mesh_resize(rt_mesh* mesh)
{
    if (!mesh) return;
    const float factor = 2;
    for (auto& v : mesh->vertices) for (int k = 0; k < 4; ++k) v.position[k] *= factor;
}

// SimpleMesh::transform, SimpleMesh.h:122-126.
void rt_mesh_transform(rt_mesh* mesh, const float rot[9], const float translation[3])
{
    if (!mesh) return;
    Mat3 r; memcpy(r.m, rot, sizeof r.m);
    for (auto& v : mesh->vertices) store3(v.position, mul(r, load3(v.position)) + load3(translation));
}

void rt_mesh_copy_vertices(const rt_mesh* mesh, float* positions, float* colors)
{
    if (!mesh) return;
    for (size_t i = 0; i < mesh->vertices.size(); ++i) {
        if (positions) memcpy(positions + 4 * i, mesh->vertices[i].position, 16);
        if (colors) memcpy(colors + 4 * i, mesh->vertices[i].color, 16);
    }
}

void rt_mesh_copy_indices(const rt_mesh* mesh, uint32_t* indices)
{
    if (mesh && indices && !mesh->indices.empty()) memcpy(indices, mesh->indices.data(), mesh->indices.size() * sizeof(uint32_t));
}

// A face may name a vertex that does not exist (yet): addFace takes any index, as SimpleMesh::addFace does
// (SimpleMesh.h:99-104).  The routines that follow the indices check them first.
static bool faces_in_range(const rt_mesh* mesh)
{
    const size_t nv = mesh->vertices.size();
    for (uint32_t i : mesh->indices) if (i >= nv) return false;
    return true;
}

// The loop of src/main.cpp:113-122.
int rt_mesh_to_triangles(const rt_mesh* mesh, rt_triangle* out)
{
    if (!mesh || !out) return RT_ERR_INVALID_ARG;
    if (!faces_in_range(mesh)) return RT_ERR_INVALID_ARG;
    for (size_t t = 0; t < mesh->indices.size() / 3; ++t) {
        const float* v0 = mesh->vertices[mesh->indices[3 * t]].position;
        const float* v1 = mesh->vertices[mesh->indices[3 * t + 1]].position;
        const float* v2 = mesh->vertices[mesh->indices[3 * t + 2]].position;
        rt_make_triangle(&out[t], v0, v1, v2);
    }
    return RT_OK;
}


// Mean of the three vertex colours of each face (Vertex::color, SimpleMesh.h:27).
int rt_mesh_triangle_colors(const rt_mesh* mesh, float* rgb)
{
    if (!mesh || !rgb) return RT_ERR_INVALID_ARG;
    if (!faces_in_range(mesh)) return RT_ERR_INVALID_ARG;
    for (size_t t = 0; t < mesh->indices.size() / 3; ++t) {
        const float* c0 = mesh->vertices[mesh->indices[3 * t]].color;
        const float* c1 = mesh->vertices[mesh->indices[3 * t + 1]].color;
        const float* c2 = mesh->vertices[mesh->indices[3 * t + 2]].color;
        for (int k = 0; k < 3; ++k) rgb[3 * t + k] = ((c0[k] + c1[k]) + c2[k]) / 3.0f;
    }
    return RT_OK;
}

int rt_mesh_create(rt_mesh** out)                                // SimpleMesh(), SimpleMesh.h:42
{
    if (!out) return RT_ERR_INVALID_ARG;
    *out = new (std::nothrow) rt_mesh();
    return *out ? RT_OK : RT_ERR_OUT_OF_MEMORY;
}

void rt_mesh_clear(rt_mesh* mesh)                                // SimpleMesh.h:47-50
{
    if (!mesh) return;
    mesh->vertices.clear();
    mesh->indices.clear();
}

uint32_t rt_mesh_add_vertex(rt_mesh* mesh, const float position[4], const float color[4])   // SimpleMesh.h:93-97
{
    if (!mesh || !position) return 0xffffffffu;
    rt_mesh::Vertex v{};
    memcpy(v.position, position, 16);
    if (color) memcpy(v.color, color, 16);
    try { mesh->vertices.push_back(v); } catch (...) { return 0xffffffffu; }
    return (uint32_t)mesh->vertices.size() - 1u;
}

uint32_t rt_mesh_add_face(rt_mesh* mesh, uint32_t idx0, uint32_t idx1, uint32_t idx2)      // SimpleMesh.h:99-104
{
    if (!mesh) return 0xffffffffu;
    try {
        mesh->indices.reserve(mesh->indices.size() + 3u);
        mesh->indices.push_back(idx0); mesh->indices.push_back(idx1); mesh->indices.push_back(idx2);
    } catch (...) { return 0xffffffffu; }
    return (uint32_t)(mesh->indices.size() / 3) - 1u;
}

void rt_mesh_set_color(rt_mesh* mesh, float r, float g, float b, float a)                  // SimpleMesh.h:87-91
{
    if (!mesh) return;
    for (auto& v : mesh->vertices) { v.color[0] = r; v.color[1] = g; v.color[2] = b; v.color[3] = a; }
}

// SimpleMesh::writeMesh, SimpleMesh.h:217-245.  Numbers are formatted as an ostream's default
// does (%g); colours go out as int(colour), which after loadMesh's division by 255 is 0 or 1.
int rt_mesh_write(const rt_mesh* mesh, const char* path)
{
    if (!mesh || !path) return RT_ERR_INVALID_ARG;
    FILE* f = fopen(path, "w");
    if (!f) return RT_ERR_IO;                                    // :220
    fprintf(f, "COFF\n%zu %zu 0\n", mesh->vertices.size(), mesh->indices.size() / 3);   // :223-224
    for (const auto& v : mesh->vertices) {
        const bool finite = std::isfinite(v.position[0]) && std::isfinite(v.position[1]) && std::isfinite(v.position[2]) &&
                            std::isfinite(v.position[3]);         // :229
        if (finite)
            fprintf(f, "%g %g %g %d %d %d %d\n", v.position[0], v.position[1], v.position[2],
                    (int)v.color[0], (int)v.color[1], (int)v.color[2], (int)v.color[3]);   // :230-231
        else
            fprintf(f, "0.0 0.0 0.0 0 0 0 0\n");                 // :233
    }
    for (size_t t = 0; t < mesh->indices.size() / 3; ++t)
        fprintf(f, "3 %u %u %u\n", mesh->indices[3 * t], mesh->indices[3 * t + 1], mesh->indices[3 * t + 2]);   // :238
    const bool ok = !ferror(f);
    return (fclose(f) == 0 && ok) ? RT_OK : RT_ERR_IO;
}

// SimpleMesh::joinMeshes, SimpleMesh.h:251-288.
int rt_mesh_join(rt_mesh** out, const rt_mesh* mesh1, const rt_mesh* mesh2, const float pose[16])
{
    if (!out || !mesh1 || !mesh2) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    rt_mesh* joined = new (std::nothrow) rt_mesh();
    if (!joined) return RT_ERR_OUT_OF_MEMORY;
    const uint32_t n1 = (uint32_t)mesh1->vertices.size();
    try {
    joined->vertices.reserve(n1 + mesh2->vertices.size());
    for (const auto& v1 : mesh1->vertices) {                     // :270-276
        rt_mesh::Vertex v{};
        if (pose) {
            for (int i = 0; i < 4; ++i)
                v.position[i] = ((pose[i] * v1.position[0] + pose[4 + i] * v1.position[1]) + pose[8 + i] * v1.position[2]) +
                                pose[12 + i] * v1.position[3];
        } else {
            memcpy(v.position, v1.position, 16);
        }
        memcpy(v.color, v1.color, 16);
        joined->vertices.push_back(v);
    }
    for (const auto& v2 : mesh2->vertices) joined->vertices.push_back(v2);                  // :277
    joined->indices = mesh1->indices;                                                       // :280
    for (uint32_t idx : mesh2->indices) joined->indices.push_back(idx + n1);                // :281-285
    } catch (...) { delete joined; return RT_ERR_OUT_OF_MEMORY; }
    *out = joined;
    return RT_OK;
}

// SimpleMesh::sphere, SimpleMesh.h:293-317: an icosahedron; vertex and face order as there, so
// that triangle indices agree.
int rt_mesh_sphere(rt_mesh** out, const float center[3], float scale, const float color[4])
{
    if (!out || !center) return RT_ERR_INVALID_ARG;
    *out = nullptr;
    rt_mesh* mesh = new (std::nothrow) rt_mesh();
    if (!mesh) return RT_ERR_OUT_OF_MEMORY;
    const double X = 0.525731, Z = 0.850651;                     // :298-299
    const double verts[12][3] = { { -X, 0, Z }, { X, 0, Z }, { -X, 0, -Z }, { X, 0, -Z }, { 0, Z, X }, { 0, Z, -X },
                                  { 0, -Z, X }, { 0, -Z, -X }, { Z, X, 0 }, { -Z, X, 0 }, { Z, -X, 0 }, { -Z, -X, 0 } };
    const uint32_t faces[20][3] = { { 0, 4, 1 }, { 0, 9, 4 }, { 9, 5, 4 }, { 4, 5, 8 }, { 4, 8, 1 }, { 8, 10, 1 }, { 8, 3, 10 },
                                    { 5, 3, 8 }, { 5, 2, 3 }, { 2, 7, 3 }, { 7, 10, 3 }, { 7, 6, 10 }, { 7, 11, 6 }, { 11, 0, 6 },
                                    { 0, 1, 6 }, { 6, 1, 10 }, { 9, 0, 11 }, { 9, 11, 2 }, { 9, 2, 5 }, { 7, 2, 11 } };   // :300-301
    const float default_color[4] = { 0, 0, 1, 1 };               // :293
    const float* col = color ? color : default_color;
    try {
    for (int i = 0; i < 12; ++i) {                               // :304-309
        rt_mesh::Vertex v{};
        for (int k = 0; k < 3; ++k) v.position[k] = center[k] + scale * (float)verts[i][k];
        v.position[3] = 1.0f + scale * 0.0f;
        memcpy(v.color, col, 16);
        mesh->vertices.push_back(v);
    }
    for (int i = 0; i < 20; ++i) for (int k = 0; k < 3; ++k) mesh->indices.push_back(faces[i][k]);   // :312-314
    } catch (...) { delete mesh; return RT_ERR_OUT_OF_MEMORY; }
    *out = mesh;
    return RT_OK;
}

}  // extern "C"
