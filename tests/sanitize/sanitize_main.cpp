// sanitize_main.cpp — the oracle (oracle/rt_oracle.c) and the host-side scene code (ray_tracing_amd/csrc/rt_host.cpp)
// driven from one plain program built with -fsanitize=address,undefined (tests/sanitize/Makefile): every routine
// the CPU tests reach through ctypes is run here on small inputs, plus malformed mesh files.  No GPU, no Python:
// a sanitizer runtime cannot be loaded into the test runner's interpreter after the fact.
//   usage: sanitize_main <mesh.off> <scratch directory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
extern "C" {
#include "../../oracle/rt_oracle.h"
}

static int failures = 0;
#define CHECK(x) do { if (!(x)) { std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #x); ++failures; } } while (0)

static void write_text(const std::string& path, const std::string& text)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::perror(path.c_str()); std::exit(2); }
    std::fwrite(text.data(), 1, text.size(), f);
    std::fclose(f);
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s mesh.off scratch_dir\n", argv[0]); return 2; }
    const std::string mesh_path = argv[1], dir = argv[2];

    // ---- host-side scene construction
    float cam[RT_CAMERA_FLOATS];
    const float pos[3] = { 1, 1, 6 };
    rt_camera_init(cam, 64, 48, 0.05f, pos);
    rt_camera_rotate(cam, 64, 48, 0.05f, 10.0f, -5.0f);
    for (int d = 0; d < 4; ++d) rt_camera_move(cam, 64, 48, 0.05f, d);
    rt_camera_init(cam, 64, 48, 0.05f, pos);

    rt_material lam = { { 0.8f, 0.8f, 0.0f }, RT_MAT_LAMBERTIAN, { 0, 0, 0, 0 } };
    rt_material metal = { { 0.9f, 0.9f, 0.9f }, RT_MAT_METAL, { 0, 0.1f, 0, 0 } };
    rt_material light = { { 4, 4, 4 }, RT_MAT_LIGHT, { 0, 0, 0, 0 } };
    std::vector<rt_shape> shapes(5);
    const float c0[3] = { 0, 0, -4 }, c1[3] = { 0, -100.5f, -4 }, c2[3] = { 1, 2, -2 }, c3[3] = { -1.5f, 0.5f, -4 }, c4[3] = { 2, 0, -5 };
    const float sides[3] = { 1, 2, 1 };
    rt_make_circle(&shapes[0], c0, 0.5f, &lam);
    rt_make_circle(&shapes[1], c1, 100.0f, &metal);
    rt_make_circle(&shapes[2], c2, 0.5f, &light);
    rt_make_cube(&shapes[3], c3, &lam);
    rt_cube_rotate(&shapes[3], 0.1f);
    rt_make_box(&shapes[4], c4, sides, &lam);

    rt_mesh* mesh = nullptr;
    CHECK(rt_mesh_load(&mesh, mesh_path.c_str()) == RT_OK && mesh);
    rt_mesh_normalize(mesh);
    rt_mesh_resize(mesh);
    const float rot[9] = { 0.5f, 0, 0, 0, 0, 0.5f, 0, -0.5f, 0 }, tr[3] = { 0.5f, 0.7f, -2 };
    rt_mesh_transform(mesh, rot, tr);
    const uint32_t nv = rt_mesh_vertex_count(mesh), nt = rt_mesh_triangle_count(mesh);
    CHECK(nv > 0 && nt > 0);
    std::vector<float> vp(4 * (size_t)nv), vc(4 * (size_t)nv), tcol(3 * (size_t)nt);
    std::vector<uint32_t> idx(3 * (size_t)nt);
    std::vector<rt_triangle> tris(nt);
    rt_mesh_copy_vertices(mesh, vp.data(), vc.data());
    rt_mesh_copy_indices(mesh, idx.data());
    rt_mesh_to_triangles(mesh, tris.data());
    rt_mesh_triangle_colors(mesh, tcol.data());
    rt_mesh_set_color(mesh, 1, 0, 0, 1);
    CHECK(rt_mesh_write(mesh, (dir + "/copy.off").c_str()) == RT_OK);
    rt_mesh* again = nullptr;
    CHECK(rt_mesh_load(&again, (dir + "/copy.off").c_str()) == RT_OK && rt_mesh_triangle_count(again) == nt);
    rt_mesh *ball = nullptr, *joined = nullptr;
    const float centre[3] = { 0, 1, -3 }, white[4] = { 1, 1, 1, 1 };
    CHECK(rt_mesh_sphere(&ball, centre, 0.5f, white) == RT_OK && rt_mesh_triangle_count(ball) == 20);
    CHECK(rt_mesh_join(&joined, ball, again, nullptr) == RT_OK && rt_mesh_triangle_count(joined) == nt + 20);
    rt_mesh* hand = nullptr;
    CHECK(rt_mesh_create(&hand) == RT_OK);
    const float p0[4] = { 0, 0, 0, 1 }, p1[4] = { 1, 0, 0, 1 }, p2[4] = { 0, 1, 0, 1 };
    rt_mesh_add_vertex(hand, p0, white); rt_mesh_add_vertex(hand, p1, white); rt_mesh_add_vertex(hand, p2, white);
    CHECK(rt_mesh_add_face(hand, 0, 1, 2) == 0);
    CHECK(rt_mesh_add_face(hand, 0, 1, 3) == 1);                               // a face before its vertex: taken, as SimpleMesh::addFace takes it
    std::vector<rt_triangle> two(rt_mesh_triangle_count(hand));
    std::vector<float> two_rgb(3 * two.size());
    CHECK(rt_mesh_to_triangles(hand, two.data()) == RT_ERR_INVALID_ARG);       // ... and refused where the indices are followed
    CHECK(rt_mesh_triangle_colors(hand, two_rgb.data()) == RT_ERR_INVALID_ARG);
    const float p3[4] = { 1, 1, 0, 1 };
    rt_mesh_add_vertex(hand, p3, white);
    CHECK(rt_mesh_to_triangles(hand, two.data()) == RT_OK);
    CHECK(rt_mesh_triangle_colors(hand, two_rgb.data()) == RT_OK);
    rt_mesh_clear(hand);
    // malformed files: each must be refused (or read) without touching memory it does not own
    const char* bad[] = {
        "", "OFF\n", "OFF\n3 1 0\n0 0 0\n1 0 0\n", "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n", "OFF\n4000000000 4000000000 0\n",
        "COFF\n3 1 0\n0 0 0 255 0 0 255\n1 0 0 255 0 0 255\n0 1 0 255 0 0\n3 0 1 2\n", "OFF\n-1 -1 0\n", "PLY\n", "OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n4 0 1 2 0\n",
    };
    for (size_t i = 0; i < sizeof bad / sizeof *bad; ++i) {
        const std::string path = dir + "/bad" + std::to_string(i) + ".off";
        write_text(path, bad[i]);
        rt_mesh* m = nullptr;
        const int rc = rt_mesh_load(&m, path.c_str());
        if (rc == RT_OK) { std::vector<rt_triangle> t(rt_mesh_triangle_count(m)); rt_mesh_to_triangles(m, t.data()); rt_mesh_free(m); }
        else CHECK(m == nullptr);
    }
    rt_mesh* none = nullptr;
    CHECK(rt_mesh_load(&none, (dir + "/does_not_exist.off").c_str()) != RT_OK);
    rt_mesh_free(mesh); rt_mesh_free(again); rt_mesh_free(ball); rt_mesh_free(joined); rt_mesh_free(hand);

    // ---- the oracle: both rule sets, both arithmetics, both generators, every entry point
    static_assert(sizeof(orc_shape) == sizeof(rt_shape) && sizeof(orc_triangle) == sizeof(rt_triangle), "the oracle reads the same layouts");
    const uint32_t W = 48, H = 36;
    for (uint32_t sem = 0; sem < 2; ++sem)
        for (uint32_t arith = 0; arith < 2; ++arith)
            for (uint32_t rng = 0; rng < 2; ++rng) {
                orc_params p;
                std::memset(&p, 0, sizeof p);
                p.width = W; p.height = H; p.spp = 3; p.max_depth = 6; p.grid_n = 1; p.seed = 7; p.semantics = sem;
                p.triangle_albedo[0] = 1.0f; p.color_order = ORC_COLOR_FORWARD; p.unit_sampler = arith ? ORC_UNIT_GAUSS : ORC_UNIT_DIRECT;
                p.shard_count = 1; p.band_rows = 16; p.threads = 2; p.rng = rng; p.arith = arith;
                p.triangle_colors = rng ? tcol.data() : nullptr;
                std::vector<uint32_t> accum(4 * (size_t)W * H, 0u);
                orc_stats st;
                CHECK(orc_render(reinterpret_cast<const orc_shape*>(shapes.data()), (uint32_t)shapes.size(),
                                 reinterpret_cast<const orc_triangle*>(tris.data()), nt, cam, &p, accum.data(), &st) == 0);
                CHECK(st.samples == (uint64_t)W * H * 3 && st.terminal == st.samples);
                std::vector<float> img(4 * (size_t)W * H), mean(4 * (size_t)W * H, 0.0f);
                orc_resolve(accum.data(), W, H, 3, sem, 1, 1, img.data());
                for (float v : img) CHECK(std::isfinite(v));
                orc_convert(accum.data(), mean.data(), W, H, 3, sem, 1);
                std::vector<int32_t> hit((size_t)W * H * p.max_depth);
                std::vector<uint32_t> mat((size_t)W * H * p.max_depth);
                CHECK(orc_trace(reinterpret_cast<const orc_shape*>(shapes.data()), (uint32_t)shapes.size(),
                                reinterpret_cast<const orc_triangle*>(tris.data()), nt, cam, &p, 1, hit.data(), mat.data()) == 0);
                float o[3], d[3];
                orc_camera_ray(cam, &p, W - 1, H - 1, 2, o, d);
            }
    // an empty scene, one pixel, row shards that leave a rank without rows
    {
        orc_params p;
        std::memset(&p, 0, sizeof p);
        p.width = 1; p.height = 1; p.spp = 4; p.max_depth = 2; p.grid_n = 2; p.sample_mode = ORC_SAMPLE_GRID; p.shard_index = 1; p.shard_count = 2; p.band_rows = 16; p.threads = 1;
        uint32_t accum[4] = { 0, 0, 0, 0 };
        orc_stats st;
        CHECK(orc_render(nullptr, 0, nullptr, 0, cam, &p, accum, &st) == 0 && st.samples == 0);
    }
    if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    std::printf("sanitize_main: ok\n");
    return 0;
}
