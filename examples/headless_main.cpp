// headless_main.cpp — src/main.cpp:88-159 of jaafer06/ray_tracing without the window: the same scene
// list, the same call order (updateShapeBuffer once; per frame updateCameraBuffer, compute,
// converter), driving the HIP core through rt_host.hpp.  The running-mean image is written as a PFM
// file instead of being blitted.
//
//   headless_main [--mesh path.off] [--frames N] [--width W --height H] [--cpu-rules] [--out file.pfm]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "rt_host.hpp"

using namespace rt;

static bool writePFM(const std::string& path, const std::vector<float>& rgba, unsigned int width, unsigned int height)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    // little-endian; a PFM file holds its scanlines bottom to top, and so does the image read back from the
    // library (row 0 is the bottom of the view, the order of the reference's texture): rows go out in index order
    std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    std::vector<float> row(3 * (size_t)width);
    for (unsigned int y = 0; y < height; ++y) {
        const float* src = &rgba[4 * (size_t)y * width];
        for (unsigned int x = 0; x < width; ++x) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
        std::fwrite(row.data(), sizeof(float), row.size(), f);
    }
    std::fclose(f);
    return true;
}

int main(int argc, char** argv)
{
    unsigned int width = 1080, height = 720;                        // main.cpp:38-39
    unsigned int frames = 4;
    std::string meshPath, out = "frame.pfm";
    bool cpuRules = false;
    for (int i = 1; i < argc; ++i) {
        auto next = [&] { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--mesh")) meshPath = next();
        else if (!std::strcmp(argv[i], "--frames")) frames = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--width")) width = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) height = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else if (!std::strcmp(argv[i], "--cpu-rules")) cpuRules = true;
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    try {
        Camera camera{ width, height };                              // main.cpp:88
        RayTracingComputeShader computeShader{ camera, 0, cpuRules ? RT_SEM_CPU : RT_SEM_GLSL };   // main.cpp:93
        computeShader.shapes.push_back(Box({ -1, 12, -2 }, { 10, 1, 1 }, Light{ { 1.5, 1.5, 1.5 } }));           // :95
        computeShader.shapes.push_back(Circle({ 1, 3, -10 }, 1, Lambertian{ { 0, 1, 0 } }));                       // :98
        computeShader.shapes.push_back(Circle({ -1.5, 1.5, -4 }, 0.5, Metal{ { 1, 1, 1 } }));                     // :99
        computeShader.shapes.push_back(Circle({ 0, 1, -4 }, 0.5, Lambertian{ { 0.8f, 0.8f, 0 } }));               // :100
        computeShader.shapes.push_back(Circle({ 1.5, 1, -3 }, 1.1f, Lambertian{ { 1, 0, 0 } }));                  // :102
        computeShader.shapes.push_back(Box({ 0, -50, 0 }, { 100, 1, 1 }, Lambertian{ { 0.3f, 0.9f, 0.7f } }));    // :103
        computeShader.shapes.push_back(Box({ 3, 2, -4.5 }, { 1, 2, 1 }, Lambertian{ { 0, 1, 1 } }));              // :104
        computeShader.shapes.push_back(Circle({ -2, 1, -1 }, 0.5, Lambertian{ { 0.5, 1, 1 } }));                  // :105
        computeShader.shapes.push_back(Box({ -2, 1, -6 }, { 0.5, 0.5, 1 }, Lambertian{ { 1, 0, 0.5 } }));         // :106

        if (!meshPath.empty()) {                                     // main.cpp:108-122
            SimpleMesh mesh;
            if (!mesh.loadMesh(meshPath)) { std::fprintf(stderr, "cannot load %s\n", meshPath.c_str()); return 1; }
            mesh.normalize();
            mesh.resize();
            const float angle = (float)(-3.14159265358979323846 / 2);
            const float c = 0.5f * std::cos(angle), s = 0.5f * std::sin(angle);
            mesh.transform({ 0.5f, 0, 0, 0, c, -s, 0, s, c }, { 0.5f, 0.7f, -2 });
            computeShader.triangles = mesh.toSimpleTriangles();
        }
        computeShader.updateShapeBuffer();                           // main.cpp:124

        std::vector<float> pixels(4 * (size_t)width * height);
        unsigned int number_of_samples = 1;                          // main.cpp:42
        for (unsigned int frame = 0; frame < frames; ++frame) {     // main.cpp:135
            auto t1 = std::chrono::high_resolution_clock::now();
            computeShader.updateCameraBuffer();                      // :143
            computeShader.compute(width, height, 1);                 // :145
            computeShader.convert(number_of_samples);                // :147-149
            ++number_of_samples;                                     // :151
            computeShader.readImage(pixels.data());                  // stands in for the blit of :152
            auto t2 = std::chrono::high_resolution_clock::now();
            std::chrono::duration<double, std::milli> ms_double = t2 - t1;
            std::cout << ms_double.count() << "ms" << std::endl;     // :158
        }
        if (!writePFM(out, pixels, width, height)) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
        double mean = 0;
        for (size_t i = 0; i < pixels.size(); i += 4) mean += pixels[i] + pixels[i + 1] + pixels[i + 2];
        std::printf("%u frame(s) of %ux%u, %u shapes + %zu triangles -> %s (mean %.6f)\n", frames, width, height,
                    (unsigned)computeShader.shapes.size(), computeShader.triangles.size(), out.c_str(), mean / (3.0 * width * height));
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
