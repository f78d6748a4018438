// multi_gpu.cpp — one frame split over the GPUs of a node from a C++ host: one thread-less process, one context per
// GPU, RCCL for the single exchange.  The reference is single-GPU (one GL context, src/main.cpp:88-159); this shows
// that a host which keeps the reference's classes (rt_host.hpp) shards a frame with the C ABI alone:
//
//   rows    — GPU r traces the 16-row bands b with b % N == r of every frame (rt_set_row_shard); the exchange
//             brings each GPU's own bands to GPU 0 (ncclSend / ncclRecv inside one group: width*height*16/N bytes
//             per GPU).  The assembled frame is bit-identical to a single-GPU frame.
//   samples — GPU r traces samples [r*spp, (r+1)*spp) of every pixel (first_sample); the exchange is ONE
//             ncclReduce(sum) of the accumulators to GPU 0, a frame of N*spp samples per pixel.
//
// Then rt_set_accum_samples + rt_resolve on GPU 0.  Accumulators are caller-owned device memory bound with
// rt_bind_accum, so that the collective runs on them in place; trace and exchange of a GPU share one HIP stream.
//
//   multi_gpu [--gpus N] [--split rows|samples] [--width W --height H] [--spp S] [--frames F] [--out file.pfm]
//
// With --gpus 1 the collectives still run (a one-rank communicator), which is how this file is tested on a
// one-GPU box; the frame then equals the single-context frame of the same call.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "rt_host.hpp"

using namespace rt;

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); std::exit(1); } } while (0)

static void addScene(RayTracingComputeShader& cs)                                                  // main.cpp:95-106
{
    cs.shapes.push_back(Box({ -1, 12, -2 }, { 10, 1, 1 }, Light{ { 1.5, 1.5, 1.5 } }));
    cs.shapes.push_back(Circle({ 1, 3, -10 }, 1, Lambertian{ { 0, 1, 0 } }));
    cs.shapes.push_back(Circle({ -1.5, 1.5, -4 }, 0.5, Metal{ { 1, 1, 1 } }));
    cs.shapes.push_back(Circle({ 0, 1, -4 }, 0.5, Lambertian{ { 0.8f, 0.8f, 0 } }));
    cs.shapes.push_back(Circle({ 1.5, 1, -3 }, 1.1f, Lambertian{ { 1, 0, 0 } }));
    cs.shapes.push_back(Box({ 0, -50, 0 }, { 100, 1, 1 }, Lambertian{ { 0.3f, 0.9f, 0.7f } }));
    cs.shapes.push_back(Box({ 3, 2, -4.5 }, { 1, 2, 1 }, Lambertian{ { 0, 1, 1 } }));
    cs.shapes.push_back(Circle({ -2, 1, -1 }, 0.5, Lambertian{ { 0.5, 1, 1 } }));
    cs.shapes.push_back(Box({ -2, 1, -6 }, { 0.5, 0.5, 1 }, Lambertian{ { 1, 0, 0.5 } }));
}

int main(int argc, char** argv)
{
    int n_gpus = 1;
    unsigned width = 1080, height = 720, spp = 100, frames = 2, depth = 10;
    const unsigned band = 16;
    std::string split = "rows", out;
    for (int i = 1; i < argc; ++i) {
        auto next = [&] { if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", argv[i]); std::exit(2); } return argv[++i]; };
        if (!std::strcmp(argv[i], "--gpus")) n_gpus = std::atoi(next());
        else if (!std::strcmp(argv[i], "--split")) split = next();
        else if (!std::strcmp(argv[i], "--width")) width = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) height = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--spp")) spp = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--frames")) frames = (unsigned)std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) out = next();
        else { std::fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    const bool rows = split == "rows";
    if (!rows && split != "samples") { std::fprintf(stderr, "--split rows|samples\n"); return 2; }
    int have = 0;
    HIP_OK(hipGetDeviceCount(&have));
    if (n_gpus < 1 || n_gpus > have) { std::fprintf(stderr, "%d GPUs asked for, %d present\n", n_gpus, have); return 1; }

    try {
        std::vector<int> devices(n_gpus);
        for (int r = 0; r < n_gpus; ++r) devices[r] = r;
        std::vector<ncclComm_t> comms(n_gpus);
        NCCL_OK(ncclCommInitAll(comms.data(), n_gpus, devices.data()));

        const size_t words = (size_t)width * height * 4;
        const size_t row_words = (size_t)width * 4;
        Camera camera{ width, height };                                                           // main.cpp:88
        std::vector<std::unique_ptr<RayTracingComputeShader>> cs(n_gpus);
        std::vector<hipStream_t> streams(n_gpus);
        std::vector<uint32_t*> accum(n_gpus);
        for (int r = 0; r < n_gpus; ++r) {
            HIP_OK(hipSetDevice(r));
            HIP_OK(hipStreamCreate(&streams[r]));
            HIP_OK(hipMalloc(&accum[r], words * sizeof(uint32_t)));
            cs[r].reset(new RayTracingComputeShader(camera, r, RT_SEM_GLSL));
            addScene(*cs[r]);
            cs[r]->updateShapeBuffer();                                                           // main.cpp:124
            cs[r]->updateCameraBuffer();
            if (rt_set_stream(cs[r]->handle(), streams[r]) != RT_OK) throw std::runtime_error(rt_last_error(cs[r]->handle()));
            if (rt_bind_accum(cs[r]->handle(), accum[r], width, height) != RT_OK) throw std::runtime_error(rt_last_error(cs[r]->handle()));
            if (rows && rt_set_row_shard(cs[r]->handle(), (uint32_t)r, (uint32_t)n_gpus, band) != RT_OK) throw std::runtime_error(rt_last_error(cs[r]->handle()));
        }

        std::vector<float> pixels(words);
        const unsigned n_bands = (height + band - 1) / band;
        double total_ms = 0;
        for (unsigned frame = 0; frame < frames; ++frame) {
            auto t1 = std::chrono::high_resolution_clock::now();
            for (int r = 0; r < n_gpus; ++r) {                       // every GPU: clear, trace its share (asynchronous)
                HIP_OK(hipSetDevice(r));
                if (rt_reset_accum(cs[r]->handle()) != RT_OK) throw std::runtime_error(rt_last_error(cs[r]->handle()));
                const uint32_t first = rows ? 0u : (uint32_t)r * spp;
                if (rt_render(cs[r]->handle(), width, height, spp, depth, /*seed*/ 1 + frame, first, 0) != RT_OK)
                    throw std::runtime_error(rt_last_error(cs[r]->handle()));
            }
            NCCL_OK(ncclGroupStart());                               // the one exchange, on the streams the traces are on
            for (int r = 0; r < n_gpus; ++r) {
                if (!rows) {
                    NCCL_OK(ncclReduce(accum[r], accum[r], words, ncclUint32, ncclSum, 0, comms[r], streams[r]));
                    continue;
                }
                for (unsigned b = 0; b < n_bands; ++b) {             // band b belongs to GPU b % N and goes to GPU 0
                    const int owner = (int)(b % (unsigned)n_gpus);
                    if (owner == 0) continue;
                    const unsigned y0 = b * band, y1 = std::min(height, y0 + band);
                    const size_t at = (size_t)y0 * row_words, n = (size_t)(y1 - y0) * row_words;
                    if (r == owner) NCCL_OK(ncclSend(accum[r] + at, n, ncclUint32, 0, comms[r], streams[r]));
                    if (r == 0) NCCL_OK(ncclRecv(accum[0] + at, n, ncclUint32, owner, comms[0], streams[0]));
                }
            }
            NCCL_OK(ncclGroupEnd());
            HIP_OK(hipSetDevice(0));
            if (rt_set_accum_samples(cs[0]->handle(), rows ? spp : spp * (unsigned)n_gpus) != RT_OK) throw std::runtime_error(rt_last_error(cs[0]->handle()));
            cs[0]->resolve(pixels.data());                           // waits for GPU 0's stream
            for (int r = 1; r < n_gpus; ++r) { HIP_OK(hipSetDevice(r)); HIP_OK(hipStreamSynchronize(streams[r])); }
            auto t2 = std::chrono::high_resolution_clock::now();
            const double ms = std::chrono::duration<double, std::milli>(t2 - t1).count();
            if (frame > 0) total_ms += ms;
            std::printf("frame %u: %.3f ms\n", frame, ms);
        }
        double mean = 0;
        for (size_t i = 0; i < pixels.size(); i += 4) mean += pixels[i] + pixels[i + 1] + pixels[i + 2];
        std::printf("%d GPU(s), split %s, %ux%u, %u spp per GPU: mean %.6f, %.3f ms per frame after the first\n", n_gpus, split.c_str(),
                    width, height, spp, mean / (3.0 * width * height), frames > 1 ? total_ms / (frames - 1) : 0.0);
        if (!out.empty()) {                                          // rows bottom to top, as a PFM file wants them
            FILE* f = std::fopen(out.c_str(), "wb");
            if (!f) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
            std::fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
            for (size_t i = 0; i < pixels.size(); i += 4) std::fwrite(&pixels[i], sizeof(float), 3, f);
            std::fclose(f);
        }
        for (int r = 0; r < n_gpus; ++r) {
            HIP_OK(hipSetDevice(r));
            rt_bind_accum(cs[r]->handle(), nullptr, 0, 0);
            cs[r].reset();
            HIP_OK(hipFree(accum[r]));
            HIP_OK(hipStreamDestroy(streams[r]));
            NCCL_OK(ncclCommDestroy(comms[r]));
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
