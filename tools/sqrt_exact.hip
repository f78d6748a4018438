// sqrt_exact.hip — which short instruction sequences give the correctly rounded square root / reciprocal
// on this chip for EVERY binary32 input?  Exhaustive: all 2^31 non-negative bit patterns (sqrt) and all
// 2^32 patterns (reciprocal) are run through each candidate and compared with the IEEE result
// ((float)sqrt((double)x) is the correctly rounded binary32 root: 53 >= 2*24 + 2).
// Prints, per candidate, the number of inputs whose result differs and the first few of them.
//   hipcc -O2 -ffp-contract=off --offload-arch=gfx950 -o tools/sqrt_exact tools/sqrt_exact.hip && tools/sqrt_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

#define DEV static __device__ __forceinline__
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// v_mul_legacy_f32: a product in which 0 times anything (infinity, NaN) is 0
DEV float mul_legacy(float a, float b) { float r; asm("s_nop 0\n\tv_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// today's recipe: v_sqrt_f32 and a residual test against each neighbour
DEV float sqrt_neighbours(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    float rm = fma_(-sm, s, x), rp = fma_(-sp, s, x);
    s = (rm <= 0.0f) ? sm : s;
    s = (rp > 0.0f) ? sp : s;
    return s;
}
// A: one transcendental.  y = rsq(x); g = x*y; s = g + (x - g*g) * y/2; the products with y are "legacy"
// multiplies (0 * anything = 0), which carry x = 0 (y = inf) through to s = 0.
DEV float sqrt_rsq(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = mul_legacy(x, y);
    float h = 0.5f * y;
    float r = fma_(-g, g, x);
    return g + mul_legacy(r, h);
}
// A2: the same with a fused last step (no zero handling)
DEV float sqrt_rsq_fused(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float r = fma_(-g, g, x);
    return fma_(r, h, g);
}
// B: two transcendentals: g = sqrt(x), h = rsq(x)/2
DEV float sqrt_sqrt_rsq(float x)
{
    float g = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rsqf(x);
    float r = fma_(-g, g, x);
    return g + mul_legacy(r, h);
}
// reciprocal: y0 = rcp(x), one fused Newton step
DEV float rcp_newton(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = fma_(-x, y, 1.0f);
    return fma_(y, e, y);
}
DEV float rcp_newton2(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    float e = fma_(-x, y, 1.0f);
    y = fma_(y, e, y);
    e = fma_(-x, y, 1.0f);
    return fma_(y, e, y);
}

// 1 / sqrt_rn(q), seeded with the rsq(q) the square root itself starts from: one fused Newton step against s
DEV float inv_of_root(float q, float& s_out)
{
    float y = __builtin_amdgcn_rsqf(q);
    float g = mul_legacy(q, y);
    float h = 0.5f * y;
    float r = fma_(-g, g, q);
    float s = g + mul_legacy(r, h);
    float e = fma_(-s, y, 1.0f);
    s_out = s;
    return fma_(y, e, y);
}
// a / b from the correctly rounded reciprocal: q = a*y, one fused correction
DEV float div_newton(float a, float b)
{
    float y = rcp_newton(b);
    float q = a * y;
    float r = fma_(-b, q, a);
    return fma_(r, y, q);
}
DEV float div_fixup(float q, float b, float a) { float r; asm("v_div_fixup_f32 %0, %1, %2, %3" : "=v"(r) : "v"(q), "v"(b), "v"(a)); return r; }

struct Tally { unsigned long long bad; uint32_t first[8]; };

template <int WHICH>
__global__ void sweep_sqrt(Tally* out, uint32_t lo, uint32_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = (float)sqrt((double)x);
        const float got = WHICH == 0 ? sqrt_neighbours(x) : WHICH == 1 ? sqrt_rsq(x) : WHICH == 2 ? sqrt_rsq_fused(x) : sqrt_sqrt_rsq(x);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const unsigned long long at = atomicAdd(&out->bad, 1ull);
            if (at < 8) out->first[at] = (uint32_t)b;
        }
    }
}
template <int WHICH>
__global__ void sweep_rcp(Tally* out, uint32_t lo, uint32_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = (float)(1.0 / (double)x);               // correctly rounded: the quotient of two binary32 numbers in double, rounded once more, is safe (53 >= 2*24 + 2)
        const float got = WHICH == 0 ? rcp_newton(x) : rcp_newton2(x);
        if (__float_as_uint(want) != __float_as_uint(got) && want == want) {
            const unsigned long long at = atomicAdd(&out->bad, 1ull);
            if (at < 8) out->first[at] = (uint32_t)b;
        }
    }
}

__global__ void sweep_inv_root(Tally* out, uint32_t lo, uint32_t hi)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        float s;
        const float got = inv_of_root(x, s);
        const float want = (float)(1.0 / (double)(float)sqrt((double)x));
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const unsigned long long at = atomicAdd(&out->bad, 1ull);
            if (at < 8) out->first[at] = (uint32_t)b;
        }
    }
}
// pairs (a, b): b walks all mantissas of [1, 2) times a few exponents, a comes from a hash of (b, k)
template <int FIX>
__global__ void sweep_div(Tally* out, uint32_t rounds)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 23) * rounds; i += stride) {
        uint32_t m = (uint32_t)(i & 0x7fffffu), k = (uint32_t)(i >> 23);
        uint32_t h = m * 0x9e3779b9u + k * 0x85ebca6bu; h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
        const uint32_t eb = 97u + (h >> 27) * 2u;                    // exponent of b: 2^-30 .. 2^32
        const float b = __uint_as_float((eb << 23) | m | ((h & 1u) << 31));
        uint32_t h2 = h * 0xadb4a92du; h2 ^= h2 >> 15;
        const uint32_t ea = 97u + ((h2 >> 8) & 63u);                 // exponent of a: 2^-30 .. 2^33
        const float a = __uint_as_float((ea << 23) | (h2 >> 9) | ((h2 & 2u) << 30));
        float got = div_newton(a, b);
        if (FIX) got = div_fixup(got, b, a);
        const float want = (float)((double)a / (double)b);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const unsigned long long at = atomicAdd(&out->bad, 1ull);
            if (at < 8) out->first[at] = m;
        }
    }
}
__global__ void specials(float* out)
{
    const float in[12] = { 0.0f, -0.0f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1e-40f, -1e-40f, 3e38f, 1.1754944e-38f, 2e-38f, 1.0f, -3.0f };
    const int i = threadIdx.x;
    if (i < 12) {
        const float x = in[i];
        out[4 * i] = rcp_newton(x);
        out[4 * i + 1] = div_fixup(rcp_newton(x), x, 1.0f);
        out[4 * i + 2] = div_fixup(div_newton(2.5f, x), x, 2.5f);
        out[4 * i + 3] = div_newton(2.5f, x);
    }
}

template <typename K>
static void report(const char* name, K launch, const char* range)
{
    Tally* d; Tally h = {};
    (void)hipMalloc(&d, sizeof(Tally)); (void)hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice);
    launch(d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%-22s %-34s differing inputs: %llu", name, range, h.bad);
    for (int i = 0; i < 8 && i < (int)h.bad; ++i) printf(" %08x", h.first[i]);
    printf("\n"); fflush(stdout);
    (void)hipFree(d);
}

int main()
{
    const dim3 grid(256 * 32), block(256);
    // 2^-96 (0x0f800000) .. FLT_MAX: what the tracer's callers can pass (below 2^-96 they accept any tiny result), and zero.
    struct { uint32_t lo, hi; const char* name; } ranges[] = {
        { 0x0f800000u, 0x7f7fffffu, "[2^-96, FLT_MAX]" }, { 0u, 0u, "zero" }, { 0x00800000u, 0x0f7fffffu, "[2^-126, 2^-96)" }, { 1u, 0x007fffffu, "subnormal" } };
    for (auto& r : ranges) {
        report("sqrt neighbours", [&](Tally* d) { hipLaunchKernelGGL(sweep_sqrt<0>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
        report("sqrt rsq+legacy", [&](Tally* d) { hipLaunchKernelGGL(sweep_sqrt<1>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
        report("sqrt rsq fused", [&](Tally* d) { hipLaunchKernelGGL(sweep_sqrt<2>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
        report("sqrt sqrt+rsq", [&](Tally* d) { hipLaunchKernelGGL(sweep_sqrt<3>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
    }
    struct { uint32_t lo, hi; const char* name; } rr[] = { { 0x00800000u, 0x7e7fffffu, "positive normal, 1/x normal" }, { 0x80800000u, 0xfe7fffffu, "negative normal, 1/x normal" } };
    for (auto& r : rr) {
        report("rcp + 1 Newton", [&](Tally* d) { hipLaunchKernelGGL(sweep_rcp<0>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
        report("rcp + 2 Newton", [&](Tally* d) { hipLaunchKernelGGL(sweep_rcp<1>, grid, block, 0, 0, d, r.lo, r.hi); }, r.name);
    }
    report("1/sqrt_rn from rsq", [&](Tally* d) { hipLaunchKernelGGL(sweep_inv_root, grid, block, 0, 0, d, 0x0f800000u, 0x6f000000u); }, "[2^-96, 2^95]");
    report("a/b Newton", [&](Tally* d) { hipLaunchKernelGGL(sweep_div<0>, grid, block, 0, 0, d, 1024u); }, "2^33 pairs");
    report("a/b Newton + fixup", [&](Tally* d) { hipLaunchKernelGGL(sweep_div<1>, grid, block, 0, 0, d, 1024u); }, "2^33 pairs");
    {
        float* d; float h[48];
        (void)hipMalloc(&d, sizeof h);
        hipLaunchKernelGGL(specials, dim3(1), dim3(64), 0, 0, d);
        (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        const char* names[12] = { "0", "-0", "inf", "-inf", "nan", "1e-40", "-1e-40", "3e38", "FLT_MIN", "2e-38", "1", "-3" };
        for (int i = 0; i < 12; ++i)
            printf("x = %-8s rcp_newton %-14g + fixup %-14g   2.5/x newton+fixup %-14g newton %-14g\n", names[i], h[4 * i], h[4 * i + 1], h[4 * i + 2], h[4 * i + 3]);
        (void)hipFree(d);
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
