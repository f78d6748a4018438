// valu_rates.hip — issue-rate microbenchmark for the vector instructions the tracer leans on.
// Each wave runs LOOPS iterations of 32 independent copies of one instruction; the figure
// printed is SIMD cycles per wave-instruction at full occupancy (nominal 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define LOOPS 4096

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY32(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEFINE_KERNEL(NAME, DECL, STMT)                                            \
    __global__ __launch_bounds__(64) void NAME(float* out, float seed) {            \
        DECL                                                                        \
        for (int it = 0; it < LOOPS; ++it) { BODY32(STMT) }                         \
        float s = 0; for (int k = 0; k < 8; ++k) s += __uint_as_float(r[k]);        \
        if (s == 123.456f) out[threadIdx.x] = s;                                    \
    }

#define DECL_F uint32_t r[8]; float a = seed, b = seed * 0.5f; for (int k = 0; k < 8; ++k) r[k] = __float_as_uint(seed + k + threadIdx.x);
#define S_FMA(k)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_MUL(k)   asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_ADD(k)   asm volatile("v_add_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_MULLO(k) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_MULHI(k) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_MUL24(k) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_MAD24(k) asm volatile("v_mad_u32_u24 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_SQRT(k)  asm volatile("v_sqrt_f32 %0, %0" : "+v"(r[k]));
#define S_RCP(k)   asm volatile("v_rcp_f32 %0, %0" : "+v"(r[k]));
#define S_RSQ(k)   asm volatile("v_rsq_f32 %0, %0" : "+v"(r[k]));
#define S_XOR(k)   asm volatile("v_xor_b32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_ADDU(k)  asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_LSHR(k)  asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(r[k]));
#define S_CVT(k)   asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(r[k]));
#define S_CND(k)   asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[k]) : "v"(a) : );
#define S_CMP(k)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[k]), "v"(a) : "vcc");
#define S_CMPS(k)  asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(r[k]), "v"(a) : "s20", "s21");
#define S_MAD64(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(a), "v"(b) : "vcc");
#define S_SQRTFMA(k) asm volatile("v_sqrt_f32 %0, %0\n v_fma_f32 %1, %2, %3, %1\n v_fma_f32 %4, %2, %3, %4\n v_fma_f32 %5, %2, %3, %5" : "+v"(r[k]), "+v"(r[(k+1)&7]), "+v"(r[(k+2)&7]), "+v"(r[(k+3)&7]) : "v"(a), "v"(b));
#define S_DIV(k)   { float x = __uint_as_float(r[k]); x = a / x; r[k] = __float_as_uint(x); }
#define S_SQRTF(k) { float x = __uint_as_float(r[k]); x = __builtin_sqrtf(x); r[k] = __float_as_uint(x); }

DEFINE_KERNEL(k_fma, DECL_F, S_FMA)
DEFINE_KERNEL(k_mul, DECL_F, S_MUL)
DEFINE_KERNEL(k_add, DECL_F, S_ADD)
DEFINE_KERNEL(k_mullo, DECL_F, S_MULLO)
DEFINE_KERNEL(k_mulhi, DECL_F, S_MULHI)
DEFINE_KERNEL(k_mul24, DECL_F, S_MUL24)
DEFINE_KERNEL(k_mad24, DECL_F, S_MAD24)
DEFINE_KERNEL(k_sqrt, DECL_F, S_SQRT)
DEFINE_KERNEL(k_rcp, DECL_F, S_RCP)
DEFINE_KERNEL(k_rsq, DECL_F, S_RSQ)
DEFINE_KERNEL(k_xor, DECL_F, S_XOR)
DEFINE_KERNEL(k_addu, DECL_F, S_ADDU)
DEFINE_KERNEL(k_lshr, DECL_F, S_LSHR)
DEFINE_KERNEL(k_cvt, DECL_F, S_CVT)
DEFINE_KERNEL(k_cnd, DECL_F, S_CND)
DEFINE_KERNEL(k_cmp, DECL_F, S_CMP)
DEFINE_KERNEL(k_cmps, DECL_F, S_CMPS)
DEFINE_KERNEL(k_sqrt3fma, DECL_F, S_SQRTFMA)
DEFINE_KERNEL(k_div_ieee, DECL_F, S_DIV)
DEFINE_KERNEL(k_sqrt_ieee, DECL_F, S_SQRTF)


#define S_CND64(k) asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(r[k]) : "v"(a) : );
#define S_CNDW(k)  asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
#define S_CNDW2(k) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n s_nop 1\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
#define S_MIN(k)   asm volatile("v_min_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_MED3(k)  asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_MAX3(k)  asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_FMAS(k)  asm volatile("v_fma_f32 %0, s20, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_FMAK(k)  asm volatile("v_fmaak_f32 %0, %1, %0, 0x3fc00000" : "+v"(r[k]) : "v"(a));
#define S_RDLANE(k) asm volatile("v_readlane_b32 s20, %0, 5" : : "v"(r[k]) : "s20");
#define S_RDFIRST(k) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(r[k]) : "s20");
#define S_CVTU(k)  asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(r[k]));
#define S_FRACT(k) asm volatile("v_fract_f32 %0, %0" : "+v"(r[k]));
#define S_FLOOR(k) asm volatile("v_floor_f32 %0, %0" : "+v"(r[k]));
#define S_BFE(k)   asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(r[k]));
#define S_XAD(k)   asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_LSHLADD(k) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(r[k]) : "v"(a));
#define S_ALIGNBIT(k) asm volatile("v_alignbit_b32 %0, %0, %0, 13" : "+v"(r[k]));
#define S_ANDOR(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_MOVDPP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r[k]));
#define S_CMPX(k)  asm volatile("v_cmpx_lt_f32 %0, %1" : : "v"(a), "v"(r[k]) : "exec");
#define S_MADU16(k) asm volatile("v_mad_u32_u16 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_SALU(k)  asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");
#define S_FMASALU(k) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_add_u32 s20, s20, 1" : "+v"(r[k]) : "v"(a), "v"(b) : "s20", "scc");
#define S_FMACMP(k) asm volatile("v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %3, %1, %2, %3\n v_cmp_lt_f32 vcc, %0, %1" : "+v"(r[k]), "+v"(r[(k+1)&7]) : "v"(a), "v"(b) : "vcc");
#define S_LOG(k)   asm volatile("v_log_f32 %0, %0" : "+v"(r[k]));
#define S_SIN(k)   asm volatile("v_sin_f32 %0, %0" : "+v"(r[k]));
DEFINE_KERNEL(k_cnd64, DECL_F, S_CND64)
DEFINE_KERNEL(k_cndw, DECL_F, S_CNDW)
DEFINE_KERNEL(k_cndw_nop, DECL_F, S_CNDW2)
DEFINE_KERNEL(k_min, DECL_F, S_MIN)
DEFINE_KERNEL(k_med3, DECL_F, S_MED3)
DEFINE_KERNEL(k_max3, DECL_F, S_MAX3)
DEFINE_KERNEL(k_fma_sgpr, DECL_F, S_FMAS)
DEFINE_KERNEL(k_fmaak, DECL_F, S_FMAK)
DEFINE_KERNEL(k_readlane, DECL_F, S_RDLANE)
DEFINE_KERNEL(k_readfirst, DECL_F, S_RDFIRST)
DEFINE_KERNEL(k_cvt_u32, DECL_F, S_CVTU)
DEFINE_KERNEL(k_fract, DECL_F, S_FRACT)
DEFINE_KERNEL(k_floor, DECL_F, S_FLOOR)
DEFINE_KERNEL(k_bfe, DECL_F, S_BFE)
DEFINE_KERNEL(k_xad, DECL_F, S_XAD)
DEFINE_KERNEL(k_lshladd, DECL_F, S_LSHLADD)
DEFINE_KERNEL(k_alignbit, DECL_F, S_ALIGNBIT)
DEFINE_KERNEL(k_andor, DECL_F, S_ANDOR)
DEFINE_KERNEL(k_movdpp, DECL_F, S_MOVDPP)
DEFINE_KERNEL(k_madu16, DECL_F, S_MADU16)
DEFINE_KERNEL(k_salu, DECL_F, S_SALU)
DEFINE_KERNEL(k_fma_salu, DECL_F, S_FMASALU)
DEFINE_KERNEL(k_2fma_cmp, DECL_F, S_FMACMP)
DEFINE_KERNEL(k_log, DECL_F, S_LOG)
DEFINE_KERNEL(k_sin, DECL_F, S_SIN)

#define S_SUB(k)   asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_FMAC(k)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_FMANEG(k) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_FMAABS(k) asm volatile("v_fma_f32 %0, |%1|, %2, -%0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_AND(k)   asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_OR(k)    asm volatile("v_or_b32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_LSHL(k)  asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r[k]));
#define S_ASHR(k)  asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(r[k]));
#define S_MOV(k)   asm volatile("v_mov_b32 %0, %1" : "=v"(r[k]) : "v"(a));
#define S_MAX(k)   asm volatile("v_max_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_SUBU(k)  asm volatile("v_sub_u32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_XORSDWA(k) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r[k]));
#define S_MULLIT(k) asm volatile("v_mul_f32 %0, 0x3fc90fdb, %0" : "+v"(r[k]));
#define S_ADDSGPR(k) asm volatile("v_add_f32 %0, s20, %0" : "+v"(r[k]));
#define S_CMPU(k)  asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(r[k]), "v"(a) : "vcc");
#define S_3CMP1CND(k) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_fma_f32 %3, %1, %2, %3\n v_fma_f32 %4, %1, %2, %4\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(r[k]), "+v"(r[(k+1)&7]), "+v"(r[(k+2)&7]) : "v"(a), "v"(b) : "vcc");
#define S_MIN3(k)  asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_MINU(k)  asm volatile("v_min_u32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
#define S_RNDNE(k) asm volatile("v_rndne_f32 %0, %0" : "+v"(r[k]));
#define S_LDEXP(k) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a));
#define S_CVTI(k)  asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[k]));
#define S_MADI24(k) asm volatile("v_mad_i32_i24 %0, %1, %0, %2" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_ADDCO(k) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(r[k]) : "v"(a) : "vcc");
#define S_BFI(k)   asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
#define S_FMA4DEP(k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[0]) : "v"(a), "v"(b));
DEFINE_KERNEL(k_sub, DECL_F, S_SUB)
DEFINE_KERNEL(k_fmac, DECL_F, S_FMAC)
DEFINE_KERNEL(k_fma_neg, DECL_F, S_FMANEG)
DEFINE_KERNEL(k_fma_absneg, DECL_F, S_FMAABS)
DEFINE_KERNEL(k_and, DECL_F, S_AND)
DEFINE_KERNEL(k_or, DECL_F, S_OR)
DEFINE_KERNEL(k_lshl, DECL_F, S_LSHL)
DEFINE_KERNEL(k_ashr, DECL_F, S_ASHR)
DEFINE_KERNEL(k_mov, DECL_F, S_MOV)
DEFINE_KERNEL(k_max, DECL_F, S_MAX)
DEFINE_KERNEL(k_subu, DECL_F, S_SUBU)
DEFINE_KERNEL(k_xor_sdwa, DECL_F, S_XORSDWA)
DEFINE_KERNEL(k_mul_lit, DECL_F, S_MULLIT)
DEFINE_KERNEL(k_add_sgpr, DECL_F, S_ADDSGPR)
DEFINE_KERNEL(k_cmp_u32, DECL_F, S_CMPU)
DEFINE_KERNEL(k_cmp2fma_cnd, DECL_F, S_3CMP1CND)
DEFINE_KERNEL(k_min3, DECL_F, S_MIN3)
DEFINE_KERNEL(k_min_u32, DECL_F, S_MINU)
DEFINE_KERNEL(k_rndne, DECL_F, S_RNDNE)
DEFINE_KERNEL(k_ldexp, DECL_F, S_LDEXP)
DEFINE_KERNEL(k_cvt_i32, DECL_F, S_CVTI)
DEFINE_KERNEL(k_mad_i24, DECL_F, S_MADI24)
DEFINE_KERNEL(k_add_co, DECL_F, S_ADDCO)
DEFINE_KERNEL(k_bfi, DECL_F, S_BFI)
DEFINE_KERNEL(k_fma_dep1, DECL_F, S_FMA4DEP)
// packed fp32: 64-bit register pairs
__global__ __launch_bounds__(64) void k_pkfma(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[8], a = {seed, seed}, b = {seed * 0.5f, seed};
    for (int k = 0; k < 8; ++k) r[k] = f2{seed + k, seed - k};
    for (int it = 0; it < LOOPS; ++it) {
#define S_PK(k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b));
        BODY32(S_PK)
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += r[k].x + r[k].y;
    if (s == 123.456f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_pkmul(float* out, float seed) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 r[8], a = {seed, seed};
    for (int k = 0; k < 8; ++k) r[k] = f2{seed + k, seed - k};
    for (int it = 0; it < LOOPS; ++it) {
#define S_PKM(k) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r[k]) : "v"(a));
        BODY32(S_PKM)
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += r[k].x + r[k].y;
    if (s == 123.456f) out[threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_mad64(float* out, float seed) {
    unsigned long long q[8]; uint32_t a = (uint32_t)seed, b = a * 3u;
    for (int k = 0; k < 8; ++k) q[k] = k + threadIdx.x;
    for (int it = 0; it < LOOPS; ++it) { BODY32(S_MAD64) }
    unsigned long long s = 0; for (int k = 0; k < 8; ++k) s += q[k];
    if (s == 123456ull) out[threadIdx.x] = (float)s;
}
// uniform-address LDS reads (what a broadcast scene fetch costs)
__global__ __launch_bounds__(64) void k_ldsb128(float* out, float seed) {
    __shared__ float4 tab[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = make_float4(seed, i, 1, 2);
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    for (int it = 0; it < LOOPS; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            float4 v = tab[(it * 32 + k) & 1023];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[threadIdx.x] = acc.x;
}
// scalar loads from an 18.7 KB table (a 292-triangle mesh) — 64 B per load
__global__ __launch_bounds__(64) void k_smem64(float* out, const float* __restrict__ tab, int n16) {
    float acc = threadIdx.x;
    for (int it = 0; it < LOOPS / 8; ++it) {
        for (int k = 0; k < n16; ++k) {
            const float* p = tab + 16 * k;
            float s = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += p[j];
            acc = __builtin_fmaf(acc, 0.5f, s);
        }
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <typename F> static double run(const char* name, F launch, double wave_instrs_per_wave, int waves) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    double total = wave_instrs_per_wave * waves;
    double cyc = ms * 1e-3 * 2.4e9 * 1024.0 / total;      // SIMD cycles per wave-instruction
    printf("%-14s %8.3f ms  %6.2f SIMD-cycles/wave-instr (at 2.4 GHz, %d waves)\n", name, ms, cyc, waves);
    return cyc;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int waves = 1024 * 8 * 2;
    const double n = (double)LOOPS * 32;
#define RUN(K) run(#K, [&] { hipLaunchKernelGGL(K, dim3(waves), dim3(64), 0, 0, out, 1.5f); }, n, waves)
    RUN(k_fma); RUN(k_mul); RUN(k_add); RUN(k_pkfma); RUN(k_pkmul);
    RUN(k_mullo); RUN(k_mulhi); RUN(k_mul24); RUN(k_mad24); RUN(k_mad64);
    RUN(k_sqrt); RUN(k_rcp); RUN(k_rsq); RUN(k_xor); RUN(k_addu); RUN(k_lshr); RUN(k_cvt);
    RUN(k_cnd); RUN(k_cmp); RUN(k_cmps);
    run("k_sqrt3fma(x4)", [&] { hipLaunchKernelGGL(k_sqrt3fma, dim3(waves), dim3(64), 0, 0, out, 1.5f); }, n, waves);
    RUN(k_cnd64); RUN(k_cndw); RUN(k_cndw_nop); RUN(k_min); RUN(k_med3); RUN(k_max3); RUN(k_fma_sgpr); RUN(k_fmaak); RUN(k_readlane); RUN(k_readfirst); RUN(k_cvt_u32); RUN(k_fract); RUN(k_floor); RUN(k_bfe); RUN(k_xad); RUN(k_lshladd); RUN(k_alignbit); RUN(k_andor); RUN(k_movdpp); RUN(k_madu16); RUN(k_salu); RUN(k_fma_salu); RUN(k_2fma_cmp); RUN(k_log); RUN(k_sin);
    RUN(k_sub); RUN(k_fmac); RUN(k_fma_neg); RUN(k_fma_absneg); RUN(k_and); RUN(k_or); RUN(k_lshl); RUN(k_ashr); RUN(k_mov); RUN(k_max); RUN(k_subu); RUN(k_xor_sdwa); RUN(k_mul_lit); RUN(k_add_sgpr); RUN(k_cmp_u32); RUN(k_cmp2fma_cnd); RUN(k_min3); RUN(k_min_u32); RUN(k_rndne); RUN(k_ldexp); RUN(k_cvt_i32); RUN(k_mad_i24); RUN(k_add_co); RUN(k_bfi); RUN(k_fma_dep1);
    RUN(k_div_ieee); RUN(k_sqrt_ieee);
    RUN(k_ldsb128);
    float* tab; hipMalloc(&tab, 292 * 64); hipMemset(tab, 0, 292 * 64);
    run("k_smem64", [&] { hipLaunchKernelGGL(k_smem64, dim3(waves), dim3(64), 0, 0, out, tab, 292); }, (double)(LOOPS / 8) * 292, waves);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
