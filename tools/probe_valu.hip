// probe_valu.hip — issue rate of the vector instructions a GEMM epilogue is made of, on one SIMD of gfx950, in shader
// cycles per wave64 instruction at 1, 2 and 4 waves per SIMD (round 4: is packed fp16 arithmetic cheaper than packed /
// plain fp32 for the GELU polynomial of the bf16 / e4m3-output fc1 epilogues?).
// Every loop body is 32 independent instructions of ONE kind on 8 register chains (no dependent pair closer than 8
// instructions), repeated 256 times between two s_memtime stamps; the figure printed is
//   cycles x waves-per-SIMD / instructions  =  cycles per instruction as the SIMD sees it.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_valu.hip -o tools/probe_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY4(X) REP8(X) REP8(X) REP8(X) REP8(X)

// chains: a0..a7 (accumulators), b, c (operands).  32-bit and 64-bit (packed fp32) variants.
#define I_FMA32(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define I_PKFMA32(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n\t"
#define I_PKFMA32_S(i) "v_pk_fma_f32 %" #i ", %8, %10, %" #i "\n\t"
#define I_PKFMA32_CL(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i " clamp\n\t"
#define I_PKFMA16(i) "v_pk_fma_f16 %" #i ", %8, %9, %" #i "\n\t"
#define I_PKMUL16(i) "v_pk_mul_f16 %" #i ", %8, %" #i "\n\t"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n\t"
#define I_CVTPK16(i) "v_cvt_pkrtz_f16_f32 %" #i ", %8, %9\n\t"
#define I_CVTF32(i) "v_cvt_f32_f16 %" #i ", %8\n\t"
#define I_CVTBF(i) "v_cvt_pk_bf16_f32 %" #i ", %8, %9\n\t"
#define I_EXP(i) "v_exp_f32 %" #i ", %8\n\t"
#define I_PKADD32(i) "v_pk_add_f32 %" #i ", %8, %" #i "\n\t"
#define I_PKMUL32(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n\t"
#define I_MAX16(i) "v_pk_max_f16 %" #i ", %8, %" #i "\n\t"
#define I_FP8(i) "v_cvt_pk_fp8_f32 %" #i ", %8, %9\n\t"

// dependency distance: the same 32 packed fmas per body on 1, 2 or 4 register chains instead of 8 (round 4, after the evidence
// set: the GELU polynomial of the fc1 epilogue advances TWO chains in alternation -- what does a wave pay for that?)
#define REP8_C1(X) X(0) X(0) X(0) X(0) X(0) X(0) X(0) X(0)
#define REP8_C2(X) X(0) X(1) X(0) X(1) X(0) X(1) X(0) X(1)
#define REP8_C4(X) X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#define BODY4_C1(X) REP8_C1(X) REP8_C1(X) REP8_C1(X) REP8_C1(X)
#define BODY4_C2(X) REP8_C2(X) REP8_C2(X) REP8_C2(X) REP8_C2(X)
#define BODY4_C4(X) REP8_C4(X) REP8_C4(X) REP8_C4(X) REP8_C4(X)

template <int KIND>
__global__ void __launch_bounds__(1024) probe(unsigned long long* out, float seed) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    unsigned long long t0, t1;
    const float s = seed + (float)threadIdx.x * 1e-6f;
    if constexpr (KIND == 1 || KIND == 9 || KIND == 10 || KIND >= 20) {   // 64-bit operands
        f32x2 a0{s, s}, a1{s, s}, a2{s, s}, a3{s, s}, a4{s, s}, a5{s, s}, a6{s, s}, a7{s, s}, b{0.999f, 0.999f}, c{1e-3f, 1e-3f};
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int r = 0; r < 256; ++r) {
            if constexpr (KIND == 1) asm volatile(BODY4(I_PKFMA32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 9) asm volatile(BODY4(I_PKADD32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 20) asm volatile(BODY4_C1(I_PKFMA32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 21) asm volatile(BODY4_C2(I_PKFMA32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 22) asm volatile(BODY4_C4(I_PKFMA32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 23) { f32x2 sc = {0.5f, 0.25f}; asm volatile(BODY4(I_PKFMA32_S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(sc)); }
            if constexpr (KIND == 24) asm volatile(BODY4(I_PKFMA32_CL) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if constexpr (KIND == 10) asm volatile(BODY4(I_PKMUL32) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    } else {
        float a0 = s, a1 = s, a2 = s, a3 = s, a4 = s, a5 = s, a6 = s, a7 = s, b = 0.999f, c = 1e-3f;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int r = 0; r < 256; ++r) {
#define RUN(K, I) if constexpr (KIND == K) asm volatile(BODY4(I) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            RUN(0, I_FMA32) RUN(2, I_PKFMA16) RUN(3, I_PKMUL16) RUN(4, I_MED3) RUN(5, I_CVTPK16) RUN(6, I_CVTF32) RUN(7, I_CVTBF) RUN(8, I_EXP)
            RUN(11, I_MAX16) RUN(12, I_FP8)
#undef RUN
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, unsigned long long* d) {
    printf("%-24s", name);
    for (int waves : {4, 8, 16}) {   // waves per workgroup = 1, 2, 4 per SIMD; one workgroup per CU on every CU
        const int grid = 256;
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(waves * 64), 0, 0, d, 1.0f);
        hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(waves * 64), 0, 0, d, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid * 16);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < grid; ++b) for (int w = 0; w < waves; ++w) sum += (double)h[b * 16 + w];
        const double cyc = sum / (grid * waves);
        printf("  %d/SIMD: %6.2f", waves / 4, cyc / (256.0 * 32.0) / (waves / 4.0) * (waves / 4.0));   // per wave
        printf(" (SIMD %5.2f)", cyc / (256.0 * 32.0) / (waves / 4.0));
    }
    printf("\n");
}

int main() {
    unsigned long long* d;
    hipMalloc((void**)&d, 256 * 16 * 8);
    printf("cycles per wave64 instruction as one wave sees it (and, in brackets, per instruction as the SIMD sees it)\n");
    run<0>("v_fma_f32", d);
    run<1>("v_pk_fma_f32", d);
    run<20>("v_pk_fma_f32, 1 chain", d);
    run<21>("v_pk_fma_f32, 2 chains", d);
    run<22>("v_pk_fma_f32, 4 chains", d);
    run<23>("v_pk_fma_f32, SGPR pair", d);
    run<24>("v_pk_fma_f32 clamp", d);
    run<9>("v_pk_add_f32", d);
    run<10>("v_pk_mul_f32", d);
    run<2>("v_pk_fma_f16", d);
    run<3>("v_pk_mul_f16", d);
    run<11>("v_pk_max_f16", d);
    run<4>("v_med3_f32", d);
    run<5>("v_cvt_pkrtz_f16_f32", d);
    run<6>("v_cvt_f32_f16", d);
    run<7>("v_cvt_pk_bf16_f32", d);
    run<12>("v_cvt_pk_fp8_f32", d);
    run<8>("v_exp_f32", d);
    hipFree(d);
    return 0;
}
