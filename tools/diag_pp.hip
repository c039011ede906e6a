// diag_pp.hip — ablation harness for the ping-pong GEMM main loop (copy of kernels_gemm5.hip's loop, bf16,
// direct bf16 stores) with switchable pieces.  Timing only: ablated outputs are meaningless.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_pp.hip -o tools/diag_pp
// DIAG bits: 1 no DMA in the loop, 2 no MFMA, 4 no fragment reads, 8 no epilogue stores, 16 no barriers in loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 vec8;
typedef __attribute__((ext_vector_type(4))) __bf16 vec4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N> __device__ __forceinline__ void wv() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int DIAG> __device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    if (!(DIAG & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

template <int DIAG>
__global__ void __launch_bounds__(512, 2)
k(const __bf16* __restrict__ A, const __bf16* __restrict__ W, const float* __restrict__ bias, __bf16* __restrict__ outp,
  int M, int N, int K, int tiles_m, int tiles_n, unsigned long long* stamps) {
    constexpr int BM = 256, BN = 256, BK = 64, STAGE_BYTES = 65536, W_OFF = 32768, MI = 8, NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wn = wave & 3;
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const __bf16 *gA[4], *gW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = grp * 128 + (i * 4 + wn) * 8 + lr;
        int ra = tile_m * BM + r, rw = tile_n * BN + r;
        ra = ra < M ? ra : M - 1; rw = rw < N ? rw : N - 1;
        gA[i] = A + (int64_t)ra * K + lc * 8; gW[i] = W + (int64_t)rw * K + lc * 8;
    }
    const int dma_off = grp * 16384 + wn * 1024;
    auto issue_a = [&](int kt) {
        char* dst = smem + (kt & 1) * STAGE_BYTES + dma_off;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gA[i] + kt * BK),
                                             (void __attribute__((address_space(3)))*)(dst + i * 4096), 16, 0, 0);
    };
    auto issue_w = [&](int kt) {
        char* dst = smem + (kt & 1) * STAGE_BYTES + W_OFF + dma_off;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gW[i] + kt * BK),
                                             (void __attribute__((address_space(3)))*)(dst + i * 4096), 16, 0, 0);
    };
    const int frow = lane & 15, fq = lane >> 4;
    const int off0 = frow * 128 + (((0 | fq) ^ (frow & 7)) << 4), off1 = frow * 128 + (((4 | fq) ^ (frow & 7)) << 4);
    const int xbase = grp * 16384, wbase = W_OFF + wn * 8192;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    vec8 wf0[NI], wf1[NI], xf[MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) { wf0[ni] = vec8{}; wf1[ni] = vec8{}; }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[mi] = vec8{};
    const int nk = K / BK;
    constexpr bool DMA = !(DIAG & 1), MF = !(DIAG & 2), RD = !(DIAG & 4);
    unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
    issue_w(0); issue_a(0);
    if (nk > 1 && DMA) { issue_w(1); issue_a(1); wv<8>(); } else wv<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) bar<DIAG>();
    unsigned long long ph[5] = {0, 0, 0, 0, 0};
    unsigned long long tp = __builtin_amdgcn_s_memtime();
#define STAMP(i) if (DIAG & 32) { __builtin_amdgcn_sched_barrier(0); unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[i] += tn_ - tp; tp = tn_; __builtin_amdgcn_sched_barrier(0); }
    for (int kt = 0; kt < nk; ++kt) {
        const char* st = smem + (kt & 1) * STAGE_BYTES;
        if (RD) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) { wf0[ni] = *(const vec8*)(st + wbase + ni * 2048 + off0); wf1[ni] = *(const vec8*)(st + wbase + ni * 2048 + off1); }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const vec8*)(st + xbase + mi * 2048 + off0);
        }
        if (DMA && !(DIAG & 64) && kt >= 1 && kt + 1 < nk) issue_a(kt + 1);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        STAMP(0)
        bar<DIAG>();
        STAMP(4)
        if (MF) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[ni], xf[mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(xf[mi]));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf0[ni]));
        }
        STAMP(1)
        bar<DIAG>();
        STAMP(4)
        if (RD) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const vec8*)(st + xbase + mi * 2048 + off1);
        }
        if (DMA) {
            if (kt + 2 < nk) { issue_w(kt + 2); wv<8>(); } else if (kt + 1 < nk) wv<4>(); else wv<0>();
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        STAMP(2)
        bar<DIAG>();
        STAMP(4)
        if (MF) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[ni], xf[mi], acc[mi][ni], 0, 0, 0);
                if ((DIAG & 64) && DMA && (mi & 1) == 0 && kt + 2 < nk) {
                    const int i = mi >> 1;
                    char* dst = smem + (kt & 1) * STAGE_BYTES + dma_off;
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gA[i] + (kt + 2) * BK),
                                                     (void __attribute__((address_space(3)))*)(dst + i * 4096), 16, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(xf[mi]));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf1[ni]));
        }
        if (DMA) { if (kt + 2 < nk) { if (DIAG & 64) wv<8>(); else wv<4>(); } else wv<0>(); }
        STAMP(3)
        bar<DIAG>();
        STAMP(4)
    }
    if (grp == 0) bar<DIAG>();
    unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
    const int m0 = tile_m * BM + grp * 128 + frow, n0 = tile_n * BN + wn * 64 + fq * 4;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int m = m0 + mi * 16, n = n0 + ni * 16;
            f32x4 v = acc[mi][ni] + *(const f32x4*)(bias + n);
            vec4 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
            if (DIAG & 8) { if (v[0] == 123456.789f) *(vec4*)(outp + (int64_t)m * N + n) = o; }
            else if (m < M && n < N) *(vec4*)(outp + (int64_t)m * N + n) = o;
        }
    unsigned long long ts2 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long ts3 = __builtin_amdgcn_s_memrealtime();
    if (stamps && (DIAG & 32) && (threadIdx.x == 0 || threadIdx.x == 256)) { const int o = (threadIdx.x >> 8) * 5; for (int i = 0; i < 5; ++i) stamps[(size_t)gridDim.x * 4 + blockIdx.x * 10 + o + i] = ph[i]; }
    if (stamps && threadIdx.x == 0) { stamps[blockIdx.x * 4 + 0] = ts0; stamps[blockIdx.x * 4 + 1] = ts1; stamps[blockIdx.x * 4 + 2] = ts2; stamps[blockIdx.x * 4 + 3] = ts3; }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int DIAG>
float run(const __bf16* A, const __bf16* W, const float* b, __bf16* o, int M, int N, int K, int iters) {
    const int tm = (M + 255) / 256, tn = (N + 255) / 256;
    auto kern = k<DIAG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(512), 131072, 0, A, W, b, o, M, N, K, tm, tn, (unsigned long long*)nullptr);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(512), 131072, 0, A, W, b, o, M, N, K, tm, tn, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3f;
}

__global__ void fillk(__bf16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (__bf16)(((int)(h & 0xFFFF) - 32768) * (scale / 32768.f));
    }
}

int main() {
    const int M = 100864;
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"fc2", 768, 3072}};
    for (auto& s : shapes) {
        __bf16 *A, *W, *o; float* b;
        CK(hipMalloc(&A, (size_t)M * s.K * 2)); CK(hipMalloc(&W, (size_t)s.N * s.K * 2)); CK(hipMalloc(&o, (size_t)M * s.N * 2));
        CK(hipMalloc(&b, s.N * 4)); CK(hipMemset(b, 0, s.N * 4));
        fillk<<<4096, 256>>>(A, (size_t)M * s.K, 1, 1.0f);
        fillk<<<1024, 256>>>(W, (size_t)s.N * s.K, 2, 0.05f);
        CK(hipDeviceSynchronize());
        const double fl = 2.0 * M * s.N * s.K;
        float t;
#define R(D, label) t = run<D>(A, W, b, o, M, s.N, s.K, 20); printf("%s pp %-36s %8.1f us  %7.1f TF-equiv\n", s.name, label, t, fl / t / 1e6);
        R(0, "baseline")
        R(8, "no stores")
        R(72, "no stores, A-DMA inside C1")
        R(9, "no DMA, no stores")
        R(13, "MFMA only")
        R(29, "MFMA only, no barriers")
        R(14, "DMA only (no MFMA/reads/stores)")
        R(10, "DMA + reads (no MFMA, no stores)")
        R(12, "DMA + MFMA (no reads, no stores)")
#undef R
        {
            const int tm = (M + 255) / 256, tn = (s.N + 255) / 256, nt = tm * tn;
            unsigned long long* st; CK(hipMalloc(&st, nt * 32));
            auto kern = k<0>;
            hipLaunchKernelGGL(kern, dim3(nt), dim3(512), 131072, 0, A, W, b, o, M, s.N, s.K, tm, tn, st);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(nt * 4);
            CK(hipMemcpy(h.data(), st, nt * 32, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, tend = 0; double s_main = 0, s_epi = 0, s_drain = 0;
            for (int i = 0; i < nt; ++i) { if (h[i*4] < t0) t0 = h[i*4]; if (h[i*4+3] > tend) tend = h[i*4+3]; s_main += h[i*4+1]-h[i*4]; s_epi += h[i*4+2]-h[i*4+1]; s_drain += h[i*4+3]-h[i*4+2]; }
            printf("%s stamps: kernel span %.1f us; per tile avg: main %.2f us, epilogue issue %.2f us, store drain %.2f us (100 MHz ticks)\n", s.name, (tend - t0) / 100.0, s_main / nt / 100.0, s_epi / nt / 100.0, s_drain / nt / 100.0);
            // start-time histogram of the first 512 tiles, to see lockstep vs spread
            int hist[16] = {0}; for (int i = 0; i < nt; ++i) { unsigned long long d = (h[i*4+1] - t0) / 100; int bkt = (int)(d % 32) / 2; hist[bkt]++; }
            printf("%s mainloop-end time mod 32us histogram:", s.name); for (int i = 0; i < 16; ++i) printf(" %d", hist[i]); printf("\n");
            hipFree(st);
        }
        {
            const int tm = (M + 255) / 256, tn = (s.N + 255) / 256, nt = tm * tn;
            unsigned long long* st; CK(hipMalloc(&st, (size_t)nt * (32 + 80)));
            auto kern = k<40 + PPV>;  // stamps + no stores
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            hipLaunchKernelGGL(kern, dim3(nt), dim3(512), 131072, 0, A, W, b, o, M, s.N, s.K, tm, tn, st);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h((size_t)nt * 14);
            CK(hipMemcpy(h.data(), st, (size_t)nt * 112, hipMemcpyDeviceToHost));
            double sum[2][5] = {{0}};
            for (int i = 0; i < nt; ++i) for (int g = 0; g < 2; ++g) for (int p = 0; p < 5; ++p) sum[g][p] += (double)h[(size_t)nt * 4 + i * 10 + g * 5 + p];
            const double per = (double)nt * (s.K / 64);
            for (int g = 0; g < 2; ++g) printf("%s phase cycles per K-tile, group %d: L0 %.0f  C0 %.0f  L1 %.0f  C1 %.0f  barrier-wait %.0f (4 barriers)\n", s.name, g, sum[g][0] / per, sum[g][1] / per, sum[g][2] / per, sum[g][3] / per, sum[g][4] / per);
            hipFree(st);
        }
        hipFree(A); hipFree(W); hipFree(o); hipFree(b);
    }
    return 0;
}
