// diag_gemm.hip — standalone ablation harness for the BK=64 two-stage GEMM main loop (gfx950).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_gemm.hip -o tools/diag_gemm
// Run on the GPU box: tools/diag_gemm            (prints a table of ablated timings)
//
// DIAG bits: 1 = issue no DMA after the prologue (loads off), 2 = no MFMA (fragment reads kept alive),
//            4 = no fragment reads (MFMA on stale registers), 8 = no epilogue stores.
// Timing-only builds: outputs of ablated variants are meaningless.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int BM, int BN, int WM, int WN, int DIAG>
__global__ void __launch_bounds__(WM* WN * 64)
k(const __bf16* __restrict__ A, const __bf16* __restrict__ W, const float* __restrict__ bias, __bf16* __restrict__ out,
  int M, int N, int K, int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN, BK = 64, ROWS = BM + BN, STAGE_BYTES = ROWS * 128;
    constexpr int GROUPS_A = BM / 8, GROUPS = ROWS / 8, LOADS = GROUPS / NW, LOADS_A = GROUPS_A / NW;
    constexpr int TM = BM / WM, TN = BN / WN, MI = TM / 16, NI = TN / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = tiles_m * tiles_n, bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const __bf16* gsrc[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int gi = i * NW + wave;
        if (i < LOADS_A) { int row = tile_m * BM + gi * 8 + lr; row = row < M ? row : M - 1; gsrc[i] = A + (int64_t)row * K + lc * 8; }
        else { int row = tile_n * BN + (gi - GROUPS_A) * 8 + lr; row = row < N ? row : N - 1; gsrc[i] = W + (int64_t)row * K + lc * 8; }
    }
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int gi = i * NW + wave;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(gsrc[i] + kt * BK),
                                             (void __attribute__((address_space(3)))*)(smem + stage * STAGE_BYTES + gi * 1024), 16, 0, 0);
        }
    };
    const int frow = lane & 15, fq = lane >> 4;
    int offk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offk[ks] = frow * 128 + ((((ks << 2) | fq) ^ (frow & 7)) << 4);
    const int xbase = wm * TM * 128, wbase = BM * 128 + wn * TN * 128;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[MI], wf[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[mi] = bf16x8{};
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) wf[ni] = bf16x8{};
    const int nk = K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk && !(DIAG & 1)) issue((kt + 1) & 1, kt + 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (!(DIAG & 4)) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const bf16x8*)(st + wbase + ni * 2048 + offk[ks]);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const bf16x8*)(st + xbase + mi * 2048 + offk[ks]);
            }
            if (!(DIAG & 2)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) asm volatile("" ::"v"(xf[mi]));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) asm volatile("" ::"v"(wf[ni]));
            }
        }
    }
    const int m0 = tile_m * BM + wm * TM + frow, n0 = tile_n * BN + wn * TN + fq * 4;
    if (DIAG & 16) {  // same bytes, ideally coalesced: 16 B per lane, 1 KiB contiguous per instruction
        char* base = (char*)out + (size_t)wg * (BM * BN * 2) + wave * (TM * TN * 2);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ni += 2) {
                f32x4 v = acc[mi][ni] + acc[mi][ni + 1];
                typedef __attribute__((ext_vector_type(8))) __bf16 b8;
                b8 o;
                for (int j = 0; j < 4; ++j) { o[j] = (__bf16)v[j]; o[4 + j] = (__bf16)acc[mi][ni + 1][j]; }
                *(b8*)(base + ((mi * (NI / 2) + ni / 2) * 64 + lane) * 16) = o;
            }
        return;
    }
    if (DIAG & (32 | 64 | 128)) {  // row-major output, 16 B per lane: 128 B (32), 512 B (64) or 256 B (128) per row per instruction
        typedef __attribute__((ext_vector_type(8))) __bf16 b8;
#pragma unroll
        for (int i = 0; i < MI * NI / 2; ++i) {
            f32x4 v = acc[i / 2][(i & 1) * 2] + acc[i / 2][(i & 1) * 2 + 1];
            b8 o;
            for (int j = 0; j < 4; ++j) { o[j] = (__bf16)v[j]; o[4 + j] = (__bf16)v[3 - j]; }
            int row, colb;
            if (DIAG & 32) { row = tile_m * BM + wm * TM + i * 8 + (lane >> 3); colb = (tile_n * BN + wn * TN) * 2 + (lane & 7) * 16; }
            else if (DIAG & 64) { row = tile_m * BM + wave * 32 + i * 2 + (lane >> 5); colb = tile_n * BN * 2 + (lane & 31) * 16; }
            else { row = tile_m * BM + wm * TM + (wn >> 1) * 64 + i * 4 + (lane >> 4); colb = (tile_n * BN + (wn & 1) * 128) * 2 + (lane & 15) * 16; }
            if (row < M) *(b8*)((char*)out + (int64_t)row * N * 2 + colb) = o;
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + mi * 16;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 16;
            f32x4 v = acc[mi][ni] + *(const f32x4*)(bias + n);
            bf16x4 o;
            o[0] = (__bf16)v[0]; o[1] = (__bf16)v[1]; o[2] = (__bf16)v[2]; o[3] = (__bf16)v[3];
            if (DIAG & 8) { if (v[0] == 123456.789f) *(bf16x4*)(out + (int64_t)m * N + n) = o; }
            else if (m < M && n < N) *(bf16x4*)(out + (int64_t)m * N + n) = o;
        }
    }
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

template <int BM, int BN, int WM, int WN, int DIAG>
float run(const __bf16* A, const __bf16* W, const float* b, __bf16* o, int M, int N, int K, int iters) {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    const size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto kern = k<BM, BN, WM, WN, DIAG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(WM * WN * 64), lds, 0, A, W, b, o, M, N, K, tm, tn);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(tm * tn), dim3(WM * WN * 64), lds, 0, A, W, b, o, M, N, K, tm, tn);
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
#define R(D, label) t = run<256, 256, 2, 4, D>(A, W, b, o, M, s.N, s.K, 20); printf("%s 256x256 %-28s %8.1f us  %7.1f TF-equiv\n", s.name, label, t, fl / t / 1e6);
        R(0, "baseline")
        R(8, "no epilogue stores")
        R(16, "ideal coalesced stores")
        R(32, "row-major 128 B/row/instr")
        R(128, "row-major 256 B/row/instr")
        R(64, "row-major 512 B/row/instr")
        R(1, "no DMA in loop")
        R(9, "no DMA, no stores")
        R(2, "no MFMA")
        R(4, "no fragment reads")
        R(5, "no DMA, no frag reads (MFMA)")
        R(13, "MFMA only (no DMA/reads/st)")
        R(6, "no MFMA, no reads (DMA only)")
        R(14, "DMA+barrier only, no stores")
#undef R
        hipFree(A); hipFree(W); hipFree(o); hipFree(b);
    }
    return 0;
}
