// kernels_gemm3.hip — the software-pipelined GEMM of the ViT hot path (gfx950), variants 3 and 4.
//
// Same contract and epilogues as kernels_gemm.hip (out = epilogue(A[M,K] W[N,K]^T), 16-bit MFMA
// operands, fp32 accumulate, transposed-tile orientation).  What changes is the main loop:
//
//   * BK = 32 and S >= 3 LDS stages filled by global_load_lds_dwordx4.  S-1 K-tiles are in flight or
//     landed ahead of the one being multiplied; the wait before the barrier is a COUNTED
//     s_waitcnt vmcnt((S-2)*LOADS), so DMA of later tiles stays in flight ACROSS the barrier
//     (raw s_barrier, never __syncthreads: its fence would drain the DMA queue).
//   * two statically named fragment register sets: right after the barrier of step k the wave
//     issues the ds_read_b128 of tile k+1 into the other set, then runs the 32 MFMAs of tile k.
//     LDS latency is hidden behind a full MFMA block; an MFMA never waits for a read issued in
//     the same step.
//   * variant 3: 256x128 tile, 4 waves (one per SIMD), 72 KiB LDS -> TWO workgroups per CU.  The two
//     co-resident workgroups are independent, so one's epilogue (bias/GELU/residual VALU + stores),
//     prologue and barrier skew overlap the other's MFMA stream on the same SIMD.
//     variant 4: 256x256 tile, 8 waves, 4 stages (128 KiB), one workgroup per CU (less L2->LDS
//     traffic per flop; no inter-workgroup overlap).
//   * LDS rows are 64 B (32 x 16-bit) = 4 chunks of 16 B; chunk c of row r is stored at chunk
//     c ^ ((4 - (r>>2)) & 3).  As in the BK=64 kernel the DMA destination is linear, the swizzle is
//     applied to the lane's global source address and again on the ds_read side: conflict-free for
//     all four 16-lane groups of ds_read_b128 (checked against the §LDS bank rule).
#include "gemm_epilogue.h"
#include "vh_kernels.h"

namespace vh {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <typename T, int BM, int BN, int WM, int WN, int S, int EPI>
__global__ void __launch_bounds__(WM* WN * 64, 2)
gemm_nt_pipe_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W,
                    const float* __restrict__ bias, void* __restrict__ outp, int M, int N, int K,
                    const float* __restrict__ aux, int aux_i, int tiles_m, int tiles_n) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int NW = WM * WN;
    constexpr int BK = 32;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE_BYTES = ROWS * 64;
    constexpr int GROUPS_A = BM / 16;  // 1-KiB groups: 16 rows x 64 B
    constexpr int GROUPS = ROWS / 16;
    constexpr int LOADS = GROUPS / NW;
    constexpr int LOADS_A = GROUPS_A / NW;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    static_assert(GROUPS % NW == 0 && GROUPS_A % NW == 0, "tile/wave mismatch");
    static_assert(S >= 3 && S <= 4, "3 or 4 stages");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- DMA source pointers --------------------------------------------------------------
    const int lr = lane >> 2;                            // row inside the 16-row group
    const int lc = (lane & 3) ^ ((4 - (lr >> 2)) & 3);   // logical chunk landing in slot lane&3
    const elem* gsrc[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int gi = i * NW + wave;
        if (i < LOADS_A) {
            int row = tile_m * BM + gi * 16 + lr;
            row = row < M ? row : M - 1;
            gsrc[i] = A + (int64_t)row * K + lc * 8;
        } else {
            int row = tile_n * BN + (gi - GROUPS_A) * 16 + lr;
            row = row < N ? row : N - 1;
            gsrc[i] = W + (int64_t)row * K + lc * 8;
        }
    }
    auto issue = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const int gi = i * NW + wave;
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1)))*)(gsrc[i] + kt * BK),
                (void __attribute__((address_space(3)))*)(smem + stage * STAGE_BYTES + gi * 1024), 16, 0, 0);
        }
    };

    // ---- fragment reads ---------------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    const int foff = frow * 64 + ((fq ^ ((4 - (frow >> 2)) & 3)) << 4);
    const int xbase = wm * TM * 64 + foff;
    const int wbase = BM * 64 + wn * TN * 64 + foff;
    struct Frags {
        vec8 x[MI];
        vec8 w[NI];
    };
    auto read_frags = [&](Frags& f, int stage) {
        const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) f.w[ni] = *(const vec8*)(st + wbase + ni * 1024);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) f.x[mi] = *(const vec8*)(st + xbase + mi * 1024);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;

    // ---- prologue: fill all S stages, make tile 0 visible, fetch its fragments ---------------
#pragma unroll
    for (int t = 0; t < S; ++t)
        if (t < nk) issue(t, t);
    if (nk >= S) wait_vmcnt<(S - 1) * LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    read_frags(fa, 0);

    // one K-step: tile kt's fragments are in `cur`; prefetch tile kt+1 into `nxt`
    auto mfma_block = [&](const Frags& cur) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T::mfma16(cur.w[ni], cur.x[mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
    };
    // a step that has a successor (kt + 1 < nk): no data-dependent join between the prefetch reads and
    // the MFMA block, so the compiler's lgkmcnt bookkeeping stays exact
    auto step = [&](Frags& cur, Frags& nxt, int kt, int stage_next, int stage_free) {
        // tile kt+1 must have landed: allow only the tiles after it to stay in flight
        const int later = nk - 2 - kt;  // tiles issued after kt+1 (at most S-2)
        if (later >= S - 2) wait_vmcnt<(S - 2) * LOADS>();
        else if (S == 4 && later == 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): my reads of tile kt are complete
        __builtin_amdgcn_s_barrier();        // => tile kt+1 visible, stage of tile kt free
        if (kt + S < nk) issue(stage_free, kt + S);
        read_frags(nxt, stage_next);
        mfma_block(cur);
    };

    int st = 0;  // stage of tile kt
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
        const int s1 = st + 1 == S ? 0 : st + 1;
        const int s2 = s1 + 1 == S ? 0 : s1 + 1;
        step(fa, fb, kt, s1, st);
        step(fb, fa, kt + 1, s2, s1);
        st = s2;
    }
    if (kt + 2 == nk) {
        const int s1 = st + 1 == S ? 0 : st + 1;
        step(fa, fb, kt, s1, st);
        mfma_block(fb);
    } else {
        mfma_block(fa);
    }

    gemm_epilogue<T, EPI, MI, NI>(acc, bias, outp, M, N, tile_m * BM + wm * TM, tile_n * BN + wn * TN, lane, aux, aux_i,
                                  (tile_m + 1) * BM <= M && (tile_n + 1) * BN <= N, smem, wave);
}

template <typename T, int BM, int BN, int WM, int WN, int S, int EPI>
static hipError_t launch_pipe(const GemmArgs& g, hipStream_t s) {
    const int tiles_m = (int)((g.M + BM - 1) / BM), tiles_n = (g.N + BN - 1) / BN;
    constexpr size_t lds = (size_t)S * (BM + BN) * 64;
    auto k = gemm_nt_pipe_kernel<T, BM, BN, WM, WN, S, EPI>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, s, (const typename T::elem*)g.a,
                       (const typename T::elem*)g.w, g.bias, g.out, (int)g.M, g.N, g.K, g.aux, g.aux_i, tiles_m, tiles_n);
    return hipGetLastError();
}

template <typename T, int EPI>
hipError_t launch_gemm_pipelined(const GemmArgs& g, int variant, hipStream_t s) {
    if (variant == 4) return launch_pipe<T, 256, 256, 2, 4, 4, EPI>(g, s);
    return launch_pipe<T, 256, 128, 2, 2, 3, EPI>(g, s);
}

#define VH_INST(T)                                                                                  \
    template hipError_t launch_gemm_pipelined<T, VH_EPI_BIAS>(const GemmArgs&, int, hipStream_t);      \
    template hipError_t launch_gemm_pipelined<T, VH_EPI_BIAS_GELU>(const GemmArgs&, int, hipStream_t); \
    template hipError_t launch_gemm_pipelined<T, VH_EPI_BIAS_RESID>(const GemmArgs&, int, hipStream_t);\
    template hipError_t launch_gemm_pipelined<T, VH_EPI_BIAS_F32>(const GemmArgs&, int, hipStream_t);  \
    template hipError_t launch_gemm_pipelined<T, VH_EPI_PATCH>(const GemmArgs&, int, hipStream_t);
VH_INST(BF16)
VH_INST(FP16)

}  // namespace vh
