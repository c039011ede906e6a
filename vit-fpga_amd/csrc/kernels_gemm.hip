// kernels_gemm.hip — the dense contraction of the ViT hot path, hand-written for gfx950.
//
//   out = epilogue( A[M,K] x W[N,K]^T ),  A/W 16-bit (bf16 or fp16), fp32 accumulate (MFMA)
//
// This one kernel family carries 92 % of the forward's FLOPs: patch embedding (conv as GEMM),
// fused q|k|v projection, attention output projection, fc1 (+GELU) and fc2 (+residual), and
// the classifier head.  It is what replaces the per-layer body of the reference's
// `network_v1` task (netFPGA.cpp:275; a_l = act(W_l a_{l-1} + b_l), weights row-major
// [n_out, n_in], netFPGA.cpp:91-106) for a whole batch of token rows at once.
//
// Design (CDNA4):
//   * tile BM x BN x 64, NW = WM*WN waves of 64 lanes; v_mfma_f32_16x16x32_{bf16,f16}.
//   * operands staged global -> LDS with global_load_lds_dwordx4 (no VGPR round trip),
//     two LDS stages, one barrier per K-tile: the loads of tile k+1 are in flight while
//     tile k is multiplied.
//   * LDS rows are 128 B (64 x 16-bit); 16-B chunks are XOR-swizzled with (row & 7).  The DMA
//     writes LDS linearly, so the swizzle is applied to the per-lane GLOBAL source address and
//     again on the ds_read_b128 side (same involution) -> conflict-free fragment reads.
//   * the MFMA "A" operand is the W fragment and "B" the activation fragment, i.e. the wave
//     computes the transposed tile.  Each lane then owns 4 CONSECUTIVE output columns of one
//     row, so every epilogue access is an 8-byte (16-bit out) or 16-byte (fp32) vector.
//   * epilogues fused: +bias, GELU(erf), fp32 residual read-modify-write, patch-row remap
//     + position embedding.
//   * 1-D grid remapped so that the workgroups that share an XCD (blockIdx % 8) walk a
//     contiguous run of tiles, n fastest: neighbours reuse the same A panel from that XCD's L2.
//   * ragged M / N: loads clamp the row index, stores are predicated (N % 4 == 0, K % 64 == 0).
#include <cstdlib>

#include "gemm_epilogue.h"
#include "vh_kernels.h"

namespace vh {

template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ void __launch_bounds__(WM* WN * 64)
gemm_nt_kernel(const typename T::elem* __restrict__ A, const typename T::elem* __restrict__ W,
               const float* __restrict__ bias, void* __restrict__ outp, int M, int N, int K,
               const float* __restrict__ aux, int aux_i, int tiles_m, int tiles_n, const float* __restrict__ stats,
               void* __restrict__ out16, float* __restrict__ partials, int64_t prow) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int NW = WM * WN;
    constexpr int BK = 64;
    constexpr int ROWS = BM + BN;
    constexpr int STAGE_BYTES = ROWS * 128;
    constexpr int GROUPS_A = BM / 8;          // 1-KiB row groups (8 rows x 128 B) of the A part
    constexpr int GROUPS = ROWS / 8;
    constexpr int LOADS = GROUPS / NW;        // DMA instructions per wave per stage
    constexpr int LOADS_A = GROUPS_A / NW;
    constexpr int TM = BM / WM, TN = BN / WN; // wave tile
    constexpr int MI = TM / 16, NI = TN / 16;
    static_assert(GROUPS % NW == 0 && GROUPS_A % NW == 0, "tile/wave mismatch");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile assignment (bijective for any grid size) -----------------------
    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;

    // ---- DMA source pointers: lane -> (row, swizzled 16-B chunk) -------------------------
    const int lr = lane >> 3;                 // row inside the 8-row group
    const int lc = (lane & 7) ^ lr;           // logical chunk that lands in physical slot lane&7
    const elem* gsrc[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int gi = i * NW + wave;
        if (i < LOADS_A) {
            int row = tile_m * BM + gi * 8 + lr;
            row = row < M ? row : M - 1;
            gsrc[i] = A + (int64_t)row * K + lc * 8;
        } else {
            int row = tile_n * BN + (gi - GROUPS_A) * 8 + lr;
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
                (void __attribute__((address_space(3)))*)(smem + stage * STAGE_BYTES + gi * 1024),
                16, 0, 0);
        }
    };

    // ---- fragment read offsets ------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    int offk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offk[ks] = frow * 128 + ((((ks << 2) | fq) ^ (frow & 7)) << 4);
    const int xbase = wm * TM * 128;
    const int wbase = BM * 128 + wn * TN * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    issue(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // tile kt landed for every wave; everyone is done reading the other stage
        if (kt + 1 < nk) issue((kt + 1) & 1, kt + 1);
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 xf[MI], wf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *(const vec8*)(st + wbase + ni * 2048 + offk[ks]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                xf[mi] = *(const vec8*)(st + xbase + mi * 2048 + offk[ks]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T::mfma16(wf[ni], xf[mi], acc[mi][ni]);
        }
    }

    // ---- epilogue: lane owns rows m = .. + (lane&15), 4 consecutive columns n = .. + 4*(lane>>4)
    const EpiArgs e{bias, outp, M, N, aux, aux_i, stats, out16, partials, prow};
    gemm_epilogue<T, EPI, MI, NI>(acc, e, tile_m * BM + wm * TM, tile_n * BN + wn * TN, lane, (tile_n + 1) * BN <= N,
                                  (tile_m + 1) * BM <= M, smem, wave);
}

// ---- host side ---------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int EPI>
static hipError_t launch_one(const GemmArgs& g, hipStream_t s) {
    const int tiles_m = (int)((g.M + BM - 1) / BM), tiles_n = (g.N + BN - 1) / BN;
    constexpr size_t lds = 2 * (size_t)(BM + BN) * 128;
    auto k = gemm_nt_kernel<T, BM, BN, WM, WN, EPI>;
    static LdsDone lds_done;  // per instantiation, per device
    if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, lds_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, s,
                       (const typename T::elem*)g.a, (const typename T::elem*)g.w, g.bias, g.out,
                       (int)g.M, g.N, g.K, g.aux, g.aux_i, tiles_m, tiles_n, g.stats, g.out16, g.partials, gemm_prow(g));
    return hipGetLastError();
}

template <typename T, int EPI>
hipError_t launch_gemm_pingpong(const GemmArgs& g, int mode, hipStream_t s);  // kernels_gemm5.hip; mode = variant - 5

template <typename T, int EPI>
static hipError_t launch_tile(const GemmArgs& g, int variant, hipStream_t s) {
    if (variant >= 5 && variant <= 7) return launch_gemm_pingpong<T, EPI>(g, variant - 5, s);
    if (variant == 2) return launch_one<T, 256, 256, 2, 4, EPI>(g, s);
    return launch_one<T, 128, 128, 2, 2, EPI>(g, s);
}

template <typename T>
static hipError_t launch_epi(const GemmArgs& g, int variant, hipStream_t s) {
    switch (g.epilogue) {
    case VH_EPI_BIAS: return launch_tile<T, VH_EPI_BIAS>(g, variant, s);
    case VH_EPI_BIAS_GELU: return launch_tile<T, VH_EPI_BIAS_GELU>(g, variant, s);
    case VH_EPI_BIAS_RESID: return launch_tile<T, VH_EPI_BIAS_RESID>(g, variant, s);
    case VH_EPI_BIAS_F32: return launch_tile<T, VH_EPI_BIAS_F32>(g, variant, s);
    case VH_EPI_PATCH: return launch_tile<T, VH_EPI_PATCH>(g, variant, s);
    case VH_EPI_LNFOLD: return launch_tile<T, VH_EPI_LNFOLD>(g, variant, s);
    case VH_EPI_LNFOLD_GELU: return launch_tile<T, VH_EPI_LNFOLD_GELU>(g, variant, s);
    case VH_EPI_RESID_LN: return launch_tile<T, VH_EPI_RESID_LN>(g, variant, s);
    case VH_EPI_RESID_SPLIT: return launch_tile<T, VH_EPI_RESID_SPLIT>(g, variant, s);
    case VH_EPI_PATCH_SPLIT: return launch_tile<T, VH_EPI_PATCH_SPLIT>(g, variant, s);
    default: return hipErrorInvalidValue;
    }
}

// which ping-pong form a large shape takes.  Default: the PERSISTENT form (6: one workgroup per CU walks the tiles; the DMA
// stream runs on across tile boundaries, so a tile's first K-tiles are in LDS when its loop starts and no launch gap or
// prologue sits between tiles).  Measured on ViT-B/16 b512, same-box A/B against the one-tile-per-workgroup form (5):
// q|k|v -9 %, fc1 -6 %, out-proj -5 %, fc2 +-0.  The launcher itself falls back to form 5 where the persistent form does not
// apply (patch-row remap, N % 256 != 0, fewer than two K-tiles) and hands a ragged last row of tiles to it as a tile
// range.  VH_GEMM_PP = 5 / 6 / 7 forces one form (A/B runs, tests).
int gemm_pp_variant(int epilogue) {
    (void)epilogue;
    static const int v = [] { const int e = env_int("VH_GEMM_PP", 0); return e < 5 || e > 7 ? 0 : e; }();
    return v ? v : 6;
}
int gemm_pick_variant(int64_t M, int N, int epilogue) {
    // swept at batch 16..128 (ViT-B/16): 128 is best, +13 % at batch 64 over 256
    static const int min_tiles = [] { const int e = env_int("VH_PP_MIN_TILES", 128); return e < 1 ? 1 : e; }();
    const int64_t t256 = ((M + 255) / 256) * ((N + 255) / 256);
    return t256 >= min_tiles ? gemm_pp_variant(epilogue) : 1;  // the 256x256 ping-pong kernel needs enough tiles for the 256 CUs
}

// the persistent ping-pong form runs (launch_pp): enough tiles for the ping-pong kernel, whole 256 x 256 tiles, at least four K-tiles (the forms that prefetch their epilogue's first loads need a W(k+2) slot in K-tile 1)
bool gemm_tiled_applies(int64_t M, int N, int K) {
    return M > 0 && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K / 64 >= 4 && gemm_pick_variant(M, N, VH_EPI_BIAS) == 6;
}

bool gemm_tiled_applies_f8(int64_t M, int N, int K) {   // (launch_gemm_fp8 takes the persistent form whenever whole tiles allow it)
    return M > 0 && M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && K / 128 >= 4 && gemm_pp_variant(VH_EPI_BIAS) == 6;
}

const char* gemm_check(const GemmArgs& g) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0) return "gemm: empty shape";
    if (g.K % 64) return "gemm: K must be a multiple of 64";
    if (g.N % 4) return "gemm: N must be a multiple of 4";
    if (g.M > 0x7fffffff) return "gemm: M too large";
    if (g.epilogue < 0 || g.epilogue > VH_EPI_PATCH_SPLIT) return "gemm: unknown epilogue";
    if ((g.epilogue == VH_EPI_LNFOLD || g.epilogue == VH_EPI_LNFOLD_GELU) && (!g.stats || !g.aux)) return "gemm: LNFOLD needs stats and c (aux)";
    if ((g.epilogue == VH_EPI_RESID_LN || g.epilogue == VH_EPI_RESID_SPLIT) && (!g.out16 || !g.partials || g.N % 256))
        return "gemm: RESID_LN / RESID_SPLIT need out16, partials and N % 256 == 0";
    if (g.epilogue == VH_EPI_PATCH && (!g.aux || g.aux_i <= 0)) return "gemm: EPI_PATCH needs pos-emb and patches/image";
    if (g.epilogue == VH_EPI_PATCH_SPLIT && (!g.aux || g.aux_i <= 0 || !g.out16 || !g.partials || g.N % 256 || g.prow < 0))
        return "gemm: EPI_PATCH_SPLIT needs pos-emb, patches/image, the lo plane, partials and N % 256 == 0";
    if (g.dtype != VH_DTYPE_BF16 && g.dtype != VH_DTYPE_FP16) return "gemm: dtype";
    if (g.variant < 0 || g.variant > 7) return "gemm: variant";
    if (g.variant == 3 || g.variant == 4) return "gemm: variants 3/4 (BK=32 pipeline) were removed";
    if (!g.a || !g.w || !g.bias || !g.out) return "gemm: null pointer";
    return nullptr;
}

hipError_t launch_gemm(const GemmArgs& g, hipStream_t s) {
    if (gemm_check(g)) return hipErrorInvalidValue;
    const int variant = g.variant ? g.variant : gemm_pick_variant(g.M, g.N, g.epilogue);
    if ((g.out_tiled || g.ab_tiled) && variant != 6) return hipErrorInvalidValue;   // tiled layouts exist in the persistent form only
    if ((g.tile_count || g.tile_begin) && variant != 5 && variant != 7) return hipErrorInvalidValue;  // tile ranges: ping-pong forms only
    return g.dtype == VH_DTYPE_BF16 ? launch_epi<BF16>(g, variant, s) : launch_epi<FP16>(g, variant, s);
}

}  // namespace vh
