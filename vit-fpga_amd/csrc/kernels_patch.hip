// kernels_patch.hip — patch embedding as ONE kernel: the conv-as-GEMM with the patch gather inside its A loader (round 4).
//
//   hi, lo, partial sums = PATCH_SPLIT epilogue( im2col(image)[B*NP, KP] x Wp[D, KP]^T )          (gemm_epilogue.h)
//
// The patch matrix never exists in memory.  With NHWC fp32 images a patch row (patch x channels values) is contiguous, the K
// order of the permuted patch kernel is (ky, kx, c), so an 8-element chunk of a GEMM row is 8 consecutive floats of the image
// as long as 8 divides patch x channels (48 for 16 x 16 x 3): GEMM row m = (image b, patch row py, patch column px), chunk
// k0 = 8 j  ->  image[b][py * patch + k0 / (patch*ch)][px * patch * ch + k0 % (patch*ch) ...+7].  What LDS-DMA cannot do is CONVERT
// (the image is fp32, the MFMA operand 16 bit), so the A operand takes the register path: two 16-byte loads per chunk, one
// packed conversion, one ds_write_b128 into the same XOR-swizzled 128-byte-row image the DMA'd operands use; W stays on
// global_load_lds.  Register staging in the T14 order: the loads of K-tile k+1 are issued before K-tile k is multiplied and
// written to the other stage behind it.  128 x 128 tiles, four waves (64 x 64 each, 64 accumulator registers: room for the 32
// staging registers), two workgroups per CU.
//
// Same arithmetic as im2col_kernel + the GEMM: the same RNE conversion of every pixel, the same k order inside every MFMA,
// the same epilogue -- logits are bit-identical to the two-kernel path (tests/test_gpu_vit.py).
#include <cstdlib>

#include "gemm_epilogue.h"
#include "vh_kernels.h"

namespace vh {

template <typename T>
__global__ void __launch_bounds__(256, 2)
patch_gemm_fused_kernel(const float* __restrict__ img, const typename T::elem* __restrict__ W, const float* __restrict__ bias,
                        void* __restrict__ hi, void* __restrict__ lo, float* __restrict__ partials, const float* __restrict__ pos,
                        int M, int N, int K, int image, int patch, int ch, int tiles_m, int tiles_n, int64_t prow) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int BM = 128, BN = 128, NW = 4, WN = 2, BK = 64;
    constexpr int STAGE_BYTES = (BM + BN) * 128;
    constexpr int MI = 4, NI = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int nwg = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    const int tile_m = wg / tiles_n, tile_n = wg - tile_m * tiles_n;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int lr = lane >> 3, pc = lane & 7;      // row inside an 8-row group, physical 16-byte slot

    // ---- A: this thread's four chunks = rows (i * 4 + wave) * 8 + lr of the tile, logical chunk lc = pc ^ lr -------------------
    const int g = image / patch, np = g * g, prc = patch * ch;   // patches per side / image, floats per patch row
    const int lc = pc ^ lr;
    const float* arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = tile_m * BM + (i * NW + wave) * 8 + lr;
        m = m < M ? m : M - 1;
        const int b = m / np, p = m - b * np, py = p / g, px = p - py * g;
        arow[i] = img + (((int64_t)b * image + py * patch) * image + px * patch) * ch;
    }
    const int img_row = image * ch;                               // floats per image row
    f32x4 areg[4][2];
    auto load_a = [&](int kt) {
        const int k0 = kt * BK + lc * 8;                          // first of the chunk's 8 k-elements
        const int ky = k0 / prc, off = ky * img_row + (k0 - ky * prc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            areg[i][0] = *(const f32x4*)(arow[i] + off);
            areg[i][1] = *(const f32x4*)(arow[i] + off + 4);
        }
    };
    auto write_a = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const typename T::vec4 l = pack4<T>(areg[i][0][0], areg[i][0][1], areg[i][0][2], areg[i][0][3]);
            const typename T::vec4 h = pack4<T>(areg[i][1][0], areg[i][1][1], areg[i][1][2], areg[i][1][3]);
            const vec8 v = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
            *(vec8*)(smem + stage * STAGE_BYTES + ((i * NW + wave) * 8 + lr) * 128 + pc * 16) = v;
        }
    };
    // ---- W: LDS-DMA, four 1-KiB pieces per wave and K-tile (rows BM.. of the stage) ----------------------------------------------
    const elem* wsrc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row = tile_n * BN + (i * NW + wave) * 8 + lr;
        row = row < N ? row : N - 1;
        wsrc[i] = W + (int64_t)row * K + lc * 8;
    }
    auto issue_w = [&](int stage, int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wsrc[i] + kt * BK),
                                             (void __attribute__((address_space(3)))*)(smem + stage * STAGE_BYTES + BM * 128 + (i * NW + wave) * 1024),
                                             16, 0, 0);
    };

    const int frow = lane & 15, fq = lane >> 4;
    int offk[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offk[ks] = frow * 128 + ((((ks << 2) | fq) ^ (frow & 7)) << 4);
    const int xbase = wm * 64 * 128;
    const int wbase = BM * 128 + wn * 64 * 128;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    load_a(0);
    issue_w(0, 0);
    write_a(0);                                            // (the compiler waits for the loads it knows)
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();   // K-tile kt is in LDS for every wave (A written, W landed); everyone is done reading the other stage
        if (kt + 1 < nk) {
            load_a(kt + 1);                                // in flight under this K-tile's MFMAs
            issue_w((kt + 1) & 1, kt + 1);
        }
        const char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 xf[MI], wf[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(st + wbase + ni * 2048 + offk[ks]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) xf[mi] = *(const vec8*)(st + xbase + mi * 2048 + offk[ks]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = T::mfma16(wf[ni], xf[mi], acc[mi][ni]);
        }
        if (kt + 1 < nk) write_a((kt + 1) & 1);            // the other stage: its last reader passed the barrier above
    }

    const EpiArgs e{bias, hi, M, N, pos, np, nullptr, lo, partials, prow};
    gemm_epilogue<T, VH_EPI_PATCH_SPLIT, MI, NI>(acc, e, tile_m * BM + wm * 64, tile_n * BN + wn * 64, lane, true, (tile_m + 1) * BM <= M, smem, wave);
}

// patch embedding straight into the split residual planes; images NHWC fp32.  Needs 8 | patch * channels, 64 | KP, 128 | dim.
bool patch_fused_supported(int image, int patch, int channels, int dim) {
    return image % patch == 0 && (patch * channels) % 8 == 0 && (patch * patch * channels) % 64 == 0 && dim % 128 == 0;
}
hipError_t launch_patch_fused(const float* images, int batch, int image, int patch, int channels, const void* wp16, const float* bias,
                              const float* pos, void* hi, void* lo, float* partials, int64_t prow, int dim, int dtype, hipStream_t s) {
    if (!patch_fused_supported(image, patch, channels, dim) || batch <= 0) return hipErrorInvalidValue;
    const int g = image / patch, M = batch * g * g, K = patch * patch * channels;
    const int tiles_m = (M + 127) / 128, tiles_n = dim / 128;
    constexpr size_t lds = 2 * 256 * 128;
    const dim3 grid((unsigned)(tiles_m * tiles_n)), block(256);
    if (dtype == VH_DTYPE_BF16) {
        auto k = patch_gemm_fused_kernel<BF16>;
        static LdsDone done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, block, lds, s, images, (const BF16::elem*)wp16, bias, hi, lo, partials, pos, M, dim, K, image, patch, channels,
                           tiles_m, tiles_n, prow);
    } else if (dtype == VH_DTYPE_FP16) {
        auto k = patch_gemm_fused_kernel<FP16>;
        static LdsDone done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, grid, block, lds, s, images, (const FP16::elem*)wp16, bias, hi, lo, partials, pos, M, dim, K, image, patch, channels,
                           tiles_m, tiles_n, prow);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace vh
