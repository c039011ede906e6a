// gemm_epilogue.h — the fused epilogues shared by every GEMM kernel variant.
//
// Accumulator layout (transposed-tile orientation, see kernels_gemm.hip): for the 16x16 tile
// (mi, ni) of a wave, lane l holds rows m = m0 + 16*mi + (l & 15) and the 4 consecutive columns
// n = n0 + 16*ni + 4*(l >> 4) + {0,1,2,3}.
#pragma once

#include "vh_common.h"

namespace vh {

// Phi(v) = 0.5 * erfc(-v / sqrt(2)) with erfc from Abramowitz & Stegun 7.1.26
// (|abs error| <= 1.5e-7 on erf, i.e. below fp32 rounding of the surrounding arithmetic and four
// orders of magnitude below the 16-bit rounding of the GELU output).  ~17 VALU ops per element
// instead of ~45 for libm's erff: the fc1 epilogue evaluates 128 of these per lane per tile.
__device__ __forceinline__ float gelu_fast(float v) {
    // with x = |v|/sqrt(2): h = 0.5*erfc(x) = 0.5*poly(t)*t*exp(-x^2), t = 1/(1 + 0.3275911 x);
    // gelu(v) = v*Phi(v) = max(v,0) - |v|*h   (both signs; no cancellation for v << 0).  13 VALU ops.
    const float u = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.2316418882f, u, 1.0f));
    float p = fmaf(0.5307027145f, t, -0.7265760135f);
    p = fmaf(p, t, 0.7107068705f);
    p = fmaf(p, t, -0.142248368f);
    p = fmaf(p, t, 0.127414796f);
    p *= t;
    const float h = p * __builtin_amdgcn_exp2f(v * v * -0.72134752044f);
    return fmaf(-u, h, fmaxf(v, 0.f));
}

template <typename T, int EPI, int MI, int NI, bool GUARD>
__device__ __forceinline__ void gemm_epilogue_impl(const f32x4 (&acc)[MI][NI], const float* __restrict__ bias,
                                              void* __restrict__ outp, int M, int N, int m0, int n0,
                                              const float* __restrict__ aux, int aux_i) {
    using elem = typename T::elem;
    f32x4 bv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + ni * 16;
        bv[ni] = (!GUARD || n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + mi * 16;
        if (GUARD && m >= M) continue;
        int64_t orow = m;
        const float* posrow = nullptr;
        if constexpr (EPI == VH_EPI_PATCH) {
            const int img = m / aux_i, p = m - img * aux_i;
            orow = (int64_t)img * (aux_i + 1) + 1 + p;
            posrow = aux + (int64_t)(1 + p) * N;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 16;
            if (GUARD && n >= N) continue;
            f32x4 v = acc[mi][ni] + bv[ni];
            if constexpr (EPI == VH_EPI_BIAS) {
                *(typename T::vec4*)((elem*)outp + orow * N + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            } else if constexpr (EPI == VH_EPI_BIAS_GELU) {
                *(typename T::vec4*)((elem*)outp + orow * N + n) =
                    pack4<T>(gelu_fast(v[0]), gelu_fast(v[1]), gelu_fast(v[2]), gelu_fast(v[3]));
            } else if constexpr (EPI == VH_EPI_BIAS_RESID) {
                f32x4* p = (f32x4*)((float*)outp + orow * N + n);
                *p = *p + v;
            } else if constexpr (EPI == VH_EPI_BIAS_F32) {
                *(f32x4*)((float*)outp + orow * N + n) = v;
            } else {  // VH_EPI_PATCH
                *(f32x4*)((float*)outp + orow * N + n) = v + *(const f32x4*)(posrow + n);
            }
        }
    }
}

// ---- LDS-staged epilogue for full tiles -----------------------------------------------------------
// The accumulator layout gives a lane 4 consecutive columns of 16 different rows, so a direct store
// instruction touches 16 rows x 32 B (16-bit out) or x 64 B (fp32): partial lines, measured 3.5x
// slower than full-line stores on the QKV shape.  Instead every wave transposes its own TM x 64 sub-tile
// through a PRIVATE 16 KiB slice of the (now idle) operand stages and writes/reads HBM in whole
// 128-B (16-bit) or 256-B (fp32) row segments, 16 B per lane.  Only one workgroup barrier is needed
// (before the first LDS write: other waves may still be reading the last stage); write -> read-back is
// wave-private.  `sw` = this wave's slice, m_w/n_w = first row/column of the wave's sub-tile.
// SMI = 16-row blocks staged per pass for 16-bit output (slice = SMI*2 KiB per wave); fp32 stages SMI/2.
template <typename T, int EPI, int MI, int NI, int SMI = MI>
__device__ __forceinline__ void gemm_epilogue_staged(const f32x4 (&acc)[MI][NI], const float* __restrict__ bias,
                                                     void* __restrict__ outp, int N, int m_w, int n_w, int lane,
                                                     char* sw) {
    static_assert(NI == 4, "staged epilogue assumes a 64-column wave tile");
    static_assert(MI % SMI == 0 && SMI % 2 == 0, "slice must divide the wave tile");
    using elem = typename T::elem;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 bv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bv[ni] = *(const f32x4*)(bias + n_w + ni * 16 + fq * 4);

    if constexpr (EPI == VH_EPI_BIAS || EPI == VH_EPI_BIAS_GELU) {
        // rows of 64 x 16-bit = 128 B = 8 chunks of 16 B; chunk c of row r lives at chunk c ^ (r & 7)
        const int rr = lane >> 3, pc = lane & 7;
#pragma unroll
        for (int h = 0; h < MI / SMI; ++h) {
#pragma unroll
            for (int mi = 0; mi < SMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    f32x4 v = acc[h * SMI + mi][ni] + bv[ni];
                    if constexpr (EPI == VH_EPI_BIAS_GELU) {
                        v[0] = gelu_fast(v[0]); v[1] = gelu_fast(v[1]); v[2] = gelu_fast(v[2]); v[3] = gelu_fast(v[3]);
                    }
                    const int c = ni * 2 + (fq >> 1);
                    *(typename T::vec4*)(sw + r * 128 + ((c ^ (r & 7)) << 4) + (fq & 1) * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            }
#pragma unroll
            for (int i = 0; i < SMI * 2; ++i) {
                const int r = i * 8 + rr;
                const u32x4 v = *(const u32x4*)(sw + r * 128 + (pc << 4));
                const int n = n_w + ((pc ^ (r & 7)) << 3);
                *(u32x4*)((elem*)outp + (int64_t)(m_w + h * SMI * 16 + r) * N + n) = v;
            }
        }
    } else {
        // fp32 out: rows of 64 floats = 256 B = 16 chunks, chunk ^ (r & 15); SMI/2 row blocks per pass
        constexpr int FMI = SMI / 2;
        const int rr = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int h = 0; h < MI / FMI; ++h) {
#pragma unroll
            for (int mi = 0; mi < FMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *(f32x4*)(sw + r * 256 + (((ni * 4 + fq) ^ (r & 15)) << 4)) = acc[h * FMI + mi][ni] + bv[ni];
            }
#pragma unroll
            for (int i = 0; i < FMI * 4; ++i) {
                const int r = i * 4 + rr;
                f32x4 v = *(const f32x4*)(sw + r * 256 + (pc << 4));
                const int n = n_w + ((pc ^ (r & 15)) << 2);
                f32x4* p = (f32x4*)((float*)outp + (int64_t)(m_w + h * FMI * 16 + r) * N + n);
                if constexpr (EPI == VH_EPI_BIAS_RESID) v = v + *p;
                *p = v;
            }
        }
    }
}

// Full tiles (the common case) take the staged, unpredicated path; ragged tiles and the patch-row remap
// store directly with per-element predicates.  `smem`/`wave_slice_bytes`: the kernel's dynamic LDS, which
// must hold NW * SMI * 2 KiB from `smem` on.  With BARRIER it contains a workgroup barrier (needed when other
// waves may still be reading the operand stages): call it from uniform control flow only.
template <typename T, int EPI, int MI, int NI, int SMI = MI, bool BARRIER = true>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[MI][NI], const float* __restrict__ bias,
                                              void* __restrict__ outp, int M, int N, int m_w, int n_w, int lane,
                                              const float* __restrict__ aux, int aux_i, bool tile_is_full, char* smem,
                                              int wave) {
    if constexpr (EPI != VH_EPI_PATCH) {
        if (tile_is_full) {
            if constexpr (BARRIER) __syncthreads();
            gemm_epilogue_staged<T, EPI, MI, NI, SMI>(acc, bias, outp, N, m_w, n_w, lane, smem + wave * (SMI * 16 * 128));
            return;
        }
    }
    const int m0 = m_w + (lane & 15), n0 = n_w + (lane >> 4) * 4;
    if (tile_is_full) gemm_epilogue_impl<T, EPI, MI, NI, false>(acc, bias, outp, M, N, m0, n0, aux, aux_i);
    else gemm_epilogue_impl<T, EPI, MI, NI, true>(acc, bias, outp, M, N, m0, n0, aux, aux_i);
}

}  // namespace vh
