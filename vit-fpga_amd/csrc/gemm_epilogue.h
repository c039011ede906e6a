// gemm_epilogue.h — the fused epilogues shared by every GEMM kernel variant.
//
// Accumulator layout (transposed-tile orientation, see kernels_gemm.hip): for the 16x16 tile
// (mi, ni) of a wave, lane l holds rows m = m0 + 16*mi + (l & 15) and the 4 consecutive columns
// n = n0 + 16*ni + 4*(l >> 4) + {0,1,2,3}.
//
// Epilogues (VH_EPI_* in include/vithip.h):
//   BIAS / BIAS_GELU        out16 = [gelu](acc + bias)
//   BIAS_RESID / BIAS_F32   out32 (+)= acc + bias
//   PATCH                   out32[row(m)] = acc + bias + pos[tok(m)]
//   LNFOLD / LNFOLD_GELU    out16 = [gelu]( rstd_m * (acc - mean_m * c_n) + d_n )
//        LayerNorm folded into the GEMM: with W' = gamma o W the product LN(x) W^T equals
//        rstd * (x W'^T - mean * c) + d,  c_n = sum_k W'_nk,  d_n = sum_k beta_k W_nk + b_n, so the
//        GEMM runs on the RAW (16-bit rounded) residual rows and the per-row statistics enter here.
//   RESID_SPLIT             the residual stream as two 16-bit planes (x = hi + lo, hi = T(x) IS the next GEMM's operand,
//        lo = T(x - hi): 16-17 significant bits for bf16, 22 for fp16): (hi, lo) += acc + bias, plus the per-64-column row
//        sums of RESID_LN.  Moves 4 B per element each way where RESID_LN moves 4 B + its 2 B copy; a lane owns 8
//        consecutive columns so that both planes move 16 B per lane.
//   RESID_LN                out32 += acc + bias; also writes the 16-bit copy of the updated rows and,
//        per 64-column block, the row's (sum, sum of squares) — the producer side of LNFOLD.  Together
//        they remove the stand-alone LayerNorm pass (read 310 MB + write 155 MB per LN at ViT-B b512).
#pragma once

#include "vh_common.h"

namespace vh {

#ifndef VH_EPI_ABL
#define VH_EPI_ABL 0   // timing-only ablation builds (tools): 1 no partial-sum stores, 2 no 16-bit copy, 4 no fp32 store, 8 no residual loads,
                       // 16-bit forms: 16 no global stores, 32 no GELU arithmetic, 64 no epilogue at all (accumulators kept alive)
#endif
#ifndef VH_EPI_NT
#define VH_EPI_NT 1   // 16-bit / e4m3 result stores non-temporal (A/B: -DVH_EPI_NT=0 builds plain stores)
#endif
template <typename V>
__device__ __forceinline__ void epi_store(V v, V* p) {
    if constexpr (VH_EPI_NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

struct EpiArgs {
    const float* bias;   // [N]  (LNFOLD: d_n)
    void* out;           // primary output
    int M, N;
    const float* aux;    // PATCH: pos-emb [tokens, N]; LNFOLD: c_n [N]
    int aux_i;           // PATCH: patches per image
    const float* stats;  // LNFOLD: [M][2] = (mean, rstd) of every row
    void* out16;         // RESID_LN: 16-bit copy of the updated residual [M, N]
    float* partials;     // RESID_LN: [N/64][M][2] = per-64-column (sum, sum of squares)
    int64_t prow;        // PATCH_SPLIT: row stride of `partials` (token rows); the other forms use M
};

constexpr bool epi_is_16bit(int epi) {
    return epi == VH_EPI_BIAS || epi == VH_EPI_BIAS_GELU || epi == VH_EPI_LNFOLD || epi == VH_EPI_LNFOLD_GELU;
}
constexpr bool epi_has_gelu(int epi) { return epi == VH_EPI_BIAS_GELU || epi == VH_EPI_LNFOLD_GELU; }
constexpr bool epi_is_lnfold(int epi) { return epi == VH_EPI_LNFOLD || epi == VH_EPI_LNFOLD_GELU; }

// exact-erf GELU: with x = |v|/sqrt(2), h = 0.5*erfc(x) = 0.5*poly(t)*t*exp(-x^2), t = 1/(1 + 0.3275911 x)
// (Abramowitz & Stegun 7.1.26, |abs error| <= 1.5e-7 on erf: below fp32 rounding of the surrounding
// arithmetic, four orders below the 16-bit rounding of the output);  gelu(v) = v*Phi(v) = max(v,0) - |v|*h
// for both signs, no cancellation for v << 0.  13 VALU ops instead of ~45 for libm's erff: the fc1 epilogue
// evaluates 128 of these per lane per tile.
__device__ __forceinline__ float gelu_fast(float v) {
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

// The same function for a quad of values, transcendental-free and (round 4) without per-value clamps:
//     w   = clamp01(1 - v^2 / 4.5^2)                (ONE packed fma with the instruction's clamp modifier: 0 beyond |v| = 4.5)
//     R   = P(w), degree DEG                        (R = (Phi(v) - 1/2) / v on |v| <= 4.5, and its value at 4.5 beyond)
//     Phi = clamp01(1/2 + v R)                      (again the modifier: beyond 4.5 the line 1/2 + v R(0) leaves [0, 1] and saturates)
//     gelu(v) = v Phi
// i.e. DEG + 4 packed operations per PAIR of values and nothing else (round 1's form: two v_med3 + 13 packed operations for degree
// 10 in t = 2 c^2 / 4.5^2 - 1).  The fc1 epilogue is VALU-throughput-bound on exactly these instructions (DESIGN.md 4.1,
// tools/probe_valu.hip: 3.3 cycles per packed fp32 instruction with two waves on the SIMD, whatever it computes).
// P is a weighted least-squares / reweighted minimax fit of (Phi(c) - 1/2)/c on [0, 4.5] with the error weighted by c^2 (what
// gelu = v (1/2 + v R) sees), written in powers of w: Horner in w has its rounding where v^2 is SMALL (w near 1) and is exact to
// a few ulps where the weight is large (w near 0).  |gelu error|, fp32 Horner with fma, every v:
//     DEG 9: 7.7e-6  (fp16 results; round 1's degree-10 fit in t: 1.8e-5)     DEG 8: 3.4e-5  (bf16 results: half a bf16 ulp of 0.017)
//     DEG 6: 5.7e-4  (e4m3 results: a quarter of e4m3's smallest step)
// (tools: the fit is reproduced by tools/gelu_fit.py.)  Beyond the clamp the result is v or 0 exactly (true value within 3.4e-6 |v|).
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int DEG> struct GeluW;
template <> struct GeluW<9> { static constexpr float a[10] = {1.111100093e-01f, 5.562581867e-02f, 3.875113651e-02f, 7.441916317e-02f, -2.459328771e-01f,
                                                                1.095633626e+00f, -2.419960260e+00f, 3.303298712e+00f, -2.415727854e+00f, 8.016663194e-01f}; };
template <> struct GeluW<8> { static constexpr float a[9] = {1.111119911e-01f, 5.524744466e-02f, 5.048053339e-02f, -6.402773410e-02f, 5.648611188e-01f,
                                                               -1.549515009e+00f, 2.612281084e+00f, -2.240322828e+00f, 8.585962057e-01f}; };
template <> struct GeluW<6> { static constexpr float a[7] = {1.111383960e-01f, 5.238987878e-02f, 9.770081192e-02f, -3.239865899e-01f, 1.061976552e+00f,
                                                               -1.360716462e+00f, 7.577903271e-01f}; };
// d = clamp01(a * b + c) on a pair: v_pk_fma_f32 with the clamp modifier (no builtin carries it; a NaN comes out as 0 under the
// kernels' DX10_CLAMP mode, and the final v * Phi makes it a NaN again)
__device__ __forceinline__ f32x2 pk_fma_clamp01(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the two pairs of a quad advance in step: a packed fp32 op feeding the next instruction costs a wait state,
// two independent chains in alternation cost none
template <int DEG>
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 v) {
    constexpr const float* a = GeluW<DEG>::a;
    f32x2 x[2] = {f32x2{v[0], v[1]}, f32x2{v[2], v[3]}}, w[2], p[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) w[h] = x[h] * x[h];
#pragma unroll
    for (int h = 0; h < 2; ++h) w[h] = pk_fma_clamp01(w[h], f32x2{-4.938271605e-02f, -4.938271605e-02f}, f32x2{1.f, 1.f});
#pragma unroll
    for (int h = 0; h < 2; ++h) p[h] = __builtin_elementwise_fma(w[h], f32x2{a[DEG], a[DEG]}, f32x2{a[DEG - 1], a[DEG - 1]});
#pragma unroll
    for (int k = DEG - 2; k >= 0; --k)
#pragma unroll
        for (int h = 0; h < 2; ++h) p[h] = __builtin_elementwise_fma(p[h], w[h], f32x2{a[k], a[k]});
#pragma unroll
    for (int h = 0; h < 2; ++h) p[h] = pk_fma_clamp01(x[h], p[h], f32x2{0.5f, 0.5f});
#pragma unroll
    for (int h = 0; h < 2; ++h) x[h] = x[h] * p[h];
    return f32x4{x[0][0], x[0][1], x[1][0], x[1][1]};
}
// degree by result type: what the rounding of the stored value hides (fp16 11 bits, bf16 8, e4m3 4)
template <typename T> struct GeluDeg { static constexpr int value = 9; };
template <> struct GeluDeg<BF16> { static constexpr int value = 8; };
template <> struct GeluDeg<E4M3> { static constexpr int value = 6; };

// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15), result in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v);  // row_ror:8
    v += dpp_mov<0x124>(v);  // row_ror:4
    v += dpp_mov<0x122>(v);  // row_ror:2
    v += dpp_mov<0x121>(v);  // row_ror:1
    return v;
}

// value of one accumulator quad for the 16-bit epilogues; TR = the type the caller rounds the quad to (selects the GELU degree)
template <int EPI, typename TR>
__device__ __forceinline__ f32x4 epi_value16(f32x4 acc, f32x4 bv, f32x4 cv, float mean_rstd, float rstd) {
    f32x4 v;
    if constexpr (epi_is_lnfold(EPI)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(rstd, acc[j], fmaf(-mean_rstd, cv[j], bv[j]));
    } else {
        v = acc + bv;
    }
    if constexpr (epi_has_gelu(EPI) && !(VH_EPI_ABL & 32)) {
        v = gelu_poly4<GeluDeg<TR>::value>(v);
    }
    return v;
}

// ---- direct (unstaged) epilogue: ragged-N tiles, the patch-row remap, small kernels -----------------------
template <typename T, int EPI, int MI, int NI, bool GUARD>
__device__ __forceinline__ void gemm_epilogue_impl(const f32x4 (&acc)[MI][NI], const EpiArgs& e, int m0, int n0) {
    static_assert(EPI != VH_EPI_RESID_LN && EPI != VH_EPI_RESID_SPLIT && EPI != VH_EPI_PATCH_SPLIT, "RESID_LN / RESID_SPLIT / PATCH_SPLIT need the staged path (N % tile == 0)");
    using elem = typename T::elem;
    const int M = e.M, N = e.N;
    f32x4 bv[NI], cv[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + ni * 16;
        const bool ok = !GUARD || n < N;
        bv[ni] = ok ? *(const f32x4*)(e.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (epi_is_lnfold(EPI)) cv[ni] = ok ? *(const f32x4*)(e.aux + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        else cv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + mi * 16;
        if (GUARD && m >= M) continue;
        int64_t orow = m;
        const float* posrow = nullptr;
        if constexpr (EPI == VH_EPI_PATCH) {
            const int img = m / e.aux_i, p = m - img * e.aux_i;
            orow = (int64_t)img * (e.aux_i + 1) + 1 + p;
            posrow = e.aux + (int64_t)(1 + p) * N;
        }
        float rstd = 0.f, mr = 0.f;
        if constexpr (epi_is_lnfold(EPI)) {
            const float2 st = *(const float2*)(e.stats + 2 * (int64_t)m);
            rstd = st.y;
            mr = st.x * st.y;
        }
        // what the row ADDS (residual / position embedding) is loaded for all NI column blocks before the first store of
        // the row: a load placed after a store to the same array cannot be moved above it
        f32x4 addv[NI];
        if constexpr (EPI == VH_EPI_BIAS_RESID || EPI == VH_EPI_PATCH) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const int n = n0 + ni * 16;
                const bool ok = !GUARD || n < N;
                if constexpr (EPI == VH_EPI_BIAS_RESID) addv[ni] = ok ? *(const f32x4*)((const float*)e.out + orow * N + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                else addv[ni] = ok ? *(const f32x4*)(posrow + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n0 + ni * 16;
            if (GUARD && n >= N) continue;
            if constexpr (epi_is_16bit(EPI)) {
                const f32x4 v = epi_value16<EPI, T>(acc[mi][ni], bv[ni], cv[ni], mr, rstd);
                *(typename T::vec4*)((elem*)e.out + orow * N + n) = pack4<T>(v[0], v[1], v[2], v[3]);
            } else {
                const f32x4 v = acc[mi][ni] + bv[ni];
                f32x4* p = (f32x4*)((float*)e.out + orow * N + n);
                if constexpr (EPI == VH_EPI_BIAS_F32) *p = v;
                else *p = v + addv[ni];  // VH_EPI_BIAS_RESID, VH_EPI_PATCH
            }
        }
    }
}

// ---- LDS-staged epilogue -------------------------------------------------------------------------------
// The accumulator layout gives a lane 4 consecutive columns of 16 different rows, so a direct store
// instruction touches 16 rows x 32 B (16-bit out) or x 64 B (fp32): partial lines, measured 3.5x slower than
// full-line stores on the QKV shape.  Instead every wave transposes its own TM x 64 sub-tile through a PRIVATE
// slice of the (now idle) operand stages and accesses HBM in whole 128-B (16-bit) or 256-B (fp32) row
// segments, 16 B per lane.  Write -> read-back is wave-private (no barrier).  `sw` = this wave's slice,
// m_w / n_w = first row / column of the wave's sub-tile; rows >= M are skipped (ragged last M-tile).
// SMI = 16-row blocks staged per pass for 16-bit output (slice = SMI * 2 KiB per wave); fp32 stages SMI/2.
// MFULL: every row of the wave's sub-tile is inside M (all but the last row tile of a GEMM): the 16-bit store loop then
// carries no per-row predicate, so the read-backs of a pass are issued together instead of one per predicated block.
// prio_grp (>= 0: the wave group 0 / 1 of the ping-pong kernel; -1: off), 16-bit forms with VH_EPI_PRIO: the two waves of a
// SIMD run this epilogue at the same time and the vector ALU is arbitrated by priority, then AGE -- the older wave (group 0) got
// nearly the full single-wave rate, finished early and waited ~3 us at the next tile's first barrier while group 1 finished ALONE
// at the single-wave rate (4.7 cycles per packed instruction where two waves together manage one per 3.3: tools/probe_valu.hip,
// profiles/r04_c_gemm_anatomy_tile8.txt).  Alternating the priority pass by pass lets both advance and finish together.
#ifndef VH_EPI_PRIO
#define VH_EPI_PRIO 1
#endif
// OTILED (16-bit GELU forms, full tiles): the result goes to the 16-ROW-BLOCKED layout of the MLP hidden activation (below) straight
// from the registers, without the LDS transposition.
// The LN-fold epilogues' constants out of the wave's staging slice, where the K loop's LDS-DMA put them (`cpre`): 8 rows' (mean, rstd)
// at byte 8 r, d_n at 1024 + 4 n, c_n at 1280 + 4 n.  Inline asm: to the compiler an ordinary LDS read (or anything that may touch
// memory) behind a pending LDS-DMA is a possible alias, which it answers with vmcnt(0) -- draining the next tile's operand DMAs, the
// very wait this prefetch is there to avoid.  The K loop's counted waits retired these three DMAs K-tiles ago.
__device__ __forceinline__ void epi_consts_from_lds(const char* sw, int frow, int fq, f32x4 (&bv)[4], f32x4 (&cv)[4], float2 (&lnst)[8]) {
    const uint32_t la = (uint32_t)(uintptr_t)sw + (uint32_t)((fq * 4) * 4), lr = (uint32_t)(uintptr_t)sw + (uint32_t)(frow * 8);
    asm volatile("ds_read_b128 %0, %8 offset:1024\n\tds_read_b128 %1, %8 offset:1088\n\tds_read_b128 %2, %8 offset:1152\n\tds_read_b128 %3, %8 offset:1216\n\t"
                 "ds_read_b128 %4, %8 offset:1280\n\tds_read_b128 %5, %8 offset:1344\n\tds_read_b128 %6, %8 offset:1408\n\tds_read_b128 %7, %8 offset:1472"
                 : "=&v"(bv[0]), "=&v"(bv[1]), "=&v"(bv[2]), "=&v"(bv[3]), "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3]) : "v"(la));
    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\tds_read_b64 %3, %8 offset:384\n\t"
                 "ds_read_b64 %4, %8 offset:512\n\tds_read_b64 %5, %8 offset:640\n\tds_read_b64 %6, %8 offset:768\n\tds_read_b64 %7, %8 offset:896\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(lnst[0]), "=&v"(lnst[1]), "=&v"(lnst[2]), "=&v"(lnst[3]), "=&v"(lnst[4]), "=&v"(lnst[5]), "=&v"(lnst[6]), "=&v"(lnst[7]) : "v"(lr));
    // (the wait covers the b128 reads too: LDS operations return in order; their uses are tied to it here)
    asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(cv[0]), "+v"(cv[1]), "+v"(cv[2]), "+v"(cv[3]));
}

// cpre (LN-fold forms of the persistent GEMM, round 4): the epilogue's constants -- the (mean, rstd) pairs of the wave's MI * 16 rows, d_n
// and c_n of its 64 columns -- were put into the wave's staging slice `sw` by LDS-DMA during the K loop (kernels_gemm5.hip
// issue_consts: rows at byte 8 r, d at 1024 + 4 n, c at 1280 + 4 n), so they come from LDS here instead of being loaded from global
// memory behind the next tile's operand DMAs (a microsecond per tile before the first value could be computed).  They are read
// into registers before the slice is used for staging.
template <typename T, int EPI, int MI, int NI, int SMI = MI, bool MFULL = false, bool OTILED = false>
__device__ __forceinline__ void gemm_epilogue_staged(const f32x4 (&acc)[MI][NI], const EpiArgs& e, int m_w, int n_w,
                                                     int lane, char* sw, int prio_grp = -1, bool cpre = false) {
    static_assert(NI == 4, "staged epilogue assumes a 64-column wave tile");
    static_assert(MI % SMI == 0 && SMI % 2 == 0, "slice must divide the wave tile");
    using elem = typename T::elem;
    const int M = e.M, N = e.N;
    const int frow = lane & 15, fq = lane >> 4;
    f32x4 bv[NI], cv[NI];
    float2 lnst[epi_is_lnfold(EPI) ? MI : 1];   // LN-fold forms: (mean, rstd) of the lane's MI rows, all up front (read inside the row loop they
                                                // would sit behind the previous pass's stores: one exposed load latency per pass)
    auto consts_from_global = [&]() {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bv[ni] = *(const f32x4*)(e.bias + n_w + ni * 16 + fq * 4);
            if constexpr (epi_is_lnfold(EPI)) cv[ni] = *(const f32x4*)(e.aux + n_w + ni * 16 + fq * 4);
            else cv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (epi_is_lnfold(EPI)) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                int m = m_w + mi * 16 + frow;
                if constexpr (!MFULL) m = m < M ? m : M - 1;
                lnst[mi] = *(const float2*)(e.stats + 2 * (int64_t)m);
            }
        }
    };
    // (plain if / else: with a flag between the two paths the compiler sees a way from the global loads to the asm reads that
    //  overwrite their registers, and guards it with vmcnt(0))
    if constexpr (epi_is_lnfold(EPI) && MI == 8) {
        if (cpre) epi_consts_from_lds(sw, frow, fq, bv, cv, lnst);   // (wave-uniform)
        else consts_from_global();
    } else {
        consts_from_global();
    }

    if constexpr (epi_is_16bit(EPI) && (VH_EPI_ABL & 64)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
    } else if constexpr (epi_is_16bit(EPI) && epi_has_gelu(EPI) && MFULL && OTILED) {
        // The MLP hidden activation h in its TILED layout (round 4): [m / 16][N / 8 chunks][16 rows][8 values] -- the sixteen rows of a
        // 16-lane group are 256 contiguous bytes per chunk, so the result leaves the registers in 16-byte stores of whole 256-byte
        // segments with NO LDS transposition (the staged form below costs 32 ds_write_b64 + 16 ds_read_b128 per wave and tile; a
        // register path into a ROW-major result was measured slower in round 3 because a 16-lane group then touches sixteen rows).
        // A lane holds 4 consecutive columns of column blocks ni = 2j and 2j + 1; one v_permlane16_swap per packed dword (odd 16-lane
        // rows of the first operand <-> even rows of the second) leaves the lanes of an even row with the 8 consecutive columns
        // 16 (2j) + 8 h .. + 7 and those of the odd row beside it with 16 (2j + 1) + 8 h .. + 7 (h = fq >> 1): NATURAL column order
        // inside every chunk, so fc2 multiplies in the same k order as with a row-major h and the logits keep their bits.
        // h has exactly one consumer, fc2's operand DMA (kernels_gemm5.hip AT), which takes per-lane source addresses anyway.
        char* const obase = (char*)e.out + ((int64_t)(m_w >> 4) * (N >> 3) + (n_w >> 3) + (fq & 1) * 2 + (fq >> 1)) * 256 + frow * 16;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            if constexpr (VH_EPI_PRIO == 1) {
                if (prio_grp >= 0 && (mi & 1) == 0) {   // (wave-uniform) the two wave groups alternate, as in the staged form
                    if (((mi >> 1) + prio_grp) & 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
            float rstd = 0.f, mr = 0.f;
            if constexpr (epi_is_lnfold(EPI)) { rstd = lnst[mi].y; mr = lnst[mi].x * lnst[mi].y; }
#pragma unroll
            for (int j = 0; j < NI / 2; ++j) {
                const f32x4 v0 = epi_value16<EPI, T>(acc[mi][2 * j], bv[2 * j], cv[2 * j], mr, rstd);
                const f32x4 v1 = epi_value16<EPI, T>(acc[mi][2 * j + 1], bv[2 * j + 1], cv[2 * j + 1], mr, rstd);
                const u32x2 p0 = __builtin_bit_cast(u32x2, pack4<T>(v0[0], v0[1], v0[2], v0[3]));
                const u32x2 p1 = __builtin_bit_cast(u32x2, pack4<T>(v1[0], v1[1], v1[2], v1[3]));
                const auto sx = __builtin_amdgcn_permlane16_swap(p0[0], p1[0], false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(p0[1], p1[1], false, false);
                epi_store(u32x4{sx[0], sy[0], sx[1], sy[1]}, (u32x4*)(obase + (int64_t)mi * (N >> 3) * 256 + j * 1024));
            }
        }
        if constexpr (VH_EPI_PRIO == 1) { if (prio_grp >= 0) __builtin_amdgcn_s_setprio(0); }
    } else if constexpr (epi_is_16bit(EPI)) {
        // rows of 64 x 16-bit = 128 B = 8 chunks of 16 B; chunk c of row r lives at chunk c ^ (r & 7)
        const int rr = lane >> 3, pc = lane & 7;
        // LNFOLD: the (mean, rstd) pairs of the lane's MI rows are loaded up front -- read inside the loop they sit behind
        // the previous pass's stores (possible alias), one exposed global-load latency per pass
#pragma unroll
        for (int h = 0; h < MI / SMI; ++h) {
            if constexpr (VH_EPI_PRIO == 1) {
                if (prio_grp >= 0) {   // (wave-uniform)
                    if ((h + prio_grp) & 1) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
#pragma unroll
            for (int mi = 0; mi < SMI; ++mi) {
                if constexpr (VH_EPI_PRIO == 2) {   // (A/B: flip per 16-row block instead of per pass)
                    if (prio_grp >= 0) {
                        if ((h * SMI + mi + prio_grp) & 1) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                }
                const int r = mi * 16 + frow;
                float rstd = 0.f, mr = 0.f;
                if constexpr (epi_is_lnfold(EPI)) {
                    const float2 st = lnst[h * SMI + mi];
                    rstd = st.y;
                    mr = st.x * st.y;
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const f32x4 v = epi_value16<EPI, T>(acc[h * SMI + mi][ni], bv[ni], cv[ni], mr, rstd);
                    const int c = ni * 2 + (fq >> 1);
                    *(typename T::vec4*)(sw + r * 128 + ((c ^ (r & 7)) << 4) + (fq & 1) * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
                }
            }
#pragma unroll
            for (int i = 0; i < SMI * 2; ++i) {
                const int r = i * 8 + rr;
                const u32x4 v = *(const u32x4*)(sw + r * 128 + (pc << 4));
                const int n = n_w + ((pc ^ (r & 7)) << 3);
                const int m = m_w + h * SMI * 16 + r;
                if constexpr (VH_EPI_ABL & 16) asm volatile("" :: "v"(v));
                else if constexpr (OTILED) {
                    // q|k|v HEAD-MAJOR, [N / 64][M][64] (round 4; whole tiles): the wave's 64 columns are one head's q, k or v, whose rows
                    // then lie 128 B apart -- this store instruction writes 1 KiB of contiguous memory (8 rows), and attention's operand
                    // DMA reads it the same way (kernels_attn.hip IHM).  Same values, another address.
                    epi_store(v, (u32x4*)((elem*)e.out + ((int64_t)(n_w >> 6) * M + m) * 64 + ((pc ^ (r & 7)) << 3)));
                }
                else if (MFULL || m < M) epi_store(v, (u32x4*)((elem*)e.out + (int64_t)m * N + n));
            }
        }
        if constexpr (VH_EPI_PRIO) { if (prio_grp >= 0) __builtin_amdgcn_s_setprio(0); }
    } else if constexpr (EPI == VH_EPI_RESID_SPLIT && std::is_same<T, E4M3>::value) {
        // split residual of the fp8 path: hi = e4m3(x), one byte, which IS the next GEMM's A operand; lo = bf16(x - hi), so the
        // pair carries 4 + 8 significant bits of x (the same 12 as the bf16 path's planes) in 3 bytes: a residual update
        // moves 3 B per element each way instead of the fp32 array's 4 B plus the 1 B operand copy of RESID_LN.  Staged and
        // read back exactly like the 16-bit form below (8 columns per lane: 8 B of hi, 16 B of lo).
        using lvec8 = typename BF16::vec8;
        constexpr int FMI = SMI / 2, NP = MI / FMI, NL = FMI * 2;
        const int rr = lane >> 3, pc = lane & 7;
        uint8_t* const hi = (uint8_t*)e.out;
        typename BF16::elem* const lo = (typename BF16::elem*)e.out16;
        struct Add8 { u32x2 h; lvec8 l; };
        Add8 xa[2][NL];
        auto where = [&](int h, int i, int& m) {
            m = m_w + h * FMI * 16 + i * 8 + rr;
            return (int64_t)m * N + n_w + 8 * pc;
        };
        auto dec4 = [](uint32_t w) {
            typedef float f32x2_ __attribute__((ext_vector_type(2)));
            const f32x2_ a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
            return f32x4{a[0], a[1], b[0], b[1]};
        };
        auto load_pass = [&](int h, Add8 (&a)[NL]) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                int m;
                const int64_t off = where(h, i, m);
                if ((MFULL || m < M) && !(VH_EPI_ABL & 8)) { a[i].h = *(const u32x2*)(hi + off); a[i].l = *(const lvec8*)(lo + off); }
                else { a[i].h = u32x2{0u, 0u}; a[i].l = lvec8{}; }
            }
        };
        load_pass(0, xa[0]);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            if (h + 1 < NP) load_pass(h + 1, xa[(h + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < FMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *(f32x4*)(sw + r * 256 + (((ni * 4 + fq) ^ (r & 15)) << 4)) = acc[h * FMI + mi][ni] + bv[ni];
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int r = i * 8 + rr;
                const f32x4 v0 = *(const f32x4*)(sw + r * 256 + (((2 * pc) ^ (r & 15)) << 4));
                const f32x4 v1 = *(const f32x4*)(sw + r * 256 + (((2 * pc + 1) ^ (r & 15)) << 4));
                int m;
                const int64_t off = where(h, i, m);
                const bool ok = MFULL || m < M;
                const Add8& a = xa[h & 1][i];
                const f32x4 h0 = dec4(a.h[0]), h1 = dec4(a.h[1]);
                float v[8];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = (j < 4 ? v0[j] : v1[j - 4]) + ((j < 4 ? h0[j] : h1[j - 4]) + (float)a.l[j]);
                    s1 += v[j];
                    s2 = fmaf(v[j], v[j], s2);
                }
                const u32x2 hn{pack4_e4m3(v[0], v[1], v[2], v[3]), pack4_e4m3(v[4], v[5], v[6], v[7])};   // saturating: lo carries what is cut off
                const f32x4 d0 = dec4(hn[0]), d1 = dec4(hn[1]);
                lvec8 ln;
#pragma unroll
                for (int j = 0; j < 8; ++j) ln[j] = (typename BF16::elem)(v[j] - (j < 4 ? d0[j] : d1[j - 4]));
                if (ok && !(VH_EPI_ABL & 4)) { *(u32x2*)(hi + off) = hn; *(lvec8*)(lo + off) = ln; }
                s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);     // quad_perm [1,0,3,2]
                s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);     // quad_perm [2,3,0,1]
                s1 += dpp_mov<0x141>(s1); s2 += dpp_mov<0x141>(s2);   // row_half_mirror
                if (ok && pc == 0 && !(VH_EPI_ABL & 1)) *(float2*)(e.partials + 2 * ((int64_t)(n_w >> 6) * M + m)) = make_float2(s1, s2);
            }
        }
    } else if constexpr (EPI == VH_EPI_RESID_SPLIT || EPI == VH_EPI_PATCH_SPLIT) {
        // split residual: stage acc + bias as fp32 rows (256 B, chunk ^ (r & 15)) like the fp32 form, read back EIGHT columns per
        // lane (two chunks): the hi plane is accessed 16 B per lane / 128 B per row, the one-byte lo plane 8 B per lane / 64 B per row.  What the row
        // ADDS -- its own planes (RESID_SPLIT) or the position embedding's fp32 row (PATCH_SPLIT: the patch embedding lands
        // directly in the split form, on the remapped row m + m / patches + 1, with the first row statistics) -- is loaded for
        // pass h+1 before pass h is processed (same reason as in the fp32 form below).
        using vec8 = typename T::vec8;
        constexpr bool PATCHS = EPI == VH_EPI_PATCH_SPLIT;
        constexpr int FMI = SMI / 2, NP = MI / FMI, NL = FMI * 2;
        const int rr = lane >> 3, pc = lane & 7;
        elem* const hi = (elem*)e.out;
        uint8_t* const lo = (uint8_t*)e.out16;   // one e4m3 byte per element (Lo8<T>, vh_common.h)
        const int64_t prow = PATCHS ? e.prow : (int64_t)M;
        // 8 fp32 addends per lane and line: RESID_SPLIT converts its two 16-bit planes on use, PATCH_SPLIT loads them as is
        struct Add { vec8 h; u32x2 l; f32x4 p0, p1; };
        Add xa[2][NL];
        // PATCH_SPLIT: GEMM row m = image * patches + p lands on token row m + image + 1 and adds pos row 1 + p.  One scalar
        // division per wave tile (m_w is wave-uniform), then a carry per row instead of a vector division per line.
        const int q0 = PATCHS ? m_w / e.aux_i : 0, r0 = PATCHS ? m_w - q0 * e.aux_i : 0;
        auto where = [&](int h, int i, int& m, int64_t& orow, int& p) {   // GEMM row m -> output row, element offset of the lane's 8 columns
            const int dr = h * FMI * 16 + i * 8 + rr;
            m = m_w + dr;
            p = 0;
            if constexpr (PATCHS) {
                int img = q0;
                p = r0 + dr;
                while (p >= e.aux_i) { p -= e.aux_i; ++img; }
                orow = (int64_t)m + img + 1;
            } else {
                orow = m;
            }
            return orow * N + n_w + 8 * pc;
        };
        auto load_pass = [&](int h, Add (&a)[NL]) {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                int m, p;
                int64_t orow;
                const int64_t off = where(h, i, m, orow, p);
                if constexpr (PATCHS) {
                    const float* pr = e.aux + (int64_t)(1 + p) * N + n_w + 8 * pc;   // (a row of the table whatever m is: no bound to check)
                    a[i].p0 = *(const f32x4*)pr;
                    a[i].p1 = *(const f32x4*)(pr + 4);
                } else {
                    if ((MFULL || m < M) && !(VH_EPI_ABL & 8)) { a[i].h = *(const vec8*)(hi + off); a[i].l = *(const u32x2*)(lo + off); }
                    else { a[i].h = vec8{}; a[i].l = u32x2{0u, 0u}; }
                }
            }
        };
        load_pass(0, xa[0]);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            if (h + 1 < NP) load_pass(h + 1, xa[(h + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < FMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *(f32x4*)(sw + r * 256 + (((ni * 4 + fq) ^ (r & 15)) << 4)) = acc[h * FMI + mi][ni] + bv[ni];
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int r = i * 8 + rr;
                const f32x4 v0 = *(const f32x4*)(sw + r * 256 + (((2 * pc) ^ (r & 15)) << 4));
                const f32x4 v1 = *(const f32x4*)(sw + r * 256 + (((2 * pc + 1) ^ (r & 15)) << 4));
                int m, p;
                int64_t orow;
                const int64_t off = where(h, i, m, orow, p);
                const bool ok = MFULL || m < M;   // (whole tiles: no exec-mask branch around every access)
                const Add& a = xa[h & 1][i];
                float v[8], ln[8];
                vec8 hn;
                float s1 = 0.f, s2 = 0.f;
                f32x4 l0 = f32x4{0.f, 0.f, 0.f, 0.f}, l1 = l0;
                if constexpr (!PATCHS) { l0 = lo8_unpack4<T>(a.l[0]); l1 = lo8_unpack4<T>(a.l[1]); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // RESID_SPLIT: x = hi + lo (exact in fp32 up to its 24 bits), then ONE rounding of the sum with the update
                    const float add = PATCHS ? (j < 4 ? a.p0[j] : a.p1[j - 4]) : ((float)a.h[j] + (j < 4 ? l0[j] : l1[j - 4]));
                    v[j] = (j < 4 ? v0[j] : v1[j - 4]) + add;
                    hn[j] = (elem)v[j];
                    ln[j] = v[j] - (float)hn[j];
                    s1 += v[j];
                    s2 = fmaf(v[j], v[j], s2);
                }
                if (ok && !(VH_EPI_ABL & 4)) {
                    *(vec8*)(hi + off) = hn;
                    *(u32x2*)(lo + off) = u32x2{lo8_pack4<T>(ln[0], ln[1], ln[2], ln[3]), lo8_pack4<T>(ln[4], ln[5], ln[6], ln[7])};
                }
                // the 8 lanes of a row: quad butterflies, then the mirrored half-row (lane i <-> 7 - i) joins the two quads
                s1 += dpp_mov<0xB1>(s1); s2 += dpp_mov<0xB1>(s2);     // quad_perm [1,0,3,2]
                s1 += dpp_mov<0x4E>(s1); s2 += dpp_mov<0x4E>(s2);     // quad_perm [2,3,0,1]
                s1 += dpp_mov<0x141>(s1); s2 += dpp_mov<0x141>(s2);   // row_half_mirror
                if (ok && pc == 0 && !(VH_EPI_ABL & 1)) *(float2*)(e.partials + 2 * ((int64_t)(n_w >> 6) * prow + orow)) = make_float2(s1, s2);
            }
        }
    } else {
        // fp32 out: rows of 64 floats = 256 B = 16 chunks, chunk ^ (r & 15); SMI/2 row blocks per pass.
        // The residual values of pass h+1 are loaded BEFORE pass h is staged, added and stored: written as
        // "load, add, store" per line the loads could not move above the previous line's store (same array), so a
        // tile's epilogue was a chain of 32 dependent HBM round trips -- latency-bound at a third of the HBM rate.
        constexpr int FMI = SMI / 2, NP = MI / FMI, NL = FMI * 4;
        constexpr bool RESID = EPI == VH_EPI_BIAS_RESID || EPI == VH_EPI_RESID_LN;
        const int rr = lane >> 4, pc = lane & 15;
        f32x4 xv[2][RESID ? NL : 1];
        auto line = [&](int h, int i, int& m, int& n) {
            const int r = i * 4 + rr;
            n = n_w + ((pc ^ (r & 15)) << 2);
            m = m_w + h * FMI * 16 + r;
            return (f32x4*)((float*)e.out + (int64_t)m * N + n);
        };
        auto load_pass = [&](int h, f32x4 (&x)[RESID ? NL : 1]) {
            if constexpr (RESID) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    int m, n;
                    const f32x4* p = line(h, i, m, n);
                    x[i] = (m < M && !(VH_EPI_ABL & 8)) ? *p : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        };
        load_pass(0, xv[0]);
#pragma unroll
        for (int h = 0; h < NP; ++h) {
            if (h + 1 < NP) load_pass(h + 1, xv[(h + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < FMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    *(f32x4*)(sw + r * 256 + (((ni * 4 + fq) ^ (r & 15)) << 4)) = acc[h * FMI + mi][ni] + bv[ni];
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int r = i * 4 + rr;
                f32x4 v = *(const f32x4*)(sw + r * 256 + (pc << 4));
                int m, n;
                f32x4* p = line(h, i, m, n);
                const bool ok = m < M;
                if constexpr (RESID) v = v + xv[h & 1][i];
                if (ok && !(VH_EPI_ABL & 4)) *p = v;
                if constexpr (EPI == VH_EPI_RESID_LN) {
                    // producer side of the folded LayerNorm: 16-bit copy + (sum, sumsq) of these 64 columns.
                    // The 16 lanes of a DPP row hold one matrix row.
                    if (ok && !(VH_EPI_ABL & 2)) *(typename T::vec4*)((elem*)e.out16 + (int64_t)m * N + n) = pack4<T>(v[0], v[1], v[2], v[3]);
                    const float s1 = row16_sum((v[0] + v[1]) + (v[2] + v[3]));
                    const float s2 = row16_sum(fmaf(v[0], v[0], v[1] * v[1]) + fmaf(v[2], v[2], v[3] * v[3]));
                    if (ok && pc == 0 && !(VH_EPI_ABL & 1)) *(float2*)(e.partials + 2 * ((int64_t)(n_w >> 6) * M + m)) = make_float2(s1, s2);
                }
            }
        }
    }
}

// Tiles whose 256/128 columns are all inside N take the staged path (rows are guarded inside); ragged-N tiles
// and the patch-row remap store directly with per-element predicates.  `smem`: LDS holding NW * SMI * 2 KiB
// from that address on.  With BARRIER it contains a workgroup barrier (needed when other waves may still be
// reading the operand stages): call it from uniform control flow only.
template <typename T, int EPI, int MI, int NI, int SMI = MI, bool BARRIER = true>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[MI][NI], const EpiArgs& e, int m_w, int n_w, int lane,
                                              bool n_full, bool m_full, char* smem, int wave) {
    if constexpr (EPI != VH_EPI_PATCH) {
        if (EPI == VH_EPI_RESID_LN || EPI == VH_EPI_RESID_SPLIT || EPI == VH_EPI_PATCH_SPLIT || n_full) {
            if constexpr (BARRIER) __syncthreads();
            if (m_full && epi_is_16bit(EPI)) gemm_epilogue_staged<T, EPI, MI, NI, SMI, true>(acc, e, m_w, n_w, lane, smem + wave * (SMI * 16 * 128));
            else gemm_epilogue_staged<T, EPI, MI, NI, SMI, false>(acc, e, m_w, n_w, lane, smem + wave * (SMI * 16 * 128));
            return;
        }
    }
    if constexpr (EPI != VH_EPI_RESID_LN && EPI != VH_EPI_RESID_SPLIT && EPI != VH_EPI_PATCH_SPLIT) {
        const int m0 = m_w + (lane & 15), n0 = n_w + (lane >> 4) * 4;
        if (n_full && m_full) gemm_epilogue_impl<T, EPI, MI, NI, false>(acc, e, m0, n0);
        else gemm_epilogue_impl<T, EPI, MI, NI, true>(acc, e, m0, n0);
    }
}

// ---- e4m3 output (fp8 path, fc1): v = gelu(acc + bias) -> 1 byte -----------------------------------------
// A lane's accumulator quad is 4 consecutive columns = one dword of e4m3.  Staged like the 16-bit form: the
// wave's 64-column rows are 64 B = 4 chunks of 16 B, chunk c of row r at position c ^ ((r >> 1) & 3); read back
// 16 B per lane, 4 lanes per row, and stored as whole 64-B row segments.  Ragged tiles store dwords directly.
// OTILED (whole tiles): the e4m3 hidden activation in the tiled layout of the e4m3 A operand, [m / 16][N / 16 chunks][16 rows][16 B]
// (a chunk = 16 k-bytes: the same 2 KiB per row block and K-tile as the 16-bit layout).  A lane holds one dword (4 columns) of each
// of the wave's four 16-column blocks; a 4 x 4 transpose over the four 16-lane rows -- two v_permlane32_swap, two
// v_permlane16_swap -- leaves the lanes of row q with the 16 consecutive bytes of block q: one 16-byte store per lane, 1 KiB of
// contiguous memory per instruction, no LDS staging.
template <int EPI, int MI, int NI, int SMI, bool MFULL = false, bool OTILED = false>
__device__ __forceinline__ void gemm_epilogue8(const f32x4 (&acc)[MI][NI], const EpiArgs& e, int m_w, int n_w, int lane,
                                               bool n_full, char* smem, int wave, bool cpre = false) {
    static_assert(NI == 4 && MI % SMI == 0, "64-column wave tile");
    static_assert(EPI == VH_EPI_BIAS || EPI == VH_EPI_BIAS_GELU || EPI == VH_EPI_LNFOLD_GELU, "8-bit output: bias, bias+GELU or LN-fold+GELU");
    const int M = e.M, N = e.N;
    const int frow = lane & 15, fq = lane >> 4;
    uint8_t* const out = (uint8_t*)e.out;
    // LNFOLD_GELU (fp8 path with the folded LayerNorm): v = gelu(rstd_m * (acc - mean_m * c_n) + d_n), as epi_value16
    auto value = [&](const f32x4& a, const f32x4& b, const f32x4& cq = f32x4{0.f, 0.f, 0.f, 0.f}, float mean_rstd = 0.f, float rstd = 0.f) {
        f32x4 v = epi_value16<EPI, E4M3>(a, b, cq, mean_rstd, rstd);
        return pack4_e4m3(v[0], v[1], v[2], v[3]);
    };
    if (n_full || epi_is_lnfold(EPI)) {   // LN fold: staged form only, N % tile == 0 guaranteed
        f32x4 bv[NI], cv[NI];
        float2 lnst[epi_is_lnfold(EPI) ? MI : 1];
        bool have = false;
        if constexpr (epi_is_lnfold(EPI) && MI == 8) { if (cpre) { epi_consts_from_lds(smem + wave * (SMI * 16 * 128), frow, fq, bv, cv, lnst); have = true; } }   // (wave-uniform)
        if (!have) {
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                bv[ni] = *(const f32x4*)(e.bias + n_w + ni * 16 + fq * 4);
                if constexpr (epi_is_lnfold(EPI)) cv[ni] = *(const f32x4*)(e.aux + n_w + ni * 16 + fq * 4);
                else cv[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if constexpr (epi_is_lnfold(EPI)) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    int m = m_w + mi * 16 + frow;
                    m = m < M ? m : M - 1;
                    lnst[mi] = *(const float2*)(e.stats + 2 * (int64_t)m);
                }
            }
        }
        if constexpr (OTILED && MFULL) {
            uint8_t* const obase = out + ((int64_t)(m_w >> 4) * (N >> 4) + (n_w >> 4) + fq) * 256 + frow * 16;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                uint32_t x[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (epi_is_lnfold(EPI)) x[ni] = value(acc[mi][ni], bv[ni], cv[ni], lnst[mi].x * lnst[mi].y, lnst[mi].y);
                    else x[ni] = value(acc[mi][ni], bv[ni]);
                }
                // halves first (rows 2, 3 of x0 / x1 <-> rows 0, 1 of x2 / x3), then odd / even rows inside the halves
                const auto a = __builtin_amdgcn_permlane32_swap(x[0], x[2], false, false);
                const auto b = __builtin_amdgcn_permlane32_swap(x[1], x[3], false, false);
                const auto lo = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
                epi_store(u32x4{lo[0], lo[1], hi[0], hi[1]}, (u32x4*)(obase + (int64_t)mi * (N >> 4) * 256));
            }
            return;
        }
        char* sw = smem + wave * (SMI * 16 * 128);
        const int rr = lane >> 2, pc = lane & 3;
#pragma unroll
        for (int h = 0; h < MI / SMI; ++h) {
#pragma unroll
            for (int mi = 0; mi < SMI; ++mi) {
                const int r = mi * 16 + frow;
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (epi_is_lnfold(EPI)) {
                        const float2 st = lnst[h * SMI + mi];
                        *(uint32_t*)(sw + r * 64 + ((ni ^ ((r >> 1) & 3)) << 4) + fq * 4) = value(acc[h * SMI + mi][ni], bv[ni], cv[ni], st.x * st.y, st.y);
                    } else {
                        *(uint32_t*)(sw + r * 64 + ((ni ^ ((r >> 1) & 3)) << 4) + fq * 4) = value(acc[h * SMI + mi][ni], bv[ni]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SMI; ++i) {
                const int r = i * 16 + rr;
                const u32x4 v = *(const u32x4*)(sw + r * 64 + (pc << 4));
                const int n = n_w + ((pc ^ ((r >> 1) & 3)) << 4);
                const int m = m_w + h * SMI * 16 + r;
                if (MFULL || m < M) epi_store(v, (u32x4*)(out + (int64_t)m * N + n));
            }
        }
        return;
    }
    if constexpr (!epi_is_lnfold(EPI)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int n = n_w + ni * 16 + fq * 4;
            if (n >= N) continue;
            const f32x4 bv = *(const f32x4*)(e.bias + n);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int m = m_w + mi * 16 + frow;
                if (m < M) *(uint32_t*)(out + (int64_t)m * N + n) = value(acc[mi][ni], bv);
            }
        }
    }
}

}  // namespace vh
