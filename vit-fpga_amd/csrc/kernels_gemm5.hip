// kernels_gemm5.hip — the ping-pong GEMM of the ViT hot path (gfx950), variants 5 (one tile per workgroup)
// and 6 (persistent: one workgroup per CU walks the tiles and prefetches across tile boundaries).
//
// Same contract, operands, swizzled BK=64 LDS image and fused epilogues as kernels_gemm.hip; 256x256 tile,
// 8 waves, two 64 KiB stages.  What changes is WHO uses the matrix pipe WHEN.
//
// Ablation of the plain two-stage loop (tools/diag_gemm.hip, QKV shape) showed its three costs ADD instead
// of overlapping: MFMA only 180 us, DMA only 176 us, MFMA + fragment reads 239 us, all three 310 us —
// because all 8 waves run [barrier, issue DMA, read fragments, MFMA] in lockstep, so each SIMD's matrix pipe
// idles whenever its two waves are both loading.  Here the two waves of a SIMD work in opposite phases:
//
//   wave group G0 = waves 0-3 (tile rows 0..127), G1 = waves 4-7 (rows 128..255); wave i and i+4 share a SIMD.
//   per K-tile a wave runs four phases, each closed by s_barrier:   L0  C0  L1  C1
//      L0: ds_read W fragments (both k-steps) + X fragments of k-step 0, issue DMA of its A half of tile k+1
//      C0: 32 MFMAs (k-step 0)
//      L1: ds_read X fragments of k-step 1, issue DMA of its W half of tile k+2
//      C1: 32 MFMAs (k-step 1)
//   G1 runs one phase behind G0 (one extra barrier up front), so in every phase one wave of each SIMD
//   multiplies while the other reads LDS and feeds the DMA queue.
//
// Hazard bookkeeping (p = global phase; G0: L0 4k, C0 4k+1, L1 4k+2, C1 4k+3; G1: +1):
//   * every L phase ends with lgkmcnt(0) before its barrier  => reads issued in phase p are complete for
//     everybody after the barrier that closes p.
//   * RAW (DMA -> ds_read): the issuing wave's counted vmcnt, then a barrier the readers pass.
//       end of L1(k): vmcnt(8|4|0) -> this wave's W half of tile k+1 landed   (read from p = 4k+4)
//       end of C1(k): vmcnt(4|0)   -> this wave's A half of tile k+1 landed   (read by its own group)
//   * WAR (ds_read -> DMA): W of a stage is read only in L0 (both groups: phases 4k, 4k+1), refilled from
//     L1 (4k+2 / 4k+3); a group's A half is read in its L0/L1 and refilled by the same group in its next L0.
//   DMA of a tile is in flight for >= 4 phases (~2000 cycles) before it is needed; the waits are counted, the
//   barriers raw, so later tiles' DMA stays in flight across them.
//
// Persistent mode: after the last phase of an output tile both stages are idle.  The workgroup issues the DMA
// of the NEXT tile's K-tile 0 into stage 0, runs the epilogue through 8 KiB-per-wave slices of stage 1, then
// issues K-tile 1 into stage 1: the next tile's first operands are already in LDS when its loop starts, and no
// workgroup launch/drain sits between tiles.
//
// fp8 form (F8 = true, VH_DTYPE_FP8): A and W are OCP e4m3 bytes, a K-tile is still 128 B per row (128 elements),
// so the DMA, the LDS image, the swizzle and the hazard table are byte for byte the ones above.  The matrix
// instruction is v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (twice the cycles of the bf16
// 16x16x32 form at four times the K: 2x the bf16 rate); lane l carries row l&15 and the 32 k-bytes of group
// l>>4 (tools/probe_fp8.hip), i.e. chunks 2g and 2g+1 of the 128-B row.  One K-tile is ONE k-step, so the two
// compute phases split the wave tile by rows instead of by k:
//      L0: W fragments + X fragments of row blocks 0-3     C0: 16 MFMAs (row blocks 0-3)
//      L1: X fragments of row blocks 4-7                   C1: 16 MFMAs (row blocks 4-7)
// (same 512 matrix cycles per phase, same 64 fragment registers).  W carries one fp32 scale per output channel
// (`aux`), applied to the accumulators before the epilogue; activations are unscaled (vh_common.h, E4M3).
#include <cstdlib>
#include <type_traits>

#include "gemm_epilogue.h"
#include "vh_kernels.h"

#ifndef VH_PP_SMI
#define VH_PP_SMI 2   // 16-row blocks staged per epilogue pass (slice = VH_PP_SMI * 2 KiB per wave); 2 vs 4: fc1 -1.2 %, stores spread finer under the VALU work
#endif

namespace vh {

#ifndef VH_MAIN_ABL
#define VH_MAIN_ABL 0   // timing-only ablation builds (tools/build_abl.sh), 16-bit form: 1 no MFMAs, 2 no fragment reads, 4 fragment
                        // reads in the workgroup's first K-tile only (real operands, then none), 8 no DMA after the prologue, 16 no barriers, 32 no s_setprio(1),
                        // 64 (round 4) the K-tile of a 256x128 workgroup tile with 64x64 wave tiles -- the geometry that has room for a
                        // SECOND accumulator set -- emulated in this loop: half the MFMAs (16 per compute phase), half the activation
                        // fragment reads, and 6 instead of 8 DMA pieces per wave and K-tile (the W operand is 128 rows: 2 pieces per wave)
#endif
#ifndef VH_PP_CPRE
#define VH_PP_CPRE 1   // A/B: -DVH_PP_CPRE=0 loads the LN-fold epilogues' constants from global memory inside the epilogue
#endif
constexpr int kAblHalf = (VH_MAIN_ABL & 64) ? 1 : 0;
constexpr int kWPieces = kAblHalf ? 2 : 4;   // DMA pieces of the W operand per wave and K-tile (the counted waits follow)
// ---- DIAGNOSTIC BUILD ONLY (-DVH_DIAG_STAMPS -> libvithip_diag.so, tools/gemm_anatomy.py) ---------------------------
// Wave 0 of every workgroup stamps s_memrealtime (100 MHz, chip-wide) at: kernel entry, first K-tile visible, end of
// the main loop, epilogue issued, stores drained; s_memtime (shader clock) around the main loop (in-kernel clock =
// d(s_memtime) / d(s_memrealtime) x 100 MHz, MI355X_MICROARCH.md "DVFS give-back" item 6); and the hardware id of its
// CU.  The stamps go to a ring buffer of their own that nothing else reads; no output depends on them.  In the
// product build (no VH_DIAG_STAMPS) none of this exists: no parameter, no instruction.
#ifdef VH_DIAG_STAMPS
#ifndef VH_DIAG_KT
#define VH_DIAG_KT -1      // which K-tile of the stamped tile gets the phase stamps (-1: the middle one; -DVH_DIAG_KT=0: the first, behind the
                           // previous tile's epilogue and its stores)
#endif
#ifndef VH_DIAG_REC_IT
#define VH_DIAG_REC_IT 1   // which tile of a persistent workgroup is stamped (-DVH_DIAG_REC_IT=8: deep in the steady state)
#endif
#define VH_STAMP_PARAM , unsigned long long* __restrict__ stamps
__device__ __forceinline__ unsigned long long diag_rt() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long diag_ct() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long diag_hwid() {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return ((unsigned long long)xcc << 32) | hw;
}
#define VH_STAMP(i, expr) do { __builtin_amdgcn_sched_barrier(0); st_[i] = (expr); __builtin_amdgcn_sched_barrier(0); } while (0)
// Phase stamps of ONE K-tile (the middle one) of the stamped tile, written straight to the stamp record's words 8-15 by
// lane 0 (nothing is kept in registers across the K loop): shader clock at  0 top of L0 | 1 L0's reads back, DMA issued |
// 2 past L0's barrier | 3 C0's MFMAs issued | 4 past C0's barrier | 5 L1's reads back, DMA issued, counted wait passed |
// 6 past L1's barrier | 7 C1's MFMAs issued and its counted wait passed.  (The stores are vector-memory operations of
// their own: they make the K-tile's counted waits stricter, never laxer.)
#define VH_PSTAMP(i) do { if (pst_) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = diag_ct(); \
    if (lane == 0) pst_[i] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define VH_PSTAMP(i) do { } while (0)
#define VH_STAMP_PARAM
#define VH_STAMP(i, expr) do { } while (0)
#endif

// which persistent instantiations prefetch their epilogue's constants through LDS-DMA (issue_consts): the LN-fold epilogues, except
// fc1 with e4m3 operands (that instantiation sits at 256 registers and the extra live values spill into the K loop).  (The same for
// the split-residual epilogue's bias and first-pass planes was built and measured: out-proj -0.8 %, fc2 +2.3 % -- the first wait for
// an ordinary load then comes BEHIND the first pass's stores and, being vmcnt(0), waits for their acknowledgements too.  Removed.)
template <int EPI, bool F8>
__host__ __device__ constexpr bool pp_uses_cpre() {
    return VH_PP_CPRE && epi_is_lnfold(EPI) && !(F8 && EPI == VH_EPI_LNFOLD_GELU);
}

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
#if VH_MAIN_ABL & 16   // (with 12 only: no memory operation is left to order) the two waves of a SIMD issue as they come
    return;
#endif
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int xcd_remap(int bid, int n) {  // bijective on [0, n)
    const int xcd = bid & 7, qd = n >> 3, rm = n & 7;
    return (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
}

// OT: the 16-bit GELU result goes to the tiled layout of the MLP hidden activation (gemm_epilogue.h); AT: A and W are read from it
// (16-row blocks: the per-lane DMA offsets and the K-tile stride change, nothing inside the K loop does).  Persistent form only.
template <typename T, int EPI, bool PERSIST, bool F8 = false, int AST = 2, bool OT = false, bool AT = false>
__global__ void __launch_bounds__(512, 2)
gemm_nt_pp_kernel(const void* __restrict__ Av, const void* __restrict__ Wv,
                  const float* __restrict__ bias, void* __restrict__ outp, int M, int N, int K,
                  const float* __restrict__ aux, int aux_i, int tiles_m, int tiles_n, const float* __restrict__ stats,
                  void* __restrict__ out16, float* __restrict__ partials, int tile0, int sn,
                  const float* __restrict__ wscale, int64_t prow VH_STAMP_PARAM) {
    using vec8 = typename T::vec8;
#ifdef VH_DIAG_STAMPS
    // the iteration whose stamps are kept: the workgroup's only tile, or the SECOND tile of a persistent workgroup
    // (steady state: its K-tile 0 was prefetched under the previous tile's epilogue)
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int it_ = 0;
    constexpr int rec_it_ = PERSIST ? VH_DIAG_REC_IT : 0;
#define first_tile_ (it_ == rec_it_)
#endif
    constexpr int BM = 256, BN = 256;
    constexpr int KT_BYTES = 128;                  // one K-tile row: 64 16-bit or 128 8-bit elements
    static_assert(!(OT || AT) || PERSIST, "tiled layouts: persistent form only");
    constexpr int KT_STEP = AT ? 2048 : KT_BYTES;  // source bytes from one K-tile to the next (tiled: 8 chunks x 256 B)
    constexpr int STAGE_BYTES = 65536, W_OFF = 32768;
    // AST = number of LDS stages of the A (activation) operand.  2: stage s = [A 32 KiB | W 32 KiB] at s * 64 KiB.
    // 3 (variant 7): W stages at 0 / 32 KiB, A stages at 64 + 32 s KiB (160 KiB, the whole LDS): A of tile k+2 is
    // issued in L0(k), a full K-tile earlier than with two stages, and the counted wait that closed C1 disappears.
    static_assert(AST == 2 || (AST == 3 && !PERSIST), "three A stages: one tile per workgroup only");
    const char* const A = (const char*)Av;
    const char* const W = (const char*)Wv;
    const int64_t row_bytes = (int64_t)K * (F8 ? 1 : 2);
    constexpr int MI = 8, NI = 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int ntiles = tiles_m * tiles_n;
    // one tile per workgroup: the grid covers tiles [tile0, tile0 + gridDim.x) of the n-fastest tile order (a launch
    // may cover a sub-range: GemmArgs::tile_begin / tile_count); persistent: tile0 == 0, stride = grid
    const int stride = (int)gridDim.x;
    int t = tile0 + xcd_remap(blockIdx.x, stride);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, wn = wave & 3;

    // ---- DMA: this wave moves 4 x 1 KiB of its group's A half and 4 x 1 KiB of its group's W half -------
    // A source address = wave-uniform panel base (first row of the tile, 64-bit, scalar registers) + a per-lane
    // 32-bit byte offset (row inside the tile x row_bytes + swizzled 16-B chunk): 8 address registers per wave
    // instead of 16 (4 in the persistent form) and no 64-bit vector adds per DMA.
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    const int rb = (int)row_bytes;
    struct TileSrc { const char *a, *w; uint32_t oa[4], ow[4]; };
    int tile_m = 0, tile_n = 0;
    auto tile_coords = [&](int tt, int& tm, int& tn) {
        // Tile order.  sn == 0: n fastest over all tiles_n column tiles.  sn > 0: super-columns of sn column tiles, m
        // fastest-but-one inside each: the workgroups of an XCD then share sn W panels (which stay in its 4 MiB L2
        // from round to round) instead of all tiles_n (fc1: 12 panels = 4.7 MB, re-fetched from beyond L2 every round).
        if (sn > 0) {
            const int blk = tiles_m * sn, sc = tt / blk, r = tt - sc * blk;
            const int width = tiles_n - sc * sn < sn ? tiles_n - sc * sn : sn;
            tm = r / width;
            tn = sc * sn + (r - tm * width);
        } else {
            tm = tt / tiles_n;
            tn = tt - tm * tiles_n;
        }
    };
    auto base_a = [&](int tm) { return A + (int64_t)tm * BM * row_bytes; };
    auto base_w = [&](int tn) { return W + (int64_t)tn * BN * row_bytes; };
    TileSrc cur;
    auto setup_tile = [&](int tt) {
        tile_coords(tt, tile_m, tile_n);
        cur.a = base_a(tile_m);
        cur.w = base_w(tile_n);
        // last valid row inside the tile (rows beyond it replicate it).  Persistent form: full tiles only, so the per-lane
        // offsets are the same for A and W and for every tile -- only the scalar bases change from tile to tile.
        const int ma = PERSIST ? 255 : M - 1 - tile_m * BM, mw = PERSIST ? 255 : N - 1 - tile_n * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = grp * 128 + (i * 4 + wn) * 8 + lr;
            // AT: the source is [16-row block][K / 8 chunks][16 rows][16 B], and the LDS image of a K-tile keeps that order (8 chunks x
            // 256 B per row block): piece p = i * 4 + wn of the group's half is row block p >> 1, chunks 4 (p & 1) .. + 3 -- ONE KiB OF
            // CONTIGUOUS SOURCE per DMA instruction, lane-linear on both sides (a first form that kept the row-major LDS image read
            // 16-byte pieces 256 B apart from adjacent lanes: fc2 +35 %, the coalescer works on adjacent lanes)
            if constexpr (AT) cur.oa[i] = (uint32_t)((grp * 8 + ((i * 4 + wn) >> 1)) * (rb * 16) + ((i * 4 + wn) & 1) * 1024 + lane * 16);
            else
            cur.oa[i] = (uint32_t)((r < ma ? r : ma) * rb + lc * 16);
            if constexpr (!PERSIST) cur.ow[i] = (uint32_t)((r < mw ? r : mw) * rb + lc * 16);
        }
    };
    const int dma_off = grp * 16384 + wn * 1024;  // + i * 4096
    // LDS stage of K-tile kt of the current tile.  One tile per workgroup: kt & 1.  Persistent form: the K-tiles of
    // consecutive tiles form ONE stream through the two stages, so a tile's K-tile 0 sits in stage `par` (the parity of
    // the K-tiles that went before).
    int par = 0;
    auto a_off = [&](int kt, int kt3) { return AST == 2 ? ((par + kt) & 1) * STAGE_BYTES : 2 * W_OFF + kt3 * 32768; };
    auto w_off = [&](int kt) { return AST == 2 ? ((par + kt) & 1) * STAGE_BYTES + W_OFF : (kt & 1) * W_OFF; };
    bool dma_on = true;   // VH_MAIN_ABL & 8 only
    auto dma4 = [&](const char* base, const uint32_t (&off)[4], char* dst, int pieces = 4) {
        if constexpr (VH_MAIN_ABL & 8) { if (!dma_on) return; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < pieces)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(base + off[i]),
                                             (void __attribute__((address_space(3)))*)(dst + i * 4096), 16, 0, 0);
    };
    auto issue_a = [&](int kt, int kt3 = 0) { dma4(cur.a + (int64_t)kt * KT_STEP, cur.oa, smem + a_off(kt, kt3) + dma_off); };
    auto issue_w = [&](int kt) { dma4(cur.w + (int64_t)kt * KT_STEP, PERSIST ? cur.oa : cur.ow, smem + w_off(kt) + dma_off, kWPieces); };

    // ---- fragment addresses ----------------------------------------------------------------------------------
    const int frow = lane & 15, fq = lane >> 4;
    // 16-bit: k-step 0 / 1 = chunk fq / 4+fq; fp8: the lane's 32 k-bytes = chunks 2fq and 2fq+1
    // (AT: row block = 8 chunks x [16 rows x 16 B]; a fragment read takes 4 chunks x 16 rows = 1 KiB of contiguous LDS: no swizzle needed)
    const int off0 = AT ? (F8 ? 2 * fq : fq) * 256 + frow * 16 : frow * 128 + (((F8 ? 2 * fq : fq) ^ (frow & 7)) << 4);
    const int off1 = AT ? (F8 ? 2 * fq + 1 : 4 + fq) * 256 + frow * 16 : frow * 128 + (((F8 ? 2 * fq + 1 : 4 | fq) ^ (frow & 7)) << 4);
    const int xbase = grp * 16384;          // rows 128*grp ..  (inside an A stage)
    const int wbase = wn * 8192;            // rows 64*wn ..    (inside a W stage)

    const int nk = (int)(row_bytes / KT_BYTES);
    // Constants of the LN-fold epilogues prefetched through LDS (persistent form;
    // K-tile 1 must have a W(k+2) slot: the launcher sends shapes with fewer than four K-tiles to the one-tile form).  A compile-time
    // property of the instantiation, so that the epilogue has no second path whose loads the compiler would have to fence.
    constexpr bool cpre = pp_uses_cpre<EPI, F8>() && PERSIST && AST == 2;

    // ---- prologue of the first tile: K-tiles 0 and 1 ------------------------------------------------------------
    setup_tile(t);
    issue_w(0);
    issue_a(0);
    if (nk > 1) {
        issue_w(1);
        issue_a(1, 1);
    }
    if (nk > 1) pp_wait_vmcnt<4 + kWPieces>();  // K-tile 0 landed, K-tile 1 may fly
    else pp_wait_vmcnt<0>();
    pp_barrier();                    // K-tile 0 visible
    if constexpr (VH_MAIN_ABL & 8) dma_on = false;

    // Persistent form (nk >= 2, full tiles only: the launcher sees to both): the DMA stream does not stop at a tile
    // boundary.  The slots of the schedule that would fetch K-tiles nk and nk + 1 of the current tile fetch K-tiles 0 and
    // 1 of the NEXT tile instead -- A(next, 0) in L0(nk-1), W(next, 0) in L1(nk-2), W(next, 1) in L1(nk-1), A(next, 1) in
    // L0 of the next tile's K-tile 0 -- with the usual counted waits, so a tile's first operands are in LDS when its loop
    // starts: no per-tile prologue, no launch gap, no burst of DMAs.  The epilogue stages through an LDS region of its own
    // behind the two stages (160 KiB in this form), and its loads and stores simply join the in-order queue: its leading
    // loads wait for W(next, 1), issued one phase earlier; its stores are retired by the next tile's first counted wait.
    bool first = true;   // first tile of this workgroup: A(1) came from the prologue above
    while (true) {
#ifdef VH_DIAG_STAMPS
        if (first_tile_) VH_STAMP(0, diag_rt());
#endif
        f32x4 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int t_next = t + stride;
        const bool has_next = PERSIST && t_next < ntiles;
        int tm_n = 0, tn_n = 0;
        if (has_next) tile_coords(t_next, tm_n, tn_n);

        if (grp == 1) pp_barrier();   // G1 runs one phase behind
#ifdef VH_DIAG_STAMPS
        if (first_tile_) { VH_STAMP(1, diag_rt()); VH_STAMP(5, diag_ct()); }
#endif

        // the DMA slots of one K-tile (AST == 2 unless noted)
        auto l0_issue = [&](int kt, int k3) {
            if (AST == 3) { if (kt + 2 < nk) issue_a(kt + 2, k3 == 0 ? 2 : k3 - 1); }
            else if (kt + 1 < nk) { if (kt >= 1 || (PERSIST && !first)) issue_a(kt + 1); }
            else if (has_next) {   // kt == nk - 1: no A issue of this tile is left; the A base moves on to the next tile
                cur.a = base_a(tm_n);
                dma4(cur.a, cur.oa, smem + a_off(nk, 0) + dma_off);
            }
        };
        // LN-fold forms (16-bit persistent): the epilogue's constants of THIS tile -- 128 (mean, rstd) pairs, 64 d_n, 64 c_n per wave --
        // go into the wave's own staging slice by three more DMA instructions in L1 of K-tile 1 (the registers of the first
        // weight fragments are free there), behind W(k+2): the two waits of that K-tile may leave three more operations in flight,
        // every later wait finds them landed (gemm_epilogue.h `cpre`)
        auto issue_consts = [&]() {
            char* const slice = smem + 2 * STAGE_BYTES + wave * 4096;
            const int m0 = tile_m * BM + grp * 128, n0 = tile_n * BN + wn * 64;
            if constexpr (F8) {
                // e4m3 operands: the K loop has no register for three 64-bit address pairs (they spilled 12): scalar bases + ONE
                // per-lane offset register through the asm helpers; the weight scales of the 64 columns travel too (kNC = 4)
                const uint32_t la = (uint32_t)(uintptr_t)slice;
                uint32_t l4 = (uint32_t)lane * 4u;
                asm_lds_dma4(bias + n0, l4, la + 1024);
                asm_lds_dma4(aux + n0, l4, la + 1280);
                asm_lds_dma4(wscale + n0, l4, la + 1536);
                l4 *= 4u;
                asm_lds_dma16((const char*)stats + (int64_t)m0 * 8, l4, la);
                return;
            }
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)((const char*)stats + ((int64_t)m0 * 8 + lane * 16)),
                                             (void __attribute__((address_space(3)))*)slice, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(bias + n0 + lane),
                                             (void __attribute__((address_space(3)))*)(slice + 1024), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(aux + n0 + lane),
                                             (void __attribute__((address_space(3)))*)(slice + 1280), 4, 0, 0);
        };
        constexpr int kNC = F8 ? 4 : 3;   // DMA instructions of issue_consts
        auto l1_issue_wait = [&](int kt) {
            if (kt + 2 < nk) {
                issue_w(kt + 2);
                if (cpre && kt == 1) { issue_consts(); pp_wait_vmcnt<4 + kWPieces + kNC>(); }
                else
                pp_wait_vmcnt<4 + kWPieces>();   // all but A(k+1) | W(k+2) (AST 3: A(k+2), W(k+2)) => W(k+1) landed
            } else if (has_next) {
                if (kt + 2 == nk) {   // no W issue of this tile is left; the W base moves on to the next tile
                    cur.w = base_w(tn_n);
                    dma4(cur.w, cur.oa, smem + w_off(nk) + dma_off, kWPieces);
                } else {
                    dma4(cur.w + KT_STEP, cur.oa, smem + w_off(nk + 1) + dma_off, kWPieces);
                }
                pp_wait_vmcnt<4 + kWPieces>();
            } else if (AST == 2 && kt + 1 < nk) {
                pp_wait_vmcnt<4>();
            } else {
                pp_wait_vmcnt<0>();
            }
        };
        auto c1_wait = [&](int kt) {
            if (AST == 2) {
                if (cpre && kt == 1) pp_wait_vmcnt<kWPieces + kNC>();     // (+ the constant loads behind W(k+2))
                else
                if (kt + 2 < nk || has_next) pp_wait_vmcnt<kWPieces>();   // all but W(k+2) => A(k+1) landed
                else pp_wait_vmcnt<0>();
            }
        };

        if constexpr (F8) {
            i32x8 wf[NI], xf[MI / 2];
            const int unit = 0x7F7F7F7F;  // E8M0 1.0 for every 32-element block
            auto ld8 = [&](const char* p) {
                const u32x4 lo = *(const u32x4*)(p + off0), hi = *(const u32x4*)(p + off1);
                return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
            for (int kt = 0, k3 = 0; kt < nk; ++kt, k3 = k3 == 2 ? 0 : k3 + 1) {
                const char* sa = smem + a_off(kt, k3) + xbase;
                const char* sw = smem + w_off(kt) + wbase;
                // ---- L0 ----
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) wf[ni] = ld8(sw + ni * 2048);
#pragma unroll
                for (int mi = 0; mi < MI / 2; ++mi) xf[mi] = ld8(sa + mi * 2048);
                l0_issue(kt, k3);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                pp_barrier();
                // ---- C0 ----
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < MI / 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], xf[mi], acc[mi][ni], 0, 0, 0, unit, 0, unit);
                __builtin_amdgcn_s_setprio(0);
                pp_barrier();
                // ---- L1 ----
#pragma unroll
                for (int mi = 0; mi < MI / 2; ++mi) xf[mi] = ld8(sa + (MI / 2 + mi) * 2048);
                l1_issue_wait(kt);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                pp_barrier();
                // ---- C1 ----
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int mi = 0; mi < MI / 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[MI / 2 + mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ni], xf[mi], acc[MI / 2 + mi][ni], 0, 0, 0, unit, 0, unit);
                __builtin_amdgcn_s_setprio(0);
                c1_wait(kt);
                pp_barrier();
            }
        } else {
        vec8 wf0[NI], wf1[NI], xf[MI];
        constexpr int MI_A = kAblHalf ? MI / 2 : MI;   // row blocks of the wave tile that are read and multiplied (ablation 64: half)
        for (int kt = 0, k3 = 0; kt < nk; ++kt, k3 = k3 == 2 ? 0 : k3 + 1) {
            const char* sa = smem + a_off(kt, k3) + xbase;
            const char* sw = smem + w_off(kt) + wbase;
#ifdef VH_DIAG_STAMPS
            unsigned long long* const pst_ = (first_tile_ && stamps && kt == (VH_DIAG_KT < 0 ? nk / 2 : VH_DIAG_KT)) ? stamps + ((size_t)blockIdx.x * 8 + wave) * 16 + 8 : nullptr;
#endif
            VH_PSTAMP(0);
            // ---- L0 ------------------------------------------------------------------------------------------------
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                if constexpr (VH_MAIN_ABL & 2) { asm volatile("" : "=v"(wf0[ni])); asm volatile("" : "=v"(wf1[ni])); continue; }
                if constexpr (VH_MAIN_ABL & 4) { if (!(first && kt == 0)) { asm volatile("" : "+v"(wf0[ni])); asm volatile("" : "+v"(wf1[ni])); continue; } }
                wf0[ni] = *(const vec8*)(sw + ni * 2048 + off0);
                wf1[ni] = *(const vec8*)(sw + ni * 2048 + off1);
            }
#pragma unroll
            for (int mi = 0; mi < MI_A; ++mi) {
                if constexpr (VH_MAIN_ABL & 2) { asm volatile("" : "=v"(xf[mi])); continue; }
                if constexpr (VH_MAIN_ABL & 4) { if (!(first && kt == 0)) { asm volatile("" : "+v"(xf[mi])); continue; } }
                xf[mi] = *(const vec8*)(sa + mi * 2048 + off0);
            }
            l0_issue(kt, k3);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            VH_PSTAMP(1);
            pp_barrier();
            VH_PSTAMP(2);
            // ---- C0 ------------------------------------------------------------------------------------------------
            if constexpr (!(VH_MAIN_ABL & 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < MI_A; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (VH_MAIN_ABL & 1) asm volatile("" : "+v"(acc[mi][ni]) : "v"(wf0[ni]), "v"(xf[mi]));
                    else acc[mi][ni] = T::mfma16(wf0[ni], xf[mi], acc[mi][ni]);
                }
            __builtin_amdgcn_s_setprio(0);
            VH_PSTAMP(3);
            pp_barrier();
            VH_PSTAMP(4);
            // ---- L1 ------------------------------------------------------------------------------------------------
#pragma unroll
            for (int mi = 0; mi < MI_A; ++mi) {
                if constexpr (VH_MAIN_ABL & 2) { asm volatile("" : "=v"(xf[mi])); continue; }
                if constexpr (VH_MAIN_ABL & 4) { asm volatile("" : "+v"(xf[mi])); continue; }
                xf[mi] = *(const vec8*)(sa + mi * 2048 + off1);
            }
            l1_issue_wait(kt);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            VH_PSTAMP(5);
            pp_barrier();
            VH_PSTAMP(6);
            // ---- C1 ------------------------------------------------------------------------------------------------
            if constexpr (!(VH_MAIN_ABL & 32)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < MI_A; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    if constexpr (VH_MAIN_ABL & 1) asm volatile("" : "+v"(acc[mi][ni]) : "v"(wf1[ni]), "v"(xf[mi]));
                    else acc[mi][ni] = T::mfma16(wf1[ni], xf[mi], acc[mi][ni]);
                }
            __builtin_amdgcn_s_setprio(0);
            c1_wait(kt);
            VH_PSTAMP(7);
            pp_barrier();
        }
        }
        if (grp == 0) pp_barrier();  // G0 waits for G1's last phase: every LDS read of this tile is complete
#ifdef VH_DIAG_STAMPS
        if (first_tile_) { VH_STAMP(6, diag_ct()); VH_STAMP(2, diag_rt()); }
#endif

        // ---- tile boundary --------------------------------------------------------------------------------------
        const int m_w = tile_m * BM + grp * 128, n_w = tile_n * BN + wn * 64;
        const bool m_full = (tile_m + 1) * BM <= M, n_full = (tile_n + 1) * BN <= N;
        // one tile per workgroup: the epilogue stages through the (idle) stage 1; persistent: through its own region
        char* const stage_epi = smem + (PERSIST ? 2 * STAGE_BYTES : STAGE_BYTES);
        const EpiArgs e{bias, outp, M, N, aux, aux_i, stats, out16, partials, prow};
        // Everything the epilogue derives from the lane id is derived HERE, per tile: in the persistent form the compiler
        // would otherwise hoist those lane-constant addresses out of the tile loop and keep them alive through the K loop,
        // which has no register to spare (spills inside the K loop are vector-memory operations: they would break its
        // counted waits).
        int lane_e = lane;
        if constexpr (PERSIST) asm volatile("" : "+v"(lane_e));
        if constexpr (F8) {
            // per-output-channel weight scale (`wscale`): the lane's 4 consecutive columns of column block ni.  The product is
            // rounded on its own (no contraction with the epilogue's "+ bias" into an fma): every form of the kernel then
            // produces the same bits, whatever the compiler's inlining context makes of the two statements.
#pragma clang fp contract(off)
            const int fq_e = lane_e >> 4;
            f32x4 wsv[NI];
            if (cpre) {   // (wave-uniform) prefetched beside the other constants; gemm_epilogue.h epi_consts_from_lds says why this is asm
                const uint32_t la = (uint32_t)(uintptr_t)(smem + 2 * STAGE_BYTES + wave * 4096) + (uint32_t)(fq_e * 16);
                static_assert(NI == 4, "64-column wave tile");
                asm volatile("ds_read_b128 %0, %4 offset:1536\n\tds_read_b128 %1, %4 offset:1600\n\tds_read_b128 %2, %4 offset:1664\n\tds_read_b128 %3, %4 offset:1728\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&v"(wsv[0]), "=&v"(wsv[1]), "=&v"(wsv[2]), "=&v"(wsv[3]) : "v"(la));
            } else {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const int n = n_w + ni * 16 + fq_e * 4;
                    wsv[ni] = n < N ? *(const f32x4*)(wscale + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = acc[mi][ni] * wsv[ni];
            }
        }
        if constexpr (PERSIST) {
            // full tiles only: ONE straight-line staged epilogue.  (With a ragged variant beside it hipcc hoisted the shared
            // leading load above the branch between the two, saw a path on which it is never waited for, and drained the
            // DMA queue -- vmcnt(0) -- before the next tile's first fragment reads overwrote its register.)
            // (fp32 results: 16 rows x 256 B per pass, so that the 8 waves' slices fit the 32 KiB behind the stages)
            constexpr int SLICE = VH_PP_SMI * 16 * 128;
            static_assert(VH_PP_SMI == 2, "staging region of the persistent form: 8 waves x 4 KiB");
            // fp8 operands: GELU results leave as e4m3 (the next GEMM's A operand); RESID_LN writes an e4m3 copy of the rows, RESID_SPLIT
            // keeps the residual itself as an e4m3 plane (that operand) + a bf16 plane
            if constexpr (F8 && epi_has_gelu(EPI))
                gemm_epilogue8<EPI, MI, NI, VH_PP_SMI, true, OT>(acc, e, m_w, n_w, lane_e, true, stage_epi, wave, cpre);
            else if constexpr (F8 && (EPI == VH_EPI_RESID_LN || EPI == VH_EPI_RESID_SPLIT))
                gemm_epilogue_staged<E4M3, EPI, MI, NI, VH_PP_SMI, true>(acc, e, m_w, n_w, lane_e, stage_epi + wave * SLICE);
            else
                gemm_epilogue_staged<T, EPI, MI, NI, VH_PP_SMI, true, OT>(acc, e, m_w, n_w, lane_e, stage_epi + wave * SLICE, epi_is_16bit(EPI) ? grp : -1, cpre);
        } else {
            if constexpr (F8 && epi_has_gelu(EPI))
                gemm_epilogue8<EPI, MI, NI, VH_PP_SMI>(acc, e, m_w, n_w, lane_e, n_full, stage_epi, wave);
            else if constexpr (F8 && (EPI == VH_EPI_RESID_LN || EPI == VH_EPI_RESID_SPLIT))
                gemm_epilogue<E4M3, EPI, MI, NI, 4, false>(acc, e, m_w, n_w, lane_e, n_full, m_full, stage_epi, wave);
            else
                gemm_epilogue<T, EPI, MI, NI, (epi_is_16bit(EPI) ? VH_PP_SMI : 4), false>(acc, e, m_w, n_w, lane_e, n_full, m_full, stage_epi, wave);   // fp32 forms: 32 rows per pass
        }
#ifdef VH_DIAG_STAMPS
        if (first_tile_) VH_STAMP(3, diag_rt());
        if (!has_next && first_tile_) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores drained (one tile per workgroup)
        if (first_tile_) {
            VH_STAMP(4, diag_rt());
            st_[7] = diag_hwid();
            if (stamps && lane == 0) {   // one record per wave
#pragma unroll
                for (int i = 0; i < 8; ++i) stamps[((size_t)blockIdx.x * 8 + wave) * 16 + i] = st_[i];
            }
        }
        ++it_;
#endif
        if (!has_next) break;
        par = (par + nk) & 1;
        first = false;
        t = t_next;
        tile_m = tm_n;
        tile_n = tn_n;
    }
}

// super-column width of the tile order for a GEMM with `tiles_n` column tiles (0 = plain n-fastest); VH_PP_SN overrides
static int gemm_super_columns(int tiles_n) {
    static const int env = [] { const int e = env_int("VH_PP_SN", -1); return e < -1 ? -1 : e; }();   // -1 = automatic
    if (env >= 0) return env > 0 && env < tiles_n ? env : 0;
    // measured on ViT-B/16 b512 (tools/ab_supercol.sh): 9 column tiles (q|k|v) best at 3, 12 (fc1) at 4; 3 (proj, fc2)
    // need none.  Widths that divide tiles_n keep every super-column full.
    if (tiles_n < 6) return 0;
    if (tiles_n % 4 == 0) return 4;
    if (tiles_n % 3 == 0) return 3;
    return 4;
}

#ifdef VH_DIAG_STAMPS
// ring of the most recent launches' stamps (host bookkeeping; one device buffer of slots x max_wgs x 8 words)
struct DiagRec { int64_t M; int N, K, epi, f8, grid, variant; };
static unsigned long long* g_diag_buf = nullptr;
static int g_diag_slots = 0, g_diag_max_wgs = 0;
static long long g_diag_count = 0;
static DiagRec g_diag_rec[1024];
static unsigned long long* diag_next(const GemmArgs& g, int epi, bool f8, int grid, int variant) {
    if (!g_diag_buf || grid > g_diag_max_wgs) return nullptr;
    const int slot = (int)(g_diag_count % g_diag_slots);
    g_diag_rec[slot] = DiagRec{g.M, g.N, g.K, epi, f8 ? 1 : 0, grid, variant};
    ++g_diag_count;
    return g_diag_buf + (size_t)slot * g_diag_max_wgs * 128;
}
#define VH_STAMP_ARG(g, epi, f8, grid, variant) , diag_next(g, epi, f8, grid, variant)
#else
#define VH_STAMP_ARG(g, epi, f8, grid, variant)
#endif

// mode 0: one tile per workgroup (variant 5); 1: persistent (6); 2: one tile per workgroup, three A stages (7)
template <typename T, int EPI, bool F8, bool PERSIST, int AST, bool OT = false, bool AT = false>
static hipError_t launch_pp_one(const GemmArgs& g, int grid, int tiles_m, int tiles_n, hipStream_t s) {
    constexpr size_t lds = (AST == 3 || PERSIST) ? 163840 : 131072;
    auto k = gemm_nt_pp_kernel<T, EPI, PERSIST, F8, AST, OT, AT>;
    static LdsDone lds_done;  // per instantiation, per device
    if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, lds_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, s, g.a, g.w, g.bias, g.out, (int)g.M, g.N, g.K, g.aux, g.aux_i, tiles_m,
                       tiles_n, g.stats, g.out16, g.partials, PERSIST ? 0 : g.tile_begin, (g.tile_count || g.tile_begin) ? 0 : gemm_super_columns(tiles_n),
                       F8 ? (g.wscale ? g.wscale : g.aux) : (const float*)nullptr, gemm_prow(g)
                       VH_STAMP_ARG(g, EPI, F8, grid, PERSIST ? 6 : (AST == 3 ? 7 : 5)));
    return hipGetLastError();
}

template <typename T, int EPI, bool F8>
static hipError_t launch_pp(const GemmArgs& g, int mode, hipStream_t s) {
    if constexpr (EPI == VH_EPI_PATCH) { if (mode == 1) mode = 0; }   // the patch-row remap has no staged (full-line) epilogue
    const int tiles_m = (int)((g.M + 255) / 256), tiles_n = (g.N + 255) / 256;
    int ntiles = tiles_m * tiles_n;
    if (g.tile_count > 0) {   // a sub-range of the tiles (one tile per workgroup forms only)
        if (mode == 1 || g.tile_begin < 0 || g.tile_begin + g.tile_count > ntiles) return hipErrorInvalidValue;
        ntiles = g.tile_count;
    } else if (g.tile_begin != 0) {
        return hipErrorInvalidValue;
    }
    if (mode == 2) return launch_pp_one<T, EPI, F8, false, 3>(g, ntiles, tiles_m, tiles_n, s);
    if constexpr (EPI != VH_EPI_PATCH) {
        // the persistent form runs FULL tiles of at least two K-tiles only: N must be a multiple of the tile, and a ragged
        // last row of tiles (M % 256 != 0) goes to the one-tile-per-workgroup form as a tile range behind it
        const int full_m = (int)(g.M / 256);
        const int nk = g.K / (F8 ? 128 : 64);
        if (mode == 1 && (g.N % 256 != 0 || full_m == 0 || nk < (pp_uses_cpre<EPI, F8>() ? 4 : 2))) mode = 0;
        // A ragged last row of tiles would run as a second launch BEHIND the persistent one: one more tile time on 3-12
        // CUs while the rest idle (batch 128: fc2 2.74 ms per forward against 1.96 ms in the one-tile form).  The
        // persistent form gains ~3 %, so it is worth that only when forced (variant 6 asked for explicitly).
        if (mode == 1 && full_m != tiles_m && g.variant != 6) mode = 0;
        if ((g.out_tiled || g.ab_tiled) && (mode != 1 || full_m != tiles_m)) return hipErrorInvalidValue;   // tiled layouts: persistent form on whole tiles only
        if (mode == 1) {
            const int num_cu = device_num_cu();   // of the device this launch goes to (a group has one thread per device)
            if (!num_cu) return hipErrorUnknown;
            const int nfull = full_m * tiles_n;
            // (LNFOLD / BIAS with out_tiled: the q|k|v projection's HEAD-MAJOR result, gemm_epilogue.h)
            // (e4m3 operands: the GELU forms write the tiled e4m3 hidden activation, RESID_SPLIT reads it; LNFOLD's bf16 q|k|v goes head-major like the 16-bit path's)
            if constexpr (EPI == VH_EPI_LNFOLD_GELU || EPI == VH_EPI_BIAS_GELU || EPI == VH_EPI_LNFOLD || (!F8 && EPI == VH_EPI_BIAS)) {
                if (g.out_tiled) return g.ab_tiled ? hipErrorInvalidValue : launch_pp_one<T, EPI, F8, true, 2, true, false>(g, nfull < num_cu ? nfull : num_cu, full_m, tiles_n, s);
            }
            if constexpr (F8 ? EPI == VH_EPI_RESID_SPLIT : (EPI == VH_EPI_RESID_SPLIT || EPI == VH_EPI_BIAS)) {
                if (g.ab_tiled) return g.out_tiled ? hipErrorInvalidValue : launch_pp_one<T, EPI, F8, true, 2, false, true>(g, nfull < num_cu ? nfull : num_cu, full_m, tiles_n, s);
            }
            if (g.out_tiled || g.ab_tiled) return hipErrorInvalidValue;
            if (hipError_t e = launch_pp_one<T, EPI, F8, true, 2>(g, nfull < num_cu ? nfull : num_cu, full_m, tiles_n, s); e != hipSuccess) return e;
            if (full_m == tiles_m) return hipSuccess;
            GemmArgs tail = g;   // tiles [full_m * tiles_n, ntiles) of the n-fastest order = the ragged last row of tiles
            tail.tile_begin = nfull;
            tail.tile_count = tiles_n;
            return launch_pp_one<T, EPI, F8, false, 2>(tail, tiles_n, tiles_m, tiles_n, s);
        }
    }
    if (g.out_tiled || g.ab_tiled) return hipErrorInvalidValue;
    return launch_pp_one<T, EPI, F8, false, 2>(g, ntiles, tiles_m, tiles_n, s);
}

template <typename T, int EPI>
hipError_t launch_gemm_pingpong(const GemmArgs& g, int mode, hipStream_t s) {
    return launch_pp<T, EPI, false>(g, mode, s);
}

#define VH_INST(T)                                                                                     \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_BIAS>(const GemmArgs&, int, hipStream_t);       \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_BIAS_GELU>(const GemmArgs&, int, hipStream_t);  \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_BIAS_RESID>(const GemmArgs&, int, hipStream_t); \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_BIAS_F32>(const GemmArgs&, int, hipStream_t);   \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_PATCH>(const GemmArgs&, int, hipStream_t);       \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_LNFOLD>(const GemmArgs&, int, hipStream_t);      \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_LNFOLD_GELU>(const GemmArgs&, int, hipStream_t); \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_RESID_LN>(const GemmArgs&, int, hipStream_t);   \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_RESID_SPLIT>(const GemmArgs&, int, hipStream_t); \
    template hipError_t launch_gemm_pingpong<T, VH_EPI_PATCH_SPLIT>(const GemmArgs&, int, hipStream_t);
VH_INST(BF16)
VH_INST(FP16)

// fp8 operands: 16-bit results are bf16; fc1 writes e4m3 (gemm_epilogue8)
hipError_t launch_gemm_fp8(const GemmArgs& g, hipStream_t s) {
    const int v = g.variant ? g.variant : gemm_pp_variant(g.epilogue);
    const int mode = v == 7 ? 2 : (v == 6 && !g.tile_count && !g.tile_begin ? 1 : 0);
    switch (g.epilogue) {
        case VH_EPI_BIAS: return launch_pp<BF16, VH_EPI_BIAS, true>(g, mode, s);
        case VH_EPI_BIAS_GELU: return launch_pp<BF16, VH_EPI_BIAS_GELU, true>(g, mode, s);
        case VH_EPI_BIAS_RESID: return launch_pp<BF16, VH_EPI_BIAS_RESID, true>(g, mode, s);
        case VH_EPI_BIAS_F32: return launch_pp<BF16, VH_EPI_BIAS_F32, true>(g, mode, s);
        // folded LayerNorm on e4m3 operands: `aux` = c_n, `wscale` = the weight scales, `stats` = (mean, rstd) per row
        case VH_EPI_LNFOLD: return g.wscale && g.stats && g.aux ? launch_pp<BF16, VH_EPI_LNFOLD, true>(g, mode, s) : hipErrorInvalidValue;
        case VH_EPI_LNFOLD_GELU: return g.wscale && g.stats && g.aux && g.N % 256 == 0 ? launch_pp<BF16, VH_EPI_LNFOLD_GELU, true>(g, mode, s) : hipErrorInvalidValue;
        case VH_EPI_RESID_LN: return g.out16 && g.partials && g.N % 256 == 0 ? launch_pp<BF16, VH_EPI_RESID_LN, true>(g, mode, s) : hipErrorInvalidValue;
        // split residual of the fp8 path: out = the e4m3 hi plane (the next GEMM's operand), out16 = the bf16 lo plane
        case VH_EPI_RESID_SPLIT: return g.out16 && g.partials && g.N % 256 == 0 ? launch_pp<BF16, VH_EPI_RESID_SPLIT, true>(g, mode, s) : hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

}  // namespace vh

#ifdef VH_DIAG_STAMPS
// exported by libvithip_diag.so only (not declared in include/vithip.h; bound by tools/gemm_anatomy.py)
extern "C" int vh_diag_stamps_arm(int slots, int max_wgs) {
    using namespace vh;
    if (slots < 1 || slots > 1024 || max_wgs < 1) return 1;
    if (g_diag_buf) { hipFree(g_diag_buf); g_diag_buf = nullptr; }
    if (hipMalloc((void**)&g_diag_buf, (size_t)slots * max_wgs * 1024) != hipSuccess) return 2;
    if (hipMemset(g_diag_buf, 0, (size_t)slots * max_wgs * 1024) != hipSuccess) return 2;
    g_diag_slots = slots; g_diag_max_wgs = max_wgs; g_diag_count = 0;
    return 0;
}
// age 0 = the most recent launch; copies grid x 8 waves x 16 words (0-7 tile stamps, 8-15 phase stamps of one K-tile); meta = {M, N, K, epi, f8, grid, variant}
extern "C" int vh_diag_stamps_read(int age, unsigned long long* host, int max_wgs, long long* meta) {
    using namespace vh;
    if (!g_diag_buf || age < 0 || age >= g_diag_slots || age >= g_diag_count) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    const int slot = (int)((g_diag_count - 1 - age) % g_diag_slots);
    const DiagRec& r = g_diag_rec[slot];
    if (r.grid > max_wgs) return 3;
    if (hipMemcpy(host, g_diag_buf + (size_t)slot * g_diag_max_wgs * 128, (size_t)r.grid * 1024, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    meta[0] = r.M; meta[1] = r.N; meta[2] = r.K; meta[3] = r.epi; meta[4] = r.f8; meta[5] = r.grid; meta[6] = r.variant;
    return 0;
}
extern "C" long long vh_diag_stamps_count(void) { return vh::g_diag_count; }
#endif
