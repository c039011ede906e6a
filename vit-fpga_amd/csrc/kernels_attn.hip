// kernels_attn.hip — fused multi-head attention of the ViT hot path (gfx950).
//
//   out[b, t, h*64:(h+1)*64] = softmax( q_h k_h^T ) v_h        (q pre-scaled by 64^-1/2 * log2(e): the
//                                                               scores leave the MFMA in the exp2 domain)
//
// reads the fused projection output qkv [batch*T, 3*H*64] (16-bit), writes [batch*T, H*64].
// Head dimension is fixed at 64 (ViT-Ti/S/B/L/H all use 64).
//
// Design (CDNA4):
//   * work item = (image, head, query slab of NW*32 rows); each wave owns 32 query rows.
//   * the whole K and V of the head are staged into LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR
//     round trip); K is XOR-swizzled for conflict-free ds_read_b128 row reads, V for conflict-free
//     ds_read_b64_tr_b16 transposed reads — the DMA destination is linear, so both swizzles are applied to
//     the lane's global SOURCE address.
//   * persistent workgroups (when two K/V images fit in LDS): a workgroup walks its items and DMA-prefetches
//     the next item's K/V into the other LDS buffer (and its Q fragments into registers) while it computes
//     the current one.  The kernel is HBM-bound (620 MB per launch for ViT-B/16 at batch 512); the one-shot
//     form (load, barrier, compute, store) only reached 55 % of the HBM rate because each workgroup's loads
//     and math serialise and only two workgroups fit per CU.
//   * S^T = K Q^T with v_mfma_f32_32x32x16 (keys on the accumulator rows, the query on the lane): a query's
//     scores live in ONE lane pair (l, l^32), so the softmax row max / sum are 15 in-register ops + one
//     cross-half shuffle — a wavefront reduction, no LDS.
//   * the S^T accumulator, converted to 16-bit in registers, IS the B operand of O^T = V^T P^T (same lane,
//     k-order of the accumulator rows); V^T fragments come from the hardware transposing LDS read.
//     Scores/probabilities never touch LDS or HBM.
//   * online softmax over 32-key tiles in the exp2 domain, fp32 statistics, keys >= T masked.
#include <cstdlib>
#include <type_traits>

#include "vh_kernels.h"

namespace vh {

// the running shift of a query row is re-centred only when a tile's scores exceed it by more than 2^kTau: between
// re-centrings p = exp2(s - shift) <= 2^kTau, which fp32 sums and 16-bit P operands (fp16 max 65504) hold easily
constexpr float kTau = 8.0f;

// lanes l and l^32 hold the two halves of a query's row: combine them with one v_permlane32_swap
// (a VALU exchange of the wave's halves) instead of a ds_bpermute round trip through the LDS crossbar
// v_permlane32_swap vdst, src exchanges lanes 32-63 of vdst with lanes 0-31 of src IN PLACE, so the two
// operands must be different registers: with one register it degenerates to a plain half swap.  The copy is
// made opaque so the compiler cannot fold the operands back together.
__device__ __forceinline__ void swap_halves(float v, float& lo_everywhere, float& hi_everywhere) {
    // Whole exchange in one asm statement (hipcc 7.2 returned the same register for both results of
    // __builtin_amdgcn_permlane32_swap here).  s_nop 1 = the 2 wait states of the "VALU write ->
    // v_permlane read" hazard, which nobody pads inside inline asm; the trailing one covers the readers.
    float a = v, b;
    asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "=&v"(b));
    lo_everywhere = a;  // {v[0..31], v[0..31]}
    hi_everywhere = b;  // {v[32..63], v[32..63]}
}
__device__ __forceinline__ float cross_half_max(float v) {
    float a, b;
    swap_halves(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float cross_half_sum(float v) {
    float a, b;
    swap_halves(v, a, b);
    return a + b;
}

template <typename T, bool PERSIST, typename TO = T>   // TO: output element (T, or E4M3 on the fp8 path)
__global__ void __launch_bounds__(1024, PERSIST ? 2 : 4)   // one-shot form: <= 128 VGPRs, two 7-wave workgroups per CU put 4 waves on a SIMD
attention_kernel(const typename T::elem* __restrict__ qkv, typename TO::elem* __restrict__ out,
                 int tokens, int heads, int slabs, int ntiles, int nitems) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kv_bytes = ntiles * 4096;      // one K (or V) image
    const int buf_bytes = 2 * kv_bytes;      // K then V

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int l31 = lane & 31, hl = lane >> 5;
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;

    // ---- DMA of one item's K and V: 1 KiB pieces = 8 rows x 128 B, lane -> (row, swizzled chunk) -------
    const int lr = lane >> 3, pc = lane & 7;
    const int ngroups = ntiles * 4;          // 8-row groups per matrix
    auto issue_kv = [&](int item, int buf) {
        const int bh = item / slabs;
        const int b = bh / heads, h = bh - b * heads;
        const elem* base = qkv + (int64_t)b * tokens * ld + h * 64;
        char* kdst = smem + buf * buf_bytes;
        for (int g = wave; g < ngroups; g += nw) {
            const int row = g * 8 + lr;
            const int rsrc = row < tokens ? row : tokens - 1;   // rows >= tokens replicate the last row
            const elem* src = base + rsrc * ld;
            const int ck = pc ^ ((row >> 1) & 7);
            const int cv = pc ^ (((row >> 1) & 1) << 2);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + D + ck * 8),
                                             (void __attribute__((address_space(3)))*)(kdst + g * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 2 * D + cv * 8),
                                             (void __attribute__((address_space(3)))*)(kdst + kv_bytes + g * 1024), 16, 0, 0);
        }
    };
    auto load_q = [&](int item, vec8 (&qf)[4]) {
        const int bh = item / slabs, slab = item - bh * slabs;
        const int b = bh / heads, h = bh - b * heads;
        int qrow = (slab * nw + wave) * 32 + l31;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const elem* qp = qkv + ((int64_t)b * tokens + qrow) * ld + h * 64 + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const vec8*)(qp + 16 * ks);
    };

    // per-lane LDS offsets
    const int koff = l31 * 128;
    const int kswz = (l31 >> 1) & 7;
    const int g4 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int vrow0 = 4 * (g4 >> 1) + tq;             // + 32*kt + 16*s (+8)
    const int vcolb = (16 * (g4 & 1) + 4 * tp) * 2;   // byte offset inside the row, + 64*db

    const int stride = PERSIST ? (int)gridDim.x : nitems;
    int item = blockIdx.x;
    int buf = 0;
    vec8 qf[4], qn[4];
    issue_kv(item, 0);
    load_q(item, qf);

    while (true) {
        const int next = item + stride;
        const bool has_next = PERSIST && next < nitems;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this item's K/V (and Q) have landed
        __syncthreads();                                  // ... for every wave; previous item's reads are done
        if (has_next) {
            issue_kv(next, buf ^ 1);                      // flies under this item's math
            load_q(next, qn);
        }

        const int bh = item / slabs, slab = item - bh * slabs;
        const int b = bh / heads, h = bh - b * heads;
        const int q0 = (slab * nw + wave) * 32;
        const char* Ks = smem + buf * buf_bytes;
        const char* Vs = Ks + kv_bytes;

        if (q0 < tokens) {  // wave-uniform
            f32x16 o0, o1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
            float m2 = -INFINITY;  // running row max, log2 domain
            float lsum = 0.f;      // this lane's half of the row sum

            // lane-constant LDS addresses: K row reads (4 swizzled chunks) and the transposed V reads
            // (two bases: the 32-column block db flips bit 6 of the swizzled byte offset)
            const char* kbase = Ks + koff;
            int kofs[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kofs[ks] = ((2 * ks + hl) ^ kswz) << 4;
            const int vx = ((vrow0 >> 1) & 1) << 6;   // same for rows +8, +16, +32*kt
            const char* vb0 = Vs + vrow0 * 128 + (vcolb ^ vx);
            const char* vb1 = Vs + vrow0 * 128 + ((vcolb + 64) ^ vx);

            // S^T tile: 32 keys x 32 queries
            auto qk = [&](int kt) {
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
                const char* kp = kbase + kt * 4096;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = T::mfma32(*(const vec8*)(kp + kofs[ks]), qf[ks], s);
                return s;
            };
            // V^T fragments of one 32-key tile: [k-step][column block] (hardware-transposing reads)
            struct VFrag { vec8 f[2][2]; };
            auto load_v = [&](int kt) {
                VFrag v;
                const char* v0 = vb0 + kt * 4096;
                const char* v1 = vb1 + kt * 4096;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const vec4 a0 = T::tr_read(v0 + ks * 2048), c0 = T::tr_read(v0 + ks * 2048 + 1024);
                    const vec4 a1 = T::tr_read(v1 + ks * 2048), c1 = T::tr_read(v1 + ks * 2048 + 1024);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v.f[ks][0][j] = a0[j]; v.f[ks][0][4 + j] = c0[j]; v.f[ks][1][j] = a1[j]; v.f[ks][1][4 + j] = c1[j]; }
                }
                return v;
            };
            auto softmax_pv = [&](int kt, f32x16 s, const VFrag& vfr, auto masked) {
                if constexpr (decltype(masked)::value) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                        if (key >= tokens) s[r] = -INFINITY;
                    }
                }
                // ---- online softmax (row = query = lane pair) --------------------------------------------
                float mx = s[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
                mx = cross_half_max(mx);
                const float mnew = fmaxf(m2, mx);
                // rescale the running sums only when some row's maximum moved (exact: alpha == 1 otherwise)
                if (__builtin_amdgcn_ballot_w64(mnew != m2)) {
                    const float alpha = __builtin_amdgcn_exp2f(m2 - mnew);
                    lsum *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                    m2 = mnew;
                }
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(s[r] - mnew);
                    psum += s[r];
                }
                lsum += psum;
                // ---- O^T += V^T P^T ----------------------------------------------------------------------
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    vec8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (elem)s[8 * ks + j];
                    o0 = T::mfma32(vfr.f[ks][0], pf, o0);
                    o1 = T::mfma32(vfr.f[ks][1], pf, o1);
                }
            };
            // software pipeline: the V fragments of tile k and the score MFMAs of tile k+1 are issued ahead of the
            // softmax of tile k, so LDS latency and the matrix pipe sit under the wave's own exp/max/convert stream
            f32x16 s_cur = qk(0);
            for (int kt = 0; kt + 1 < ntiles; ++kt) {
                const VFrag vfr = load_v(kt);
                const f32x16 s_nxt = qk(kt + 1);
                softmax_pv(kt, s_cur, vfr, std::false_type{});
                s_cur = s_nxt;
            }
            {
                const VFrag vfr = load_v(ntiles - 1);
                softmax_pv(ntiles - 1, s_cur, vfr, std::true_type{});   // only the last tile can hold keys >= tokens
            }

            // ---- normalise and store: lane holds O[q][32*db + 8*rg + 4*hl + 0..3] ------------------
            const float ltot = cross_half_sum(lsum);
            const float inv = 1.0f / ltot;
            const int q = q0 + l31;
            if (q < tokens) {
                typename TO::elem* op = out + ((int64_t)b * tokens + q) * D + h * 64 + 4 * hl;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    *(typename TO::vec4*)(op + 8 * rg) = pack4<TO>(o0[4 * rg] * inv, o0[4 * rg + 1] * inv, o0[4 * rg + 2] * inv, o0[4 * rg + 3] * inv);
                    *(typename TO::vec4*)(op + 32 + 8 * rg) = pack4<TO>(o1[4 * rg] * inv, o1[4 * rg + 1] * inv, o1[4 * rg + 2] * inv, o1[4 * rg + 3] * inv);
                }
            }
        }
        if (!has_next) break;
        item = next;
        buf ^= 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = qn[ks];
    }
}

// ---- ring form -------------------------------------------------------------------------------------------------------
// Persistent workgroups with ONE K/V image in LDS that is refilled tile by tile: as soon as every wave is done with
// key tile t-1 of the current item (a workgroup barrier at the top of tile t), waves 0-3 DMA the NEXT item's tile t-1
// into the freed slot.  K/V traffic therefore flows during the whole of an item's math instead of in a burst before
// it (the one-shot form: load, barrier, compute, store; only the co-resident second workgroup overlapped them), at
// the LDS cost of one image (2 workgroups per CU at T = 197, as before).  All waits are vmcnt(0) placed where the
// youngest outstanding DMA is at least one tile old:
//   item top        : slots 0..nt-2 (issued during the previous item) + the next Q fragments  -> barrier -> slot nt-1 issued
//   top of tile nt-1: slot nt-1
// Softmax without per-score shifts: q arrives scaled by 64^-1/2 * log2(e) and the score accumulators START at
// -shift (the row's tile-0 maximum), so a tile's probabilities are exp2(acc) directly; the shift moves (rescaling
// O and l) only when a later tile exceeds it by 2^kTau.  The last tile computes only its valid 8-key groups.
#ifndef VH_ATTN_ABL
#define VH_ATTN_ABL 0   // timing ablations (tools/ab_attn_abl.sh; results are wrong): 1 no exp, 2 no refill DMA, 4 no tile barriers, 8 no PV, 16 no QK, 32 no LDS reads, 64 no max
#endif
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS reads of the slot about to be refilled are done
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
// One LDS-DMA instruction (64 lanes x 16 B -> 1 KiB at the LDS address in M0), written as inline asm on purpose: hipcc
// makes every ds_read_b64_tr_b16 (an intrinsic without a memory operand) wait for vmcnt(0) while a global_load_lds
// it knows of is in flight, which would drain the ring at every tile.  The kernel's own waits (ring_wait_dma) cover
// the DMA; the compiler's vmcnt arithmetic for ordinary loads only gets stricter by not knowing these.  The statement
// itself, with the wait states its scalar operands need, lives in vh_common.h (asm_lds_dma16).
__device__ __forceinline__ void ring_dma16(const void* base, uint32_t lane_off, uint32_t lds_addr) {
    asm_lds_dma16(base, lane_off, lds_addr);
}
#ifdef VH_DIAG_STAMPS
// diagnostic build only (tools/attn_anatomy.py): per-wave shader-clock totals of the phases of the ring kernel
__device__ unsigned long long g_attn_diag[1024 * 16 * 16];   // per wave: 10 phase totals, start/end s_memrealtime, hw id, 3 debug words
__device__ __forceinline__ unsigned long long attn_rt() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ unsigned long long attn_ct() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define VH_ATT_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = attn_ct(); ph_[i] += n_ - last_; last_ = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define VH_ATT_T(i) do { } while (0)
#endif
// s_waitcnt vmcnt(n) for a wave-uniform run-time n (0..40: the counts of the staged form for up to 14 key tiles)
__device__ __forceinline__ void ring_wait_vm(int n) {
#define VH_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        VH_W(1) VH_W(2) VH_W(3) VH_W(4) VH_W(5) VH_W(6) VH_W(7) VH_W(8) VH_W(9) VH_W(10) VH_W(11) VH_W(12) VH_W(13) VH_W(14)
        VH_W(15) VH_W(16) VH_W(17) VH_W(18) VH_W(19) VH_W(20) VH_W(21) VH_W(22) VH_W(23) VH_W(24) VH_W(25) VH_W(26) VH_W(27)
        VH_W(28) VH_W(29) VH_W(30) VH_W(31) VH_W(32) VH_W(33) VH_W(34) VH_W(35) VH_W(36) VH_W(37) VH_W(38) VH_W(39) VH_W(40)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef VH_W
}
__device__ __forceinline__ void ring_wait_dma() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// QS ("Q staged", 96 < T <= ~208, one slab): the NEXT item's Q rows are DMA'd into an LDS staging area early in the
// current item (tile 1) and each wave reads its four fragments from there at the item top, and the K/V images are
// trimmed to ceil(T/8) 8-row groups (3 x 25 + 3 KiB = 78 KiB at T = 197: still two workgroups per CU).  Nothing a wave
// waits for is then younger than about five tiles: every wait is a COUNTED vmcnt that leaves the younger DMAs and the
// previous item's output stores in flight.  Per DMA wave and item i (nt tiles, nw waves) the issue order is
//   item top : slot nt-1 (i)                  2      tile 1 top: Q (i+1) nw, slot 0 (i+1) 2
//   tile t top, t = 2..nt-1: slot t-1 (i+1)   2      item end  : 8 stores (NST in the code: 4 since round 3 for 16-bit results)
// so a wait for X may leave outstanding:  item top (Q, slot 0): 2(nt-2) + 8;  tile 1 (slot 1): 2(nt-3) + 8 + 2;
// tile t in 2..nt-2 (slot t): 2nt + nw + 4;  last tile (slot nt-1): nw + 2 + 2(nt-3).  The last item of a workgroup
// issues no refills and waits with vmcnt(0).
// QS with a ticket counter (`ticket` != null, nw >= 5): a workgroup's first two items are static (blockIdx, + grid), the
// rest are drawn from a device counter zeroed before the launch.  The two workgroups of a CU do NOT progress equally
// (the older one wins the issue arbitration; measured lifetimes 108-161 us for equal static shares), so equal shares
// left every CU half empty for the last quarter of the launch.  The last wave (not a DMA wave) draws the ticket for item
// i+2 at the top of item i; the result is picked up at the top of item i+1 (vmcnt(8): its stores are younger), handed
// to the other waves through one LDS word across the item-top barrier, and becomes that item's prefetch target.
// OTILE (16-bit results): the output goes to the 16-row-blocked layout [m / 16][D / 8 chunks][16 rows][8 values] that the
// out-projection's operand DMA reads as 1 KiB of contiguous source per instruction (kernels_gemm5.hip AT): the 32 query rows a
// half-wave stores per instruction are then two or three contiguous runs of 16-byte pieces instead of 32 pieces a row apart.
// IHM (round 4): q|k|v arrives HEAD-MAJOR, [3][heads][hm_rows][64] -- the projection's epilogue writes it so (gemm_epilogue.h: a
// wave's 64 output columns are exactly one head's q, k or v) -- instead of [rows][3 * heads * 64]: a head's rows are 128 B apart,
// so every DMA piece (8 rows) is ONE KiB of contiguous source instead of eight 128-byte segments 3 * D * 2 bytes apart.
template <typename T, typename TO = T, bool QS = false, bool OTILE = false, bool IHM = false>
__global__ void __launch_bounds__(1024, 4)
attention_ring_kernel(const typename T::elem* __restrict__ qkv, typename TO::elem* __restrict__ out,
                      int tokens, int heads, int slabs, int ntiles, int nitems, unsigned int* __restrict__ ticket,
                      unsigned int heads_rcp, int64_t hm_rows) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // K image, then V image; slot t = bytes [t * 4096, (t + 1) * 4096) of each.  QS: images of G = ceil(T/8) groups (the
    // last tile's missing groups alias the start of the next area: finite data under masked keys), then the Q staging
    // (K rows of garbage only make garbage scores for masked keys; V gets an even number of groups so that the 16-key
    // k-step of the last tile multiplies its zero probabilities with real, finite rows)
    const int G = (tokens + 7) >> 3, G2 = (G + 1) & ~1;
    const int kv_bytes = QS ? G * 1024 : ntiles * 4096;          // K image = offset of the V image
    const int q_off = kv_bytes + (QS ? G2 * 1024 : kv_bytes);    // end of the V image = offset of the Q staging (QS)
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;   // LDS byte address of the ring

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int l31 = lane & 31, hl = lane >> 5;
    const int D = heads * 64;
    const int64_t ld = IHM ? 64 : 3 * (int64_t)D;
    // byte offset of the K plane behind the Q plane (V: twice that): the next D columns of the row, or the next `heads` head blocks
    const uint32_t koff2 = IHM ? (uint32_t)(hm_rows * D * 2) : (uint32_t)(D * 2);
    auto head_base = [&](int b, int h) {
        if constexpr (IHM) return qkv + ((int64_t)h * hm_rows + (int64_t)b * tokens) * 64;
        else return qkv + (int64_t)b * tokens * ld + h * 64;
    };

    // ---- DMA of one key tile (32 rows of K and of V = 8 pieces of 1 KiB): waves 0-3 move one K and one V piece each
    auto kv_base = [&](int item) {
        if (QS) {   // one slab; b = item / heads by the host's reciprocal ceil(2^32 / heads) (exact for item * heads < 2^32; heads == 1 has no 32-bit reciprocal)
            const int b = (heads == 1 ? item : (int)__umulhi((unsigned)item, heads_rcp)), h = item - b * heads;
            return head_base(b, h);
        }
        const int bh = item / slabs;
        const int b = bh / heads, h = bh - b * heads;
        return head_base(b, h);
    };
    // Source address = wave-uniform base (scalar registers) + a 32-bit per-lane byte offset that is recomputed from the
    // lane id at every issue (the id is laundered so that the compiler cannot hoist twelve address registers out of
    // the item loop and spill them: a spill reload inside the loop would wait for every DMA in flight).
    auto issue_tile = [&](const elem* base, int t) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lr_ = ln >> 3, pc_ = ln & 7;
        // trimmed images (QS): a surplus issue re-loads the image's last group (keeps the DMA count per slot uniform)
        const int g = 4 * t + wave;
        const int gk = QS && g >= G ? G - 1 : g, gv = QS && g >= G2 ? G2 - 1 : g;
        const int rowk = gk * 8 + lr_, rowv = gv * 8 + lr_;
        const int rk = rowk < tokens ? rowk : tokens - 1;   // rows >= tokens replicate the last row
        const int rv = rowv < tokens ? rowv : tokens - 1;
        const int ck = pc_ ^ ((rowk >> 1) & 7);
        const int cv = pc_ ^ (((rowv >> 1) & 1) << 2);
        const uint32_t ok = (uint32_t)(rk * (int)ld + ck * 8) * 2u + koff2;
        const uint32_t ov = (uint32_t)(rv * (int)ld + cv * 8) * 2u + 2u * koff2;
        ring_dma16(base, ok, lds0 + gk * 1024);
        ring_dma16(base, ov, lds0 + kv_bytes + gv * 1024);
    };
    // QS: the Q rows of an item (4 * nw groups, rows >= tokens replicate the last) -> staging, K-style swizzle
    auto issue_q = [&](const elem* base) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lr_ = ln >> 3, pc_ = ln & 7;
        for (int j = 0; j < nw; ++j) {
            const int g = 4 * j + wave;
            const int row = g * 8 + lr_;
            const int rsrc = row < tokens ? row : tokens - 1;
            const int ck = pc_ ^ ((row >> 1) & 7);
            ring_dma16(base, (uint32_t)(rsrc * (int)ld + ck * 8) * 2u, lds0 + q_off + g * 1024);
        }
    };
    auto load_q = [&](int item, vec8 (&qf)[4]) {
        const int bh = item / slabs, slab = item - bh * slabs;
        const int b = bh / heads, h = bh - b * heads;
        int qrow = (slab * nw + wave) * 32 + l31;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const elem* qp = head_base(b, h) + (int64_t)qrow * ld + 8 * hl;
        // Inline asm, in place: the item-top wait leaves the 8 output stores issued behind these loads in flight
        // (vmcnt(8)), which hipcc's own accounting for an ordinary load cannot express across the predicated store block
        // (it falls back to vmcnt(0) = wait for the stores just issued, every item).  ring_wait_item() orders the uses.
        asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                     "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                     : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]) : "v"(qp) : "memory");
    };

    // per-lane LDS addresses: K row reads (4 swizzled chunks) and the transposed V reads (two bases: the 32-column
    // block db flips bit 6 of the swizzled byte offset)
    const int kswz = (l31 >> 1) & 7;
    const int g4 = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int vrow0 = 4 * (g4 >> 1) + tq;             // + 32*kt + 16*s (+8)
    const int vcolb = (16 * (g4 & 1) + 4 * tp) * 2;   // byte offset inside the row, + 64*db
    // kept as two byte offsets; the other K chunks and the second V column block are XORs of them (bits 5-6)
    const int k0 = l31 * 128 + ((hl ^ kswz) << 4);    // chunk (2*ks + hl) ^ kswz = this ^ (ks << 5)
    const int vx = ((vrow0 >> 1) & 1) << 6;           // same for rows +8, +16, +32*kt
    const int v0 = kv_bytes + vrow0 * 128 + (vcolb ^ vx);   // column block 1: ^ 64  (vcolb < 64)

    // A workgroup walks HEADS (blockIdx, + grid, ...) and, inside a head, its query slabs back to back: the K/V image
    // loaded for slab 0 serves every slab of the head (T = 577: two), and the ring is refilled with the next head's rows
    // only during the head's LAST slab.  item = head * slabs + slab throughout.
    const int stride = (int)gridDim.x;
    int item = blockIdx.x * slabs;
    const elem* cbase = kv_base(item);
    vec8 qf[4];
    if (wave < 4) {
        if (QS) issue_q(cbase);
        for (int t = 0; t < ntiles; ++t) issue_tile(cbase, t);
    }
    if (!QS) load_q(item, qf);
    bool first = true;
    // QS: outstanding operations a wait may leave behind (see the table above)
    // NST = output store instructions per wave and item: 4 x 16 bytes per lane for 16-bit results (the widened form at the
    // item end), 8 x 4 bytes for e4m3.  Every "8" of the table above is NST here.
    constexpr int NST = 4;   // (16-bit results: 4 x 16 bytes per lane; e4m3: 4 x 8 bytes)
    const int w_top = 2 * (ntiles - 2) + NST, w_t1 = 2 * (ntiles - 3) + NST + 2, w_mid = 2 * ntiles + nw - 4 + NST, w_last = nw + 2 + 2 * (ntiles - 3);
    bool stored = false;   // did this wave issue the NST stores of the previous item (wave-uniform)

#ifdef VH_DIAG_STAMPS
    const unsigned long long rt0_ = attn_rt();
    unsigned int dbg_[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    int dbg_n_ = 0;
    unsigned long long ph_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = attn_ct();
#endif
    const bool dyn = QS && ticket != nullptr && nw >= 5;   // wave-uniform
    const int tk_off = q_off + 4 * nw * 1024;              // the LDS word behind the Q staging
    unsigned int tkv = 0;                                  // ticket wave: the ticket in flight
    while (true) {
        const int slab_now = QS ? 0 : item - (item / slabs) * slabs;
        const bool last_slab = QS || slab_now + 1 == slabs;
        int next = QS ? item + stride : (last_slab ? (item / slabs + stride) * slabs : item + 1);
        if (QS && dyn && !first) {
            if (wave == nw - 1) {   // the ticket drawn at the top of the previous item (older than that item's 8 stores)
                if (stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(tkv));
                if (lane == 0) *(volatile unsigned int __attribute__((address_space(3)))*)(uintptr_t)(lds0 + tk_off) = 2u * (unsigned)stride + tkv;
            }
        }
        const int b = QS ? (heads == 1 ? item : (int)__umulhi((unsigned)item, heads_rcp)) : (item / slabs) / heads;
        const int slab = slab_now;
        const int h = (QS ? item : item / slabs) - b * heads;
        const int q0 = (slab * nw + wave) * 32;

        // slots 0..nt-2 of this item (first item: all) and Q have landed; the previous item's 8 output stores, the
        // youngest operations of a wave that had rows to store, stay in flight
        VH_ATT_T(7);   // stores, bookkeeping
        if (QS) {
            ring_wait_vm(first ? 0 : w_top);   // (everything it may leave behind was issued during the previous item)
        } else {
            if (first || !stored) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NST) : "memory");
            asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));   // uses of the Q fragments stay behind the wait
        }
        VH_ATT_T(0);   // item-top wait
        ring_barrier();                                   // ... for every wave; everyone is done with the previous item
        VH_ATT_T(1);   // item-top barrier
        if (QS && dyn && !first) next = __builtin_amdgcn_readfirstlane(*(volatile const int __attribute__((address_space(3)))*)(uintptr_t)(lds0 + tk_off));
#if defined(VH_DIAG_STAMPS) && defined(VH_ATTN_TKDBG)
        // ticket debugging: record what the exchange delivered, but keep walking the static shares
        if (QS && dyn && !first) { if (dbg_n_ < 3) dbg_[dbg_n_++] = (unsigned)next; next = item + stride; }
#endif
        const bool has_next = (unsigned)next < (unsigned)nitems;
        const bool refill = has_next && last_slab;            // the next item belongs to another head: stream its K/V in
        const elem* nbase = kv_base(has_next ? next : item);
        if (QS && dyn && has_next && wave == nw - 1 && lane == 0) {   // draw the ticket of the item after `next` (one lane)
            // (vh_common.h: the statement opens with s_nop 4 -- without it the atomic went out with a stale upper address
            //  half after an SGPR reload and the launch died with a memory aperture violation)
            asm_atomic_add_ret_issue(ticket, 1u, tkv);
        }
        if (!(VH_ATTN_ABL & 2) && !first && slab == 0 && wave < 4) issue_tile(cbase, ntiles - 1);   // (a later slab finds the whole image in place)
        if (QS) {   // this wave's Q fragments out of the staging area (free again after the barrier at the top of tile 1)
            int qa = q_off + wave * 4096 + k0;
            if constexpr (!std::is_same<T, TO>::value) {
                // e4m3 results: one register short at 128 -- the four lane-constant addresses below were hoisted out of the item
                // loop and SPILLED, and their reloads (vector-memory operations) made every item top wait for vmcnt(0).
                // Recomputed from a laundered lane id per item instead (3 VALU operations).
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const int r31 = ln & 31;
                qa = q_off + wave * 4096 + r31 * 128 + ((((ln >> 5)) ^ ((r31 >> 1) & 7)) << 4);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const vec8*)(smem + (qa ^ (ks << 5)));
        }

        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        float negm = 0.f;      // - (shift of this lane's query row), exp2 domain
        float lsum = 0.f;      // this lane's half of the row sum

        // S^T tile: 32 keys x 32 queries, accumulators start at `init` (= -shift)
        auto qk = [&](int kt, float init) {
            // all four K fragments requested up front (hipcc, register-shy at this occupancy, otherwise reads one, waits, multiplies)
            int ka = k0 + kt * 4096;
            // (e4m3 results: the three other chunk addresses are XORs of this one; kept opaque so that they are recomputed per
            //  tile instead of being hoisted into three more registers this instantiation does not have)
            if constexpr (!std::is_same<T, TO>::value) asm volatile("" : "+v"(ka));
            vec8 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (VH_ATTN_ABL & 32) kf[ks] = qf[ks]; else if (VH_ATTN_ABL & 128) kf[ks] = *(const vec8*)(smem + ((ka ^ (ks << 5)) & 0x3ff0)); else
                kf[ks] = *(const vec8*)(smem + (ka ^ (ks << 5)));
            }
            __builtin_amdgcn_sched_barrier(0);
            // 8 x v_mov_b64 of the (init, init) pair: left to itself hipcc splats with 16 v_mov_b32 + 8 v_mov_b64
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef float f32x8 __attribute__((ext_vector_type(8)));
            const f32x2 i2 = {init, init};
            f32x2 p[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(i2));
            const f32x4 q0 = __builtin_shufflevector(p[0], p[1], 0, 1, 2, 3), q1 = __builtin_shufflevector(p[2], p[3], 0, 1, 2, 3);
            const f32x4 q2 = __builtin_shufflevector(p[4], p[5], 0, 1, 2, 3), q3 = __builtin_shufflevector(p[6], p[7], 0, 1, 2, 3);
            const f32x8 h0 = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), h1 = __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7);
            f32x16 s = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (VH_ATTN_ABL & 16) s[ks] += (float)kf[ks][0] * (float)qf[ks][0];
                else s = T::mfma32(kf[ks], qf[ks], s);
            }
            return s;
        };
        struct VFrag { vec8 f[2][2]; };   // V^T fragments of one 32-key tile: [k-step][column block]
        auto load_v = [&](int kt) {
            VFrag v;
            if (VH_ATTN_ABL & 32) { for (int ks = 0; ks < 2; ++ks) { v.f[ks][0] = qf[ks]; v.f[ks][1] = qf[ks + 2]; } return v; }
            const char* p0 = smem + (v0 + kt * 4096);
            const char* p1 = smem + ((v0 + kt * 4096) ^ 64);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const vec4 a0 = T::tr_read(p0 + ks * 2048), c0 = T::tr_read(p0 + ks * 2048 + 1024);
                const vec4 a1 = T::tr_read(p1 + ks * 2048), c1 = T::tr_read(p1 + ks * 2048 + 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v.f[ks][0][j] = a0[j]; v.f[ks][0][4 + j] = c0[j]; v.f[ks][1][j] = a1[j]; v.f[ks][1][4 + j] = c1[j]; }
            }
            return v;
        };
        // one tile of softmax + O^T += V^T P^T.  `s` holds scores - shift.
        auto softmax_pv = [&](int kt, f32x16& s, const VFrag& vfr, auto first_c, auto tail_c) {
            constexpr bool FIRST = decltype(first_c)::value, TAIL = decltype(tail_c)::value;
            int ng = 4;                                   // 8-key groups of this tile that hold keys < tokens
            if constexpr (TAIL) {
                const int valid = tokens - kt * 32;
                ng = (valid + 7) >> 3;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (g < ng) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (8 * g + j + 4 * hl >= valid) s[4 * g + j] = -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (!TAIL || g < ng) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[4 * g + j]);
                }
            if (VH_ATTN_ABL & 64) mx = s[kt & 15];
            else
            mx = cross_half_max(mx);
            if constexpr (FIRST) {
                negm = -mx;                               // the row's shift = its maximum over tile 0
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] -= mx;
            } else if (__builtin_amdgcn_ballot_w64(mx > kTau)) {   // rare: some row outgrew its shift by 2^kTau
                const float delta = fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                negm -= delta;
                lsum *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; s[r] -= delta; }
            }
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 psum = {0.f, 0.f};                      // two chains, packed adds
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!TAIL || g < ng) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[4 * g + j] = (VH_ATTN_ABL & 1) ? s[4 * g + j] * 0.001f + 1.0f : __builtin_amdgcn_exp2f(s[4 * g + j]);
                    psum += f32x2{s[4 * g], s[4 * g + 1]};
                    psum += f32x2{s[4 * g + 2], s[4 * g + 3]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[4 * g + j] = 0.f;
                }
            }
            lsum += psum[0] + psum[1];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (!TAIL || 2 * ks < ng) {
                    vec8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (elem)s[8 * ks + j];
                    if (VH_ATTN_ABL & 8) { o0[ks] += (float)pf[0]; o1[ks] += (float)vfr.f[ks][0][0] + (float)vfr.f[ks][1][0]; }
                    else {
                    o0 = T::mfma32(vfr.f[ks][0], pf, o0);
                    o1 = T::mfma32(vfr.f[ks][1], pf, o1);
                    }
                }
            }
        };

        // No software pipeline across tiles (it costs 16 registers this kernel does not have at 4 waves per SIMD): a
        // wave's K reads and score MFMAs sit under the exp/max/convert streams of the SIMD's other waves.  The V
        // fragments are requested before the score MFMAs so that they arrive during the softmax.
        // A wave whose 32 rows all lie beyond `tokens` (T = 577: one of 20) runs the clamped last row and stores nothing:
        // no wave-dependent control flow around the barriers, the DMA duty or the Q prefetch.
        {
            const VFrag vfr = load_v(0);
            f32x16 sc = qk(0, 0.f);
            softmax_pv(0, sc, vfr, std::true_type{}, std::false_type{});
        }
        VH_ATT_T(2);   // Q fetch + tile 0
        for (int kt = 1; kt + 1 < ntiles; ++kt) {
            if (QS) ring_wait_vm(kt == 1 ? w_t1 : !has_next ? 0 : w_mid);   // this wave's pieces of slot kt
            VH_ATT_T(4);   // middle tiles: compute (and the wait, QS)
            if (!(VH_ATTN_ABL & 4)) ring_barrier();       // tile kt-1 is finished everywhere
            VH_ATT_T(3);   // middle tiles: barrier
            if (QS && kt == 1 && has_next && wave < 4) issue_q(nbase);
            if (!(VH_ATTN_ABL & 2) && refill && wave < 4) issue_tile(nbase, kt - 1);
            const VFrag vfr = load_v(kt);
            f32x16 sc = qk(kt, negm);
            softmax_pv(kt, sc, vfr, std::false_type{}, std::false_type{});
        }
        VH_ATT_T(4);
        if (QS) ring_wait_vm(!has_next ? 0 : w_last);     // this wave's pieces of slot nt-1
        else ring_wait_dma();                             // (youngest DMA in flight: one tile old)
        VH_ATT_T(5);   // last-tile wait
        if (!(VH_ATTN_ABL & 4)) ring_barrier();           // ... visible to every wave; tile nt-2 is finished everywhere
        VH_ATT_T(6);   // last-tile barrier
        if (!(VH_ATTN_ABL & 2) && refill && wave < 4) issue_tile(nbase, ntiles - 2);
        {
            const VFrag vfr = load_v(ntiles - 1);
            f32x16 sc = qk(ntiles - 1, negm);
            if (!QS && has_next) load_q(next, qf);        // the last score MFMAs are issued: qf is free
            softmax_pv(ntiles - 1, sc, vfr, std::false_type{}, std::true_type{});
            VH_ATT_T(8);   // last tile

            // ---- normalise and store: lane holds O[q][32*db + 8*rg + 4*hl + 0..3] ------------------
            const float ltot = cross_half_sum(lsum);
            const float inv = __builtin_amdgcn_rcpf(ltot);   // 1 ulp; the result is rounded to 16 (8) bits right after
            const int q = q0 + l31;
            if constexpr (sizeof(typename TO::elem) == 2) {
                // 16-bit output: the two halves of the wave hold the two quads of every 8-column group of a row, so the
                // natural store is 8 x 8 bytes per lane.  One v_permlane32_swap per packed dword on a PAIR of groups
                // (rg, rg + 1) leaves lanes 0-31 with the whole group rg and lanes 32-63 with the whole group rg + 1
                // (cdna_hip_programming.md T21): 4 x 16 bytes per lane, half the store instructions for the same bytes --
                // the phase is store-issue-bound (15 % of a wave's time in r02_c_attn_anatomy.txt).  The swaps run on every
                // lane (no divergence around a cross-lane instruction); only the stores are predicated.
                const int64_t mrow = (int64_t)b * tokens + (q < tokens ? q : tokens - 1);
                // row-major: row mrow, column h * 64 + 8 * hl (+ 32 db + 8 rg); tiled: chunk h * 8 + hl (+ 4 db + rg) of row block mrow >> 4
                typename TO::elem* const op = OTILE ? out + ((mrow >> 4) * (D >> 3) + h * 8 + hl) * 128 + (mrow & 15) * 8
                                                    : out + mrow * D + h * 64 + 8 * hl;
                constexpr int CH = OTILE ? 128 : 8;   // elements from one 8-column chunk to the next
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f32x16& o = db ? o1 : o0;
#pragma unroll
                    for (int rg = 0; rg < 4; rg += 2) {
                        const u32x2 ga = __builtin_bit_cast(u32x2, pack4<TO>(o[4 * rg] * inv, o[4 * rg + 1] * inv, o[4 * rg + 2] * inv, o[4 * rg + 3] * inv));
                        const u32x2 gb = __builtin_bit_cast(u32x2, pack4<TO>(o[4 * rg + 4] * inv, o[4 * rg + 5] * inv, o[4 * rg + 6] * inv, o[4 * rg + 7] * inv));
                        const auto sx = __builtin_amdgcn_permlane32_swap(ga[0], gb[0], false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(ga[1], gb[1], false, false);
                        if (q < tokens) *(u32x4*)(op + (4 * db + rg) * CH) = u32x4{sx[0], sy[0], sx[1], sy[1]};
                    }
                }
            } else {
                // e4m3 output (fp8 path): a lane's quad of a group is ONE dword; the same half exchange on a pair of groups
                // leaves lanes 0-31 with the 8 bytes of group rg and lanes 32-63 with those of group rg + 1: 4 x 8 bytes per
                // lane instead of 8 x 4 (round 4; NST follows)
                // row-major: row mrow, byte h * 64 + 8 * hl (+ 32 db + 8 rg); tiled (the e4m3 operand's layout, 16-byte chunks): the pair of
                // groups (rg, rg + 1) IS chunk h * 4 + 2 db + rg / 2 of the row, its halves in lanes l and l + 32 -- the same 8-byte stores
                const int64_t mrow = (int64_t)b * tokens + (q < tokens ? q : tokens - 1);
                typename TO::elem* const op = OTILE ? out + ((mrow >> 4) * (D >> 4) + h * 4) * 256 + (mrow & 15) * 16 + 8 * hl
                                                    : out + mrow * D + h * 64 + 8 * hl;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const f32x16& o = db ? o1 : o0;
#pragma unroll
                    for (int rg = 0; rg < 4; rg += 2) {
                        const uint32_t ga = pack4<TO>(o[4 * rg] * inv, o[4 * rg + 1] * inv, o[4 * rg + 2] * inv, o[4 * rg + 3] * inv);
                        const uint32_t gb = pack4<TO>(o[4 * rg + 4] * inv, o[4 * rg + 5] * inv, o[4 * rg + 6] * inv, o[4 * rg + 7] * inv);
                        const auto sx = __builtin_amdgcn_permlane32_swap(ga, gb, false, false);
                        if (q < tokens) *(u32x2*)(op + (OTILE ? (2 * db + (rg >> 1)) * 256 : 32 * db + 8 * rg)) = u32x2{sx[0], sx[1]};
                    }
                }
            }
        }
        VH_ATT_T(9);   // normalise + stores
        if (!has_next) break;
        stored = q0 < tokens;
        item = next;
        cbase = nbase;
        first = false;
    }
#ifdef VH_DIAG_STAMPS
    VH_ATT_T(7);
    if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 10; ++i) g_attn_diag[((size_t)blockIdx.x * 16 + wave) * 16 + i] = ph_[i];
        unsigned int hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned int xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_attn_diag[((size_t)blockIdx.x * 16 + wave) * 16 + 10] = rt0_;
        g_attn_diag[((size_t)blockIdx.x * 16 + wave) * 16 + 11] = attn_rt();
        g_attn_diag[((size_t)blockIdx.x * 16 + wave) * 16 + 12] = ((unsigned long long)xcc << 32) | hw;
        for (int i = 0; i < 3; ++i) g_attn_diag[((size_t)blockIdx.x * 16 + wave) * 16 + 13 + i] = dbg_[i];
    }
#endif
}

#ifdef VH_DIAG_STAMPS
extern "C" int vh_diag_attn_read(unsigned long long* host, int n_words) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_diag), (size_t)n_words * 8);
}
#endif

size_t attention_lds_bytes(int tokens) { return (size_t)((tokens + 31) / 32) * 8192; }

// which form launch_attn_t picks for (tokens, batch, heads): 2 = staged-Q ring, 1 = ring, 0 = one-shot / persistent double-buffered
static int attn_form(int batch, int tokens, int heads) {
    const int ntiles = (tokens + 31) / 32, nqb = ntiles;
    static const int max_waves = [] { const int e = env_int("VH_ATTN_WAVES", 16); return e < 1 || e > 16 ? 16 : e; }();
    const int slabs = (nqb + max_waves - 1) / max_waves, nw = (nqb + slabs - 1) / slabs;
    static const int want_ring = env_int("VH_ATTN_RING", 1);
    const int G = (tokens + 7) / 8, G2 = (G + 1) & ~1;
    const size_t qs_lds = (size_t)(G + G2 + 4 * nw) * 1024 + 16;
    if (attention_lds_bytes(tokens) > 160 * 1024) return -1;
    if (want_ring >= 1 && want_ring != 2 && slabs == 1 && nw >= 4 && 2 * qs_lds <= 160 * 1024) return 2;
    if (want_ring && ntiles >= 3 && nw >= 4) return 1;
    (void)batch; (void)heads;
    return 0;
}
bool attention_tiled_applies(int batch, int tokens, int heads) { return attn_form(batch, tokens, heads) >= 1 && heads % 2 == 0; }

template <typename T, typename TO = T, bool OTILE = false, bool IHM = false>
static hipError_t launch_attn_t(const void* qkv, int batch, int tokens, int heads, void* out, unsigned int* ticket, hipStream_t s, bool tk_zeroed,
                                int64_t hm_rows = 0) {
    const int ntiles = (tokens + 31) / 32;
    const int nqb = ntiles;
    // waves per workgroup (each owns 32 queries): VH_ATTN_WAVES, default 16 (T = 577: 10 waves share one K/V image instead of 7)
    static const int max_waves = [] { const int e = env_int("VH_ATTN_WAVES", 16); return e < 1 || e > 16 ? 16 : e; }();
    const int slabs = (nqb + max_waves - 1) / max_waves;
    const int nw = (nqb + slabs - 1) / slabs;
    const size_t one = attention_lds_bytes(tokens);
    if (one > 160 * 1024) return hipErrorInvalidValue;
    const int nitems = batch * heads * slabs;
    const int num_cu = device_num_cu();
    if (!num_cu) return hipErrorUnknown;
    // persistent double-buffered form when two K/V images fit and there is more than one item per CU
    // VH_ATTN_PERSIST=1 selects it; measured slower at T=197 (7 waves/CU leave the softmax latency-bound), so
    // the default is the one-shot form with two workgroups per CU.
    static const int want_persist = env_int("VH_ATTN_PERSIST", 0);
    const bool persist = want_persist && 2 * one <= 160 * 1024 && nitems > num_cu;
    // ring forms (default): need the four DMA waves and at least three key tiles; VH_ATTN_RING=0 selects the one-shot
    // form, 2 the ring without Q staging where the staged form would be used
    static const int want_ring = env_int("VH_ATTN_RING", 1);
    // staged-Q ring (counted waits): one slab of at least four waves whose trimmed images + staging fit twice in a CU
    const int G = (tokens + 7) / 8, G2 = (G + 1) & ~1;
    const size_t qs_lds = (size_t)(G + G2 + 4 * nw) * 1024 + 16;   // + the ticket word
    if (want_ring >= 1 && want_ring != 2 && slabs == 1 && nw >= 4 && 2 * qs_lds <= 160 * 1024) {
        auto k = attention_ring_kernel<T, TO, true, OTILE, IHM>;
        static LdsDone lds_done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, qs_lds, lds_done); e != hipSuccess) return e;
        const int grid = nitems < 2 * num_cu ? nitems : 2 * num_cu;
        static const int want_dyn = env_int("VH_ATTN_DYN", 1);   // VH_ATTN_DYN=0: equal static shares
        unsigned int* tk = want_dyn && nw >= 5 && nitems > 2 * grid ? ticket : nullptr;
        if (tk && !tk_zeroed) { if (hipError_t e = hipMemsetAsync(tk, 0, sizeof(unsigned int), s); e != hipSuccess) return e; }
        const unsigned int heads_rcp = (unsigned int)(((1ull << 32) + (unsigned)heads - 1) / (unsigned)heads);
        hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), qs_lds, s, (const typename T::elem*)qkv, (typename TO::elem*)out,
                           tokens, heads, slabs, ntiles, nitems, tk, heads_rcp, hm_rows);
        return hipGetLastError();
    }
    if (want_ring && ntiles >= 3 && nw >= 4) {
        auto k = attention_ring_kernel<T, TO, false, OTILE, IHM>;
        static LdsDone lds_done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, one, lds_done); e != hipSuccess) return e;
        const int per_cu = (int)(160 * 1024 / one);       // co-resident workgroups per CU by LDS
        int grid = num_cu * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
        if (grid > batch * heads) grid = batch * heads;   // a workgroup walks heads; the slabs of a head share its K/V image
        hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), one, s, (const typename T::elem*)qkv, (typename TO::elem*)out,
                           tokens, heads, slabs, ntiles, nitems, (unsigned int*)nullptr, 0u, hm_rows);
        return hipGetLastError();
    }
    if constexpr (OTILE) return hipErrorInvalidValue;   // the tiled output exists in the ring forms only (attention_tiled_applies)
    else
    if constexpr (!std::is_same<T, TO>::value) {
        auto k = attention_kernel<T, false, TO>;
        static LdsDone lds_done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, one, lds_done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(nitems), dim3(nw * 64), one, s, (const typename T::elem*)qkv, (typename TO::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
        return hipGetLastError();
    } else
    if (persist) {
        const size_t lds = 2 * one;
        auto k = attention_kernel<T, true>;
        static LdsDone lds_done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, lds_done); e != hipSuccess) return e;
        const int per_cu = (int)(160 * 1024 / lds);   // co-resident workgroups per CU by LDS
        int grid = num_cu * (per_cu < 1 ? 1 : per_cu);
        if (grid > nitems) grid = nitems;
        hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, s, (const typename T::elem*)qkv, (typename T::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
    } else {
        auto k = attention_kernel<T, false>;
        static LdsDone lds_done;
        if (hipError_t e = ensure_dynamic_lds((const void*)k, one, lds_done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(nitems), dim3(nw * 64), one, s, (const typename T::elem*)qkv, (typename T::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
    }
    return hipGetLastError();
}

// ---- attention of the class token only (VH_FLAG_CLS_TAIL: the last layer computes what the logits need) ----------------
// One wave per (image, head): the query is row 0 of the image (64 values, already scaled to the log2 domain like every q),
// 8 lanes share a key row (16 bytes each: a wave instruction reads 8 whole rows), fp32 dot products, exp2-domain softmax over
// the wave, then the same 8-lanes-per-row walk over the value rows with 8 output dimensions per lane.  Reads K and V once
// (2/3 of q|k|v), writes [batch][dim]; no MFMA -- 6 144 items x 25 K MACs at ViT-B b512.
template <typename T>
__global__ void __launch_bounds__(256)
attention_cls_kernel(const typename T::elem* __restrict__ qkv, typename T::elem* __restrict__ out, int batch, int tokens, int heads) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    constexpr int MAXT = 1024;
    __shared__ float sc[4][MAXT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= batch * heads) return;
    const int b = item / heads, h = item - b * heads;
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const elem* const base = qkv + (int64_t)b * tokens * ld + h * 64;
    // 8 lanes per row: lane = 8 * g + c reads the 16-byte chunk c of row 8 * step + g, so one wave instruction covers 8 whole
    // 128-byte rows
    const int g = lane >> 3, c = lane & 7;
    float q[8];
    {
        const vec8 v = *(const vec8*)(base + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = (float)v[j];
    }
    float mx = -INFINITY;
    constexpr int U = 4;   // row groups per step: U independent 16-byte loads in flight per lane (the walk is latency-bound)
    for (int t0 = 0; t0 < tokens; t0 += 8 * U) {
        vec8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 8 * u + g;
            v[u] = *(const vec8*)(base + (t < tokens ? t : tokens - 1) * ld + D + c * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 8 * u + g;
            float sv = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sv = fmaf(q[j], (float)v[u][j], sv);
            sv += __shfl_xor(sv, 1); sv += __shfl_xor(sv, 2); sv += __shfl_xor(sv, 4);   // the row's 8 lanes
            if (t < tokens) {
                if (c == 0) sc[wv][t] = sv;
                mx = fmaxf(mx, sv);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __builtin_amdgcn_s_waitcnt(0xC07F);   // the wave's LDS writes have landed (its slice is private: no barrier)
    float sum = 0.f;
    for (int t = lane; t < tokens; t += 64) {
        const float p = __builtin_amdgcn_exp2f(sc[wv][t] - mx);
        sc[wv][t] = p;
        sum += p;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // P V: the lane accumulates its 8 output dimensions (chunk c) over the rows g, g + 8, ...; the 8 row groups are added at the end
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t0 = 0; t0 < tokens; t0 += 8 * U) {
        vec8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 8 * u + g;
            v[u] = *(const vec8*)(base + (t < tokens ? t : tokens - 1) * ld + 2 * D + c * 8);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + 8 * u + g;
            const float p = t < tokens ? sc[wv][t] : 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(p, (float)v[u][j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[j] += __shfl_xor(acc[j], 8); acc[j] += __shfl_xor(acc[j], 16); acc[j] += __shfl_xor(acc[j], 32);
    }
    if (g == 0) {
        const float r = 1.0f / sum;
        vec8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (elem)(acc[j] * r);
        *(vec8*)(out + (int64_t)b * D + h * 64 + c * 8) = o;
    }
}
hipError_t launch_attention_cls(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype, hipStream_t s) {
    if (batch <= 0 || tokens <= 0 || tokens > 1024 || heads <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((batch * heads + 3) / 4));
    if (dtype == VH_DTYPE_BF16)
        hipLaunchKernelGGL(attention_cls_kernel<BF16>, grid, dim3(256), 0, s, (const BF16::elem*)qkv16, (BF16::elem*)out16, batch, tokens, heads);
    else if (dtype == VH_DTYPE_FP16)
        hipLaunchKernelGGL(attention_cls_kernel<FP16>, grid, dim3(256), 0, s, (const FP16::elem*)qkv16, (FP16::elem*)out16, batch, tokens, heads);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_attention(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype,
                            unsigned int* ticket, hipStream_t s, bool ticket_zeroed, bool out_tiled, int64_t in_hm_rows) {
    if (batch <= 0 || tokens <= 0 || heads <= 0) return hipErrorInvalidValue;
    if (in_hm_rows) {   // head-major q|k|v [3][heads][in_hm_rows][64] (the persistent projection's layout): ring forms with the tiled output only
        if (!out_tiled || in_hm_rows < (int64_t)batch * tokens || in_hm_rows * heads * 64 * 4 >= (1ll << 32)) return hipErrorInvalidValue;
        if (!attention_tiled_applies(batch, tokens, heads)) return hipErrorInvalidValue;
        if (dtype == VH_DTYPE_FP8) return launch_attn_t<BF16, E4M3, true, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed, in_hm_rows);
        return dtype == VH_DTYPE_BF16 ? launch_attn_t<BF16, BF16, true, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed, in_hm_rows)
                                      : launch_attn_t<FP16, FP16, true, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed, in_hm_rows);
    }
    if (out_tiled) {
        if (!attention_tiled_applies(batch, tokens, heads)) return hipErrorInvalidValue;
        if (dtype == VH_DTYPE_FP8) return launch_attn_t<BF16, E4M3, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed);   // (tiled e4m3: 16-byte chunks)
        return dtype == VH_DTYPE_BF16 ? launch_attn_t<BF16, BF16, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed)
                                      : launch_attn_t<FP16, FP16, true>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed);
    }
    if (dtype == VH_DTYPE_FP8) return launch_attn_t<BF16, E4M3>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed);  // bf16 in, e4m3 out
    return dtype == VH_DTYPE_BF16 ? launch_attn_t<BF16>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed)
                                  : launch_attn_t<FP16>(qkv16, batch, tokens, heads, out16, ticket, s, ticket_zeroed);
}

}  // namespace vh
