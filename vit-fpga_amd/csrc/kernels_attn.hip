// kernels_attn.hip — fused multi-head attention of the ViT hot path (gfx950).
//
//   out[b, t, h*64:(h+1)*64] = softmax( q_h k_h^T ) v_h        (q pre-scaled by 64^-1/2)
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

constexpr float kLog2e = 1.4426950408889634f;

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
                const float mnew = fmaxf(m2, mx * kLog2e);
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
                    s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], kLog2e, -mnew));
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

size_t attention_lds_bytes(int tokens) { return (size_t)((tokens + 31) / 32) * 8192; }

template <typename T, typename TO = T>
static hipError_t launch_attn_t(const void* qkv, int batch, int tokens, int heads, void* out, hipStream_t s) {
    const int ntiles = (tokens + 31) / 32;
    const int nqb = ntiles;
    static int max_waves = 0;   // waves per workgroup (each owns 32 queries): VH_ATTN_WAVES, default 16 (T = 577: 10 waves share one K/V image instead of 7)
    if (!max_waves) { const char* e = getenv("VH_ATTN_WAVES"); max_waves = e ? atoi(e) : 16; if (max_waves < 1 || max_waves > 16) max_waves = 16; }
    const int slabs = (nqb + max_waves - 1) / max_waves;
    const int nw = (nqb + slabs - 1) / slabs;
    const size_t one = attention_lds_bytes(tokens);
    if (one > 160 * 1024) return hipErrorInvalidValue;
    const int nitems = batch * heads * slabs;
    static int num_cu = 0;
    if (!num_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return hipErrorUnknown;
        num_cu = prop.multiProcessorCount;
    }
    // persistent double-buffered form when two K/V images fit and there is more than one item per CU
    // VH_ATTN_PERSIST=1 selects it; measured slower at T=197 (7 waves/CU leave the softmax latency-bound), so
    // the default is the one-shot form with two workgroups per CU.
    static int want_persist = -1;
    if (want_persist < 0) { const char* e = getenv("VH_ATTN_PERSIST"); want_persist = e ? atoi(e) : 0; }
    const bool persist = want_persist && 2 * one <= 160 * 1024 && nitems > num_cu;
    if constexpr (!std::is_same<T, TO>::value) {
        auto k = attention_kernel<T, false, TO>;
        static int lds_done[kMaxDevices] = {0};
        if (hipError_t e = ensure_dynamic_lds((const void*)k, one, lds_done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(nitems), dim3(nw * 64), one, s, (const typename T::elem*)qkv, (typename TO::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
        return hipGetLastError();
    } else
    if (persist) {
        const size_t lds = 2 * one;
        auto k = attention_kernel<T, true>;
        static int lds_done[kMaxDevices] = {0};
        if (hipError_t e = ensure_dynamic_lds((const void*)k, lds, lds_done); e != hipSuccess) return e;
        const int per_cu = (int)(160 * 1024 / lds);   // co-resident workgroups per CU by LDS
        int grid = num_cu * (per_cu < 1 ? 1 : per_cu);
        if (grid > nitems) grid = nitems;
        hipLaunchKernelGGL(k, dim3(grid), dim3(nw * 64), lds, s, (const typename T::elem*)qkv, (typename T::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
    } else {
        auto k = attention_kernel<T, false>;
        static int lds_done[kMaxDevices] = {0};
        if (hipError_t e = ensure_dynamic_lds((const void*)k, one, lds_done); e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(nitems), dim3(nw * 64), one, s, (const typename T::elem*)qkv, (typename T::elem*)out,
                           tokens, heads, slabs, ntiles, nitems);
    }
    return hipGetLastError();
}

hipError_t launch_attention(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype,
                            hipStream_t s) {
    if (batch <= 0 || tokens <= 0 || heads <= 0) return hipErrorInvalidValue;
    if (dtype == VH_DTYPE_FP8) return launch_attn_t<BF16, E4M3>(qkv16, batch, tokens, heads, out16, s);  // bf16 in, e4m3 out
    return dtype == VH_DTYPE_BF16 ? launch_attn_t<BF16>(qkv16, batch, tokens, heads, out16, s)
                                  : launch_attn_t<FP16>(qkv16, batch, tokens, heads, out16, s);
}

}  // namespace vh
