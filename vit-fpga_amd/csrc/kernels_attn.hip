// kernels_attn.hip — fused multi-head attention of the ViT hot path (gfx950).
//
//   out[b, t, h*64:(h+1)*64] = softmax( q_h k_h^T ) v_h        (q pre-scaled by 64^-1/2)
//
// reads the fused projection output qkv [batch*T, 3*H*64] (16-bit), writes [batch*T, H*64].
// Head dimension is fixed at 64 (ViT-Ti/S/B/L/H all use 64).
//
// Design (CDNA4):
//   * one workgroup per (image, head, query-slab); NW waves, each wave owns 32 query rows.
//   * the whole K and V of the head are staged ONCE into LDS (T<=~600 keys: 2*T*128 B),
//     K XOR-swizzled for conflict-free ds_read_b128 row reads, V swizzled for conflict-free
//     ds_read_b64_tr_b16 transposed reads.
//   * S^T = K Q^T with v_mfma_f32_32x32x16 (keys on the accumulator rows, the query on the
//     lane): a query's scores live in ONE lane pair (l, l^32), so the softmax row max / sum
//     are 15 in-register ops + one cross-half shuffle — a wavefront reduction, no LDS.
//   * the S^T accumulator, converted to 16-bit in registers, IS the B operand of
//     O^T = V^T P^T (same lane, k-order of the accumulator rows); V^T fragments come from the
//     hardware transposing LDS read.  Scores/probabilities never touch LDS or HBM.
//   * online softmax over 32-key tiles in the exp2 domain, fp32 statistics, keys >= T masked.
#include "vh_kernels.h"

namespace vh {

constexpr float kLog2e = 1.4426950408889634f;

template <typename T>
__global__ void __launch_bounds__(512)
attention_kernel(const typename T::elem* __restrict__ qkv, typename T::elem* __restrict__ out,
                 int tokens, int heads, int slabs, int ntiles) {
    using elem = typename T::elem;
    using vec8 = typename T::vec8;
    using vec4 = typename T::vec4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + ntiles * 4096;

    const int bh = blockIdx.x / slabs, slab = blockIdx.x - bh * slabs;
    const int b = bh / heads, h = bh - b * heads;
    const int D = heads * 64;
    const int64_t ld = 3 * (int64_t)D;
    const elem* base = qkv + (int64_t)b * tokens * ld + h * 64;

    // ---- stage K and V of this head into LDS (rows >= tokens replicate the last row) ------
    const int nchunks = ntiles * 32 * 8;
    for (int idx = threadIdx.x; idx < nchunks; idx += blockDim.x) {
        const int row = idx >> 3, c = idx & 7;
        const int rsrc = row < tokens ? row : tokens - 1;
        const elem* src = base + rsrc * ld + c * 8;
        const u32x4 kv = *(const u32x4*)(src + D);
        const u32x4 vv = *(const u32x4*)(src + 2 * D);
        *(u32x4*)(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kv;
        *(u32x4*)(Vs + row * 128 + ((c ^ (((row >> 1) & 1) << 2)) << 4)) = vv;
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int l31 = lane & 31, hl = lane >> 5;
    const int q0 = (slab * nw + wave) * 32;

    // ---- Q fragments: B operand of S^T = K Q^T, lane holds Q[q0 + l31][16*ks + 8*hl ..+7] ----
    vec8 qf[4];
    {
        int qrow = q0 + l31;
        qrow = qrow < tokens ? qrow : tokens - 1;
        const elem* qp = base + qrow * ld + 8 * hl;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const vec8*)(qp + 16 * ks);
    }
    __syncthreads();
    if (q0 >= tokens) return;  // wave-uniform; after the only barrier

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m2 = -INFINITY;  // running row max, log2 domain
    float lsum = 0.f;      // this lane's half of the row sum

    // per-lane LDS offsets
    const int koff = l31 * 128;                 // + ((c ^ ((l31>>1)&7)) << 4)
    const int kswz = (l31 >> 1) & 7;
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    // transposed V read: lane supplies &V[kbase + tq][dcol0 + 4*tp], dcol0 = 32*db + 16*(g&1)
    const int vrow0 = 4 * (g >> 1) + tq;        // + 32*kt + 16*s (+8)
    const int vcolb = (16 * (g & 1) + 4 * tp) * 2;  // byte offset inside the row, + 64*db

    for (int kt = 0; kt < ntiles; ++kt) {
        // ---- S^T tile: 32 keys x 32 queries -------------------------------------------------
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const char* kp = Ks + kt * 4096 + koff;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const vec8 kf = *(const vec8*)(kp + (((2 * ks + hl) ^ kswz) << 4));
            s = T::mfma32(kf, qf[ks], s);
        }
        if (kt == ntiles - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                if (key >= tokens) s[r] = -INFINITY;
            }
        }
        // ---- online softmax (row = query = lane pair) ----------------------------------------
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mnew = fmaxf(m2, mx * kLog2e);
        const float alpha = __builtin_amdgcn_exp2f(m2 - mnew);
        m2 = mnew;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = __builtin_amdgcn_exp2f(s[r] * kLog2e - mnew);
            psum += s[r];
        }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        // ---- O^T += V^T P^T ----------------------------------------------------------------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            vec8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (elem)s[8 * ks + j];
            const int r_lo = kt * 32 + 16 * ks + vrow0, r_hi = r_lo + 8;
            const char* plo = Vs + r_lo * 128;
            const char* phi = Vs + r_hi * 128;
            const int xlo = ((r_lo >> 1) & 1) << 6, xhi = ((r_hi >> 1) & 1) << 6;
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int cb = vcolb + 64 * db;
                const vec4 a = T::tr_read(plo + (cb ^ xlo));
                const vec4 c = T::tr_read(phi + (cb ^ xhi));
                vec8 vf;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vf[j] = a[j]; vf[4 + j] = c[j]; }
                if (db == 0) o0 = T::mfma32(vf, pf, o0); else o1 = T::mfma32(vf, pf, o1);
            }
        }
    }

    // ---- normalise and store: lane holds O[q][32*db + 8*rg + 4*hl + 0..3] ------------------
    const float ltot = lsum + __shfl_xor(lsum, 32);
    const float inv = 1.0f / ltot;
    const int q = q0 + l31;
    if (q < tokens) {
        elem* op = out + ((int64_t)b * tokens + q) * D + h * 64 + 4 * hl;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            *(vec4*)(op + 8 * rg) = pack4<T>(o0[4 * rg] * inv, o0[4 * rg + 1] * inv, o0[4 * rg + 2] * inv, o0[4 * rg + 3] * inv);
            *(vec4*)(op + 32 + 8 * rg) = pack4<T>(o1[4 * rg] * inv, o1[4 * rg + 1] * inv, o1[4 * rg + 2] * inv, o1[4 * rg + 3] * inv);
        }
    }
}

size_t attention_lds_bytes(int tokens) { return (size_t)((tokens + 31) / 32) * 8192; }

template <typename T>
static hipError_t launch_attn_t(const void* qkv, int batch, int tokens, int heads, void* out, hipStream_t s) {
    const int ntiles = (tokens + 31) / 32;
    const int nqb = ntiles;
    const int slabs = (nqb + 7) / 8;
    const int nw = (nqb + slabs - 1) / slabs;
    const size_t lds = attention_lds_bytes(tokens);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto k = attention_kernel<T>;
    static size_t lds_max = 0;
    if (lds > lds_max) {
        hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_max = lds;
    }
    hipLaunchKernelGGL(k, dim3(batch * heads * slabs), dim3(nw * 64), lds, s,
                       (const typename T::elem*)qkv, (typename T::elem*)out, tokens, heads, slabs, ntiles);
    return hipGetLastError();
}

hipError_t launch_attention(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype,
                            hipStream_t s) {
    if (batch <= 0 || tokens <= 0 || heads <= 0) return hipErrorInvalidValue;
    return dtype == VH_DTYPE_BF16 ? launch_attn_t<BF16>(qkv16, batch, tokens, heads, out16, s)
                                  : launch_attn_t<FP16>(qkv16, batch, tokens, heads, out16, s);
}

}  // namespace vh
