// vh_common.h — shared device/host helpers of libvithip (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vithip.h"

namespace vh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

// 16-bit MFMA operand types.  All kernels are templated on one of these two.
struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    static constexpr int id = VH_DTYPE_BF16;
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ vec4 tr_read(const void* lds_addr) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
            (vec4 __attribute__((address_space(3)))*)(uintptr_t)lds_addr);
    }
};
struct FP16 {
    using elem = _Float16;
    using vec8 = f16x8;
    using vec4 = f16x4;
    static constexpr int id = VH_DTYPE_FP16;
    static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ vec4 tr_read(const void* lds_addr) {
        typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
        h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
            (h4 __attribute__((address_space(3)))*)(uintptr_t)lds_addr);
        return __builtin_bit_cast(vec4, r);
    }
};

// fp32 "output type" of the kernels templated on their result element (final LayerNorm of the CLS rows -> fp32 head)
struct F32OUT {
    using elem = float;
    using vec4 = f32x4;
};

// four fp32 -> one 8-byte vector of the storage type.  Converted as two PAIRS: element-wise casts made hipcc emit one
// single conversion per value plus v_perm/v_alignbit to assemble the halves (3 instructions per pair instead of 1).
template <typename T>
__device__ __forceinline__ typename T::vec4 pack4(float a, float b, float c, float d) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef typename T::elem e2_ __attribute__((ext_vector_type(2)));
    const e2_ lo = __builtin_convertvector(f32x2_{a, b}, e2_), hi = __builtin_convertvector(f32x2_{c, d}, e2_);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// 8-bit OUTPUT type of the fp8 path (VH_DTYPE_FP8): OCP e4m3fn, no scale, saturating at +-448.
// v_cvt_pk_fp8_f32 rounds to nearest even but turns |x| > 464 into NaN (tools/probe_fp8.hip), hence the clamp.
// Only the kernels that PRODUCE a GEMM operand are instantiated on it (LayerNorm, attention output, cast, the
// fc1 epilogue); the MFMA side of the format lives in kernels_gemm5.hip.
typedef __attribute__((ext_vector_type(8))) int i32x8;
struct E4M3 {
    using elem = uint8_t;
    using vec4 = uint32_t;
    static constexpr int id = VH_DTYPE_FP8;
};
__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f);
    b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f);
    d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}
template <>
__device__ __forceinline__ uint32_t pack4<E4M3>(float a, float b, float c, float d) {
    return pack4_e4m3(a, b, c, d);
}
template <>
__device__ __forceinline__ f32x4 pack4<F32OUT>(float a, float b, float c, float d) {
    return f32x4{a, b, c, d};
}

// ---- lo plane of the split residual (x = hi + lo): ONE e4m3 byte per element ----------------------------
// hi = T(x) is the next GEMM's operand; lo carries what that rounding dropped.  |x - hi| <= |x| * 2^-8 (bf16) / 2^-11
// (fp16), so lo is stored SCALED by 32 / 256: the byte then holds a value of at most |x| / 8 -- it saturates only beyond
// |x| = 3 584, well past the "massive activations" of trained ViTs, and underflows below 2^-10 / scale = 3e-5 / 4e-6
// absolute, far below any element's share of a row -- and e4m3's 4 significant bits put the pair at 12 (bf16) / 15 (fp16)
// significant bits of x: the rounding a residual update injects, 2^-13 / 2^-16 relative, is 16x below the 2^-9 / 2^-12 the
// operand rounding of every GEMM input injects anyway -- for 6 instead of 8 bytes per element and update.  Powers of two:
// scaling is exact.
template <typename T> struct Lo8;
template <> struct Lo8<BF16> { static constexpr float scale = 32.f, inv = 1.f / 32.f; };
template <> struct Lo8<FP16> { static constexpr float scale = 256.f, inv = 1.f / 256.f; };
template <typename T>
__device__ __forceinline__ uint32_t lo8_pack4(float a, float b, float c, float d) {   // the four residues x - hi
    return pack4_e4m3(a * Lo8<T>::scale, b * Lo8<T>::scale, c * Lo8<T>::scale, d * Lo8<T>::scale);
}
template <typename T>
__device__ __forceinline__ f32x4 lo8_unpack4(uint32_t w) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    return f32x4{a[0] * Lo8<T>::inv, a[1] * Lo8<T>::inv, b[0] * Lo8<T>::inv, b[1] * Lo8<T>::inv};
}
template <typename T>
__device__ __forceinline__ uint8_t lo8_pack1(float a) {
    a = __builtin_amdgcn_fmed3f(a * Lo8<T>::scale, -448.f, 448.f);
    return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, a, 0, false) & 0xFF);
}
template <typename T>
__device__ __forceinline__ float lo8_unpack1(uint8_t b) { return __builtin_amdgcn_cvt_f32_fp8((int)b, 0) * Lo8<T>::inv; }

// ---- inline-asm vector-memory instructions with scalar operands: ONE place for the hazard rule --------------------------
// hipcc's hazard recogniser does not look into an asm string.  A scalar operand ("s") of a VMEM instruction inside one may
// have been written by a VALU instruction immediately in front of the statement -- v_readlane / v_readfirstlane when the
// register allocator reloads a spilled SGPR pair -- and "VALU writes SGPR -> VMEM reads that SGPR" needs 5 wait states.
// Without them the instruction goes out with a stale register half (round 2: an atomic with a stale upper address half,
// memory aperture violation).  Every asm VMEM instruction of this library that takes an "s" operand is therefore issued
// through one of the helpers below, whose strings open with `s_nop 4` (it also covers the wait state between an M0 write
// and the LDS-DMA that reads M0).
//
// One LDS-DMA instruction: 64 lanes x 16 B from `base + lane_off` -> 1 KiB at LDS byte address `lds_addr` (lane-linear).
// Written as asm on purpose where the compiler must NOT know about the transfer: hipcc answers an LDS read that may alias
// a global_load_lds it knows of with s_waitcnt vmcnt(0), which drains every DMA in flight.  The caller owns the
// vmcnt / barrier bookkeeping that orders the ds_read behind the transfer.
__device__ __forceinline__ void asm_lds_dma16(const void* base, uint32_t lane_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr), "v"(lane_off), "s"(base) : "memory", "m0");
}
// The same with 4 bytes per lane: 64 lanes x 4 B from `base + lane_off` -> 256 B at `lds_addr`.
__device__ __forceinline__ void asm_lds_dma4(const void* base, uint32_t lane_off, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dword %1, %2"
                 :: "s"(lds_addr), "v"(lane_off), "s"(base) : "memory", "m0");
}
// Returning atomic add on the wave-uniform address `base`, ISSUE only: `ret` is written when the operation completes,
// i.e. the caller waits (vmcnt) before the first use of `ret` and keeps `ret`'s register untouched until then (the
// attention work queue draws its ticket this way, one lane active).
__device__ __forceinline__ void asm_atomic_add_ret_issue(unsigned int* base, uint32_t add, uint32_t& ret) {
    const uint32_t zero = 0;
    asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(ret) : "v"(zero), "v"(add), "s"(base) : "memory");
}

__device__ __forceinline__ float gelu_erf(float v) {
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// ---- synthetic-data generator (DESIGN.md "synthetic data") -----------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ __forceinline__ uint64_t rng_stream(uint64_t seed, uint32_t tensor_id) {
    return mix64(mix64(seed) ^ (uint64_t)tensor_id);
}
__host__ __device__ __forceinline__ float rng_uniform(uint64_t stream, uint64_t i) {
    const uint64_t w = mix64(stream ^ i);
    const int32_t u = (int32_t)(w >> 40) - (1 << 23);
    return (float)u * (1.0f / 8388608.0f);
}
// offset + IrwinHall4 * sigma; `scale` = (double)sigma / 37837.22725
__host__ __device__ __forceinline__ float rng_ih4(uint64_t stream, uint64_t i, double scale,
                                                  float offset) {
    const uint64_t w = mix64(stream ^ i);
    const int32_t s = (int32_t)(w & 0xFFFF) + (int32_t)((w >> 16) & 0xFFFF) +
                      (int32_t)((w >> 32) & 0xFFFF) + (int32_t)((w >> 48) & 0xFFFF) - 131070;
    const float v = (float)((double)s * scale);
    return offset + v;
}
constexpr double kIH4Std = 37837.22725;

// tensor ids of the canonical blob (same numbers as oracle/ and tests/, by specification)
enum : uint32_t { TID_PATCH_W = 1, TID_PATCH_B = 2, TID_CLS = 3, TID_POS = 4, TID_LAYER0 = 16,
                  TID_FINAL = 0x7000, TID_IMAGES = 0x100 };

}  // namespace vh
