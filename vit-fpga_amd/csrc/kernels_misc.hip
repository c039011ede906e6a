// kernels_misc.hip — the HBM-bound pieces around the GEMMs (gfx950): LayerNorm, patch
// gather (im2col + cast), CLS-row initialisation, casts, weight re-layout, the synthetic-data
// generator and the fp32 dense layer of MLP mode.  All of them move 16 bytes per lane.
#include <type_traits>

#include "vh_kernels.h"

namespace vh {

// ---- LayerNorm: one wave per row, row kept in registers, fp32 statistics ------------------
// y[r,:] = (x[r,:] - mean) * rstd * gamma + beta  -> 16-bit.  dim % 4 == 0, dim <= 1024*?.
template <typename T, int CH>  // CH = float4 chunks per lane (dim <= 256*CH)
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, int64_t rows, int dim, int64_t row_stride,
                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                 typename T::elem* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = dim >> 2;
    const f32x4* xr = (const f32x4*)(x + row * row_stride);
    f32x4 v[CH];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nchunk ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)dim;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; var += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)dim + eps);
    typename T::elem* orow = out + row * dim;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const f32x4 gm = ((const f32x4*)gamma)[c], bt = ((const f32x4*)beta)[c];
            *(typename T::vec4*)(orow + 4 * c) =
                pack4<T>((v[i][0] - mean) * rstd * gm[0] + bt[0], (v[i][1] - mean) * rstd * gm[1] + bt[1],
                         (v[i][2] - mean) * rstd * gm[2] + bt[2], (v[i][3] - mean) * rstd * gm[3] + bt[3]);
        }
    }
}

template <typename T>
static hipError_t ln_t(const float* x, int64_t rows, int dim, int64_t stride, const float* g, const float* b,
                       float eps, void* out, hipStream_t s) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    auto o = (typename T::elem*)out;
    if (dim <= 256) hipLaunchKernelGGL((layernorm_kernel<T, 1>), grid, block, 0, s, x, rows, dim, stride, g, b, eps, o);
    else if (dim <= 512) hipLaunchKernelGGL((layernorm_kernel<T, 2>), grid, block, 0, s, x, rows, dim, stride, g, b, eps, o);
    else if (dim <= 768) hipLaunchKernelGGL((layernorm_kernel<T, 3>), grid, block, 0, s, x, rows, dim, stride, g, b, eps, o);
    else if (dim <= 1024) hipLaunchKernelGGL((layernorm_kernel<T, 4>), grid, block, 0, s, x, rows, dim, stride, g, b, eps, o);
    else if (dim <= 2048) hipLaunchKernelGGL((layernorm_kernel<T, 8>), grid, block, 0, s, x, rows, dim, stride, g, b, eps, o);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, int64_t rows, int dim, int64_t row_stride, const float* gamma,
                            const float* beta, float eps, void* out16, int dtype, hipStream_t s) {
    if (rows <= 0 || dim <= 0 || (dim & 3) || (row_stride & 3)) return hipErrorInvalidValue;
    if (dtype == VH_DTYPE_F32_INTERNAL) return ln_t<F32OUT>(x, rows, dim, row_stride, gamma, beta, eps, out16, s);
    if (dtype == VH_DTYPE_FP8) return ln_t<E4M3>(x, rows, dim, row_stride, gamma, beta, eps, out16, s);  // 1 byte / element
    return dtype == VH_DTYPE_BF16 ? ln_t<BF16>(x, rows, dim, row_stride, gamma, beta, eps, out16, s)
                                  : ln_t<FP16>(x, rows, dim, row_stride, gamma, beta, eps, out16, s);
}

// ---- im2col + cast: NHWC fp32 image -> patch matrix [batch*np, patch*patch*ch] 16-bit -------
// one float4 per thread; a patch row is `patch*ch` contiguous floats of the image, and lands as
// `patch*ch` contiguous 16-bit elements of the patch matrix, so both sides are coalesced.
template <typename T>
__global__ void __launch_bounds__(256)
im2col_kernel(const float* __restrict__ in, typename T::elem* __restrict__ out, int64_t n4, int image,
              int patch, int ch) {
    const int rowf4 = image * ch / 4;      // float4 per image row
    const int prf4 = patch * ch / 4;       // float4 per patch row
    const int g = image / patch;
    const int kp = patch * patch * ch;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n4; f += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rowidx = f / rowf4;  // b*image + y
        const int q = (int)(f - rowidx * rowf4);
        const int64_t b = rowidx / image;
        const int y = (int)(rowidx - b * image);
        const int px = q / prf4, r = q - px * prf4;
        const int py = y / patch, ky = y - py * patch;
        const f32x4 v = ((const f32x4*)in)[f];
        const int64_t orow = (b * g + py) * g + px;
        *(typename T::vec4*)(out + orow * kp + ky * patch * ch + r * 4) = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}

hipError_t launch_im2col(const float* in, int batch, int image, int patch, int channels, void* out16, int dtype,
                         hipStream_t s) {
    if ((patch * channels) % 4 || image % patch) return hipErrorInvalidValue;
    const int64_t n4 = (int64_t)batch * image * image * channels / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 16384 ? (n4 + 255) / 256 : 16384);
    if (dtype == VH_DTYPE_BF16)
        hipLaunchKernelGGL(im2col_kernel<BF16>, dim3(grid), dim3(256), 0, s, in, (BF16::elem*)out16, n4, image, patch, channels);
    else
        hipLaunchKernelGGL(im2col_kernel<FP16>, dim3(grid), dim3(256), 0, s, in, (FP16::elem*)out16, n4, image, patch, channels);
    return hipGetLastError();
}

// ---- x[b, 0, :] = cls + pos[0] ------------------------------------------------------------------
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                int batch, int tokens, int dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)batch * dim) return;
    const int b = (int)(i / dim), d = (int)(i - (int64_t)b * dim);
    x[(int64_t)b * tokens * dim + d] = cls[d] + pos[d];
}
hipError_t launch_cls_rows(float* x, const float* cls, const float* pos, int batch, int tokens, int dim, hipStream_t s) {
    const int64_t n = (int64_t)batch * dim;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, cls, pos, batch, tokens, dim);
    return hipGetLastError();
}

// the same rows for the SPLIT residual (PATCH_SPLIT path): planes hi = T(x), lo = T(x - hi) of x = cls + pos[0], and the
// row's (sum, sum of squares) per 64-column block -- what the patch GEMM's epilogue writes for every other row.  One wave
// per image; a lane owns column 64 * blk + lane of every block; fixed-order butterfly sums.
template <typename T>
__global__ void __launch_bounds__(256)
cls_rows_split_kernel(typename T::elem* __restrict__ hi, uint8_t* __restrict__ lo, float* __restrict__ partials, int64_t prow,
                      const float* __restrict__ cls, const float* __restrict__ pos, int batch, int tokens, int dim) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    const int64_t row = (int64_t)b * tokens;
    for (int blk = 0; blk * 64 < dim; ++blk) {
        const int d = blk * 64 + lane;
        float v = 0.f;
        if (d < dim) {
            v = cls[d] + pos[d];
            const typename T::elem h = (typename T::elem)v;
            hi[row * dim + d] = h;
            lo[row * dim + d] = lo8_pack1<T>(v - (float)h);
        }
        float s1 = v, s2 = v * v;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        if (lane == 0) *(float2*)(partials + 2 * ((int64_t)blk * prow + row)) = make_float2(s1, s2);
    }
}
hipError_t launch_cls_rows_split(void* hi, void* lo, float* partials, int64_t prow, const float* cls, const float* pos, int batch,
                                 int tokens, int dim, int dtype, hipStream_t s) {
    if (batch <= 0 || dim % 64) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((batch + 3) / 4));
    if (dtype == VH_DTYPE_BF16)
        hipLaunchKernelGGL(cls_rows_split_kernel<BF16>, grid, dim3(256), 0, s, (BF16::elem*)hi, (uint8_t*)lo, partials, prow, cls, pos, batch, tokens, dim);
    else
        hipLaunchKernelGGL(cls_rows_split_kernel<FP16>, grid, dim3(256), 0, s, (FP16::elem*)hi, (uint8_t*)lo, partials, prow, cls, pos, batch, tokens, dim);
    return hipGetLastError();
}

// ---- fp32 -> 16-bit cast -----------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) cast_kernel(const float* __restrict__ in, typename T::elem* __restrict__ out, int64_t n4) {
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < n4; f += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = ((const f32x4*)in)[f];
        ((typename T::vec4*)out)[f] = pack4<T>(v[0], v[1], v[2], v[3]);
    }
}
hipError_t launch_cast(const float* in, void* out16, int64_t n, int dtype, hipStream_t s) {
    if (n <= 0 || (n & 3)) return hipErrorInvalidValue;
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    if (dtype == VH_DTYPE_FP8) hipLaunchKernelGGL(cast_kernel<E4M3>, dim3(grid), dim3(256), 0, s, in, (E4M3::elem*)out16, n4);
    else if (dtype == VH_DTYPE_BF16) hipLaunchKernelGGL(cast_kernel<BF16>, dim3(grid), dim3(256), 0, s, in, (BF16::elem*)out16, n4);
    else hipLaunchKernelGGL(cast_kernel<FP16>, dim3(grid), dim3(256), 0, s, in, (FP16::elem*)out16, n4);
    return hipGetLastError();
}

// W fp32 [rows, cols] -> 16-bit in the TILED layout the MLP hidden activation has (gemm_epilogue.h OTILED, kernels_gemm5.hip AT):
// [row / 16][cols / 8 chunks][16 rows][8 values], natural column order.  One thread per destination chunk.
template <typename T>
__global__ void __launch_bounds__(256) cast_tiled_w_kernel(const float* __restrict__ w, typename T::elem* __restrict__ out, int rows, int cols) {
    const int64_t nchunk = (int64_t)rows * (cols >> 3);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        const int r16 = (int)(i & 15);
        const int64_t bc = i >> 4;                        // block * (cols / 8) + chunk
        const int chunk = (int)(bc % (cols >> 3)), block = (int)(bc / (cols >> 3));
        const float* src = w + (int64_t)(block * 16 + r16) * cols + chunk * 8;
        const f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 4);
        const typename T::vec4 lo = pack4<T>(a[0], a[1], a[2], a[3]), hi = pack4<T>(b[0], b[1], b[2], b[3]);
        *(typename T::vec8*)(out + i * 8) = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
}
hipError_t launch_cast_tiled_w(const float* w, int rows, int cols, void* w16, int dtype, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 7)) return hipErrorInvalidValue;
    const int64_t nchunk = (int64_t)rows * (cols >> 3);
    const unsigned grid = (unsigned)((nchunk + 255) / 256 < 8192 ? (nchunk + 255) / 256 : 8192);
    if (dtype == VH_DTYPE_BF16) hipLaunchKernelGGL(cast_tiled_w_kernel<BF16>, dim3(grid), dim3(256), 0, s, w, (BF16::elem*)w16, rows, cols);
    else if (dtype == VH_DTYPE_FP16) hipLaunchKernelGGL(cast_tiled_w_kernel<FP16>, dim3(grid), dim3(256), 0, s, w, (FP16::elem*)w16, rows, cols);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// e4m3 weights (one byte per element, rows of `cols` bytes) once more in the tiled layout of the e4m3 operand:
// [rows / 16][cols / 16 chunks][16 rows][16 B] -- a copy of 16-byte pieces, nothing is converted
__global__ void __launch_bounds__(256) tile_bytes_kernel(const u32x4* __restrict__ w, u32x4* __restrict__ out, int rows, int cols) {
    const int cpr = cols >> 4;   // chunks per row
    const int64_t nchunk = (int64_t)rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nchunk; i += (int64_t)gridDim.x * blockDim.x) {
        const int r16 = (int)(i & 15);
        const int64_t bc = i >> 4;                        // block * cpr + chunk
        const int chunk = (int)(bc % cpr), block = (int)(bc / cpr);
        out[i] = w[(int64_t)(block * 16 + r16) * cpr + chunk];
    }
}
hipError_t launch_tile_bytes(const void* w8, int rows, int cols, void* out, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || (rows & 15) || (cols & 15)) return hipErrorInvalidValue;
    const int64_t nchunk = (int64_t)rows * (cols >> 4);
    const unsigned grid = (unsigned)((nchunk + 255) / 256 < 8192 ? (nchunk + 255) / 256 : 8192);
    hipLaunchKernelGGL(tile_bytes_kernel, dim3(grid), dim3(256), 0, s, (const u32x4*)w8, (u32x4*)out, rows, cols);
    return hipGetLastError();
}

// ---- synthetic data ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ out, int64_t n, uint64_t stream, int kind,
                                                   double scale, float offset) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v;
        if (kind == 0) v = rng_uniform(stream, (uint64_t)i);
        else if (kind == 1) v = rng_ih4(stream, (uint64_t)i, scale, offset);
        else v = offset;
        out[i] = v;
    }
}
hipError_t launch_fill(float* out, int64_t n, uint64_t seed, uint32_t tensor_id, int kind, float sigma, float offset,
                       hipStream_t s) {
    if (n <= 0) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    hipLaunchKernelGGL(fill_kernel, dim3(grid), dim3(256), 0, s, out, n, rng_stream(seed, tensor_id), kind,
                       (double)sigma / kIH4Std, offset);
    return hipGetLastError();
}

// ---- fp8 weight quantisation: one wave per row -------------------------------------------------------------
// s0 = amax_r * (1/448) (1 if the row is all zero), q = rne_e4m3(w / s0) (IEEE division: the oracle's quantiser
// divides too, so both produce the same byte), scale[r] = s0 * post.
__global__ void __launch_bounds__(256)
quantize_rows_kernel(const float* __restrict__ w, int rows, int cols, float post, uint8_t* __restrict__ w8,
                     float* __restrict__ scale) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const f32x4* wr = (const f32x4*)(w + (int64_t)r * cols);
    const int n4 = cols >> 2;
    float amax = 0.f;
    for (int c = lane; c < n4; c += 64) {
        const f32x4 v = wr[c];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float s0 = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    uint32_t* orow = (uint32_t*)(w8 + (int64_t)r * cols);
    for (int c = lane; c < n4; c += 64) {
        const f32x4 v = wr[c];
        orow[c] = pack4_e4m3(__fdiv_rn(v[0], s0), __fdiv_rn(v[1], s0), __fdiv_rn(v[2], s0), __fdiv_rn(v[3], s0));
    }
    if (lane == 0) scale[r] = s0 * post;
}
hipError_t launch_quantize_rows(const float* w, int rows, int cols, float post, void* w8, float* scale, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || (cols & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, rows, cols, post,
                       (uint8_t*)w8, scale);
    return hipGetLastError();
}

// Weight-only e4m3 (VH_FLAG_W8_E4M3): out[r, :] = decode(quantise(w[r, :])) * s0 with the quantiser above -- the values
// a GEMM would see if the weight matrix were stored as e4m3 bytes + one scale per output channel and dequantised on its
// way to the matrix core.  The 16-bit weight preparation (cast / q|k|v packing / LayerNorm fold) then runs on `out`.
__global__ void __launch_bounds__(256)
fake_quant_rows_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const f32x4* wr = (const f32x4*)(w + (int64_t)r * cols);
    const int n4 = cols >> 2;
    float amax = 0.f;
    for (int c = lane; c < n4; c += 64) {
        const f32x4 v = wr[c];
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float s0 = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    f32x4* orow = (f32x4*)(out + (int64_t)r * cols);
    for (int c = lane; c < n4; c += 64) {
        const f32x4 v = wr[c];
        const int q = (int)pack4_e4m3(__fdiv_rn(v[0], s0), __fdiv_rn(v[1], s0), __fdiv_rn(v[2], s0), __fdiv_rn(v[3], s0));
        orow[c] = f32x4{__builtin_amdgcn_cvt_f32_fp8(q, 0) * s0, __builtin_amdgcn_cvt_f32_fp8(q, 1) * s0,
                        __builtin_amdgcn_cvt_f32_fp8(q, 2) * s0, __builtin_amdgcn_cvt_f32_fp8(q, 3) * s0};
    }
}
hipError_t launch_fake_quant_rows(const float* w, int rows, int cols, float* out, hipStream_t s) {
    if (rows <= 0 || cols <= 0 || (cols & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fake_quant_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, rows, cols, out);
    return hipGetLastError();
}

// ---- weight re-layout ---------------------------------------------------------------------------
// Wqkv[3D, D] = [q * q_scale ; k ; v] (16-bit), bqkv[3D] = [bq * q_scale ; bk ; bv] (fp32).
// q_scale = 64^-1/2 = 0.125 is a power of two, so folding it into the weights is exact.
template <typename T>
__global__ void __launch_bounds__(256)
pack_qkv_kernel(const float* __restrict__ qw, const float* __restrict__ qb, const float* __restrict__ kw,
                const float* __restrict__ kb, const float* __restrict__ vw, const float* __restrict__ vb, int dim,
                float q_scale, typename T::elem* __restrict__ w16, float* __restrict__ b32) {
    const int64_t dd = (int64_t)dim * dim;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * dd) {
        const int which = (int)(i / dd);
        const int64_t j = i - which * dd;
        const float v = which == 0 ? qw[j] * q_scale : (which == 1 ? kw[j] : vw[j]);
        w16[i] = (typename T::elem)v;
    }
    if (i < 3 * dim) {
        const int which = (int)(i / dim), j = (int)(i - (int64_t)which * dim);
        b32[i] = which == 0 ? qb[j] * q_scale : (which == 1 ? kb[j] : vb[j]);
    }
}
hipError_t launch_pack_qkv(const float* qw, const float* qb, const float* kw, const float* kb, const float* vw,
                           const float* vb, int dim, float q_scale, void* w16, float* b32, int dtype, hipStream_t s) {
    const int64_t n = 3 * (int64_t)dim * dim;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == VH_DTYPE_BF16)
        hipLaunchKernelGGL(pack_qkv_kernel<BF16>, grid, block, 0, s, qw, qb, kw, kb, vw, vb, dim, q_scale, (BF16::elem*)w16, b32);
    else
        hipLaunchKernelGGL(pack_qkv_kernel<FP16>, grid, block, 0, s, qw, qb, kw, kb, vw, vb, dim, q_scale, (FP16::elem*)w16, b32);
    return hipGetLastError();
}

// conv kernel [D][c][ky][kx] fp32 -> [D][(ky*P + kx)*C + c] 16-bit: the k-order of an NHWC patch
template <typename T>
__global__ void __launch_bounds__(256)
permute_patch_kernel(const float* __restrict__ w, int dim, int ch, int patch, typename T::elem* __restrict__ o) {
    const int kp = patch * patch * ch;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)dim * kp) return;
    const int d = (int)(i / kp), k = (int)(i - (int64_t)d * kp);
    const int c = k % ch, kx = (k / ch) % patch, ky = k / (ch * patch);
    o[i] = (typename T::elem)w[(((int64_t)d * ch + c) * patch + ky) * patch + kx];
}
hipError_t launch_permute_patch(const float* w, int dim, int channels, int patch, void* w16, int dtype, hipStream_t s) {
    const int64_t n = (int64_t)dim * patch * patch * channels;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == VH_DTYPE_BF16) hipLaunchKernelGGL(permute_patch_kernel<BF16>, grid, block, 0, s, w, dim, channels, patch, (BF16::elem*)w16);
    else hipLaunchKernelGGL(permute_patch_kernel<FP16>, grid, block, 0, s, w, dim, channels, patch, (FP16::elem*)w16);
    return hipGetLastError();
}

// ---- MLP mode: one dense layer, fp32 exact products, one wave per output neuron ----------------------
// y[v, j] = act(b[j] + sum_k W[j, k] x[v, k]) — the body of the reference's network_v1 layer loop
// (argument layout netFPGA.cpp:427-436; weights row-major [n_out, n_in], netFPGA.cpp:94-105).
__device__ __forceinline__ float act_apply(int act, float v) {
    switch (act) {
    case VH_ACT_RELU2: return fminf(fmaxf(v, 0.f), 1.f);
    case VH_ACT_RELU: return fmaxf(v, 0.f);
    case VH_ACT_HARDTANH: return fminf(fmaxf(v, -1.f), 1.f);
    case VH_ACT_GELU: return gelu_erf(v);
    default: return v;
    }
}
// `z` (optional): the pre-activations, kept for the backward pass of MLP-mode training
__global__ void __launch_bounds__(256)
dense_layer_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ x,
                   float* __restrict__ y, int n_in, int n_out, int n_vec, int act, float* __restrict__ z) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n_out) return;
    const float* wr = w + (int64_t)j * n_in;
    for (int v = 0; v < n_vec; ++v) {
        const float* xv = x + (int64_t)v * n_in;
        float s = 0.f;
        for (int k = lane; k < n_in; k += 64) s = fmaf(wr[k], xv[k], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float pre = s + b[j];
            if (z) z[(int64_t)v * n_out + j] = pre;
            y[(int64_t)v * n_out + j] = act_apply(act, pre);
        }
    }
}
hipError_t launch_dense_layer(const float* w, const float* b, const float* x, float* y, int n_in, int n_out, int n_vec,
                              int activation, hipStream_t s, float* z) {
    hipLaunchKernelGGL(dense_layer_kernel, dim3((unsigned)((n_out + 3) / 4)), dim3(256), 0, s, w, b, x, y, n_in, n_out,
                       n_vec, activation, z);
    return hipGetLastError();
}

// ---- MLP-mode training (init_gradient / launch_gradient, netFPGA.cpp:518-580: commented-out code in the reference; the
// definitions are this build's -- include/vithip.h, oracle/mlp_oracle.c).  Small fp32 kernels, every sum in a fixed order.
__device__ __forceinline__ float act_deriv(int act, float z) {
    switch (act) {
    case VH_ACT_RELU2: return (z > 0.f && z < 1.f) ? 1.f : 0.f;
    case VH_ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case VH_ACT_HARDTANH: return (z > -1.f && z < 1.f) ? 1.f : 0.f;
    case VH_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.39894228040143267794f * expf(-0.5f * z * z);
    default: return 1.f;
    }
}
// last layer: d = (a - t) * act'(z); err = sum |a - t| over all sets and outputs.  ONE workgroup: every thread adds its
// elements in index order, then a fixed tree -- the same bits on every run.
__global__ void __launch_bounds__(1024)
mlp_out_delta_kernel(const float* __restrict__ a, const float* __restrict__ z, const float* __restrict__ t, float* __restrict__ d,
                     int64_t n, int act, float* __restrict__ err) {
    __shared__ float part[1024];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float e = a[i] - t[i];
        s += fabsf(e);
        d[i] = e * act_deriv(act, z[i]);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *err = part[0];
}
// hidden layer: d_prev[j][i] = act'(z_prev[j][i]) * sum_o W[o][i] d[j][o]   (W of the layer ABOVE, before its update)
__global__ void __launch_bounds__(256)
mlp_back_delta_kernel(const float* __restrict__ w, const float* __restrict__ d, const float* __restrict__ z_prev,
                      float* __restrict__ d_prev, int n_in, int n_out, int act) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= n_in) return;
    const float* dj = d + (int64_t)j * n_out;
    float s = 0.f;
    for (int o = 0; o < n_out; ++o) s = fmaf(w[(int64_t)o * n_in + i], dj[o], s);
    d_prev[(int64_t)j * n_in + i] = s * act_deriv(act, z_prev[(int64_t)j * n_in + i]);
}
// W[o][k] -= scale * sum_j d[j][o] x[j][k];  b[o] -= scale * sum_j d[j][o]     (scale = multiplier / n_sets)
__global__ void __launch_bounds__(256)
mlp_update_kernel(float* __restrict__ w, float* __restrict__ b, const float* __restrict__ d, const float* __restrict__ x,
                  int n_in, int n_out, int n_sets, float scale) {
    const int k = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
    if (k < n_in) {
        float g = 0.f;
        for (int j = 0; j < n_sets; ++j) g = fmaf(d[(int64_t)j * n_out + o], x[(int64_t)j * n_in + k], g);
        w[(int64_t)o * n_in + k] -= scale * g;
    }
    if (k == 0) {
        float gb = 0.f;
        for (int j = 0; j < n_sets; ++j) gb += d[(int64_t)j * n_out + o];
        b[o] -= scale * gb;
    }
}
hipError_t launch_mlp_out_delta(const float* a, const float* z, const float* t, float* d, int64_t n, int act, float* err, hipStream_t s) {
    hipLaunchKernelGGL(mlp_out_delta_kernel, dim3(1), dim3(1024), 0, s, a, z, t, d, n, act, err);
    return hipGetLastError();
}
hipError_t launch_mlp_back_delta(const float* w, const float* d, const float* z_prev, float* d_prev, int n_in, int n_out, int n_sets,
                                 int act, hipStream_t s) {
    hipLaunchKernelGGL(mlp_back_delta_kernel, dim3((unsigned)((n_in + 255) / 256), (unsigned)n_sets), dim3(256), 0, s, w, d, z_prev, d_prev,
                       n_in, n_out, act);
    return hipGetLastError();
}
hipError_t launch_mlp_update(float* w, float* b, const float* d, const float* x, int n_in, int n_out, int n_sets, float scale, hipStream_t s) {
    hipLaunchKernelGGL(mlp_update_kernel, dim3((unsigned)((n_in + 255) / 256), (unsigned)n_out), dim3(256), 0, s, w, b, d, x, n_in, n_out,
                       n_sets, scale);
    return hipGetLastError();
}

// ---- folded LayerNorm: row statistics and weight folding -------------------------------------------------------
// (the GEMM epilogues LNFOLD / RESID_LN of gemm_epilogue.h are the other half)

__device__ __forceinline__ void ln_guard_update(float ratio, unsigned int* guard);
// x fp32 [rows, dim] -> plain 16-bit cast + (mean, rstd) per row; one wave per row, like layernorm_kernel.
// Used once per forward (first layer: its input comes from the patch embedding, not from a RESID_LN epilogue).
// `amax_guard` (e4m3 rows only, optional): running maximum of |x| over the first `guard_rows` rows -- the raw rows are the
// q|k|v / fc1 operand of the folded fp8 path and e4m3 saturates at 448 (vithip_api.hip, the guard's second word).
template <typename T, int CH>
__global__ void __launch_bounds__(256)
rowstats_cast_kernel(const float* __restrict__ x, int64_t rows, int dim, float eps, typename T::elem* __restrict__ x16,
                     float* __restrict__ stats, void* __restrict__ xlo_,   // xlo != NULL: the split residual's lo plane (16-bit T: one byte per element; T = e4m3: bf16)
                     int64_t guard_rows, unsigned int* __restrict__ amax_guard) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = dim >> 2;
    const f32x4* xr = (const f32x4*)(x + row * dim);
    f32x4 v[CH];
    float sum = 0.f, amax = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nchunk ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        if constexpr (std::is_same<T, E4M3>::value)
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
    }
    if constexpr (std::is_same<T, E4M3>::value) {
        if (amax_guard) ln_guard_update(row < guard_rows ? amax : 0.f, amax_guard);   // whole wave = one row: uniform
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)dim;
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[i][j] - mean; var += d * d; }
            const typename T::vec4 hq = pack4<T>(v[i][0], v[i][1], v[i][2], v[i][3]);
            *(typename T::vec4*)(x16 + row * dim + 4 * c) = hq;
            if constexpr (!std::is_same<T, E4M3>::value) {
                uint8_t* const xlo = (uint8_t*)xlo_;
                if (xlo) *(uint32_t*)(xlo + row * dim + 4 * c) = lo8_pack4<T>(v[i][0] - (float)hq[0], v[i][1] - (float)hq[1],
                                                                            v[i][2] - (float)hq[2], v[i][3] - (float)hq[3]);
            } else {
                typename BF16::elem* const xlo = (typename BF16::elem*)xlo_;
                if (xlo) {   // fp8 path: hi = the e4m3 bytes just written, lo = bf16 of what they dropped
                    const uint32_t w = (uint32_t)hq;
                    *(typename BF16::vec4*)(xlo + row * dim + 4 * c) =
                        pack4<BF16>(v[i][0] - __builtin_amdgcn_cvt_f32_fp8((int)w, 0), v[i][1] - __builtin_amdgcn_cvt_f32_fp8((int)w, 1),
                                    v[i][2] - __builtin_amdgcn_cvt_f32_fp8((int)w, 2), v[i][3] - __builtin_amdgcn_cvt_f32_fp8((int)w, 3));
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    if (lane == 0) *(float2*)(stats + 2 * row) = make_float2(mean, 1.0f / sqrtf(var / (float)dim + eps));
}

template <typename T>
static hipError_t rowstats_t(const float* x, int64_t rows, int dim, float eps, void* x16, float* stats, hipStream_t s, void* xlo = nullptr,
                             int64_t guard_rows = 0, unsigned int* amax_guard = nullptr) {
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    auto o = (typename T::elem*)x16;
    void* const lo = xlo;
    if (dim <= 256) hipLaunchKernelGGL((rowstats_cast_kernel<T, 1>), grid, block, 0, s, x, rows, dim, eps, o, stats, lo, guard_rows, amax_guard);
    else if (dim <= 512) hipLaunchKernelGGL((rowstats_cast_kernel<T, 2>), grid, block, 0, s, x, rows, dim, eps, o, stats, lo, guard_rows, amax_guard);
    else if (dim <= 768) hipLaunchKernelGGL((rowstats_cast_kernel<T, 3>), grid, block, 0, s, x, rows, dim, eps, o, stats, lo, guard_rows, amax_guard);
    else if (dim <= 1024) hipLaunchKernelGGL((rowstats_cast_kernel<T, 4>), grid, block, 0, s, x, rows, dim, eps, o, stats, lo, guard_rows, amax_guard);
    else if (dim <= 2048) hipLaunchKernelGGL((rowstats_cast_kernel<T, 8>), grid, block, 0, s, x, rows, dim, eps, o, stats, lo, guard_rows, amax_guard);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t launch_rowstats_cast(const float* x, int64_t rows, int dim, float eps, void* x16, float* stats, int dtype,
                                hipStream_t s, int64_t guard_rows, unsigned int* amax_guard) {
    if (rows <= 0 || dim <= 0 || (dim & 3)) return hipErrorInvalidValue;
    if (dtype == VH_DTYPE_FP8) return rowstats_t<E4M3>(x, rows, dim, eps, x16, stats, s, nullptr, guard_rows, amax_guard);   // e4m3 copy of the raw rows (fp8 path, folded LN)
    return dtype == VH_DTYPE_BF16 ? rowstats_t<BF16>(x, rows, dim, eps, x16, stats, s)
                                  : rowstats_t<FP16>(x, rows, dim, eps, x16, stats, s);
}

hipError_t launch_rowstats_split(const float* x, int64_t rows, int dim, float eps, void* hi, void* lo, float* stats, int dtype,
                                 hipStream_t s, int64_t guard_rows, unsigned int* amax_guard) {
    if (rows <= 0 || dim <= 0 || (dim & 3) || !lo) return hipErrorInvalidValue;
    if (dtype == VH_DTYPE_FP8) return rowstats_t<E4M3>(x, rows, dim, eps, hi, stats, s, lo, guard_rows, amax_guard);   // e4m3 hi plane + bf16 lo plane
    return dtype == VH_DTYPE_BF16 ? rowstats_t<BF16>(x, rows, dim, eps, hi, stats, s, lo)
                                  : rowstats_t<FP16>(x, rows, dim, eps, hi, stats, s, lo);
}

// LayerNorm of selected rows of the SPLIT residual (x = hi + lo: a 16-bit plane and a one-byte plane, Lo8<T>) -> fp32: the final LayerNorm of the
// CLS rows in front of the fp32 head.  One wave per row, two-pass statistics like layernorm_kernel.
template <typename T>
__global__ void __launch_bounds__(256)
layernorm_split_kernel(const typename T::elem* __restrict__ hi, const void* __restrict__ lo_, int64_t rows, int dim,
                       int64_t row_stride, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                       float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const typename T::elem* hr = hi + row * row_stride;
    // x[k] = hi + lo: 16-bit hi + scaled e4m3 byte, or (fp8 path) e4m3 hi + bf16 lo
    auto xk = [&](int k) {
        if constexpr (std::is_same<T, E4M3>::value)
            return __builtin_amdgcn_cvt_f32_fp8((int)hr[k], 0) + (float)((const typename BF16::elem*)lo_ + row * row_stride)[k];
        else
            return (float)hr[k] + lo8_unpack1<T>(((const uint8_t*)lo_ + row * row_stride)[k]);
    };
    float sum = 0.f;
    for (int k = lane; k < dim; k += 64) sum += xk(k);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)dim;
    float var = 0.f;
    for (int k = lane; k < dim; k += 64) { const float d = (xk(k)) - mean; var += d * d; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
    const float rstd = 1.0f / sqrtf(var / (float)dim + eps);
    for (int k = lane; k < dim; k += 64) out[row * dim + k] = ((xk(k)) - mean) * rstd * gamma[k] + beta[k];
}
hipError_t launch_layernorm_split(const void* hi, const void* lo, int64_t rows, int dim, int64_t row_stride, const float* gamma,
                                  const float* beta, float eps, float* out32, int dtype, hipStream_t s) {
    if (rows <= 0 || dim <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == VH_DTYPE_FP8)
        hipLaunchKernelGGL(layernorm_split_kernel<E4M3>, grid, block, 0, s, (const E4M3::elem*)hi, lo, rows, dim, row_stride, gamma, beta, eps, out32);
    else if (dtype == VH_DTYPE_BF16)
        hipLaunchKernelGGL(layernorm_split_kernel<BF16>, grid, block, 0, s, (const BF16::elem*)hi, (const void*)lo, rows, dim, row_stride, gamma, beta, eps, out32);
    else
        hipLaunchKernelGGL(layernorm_split_kernel<FP16>, grid, block, 0, s, (const FP16::elem*)hi, (const void*)lo, rows, dim, row_stride, gamma, beta, eps, out32);
    return hipGetLastError();
}

// partials [nblk][rows][2] = (sum, sum of squares) over 64-column blocks -> stats [rows][2] = (mean, rstd).
// Fixed summation order (block 0, 1, ...) keeps the result independent of how the producing tiles were scheduled.
// `guard` (optional): the run-time check on the folded LayerNorm.  The folded GEMM multiplies the rounded
// UNCENTRED rows, so its rounding error relative to the centred signal grows with sqrt(1 + (mean/sigma)^2); every row of
// the first `guard_rows` rows (the real ones: rows behind them are tile padding) contributes |mean| * rstd to a running
// maximum kept as the bits of a non-negative float (ordered like unsigned integers; a NaN ranks above everything and
// trips the guard too).  A wave first READS the word (an L2 hit shared by everybody) and issues its atomic only when it
// would raise it: after the first forward the maximum stands and no atomic is issued at all (one atomic per wave
// unconditionally -- 1 576 on one address per launch -- tripled the kernel's time).
__device__ __forceinline__ void ln_guard_update(float ratio, unsigned int* guard) {
    unsigned int b = __float_as_uint(ratio) & 0x7FFFFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)b, o); b = t > b ? t : b; }
    if ((threadIdx.x & 63) == 0 && b > __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(guard, b);
}
// `amax_guard` (optional; the fp8 path, whose folded operand is the RAW row as e4m3, saturating at 448): running maximum of an
// upper bound of |x| over the real rows -- the square root of the largest 64-column block's sum of squares (>= the block's
// largest element, <= 8 x its rms: it overshoots only where a block already carries several large elements).
__global__ void __launch_bounds__(256)
finalize_stats_kernel(const float* __restrict__ partials, int nblk, int64_t rows, int dim, float eps, float* __restrict__ stats,
                      int64_t guard_rows, unsigned int* __restrict__ guard, unsigned int* __restrict__ amax_guard) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float ratio = 0.f, bound = 0.f;
    if (r < rows) {
        float s1 = 0.f, s2 = 0.f, b2 = 0.f;
        // Sixteen blocks' pairs are requested TOGETHER, then added in block order.  Written as "load, add" per block the loop was
        // a chain of nblk dependent round trips to memory (the compiler cannot unroll a run-time trip count past its waits):
        // 11.5 us per launch for 10 MB, twenty-four launches per forward = 1.4 % of it (round 4: ~3 us).  Missing blocks read as
        // (0, 0): adding zero changes no sum, so the bits are the ones of the sequential loop.
        constexpr int U = 16;
        for (int b0 = 0; b0 < nblk; b0 += U) {
            float2 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                p[u] = b0 + u < nblk ? *(const float2*)(partials + 2 * ((int64_t)(b0 + u) * rows + r)) : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                s1 += p[u].x;
                s2 += p[u].y;
                b2 = fmaxf(b2, p[u].y);
            }
        }
        const float mean = s1 / (float)dim;
        const float var = fmaxf(s2 / (float)dim - mean * mean, 0.f);
        const float rstd = 1.0f / sqrtf(var + eps);
        *(float2*)(stats + 2 * r) = make_float2(mean, rstd);
        if (r < guard_rows) { ratio = fabsf(mean) * rstd; bound = sqrtf(b2); }
    }
    if (guard) ln_guard_update(ratio, guard);   // (whole waves reach this: no early return above)
    if (amax_guard) ln_guard_update(bound, amax_guard);
}
hipError_t launch_finalize_stats(const float* partials, int nblk, int64_t rows, int dim, float eps, float* stats, hipStream_t s,
                                 int64_t guard_rows, unsigned int* guard, unsigned int* amax_guard) {
    if (rows <= 0 || nblk <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finalize_stats_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, partials, nblk, rows, dim, eps, stats,
                       guard_rows, guard, amax_guard);
    return hipGetLastError();
}
// the same check on statistics that already exist (layer 0: written by the rowstats kernels)
__global__ void __launch_bounds__(256)
ln_guard_kernel(const float* __restrict__ stats, int64_t rows, unsigned int* __restrict__ guard) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float ratio = 0.f;
    if (r < rows) {
        const float2 st = *(const float2*)(stats + 2 * r);
        ratio = fabsf(st.x) * st.y;
    }
    ln_guard_update(ratio, guard);
}
hipError_t launch_ln_guard(const float* stats, int64_t rows, unsigned int* guard, hipStream_t s) {
    if (rows <= 0 || !guard) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ln_guard_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, stats, rows, guard);
    return hipGetLastError();
}

// one wave per weight row n: W'[n,k] = T(scale*gamma[k]*W[n,k]); c[n] = sum_k float(W'[n,k]) (of the ROUNDED
// values, so that mean*c cancels exactly what the MFMA accumulates); d[n] = scale*(sum_k beta[k]*W[n,k] + b[n])
template <typename T>
__global__ void __launch_bounds__(256)
fold_ln_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
               const float* __restrict__ beta, int rows, int dim, float scale, typename T::elem* __restrict__ w16,
               float* __restrict__ c, float* __restrict__ d) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= rows) return;
    const float* wr = w + (int64_t)n * dim;
    float cs = 0.f, ds = 0.f;
    for (int k = lane; k < dim; k += 64) {
        const float wv = wr[k];
        const typename T::elem q = (typename T::elem)(scale * gamma[k] * wv);
        w16[(int64_t)n * dim + k] = q;
        cs += (float)q;
        ds = fmaf(beta[k], wv, ds);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cs += __shfl_xor(cs, o); ds += __shfl_xor(ds, o); }
    if (lane == 0) { c[n] = cs; d[n] = scale * (ds + b[n]); }
}
// The same fold for e4m3 weights (fp8 path): W'[n,:] = scale * gamma o W[n,:] goes through the row quantiser
// (quantize_rows_kernel: s0 = amax / 448, bytes = rne_e4m3(W' / s0)); wscale[n] = s0; c[n] = s0 * sum_k decode(byte_k)
// (the sum of what the scaled MFMA multiplies, so that mean * c cancels it exactly); d[n] as above.  dim % 4 == 0.
__global__ void __launch_bounds__(256)
fold_ln_f8_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ gamma,
                  const float* __restrict__ beta, int rows, int dim, float scale, uint8_t* __restrict__ w8,
                  float* __restrict__ wscale, float* __restrict__ c, float* __restrict__ d) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= rows) return;
    const f32x4* wr = (const f32x4*)(w + (int64_t)n * dim);
    const f32x4* gr = (const f32x4*)gamma;
    const f32x4* br = (const f32x4*)beta;
    const int n4 = dim >> 2;
    float amax = 0.f, ds = 0.f;
    for (int k = lane; k < n4; k += 64) {
        const f32x4 wv = wr[k], g = gr[k], be = br[k];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            amax = fmaxf(amax, fabsf(scale * g[j] * wv[j]));
            ds = fmaf(be[j], wv[j], ds);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { amax = fmaxf(amax, __shfl_xor(amax, o)); ds += __shfl_xor(ds, o); }
    const float s0 = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    uint32_t* orow = (uint32_t*)(w8 + (int64_t)n * dim);
    float cs = 0.f;
    for (int k = lane; k < n4; k += 64) {
        const f32x4 wv = wr[k], g = gr[k];
        const int q = (int)pack4_e4m3(__fdiv_rn(scale * g[0] * wv[0], s0), __fdiv_rn(scale * g[1] * wv[1], s0),
                                      __fdiv_rn(scale * g[2] * wv[2], s0), __fdiv_rn(scale * g[3] * wv[3], s0));
        orow[k] = (uint32_t)q;
        cs += (__builtin_amdgcn_cvt_f32_fp8(q, 0) + __builtin_amdgcn_cvt_f32_fp8(q, 1)) +
              (__builtin_amdgcn_cvt_f32_fp8(q, 2) + __builtin_amdgcn_cvt_f32_fp8(q, 3));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cs += __shfl_xor(cs, o);
    if (lane == 0) { wscale[n] = s0; c[n] = s0 * cs; d[n] = scale * (ds + b[n]); }
}
hipError_t launch_fold_ln_f8(const float* w, const float* b, const float* gamma, const float* beta, int rows, int dim,
                             float scale, void* w8, float* wscale, float* c, float* d, hipStream_t s) {
    if (rows <= 0 || dim <= 0 || (dim & 3)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_ln_f8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, b, gamma, beta, rows, dim, scale,
                       (uint8_t*)w8, wscale, c, d);
    return hipGetLastError();
}

hipError_t launch_fold_ln(const float* w, const float* b, const float* gamma, const float* beta, int rows, int dim,
                          float scale, void* w16, float* c, float* d, int dtype, hipStream_t s) {
    if (rows <= 0 || dim <= 0) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype == VH_DTYPE_BF16) hipLaunchKernelGGL(fold_ln_kernel<BF16>, grid, block, 0, s, w, b, gamma, beta, rows, dim, scale, (BF16::elem*)w16, c, d);
    else hipLaunchKernelGGL(fold_ln_kernel<FP16>, grid, block, 0, s, w, b, gamma, beta, rows, dim, scale, (FP16::elem*)w16, c, d);
    return hipGetLastError();
}


// ---- 3x3 image filter on 8-bit single-channel frames (the reference's filter_image pipeline) --------------------
// The reference's kernel `image_process` is absent (netFPGA.cpp:305 names it, no source, no bitstream), so its
// arithmetic is a documented choice here: KIND 0 = 3x3 binomial blur (1 2 1 / 2 4 2 / 1 2 1, +8 >> 4), KIND 1 =
// Sobel |gx| + |gy| saturated to 255; borders replicate the edge pixel.  Integer arithmetic: bit-exact against
// the oracle.  HBM-bound (1 byte in + 1 byte out per pixel); a thread produces 4 adjacent pixels from 3 x 6 inputs.
template <int KIND>
__global__ void __launch_bounds__(256) filter3x3_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int h, int w) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= w) return;
    int p[3][6];
    const bool fast = (w & 3) == 0 && x0 >= 4 && x0 + 8 <= w;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        int yy = y + r - 1;
        yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
        const uint8_t* row = in + (int64_t)yy * w;
        if (fast) {
            const uint32_t a = *(const uint32_t*)(row + x0 - 4), b = *(const uint32_t*)(row + x0), c = *(const uint32_t*)(row + x0 + 4);
            p[r][0] = a >> 24;
            p[r][1] = b & 0xFF; p[r][2] = (b >> 8) & 0xFF; p[r][3] = (b >> 16) & 0xFF; p[r][4] = b >> 24;
            p[r][5] = c & 0xFF;
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                int xx = x0 + i - 1;
                xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                p[r][i] = row[xx];
            }
        }
    }
    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int v;
        if (KIND == 0) {
            v = (p[0][i] + 2 * p[0][i + 1] + p[0][i + 2] + 2 * (p[1][i] + 2 * p[1][i + 1] + p[1][i + 2]) + p[2][i] + 2 * p[2][i + 1] + p[2][i + 2] + 8) >> 4;
        } else {
            const int gx = (p[0][i + 2] + 2 * p[1][i + 2] + p[2][i + 2]) - (p[0][i] + 2 * p[1][i] + p[2][i]);
            const int gy = (p[2][i] + 2 * p[2][i + 1] + p[2][i + 2]) - (p[0][i] + 2 * p[0][i + 1] + p[0][i + 2]);
            v = (gx < 0 ? -gx : gx) + (gy < 0 ? -gy : gy);
            v = v > 255 ? 255 : v;
        }
        packed |= (uint32_t)v << (8 * i);
    }
    uint8_t* o = out + (int64_t)y * w + x0;
    if ((w & 3) == 0) *(uint32_t*)o = packed;
    else
        for (int i = 0; i < 4 && x0 + i < w; ++i) o[i] = (uint8_t)(packed >> (8 * i));
}
hipError_t launch_filter3x3(const uint8_t* in, uint8_t* out, int h, int w, int kind, hipStream_t s) {
    if (h <= 0 || w <= 0 || (kind != 0 && kind != 1)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((w + 1023) / 1024), (unsigned)h), block(256);
    if (kind == 0) hipLaunchKernelGGL(filter3x3_kernel<0>, grid, block, 0, s, in, out, h, w);
    else hipLaunchKernelGGL(filter3x3_kernel<1>, grid, block, 0, s, in, out, h, w);
    return hipGetLastError();
}

// ---- classifier head in fp32 --------------------------------------------------------------------------------------
//   logits[b, c] = sum_k y[b, k] * W[c, k] + bias[c]        y = final-LayerNorm'd CLS rows (fp32), W / bias = the canonical
// fp32 tensors of the blob, read in place.
// The head is 0.002 % of a forward's FLOPs but the LAST rounding point in front of the logits: with 16-bit operands its
// two roundings (CLS rows, head weights) alone were 7 % + of the logit error variance (tools/parity_attribution.py).
// v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain (k-ordered), so this GEMM adds no operand rounding at all.
// One wave = 16 images x 64 classes: the W rows are the MFMA "A" operand, so a lane ends up with 4 consecutive classes of
// one image (one 16-byte store).  A lane's float4 load covers k = kb + 4 (lane >> 4) + {0..3}; element j of the A and of
// the B load go into MFMA j of the group, so both operands see the same k: the 4 x 4 (lane group, j) pairs cover
// kb .. kb + 15 exactly once.
__global__ void __launch_bounds__(256)
head_f32_kernel(const float* __restrict__ y, const float* __restrict__ w, const float* __restrict__ bias,
                float* __restrict__ out, int batch, int classes, int dim, int ncg) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int bb = gw / ncg, cg = gw - bb * ncg;
    if (bb * 16 >= batch) return;
    const int r = lane & 15, g = lane >> 4;
    int b = bb * 16 + r;
    b = b < batch ? b : batch - 1;
    const float* yp = y + (int64_t)b * dim + 4 * g;
    const float* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = cg * 64 + i * 16 + r;
        c = c < classes ? c : classes - 1;
        wp[i] = w + (int64_t)c * dim + 4 * g;
    }
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < dim; kb += 16) {
        const f32x4 yv = *(const f32x4*)(yp + kb);
        f32x4 wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = *(const f32x4*)(wp[i] + kb);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][j], yv[j], acc[i], 0, 0, 0);
    }
    // D[row = class (lane >> 4) * 4 + reg][col = image lane & 15]
    const int bo = bb * 16 + r;
    if (bo < batch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cg * 64 + i * 16 + 4 * g;
            if (c < classes) {   // classes % 4 == 0: a quad is inside or outside as a whole
                const f32x4 bv = *(const f32x4*)(bias + c);
                *(f32x4*)(out + (int64_t)bo * classes + c) = acc[i] + bv;
            }
        }
    }
}

hipError_t launch_head_f32(const float* y, const float* w, const float* bias, float* out, int batch, int classes, int dim,
                           hipStream_t s) {
    if (batch <= 0 || classes <= 0 || (classes & 3) || dim <= 0 || (dim & 15)) return hipErrorInvalidValue;
    const int ncg = (classes + 63) / 64, nb = (batch + 15) / 16;
    const int waves = ncg * nb;
    hipLaunchKernelGGL(head_f32_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, y, w, bias, out, batch, classes, dim, ncg);
    return hipGetLastError();
}

}  // namespace vh
