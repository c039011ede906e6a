// vh_kernels.h — internal launch API between the C-ABI layer (vithip_api.hip) and the kernels.
#pragma once

#include <atomic>
#include <cstdlib>

#include "vh_common.h"

namespace vh {

struct GemmArgs {
    const void* a;      // [M, K] 16-bit, K contiguous
    const void* w;      // [N, K] 16-bit, K contiguous
    const float* bias;  // [N]
    void* out;          // 16-bit or fp32 [M(or remapped rows), N]
    int64_t M;
    int N, K;
    int epilogue;       // VH_EPI_*
    const float* aux;   // EPI_PATCH: pos-emb [tokens, N]
    int aux_i;          // EPI_PATCH: patches per image
    int dtype;          // VH_DTYPE_*
    int variant;        // 0 auto, 1 = 128x128, 2 = 256x256 two-stage, 5 = ping-pong, 6 = persistent ping-pong
    const float* stats = nullptr;  // LNFOLD*: [M][2] (mean, rstd)
    void* out16 = nullptr;         // RESID_LN: 16-bit copy of the updated rows
    float* partials = nullptr;     // RESID_LN: [N/64][M][2]
    const float* wscale = nullptr; // launch_gemm_fp8 with an LNFOLD* epilogue: the per-output-channel weight scales [N]
                                   // (`aux` carries c_n there); every other fp8 epilogue passes them in `aux`
    // ping-pong forms 5 / 7 only: launch tiles [tile_begin, tile_begin + tile_count) of the n-fastest 256x256 tile order
    // (tile t = (t / tiles_n, t % tiles_n)); tile_count == 0 = all tiles.  Lets a caller split off the last, partly
    // filled round of tiles and overlap it with other work (vithip_api.hip, tail overlap).
    int tile_begin = 0, tile_count = 0;
    // PATCH_SPLIT: row stride of `partials` = number of TOKEN rows the statistics buffer is laid out for (0 = images x tokens)
    int64_t prow = 0;
    // The MLP hidden activation in its TILED layout (gemm_epilogue.h, OTILED; persistent 16-bit form only: the launcher
    // refuses otherwise -- ask gemm_tiled_applies first).  out_tiled: an LNFOLD_GELU / BIAS_GELU launch writes it; ab_tiled: A and W
    // of this launch are both in that layout (16-row blocks).
    int out_tiled = 0, ab_tiled = 0;
};
// does a 16-bit GEMM of this shape take the persistent form, the only one that reads / writes the tiled layout?
bool gemm_tiled_applies(int64_t M, int N, int K);
// W fp32 [rows, cols] -> 16-bit in the tiled layout (16-row blocks, natural column order)
hipError_t launch_cast_tiled_w(const float* w, int rows, int cols, void* w16, int dtype, hipStream_t stream);
// e4m3 operands (launch_gemm_fp8): the same question for the 128-byte K-tiles of the fp8 form, and the byte matrix -> tiled copy
bool gemm_tiled_applies_f8(int64_t M, int N, int K);
hipError_t launch_tile_bytes(const void* w8, int rows, int cols, void* out, hipStream_t stream);

// row stride of the partial-sum buffer a launch writes: PATCH_SPLIT lands on token rows (images x (patches + 1) unless the
// caller's buffer is laid out for more, e.g. rows padded to whole tiles); the other forms use their own M inside the kernel
inline int64_t gemm_prow(const GemmArgs& g) {
    if (g.epilogue != VH_EPI_PATCH_SPLIT) return g.M;
    return g.prow > 0 ? g.prow : (g.M / g.aux_i) * (int64_t)(g.aux_i + 1);
}

// every launcher only enqueues on `stream`; returns hipSuccess or the launch error
hipError_t launch_gemm(const GemmArgs& g, hipStream_t stream);
// fp8 form (kernels_gemm5.hip): a, w = e4m3 [M,K] / [N,K]; aux = per-output-channel weight scale [N];
// epilogue BIAS (bf16 out), BIAS_GELU (e4m3 out), BIAS_RESID / BIAS_F32 (fp32 out).  K % 128 == 0, N % 4 == 0.
hipError_t launch_gemm_fp8(const GemmArgs& g, hipStream_t stream);
const char* gemm_check(const GemmArgs& g);  // NULL if the shape is supported, else the reason
int gemm_pick_variant(int64_t M, int N, int epilogue);
int gemm_pp_variant(int epilogue);  // 5, 6 or 7 (default 6 = persistent; env VH_GEMM_PP forces one)

hipError_t launch_layernorm(const float* x, int64_t rows, int dim, int64_t row_stride,
                            const float* gamma, const float* beta, float eps, void* out16,
                            int dtype, hipStream_t stream);
// dtype = VH_DTYPE_* (16-bit / e4m3 result) or VH_DTYPE_F32_INTERNAL (fp32 result: the final LayerNorm in front of the head)
constexpr int VH_DTYPE_F32_INTERNAL = 100;
// classifier head, fp32 operands (exact-fp32 MFMA): logits[b, c] = y[b, :] . w[c, :] + bias[c]; classes % 4 == 0, dim % 16 == 0
hipError_t launch_head_f32(const float* y, const float* w, const float* bias, float* out, int batch, int classes, int dim,
                           hipStream_t stream);
// q columns of qkv16 arrive scaled by kAttnQScale = 64^-1/2 * log2(e) (folded into Wq and bq when the weights are
// prepared): the score MFMAs then produce log2-domain scores and the softmax is exp2 without a multiply per score
constexpr float kAttnQScale = 0.125f * 1.4426950408889634f;
// ticket: 4 bytes of device scratch owned by the caller's stream (work-queue counter of the staged ring form; zeroed by
// the launch unless the caller says it already is: a forward zeroes one word per layer with ONE memset), or nullptr for
// equal static shares per workgroup
// out_tiled (16-bit results, ring forms: ask attention_tiled_applies): the output in the 16-row-blocked layout the out-projection's
// tiled operand DMA reads ([rows / 16][dim / 8][16][8], rows = batch * tokens rounded up to 16 by the caller's buffer)
hipError_t launch_attention(const void* qkv16, int batch, int tokens, int heads, void* out16,
                            int dtype, unsigned int* ticket, hipStream_t stream, bool ticket_zeroed = false, bool out_tiled = false,
                            int64_t in_hm_rows = 0);   // in_hm_rows != 0: q|k|v is head-major, [3][heads][in_hm_rows][64] (needs out_tiled)
bool attention_tiled_applies(int batch, int tokens, int heads);
// class-token query only: out16 [batch][heads * 64] (VH_FLAG_CLS_TAIL); 16-bit dtypes, tokens <= 1024
hipError_t launch_attention_cls(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype, hipStream_t stream);
size_t attention_lds_bytes(int tokens);
hipError_t launch_im2col(const float* in_nhwc, int batch, int image, int patch, int channels,
                         void* out16, int dtype, hipStream_t stream);
// patch embedding with the gather inside the GEMM's A loader (kernels_patch.hip): NHWC fp32 images -> the split residual's
// planes + the first row statistics' partial sums, no patch matrix in memory
bool patch_fused_supported(int image, int patch, int channels, int dim);
hipError_t launch_patch_fused(const float* images, int batch, int image, int patch, int channels, const void* wp16, const float* bias,
                              const float* pos, void* hi, void* lo, float* partials, int64_t prow, int dim, int dtype, hipStream_t stream);
hipError_t launch_cls_rows(float* x, const float* cls, const float* pos, int batch, int tokens,
                           int dim, hipStream_t stream);
// the class-token rows of the SPLIT residual: (hi, lo) planes of cls + pos[0] and their per-64-column partial sums
hipError_t launch_cls_rows_split(void* hi, void* lo, float* partials, int64_t prow, const float* cls, const float* pos, int batch,
                                 int tokens, int dim, int dtype, hipStream_t stream);
hipError_t launch_cast(const float* in, void* out16, int64_t n, int dtype, hipStream_t stream);
hipError_t launch_fill(float* out, int64_t n, uint64_t seed, uint32_t tensor_id, int kind,
                       float sigma, float offset, hipStream_t stream);
// fp8 weight quantisation: w fp32 [rows, cols] -> e4m3 [rows, cols] with scale[r] = post * amax_r / 448 (amax 0 -> post),
// q = rne_e4m3(w / (amax_r / 448)); `post` folds a constant factor into the scale (the q rows' softmax scale)
// out = decode(e4m3(w / s0)) * s0 per row, s0 = amax_row / 448: the values of a weight-only e4m3 matrix (VH_FLAG_W8_E4M3)
hipError_t launch_fake_quant_rows(const float* w, int rows, int cols, float* out, hipStream_t stream);
hipError_t launch_quantize_rows(const float* w, int rows, int cols, float post, void* w8, float* scale, hipStream_t stream);
// weight preparation (fp32 canonical tensors -> compute layout)
hipError_t launch_pack_qkv(const float* qw, const float* qb, const float* kw, const float* kb,
                           const float* vw, const float* vb, int dim, float q_scale, void* w16,
                           float* b32, int dtype, hipStream_t stream);
hipError_t launch_permute_patch(const float* w_nchw, int dim, int channels, int patch, void* w16,
                                int dtype, hipStream_t stream);
// folded LayerNorm helpers
// amax_guard (optional, e4m3 rows only): running maximum of |x| over the first guard_rows rows, as float bits
hipError_t launch_rowstats_cast(const float* x, int64_t rows, int dim, float eps, void* x16, float* stats, int dtype,
                                hipStream_t stream, int64_t guard_rows = 0, unsigned int* amax_guard = nullptr);
// guard (optional): running maximum of |mean| * rstd over the first `guard_rows` rows, as float bits (kernels_misc.hip)
// amax_guard (optional): running maximum of an upper bound of |x| (largest 64-column block's root sum of squares)
hipError_t launch_finalize_stats(const float* partials, int nblk, int64_t rows, int dim, float eps, float* stats,
                                 hipStream_t stream, int64_t guard_rows = 0, unsigned int* guard = nullptr,
                                 unsigned int* amax_guard = nullptr);
hipError_t launch_ln_guard(const float* stats, int64_t rows, unsigned int* guard, hipStream_t stream);
// split residual (x = hi + lo, two 16-bit planes; gemm_epilogue.h RESID_SPLIT): fp32 rows -> planes + row statistics, and
// the fp32 LayerNorm of selected rows (the CLS rows) of the planes
hipError_t launch_rowstats_split(const float* x, int64_t rows, int dim, float eps, void* hi, void* lo, float* stats, int dtype,
                                 hipStream_t stream, int64_t guard_rows = 0, unsigned int* amax_guard = nullptr);
hipError_t launch_layernorm_split(const void* hi, const void* lo, int64_t rows, int dim, int64_t row_stride, const float* gamma,
                                  const float* beta, float eps, float* out32, int dtype, hipStream_t stream);
// fp8 path: the folded weights as e4m3 rows + scales (wscale), c = wscale * sum of the decoded row, d as launch_fold_ln
hipError_t launch_fold_ln_f8(const float* w, const float* b, const float* gamma, const float* beta, int rows, int dim,
                             float scale, void* w8, float* wscale, float* c, float* d, hipStream_t stream);
hipError_t launch_fold_ln(const float* w, const float* b, const float* gamma, const float* beta, int rows, int dim,
                          float scale, void* w16, float* c, float* d, int dtype, hipStream_t stream);
// MLP mode: y = act(W x + b), W [n_out, n_in] fp32
hipError_t launch_dense_layer(const float* w, const float* b, const float* x, float* y, int n_in,
                              int n_out, int n_vec, int activation, hipStream_t stream, float* z = nullptr);
// MLP-mode training (kernels_misc.hip): last-layer delta + error, hidden-layer delta, parameter update
hipError_t launch_mlp_out_delta(const float* a, const float* z, const float* t, float* d, int64_t n, int act, float* err,
                                hipStream_t stream);
hipError_t launch_mlp_back_delta(const float* w, const float* d, const float* z_prev, float* d_prev, int n_in, int n_out,
                                 int n_sets, int act, hipStream_t stream);
hipError_t launch_mlp_update(float* w, float* b, const float* d, const float* x, int n_in, int n_out, int n_sets, float scale,
                             hipStream_t stream);

// 3x3 filter on 8-bit frames (filter_image): kind 0 = binomial blur, 1 = Sobel |gx|+|gy|
hipError_t launch_filter3x3(const uint8_t* in, uint8_t* out, int h, int w, int kind, hipStream_t stream);

// ---- launcher-side caches: reached concurrently from the device-group member threads (vh_group_*, one host thread per
// device), so none of them is a plain static.  Environment knobs are function-local `static const` values (C++11: the
// initialiser runs exactly once, other threads wait for it); per-device facts are tables of relaxed atomics indexed by
// the device ordinal (every writer stores the same value).
constexpr int kMaxDevices = 64;
inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
// number of CUs of the CURRENT device (0 on error)
inline int device_num_cu() {
    static std::atomic<int> table[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    int n = table[dev].load(std::memory_order_relaxed);
    if (n) return n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
    table[dev].store(n, std::memory_order_relaxed);
    return n;
}
// Opt a kernel in to `bytes` of dynamic LDS on the CURRENT device.  The attribute is per device, so the record of
// what has been set is too: `done[d]` = bytes already granted on device d (one table per kernel instantiation).
using LdsDone = std::atomic<int>[kMaxDevices];
inline hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes, LdsDone& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    if ((size_t)done[dev].load(std::memory_order_acquire) >= bytes) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done[dev].store((int)bytes, std::memory_order_release);
    return e;
}

}  // namespace vh
