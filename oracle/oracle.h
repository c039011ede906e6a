/*
 * oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU (plain C, fp32) restatement of the hot path that libvithip.so implements on the
 * GPU.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * liboracle.so; nothing under vit-fpga_amd/ links, includes or calls it.
 *
 * PARITY UNPINNED (by the reference).  /root/reference holds no Vision Transformer, no
 * kernel source for `network_v1` (named at src/netFPGA.cpp:250, bitstream absent), no CPU
 * path, no tests and no golden vectors.  What the reference does
 * define, and what this oracle follows line by line, is the MLP-mode data layout:
 *   - weight / bias flatten order            src/netFPGA.cpp:91-106
 *   - size arithmetic (n_params, n_neurons)  src/netFPGA.cpp:68-76
 *   - kernel argument list of network_v1     src/netFPGA.cpp:427-436, 499-502
 *   - random init formula                    src/netFPGA.cpp:82-88
 * The ViT arithmetic follows the canonical pre-LN ViT and is pinned instead by an
 * independent implementation (transformers' ViTForImageClassification, random weights,
 * built offline in the build container): tests/golden/make_golden.py writes the fixtures,
 * tests/test_oracle_golden.py checks this file against them.
 */
#ifndef VH_ORACLE_H
#define VH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_vit_config {
    int32_t image_size, patch_size, channels, dim, heads, mlp_dim, layers, classes;
    float ln_eps;
} oracle_vit_config;

/* ---- synthetic data generator (DESIGN.md "synthetic data"): independent restatement ---- */
uint64_t oracle_mix64(uint64_t x);
/* kind 0: uniform[-1,1) ; kind 1: offset + Irwin-Hall(4)*sigma ; kind 2: constant offset */
void oracle_fill(float* out, int64_t n, uint64_t seed, uint32_t tensor_id, int kind,
                 float sigma, float offset);

/* ---- canonical fp32 weight blob --------------------------------------------------------- */
size_t oracle_vit_param_count(const oracle_vit_config* c);
size_t oracle_vit_blob_bytes(const oracle_vit_config* c); /* 64-byte header + params*4 */
/* writes header + seeded synthetic tensors in canonical order */
int oracle_vit_make_blob(const oracle_vit_config* c, uint64_t seed, void* blob, size_t nbytes);

/* ---- the ViT forward --------------------------------------------------------------------- */
/* in: [batch, image, image, channels] fp32 NHWC.  logits: [batch, classes].
 * hidden (optional, may be NULL): residual stream after the last layer [batch*T, D].
 * n_layers_run < 0 runs all layers. threads <= 0 uses all cores. */
int oracle_vit_forward(const oracle_vit_config* c, const void* blob, const float* in_nhwc,
                       int batch, float* logits, float* hidden, int n_layers_run, int threads);

/* The same forward with the data flow of the device's VH_DTYPE_FP8 mode emulated (BASELINE config 5): the four
 * per-layer GEMMs take OCP e4m3 operands -- weights quantised per output channel, activations cast unscaled and
 * saturating -- with fp32 accumulation; qkv rounded to bf16; everything else fp32.  A quantised pipeline is
 * chaotic (a last-bit difference before a cast flips a 6 % rounding), so this pins the device STATISTICALLY: the
 * device must be as close to the fp32 forward as this emulation is (tests/test_gpu_fp8.py). */
int oracle_vit_forward_fp8(const oracle_vit_config* c, const void* blob, const float* in_nhwc,
                           int batch, float* logits, float* hidden, int n_layers_run, int threads);
/* the same with the LayerNorm folded into q|k|v and fc1 (e4m3 copy of the RAW residual rows as the operand) */
int oracle_vit_forward_fp8_folded(const oracle_vit_config* c, const void* blob, const float* in_nhwc,
                           int batch, float* logits, float* hidden, int n_layers_run, int threads);
/* The fp32 forward with the device's 16-bit ROUNDING POINTS emulated one by one (dtype 0 = bf16, 1 = fp16; mask bits:
 * 1 patch and per-layer weights, 2 LayerNorm output, 4 q|k|v, 8 softmax probabilities, 16 attention output, 32 GELU
 * output, 64 patch matrix, 128 final-LN'd CLS rows, 512 head weights, 256 LayerNorm folded into q|k|v / fc1 instead of bit 2).  Accumulation, residual
 * stream and statistics stay fp32.  Attributes the device's distance from the fp32 oracle to its MFMA operand
 * roundings (tools/parity_attribution.py, DESIGN.md "Numerics"). */
int oracle_vit_forward_emul16(const oracle_vit_config* c, const void* blob, const float* in_nhwc, int batch,
                              float* logits, int dtype, int mask, int threads);
/* e4m3fn: round-to-nearest-even, saturating at +-448 */
uint8_t oracle_e4m3_from_float(float f);
float oracle_e4m3_to_float(uint8_t b);
void oracle_quant_e4m3(float* x, int64_t n); /* in place: x -> decode(encode(x)) */
/* s0 = amax(row)/448 (1 for a zero row); w8 = e4m3(w/s0) (bytes, optional), wq = decoded (optional),
 * scale[row] = s0*post */
void oracle_quantize_rows(const float* w, int rows, int cols, float post, uint8_t* w8, float* wq, float* scale);

/* ---- single operators (for per-kernel parity tests) ------------------------------------- */
/* out[m,n] = sum_k a[m,k]*w[n,k] + bias[n]   (bias may be NULL) */
void oracle_linear(const float* a, const float* w, const float* bias, float* out, int64_t M,
                   int N, int K);
void oracle_gelu(float* x, int64_t n);
void oracle_layernorm(const float* x, int64_t rows, int dim, const float* gamma,
                      const float* beta, float eps, float* out);
/* qkv: [batch*tokens, 3*heads*dh] (q NOT pre-scaled; scale = dh^-0.5 applied inside);
 * out: [batch*tokens, heads*dh] */
void oracle_attention(const float* qkv, int batch, int tokens, int heads, int dh, float* out);
/* NHWC image -> [batch*np, patch*patch*channels], k-order (ky, kx, c) */
void oracle_im2col(const float* in_nhwc, int batch, int image, int patch, int channels,
                   float* out);
/* round-to-nearest-even fp32 -> bf16 / fp16 -> fp32 (for building expected values of
 * reduced-precision kernels) */
void oracle_round_bf16(float* x, int64_t n);
void oracle_round_fp16(float* x, int64_t n);

/* ---- filter_image (3x3 filter on 8-bit frames; the build's definition, see vit_oracle.c) ---- */
void oracle_filter3x3(const uint8_t* in, uint8_t* out, int h, int w, int kind);

/* ---- MLP mode (the reference's real launch_forward semantics) ---------------------------- */
/* activation codes = include/vithip.h VH_ACT_* */
int oracle_mlp_forward(int n_ins, int n_layers, const int* n_p_l, const float* params,
                       const float* bias, int activation, const float* inputs, float* outputs);
/* the reference ctor's random branch: float(rand() % 200 - 100) / 100 (netFPGA.cpp:82-88),
 * driven by an explicit LCG so it is reproducible without libc's rand() state */
void oracle_mlp_random_params(float* params, size_t n_params, float* bias, size_t n_neurons,
                              uint32_t seed);

/* Training of the dense chain (init_gradient / launch_gradient, netFPGA.cpp:518-580: commented-out code in the
 * reference -- PARITY UNPINNED; the definitions are this build's, see mlp_oracle.c).  Full-batch gradient descent on
 * 1/2 |a_L - t|^2: `params` / `bias` are updated in place, errors[it] = sum over sets and outputs of |a_L - t| BEFORE
 * the update of iteration it; an iteration with error <= error_threshold ends the loop (later entries 0).
 * set_ins [n_sets][n_ins], set_outs [n_sets][n_p_l[L-1]]. */
int oracle_mlp_train(int n_ins, int n_layers, const int* n_p_l, float* params, float* bias, int activation,
                     const float* set_ins, const float* set_outs, int n_sets, int iterations, float error_threshold,
                     float multiplier, float* errors);

#ifdef __cplusplus
}
#endif
#endif
