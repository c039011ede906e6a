/*
 * vit_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * fp32 CPU restatement of the ViT forward that libvithip.so runs on the GPU, plus the
 * synthetic-data generator and the canonical weight-blob writer.
 *
 * PARITY UNPINNED by the reference: /root/reference has no ViT arithmetic to follow
 * (SURVEY.md §0).  The algorithm below is the canonical pre-LN Vision Transformer
 * (Dosovitskiy et al.): conv patch embedding, CLS token, learned position embedding,
 * L x { x += Attn(LN(x)); x += MLP(LN(x)) }, final LN, linear head on token 0, exact-erf
 * GELU, softmax scale dh^-1/2.  It is pinned by tests/golden/ fixtures, written by
 * tests/golden/make_golden.py from transformers.ViTForImageClassification in fp32/fp64.
 *
 * The only reference facts used here: value range of inputs [-1,1) (def/defines.h:11-12)
 * and the "one dense layer = act(W.x + b), W row-major [n_out, n_in]" layout
 * (src/netFPGA.cpp:91-106), which is what every linear layer below is.
 */
#include "oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------- */
/* synthetic data generator                                                              */
/* ------------------------------------------------------------------------------------- */
uint64_t oracle_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

#define IH4_STD 37837.22725 /* std of the sum of four uniform 16-bit integers */

void oracle_fill(float* out, int64_t n, uint64_t seed, uint32_t tensor_id, int kind,
                 float sigma, float offset) {
    const uint64_t stream = oracle_mix64(oracle_mix64(seed) ^ (uint64_t)tensor_id);
    const double scale = (double)sigma / IH4_STD;
    for (int64_t i = 0; i < n; ++i) {
        const uint64_t w = oracle_mix64(stream ^ (uint64_t)i);
        if (kind == 0) {
            const int32_t u = (int32_t)(w >> 40) - (1 << 23);
            out[i] = (float)u * (1.0f / 8388608.0f);
        } else if (kind == 1) {
            const int32_t s = (int32_t)(w & 0xFFFF) + (int32_t)((w >> 16) & 0xFFFF) +
                              (int32_t)((w >> 32) & 0xFFFF) + (int32_t)((w >> 48) & 0xFFFF) -
                              131070;
            const float v = (float)((double)s * scale);
            out[i] = offset + v;
        } else {
            out[i] = offset;
        }
    }
}

/* ------------------------------------------------------------------------------------- */
/* canonical blob                                                                        */
/* ------------------------------------------------------------------------------------- */
typedef struct blob_header {
    char magic[8]; /* "VHBLOB1" */
    int32_t image_size, patch_size, channels, dim, heads, mlp_dim, layers, classes;
    float ln_eps;
    uint32_t pad[5];
} blob_header; /* 64 bytes */

static int tokens_of(const oracle_vit_config* c) {
    const int g = c->image_size / c->patch_size;
    return 1 + g * g;
}

size_t oracle_vit_param_count(const oracle_vit_config* c) {
    const size_t D = c->dim, M = c->mlp_dim, C = c->classes, T = tokens_of(c);
    const size_t kp = (size_t)c->patch_size * c->patch_size * c->channels;
    size_t n = D * kp + D + D + T * D;
    n += (size_t)c->layers * (2 * D + 4 * (D * D + D) + 2 * D + M * D + M + D * M + D);
    n += 2 * D + C * D + C;
    return n;
}

size_t oracle_vit_blob_bytes(const oracle_vit_config* c) {
    return sizeof(blob_header) + 4 * oracle_vit_param_count(c);
}

/* tensor ids of the generator, shared (by specification, not by code) with
 * vit-fpga_amd/csrc and tests/vh_synth.py */
enum { TID_PATCH_W = 1, TID_PATCH_B = 2, TID_CLS = 3, TID_POS = 4, TID_LAYER0 = 16,
       TID_FINAL = 0x7000 };

int oracle_vit_make_blob(const oracle_vit_config* c, uint64_t seed, void* blob, size_t nbytes) {
    if (nbytes < oracle_vit_blob_bytes(c)) return 1;
    blob_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "VHBLOB1", 8);
    h.image_size = c->image_size; h.patch_size = c->patch_size; h.channels = c->channels;
    h.dim = c->dim; h.heads = c->heads; h.mlp_dim = c->mlp_dim; h.layers = c->layers;
    h.classes = c->classes; h.ln_eps = c->ln_eps;
    memcpy(blob, &h, sizeof h);
    float* p = (float*)((char*)blob + sizeof h);
    const size_t D = c->dim, M = c->mlp_dim, C = c->classes, T = tokens_of(c);
    const size_t kp = (size_t)c->patch_size * c->patch_size * c->channels;
    const float sw = 0.02f, sb = 0.02f, sg = 0.05f;
#define GEN(count, tid, sigma, off) do { oracle_fill(p, (int64_t)(count), seed, (tid), 1, (sigma), (off)); p += (count); } while (0)
    GEN(D * kp, TID_PATCH_W, sw, 0.f);
    GEN(D, TID_PATCH_B, sb, 0.f);
    GEN(D, TID_CLS, sw, 0.f);
    GEN(T * D, TID_POS, sw, 0.f);
    for (int l = 0; l < c->layers; ++l) {
        const uint32_t t = TID_LAYER0 + 16u * (uint32_t)l;
        GEN(D, t + 0, sg, 1.f);      /* ln1.weight */
        GEN(D, t + 1, sb, 0.f);      /* ln1.bias   */
        GEN(D * D, t + 2, sw, 0.f);  /* q.weight   */
        GEN(D, t + 3, sb, 0.f);
        GEN(D * D, t + 4, sw, 0.f);  /* k */
        GEN(D, t + 5, sb, 0.f);
        GEN(D * D, t + 6, sw, 0.f);  /* v */
        GEN(D, t + 7, sb, 0.f);
        GEN(D * D, t + 8, sw, 0.f);  /* o */
        GEN(D, t + 9, sb, 0.f);
        GEN(D, t + 10, sg, 1.f);     /* ln2.weight */
        GEN(D, t + 11, sb, 0.f);
        GEN(M * D, t + 12, sw, 0.f); /* fc1 */
        GEN(M, t + 13, sb, 0.f);
        GEN(D * M, t + 14, sw, 0.f); /* fc2 */
        GEN(D, t + 15, sb, 0.f);
    }
    GEN(D, TID_FINAL + 0, sg, 1.f);
    GEN(D, TID_FINAL + 1, sb, 0.f);
    GEN(C * D, TID_FINAL + 2, sw, 0.f);
    GEN(C, TID_FINAL + 3, sb, 0.f);
#undef GEN
    return 0;
}

/* ------------------------------------------------------------------------------------- */
/* operators                                                                             */
/* ------------------------------------------------------------------------------------- */
static inline float dot1(const float* a, const float* w, int K) {
    float s = 0.f;
#pragma omp simd reduction(+ : s)
    for (int k = 0; k < K; ++k) s += a[k] * w[k];
    return s;
}

void oracle_linear(const float* a, const float* w, const float* bias, float* out, int64_t M,
                   int N, int K) {
    const int N4 = N & ~3;
#pragma omp parallel for schedule(static)
    for (int64_t m2 = 0; m2 < (M + 1) / 2; ++m2) {
        const int64_t m = 2 * m2;
        const int two = (m + 1 < M);
        const float* a0 = a + m * K;
        const float* a1 = two ? a0 + K : a0;
        float* o0 = out + m * N;
        float* o1 = two ? o0 + N : o0;
        for (int n = 0; n < N4; n += 4) {
            const float *w0 = w + (int64_t)n * K, *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
            float s00 = 0, s01 = 0, s02 = 0, s03 = 0, s10 = 0, s11 = 0, s12 = 0, s13 = 0;
#pragma omp simd reduction(+ : s00, s01, s02, s03, s10, s11, s12, s13)
            for (int k = 0; k < K; ++k) {
                const float x0 = a0[k], x1 = a1[k];
                s00 += x0 * w0[k]; s01 += x0 * w1[k]; s02 += x0 * w2[k]; s03 += x0 * w3[k];
                s10 += x1 * w0[k]; s11 += x1 * w1[k]; s12 += x1 * w2[k]; s13 += x1 * w3[k];
            }
            const float b0 = bias ? bias[n] : 0.f, b1 = bias ? bias[n + 1] : 0.f,
                        b2 = bias ? bias[n + 2] : 0.f, b3 = bias ? bias[n + 3] : 0.f;
            if (two) { o1[n] = s10 + b0; o1[n + 1] = s11 + b1; o1[n + 2] = s12 + b2; o1[n + 3] = s13 + b3; }
            o0[n] = s00 + b0; o0[n + 1] = s01 + b1; o0[n + 2] = s02 + b2; o0[n + 3] = s03 + b3;
        }
        for (int n = N4; n < N; ++n) {
            const float b = bias ? bias[n] : 0.f;
            if (two) o1[n] = dot1(a1, w + (int64_t)n * K, K) + b;
            o0[n] = dot1(a0, w + (int64_t)n * K, K) + b;
        }
    }
}

void oracle_gelu(float* x, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float v = x[i];
        x[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    }
}

void oracle_layernorm(const float* x, int64_t rows, int dim, const float* gamma,
                      const float* beta, float eps, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float* xr = x + r * dim;
        float* orow = out + r * dim;
        double s = 0.0;
        for (int i = 0; i < dim; ++i) s += xr[i];
        const double mean = s / dim;
        double v = 0.0;
        for (int i = 0; i < dim; ++i) { const double d = xr[i] - mean; v += d * d; }
        const float rstd = (float)(1.0 / sqrt(v / dim + (double)eps));
        const float mf = (float)mean;
        for (int i = 0; i < dim; ++i) orow[i] = (xr[i] - mf) * rstd * gamma[i] + beta[i];
    }
}

static void round16_buf(float* x, int64_t n, int dtype);
static inline float round16_1(float f, int dtype);

/* p_round: 0 = fp32 probabilities; 1 / 2 = the un-normalised exp(s - max) rounded to bf16 / fp16 before the P V
 * product while the denominator keeps the unrounded values (the device's data flow, kernels_attn.hip) */
static void attention_impl(const float* qkv, int batch, int tokens, int heads, int dh, float* out, int p_round) {
    const int D = heads * dh;
    const int64_t ld = 3 * (int64_t)D;
    const float scale = 1.0f / sqrtf((float)dh);
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)tokens);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < batch; ++b) {
            for (int h = 0; h < heads; ++h) {
                const float* base = qkv + (int64_t)b * tokens * ld + h * dh;
                for (int i = 0; i < tokens; ++i) {
                    const float* q = base + i * ld;
                    float mx = -INFINITY;
                    for (int j = 0; j < tokens; ++j) {
                        const float s = dot1(q, base + j * ld + D, dh) * scale;
                        sc[j] = s;
                        if (s > mx) mx = s;
                    }
                    float den = 0.f;
                    for (int j = 0; j < tokens; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
                    if (p_round) for (int j = 0; j < tokens; ++j) sc[j] = round16_1(sc[j], p_round - 1);
                    const float inv = 1.0f / den;
                    float* o = out + ((int64_t)b * tokens + i) * D + h * dh;
                    for (int d = 0; d < dh; ++d) o[d] = 0.f;
                    for (int j = 0; j < tokens; ++j) {
                        const float p = sc[j] * inv;
                        const float* v = base + j * ld + 2 * D;
                        for (int d = 0; d < dh; ++d) o[d] += p * v[d];
                    }
                }
            }
        }
        free(sc);
    }
}

void oracle_attention(const float* qkv, int batch, int tokens, int heads, int dh, float* out) {
    attention_impl(qkv, batch, tokens, heads, dh, out, 0);
}

void oracle_im2col(const float* in, int batch, int image, int patch, int channels, float* out) {
    const int g = image / patch;
    const int kp = patch * patch * channels;
    const int64_t np = (int64_t)g * g;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < batch * np; ++r) {
        const int64_t b = r / np;
        const int p = (int)(r % np), py = p / g, px = p % g;
        float* o = out + r * kp;
        for (int ky = 0; ky < patch; ++ky) {
            const float* src = in + (((b * image) + (py * patch + ky)) * image + px * patch) * channels;
            memcpy(o + ky * patch * channels, src, sizeof(float) * (size_t)patch * channels);
        }
    }
}

static inline float round_bf16_1(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return f; /* inf / nan pass through */
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    memcpy(&f, &u, 4);
    return f;
}
void oracle_round_bf16(float* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = round_bf16_1(x[i]);
}
void oracle_round_fp16(float* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        x[i] = _cvtsh_ss(_cvtss_sh(x[i], _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}

static inline float round16_1(float f, int dtype) {
    return dtype == 0 ? round_bf16_1(f) : _cvtsh_ss(_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC));
}
static void round16_buf(float* x, int64_t n, int dtype) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = round16_1(x[i], dtype);
}

/* ------------------------------------------------------------------------------------- */
/* forward                                                                               */
/* ------------------------------------------------------------------------------------- */
/* ---- OCP e4m3fn (fp8 path, include/vithip.h VH_DTYPE_FP8) ------------------------------------
 * Round-to-nearest-even, saturating at +-448 (the device clamps before v_cvt_pk_fp8_f32, which itself
 * rounds to nearest even: tools/probe_fp8.hip checks this routine against the instruction on 2e5 values). */
float oracle_e4m3_to_float(uint8_t b) {
    const int sg = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) v = NAN;
    else if (e == 0) v = ldexpf((float)m, -9);
    else v = ldexpf(1.0f + (float)m / 8.0f, e - 7);
    return sg ? -v : v;
}
uint8_t oracle_e4m3_from_float(float f) {
    if (isnan(f)) return 0x7F;
    const uint8_t sg = signbit(f) ? 0x80 : 0;
    const float a = fabsf(f);
    if (a >= 448.f) return sg | 0x7E;
    if (a < ldexpf(1.0f, -10)) return sg;          /* below half the smallest subnormal; the tie 2^-10 goes to even = 0 */
    int e;
    (void)frexpf(a, &e);
    int ex = e - 1;                                  /* a = 1.x * 2^ex */
    if (ex < -6) ex = -6;
    const float q = ldexpf(1.0f, ex - 3);            /* spacing of e4m3 at this magnitude */
    const float v = nearbyintf(a / q) * q;           /* exact: a/q is a small dyadic number; default mode = RNE */
    if (v >= 448.f) return sg | 0x7E;
    if (v < ldexpf(1.0f, -6)) return sg | (uint8_t)(int)(v / ldexpf(1.0f, -9));
    (void)frexpf(v, &e);
    ex = e - 1;
    const int m = (int)((v / ldexpf(1.0f, ex) - 1.0f) * 8.0f);
    return sg | (uint8_t)(((ex + 7) << 3) | m);
}
void oracle_quant_e4m3(float* x, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = oracle_e4m3_to_float(oracle_e4m3_from_float(x[i]));
}
/* the load-time weight quantiser (vh_op_quantize_rows): s0 = amax/448 (1 for a zero row), w8 = e4m3(w / s0),
 * scale = s0 * post.  w8 and wq (the decoded values, optional) may be NULL. */
void oracle_quantize_rows(const float* w, int rows, int cols, float post, uint8_t* w8, float* wq, float* scale) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float* wr = w + (size_t)r * cols;
        float amax = 0.f;
        for (int k = 0; k < cols; ++k) { const float a = fabsf(wr[k]); if (a > amax) amax = a; }
        const float s0 = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
        for (int k = 0; k < cols; ++k) {
            const uint8_t b = oracle_e4m3_from_float(wr[k] / s0);
            if (w8) w8[(size_t)r * cols + k] = b;
            if (wq) wq[(size_t)r * cols + k] = oracle_e4m3_to_float(b);
        }
        scale[r] = s0 * post;
    }
}

/* out[m,n] = scale[n] * sum_k a[m,k] wq[n,k] + bias[n] */
static void linear_scaled(const float* a, const float* wq, const float* scale, const float* bias, float* out,
                          int64_t M, int N, int K) {
    oracle_linear(a, wq, NULL, out, M, N, K);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) out[m * N + n] = out[m * N + n] * scale[n] + bias[n];
}

/* VH_DTYPE_FP8 with the LayerNorm FOLDED into the following GEMM (vithip_api.hip prepare_weights / launch_fold_ln_f8):
 * operand = e4m3 of the RAW residual rows; W' = gamma o W through the row quantiser (decoded values wq, scales sc);
 * out[m,n] = rstd_m * (sc_n * sum_k x8[m,k] wq[n,k] - mean_m * c_n) + d_n with c_n = sc_n * sum_k wq[n,k] and
 * d_n = sum_k beta_k W[n,k] + b_n; (mean, rstd) from the fp32 rows.  `x8` is scratch for rows * D floats. */
static void ln_linear_f8_folded(const float* x, int64_t rows, int D, const float* lnw, const float* lnb, const float* W,
                                const float* B, int N, float eps, float* x8, float* out) {
    float* wg = (float*)malloc(sizeof(float) * (size_t)N * D);
    float* wq = (float*)malloc(sizeof(float) * (size_t)N * D);
    float* sc = (float*)malloc(sizeof(float) * (size_t)N);
    float* cd = (float*)malloc(sizeof(float) * (size_t)N * 2);
    float* st = (float*)malloc(sizeof(float) * (size_t)rows * 2);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < D; ++k) wg[(size_t)n * D + k] = lnw[k] * W[(size_t)n * D + k];
    oracle_quantize_rows(wg, N, D, 1.0f, NULL, wq, sc);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        double cs = 0.0, ds = 0.0;
        for (int k = 0; k < D; ++k) { cs += wq[(size_t)n * D + k]; ds += (double)lnb[k] * W[(size_t)n * D + k]; }
        cd[2 * n] = (float)(cs * sc[n]);
        cd[2 * n + 1] = (float)(ds + B[n]);
    }
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < rows; ++m) {
        const float* xr = x + m * D;
        double sum = 0.0, var = 0.0;
        for (int k = 0; k < D; ++k) sum += xr[k];
        const double mean = sum / D;
        for (int k = 0; k < D; ++k) { const double d = xr[k] - mean; var += d * d; }
        st[2 * m] = (float)mean;
        st[2 * m + 1] = (float)(1.0 / sqrt(var / D + eps));
    }
    memcpy(x8, x, sizeof(float) * (size_t)rows * D);
    oracle_quant_e4m3(x8, rows * D);
    oracle_linear(x8, wq, NULL, out, rows, N, D);
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < rows; ++m)
        for (int n = 0; n < N; ++n)
            out[m * N + n] = st[2 * m + 1] * (out[m * N + n] * sc[n] - st[2 * m] * cd[2 * n]) + cd[2 * n + 1];
    free(wg); free(wq); free(sc); free(cd); free(st);
}

/* emul16: 0 = plain fp32; else 1 + dtype (1 = bf16, 2 = fp16) with `mask` choosing WHICH tensors are rounded to that
 * 16-bit type on their way into a matrix product, i.e. where the device (vit-fpga_amd/csrc) holds an MFMA operand:
 *   1 weights (patch + per-layer matrices)   2 LayerNorm output   4 q|k|v   8 softmax probabilities   16 attention output
 *   32 GELU output   64 patch matrix   128 final-LN'd CLS rows   512 head weights   256 = LayerNorm FOLDED into q|k|v and fc1 (vithip_api.hip prepare_weights:
 *   operand = the rounded RAW residual, weights = round(gamma o W), out = rstd (acc - mean c) + d) instead of bit 2.
 * Everything else (accumulation, residual stream, statistics) stays fp32, as on the device.  Used to ATTRIBUTE the
 * device's distance from the fp32 forward to its rounding points (tools/parity_attribution.py). */
static int vit_forward_impl(const oracle_vit_config* c, const void* blob, const float* in,
                       int batch, float* logits, float* hidden, int n_layers_run, int threads, int fp8, int emul16, int mask) {
    const int edt = emul16 - 1;
#define EM(bit) (emul16 && (mask & (bit)))
    blob_header h;
    memcpy(&h, blob, sizeof h);
    if (memcmp(h.magic, "VHBLOB1", 8) != 0) return 1;
    if (h.dim != c->dim || h.layers != c->layers || h.heads != c->heads ||
        h.mlp_dim != c->mlp_dim || h.classes != c->classes || h.image_size != c->image_size ||
        h.patch_size != c->patch_size || h.channels != c->channels)
        return 2;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
    const int D = c->dim, Mh = c->mlp_dim, C = c->classes, H = c->heads, dh = D / H;
    const int P = c->patch_size, CH = c->channels, g = c->image_size / P;
    const int NP = g * g, T = NP + 1, KP = P * P * CH;
    const int L = (n_layers_run < 0 || n_layers_run > c->layers) ? c->layers : n_layers_run;
    const int64_t rows = (int64_t)batch * T;

    const float* w = (const float*)((const char*)blob + sizeof h);
    const float* patch_w = w; w += (size_t)D * KP;
    const float* patch_b = w; w += D;
    const float* cls = w; w += D;
    const float* pos = w; w += (size_t)T * D;
    const float* layer0 = w;
    const size_t per_layer = (size_t)2 * D + 4 * ((size_t)D * D + D) + 2 * D + (size_t)Mh * D + Mh + (size_t)D * Mh + D;
    const float* fin = layer0 + per_layer * c->layers;

    float* x = (float*)malloc(sizeof(float) * rows * D);
    float* y = (float*)malloc(sizeof(float) * rows * D);
    float* qkv = (float*)malloc(sizeof(float) * rows * 3 * D);
    float* att = (float*)malloc(sizeof(float) * rows * D);
    float* hid = (float*)malloc(sizeof(float) * rows * Mh);
    float* col = (float*)malloc(sizeof(float) * (size_t)batch * NP * KP);
    float* pw = (float*)malloc(sizeof(float) * (size_t)D * KP);
    float* wqkv = (float*)malloc(sizeof(float) * 3 * (size_t)D * D);
    float* bqkv = (float*)malloc(sizeof(float) * 3 * (size_t)D);
    if (!x || !y || !qkv || !att || !hid || !col || !pw || !wqkv || !bqkv) return 3;
    float *wtmp = NULL, *cfold = NULL, *dfold = NULL, *stat = NULL;
    if (emul16) {
        const size_t big = (size_t)(Mh > 3 * D ? Mh : 3 * D) * D;
        wtmp = (float*)malloc(sizeof(float) * (big > (size_t)C * D ? big : (size_t)C * D));
        cfold = (float*)malloc(sizeof(float) * (size_t)(Mh > 3 * D ? Mh : 3 * D));
        dfold = (float*)malloc(sizeof(float) * (size_t)(Mh > 3 * D ? Mh : 3 * D));
        stat = (float*)malloc(sizeof(float) * rows * 2);
        if (!wtmp || !cfold || !dfold || !stat) return 3;
    }
    /* fp8 emulation: decoded e4m3 weights (largest matrix: mlp x dim) and their per-row scales */
    float *wq8 = NULL, *wsc = NULL;
    if (fp8) {
        const size_t big = (size_t)(Mh > 3 * D ? Mh : 3 * D) * D;
        wq8 = (float*)malloc(sizeof(float) * big);
        wsc = (float*)malloc(sizeof(float) * (size_t)(Mh > 3 * D ? Mh : 3 * D));
        if (!wq8 || !wsc) return 3;
    }

    /* conv kernel [D][c][ky][kx] -> [D][ky][kx][c] so that it meets the NHWC patch rows */
    for (int d = 0; d < D; ++d)
        for (int ch = 0; ch < CH; ++ch)
            for (int ky = 0; ky < P; ++ky)
                for (int kx = 0; kx < P; ++kx)
                    pw[(size_t)d * KP + (ky * P + kx) * CH + ch] =
                        patch_w[(((size_t)d * CH + ch) * P + ky) * P + kx];
    oracle_im2col(in, batch, c->image_size, P, CH, col);
    if (EM(64)) round16_buf(col, (int64_t)batch * NP * KP, edt);
    if (EM(1)) round16_buf(pw, (int64_t)D * KP, edt);
    oracle_linear(col, pw, patch_b, att /* scratch [batch*NP, D] */, (int64_t)batch * NP, D, KP);
    for (int b = 0; b < batch; ++b) {
        float* xb = x + (int64_t)b * T * D;
        for (int d = 0; d < D; ++d) xb[d] = cls[d] + pos[d];
        for (int p = 0; p < NP; ++p)
            for (int d = 0; d < D; ++d)
                xb[(int64_t)(1 + p) * D + d] = att[((int64_t)b * NP + p) * D + d] + pos[(int64_t)(1 + p) * D + d];
    }

    for (int l = 0; l < L; ++l) {
        const float* p = layer0 + per_layer * l;
        const float* ln1w = p; p += D;
        const float* ln1b = p; p += D;
        const float* qw = p; p += (size_t)D * D;
        const float* qb = p; p += D;
        const float* kw = p; p += (size_t)D * D;
        const float* kb = p; p += D;
        const float* vw = p; p += (size_t)D * D;
        const float* vb = p; p += D;
        const float* ow = p; p += (size_t)D * D;
        const float* ob = p; p += D;
        const float* ln2w = p; p += D;
        const float* ln2b = p; p += D;
        const float* f1w = p; p += (size_t)Mh * D;
        const float* f1b = p; p += Mh;
        const float* f2w = p; p += (size_t)D * Mh;
        const float* f2b = p; p += D;

        memcpy(wqkv, qw, sizeof(float) * (size_t)D * D);
        memcpy(wqkv + (size_t)D * D, kw, sizeof(float) * (size_t)D * D);
        memcpy(wqkv + 2 * (size_t)D * D, vw, sizeof(float) * (size_t)D * D);
        memcpy(bqkv, qb, sizeof(float) * D);
        memcpy(bqkv + D, kb, sizeof(float) * D);
        memcpy(bqkv + 2 * D, vb, sizeof(float) * D);

        if (fp8 == 2) {
            /* VH_DTYPE_FP8 with the folded LayerNorm (the default where dim and mlp_dim are multiples of 256) */
            ln_linear_f8_folded(x, rows, D, ln1w, ln1b, wqkv, bqkv, 3 * D, c->ln_eps, y, qkv);
            oracle_round_bf16(qkv, rows * 3 * D);
            oracle_attention(qkv, batch, T, H, dh, att);
            oracle_quant_e4m3(att, rows * D);
            oracle_quantize_rows(ow, D, D, 1.0f, NULL, wq8, wsc);
            linear_scaled(att, wq8, wsc, ob, y, rows, D, D);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
            ln_linear_f8_folded(x, rows, D, ln2w, ln2b, f1w, f1b, Mh, c->ln_eps, y, hid);
            oracle_gelu(hid, rows * Mh);
            oracle_quant_e4m3(hid, rows * Mh);
            oracle_quantize_rows(f2w, D, Mh, 1.0f, NULL, wq8, wsc);
            linear_scaled(hid, wq8, wsc, f2b, y, rows, D, Mh);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
            continue;
        }
        if (fp8) {
            /* the device's VH_DTYPE_FP8 data flow: every GEMM operand is e4m3 (weights with a per-row scale,
             * activations unscaled), fp32 accumulation, qkv stored as bf16, fp32 residual stream */
            oracle_layernorm(x, rows, D, ln1w, ln1b, c->ln_eps, y);
            oracle_quant_e4m3(y, rows * D);
            oracle_quantize_rows(wqkv, 3 * D, D, 1.0f, NULL, wq8, wsc);
            linear_scaled(y, wq8, wsc, bqkv, qkv, rows, 3 * D, D);
            oracle_round_bf16(qkv, rows * 3 * D);
            oracle_attention(qkv, batch, T, H, dh, att);
            oracle_quant_e4m3(att, rows * D);
            oracle_quantize_rows(ow, D, D, 1.0f, NULL, wq8, wsc);
            linear_scaled(att, wq8, wsc, ob, y, rows, D, D);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
            oracle_layernorm(x, rows, D, ln2w, ln2b, c->ln_eps, y);
            oracle_quant_e4m3(y, rows * D);
            oracle_quantize_rows(f1w, Mh, D, 1.0f, NULL, wq8, wsc);
            linear_scaled(y, wq8, wsc, f1b, hid, rows, Mh, D);
            oracle_gelu(hid, rows * Mh);
            oracle_quant_e4m3(hid, rows * Mh);
            oracle_quantize_rows(f2w, D, Mh, 1.0f, NULL, wq8, wsc);
            linear_scaled(hid, wq8, wsc, f2b, y, rows, D, Mh);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
            continue;
        }
        if (emul16) {
            /* a LayerNorm + linear pair: plain (LN in fp32, its output and the weights rounded on request) or folded */
#define LN_LINEAR(lnw, lnb, W, B, OUT, NOUT)                                                                         \
            if (EM(256)) {                                                                                           \
                /* folded: W' = round(gamma o W), c = row sums of W', d = W beta + b; operand = round(x) */          \
                for (int n = 0; n < (NOUT); ++n) {                                                                   \
                    double cs = 0.0, ds = 0.0;                                                                       \
                    for (int k = 0; k < D; ++k) {                                                                    \
                        const float wg = round16_1((W)[(size_t)n * D + k] * (lnw)[k], edt);                          \
                        wtmp[(size_t)n * D + k] = wg; cs += wg; ds += (double)(W)[(size_t)n * D + k] * (lnb)[k];     \
                    }                                                                                                \
                    cfold[n] = (float)cs; dfold[n] = (float)ds + (B)[n];                                             \
                }                                                                                                    \
                _Pragma("omp parallel for schedule(static)")                                                         \
                for (int64_t r = 0; r < rows; ++r) {                                                                 \
                    const float* xr = x + r * D; double s1 = 0.0, s2 = 0.0;                                          \
                    for (int k = 0; k < D; ++k) { s1 += xr[k]; s2 += (double)xr[k] * xr[k]; y[r * D + k] = round16_1(xr[k], edt); } \
                    const double mean = s1 / D; double var = s2 / D - mean * mean; if (var < 0) var = 0;             \
                    stat[2 * r] = (float)mean; stat[2 * r + 1] = (float)(1.0 / sqrt(var + (double)c->ln_eps));       \
                }                                                                                                    \
                oracle_linear(y, wtmp, NULL, OUT, rows, NOUT, D);                                                    \
                _Pragma("omp parallel for schedule(static)")                                                         \
                for (int64_t r = 0; r < rows; ++r)                                                                   \
                    for (int n = 0; n < (NOUT); ++n)                                                                 \
                        (OUT)[r * (NOUT) + n] = stat[2 * r + 1] * ((OUT)[r * (NOUT) + n] - stat[2 * r] * cfold[n]) + dfold[n]; \
            } else {                                                                                                 \
                oracle_layernorm(x, rows, D, lnw, lnb, c->ln_eps, y);                                                \
                if (EM(2)) round16_buf(y, rows * D, edt);                                                            \
                memcpy(wtmp, W, sizeof(float) * (size_t)(NOUT) * D);                                                 \
                if (EM(1)) round16_buf(wtmp, (int64_t)(NOUT) * D, edt);                                              \
                oracle_linear(y, wtmp, B, OUT, rows, NOUT, D);                                                       \
            }
            LN_LINEAR(ln1w, ln1b, wqkv, bqkv, qkv, 3 * D)
            if (EM(4)) round16_buf(qkv, rows * 3 * D, edt);
            attention_impl(qkv, batch, T, H, dh, att, EM(8) ? 1 + edt : 0);
            if (EM(16)) round16_buf(att, rows * D, edt);
            memcpy(wtmp, ow, sizeof(float) * (size_t)D * D);
            if (EM(1)) round16_buf(wtmp, (int64_t)D * D, edt);
            oracle_linear(att, wtmp, ob, y, rows, D, D);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
            LN_LINEAR(ln2w, ln2b, f1w, f1b, hid, Mh)
            oracle_gelu(hid, rows * Mh);
            if (EM(32)) round16_buf(hid, rows * Mh, edt);
            memcpy(wtmp, f2w, sizeof(float) * (size_t)D * Mh);
            if (EM(1)) round16_buf(wtmp, (int64_t)D * Mh, edt);
            oracle_linear(hid, wtmp, f2b, y, rows, D, Mh);
#pragma omp parallel for schedule(static)
            for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
#undef LN_LINEAR
            continue;
        }
        oracle_layernorm(x, rows, D, ln1w, ln1b, c->ln_eps, y);
        oracle_linear(y, wqkv, bqkv, qkv, rows, 3 * D, D);
        oracle_attention(qkv, batch, T, H, dh, att);
        oracle_linear(att, ow, ob, y, rows, D, D);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
        oracle_layernorm(x, rows, D, ln2w, ln2b, c->ln_eps, y);
        oracle_linear(y, f1w, f1b, hid, rows, Mh, D);
        oracle_gelu(hid, rows * Mh);
        oracle_linear(hid, f2w, f2b, y, rows, D, Mh);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < rows * D; ++i) x[i] += y[i];
    }
    if (hidden) memcpy(hidden, x, sizeof(float) * rows * D);

    /* final LN on the CLS rows only, then the head */
    for (int b = 0; b < batch; ++b)
        oracle_layernorm(x + (int64_t)b * T * D, 1, D, fin, fin + D, c->ln_eps, y + (int64_t)b * D);
    if (EM(128)) round16_buf(y, (int64_t)batch * D, edt);
    if (EM(512)) {
        memcpy(wtmp, fin + 2 * D, sizeof(float) * (size_t)C * D);
        round16_buf(wtmp, (int64_t)C * D, edt);
        oracle_linear(y, wtmp, fin + 2 * D + (size_t)C * D, logits, batch, C, D);
    } else
    oracle_linear(y, fin + 2 * D, fin + 2 * D + (size_t)C * D, logits, batch, C, D);

    free(x); free(y); free(qkv); free(att); free(hid); free(col); free(pw); free(wqkv); free(bqkv);
    free(wq8); free(wsc); free(wtmp); free(cfold); free(dfold); free(stat);
#undef EM
    return 0;
}

/* 3x3 filter on an 8-bit frame, replicated borders (include/vithip.h, vh_filter_*): kind 0 = binomial blur
 * (1 2 1 / 2 4 2 / 1 2 1, +8 >> 4), kind 1 = Sobel |gx| + |gy| saturated.  The reference's `image_process`
 * kernel is absent (netFPGA.cpp:305), so this is the build's definition, not the reference's: PARITY UNPINNED. */
void oracle_filter3x3(const uint8_t* in, uint8_t* out, int h, int w, int kind) {
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int p[3][3];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    int yy = y + r - 1, xx = x + c - 1;
                    yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
                    xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                    p[r][c] = in[(size_t)yy * w + xx];
                }
            int v;
            if (kind == 0) {
                v = (p[0][0] + 2 * p[0][1] + p[0][2] + 2 * p[1][0] + 4 * p[1][1] + 2 * p[1][2] + p[2][0] + 2 * p[2][1] + p[2][2] + 8) >> 4;
            } else {
                const int gx = (p[0][2] + 2 * p[1][2] + p[2][2]) - (p[0][0] + 2 * p[1][0] + p[2][0]);
                const int gy = (p[2][0] + 2 * p[2][1] + p[2][2]) - (p[0][0] + 2 * p[0][1] + p[0][2]);
                v = abs(gx) + abs(gy);
                if (v > 255) v = 255;
            }
            out[(size_t)y * w + x] = (uint8_t)v;
        }
}

int oracle_vit_forward(const oracle_vit_config* c, const void* blob, const float* in,
                       int batch, float* logits, float* hidden, int n_layers_run, int threads) {
    return vit_forward_impl(c, blob, in, batch, logits, hidden, n_layers_run, threads, 0, 0, 0);
}
int oracle_vit_forward_emul16(const oracle_vit_config* c, const void* blob, const float* in, int batch, float* logits,
                              int dtype, int mask, int threads) {
    if (dtype != 0 && dtype != 1) return 4;
    return vit_forward_impl(c, blob, in, batch, logits, NULL, -1, threads, 0, 1 + dtype, mask);
}
int oracle_vit_forward_fp8(const oracle_vit_config* c, const void* blob, const float* in,
                           int batch, float* logits, float* hidden, int n_layers_run, int threads) {
    return vit_forward_impl(c, blob, in, batch, logits, hidden, n_layers_run, threads, 1, 0, 0);
}
int oracle_vit_forward_fp8_folded(const oracle_vit_config* c, const void* blob, const float* in,
                                  int batch, float* logits, float* hidden, int n_layers_run, int threads) {
    return vit_forward_impl(c, blob, in, batch, logits, hidden, n_layers_run, threads, 2, 0, 0);
}
