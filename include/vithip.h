/*
 * vithip.h — C ABI of libvithip.so, the MI355X (gfx950) backend that replaces the
 * device side of LimpBunion22/VIT-FPGA's `fpga::net_fpga`.
 *
 * The reference has no FFI layer of its own: `src/netFPGA.cpp` talks to OpenCL
 * directly.  Every entry point below therefore cites the reference *call site*
 * (file:line under /root/reference) whose job it takes over.  Signatures carry only
 * plain pointers, sizes and PODs (no HIP, torch or C++ types) so the header is usable
 * from C, from plain g++ (host/netHIP.cpp), and from ctypes/cgo/JNI style bindings
 * (INTEGRATION.md shows the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns VH_OK (0) or a VH_ERR_* code; the human-readable reason is
 *     available through vh_last_error().  Nothing here calls exit(): the reference's
 *     fatal-on-error convention (aocl_utils::checkError -> cleanup() -> exit,
 *     netFPGA.cpp:274-278) is re-created, if wanted, by the C++ class above this ABI.
 *   - "dev" pointers are device (HBM) addresses of the context's GPU, "host" pointers are
 *     ordinary process memory.  The caller owns every buffer it passes in.
 *   - a context is not re-entrant; distinct contexts may be used from distinct threads.
 */
#ifndef VITHIP_H
#define VITHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VH_ABI_VERSION 1

/* status codes */
#define VH_OK 0
#define VH_ERR_INVALID 1     /* bad argument / shape the kernels do not support       */
#define VH_ERR_HIP 2         /* a HIP runtime call failed (message has hipGetErrorString) */
#define VH_ERR_STATE 3       /* call order violated (e.g. forward before weights)     */
#define VH_ERR_UNSUPPORTED 4 /* valid request that this build does not implement      */
#define VH_ERR_NO_DEVICE 5   /* no gfx950 device visible                              */
#define VH_ERR_RING_FULL 6   /* vh_ring_submit: every slot is in flight (reference: "PILA LLENA") */
#define VH_ERR_RING_EMPTY 7  /* vh_ring_collect: nothing in flight (reference: "PILA VACIA")      */

/* arithmetic type of the dense contractions (MFMA operand type; accumulation is fp32) */
#define VH_DTYPE_BF16 0
#define VH_DTYPE_FP16 1
/* fp8 GEMMs (BASELINE config 5): the four per-layer GEMMs run on v_mfma_scale_f32_16x16x128_f8f6f4 with OCP
 * e4m3 operands -- weights quantised at load with one fp32 scale per output channel, activations cast unscaled
 * (saturating at +-448) by the kernel that produces them; fp32 accumulation, fp32 residual stream; patch embedding,
 * attention and the head stay bf16.  Needs dim % 128 == 0 and mlp_dim % 128 == 0. */
#define VH_DTYPE_FP8 2

/* activation selector of MLP mode.  The reference stores `activations = 1 // RELU2`
 * (netFPGA.cpp:79) but never defines it (the network_v1 kernel source is absent), so the
 * numeric codes below are this build's definition; see DESIGN.md "parity unpinned". */
#define VH_ACT_IDENTITY 0
#define VH_ACT_RELU2 1    /* min(max(x,0), MAX_RANGE=1)  (def/defines.h:11-12 value range) */
#define VH_ACT_RELU 2
#define VH_ACT_HARDTANH 3 /* clamp(x, MIN_RANGE=-1, MAX_RANGE=1) */
#define VH_ACT_GELU 4     /* exact erf GELU */

/* Vision-Transformer shape.  Replaces the role `net::net_data` (def/defines.h:14-23) plays
 * for the MLP: it is what sizes the device buffers (_init_kernel, netFPGA.cpp:402-441). */
typedef struct vh_config {
    int32_t image_size; /* square input side, e.g. 224                     */
    int32_t patch_size; /* e.g. 16                                         */
    int32_t channels;   /* 3                                               */
    int32_t dim;        /* D, hidden size (multiple of 64)                 */
    int32_t heads;      /* H, dim/heads must be 64                         */
    int32_t mlp_dim;    /* M (multiple of 64)                              */
    int32_t layers;     /* L                                               */
    int32_t classes;    /* C (multiple of 4)                               */
    int32_t dtype;      /* VH_DTYPE_*                                      */
    int32_t max_batch;  /* workspace is sized for this many images         */
    float ln_eps;       /* 1e-6 for the canonical ViT                      */
    int32_t flags;      /* VH_FLAG_* (0 = library defaults); unknown bits are rejected */
} vh_config;

/* vh_config.flags.  LayerNorm folding: with dim and mlp_dim multiples of 256 the two LayerNorms of a layer can be folded
 * into the neighbouring GEMMs (DESIGN.md "LayerNorm folded into the GEMMs"): the GEMM then multiplies the RAW
 * 16-bit-rounded residual rows, whose rounding error relative to the centred signal grows with
 * sqrt(1 + (row mean / row sigma)^2).  Three settings:
 *   flags == 0 (default)   the GUARDED fold: folded, and every forward measures max |mean| / sigma over its rows
 *                          (vh_get_ln_guard).  The decision is taken first WHEN THE WEIGHTS ARE LOADED: every vh_load_weights* /
 *                          vh_init_weights_seeded ends with one calibration forward of a seeded image, and a checkpoint whose
 *                          rows exceed the threshold (0.5; environment VH_LN_GUARD) -- a property of the weights far more than
 *                          of the image -- leaves the load on the stand-alone LayerNorm, so no entry point, asynchronous
 *                          ones included, ever returns a batch from a fold the weights had tripped.  Behind that, every
 *                          forward still measures: once a COMPLETED forward has exceeded the threshold because of its DATA
 *                          the context switches for good (the weights are prepared again from the resident blob) at the
 *                          next forward entry or vh_synchronize; the synchronous vh_forward repeats the very forward that
 *                          tripped it, the asynchronous entry points have by then delivered that batch from the folded
 *                          path (vh_get_ln_guard reports `tripped`), and from then on logit bits differ from a context
 *                          that never tripped -- bit-reproducibility across runs holds for ON / OFF, and for the default
 *                          as long as no input trips the backstop.
 *                          VH_DTYPE_FP8 contexts guard a second quantity: the folded GEMMs multiply the RAW residual rows
 *                          as e4m3, which saturates at 448, so the statistics kernels also track max |x| (vh_get_fp8_guard)
 *                          and rows beyond 448 switch the context to the stand-alone LayerNorm operand (normalised values
 *                          fit e4m3) in the same way, calibration forward included.
 *   VH_FLAG_LN_FOLD_ON     always folded: the explicit throughput choice.  The guard still measures, never switches.
 *   VH_FLAG_LN_FOLD_OFF    always the stand-alone LayerNorm kernel.
 * With ON or OFF the path depends on the model shape and the flags ONLY (never on max_batch or on the data), so a given
 * image gives the same logit bits from every context of one configuration; with the default it depends, in addition, on
 * whether the guard has tripped (vh_get_ln_fold tells).  Environment overrides, honoured only when flags == 0 and meant
 * for A/B tools: VH_LN_FOLD=0|1 (forces the path and disables the switch), VH_RESID_SPLIT=0 (fp32 residual stream instead
 * of the two 16-bit planes). */
#define VH_FLAG_LN_FOLD_OFF 1 /* always run the stand-alone LayerNorm kernel                */
#define VH_FLAG_LN_FOLD_ON 2  /* always fold where the shapes allow it (no run-time switch)  */
/* Weight-only e4m3 (16-bit dtypes only): the q, k, v, out-projection, fc1 and fc2 matrices are quantised to OCP e4m3
 * with one fp32 scale per output channel when the weights are loaded and dequantised again before the 16-bit
 * preparation, so the GEMMs multiply 16-bit activations with e4m3-valued weights (SURVEY.md section 7 option (a); the
 * other reading of "fp8", both operands e4m3 on the scaled-MFMA path, is VH_DTYPE_FP8).  Throughput is that of the
 * 16-bit dtype: at batch 512 the weights are 0.3 % of a forward's HBM traffic, which is all a 1-byte copy would save. */
#define VH_FLAG_W8_E4M3 4
/* Class-token tail (opt-in): the logits depend on the LAST layer's class-token row only, so with this flag the last layer runs
 * attention for that one query (all keys / values) and out-proj, fc1, fc2, the final LayerNorm and the head on `batch` rows
 * instead of batch x tokens.  Logits agree with the default path to rounding (not bitwise: the one-query attention is a
 * different kernel); the residual rows of the other tokens are NOT updated by the last layer (vh_debug_read of the hidden
 * state returns them as of the layer before).  Folded 16-bit path only (ignored elsewhere).  Never used by the default bench
 * line: a forward that skips rows is reported as its own `cls_tail` object. */
#define VH_FLAG_CLS_TAIL 8

typedef struct vh_ctx vh_ctx; /* opaque ViT context (device, stream, weights, workspace) */
typedef struct vh_mlp vh_mlp; /* opaque MLP-mode context (the reference's real semantics)  */

/* ---- library / device ---------------------------------------------------------------- */
int vh_abi_version(void);
/* replaces clGetPlatformIDs/clGetDeviceIDs (netFPGA.cpp:371-377) */
int vh_device_count(int* count);
/* last error of the calling thread (ctx may be NULL) or of that context */
const char* vh_last_error(const vh_ctx* ctx);

/* ---- raw device memory helpers (so that bindings need no HIP of their own) ------------ */
int vh_malloc(int device, size_t nbytes, void** dev_ptr);
int vh_free(int device, void* dev_ptr);
int vh_memcpy_h2d(int device, void* dev_dst, const void* host_src, size_t nbytes);
int vh_memcpy_d2h(int device, void* host_dst, const void* dev_src, size_t nbytes);
int vh_device_synchronize(int device);
int vh_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes);   /* hipMemGetInfo */

/* ---- ViT context ---------------------------------------------------------------------- */
/* replaces _init_program + _init_kernel(const char*) (netFPGA.cpp:367-441): device,
 * stream and every device buffer, sized from the net shape. */
int vh_create(const vh_config* cfg, int device, vh_ctx** out);
/* replaces cleanup() + ~net_fpga (netFPGA.cpp:613-651); frees ALL device memory. */
int vh_destroy(vh_ctx* ctx);
int vh_get_config(const vh_ctx* ctx, vh_config* out);
/* *on = 1 when this context folds its LayerNorms into the GEMMs (vh_config.flags, model shape, dtype) */
int vh_get_ln_fold(const vh_ctx* ctx, int* on);
/* the fold's run-time guard (see VH_FLAG_LN_FOLD_*): *max_ratio = the largest |row mean| / row sigma any LayerNorm input
 * row has shown since the weights were loaded (0 when the context never folded), *threshold = the switch point,
 * *tripped = 1 once it was exceeded.  Synchronises the context's stream.  Any pointer may be NULL. */
int vh_get_ln_guard(vh_ctx* ctx, float* max_ratio, float* threshold, int* tripped);
/* VH_DTYPE_FP8 contexts: *max_abs = the largest |x| (an upper bound of it: the root sum of squares of a row's largest
 * 64-column block; exact for layer 0) any residual row has shown since the weights were loaded, *limit = 448 (e4m3).  Beyond
 * the limit the guarded fold switches to the stand-alone LayerNorm (vh_get_ln_guard reports `tripped`).  0 for 16-bit
 * contexts.  Synchronises the context's stream.  Either pointer may be NULL. */
int vh_get_fp8_guard(vh_ctx* ctx, float* max_abs, float* limit);

/* Weight blob = fp32 tensors in canonical order (DESIGN.md "weight blob") preceded by a
 * 64-byte header.  Replaces _load_params (netFPGA.cpp:484-515): uploads, converts to the
 * MFMA operand type, permutes the patch kernel to NHWC order and fuses q|k|v. */
size_t vh_weight_blob_bytes(const vh_config* cfg);
int vh_load_weights(vh_ctx* ctx, const void* host_blob, size_t nbytes);
/* same, blob already resident on this context's GPU (e.g. after an RCCL broadcast) */
int vh_load_weights_device(vh_ctx* ctx, const void* dev_blob, size_t nbytes);
/* deterministic synthetic weights generated on the device (generator: DESIGN.md "synthetic
 * data"); replaces the reference's `random` ctor branch (netFPGA.cpp:82-88) for ViT mode. */
int vh_init_weights_seeded(vh_ctx* ctx, uint64_t seed);
/* canonical fp32 blob currently loaded, copied back to the host / to a device buffer
 * (the inverse direction of the ctor flatten, cf. get_net_data netFPGA.cpp:206-237) */
int vh_export_weights(vh_ctx* ctx, void* host_blob, size_t nbytes);
int vh_export_weights_device(vh_ctx* ctx, void* dev_blob, size_t nbytes);

/* Weight blob on disk (the reference's own attempt at reading weights back, get_net_data
 * netFPGA.cpp:206-237, is a broken TODO and it has no file format).  The file is the canonical blob byte for
 * byte, plus an FNV-1a-64 checksum of the parameter bytes in header words that memory blobs leave zero.
 *   vh_blob_file_config: host only (no device needed) -- validates magic / shape / file length and fills the model
 *                        fields of *cfg (dtype = VH_DTYPE_BF16, max_batch = 1: set what you need before vh_create);
 *   vh_save_weights_file: writes <path>.tmp then renames; vh_load_weights_file: verifies length, shape, checksum. */
int vh_blob_file_config(const char* path, vh_config* cfg);
/* host only: the file as a MEMORY blob (what vh_load_weights takes): length, header and checksum verified, checksum
 * words cleared.  nbytes must equal vh_weight_blob_bytes of the file's configuration. */
int vh_blob_file_read(const char* path, void* host_blob, size_t nbytes);
int vh_save_weights_file(vh_ctx* ctx, const char* path);
int vh_load_weights_file(vh_ctx* ctx, const char* path);

/* The hot path.  Replaces launch_forward's device section (netFPGA.cpp:262-284):
 * in  = batch x image x image x channels fp32, NHWC, host memory;
 * out = batch x classes fp32 logits, host memory.  Synchronous, like the blocking read. */
int vh_forward(vh_ctx* ctx, const float* in_nhwc_host, int batch, float* logits_host);
/* same with both buffers resident in HBM (zero-copy boundary used by bench.py). Enqueues on
 * the context's stream and waits for completion. */
int vh_forward_device(vh_ctx* ctx, const float* in_nhwc_dev, int batch, float* logits_dev);
/* enqueue only; pair with vh_synchronize().  `steps` back-to-back forwards of the same
 * buffers are enqueued (the timed region of bench.py). */
int vh_forward_device_async(vh_ctx* ctx, const float* in_nhwc_dev, int batch,
                            float* logits_dev, int steps);
int vh_synchronize(vh_ctx* ctx);
/* uniform[-1,1) synthetic images written straight into HBM (value range of the reference,
 * def/defines.h:11-12) */
int vh_fill_input_seeded(vh_ctx* ctx, uint64_t seed, int batch, float* in_nhwc_dev);

/* ---- pipelined host path ------------------------------------------------------------------------------------
 * A ring of in-flight batches, modelled on the reference's only asynchronous pattern: the 24-slot ring of
 * filter_image / get_filtered_image (netFPGA.cpp:292-365, ring state :47-56).  vh_ring_submit enqueues
 * H2D copy -> forward -> D2H copy of one batch into the next free slot and returns at once; vh_ring_collect waits
 * for the OLDEST submitted batch and copies its logits out (FIFO).  Copies run on their own streams from pinned
 * staging buffers, so the upload of batch i+1 and the download of batch i-1 overlap the forward of batch i and the
 * host-pointer rate approaches the device-resident rate.  Full ring / empty ring are returned as
 * VH_ERR_RING_FULL / VH_ERR_RING_EMPTY (the reference prints and drops the frame, :330-333, :358-361).
 *   vh_ring_input gives the pinned staging buffer of the slot the next submit will use: fill it in place and
 *   submit with in_nhwc_host = NULL to skip the extra host copy. */
int vh_ring_create(vh_ctx* ctx, int slots, int batch_per_slot);
int vh_ring_destroy(vh_ctx* ctx);
int vh_ring_free_slots(const vh_ctx* ctx, int* n);
int vh_ring_input(vh_ctx* ctx, float** pinned_in_nhwc);
int vh_ring_submit(vh_ctx* ctx, const float* in_nhwc_host, int batch);
int vh_ring_collect(vh_ctx* ctx, float* logits_host, int* batch);

/* hipGraph replay.  With enable != 0 the launch sequence of a forward is captured once per (input pointer, logits
 * pointer, batch) and replayed with hipGraphLaunch; the first forward at a given batch size still runs eagerly.
 * Results are bit-identical.  This is for small batches, which are launch-bound (ViT-B/16, batch 1: ~100 launches);
 * at batch 512 it changes nothing.  Stage timing (vh_set_stage_timing) bypasses the graph.  Environment default:
 * VH_GRAPH=1.  Counterpart in the reference: none (one clEnqueueTask per forward, netFPGA.cpp:275). */
int vh_set_graph(vh_ctx* ctx, int enable);
int vh_get_graph(const vh_ctx* ctx, int* enabled, int* cached_graphs);

/* Concurrency inside one forward: the batch is split into `n` contiguous parts (1..4, default 1, environment
 * VH_STREAMS) that are enqueued on separate streams and joined at the end of every forward.  Images are
 * independent, so the logits are bit-identical for every n; with n = 2 the HBM-bound stages and the partly
 * filled last tile round of one half overlap the MFMA-bound stages of the other (+5 % images/s at ViT-B b512).
 * Off by default: two kernels then share the device, so a per-launch duration no longer measures one kernel. */
int vh_set_streams(vh_ctx* ctx, int n);
int vh_get_streams(const vh_ctx* ctx, int* n);

/* observability: replaces forward_performance / get_forward_performance
 * (netFPGA.cpp:262-264,280-284,603-611).  us = host wall time of the last vh_forward*,
 * kernel_ms = device time between hip events around the last forward's kernels. */
int vh_last_forward_us(const vh_ctx* ctx, int64_t* us);
int vh_last_kernel_ms(vh_ctx* ctx, double* ms);
/* device time of ONE launch of the dominant GEMM kernel class, averaged over the launches
 * of the last forward (hip events on the context's stream); used for bench.py's roofline */
int vh_profile_forward(vh_ctx* ctx, const float* in_nhwc_dev, int batch, float* logits_dev,
                       double* stage_ms, int n_stage_slots, int* n_stages_written);
const char* vh_stage_name(int stage_index);
/* Time every launch of ONE stage (index as in vh_stage_name; -1 = off) with hip events on the
 * context's stream during the following vh_forward_device_async calls, then read the average /
 * minimum launch duration and the number of launches measured. */
int vh_set_stage_timing(vh_ctx* ctx, int stage_index);
int vh_get_stage_timing(vh_ctx* ctx, double* avg_ms, double* min_ms, int* launches);
/* Per-STEP device times of the last vh_forward_device_async call: with step timing enabled the call records one hip
 * event at every step boundary (K + 1 for K steps, on the context's stream); vh_get_step_timing synchronises and writes
 * the first min(*steps, max_steps) step durations in ms (bench.py: median and min beside the mean). */
int vh_set_step_timing(vh_ctx* ctx, int enable);
int vh_get_step_timing(vh_ctx* ctx, double* step_ms, int max_steps, int* steps);

/* debug taps: copy an internal activation of the LAST forward to the host as fp32.
 * what: 0 = residual stream x [batch*T, D] after the last layer run,
 *       1 = final-LN'd CLS rows [batch, D],
 *       2 = one float: how many residual GEMMs of the last forward ran as a split launch (VH_TAIL_OVERLAP=1),
 *       3 = one float: 1 when the last forward kept the MLP hidden activation in its tiled (16-row-blocked) layout -- the
 *           default wherever both MLP GEMMs take the persistent form (16-bit folded path, whole 256-row tiles; VH_H_TILED=0 at
 *           context creation keeps it row-major): same values, same logit bits, fc1's result leaves the registers without an
 *           LDS transposition,
 *       4 = one float: 1 when the last forward kept q|k|v head-major ([3][heads][rows][64]) between the projection's epilogue
 *           and the attention kernel (default where the attention output is tiled; VH_QKV_HM=0 keeps [rows][3 D]; same bits). */
int vh_debug_read(vh_ctx* ctx, int what, float* host_out, size_t n_floats);
/* run only the first `n_layers` encoder layers on the next forwards (-1 = all) */
int vh_debug_set_layers(vh_ctx* ctx, int n_layers);

/* ---- operator-level entry points (device pointers, dtype = VH_DTYPE_*) ----------------- */
/* Each is the kernel the forward uses, exposed so that parity tests and micro-benchmarks
 * can drive it alone.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 * the call returns after the kernel has completed. */
#define VH_EPI_BIAS 0        /* out16[m,n]  = acc + bias[n]                               */
#define VH_EPI_BIAS_GELU 1   /* out16[m,n]  = gelu(acc + bias[n])                          */
#define VH_EPI_BIAS_RESID 2  /* out32[m,n] += acc + bias[n]        (fp32 residual stream)  */
#define VH_EPI_BIAS_F32 3    /* out32[m,n]  = acc + bias[n]                                */
#define VH_EPI_PATCH 4       /* out32[row(m),n] = acc + bias[n] + pos[tok(m),n]            */
#define VH_EPI_LNFOLD 5      /* out16[m,n]  = rstd[m]*(acc - mean[m]*c[n]) + d[n]   (LayerNorm folded: bias = d, aux = c) */
#define VH_EPI_LNFOLD_GELU 6 /* gelu of the above                                          */
#define VH_EPI_RESID_LN 7    /* out32 += acc + bias; out16 = 16-bit copy; partials[N/64][M][2] = row (sum, sumsq) */
#define VH_EPI_RESID_SPLIT 8 /* residual kept as TWO planes, x = hi + lo: (hi, lo) += acc + bias with hi = T(x) (16 bit) and lo = what that
                                rounding dropped, ONE byte per element: e4m3((x - hi) * 32) for bf16, * 256 for fp16 (12 / 15
                                significant bits of x in the pair).  out = hi plane (the next GEMM's A operand), out16 = lo plane
                                [M,N] bytes, partials as RESID_LN.  3 B per element each way instead of 4 B + the 2 B copy of RESID_LN */
#define VH_EPI_PATCH_SPLIT 9 /* the patch embedding written directly as the split residual: row(m) of (hi, lo) = the planes of
                                acc + bias + pos[tok(m)], plus that row's partial sums (partials[N/64][R][2], R = token rows) --
                                EPI_PATCH and the first row-statistics pass in one epilogue.  out = hi, out16 = lo, aux /
                                aux_i as EPI_PATCH; the class-token rows are not touched (vh_op_gemm_ex: R = images x tokens) */
/* out = epilogue(A[M,K] * W[N,K]^T); A and W hold `dtype` elements, K contiguous.
 * aux: EPI_PATCH -> pos-emb fp32 [tokens, N] with aux_i = patches per image.
 * variant: 0 = auto, 1 = 128x128 tile, 2 = 256x256 two-stage, 5 = 256x256 ping-pong, 6 = persistent ping-pong. */
int vh_op_gemm(const void* a16_dev, const void* w16_dev, const float* bias_dev, void* out_dev,
               int64_t M, int N, int K, int epilogue, const float* aux_dev, int aux_i,
               int dtype, int variant, void* stream);

/* fp8 GEMM (VH_DTYPE_FP8 path): a8 [M,K], w8 [N,K] OCP e4m3 bytes, w_scale [N] fp32 (per output channel),
 * out = epilogue(w_scale[n] * sum_k a8[m,k] w8[n,k] + bias[n]);  VH_EPI_BIAS -> bf16, VH_EPI_BIAS_GELU -> e4m3
 * (saturating), VH_EPI_BIAS_RESID (out += ...) / VH_EPI_BIAS_F32 -> fp32.  K % 128 == 0, N % 4 == 0. */
int vh_op_gemm_fp8(const void* a8, const void* w8, const float* w_scale, const float* bias, void* out, int64_t M,
                   int N, int K, int epilogue, int variant /* 0 auto, 5, 7 */, void* stream);
/* the same kernel with the epilogues of the folded-LayerNorm layer loop of the fp8 path (operator tests):
 *   VH_EPI_LNFOLD / VH_EPI_LNFOLD_GELU  out (bf16 / e4m3) = [gelu](rstd[m] * (w_scale[n] * acc - mean[m] * c[n]) + bias[n]);
 *                                       c_dev [N], stats_dev [M][2] = (mean, rstd); N % 256 == 0 for the GELU form
 *   VH_EPI_RESID_LN     out fp32 [M,N] += w_scale * acc + bias; out16 = its e4m3 copy; partials [N/64][M][2]
 *   VH_EPI_RESID_SPLIT  the residual kept as an e4m3 plane (out: hi = e4m3(x), the next GEMM's operand) and a bf16 plane
 *                       (out16: lo = bf16(x - hi)): (hi, lo) += w_scale * acc + bias; partials as above.  N % 256 == 0 */
int vh_op_gemm_fp8_ex(const void* a8, const void* w8, const float* w_scale, const float* bias, void* out, int64_t M,
                      int N, int K, int epilogue, const float* c_dev, const float* stats_dev, void* out16,
                      float* partials, int variant, void* stream);
/* the load-time weight quantiser: s0 = amax(row)/448 (1 for an all-zero row), w8 = rne_e4m3(w / s0),
 * scale[row] = s0 * post_scale */
int vh_op_quantize_rows(const float* w, int rows, int cols, float post_scale, void* w8, float* scale, void* stream);

/* same with the operands of the LayerNorm-folding epilogues: stats [M][2] = (mean, rstd) for LNFOLD*,
 * out16 [M,N] and partials [N/64][M][2] for RESID_LN / RESID_SPLIT (N must be a multiple of 256) */
int vh_op_gemm_ex(const void* a16_dev, const void* w16_dev, const float* bias_dev, void* out_dev,
                  int64_t M, int N, int K, int epilogue, const float* aux_dev, int aux_i,
                  const float* stats_dev, void* out16_dev, float* partials_dev,
                  int dtype, int variant, void* stream);
/* row statistics helpers of the folded LayerNorm:
 *   vh_op_rowstats_cast: x fp32 [rows, dim] -> x16 [rows, dim] (plain cast) and stats [rows][2] = (mean, rstd)
 *   vh_op_finalize_stats: partials [nblk][rows][2] (sum, sumsq over 64-column blocks) -> stats [rows][2] */
int vh_op_rowstats_cast(const float* x_dev, int64_t rows, int dim, float eps, void* x16_dev, float* stats_dev,
                        int dtype, void* stream);
int vh_op_finalize_stats(const float* partials_dev, int nblk, int64_t rows, int dim, float eps, float* stats_dev,
                         void* stream);
/* x fp32 [rows, dim] -> the two planes of the split residual (hi = T(x), 16 bit; lo = one scaled e4m3 byte per element, see
 * VH_EPI_RESID_SPLIT) and stats [rows][2] */
int vh_op_rowstats_split(const float* x_dev, int64_t rows, int dim, float eps, void* hi_dev, void* lo_dev, float* stats_dev,
                         int dtype, void* stream);
/* W'[n,k] = dtype(scale * gamma[k] * W[n,k]); c[n] = sum_k W'[n,k]; d[n] = scale * (sum_k beta[k] W[n,k] + b[n]) */
int vh_op_fold_ln(const float* w_dev, const float* b_dev, const float* gamma_dev, const float* beta_dev, int rows,
                  int dim, float scale, void* w16_dev, float* c_dev, float* d_dev, int dtype, void* stream);
/* y16[r,:] = LN(x[r*row_stride : +dim]) * gamma + beta */
int vh_op_layernorm(const float* x_dev, int64_t rows, int dim, int64_t row_stride,
                    const float* gamma_dev, const float* beta_dev, float eps, void* out16_dev,
                    int dtype, void* stream);
/* qkv16 [batch*tokens, 3*heads*64] -> out16 [batch*tokens, heads*64].  The q columns arrive pre-scaled by
 * VH_ATTN_Q_SCALE = 64^-1/2 * log2(e) (the forward folds it into Wq/bq): the kernel's softmax works in the exp2 domain. */
#define VH_ATTN_Q_SCALE 0.18033688011112042f
int vh_op_attention(const void* qkv16_dev, int batch, int tokens, int heads, void* out16_dev,
                    int dtype, void* stream);
/* NHWC fp32 images -> patch matrix [batch*np, patch*patch*channels] in `dtype` */
int vh_op_im2col(const float* in_nhwc_dev, int batch, int image, int patch, int channels,
                 void* out16_dev, int dtype, void* stream);
/* fp32 -> dtype cast of n elements (n multiple of 4) */
int vh_op_cast(const float* in_dev, void* out16_dev, int64_t n, int dtype, void* stream);
/* synthetic-data generator on the device: kind 0 = uniform[-1,1), 1 = Irwin-Hall(4) * sigma,
 * 2 = constant `sigma` */
int vh_op_fill(float* out_dev, int64_t n, uint64_t seed, uint32_t tensor_id, int kind,
               float sigma, void* stream);

/* micro-benchmark: average device time (ms) of one launch of the GEMM kernel on synthetic operands
 * generated in HBM (uniform[-1,1) activations, sigma=0.02 weights), `iters` launches between events */
int vh_bench_gemm(int device, int64_t M, int N, int K, int epilogue, int dtype, int variant, int iters,
                  double* avg_ms);

/* ---- device group: the N GPUs of one node from ONE process ---------------------------------------------------------
 * The reference drives a single device (clGetDeviceIDs(CL_DEVICE_TYPE_ACCELERATOR), netFPGA.cpp:376) with one input
 * vector per call (:266-277) and has no multi-device code; this is the C-ABI form of the image sharding the north star
 * asks for: one context + one host thread + one stream per device, ncclCommInitAll, ONE ncclBroadcast of the canonical
 * weight blob from member 0 over xGMI (the per-device upload of _load_params, netFPGA.cpp:484-515), contiguous image
 * ranges per member, per-member D2H into the caller's logits buffer; no collective on the data path.  RCCL is loaded at
 * run time and only for groups of more than one distinct device.  Listing one ordinal several times gives a REHEARSAL
 * group on one GPU (same threads, shards and blob path; the broadcast is a device-to-device copy).
 * cfg->max_batch is the per-device capacity.  A group is not re-entrant. */
typedef struct vh_group vh_group;
int vh_group_create(const vh_config* cfg, const int* devices, int n, vh_group** out);
int vh_group_destroy(vh_group* g);
int vh_group_size(const vh_group* g, int* n);
int vh_group_member(vh_group* g, int i, vh_ctx** ctx, int* device); /* the member's own context (borrowed) */
const char* vh_group_last_error(const vh_group* g);
/* image range [lo, hi) of member r for a batch split over n members (the first batch % n members take one extra) */
void vh_group_shard_bounds(int batch, int n, int r, int* lo, int* hi);
/* weights: loaded / generated on member 0, then broadcast; vh_group_broadcast_weights re-sends member 0's resident blob */
int vh_group_load_weights(vh_group* g, const void* host_blob, size_t nbytes);
int vh_group_init_weights_seeded(vh_group* g, uint64_t seed);
int vh_group_broadcast_weights(vh_group* g);
/* the hot path over the group (same buffers and meaning as vh_forward): synchronous, all members run concurrently */
int vh_group_forward(vh_group* g, const float* in_nhwc_host, int batch, float* logits_host);
/* measurement path with the inputs resident in every member's HBM (member i = shard i, seed + i) */
int vh_group_fill_inputs_seeded(vh_group* g, uint64_t seed, int batch_per_device);
int vh_group_forward_resident(vh_group* g, int batch_per_device, int steps);
int vh_group_read_logits(vh_group* g, int batch_per_device, float* logits_host); /* [n * batch_per_device, classes] */

/* ---- filter_image pipeline -------------------------------------------------------------------
 * The reference's second device entry: single-channel 8-bit frames (1080 x 1920, defines.h:31-38) pushed through
 * a kernel `image_process` with a 24-slot ring of in-flight frames (netFPGA.cpp:47-56, 292-365).  That kernel's
 * source and bitstream are absent, so WHAT it computed is unknown; the ring mechanics are reproduced and the
 * arithmetic is a documented choice: a 3x3 filter with replicated borders, integer arithmetic, bit-exact against
 * oracle_filter3x3.  Frames are `height * width` bytes, row-major. */
#define VH_FILTER_BLUR3 0   /* (1 2 1 / 2 4 2 / 1 2 1) / 16, rounded half up */
#define VH_FILTER_SOBEL3 1  /* min(255, |gx| + |gy|) */
typedef struct vh_filter vh_filter;
int vh_filter_create(int device, int height, int width, int slots, int kind, vh_filter** out);
int vh_filter_destroy(vh_filter* f);
int vh_filter_free_slots(const vh_filter* f, int* n);
int vh_filter_submit(vh_filter* f, const uint8_t* frame);   /* VH_ERR_RING_FULL when every slot is in flight */
int vh_filter_collect(vh_filter* f, uint8_t* frame);        /* oldest frame; VH_ERR_RING_EMPTY when none */
const char* vh_filter_last_error(const vh_filter* f);

/* ---- MLP mode: the reference's actual launch_forward semantics -------------------------- */
/* y_l = act(W_l y_{l-1} + b_l), weights row-major [n_out, n_in] per layer, layers and biases
 * concatenated exactly as the ctor flattens them (netFPGA.cpp:68-76, 91-106); kernel
 * argument list of network_v1 (netFPGA.cpp:427-436, 499-502). */
int vh_mlp_create(int device, int n_ins, int n_layers, const int* n_p_l, int activation,
                  vh_mlp** out);
int vh_mlp_load_params(vh_mlp* mlp, const float* params_host, size_t n_params,
                       const float* bias_host, size_t n_neurons);
/* n_vec input vectors of n_ins floats -> n_vec output vectors of n_p_l[n_layers-1] floats */
int vh_mlp_forward(vh_mlp* mlp, const float* inputs_host, int n_vec, float* outputs_host);
int vh_mlp_last_forward_us(const vh_mlp* mlp, int64_t* us);
/* Training of the dense chain: the device side of init_gradient / launch_gradient (netAbstract.h:14-15).  PARITY
 * UNPINNED: the reference's bodies are commented-out code (netFPGA.cpp:518-580) on a vector library that is not in the
 * repository.  Their loop shape is kept -- per iteration every set is back-propagated, its output error summed in
 * absolute value, the sets' gradients accumulated, normalised, applied, the accumulator reset (:552-565) -- and what
 * they leave undefined is fixed here:
 *   loss of a set       1/2 |a_L - t|^2   (delta of the last layer = (a_L - t) * act'(z_L))
 *   normalize_1         mean over the sets
 *   update              p -= multiplier * mean gradient, all layers from the deltas of the SAME parameters
 *   errors[it]          sum over sets and outputs of |a_L - t|, taken BEFORE the update of iteration `it`
 *   error_threshold     an iteration with errors[it] <= error_threshold ends the loop; later entries stay 0 (the value
 *                       the reference initialises its result with, :550)
 *   act'                identity 1; RELU2 1 on (0, 1); RELU 1 on z > 0; HARDTANH 1 on (-1, 1); GELU Phi(z) + z phi(z)
 * set_ins [n_sets][n_ins], set_outs [n_sets][n_p_l[n_layers-1]] (host); at most 65535 sets and neurons per layer.
 * vh_mlp_read_params copies the (trained) parameters back in the layout vh_mlp_load_params takes. */
int vh_mlp_init_gradient(vh_mlp* mlp, const float* set_ins_host, const float* set_outs_host, int n_sets);
int vh_mlp_launch_gradient(vh_mlp* mlp, int iterations, float error_threshold, float multiplier, float* errors_host);
int vh_mlp_read_params(vh_mlp* mlp, float* params_host, size_t n_params, float* bias_host, size_t n_neurons);
int vh_mlp_last_gradient_us(const vh_mlp* mlp, int64_t* us);
const char* vh_mlp_last_error(const vh_mlp* mlp);
int vh_mlp_destroy(vh_mlp* mlp);

#ifdef __cplusplus
}
#endif
#endif /* VITHIP_H */
