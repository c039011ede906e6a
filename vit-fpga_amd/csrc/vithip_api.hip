// vithip_api.hip — the C ABI of libvithip.so (declared in include/vithip.h).
//
// Host-side runtime of the hot path: device/stream ownership, the HBM layout (canonical fp32
// weight blob + 16-bit compute copies + one activation arena sized for max_batch), weight
// preparation, and the launch sequence of one ViT forward.  Takes over the jobs of the
// reference's _init_program/_init_kernel/_load_params/launch_forward/cleanup
// (/root/reference/src/netFPGA.cpp:239-290, 367-515, 639-651) — see the per-function notes in
// the header.  No CPU fallback exists: without a gfx950 device every compute entry point
// fails with VH_ERR_NO_DEVICE / VH_ERR_HIP.
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "vh_kernels.h"

using namespace vh;

namespace {

thread_local std::string g_err;

int fail(std::string* ctx_err, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    if (ctx_err) *ctx_err = buf;
    return code;
}

#define HIPCHK(ctxerr, expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(ctxerr, VH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

struct BlobHeader {
    char magic[8];
    int32_t image_size, patch_size, channels, dim, heads, mlp_dim, layers, classes;
    float ln_eps;
    // on-disk files only (vh_save_weights_file): FNV-1a 64 of the parameter bytes, flags bit 0 = checksum present.
    // Blobs built in memory leave all five words 0.
    uint32_t sum_lo, sum_hi, flags, pad[2];
};
static_assert(sizeof(BlobHeader) == 64, "blob header is 64 bytes");

uint64_t fnv1a64(const void* data, size_t n, uint64_t h = 0xCBF29CE484222325ull) {
    const unsigned char* p = (const unsigned char*)data;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001B3ull; }
    return h;
}

struct LayerOff {  // offsets in floats from the start of the parameter region
    size_t ln1w, ln1b, qw, qb, kw, kb, vw, vb, ow, ob, ln2w, ln2b, f1w, f1b, f2w, f2b;
};

struct Layout {
    int T, NP, KP;
    size_t patch_w, patch_b, cls, pos, lnfw, lnfb, headw, headb, total;
    std::vector<LayerOff> layer;
};

Layout make_layout(const vh_config& c) {
    Layout L;
    const size_t D = c.dim, M = c.mlp_dim, C = c.classes;
    const int g = c.image_size / c.patch_size;
    L.NP = g * g;
    L.T = L.NP + 1;
    L.KP = c.patch_size * c.patch_size * c.channels;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += n; return r; };
    L.patch_w = take(D * L.KP); L.patch_b = take(D); L.cls = take(D); L.pos = take((size_t)L.T * D);
    L.layer.resize(c.layers);
    for (int l = 0; l < c.layers; ++l) {
        LayerOff& p = L.layer[l];
        p.ln1w = take(D); p.ln1b = take(D);
        p.qw = take(D * D); p.qb = take(D); p.kw = take(D * D); p.kb = take(D);
        p.vw = take(D * D); p.vb = take(D); p.ow = take(D * D); p.ob = take(D);
        p.ln2w = take(D); p.ln2b = take(D);
        p.f1w = take(M * D); p.f1b = take(M); p.f2w = take(D * M); p.f2b = take(D);
    }
    L.lnfw = take(D); L.lnfb = take(D); L.headw = take(C * D); L.headb = take(C);
    L.total = o;
    return L;
}

// expected blob size of a configuration computed WITHOUT building the layout (no allocation): used on untrusted headers
size_t blob_bytes_of(const vh_config& c) {
    const size_t D = c.dim, M = c.mlp_dim, C = c.classes, g = (size_t)(c.image_size / c.patch_size);
    const size_t T = g * g + 1, KP = (size_t)c.patch_size * c.patch_size * c.channels;
    const size_t per_layer = 2 * D + 4 * (D * D + D) + 2 * D + (M * D + M) + (D * M + D);
    return sizeof(BlobHeader) + 4 * (D * KP + D + D + T * D + (size_t)c.layers * per_layer + 2 * D + C * D + C);
}

const char* check_config(const vh_config& c) {
    if (c.image_size <= 0 || c.patch_size <= 0 || c.image_size % c.patch_size) return "image_size must be a positive multiple of patch_size";
    if (c.channels <= 0 || (c.patch_size * c.channels) % 4) return "patch_size*channels must be a multiple of 4";
    if ((c.patch_size * c.patch_size * c.channels) % 64) return "patch_size^2*channels must be a multiple of 64";
    if (c.dim <= 0 || c.dim % 64) return "dim must be a multiple of 64";
    if (c.heads <= 0 || c.dim != c.heads * 64) return "dim/heads must be 64";
    if (c.mlp_dim <= 0 || c.mlp_dim % 64) return "mlp_dim must be a multiple of 64";
    if (c.dim > 2048) return "dim > 2048 unsupported";
    if (c.layers <= 0) return "layers must be positive";
    if (c.classes <= 0 || c.classes % 4) return "classes must be a positive multiple of 4";
    if (c.dtype != VH_DTYPE_BF16 && c.dtype != VH_DTYPE_FP16 && c.dtype != VH_DTYPE_FP8) return "dtype must be VH_DTYPE_BF16, VH_DTYPE_FP16 or VH_DTYPE_FP8";
    if (c.dtype == VH_DTYPE_FP8 && (c.dim % 128 || c.mlp_dim % 128)) return "VH_DTYPE_FP8 needs dim and mlp_dim to be multiples of 128";
    if (c.max_batch <= 0) return "max_batch must be positive";
    if (!(c.ln_eps > 0.f)) return "ln_eps must be positive";
    if (c.flags & ~(VH_FLAG_LN_FOLD_OFF | VH_FLAG_LN_FOLD_ON | VH_FLAG_W8_E4M3 | VH_FLAG_CLS_TAIL)) return "unknown bits in flags";
    if ((c.flags & VH_FLAG_W8_E4M3) && c.dtype == VH_DTYPE_FP8) return "flags: VH_FLAG_W8_E4M3 is for the 16-bit dtypes (VH_DTYPE_FP8 quantises both operands)";
    if ((c.flags & VH_FLAG_W8_E4M3) && (c.dim % 4 || c.mlp_dim % 4)) return "flags: VH_FLAG_W8_E4M3 needs dim and mlp_dim multiples of 4";
    if ((c.flags & VH_FLAG_LN_FOLD_OFF) && (c.flags & VH_FLAG_LN_FOLD_ON)) return "flags: VH_FLAG_LN_FOLD_OFF and VH_FLAG_LN_FOLD_ON exclude each other";
    // bounds that keep every size computation below far from overflow (and a crafted file header from driving an
    // allocation: vh_blob_file_config feeds this function)
    if (c.image_size > 4096 || c.patch_size > 256 || c.channels > 64) return "image_size <= 4096, patch_size <= 256, channels <= 64";
    if (c.layers > 4096 || c.classes > (1 << 20) || c.mlp_dim > (1 << 16) || c.max_batch > (1 << 20)) return "layers <= 4096, classes <= 2^20, mlp_dim <= 2^16, max_batch <= 2^20";
    const int g = c.image_size / c.patch_size;
    if (attention_lds_bytes(g * g + 1) > 160 * 1024) return "token count too large for the LDS-resident attention kernel";
    return nullptr;
}

enum Stage { ST_IM2COL, ST_PATCH, ST_CLS, ST_LN, ST_QKV, ST_ATTN, ST_PROJ, ST_FC1, ST_FC2, ST_LNF, ST_HEAD, ST_LNSTATS, ST_COUNT };
const char* kStageNames[ST_COUNT] = {"im2col", "patch_gemm", "cls_rows", "layernorm", "qkv_gemm", "attention",
                                     "proj_gemm", "fc1_gemm", "fc2_gemm", "final_layernorm", "head_gemm", "ln_stats"};

}  // namespace

struct vh_ctx {
    vh_config cfg;
    int device;
    Layout L;
    hipStream_t stream = nullptr;
    // optional extra streams: the batch is split into `nstreams` contiguous parts that run concurrently
    static constexpr int kMaxStreams = 4;
    hipStream_t xstream[kMaxStreams - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxStreams - 1] = {nullptr, nullptr, nullptr};
    int nstreams = 1;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool weights_ready = false;
    int run_layers = -1;
    // HBM: canonical blob (header + fp32 params) and the 16-bit compute copies
    char* blob = nullptr;
    float* params = nullptr;  // blob + 64
    char* w16 = nullptr;      // arena of 16-bit matrices
    void* wp16 = nullptr;     // [D, KP]
    std::vector<void*> wqkv16, wo16, w1_16, w2_16;
    // fc2's weights once more in the 16-row-blocked layout of the tiled hidden activation (h_tiled; 16-bit folded path): the MLP
    // hidden activation then leaves fc1's epilogue straight from the registers (gemm_epilogue.h OTILED) and fc2's operand DMA reads
    // both operands in that layout -- same values, same k order, same bits as the row-major path
    std::vector<void*> w2t_16, wot_16;   // (+ the out-projection's: attention writes its output tiled as well, att_tiled below)
    bool h_tiled = false;     // VH_H_TILED=0 (read when the context is created) keeps the row-major h (A/B, tests)
    bool att_tiled = true;    // VH_ATT_TILED=0 (read when the context is created) keeps the attention output row-major
    bool qkv_hm = true;       // VH_QKV_HM=0 keeps q|k|v row-major ([rows][3 D]) instead of head-major ([3][heads][rows][64]; needs att_tiled)
    bool weights_ready_tiled = false;   // the tiled copies exist for the CURRENT weights (prepared with the fold on)
    float* bqkv = nullptr;    // [layers, 3D]
    // VH_DTYPE_FP8: the four per-layer matrices hold e4m3 bytes (in the same arena) + one fp32 scale per output channel;
    // everything 16-bit (patch embedding, qkv, head) is bf16
    bool fp8 = false;
    int dt16 = VH_DTYPE_BF16;
    std::vector<float*> sqkv, so, s1, s2;
    // LayerNorm folded into the q|k|v and fc1 GEMMs (dim and mlp_dim multiples of 256):
    bool ln_fold = false;
    float* fold_cd = nullptr; // per layer: cqkv[3D] dqkv[3D] c1[M] d1[M]
    float* stats = nullptr;   // [B*T][2] (mean, rstd) of the current residual rows
    float* partials = nullptr;// [D/64][B*T][2]
    // with the fold: the residual stream lives as two planes, x = hi + lo (hi = the xn16 buffer = the GEMMs' 16-bit A operand,
    // lo = xlo16 = what that rounding dropped, one scaled e4m3 byte per element: vh_common.h Lo8): a residual GEMM then moves
    // 3 B per element each way instead of the fp32 array plus its 16-bit copy.
    // VH_RESID_SPLIT=0 (A/B tools) keeps the fp32 array.
    bool split = false;
    bool cls_tail = false;    // VH_FLAG_CLS_TAIL: the last layer computes the class-token rows only (folded 16-bit path)
    bool patch_fused = false; // VH_PATCH_FUSED=1 (read when the context is created): patch gather inside the GEMM's A loader
    void* xlo16 = nullptr;    // [B*T, D] bytes
    // Run-time guard on the fold: the kernels that produce the row statistics keep a running maximum of
    // |mean| * rstd over the REAL rows (guard_dev: the bits of a non-negative float); every forward ends with an
    // asynchronous copy of that word into pinned host memory, and every forward entry point looks at the copy before it
    // enqueues anything (no synchronisation: the value is the one of the most recently COMPLETED forward).  Beyond
    // guard_thresh the folded operand's rounding error leaves the budget of the 1e-3 tolerance; a context whose flags
    // left the choice to the library (guard_auto) then switches to the stand-alone LayerNorm for good -- the weights are
    // prepared again in the plain layout -- and the synchronous vh_forward repeats the forward that tripped it.
    bool ln_fold_cfg = false, split_cfg = false;   // what configuration + flags chose at creation (restored by a weight load)
    unsigned int* guard_dev = nullptr;
    float* guard_host = nullptr;
    float guard_thresh = 0.5f;
    bool guard_auto = false, guard_tripped = false;
    // activations (sized for max_batch)
    char* arena = nullptr;
    unsigned int* tickets = nullptr;   // [max_batch][layers] work-queue counters of the attention kernel (launch_attention)
    float* x = nullptr;       // residual stream [B*T, D] fp32
    void* xn16 = nullptr;     // LN output       [B*T, D]
    void* qkv16 = nullptr;    //                 [B*T, 3D]
    void* att16 = nullptr;    //                 [B*T, D]
    void* h16 = nullptr;      //                 [B*T, M]
    void* col16 = nullptr;    // patch matrix    [B*NP, KP]
    float* clsn32 = nullptr;  // final-LN'd CLS  [B, D] fp32 (the head runs in fp32 on the blob's own weights)
    float* in_dev = nullptr;  // staging for the host-pointer forward
    float* logits_dev = nullptr;
    int64_t last_us = 0;
    bool timed = false;
    int last_batch = 0;
    // pipelined host path (vh_ring_*): slots of pinned host staging + device buffers, copies on their own streams
    struct RingSlot {
        float *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        hipEvent_t in_done = nullptr, fwd_done = nullptr, out_done = nullptr;
        int batch = 0;
    };
    std::vector<RingSlot> ring;
    hipStream_t copy_in = nullptr, copy_out = nullptr;
    int ring_batch = 0, ring_wr = 0, ring_rd = 0, ring_used = 0;
    // tail overlap (enqueue_forward, resid_gemm_ln): helper stream + events, CU count; VH_TAIL_OVERLAP=1 enables
    hipStream_t tstream = nullptr;
    hipEvent_t ev_tail_a = nullptr, ev_tail_l = nullptr;
    bool tail_overlap = false;
    bool last_h_tiled = false;   // debug tap 3
    bool last_qkv_hm = false;    // debug tap 4
    int tail_splits = 0;   // residual GEMMs of the last forward that were launched as [full rounds] + [tail round] (debug tap 2)
    int num_cu = 256;
    // optional hipGraph replay of the forward's launch sequence (vh_set_graph): one instantiated graph per
    // (input pointer, logits pointer, batch); a batch size runs eagerly once before it is captured
    bool use_graph = false;
    struct GraphEntry { const float* in; float* out; int batch; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<GraphEntry> graphs;
    std::vector<int> graph_warm;   // batch sizes that have run eagerly (kernel attributes are set)
    // optional per-launch timing of ONE stage inside the timed region (bench.py's roofline)
    int timing_stage = -1;
    std::vector<hipEvent_t> tev;  // pool: pairs (start, stop)
    size_t tev_used = 0;
    // optional per-STEP events of the last vh_forward_device_async call (vh_set_step_timing): K + 1 boundaries
    bool step_timing = false;
    std::vector<hipEvent_t> sev;
    size_t sev_used = 0;
    std::string err;
};

// filter_image pipeline: a ring of in-flight 8-bit frames (reference: 24 slots, netFPGA.cpp:47-56, 292-365)
struct vh_filter {
    int device, height, width, kind;
    struct Slot {
        uint8_t *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        hipEvent_t in_done = nullptr, k_done = nullptr, out_done = nullptr;
    };
    std::vector<Slot> slot;
    hipStream_t copy_in = nullptr, compute = nullptr, copy_out = nullptr;
    int wr = 0, rd = 0, used = 0;
    std::string err;
};

struct vh_mlp {
    int device, n_ins, n_layers, activation;
    std::vector<int> npl;
    size_t n_params = 0, n_neurons = 0;
    int widest = 0, max_vec = 0;
    hipStream_t stream = nullptr;
    float *params = nullptr, *bias = nullptr, *buf0 = nullptr, *buf1 = nullptr;
    bool loaded = false;
    int64_t last_us = 0;
    // training (vh_mlp_init_gradient / vh_mlp_launch_gradient): the sets and, per set and neuron, pre-activation,
    // activation and delta
    int n_sets = 0;
    float *set_ins = nullptr, *set_outs = nullptr, *tz = nullptr, *ta = nullptr, *td = nullptr, *terr = nullptr;
    int64_t last_gradient_us = 0;
    std::string err;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int set_device(std::string* err, int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(err, VH_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(err, VH_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIPCHK(err, hipSetDevice(device));
    return VH_OK;
}

// convert the fp32 blob resident in ctx->blob into the compute layout
int prepare_weights(vh_ctx* c) {
    const vh_config& f = c->cfg;
    const Layout& L = c->L;
    const int D = f.dim, M = f.mlp_dim;
    hipStream_t s = c->stream;
    const float* P = c->params;
    HIPCHK(&c->err, launch_permute_patch(P + L.patch_w, D, f.channels, f.patch_size, c->wp16, c->dt16, s));
    for (int l = 0; l < f.layers && c->fp8 && c->ln_fold; ++l) {
        // folded LayerNorm on e4m3 operands: W' = gamma o W through the row quantiser, its scales, c and d (launch_fold_ln_f8)
        const LayerOff& o = L.layer[l];
        float* cd = c->fold_cd + (size_t)l * (6 * D + 2 * M);
        char* wq = (char*)c->wqkv16[l];
        const size_t dd = (size_t)D * D;
        HIPCHK(&c->err, launch_fold_ln_f8(P + o.qw, P + o.qb, P + o.ln1w, P + o.ln1b, D, D, kAttnQScale, wq, c->sqkv[l], cd, cd + 3 * D, s));
        HIPCHK(&c->err, launch_fold_ln_f8(P + o.kw, P + o.kb, P + o.ln1w, P + o.ln1b, D, D, 1.0f, wq + dd, c->sqkv[l] + D, cd + D, cd + 4 * D, s));
        HIPCHK(&c->err, launch_fold_ln_f8(P + o.vw, P + o.vb, P + o.ln1w, P + o.ln1b, D, D, 1.0f, wq + 2 * dd, c->sqkv[l] + 2 * D, cd + 2 * D, cd + 5 * D, s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.ow, D, D, 1.0f, c->wo16[l], c->so[l], s));
        if (c->h_tiled && c->wot_16[l]) HIPCHK(&c->err, launch_tile_bytes(c->wo16[l], D, D, c->wot_16[l], s));   // (att_tiled)
        HIPCHK(&c->err, launch_fold_ln_f8(P + o.f1w, P + o.f1b, P + o.ln2w, P + o.ln2b, M, D, 1.0f, c->w1_16[l], c->s1[l], cd + 6 * D, cd + 6 * D + M, s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.f2w, D, M, 1.0f, c->w2_16[l], c->s2[l], s));
        if (c->h_tiled && c->w2t_16[l]) HIPCHK(&c->err, launch_tile_bytes(c->w2_16[l], D, M, c->w2t_16[l], s));   // the same bytes, tiled (h_tiled)
    }
    for (int l = 0; l < f.layers && c->fp8 && !c->ln_fold; ++l) {
        const LayerOff& o = L.layer[l];
        // bias [bq/8 ; bk ; bv] from the 16-bit packer (its 16-bit matrix is overwritten right after), then
        // e4m3 rows + scales; the softmax scale 64^-1/2 * log2(e) (kAttnQScale) goes into the q rows' fp32 scales
        HIPCHK(&c->err, launch_pack_qkv(P + o.qw, P + o.qb, P + o.kw, P + o.kb, P + o.vw, P + o.vb, D, kAttnQScale,
                                        c->wqkv16[l], c->bqkv + (size_t)l * 3 * D, c->dt16, s));
        char* wq = (char*)c->wqkv16[l];
        const size_t dd = (size_t)D * D;
        HIPCHK(&c->err, launch_quantize_rows(P + o.qw, D, D, kAttnQScale, wq, c->sqkv[l], s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.kw, D, D, 1.0f, wq + dd, c->sqkv[l] + D, s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.vw, D, D, 1.0f, wq + 2 * dd, c->sqkv[l] + 2 * D, s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.ow, D, D, 1.0f, c->wo16[l], c->so[l], s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.f1w, M, D, 1.0f, c->w1_16[l], c->s1[l], s));
        HIPCHK(&c->err, launch_quantize_rows(P + o.f2w, D, M, 1.0f, c->w2_16[l], c->s2[l], s));
    }
    // VH_FLAG_W8_E4M3: the six matrices of a layer pass through the e4m3 quantiser (one scale per output channel) and back
    // before the 16-bit preparation -- weight-only fp8 with dequantisation at load (SURVEY.md section 7 option (a))
    float* w8tmp = nullptr;
    struct FreeOnExit { float*& p; ~FreeOnExit() { if (p) { hipFree(p); p = nullptr; } } } w8tmp_guard{w8tmp};   // every early return below frees the scratch
    const size_t dd1 = (size_t)D * D, md1 = (size_t)M * D;
    if (!c->fp8 && (f.flags & VH_FLAG_W8_E4M3)) { HIPCHK(&c->err, hipMalloc((void**)&w8tmp, (4 * dd1 + 2 * md1) * sizeof(float))); }
    for (int l = 0; l < f.layers && !c->fp8; ++l) {
        LayerOff o = L.layer[l];
        const float* P = c->params;
        if (w8tmp) {
            const size_t src[6] = {o.qw, o.kw, o.vw, o.ow, o.f1w, o.f2w};
            const int rows[6] = {D, D, D, D, M, D}, cols[6] = {D, D, D, D, D, M};
            size_t at = 0;
            for (int i = 0; i < 6; ++i) {
                HIPCHK(&c->err, launch_fake_quant_rows(c->params + src[i], rows[i], cols[i], w8tmp + at, s));
                at += (size_t)rows[i] * cols[i];
            }
            // the matrix offsets now point into w8tmp (relative to P = w8tmp - 0): rebase through pointer arithmetic
            P = w8tmp;
            o.qw = 0; o.kw = dd1; o.vw = 2 * dd1; o.ow = 3 * dd1; o.f1w = 4 * dd1; o.f2w = 4 * dd1 + md1;
        }
        const float* B = c->params;   // biases and LayerNorm parameters stay where they are
        if (c->ln_fold) {
            // W' = gamma o W (q rows also carry the softmax scale kAttnQScale), c = row sums of W', d = beta.W + b
            float* cd = c->fold_cd + (size_t)l * (6 * D + 2 * M);
            char* wq = (char*)c->wqkv16[l];
            const size_t dd2 = (size_t)D * D * 2;
            HIPCHK(&c->err, launch_fold_ln(P + o.qw, B + o.qb, B + o.ln1w, B + o.ln1b, D, D, kAttnQScale, wq, cd, cd + 3 * D, f.dtype, s));
            HIPCHK(&c->err, launch_fold_ln(P + o.kw, B + o.kb, B + o.ln1w, B + o.ln1b, D, D, 1.0f, wq + dd2, cd + D, cd + 4 * D, f.dtype, s));
            HIPCHK(&c->err, launch_fold_ln(P + o.vw, B + o.vb, B + o.ln1w, B + o.ln1b, D, D, 1.0f, wq + 2 * dd2, cd + 2 * D, cd + 5 * D, f.dtype, s));
            HIPCHK(&c->err, launch_fold_ln(P + o.f1w, B + o.f1b, B + o.ln2w, B + o.ln2b, M, D, 1.0f, c->w1_16[l], cd + 6 * D, cd + 6 * D + M, f.dtype, s));
        } else {
            HIPCHK(&c->err, launch_pack_qkv(P + o.qw, B + o.qb, P + o.kw, B + o.kb, P + o.vw, B + o.vb, D, kAttnQScale,
                                            c->wqkv16[l], c->bqkv + (size_t)l * 3 * D, f.dtype, s));
            HIPCHK(&c->err, launch_cast(P + o.f1w, c->w1_16[l], (int64_t)M * D, f.dtype, s));
        }
        HIPCHK(&c->err, launch_cast(P + o.ow, c->wo16[l], (int64_t)D * D, f.dtype, s));
        HIPCHK(&c->err, launch_cast(P + o.f2w, c->w2_16[l], (int64_t)D * M, f.dtype, s));
        if (c->h_tiled && c->w2t_16[l]) HIPCHK(&c->err, launch_cast_tiled_w(P + o.f2w, D, M, c->w2t_16[l], f.dtype, s));
        if (c->h_tiled && c->wot_16[l]) HIPCHK(&c->err, launch_cast_tiled_w(P + o.ow, D, D, c->wot_16[l], f.dtype, s));
        if (w8tmp) { HIPCHK(&c->err, hipStreamSynchronize(s)); }   // the scratch is reused by the next layer
    }
    HIPCHK(&c->err, hipStreamSynchronize(s));
    c->weights_ready = true;
    c->weights_ready_tiled = c->h_tiled && (!c->fp8 || c->ln_fold);   // (e4m3 operands: prepared in the folded branch only)
    return VH_OK;
}

// new weights: the guard starts over and the path is the configured one again
void drop_graphs(vh_ctx* c);
int guard_reset(vh_ctx* c) {
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));   // no forward in flight still writes the words
    HIPCHK(&c->err, hipMemsetAsync(c->guard_dev, 0, 2 * sizeof(unsigned int), c->stream));
    c->guard_host[0] = c->guard_host[1] = 0.f;
    c->guard_tripped = false;
    c->ln_fold = c->ln_fold_cfg;
    c->split = c->split_cfg;
    // Captured launch sequences belong to the path (and the weight layout) they were captured on: a context whose guard had
    // tripped holds graphs of the stand-alone-LayerNorm sequence, and replaying those against re-folded weights would be
    // silently wrong.  New weights, new graphs.
    drop_graphs(c);
    return VH_OK;
}

int check_blob_header(vh_ctx* c, const BlobHeader& h) {
    const vh_config& f = c->cfg;
    if (memcmp(h.magic, "VHBLOB1", 8) != 0) return fail(&c->err, VH_ERR_INVALID, "weight blob: bad magic");
    if (h.image_size != f.image_size || h.patch_size != f.patch_size || h.channels != f.channels || h.dim != f.dim ||
        h.heads != f.heads || h.mlp_dim != f.mlp_dim || h.layers != f.layers || h.classes != f.classes)
        return fail(&c->err, VH_ERR_INVALID, "weight blob: shape differs from the context's vh_config");
    return VH_OK;
}

// the launch sequence of ONE forward; `ev` (optional) receives an event after every stage
// `img0`: first image of this part inside the activation arena (a batch can be split into parts that run on
// different streams: rows of different images never interact), `s`: the stream to enqueue on.
int enqueue_forward(vh_ctx* c, const float* in, int batch, float* logits, std::vector<std::pair<int, hipEvent_t>>* ev,
                    hipStream_t s, int img0, bool allow_tail = false, bool may_pad = true) {
    const vh_config& f = c->cfg;
    const Layout& L = c->L;
    const int D = f.dim, M = f.mlp_dim, T = L.T;
    const int64_t rows = (int64_t)batch * T;
    // Row count of the per-token GEMMs of the folded layer loop: rounded up to whole 256-row tiles when nothing lives behind
    // this part's rows in the arena (a single part, or the last of several; the arena itself is padded).  The persistent
    // GEMM form runs full tiles only, and a ragged last row of tiles would send the launch to the one-tile form.  The
    // padding rows hold whatever the arena holds: rows never mix in a GEMM, the attention kernel and the head read real
    // rows only, so nothing of them reaches a logit.
    // Worth it from about two full rounds of tiles per GEMM on (measured, ViT-B bf16: batch 300 +2.5 %, batch 128 -2 %:
    // with fewer tiles the one-tile form's dynamic placement wins).
    const int64_t row_tiles = (rows + 255) / 256;
    const int64_t rows_g = (may_pad && c->ln_fold && row_tiles * ((D + 255) / 256) >= 2 * (int64_t)c->num_cu) ? row_tiles * 256 : rows;
    const float* P = c->params;
    // this part's slices of the arena
    const size_t r0 = (size_t)img0 * T, esz = 2, esz_op = c->fp8 ? 1 : 2;  // esz_op: GEMM A-operand element
    const int dt16 = c->dt16;
    float* const x = c->x + r0 * D;
    // LayerNorm-fold row statistics of this part: (mean, rstd) per row, and the [D/64][rows][2] partial sums; parts take
    // disjoint slices (a part's partial-sum block is (D/64) * rows * 2 floats, laid out for ITS row count)
    float* const stats_p = c->stats + r0 * 2;
    float* const partials_p = c->partials + (size_t)(D / 64) * r0 * 2;
    char* const xn16 = (char*)c->xn16 + r0 * D * esz_op;
    char* const xlo16 = (char*)c->xlo16 + r0 * D * (c->fp8 ? 2 : 1);   // the lo plane: one byte per element (fp8 path: bf16 beside the e4m3 hi plane)
    char* const qkv16 = (char*)c->qkv16 + r0 * 3 * D * esz;
    char* const att16 = (char*)c->att16 + r0 * D * esz_op;
    char* const h16 = (char*)c->h16 + r0 * M * esz_op;
    char* const col16 = (char*)c->col16 + (size_t)img0 * L.NP * L.KP * esz;
    float* const clsn32 = c->clsn32 + (size_t)img0 * D;
    // the attention launches' work-queue counters: one word per layer, all zeroed by ONE memset per forward (a memset per
    // launch is a 5 us fill kernel in front of every attention kernel)
    unsigned int* const tickets_part = c->tickets + (size_t)img0 * (f.layers > 0 ? f.layers : 1);
    if (f.layers > 0) HIPCHK(&c->err, hipMemsetAsync(tickets_part, 0, sizeof(unsigned int) * f.layers, s));
    // fp8 operands: the folded GEMMs multiply the RAW residual rows as e4m3, which saturates at 448 -- the statistics kernels
    // keep a running maximum of |x| (a bound of it) in the guard's second word
    unsigned int* const amax_guard = c->fp8 ? c->guard_dev + 1 : nullptr;
    auto mark = [&](int stage) -> int {
        if (!ev) return VH_OK;
        hipEvent_t e;
        HIPCHK(&c->err, hipEventCreate(&e));
        HIPCHK(&c->err, hipEventRecord(e, s));
        ev->push_back({stage, e});
        return VH_OK;
    };
    // `wscale` (fp8 operands only): the weight matrix's per-output-channel scales
    auto gemm = [&](const void* a, const void* w, const float* bias, void* out, int64_t Mr, int N, int K, int epi,
                    const float* aux, int aux_i, const float* wscale = nullptr) {
        GemmArgs g{a, w, bias, out, Mr, N, K, epi, aux, aux_i, dt16, 0};
        g.stats = stats_p;         // read by LNFOLD*, ignored otherwise
        g.out16 = epi == VH_EPI_RESID_SPLIT ? xlo16 : xn16;   // RESID_LN: the 16-bit (fp8: e4m3) copy; RESID_SPLIT: the lo plane
        g.partials = partials_p;
        if (wscale) {              // e4m3 operands (folded-LN layer loop of the fp8 path)
            g.dtype = VH_DTYPE_FP8;
            if (epi == VH_EPI_LNFOLD || epi == VH_EPI_LNFOLD_GELU) g.wscale = wscale;   // `aux` carries c_n there
            else g.aux = wscale;
            return launch_gemm_fp8(g, s);
        }
        return launch_gemm(g, s);
    };
    // hip events around every launch of the stage selected by vh_set_stage_timing()
    auto tmark = [&](int stage) -> int {
        if (stage != c->timing_stage) return VH_OK;
        if (c->tev_used == c->tev.size()) {
            hipEvent_t e;
            HIPCHK(&c->err, hipEventCreate(&e));
            c->tev.push_back(e);
        }
        HIPCHK(&c->err, hipEventRecord(c->tev[c->tev_used++], s));
        return VH_OK;
    };
    int rc;
    if (img0 == 0) c->tail_splits = 0;
    if ((rc = mark(-1))) return rc;
    const int nl = (c->run_layers < 0 || c->run_layers > f.layers) ? f.layers : c->run_layers;
    // Patch embedding with the gather inside the GEMM's A loader (kernels_patch.hip; the split residual of the 16-bit paths):
    // VH_PATCH_FUSED=1 selects it -- measured, DESIGN.md 4.4 -- the default is the im2col pass + the persistent GEMM.
    const bool fused_patch = c->patch_fused && c->split && !c->fp8 && nl > 0 && patch_fused_supported(f.image_size, f.patch_size, f.channels, D);
    if (!fused_patch) HIPCHK(&c->err, launch_im2col(in, batch, f.image_size, f.patch_size, f.channels, col16, dt16, s));
    if ((rc = mark(ST_IM2COL))) return rc;
    if (fused_patch) {
        HIPCHK(&c->err, launch_patch_fused(in, batch, f.image_size, f.patch_size, f.channels, c->wp16, P + L.patch_b, P + L.pos, xn16, xlo16,
                                           partials_p, rows_g, D, dt16, s));
        if ((rc = mark(ST_PATCH))) return rc;
        HIPCHK(&c->err, launch_cls_rows_split(xn16, xlo16, partials_p, rows_g, P + L.cls, P + L.pos, batch, T, D, dt16, s));
        if ((rc = mark(ST_CLS))) return rc;
        HIPCHK(&c->err, launch_finalize_stats(partials_p, D / 64, rows_g, D, f.ln_eps, stats_p, s, rows, c->guard_dev, amax_guard));
        if ((rc = mark(ST_LNSTATS))) return rc;
    } else if (c->split && !c->fp8 && nl > 0) {
        // Split residual: the patch embedding lands DIRECTLY in the two 16-bit planes, with the first row statistics'
        // partial sums (PATCH_SPLIT epilogue; the class-token rows from their own small kernel) -- no fp32 x, no separate
        // row-statistics pass over it (round 3: -0.1 ms per forward).  The partial sums are laid out for rows_g rows, like
        // every later layer's.
        GemmArgs g{col16, c->wp16, P + L.patch_b, xn16, (int64_t)batch * L.NP, D, L.KP, VH_EPI_PATCH_SPLIT, P + L.pos, L.NP, dt16, 0};
        g.out16 = xlo16; g.partials = partials_p; g.prow = rows_g;
        HIPCHK(&c->err, launch_gemm(g, s));
        if ((rc = mark(ST_PATCH))) return rc;
        HIPCHK(&c->err, launch_cls_rows_split(xn16, xlo16, partials_p, rows_g, P + L.cls, P + L.pos, batch, T, D, dt16, s));
        if ((rc = mark(ST_CLS))) return rc;
        HIPCHK(&c->err, launch_finalize_stats(partials_p, D / 64, rows_g, D, f.ln_eps, stats_p, s, rows, c->guard_dev, amax_guard));
        if ((rc = mark(ST_LNSTATS))) return rc;
    } else {
        HIPCHK(&c->err, gemm(col16, c->wp16, P + L.patch_b, x, (int64_t)batch * L.NP, D, L.KP, VH_EPI_PATCH, P + L.pos, L.NP));
        if ((rc = mark(ST_PATCH))) return rc;
        HIPCHK(&c->err, launch_cls_rows(x, P + L.cls, P + L.pos, batch, T, D, s));
        if ((rc = mark(ST_CLS))) return rc;
        if (c->ln_fold && nl > 0) {
            // layer 0's LN1 statistics: its input comes from the patch embedding, not from a RESID_LN epilogue
            // (fp8 path with the split residual: the patch embedding stays the bf16 fp32-out GEMM; this pass makes the planes)
            if (c->split) HIPCHK(&c->err, launch_rowstats_split(x, rows_g, D, f.ln_eps, xn16, xlo16, stats_p, c->fp8 ? VH_DTYPE_FP8 : dt16, s, rows, amax_guard));
            else HIPCHK(&c->err, launch_rowstats_cast(x, rows_g, D, f.ln_eps, xn16, stats_p, c->fp8 ? VH_DTYPE_FP8 : dt16, s, rows, amax_guard));
            HIPCHK(&c->err, launch_ln_guard(stats_p, rows, c->guard_dev, s));   // real rows only (rows_g - rows are tile padding)
            if ((rc = mark(ST_LNSTATS))) return rc;
        }
    }
    // the class-token tail needs the split planes of the folded 16-bit path, the whole model, and room in the patch-matrix buffer
    const bool tail = c->cls_tail && c->ln_fold && c->split && !c->fp8 && nl == f.layers && c->run_layers < 0 && T <= 1024 &&
                      (size_t)L.NP * L.KP * esz >= 2 * ((size_t)D * esz + 256);
    // tiled hidden activation: both MLP GEMMs must take the persistent form (whole 256-row tiles, enough of them), the 16-bit split path
    const bool h_tiled = c->h_tiled && c->split && !c->fp8 && c->weights_ready_tiled && gemm_tiled_applies(rows_g, M, D) && gemm_tiled_applies(rows_g, D, M);
    // e4m3 operands: the e4m3 hidden activation in the tiled layout of the e4m3 operand (fc1's epilogue writes it, fc2's DMA reads it)
    const bool h_tiled8 = c->h_tiled && c->split && c->fp8 && c->weights_ready_tiled && gemm_tiled_applies_f8(rows_g, M, D) && gemm_tiled_applies_f8(rows_g, D, M);
    // the attention output likewise (16-bit ring forms -> the out-projection's tiled operand DMA); VH_ATT_TILED=0 keeps it row-major
    const bool att_tiled = h_tiled && c->att_tiled && gemm_tiled_applies(rows_g, D, D) && attention_tiled_applies(batch, T, f.heads) && !tail;
    const bool att_tiled8 = h_tiled8 && c->att_tiled && gemm_tiled_applies_f8(rows_g, D, D) && attention_tiled_applies(batch, T, f.heads);
    // q|k|v head-major between the projection's epilogue and attention's operand DMA (same condition + the persistent form for N = 3 D)
    const bool qkv_hm = c->qkv_hm && ((att_tiled && gemm_tiled_applies(rows_g, 3 * D, D)) || (att_tiled8 && gemm_tiled_applies_f8(rows_g, 3 * D, D)));
    c->last_h_tiled = (h_tiled || h_tiled8) && nl > 0 && c->ln_fold;
    c->last_qkv_hm = qkv_hm && nl > 0 && c->ln_fold;
    for (int l = 0; l < nl && c->ln_fold; ++l) {
        const LayerOff& o = L.layer[l];
        const float* cd = c->fold_cd + (size_t)l * (6 * D + 2 * M);
        const int nblk = D / 64;
        if ((rc = tmark(ST_QKV))) return rc;
        const float *sq = c->fp8 ? c->sqkv[l] : nullptr, *so = c->fp8 ? c->so[l] : nullptr;
        const float *s1 = c->fp8 ? c->s1[l] : nullptr, *s2 = c->fp8 ? c->s2[l] : nullptr;
        if (qkv_hm) {
            GemmArgs gq{xn16, c->wqkv16[l], cd + 3 * D, qkv16, rows_g, 3 * D, D, VH_EPI_LNFOLD, cd, 0, c->fp8 ? VH_DTYPE_FP8 : dt16, 0};
            gq.stats = stats_p; gq.out_tiled = 1; gq.wscale = sq;   // (e4m3 operands: the weight scales; null otherwise)
            HIPCHK(&c->err, c->fp8 ? launch_gemm_fp8(gq, s) : launch_gemm(gq, s));
        } else
        HIPCHK(&c->err, gemm(xn16, c->wqkv16[l], cd + 3 * D, qkv16, rows_g, 3 * D, D, VH_EPI_LNFOLD, cd, 0, sq));
        if ((rc = tmark(ST_QKV))) return rc;
        if ((rc = mark(ST_QKV))) return rc;
        if (tail && l + 1 == nl) {
            // VH_FLAG_CLS_TAIL, last layer: only the class-token row of every image reaches the head, so attention runs for that
            // one query (all keys and values), and out-proj, fc1 and fc2 run on the `batch` class rows, gathered into compact
            // planes (the idle patch-matrix buffer).  Same kernels, same per-row arithmetic as the full layer, except the
            // attention row (a VALU kernel with another summation order): logits agree to rounding, not bitwise.
            char* const hc = col16;
            char* const lc = col16 + (((size_t)batch * D * esz + 255) & ~(size_t)255);
            HIPCHK(&c->err, launch_attention_cls(qkv16, batch, T, f.heads, att16, dt16, s));
            if ((rc = mark(ST_ATTN))) return rc;
            HIPCHK(&c->err, hipMemcpy2DAsync(hc, (size_t)D * esz, xn16, (size_t)T * D * esz, (size_t)D * esz, batch, hipMemcpyDeviceToDevice, s));
            HIPCHK(&c->err, hipMemcpy2DAsync(lc, (size_t)D, xlo16, (size_t)T * D, (size_t)D, batch, hipMemcpyDeviceToDevice, s));
            GemmArgs gp{att16, c->wo16[l], P + o.ob, hc, batch, D, D, VH_EPI_RESID_SPLIT, nullptr, 0, dt16, 0};
            gp.out16 = lc; gp.partials = partials_p;
            HIPCHK(&c->err, launch_gemm(gp, s));
            if ((rc = mark(ST_PROJ))) return rc;
            HIPCHK(&c->err, launch_finalize_stats(partials_p, nblk, batch, D, f.ln_eps, stats_p, s, batch, c->guard_dev));
            if ((rc = mark(ST_LNSTATS))) return rc;
            GemmArgs g1{hc, c->w1_16[l], cd + 6 * D + M, h16, batch, M, D, VH_EPI_LNFOLD_GELU, cd + 6 * D, 0, dt16, 0};
            g1.stats = stats_p;
            HIPCHK(&c->err, launch_gemm(g1, s));
            if ((rc = mark(ST_FC1))) return rc;
            GemmArgs g2{h16, c->w2_16[l], P + o.f2b, hc, batch, D, M, VH_EPI_RESID_SPLIT, nullptr, 0, dt16, 0};
            g2.out16 = lc; g2.partials = partials_p;
            HIPCHK(&c->err, launch_gemm(g2, s));
            if ((rc = mark(ST_FC2))) return rc;
            HIPCHK(&c->err, launch_layernorm_split(hc, lc, batch, D, D, P + L.lnfw, P + L.lnfb, f.ln_eps, clsn32, dt16, s));
            if ((rc = mark(ST_LNF))) return rc;
            HIPCHK(&c->err, launch_head_f32(clsn32, P + L.headw, P + L.headb, logits, batch, f.classes, D, s));
            if ((rc = mark(ST_HEAD))) return rc;
            c->last_batch = batch;
            return VH_OK;
        }
        if ((rc = tmark(ST_ATTN))) return rc;
        HIPCHK(&c->err, launch_attention(qkv16, batch, T, f.heads, att16, c->fp8 ? VH_DTYPE_FP8 : dt16, tickets_part + l, s, true, att_tiled || att_tiled8,
                                         qkv_hm ? (int64_t)rows_g : 0));
        if ((rc = tmark(ST_ATTN))) return rc;
        if ((rc = mark(ST_ATTN))) return rc;
        if ((rc = tmark(ST_PROJ))) return rc;
        if (att_tiled) {
            GemmArgs go{att16, c->wot_16[l], P + o.ob, xn16, rows_g, D, D, VH_EPI_RESID_SPLIT, nullptr, 0, dt16, 0};
            go.out16 = xlo16; go.partials = partials_p; go.ab_tiled = 1;
            HIPCHK(&c->err, launch_gemm(go, s));
        } else if (att_tiled8) {
            GemmArgs go{att16, c->wot_16[l], P + o.ob, xn16, rows_g, D, D, VH_EPI_RESID_SPLIT, so, 0, VH_DTYPE_FP8, 0};   // (`aux` = the weight scales)
            go.out16 = xlo16; go.partials = partials_p; go.ab_tiled = 1;
            HIPCHK(&c->err, launch_gemm_fp8(go, s));
        } else
        if (c->split) HIPCHK(&c->err, gemm(att16, c->wo16[l], P + o.ob, xn16, rows_g, D, D, VH_EPI_RESID_SPLIT, nullptr, 0, so));
        else HIPCHK(&c->err, gemm(att16, c->wo16[l], P + o.ob, x, rows_g, D, D, VH_EPI_RESID_LN, nullptr, 0, so));
        if ((rc = tmark(ST_PROJ))) return rc;
        if ((rc = mark(ST_PROJ))) return rc;
        HIPCHK(&c->err, launch_finalize_stats(partials_p, nblk, rows_g, D, f.ln_eps, stats_p, s, rows, c->guard_dev, amax_guard));
        if ((rc = mark(ST_LNSTATS))) return rc;
        if ((rc = tmark(ST_FC1))) return rc;
        if (h_tiled) {   // h in its tiled layout: written straight from fc1's registers, read by fc2's DMA (same values, same bits)
            GemmArgs g1{xn16, c->w1_16[l], cd + 6 * D + M, h16, rows_g, M, D, VH_EPI_LNFOLD_GELU, cd + 6 * D, 0, dt16, 0};
            g1.stats = stats_p; g1.out_tiled = 1;
            HIPCHK(&c->err, launch_gemm(g1, s));
        } else if (h_tiled8) {
            GemmArgs g1{xn16, c->w1_16[l], cd + 6 * D + M, h16, rows_g, M, D, VH_EPI_LNFOLD_GELU, cd + 6 * D, 0, VH_DTYPE_FP8, 0};
            g1.stats = stats_p; g1.wscale = s1; g1.out_tiled = 1;
            HIPCHK(&c->err, launch_gemm_fp8(g1, s));
        } else
        HIPCHK(&c->err, gemm(xn16, c->w1_16[l], cd + 6 * D + M, h16, rows_g, M, D, VH_EPI_LNFOLD_GELU, cd + 6 * D, 0, s1));
        if ((rc = tmark(ST_FC1))) return rc;
        if ((rc = mark(ST_FC1))) return rc;
        if ((rc = tmark(ST_FC2))) return rc;
        if (h_tiled) {
            GemmArgs g2{h16, c->w2t_16[l], P + o.f2b, xn16, rows_g, D, M, VH_EPI_RESID_SPLIT, nullptr, 0, dt16, 0};
            g2.out16 = xlo16; g2.partials = partials_p; g2.ab_tiled = 1;
            HIPCHK(&c->err, launch_gemm(g2, s));
        } else if (h_tiled8) {
            GemmArgs g2{h16, c->w2t_16[l], P + o.f2b, xn16, rows_g, D, M, VH_EPI_RESID_SPLIT, s2, 0, VH_DTYPE_FP8, 0};   // (`aux` = the weight scales)
            g2.out16 = xlo16; g2.partials = partials_p; g2.ab_tiled = 1;
            HIPCHK(&c->err, launch_gemm_fp8(g2, s));
        } else
        if (c->split) HIPCHK(&c->err, gemm(h16, c->w2_16[l], P + o.f2b, xn16, rows_g, D, M, VH_EPI_RESID_SPLIT, nullptr, 0, s2));
        else HIPCHK(&c->err, gemm(h16, c->w2_16[l], P + o.f2b, x, rows_g, D, M, l + 1 < nl ? VH_EPI_RESID_LN : VH_EPI_BIAS_RESID, nullptr, 0, s2));
        if ((rc = tmark(ST_FC2))) return rc;
        if ((rc = mark(ST_FC2))) return rc;
        if (l + 1 < nl) {
            HIPCHK(&c->err, launch_finalize_stats(partials_p, nblk, rows_g, D, f.ln_eps, stats_p, s, rows, c->guard_dev, amax_guard));
            if ((rc = mark(ST_LNSTATS))) return rc;
        }
    }
    // ---- the plain layer loop (bf16 / fp16 / fp8 operands) -----------------------------------------------------
    const bool plain = !c->ln_fold;
    const int op_dt = c->fp8 ? VH_DTYPE_FP8 : dt16;   // type of the GEMM A operands produced by LN / attention / fc1
    auto gemm_any = [&](const void* a, const void* w, const float* bias, const float* scale, void* out, int N, int K, int epi,
                        int tile_begin, int tile_count) {
        GemmArgs g{a, w, bias, out, rows, N, K, epi, c->fp8 ? scale : nullptr, 0, op_dt, 0};
        g.tile_begin = tile_begin;
        g.tile_count = tile_count;
        // a range of tiles exists in the one-tile-per-workgroup forms only: the persistent default (6) hands it to form 5
        if ((tile_begin || tile_count) && !c->fp8) g.variant = gemm_pick_variant(rows, N, epi) == 7 ? 7 : 5;
        return c->fp8 ? launch_gemm_fp8(g, s) : launch_gemm(g, s);
    };
    auto ln_rows = [&](int64_t r_begin, int64_t r_count, const float* w, const float* b, hipStream_t st) {
        return launch_layernorm(x + r_begin * D, r_count, D, D, w, b, f.ln_eps, xn16 + r_begin * D * esz_op, op_dt, st);
    };
    // Residual GEMM (x += A W^T + bias) followed by the LayerNorm of the updated rows.  Tail overlap: the tiles of a
    // 256x256 GEMM fill the 256 CUs in rounds and the last round is partly empty (N = dim = 768: 1182 tiles = 4.6
    // rounds).  The GEMM is launched as the full rounds (whole M-tiles) and then the tail round; the LayerNorm of
    // the rows finished by the first launch runs on a helper stream BESIDE the tail round, the rest after it.
    // Same kernels, same arithmetic, same bits.
    auto resid_gemm_ln = [&](const void* a, const void* w, const float* bias, const float* scale, int K, const float* lnw,
                             const float* lnb, int st_gemm) -> int {
        const int tiles_n = (D + 255) / 256, ntile = (int)((rows + 255) / 256) * tiles_n;
        int split = 0;
        const bool pp = c->fp8 || gemm_pick_variant(rows, D, VH_EPI_BIAS_RESID) >= 5;   // any 256x256 ping-pong form (5, 6, 7)
        if (allow_tail && c->tail_overlap && lnw && !ev && pp && ntile > c->num_cu && ntile % c->num_cu != 0 &&
            c->timing_stage != ST_LN && c->timing_stage != ST_PROJ && c->timing_stage != ST_FC2) {
            split = ntile / c->num_cu * c->num_cu;
            split -= split % tiles_n;
        }
        int r;
        if ((r = tmark(st_gemm))) return r;
        if (!split) {
            HIPCHK(&c->err, gemm_any(a, w, bias, scale, x, D, K, VH_EPI_BIAS_RESID, 0, 0));
            if ((r = tmark(st_gemm))) return r;
            if ((r = mark(st_gemm))) return r;
            if (!lnw) return VH_OK;
            if ((r = tmark(ST_LN))) return r;
            HIPCHK(&c->err, ln_rows(0, rows, lnw, lnb, s));
            if ((r = tmark(ST_LN))) return r;
            return mark(ST_LN);
        }
        ++c->tail_splits;
        const int64_t rows_a = (int64_t)(split / tiles_n) * 256;
        HIPCHK(&c->err, gemm_any(a, w, bias, scale, x, D, K, VH_EPI_BIAS_RESID, 0, split));
        HIPCHK(&c->err, hipEventRecord(c->ev_tail_a, s));
        HIPCHK(&c->err, gemm_any(a, w, bias, scale, x, D, K, VH_EPI_BIAS_RESID, split, ntile - split));
        HIPCHK(&c->err, hipStreamWaitEvent(c->tstream, c->ev_tail_a, 0));
        HIPCHK(&c->err, ln_rows(0, rows_a, lnw, lnb, c->tstream));
        HIPCHK(&c->err, hipEventRecord(c->ev_tail_l, c->tstream));
        HIPCHK(&c->err, ln_rows(rows_a, rows - rows_a, lnw, lnb, s));
        HIPCHK(&c->err, hipStreamWaitEvent(s, c->ev_tail_l, 0));
        return tmark(st_gemm);
    };
    if (plain && nl > 0) {
        const LayerOff& o0 = L.layer[0];
        if ((rc = tmark(ST_LN))) return rc;
        HIPCHK(&c->err, ln_rows(0, rows, P + o0.ln1w, P + o0.ln1b, s));
        if ((rc = tmark(ST_LN))) return rc;
        if ((rc = mark(ST_LN))) return rc;
    }
    for (int l = 0; l < nl && plain; ++l) {
        const LayerOff& o = L.layer[l];
        const float *sq = c->fp8 ? c->sqkv[l] : nullptr, *so = c->fp8 ? c->so[l] : nullptr;
        const float *s1 = c->fp8 ? c->s1[l] : nullptr, *s2 = c->fp8 ? c->s2[l] : nullptr;
        if ((rc = tmark(ST_QKV))) return rc;
        HIPCHK(&c->err, gemm_any(xn16, c->wqkv16[l], c->bqkv + (size_t)l * 3 * D, sq, qkv16, 3 * D, D, VH_EPI_BIAS, 0, 0));
        if ((rc = tmark(ST_QKV))) return rc;
        if ((rc = mark(ST_QKV))) return rc;
        if ((rc = tmark(ST_ATTN))) return rc;
        HIPCHK(&c->err, launch_attention(qkv16, batch, T, f.heads, att16, op_dt, tickets_part + l, s, true));
        if ((rc = tmark(ST_ATTN))) return rc;
        if ((rc = mark(ST_ATTN))) return rc;
        if ((rc = resid_gemm_ln(att16, c->wo16[l], P + o.ob, so, D, P + o.ln2w, P + o.ln2b, ST_PROJ))) return rc;
        if ((rc = tmark(ST_FC1))) return rc;
        HIPCHK(&c->err, gemm_any(xn16, c->w1_16[l], P + o.f1b, s1, h16, M, D, VH_EPI_BIAS_GELU, 0, 0));
        if ((rc = tmark(ST_FC1))) return rc;
        if ((rc = mark(ST_FC1))) return rc;
        const bool more = l + 1 < nl;   // the next layer's LN1 follows this layer's fc2
        if ((rc = resid_gemm_ln(h16, c->w2_16[l], P + o.f2b, s2, M, more ? P + L.layer[l + 1].ln1w : nullptr,
                                more ? P + L.layer[l + 1].ln1b : nullptr, ST_FC2))) return rc;
    }
    if (c->split && nl > 0)
        HIPCHK(&c->err, launch_layernorm_split(xn16, xlo16, batch, D, (int64_t)T * D, P + L.lnfw, P + L.lnfb, f.ln_eps, clsn32, c->fp8 ? VH_DTYPE_FP8 : dt16, s));
    else
        HIPCHK(&c->err, launch_layernorm(x, batch, D, (int64_t)T * D, P + L.lnfw, P + L.lnfb, f.ln_eps, clsn32, VH_DTYPE_F32_INTERNAL, s));
    if ((rc = mark(ST_LNF))) return rc;
    HIPCHK(&c->err, launch_head_f32(clsn32, P + L.headw, P + L.headb, logits, batch, f.classes, D, s));
    if ((rc = mark(ST_HEAD))) return rc;
    c->last_batch = batch;
    return VH_OK;
}

// end of a forward: the guard word travels to pinned host memory behind the forward's kernels (stream order)
int guard_publish(vh_ctx* c) {
    if (c->ln_fold) HIPCHK(&c->err, hipMemcpyAsync(c->guard_host, c->guard_dev, 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return VH_OK;
}
// top of every forward entry point: has a COMPLETED forward seen rows beyond the guard's threshold?  (NaN trips it too.)
// Second word, fp8 operands only: rows whose raw values leave e4m3's range (the folded operand would clip them).
constexpr float kE4M3Max = 448.f;
bool guard_exceeded(const vh_ctx* c) {
    if (!c->ln_fold || !c->guard_host) return false;
    const volatile float* g = c->guard_host;
    return !(g[0] <= c->guard_thresh) || (c->fp8 && !(g[1] <= kE4M3Max));
}
int prepare_weights(vh_ctx* c);
void drop_graphs(vh_ctx* c);
int guard_poll(vh_ctx* c) {
    if (!guard_exceeded(c)) return VH_OK;
    c->guard_tripped = true;
    if (!c->guard_auto) return VH_OK;   // VH_FLAG_LN_FOLD_ON: the caller's explicit choice stands; vh_get_ln_guard reports
    c->ln_fold = false;
    c->split = false;
    drop_graphs(c);
    return prepare_weights(c);          // the plain (unfolded) 16-bit / e4m3 weights, from the resident blob
}

// one complete forward of `batch` images on the context's stream(s): the (optional) concurrent parts are forked
// from and joined back into c->stream, so everything that follows on c->stream sees the finished logits
int enqueue_step(vh_ctx* c, const float* in, int batch, float* logits) {
    const size_t img_floats = (size_t)c->cfg.image_size * c->cfg.image_size * c->cfg.channels;
    const int parts = batch < c->nstreams ? batch : c->nstreams;
    int rc;
    if (parts > 1) {
        // contiguous parts of the batch on different streams: the HBM-bound stages and the partly filled
        // tail rounds of one part overlap the MFMA-bound stages of another (identical results: images
        // are independent and every per-row reduction has a fixed order)
        HIPCHK(&c->err, hipEventRecord(c->ev_fork, c->stream));
        int b0 = 0;
        for (int p = 0; p < parts; ++p) {
            const int nb = batch / parts + (p < batch % parts ? 1 : 0);
            hipStream_t st = p == 0 ? c->stream : c->xstream[p - 1];
            if (p) HIPCHK(&c->err, hipStreamWaitEvent(st, c->ev_fork, 0));
            if ((rc = enqueue_forward(c, in + (size_t)b0 * img_floats, nb, logits + (size_t)b0 * c->cfg.classes, nullptr, st, b0, false,
                                      /*may_pad: only the last part has nothing behind its rows*/ p == parts - 1))) return rc;
            if (p) {
                HIPCHK(&c->err, hipEventRecord(c->ev_join[p - 1], st));
                HIPCHK(&c->err, hipStreamWaitEvent(c->stream, c->ev_join[p - 1], 0));
            }
            b0 += nb;
        }
        c->last_batch = batch;
        return guard_publish(c);
    }
    if ((rc = enqueue_forward(c, in, batch, logits, nullptr, c->stream, 0, true))) return rc;
    return guard_publish(c);
}

void drop_graphs(vh_ctx* c) {
    for (auto& g : c->graphs) {
        if (g.exec) hipGraphExecDestroy(g.exec);
        if (g.graph) hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

// enqueue_step, or the replay of its captured launch sequence.  Small batches are launch-bound (ViT-B/16 at
// batch 1: ~100 launches, 1.27 ms eager): the graph removes the per-launch API cost and the gaps between kernels.
int run_step(vh_ctx* c, const float* in, int batch, float* logits) {
    if (int rc = guard_poll(c)) return rc;
    if (!c->use_graph || c->timing_stage >= 0) return enqueue_step(c, in, batch, logits);
    for (auto& g : c->graphs)
        if (g.in == in && g.out == logits && g.batch == batch) {
            HIPCHK(&c->err, hipGraphLaunch(g.exec, c->stream));
            c->last_batch = batch;
            return VH_OK;
        }
    bool warm = false;
    for (int b : c->graph_warm) warm |= b == batch;
    if (!warm) {  // first forward at this batch size: eager, so every kernel it uses has its attributes set
        c->graph_warm.push_back(batch);
        return enqueue_step(c, in, batch, logits);
    }
    if (c->graphs.size() >= 128) drop_graphs(c);
    vh_ctx::GraphEntry g{in, logits, batch, nullptr, nullptr};
    HIPCHK(&c->err, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_step(c, in, batch, logits);
    const hipError_t e = hipStreamEndCapture(c->stream, &g.graph);   // always close the capture
    if (rc) { if (g.graph) hipGraphDestroy(g.graph); return rc; }
    if (e != hipSuccess) return fail(&c->err, VH_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    const hipError_t ei = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
    if (ei != hipSuccess) { hipGraphDestroy(g.graph); return fail(&c->err, VH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ei)); }
    c->graphs.push_back(g);
    HIPCHK(&c->err, hipGraphLaunch(g.exec, c->stream));
    c->last_batch = batch;
    return VH_OK;
}

// Weight-load time: decide the guarded fold BEFORE the first caller's forward.  A checkpoint's common-mode offset (position
// embedding, biases) and its massive activations are properties of the weights far more than of the image, so ONE seeded
// calibration image through the freshly prepared folded path measures them; if the guard trips, the context switches to
// the stand-alone LayerNorm here, and the asynchronous entry points (vh_forward_device_async, the ring, the groups) never
// return a batch computed on a tripped fold because of the WEIGHTS.  The per-forward guard stays as the backstop for what
// only the data can cause.  ~1.5 ms per weight load (a batch-1 forward).
int guard_calibrate(vh_ctx* c) {
    if (!c->guard_auto || !c->ln_fold) return VH_OK;
    const vh_config& f = c->cfg;
    const int64_t n = (int64_t)f.image_size * f.image_size * f.channels;
    HIPCHK(&c->err, launch_fill(c->in_dev, n, 0xCA11B8A7Eull, TID_IMAGES, 0, 0.f, 0.f, c->stream));
    const int keep_stage = c->timing_stage;
    c->timing_stage = -1;
    const int rc = enqueue_step(c, c->in_dev, 1, c->logits_dev);
    c->timing_stage = keep_stage;
    if (rc) return rc;
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    return guard_poll(c);
}

int check_forward_args(vh_ctx* c, const void* in, int batch, const void* out) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    if (!in || !out) return fail(&c->err, VH_ERR_INVALID, "null buffer");
    if (batch <= 0 || batch > c->cfg.max_batch)
        return fail(&c->err, VH_ERR_INVALID, "batch %d outside 1..max_batch=%d", batch, c->cfg.max_batch);
    if (!c->weights_ready) return fail(&c->err, VH_ERR_STATE, "forward before weights were loaded");
    return VH_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int vh_ring_destroy(vh_ctx* c);

int vh_abi_version(void) { return VH_ABI_VERSION; }

int vh_device_count(int* count) {
    if (!count) return fail(nullptr, VH_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, VH_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return VH_OK;
}

const char* vh_last_error(const vh_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int vh_malloc(int device, size_t nbytes, void** p) {
    if (!p || !nbytes) return fail(nullptr, VH_ERR_INVALID, "vh_malloc: bad argument");
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipMalloc(p, nbytes));
    return VH_OK;
}
int vh_free(int device, void* p) {
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipFree(p));
    return VH_OK;
}
int vh_memcpy_h2d(int device, void* d, const void* h, size_t n) {
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    return VH_OK;
}
int vh_memcpy_d2h(int device, void* h, const void* d, size_t n) {
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    return VH_OK;
}
int vh_device_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
    if (!free_bytes || !total_bytes) return fail(nullptr, VH_ERR_INVALID, "null argument");
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipMemGetInfo(free_bytes, total_bytes));
    return VH_OK;
}
int vh_device_synchronize(int device) {
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    HIPCHK(nullptr, hipDeviceSynchronize());
    return VH_OK;
}

size_t vh_weight_blob_bytes(const vh_config* cfg) {
    if (!cfg || check_config(*cfg)) return 0;
    return blob_bytes_of(*cfg);
}

int vh_create(const vh_config* cfg, int device, vh_ctx** out) {
    if (!cfg || !out) return fail(nullptr, VH_ERR_INVALID, "vh_create: null argument");
    *out = nullptr;
    if (const char* why = check_config(*cfg)) return fail(nullptr, VH_ERR_INVALID, "vh_create: %s", why);
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, VH_ERR_NO_DEVICE, "device %d is %s; libvithip is built for gfx950 only", device, prop.gcnArchName);

    vh_ctx* c = nullptr;
    try {
        c = new vh_ctx();
        c->cfg = *cfg;
        c->L = make_layout(*cfg);
    } catch (const std::exception& ex) {   // nothing may leave an extern "C" entry point as a C++ exception
        delete c;
        return fail(nullptr, VH_ERR_INVALID, "vh_create: %s", ex.what());
    }
    c->device = device;
    const Layout& L = c->L;
    const size_t D = cfg->dim, M = cfg->mlp_dim, C = cfg->classes, B = cfg->max_batch;
    const size_t rows = B * (size_t)L.T;
    auto bail = [&](int code) { vh_destroy(c); return code; };
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fail(nullptr, VH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); return bail(VH_ERR_HIP); } } while (0)
    CK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < vh_ctx::kMaxStreams - 1; ++i) {
        CK(hipStreamCreateWithFlags(&c->xstream[i], hipStreamNonBlocking));
        CK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    {
        const char* e = getenv("VH_STREAMS");
        c->nstreams = e ? atoi(e) : 1;
        if (c->nstreams < 1) c->nstreams = 1;
        if (c->nstreams > vh_ctx::kMaxStreams) c->nstreams = vh_ctx::kMaxStreams;
    }
    CK(hipStreamCreateWithFlags(&c->tstream, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&c->ev_tail_a, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&c->ev_tail_l, hipEventDisableTiming));
    c->num_cu = prop.multiProcessorCount;
    {
        // opt-in: measured +-0.5 % on ViT-B/16 b512 (the forward is power-capped: filling idle CUs moves energy around
        // instead of saving it), so the simpler single-launch sequence stays the default
        const char* e = getenv("VH_TAIL_OVERLAP");
        c->tail_overlap = e && e[0] == '1';
    }
    {
        const char* e = getenv("VH_GRAPH");
        c->use_graph = e && e[0] == '1';
    }
    CK(hipEventCreate(&c->ev0));
    CK(hipEventCreate(&c->ev1));
    // canonical blob
    CK(hipMalloc((void**)&c->blob, sizeof(BlobHeader) + 4 * L.total));
    c->params = (float*)(c->blob + sizeof(BlobHeader));
    // 16-bit weights
    size_t w16_bytes = 0;
    auto carve16 = [&](size_t elems) { size_t o = w16_bytes; w16_bytes += align_up(elems * 2, 256); return o; };
    const size_t o_wp = carve16(D * L.KP);
    std::vector<size_t> o_qkv(cfg->layers), o_o(cfg->layers), o_1(cfg->layers), o_2(cfg->layers), o_2t(cfg->layers);
    const bool want_tiled_any = D % 256 == 0 && M % 256 == 0 && !(getenv("VH_H_TILED") && getenv("VH_H_TILED")[0] == '0');
    const bool want_tiled = cfg->dtype != VH_DTYPE_FP8 && want_tiled_any;
    const bool want_tiled8 = cfg->dtype == VH_DTYPE_FP8 && want_tiled_any;   // e4m3 operands: fc2's weights (bytes) in the tiled e4m3 layout
    for (int l = 0; l < cfg->layers; ++l) {
        o_qkv[l] = carve16(3 * D * D); o_o[l] = carve16(D * D); o_1[l] = carve16(M * D); o_2[l] = carve16(D * M);
        o_2t[l] = want_tiled ? carve16(D * M) : want_tiled8 ? carve16((D * M + 1) / 2) : 0;
    }
    std::vector<size_t> o_ot(cfg->layers);
    for (int l = 0; l < cfg->layers; ++l) o_ot[l] = want_tiled ? carve16(D * D) : want_tiled8 ? carve16((D * D + 1) / 2) : 0;
    const size_t o_bqkv = w16_bytes;
    w16_bytes += align_up((size_t)cfg->layers * 3 * D * 4, 256);
    {
        // LayerNorm folded into the neighbouring GEMMs: a property of the MODEL SHAPE and of vh_config.flags only -- never of
        // max_batch, so that a context sized for 1 image and one sized for 512 give the same logit bits for the same image
        // (and hip::net_hip, which re-creates its context when a larger batch arrives, keeps its numerics).  On by default
        // where the shapes allow it.  VH_LN_FOLD=0 / 1 (environment, A/B tools) applies only when the flags leave the choice
        // to the library.
        const bool eligible = (cfg->dim % 256 == 0) && (cfg->mlp_dim % 256 == 0);
        bool want = true;
        if (cfg->flags & VH_FLAG_LN_FOLD_OFF) want = false;
        else if (!(cfg->flags & VH_FLAG_LN_FOLD_ON)) { const char* e = getenv("VH_LN_FOLD"); if (e) want = e[0] == '1'; }
        c->ln_fold = eligible && want;
        // the library's own default (no flag, no VH_LN_FOLD in the environment) is the GUARDED fold
        c->guard_auto = c->ln_fold && !(cfg->flags & (VH_FLAG_LN_FOLD_ON | VH_FLAG_LN_FOLD_OFF)) && !getenv("VH_LN_FOLD");
        if (const char* e = getenv("VH_LN_GUARD")) { const float t = (float)atof(e); if (t > 0.f) c->guard_thresh = t; }
    }
    c->fp8 = cfg->dtype == VH_DTYPE_FP8;
    c->dt16 = c->fp8 ? VH_DTYPE_BF16 : cfg->dtype;
    {
        // (fp8 operands: the hi plane is e4m3 -- the GEMM operand itself -- and the lo plane bf16: 3 bytes per element as well)
        const char* e = getenv("VH_RESID_SPLIT");
        c->split = c->ln_fold && !(e && e[0] == '0');
        c->ln_fold_cfg = c->ln_fold;
        c->split_cfg = c->split;
        c->cls_tail = (cfg->flags & VH_FLAG_CLS_TAIL) != 0;
        c->h_tiled = want_tiled || want_tiled8;
        { const char* e = getenv("VH_ATT_TILED"); c->att_tiled = !(e && e[0] == '0'); }
        { const char* e = getenv("VH_QKV_HM"); c->qkv_hm = !(e && e[0] == '0'); }
        const char* pf = getenv("VH_PATCH_FUSED");
        c->patch_fused = pf && pf[0] == '1';
    }
    CK(hipHostMalloc((void**)&c->guard_host, 64, hipHostMallocDefault));
    *c->guard_host = 0.f;
    const size_t o_cd = w16_bytes;
    w16_bytes += align_up((size_t)cfg->layers * (6 * D + 2 * M) * 4, 256);
    const size_t o_sc = w16_bytes, sc_per_layer = 5 * D + M;  // fp8: scales of q|k|v (3D), o (D), fc1 (M), fc2 (D)
    if (c->fp8) w16_bytes += align_up((size_t)cfg->layers * sc_per_layer * 4, 256);
    CK(hipMalloc((void**)&c->w16, w16_bytes));
    c->wp16 = c->w16 + o_wp; c->bqkv = (float*)(c->w16 + o_bqkv);
    c->fold_cd = (float*)(c->w16 + o_cd);
    for (int l = 0; l < cfg->layers; ++l) {
        c->wqkv16.push_back(c->w16 + o_qkv[l]); c->wo16.push_back(c->w16 + o_o[l]);
        c->w1_16.push_back(c->w16 + o_1[l]); c->w2_16.push_back(c->w16 + o_2[l]);
        c->w2t_16.push_back(want_tiled || want_tiled8 ? c->w16 + o_2t[l] : nullptr);
        c->wot_16.push_back(want_tiled || want_tiled8 ? c->w16 + o_ot[l] : nullptr);
        if (c->fp8) {
            float* sc = (float*)(c->w16 + o_sc) + (size_t)l * sc_per_layer;
            c->sqkv.push_back(sc); c->so.push_back(sc + 3 * D); c->s1.push_back(sc + 4 * D); c->s2.push_back(sc + 4 * D + M);
        }
    }
    // activation arena
    size_t a = 0;
    auto carve = [&](size_t bytes) { size_t o = a; a += align_up(bytes, 256); return o; };
    // room for whole 256-row GEMM tiles behind the LAST part of a forward (enqueue_forward, rows_g): its rows start anywhere,
    // so the padding may reach up to 255 rows past the last real row
    const size_t rows_p = rows + 256;
    const size_t o_x = carve(rows_p * D * 4), o_xn = carve(rows_p * D * 2), o_qkvA = carve(rows_p * 3 * D * 2),
                 o_att = carve(rows_p * D * 2), o_h = carve(rows_p * M * 2), o_col = carve(B * L.NP * (size_t)L.KP * 2),
                 o_cls = carve(B * D * 4),
                 o_in = carve(B * (size_t)cfg->image_size * cfg->image_size * cfg->channels * 4), o_lg = carve(B * C * 4),
                 o_st = carve(rows_p * 2 * 4), o_pt = carve((D / 64 + 1) * rows_p * 2 * 4), o_xlo = carve(rows_p * D * 2),   // lo plane: one byte per element (16-bit paths) or bf16 (fp8 path)
                 o_tk = carve(B * 4 * (size_t)(cfg->layers > 0 ? cfg->layers : 1)),   // attention work-queue counters: one word per image and layer, a part uses its first image's
                 o_gd = carve(256);     // the LayerNorm-fold guard word
    CK(hipMalloc((void**)&c->arena, a));
    c->guard_dev = (unsigned int*)(c->arena + o_gd);
    CK(hipMemsetAsync(c->guard_dev, 0, 256, c->stream));
    c->x = (float*)(c->arena + o_x); c->xn16 = c->arena + o_xn; c->qkv16 = c->arena + o_qkvA; c->att16 = c->arena + o_att;
    c->h16 = c->arena + o_h; c->col16 = c->arena + o_col; c->clsn32 = (float*)(c->arena + o_cls);
    c->in_dev = (float*)(c->arena + o_in); c->logits_dev = (float*)(c->arena + o_lg);
    c->stats = (float*)(c->arena + o_st); c->partials = (float*)(c->arena + o_pt); c->xlo16 = c->arena + o_xlo;
    c->tickets = (unsigned int*)(c->arena + o_tk);
#undef CK
    *out = c;
    return VH_OK;
}

int vh_destroy(vh_ctx* c) {
    if (!c) return VH_OK;
    hipSetDevice(c->device);
    vh_ring_destroy(c);
    if (c->stream) hipStreamSynchronize(c->stream);
    drop_graphs(c);
    if (c->guard_host) hipHostFree(c->guard_host);
    if (c->arena) hipFree(c->arena);
    if (c->w16) hipFree(c->w16);
    if (c->blob) hipFree(c->blob);
    for (hipEvent_t e : c->tev) hipEventDestroy(e);
    for (hipEvent_t e : c->sev) hipEventDestroy(e);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    for (int i = 0; i < vh_ctx::kMaxStreams - 1; ++i) {
        if (c->xstream[i]) { hipStreamSynchronize(c->xstream[i]); hipStreamDestroy(c->xstream[i]); }
        if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->tstream) { hipStreamSynchronize(c->tstream); hipStreamDestroy(c->tstream); }
    if (c->ev_tail_a) hipEventDestroy(c->ev_tail_a);
    if (c->ev_tail_l) hipEventDestroy(c->ev_tail_l);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return VH_OK;
}

int vh_get_config(const vh_ctx* c, vh_config* out) {
    if (!c || !out) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *out = c->cfg;
    return VH_OK;
}

int vh_get_ln_fold(const vh_ctx* c, int* on) {
    if (!c || !on) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *on = c->ln_fold ? 1 : 0;
    return VH_OK;
}

int vh_get_ln_guard(vh_ctx* c, float* max_ratio, float* threshold, int* tripped) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    float seen[2] = {0.f, 0.f};
    HIPCHK(&c->err, hipMemcpy(seen, c->guard_dev, sizeof seen, hipMemcpyDeviceToHost));
    if ((!(seen[0] <= c->guard_thresh) || (c->fp8 && !(seen[1] <= kE4M3Max))) && c->ln_fold_cfg) c->guard_tripped = true;
    if (max_ratio) *max_ratio = seen[0];
    if (threshold) *threshold = c->guard_thresh;
    if (tripped) *tripped = c->guard_tripped ? 1 : 0;
    return VH_OK;
}

int vh_get_fp8_guard(vh_ctx* c, float* max_abs, float* limit) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    float seen[2] = {0.f, 0.f};
    HIPCHK(&c->err, hipMemcpy(seen, c->guard_dev, sizeof seen, hipMemcpyDeviceToHost));
    if (max_abs) *max_abs = c->fp8 ? seen[1] : 0.f;
    if (limit) *limit = kE4M3Max;
    return VH_OK;
}

int vh_load_weights(vh_ctx* c, const void* host_blob, size_t nbytes) {
    if (!c || !host_blob) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "vh_load_weights: null argument");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    if (nbytes != need) return fail(&c->err, VH_ERR_INVALID, "weight blob is %zu bytes, expected %zu", nbytes, need);
    BlobHeader h;
    memcpy(&h, host_blob, sizeof h);
    int rc = check_blob_header(c, h);
    if (rc) return rc;
    HIPCHK(&c->err, hipSetDevice(c->device));
    c->weights_ready = false;
    HIPCHK(&c->err, hipMemcpyAsync(c->blob, host_blob, nbytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = guard_reset(c))) return rc;
    if ((rc = prepare_weights(c))) return rc;
    return guard_calibrate(c);
}

int vh_load_weights_device(vh_ctx* c, const void* dev_blob, size_t nbytes) {
    if (!c || !dev_blob) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "vh_load_weights_device: null argument");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    if (nbytes != need) return fail(&c->err, VH_ERR_INVALID, "weight blob is %zu bytes, expected %zu", nbytes, need);
    HIPCHK(&c->err, hipSetDevice(c->device));
    BlobHeader h;
    HIPCHK(&c->err, hipMemcpy(&h, dev_blob, sizeof h, hipMemcpyDeviceToHost));
    int rc = check_blob_header(c, h);
    if (rc) return rc;
    c->weights_ready = false;
    if (dev_blob != c->blob) HIPCHK(&c->err, hipMemcpyAsync(c->blob, dev_blob, nbytes, hipMemcpyDeviceToDevice, c->stream));
    if ((rc = guard_reset(c))) return rc;
    if ((rc = prepare_weights(c))) return rc;
    return guard_calibrate(c);
}

int vh_init_weights_seeded(vh_ctx* c, uint64_t seed) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    HIPCHK(&c->err, hipSetDevice(c->device));
    const vh_config& f = c->cfg;
    const Layout& L = c->L;
    BlobHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "VHBLOB1", 8);
    h.image_size = f.image_size; h.patch_size = f.patch_size; h.channels = f.channels; h.dim = f.dim;
    h.heads = f.heads; h.mlp_dim = f.mlp_dim; h.layers = f.layers; h.classes = f.classes; h.ln_eps = f.ln_eps;
    c->weights_ready = false;
    HIPCHK(&c->err, hipMemcpy(c->blob, &h, sizeof h, hipMemcpyHostToDevice));
    const size_t D = f.dim, M = f.mlp_dim, C = f.classes;
    const float sw = 0.02f, sb = 0.02f, sg = 0.05f;
    float* P = c->params;
    hipStream_t s = c->stream;
#define GEN(off, count, tid, sigma, offs) HIPCHK(&c->err, launch_fill(P + (off), (int64_t)(count), seed, (tid), 1, (sigma), (offs), s))
    GEN(L.patch_w, D * L.KP, TID_PATCH_W, sw, 0.f);
    GEN(L.patch_b, D, TID_PATCH_B, sb, 0.f);
    GEN(L.cls, D, TID_CLS, sw, 0.f);
    GEN(L.pos, (size_t)L.T * D, TID_POS, sw, 0.f);
    for (int l = 0; l < f.layers; ++l) {
        const LayerOff& o = L.layer[l];
        const uint32_t t = TID_LAYER0 + 16u * (uint32_t)l;
        GEN(o.ln1w, D, t + 0, sg, 1.f); GEN(o.ln1b, D, t + 1, sb, 0.f);
        GEN(o.qw, D * D, t + 2, sw, 0.f); GEN(o.qb, D, t + 3, sb, 0.f);
        GEN(o.kw, D * D, t + 4, sw, 0.f); GEN(o.kb, D, t + 5, sb, 0.f);
        GEN(o.vw, D * D, t + 6, sw, 0.f); GEN(o.vb, D, t + 7, sb, 0.f);
        GEN(o.ow, D * D, t + 8, sw, 0.f); GEN(o.ob, D, t + 9, sb, 0.f);
        GEN(o.ln2w, D, t + 10, sg, 1.f); GEN(o.ln2b, D, t + 11, sb, 0.f);
        GEN(o.f1w, M * D, t + 12, sw, 0.f); GEN(o.f1b, M, t + 13, sb, 0.f);
        GEN(o.f2w, D * M, t + 14, sw, 0.f); GEN(o.f2b, D, t + 15, sb, 0.f);
    }
    GEN(L.lnfw, D, TID_FINAL + 0, sg, 1.f); GEN(L.lnfb, D, TID_FINAL + 1, sb, 0.f);
    GEN(L.headw, C * D, TID_FINAL + 2, sw, 0.f); GEN(L.headb, C, TID_FINAL + 3, sb, 0.f);
#undef GEN
    if (int rc = guard_reset(c)) return rc;
    if (int rc = prepare_weights(c)) return rc;
    return guard_calibrate(c);
}

int vh_export_weights(vh_ctx* c, void* host_blob, size_t nbytes) {
    if (!c || !host_blob) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (!c->weights_ready) return fail(&c->err, VH_ERR_STATE, "no weights loaded");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    if (nbytes != need) return fail(&c->err, VH_ERR_INVALID, "buffer is %zu bytes, blob is %zu", nbytes, need);
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipMemcpy(host_blob, c->blob, need, hipMemcpyDeviceToHost));
    return VH_OK;
}

// ---- weight blob on disk ---------------------------------------------------------------------
// The file IS the canonical blob (64-byte header + fp32 tensors in the fixed order of make_layout), with a
// checksum of the parameter bytes in the header words a memory blob leaves zero.  It is what vh_load_weights
// takes, what vh_export_weights returns and what the multi-GPU path broadcasts.
static int blob_file_header(const char* path, vh_config* cfg, size_t* need_out) {
    FILE* f = fopen(path, "rb");
    if (!f) return fail(nullptr, VH_ERR_INVALID, "cannot open %s", path);
    BlobHeader h;
    const size_t got = fread(&h, 1, sizeof h, f);
    long long fsize = -1;
    if (fseek(f, 0, SEEK_END) == 0) fsize = ftell(f);
    fclose(f);
    if (got != sizeof h || memcmp(h.magic, "VHBLOB1", 8) != 0) return fail(nullptr, VH_ERR_INVALID, "%s: not a VHBLOB1 file", path);
    vh_config c;
    memset(&c, 0, sizeof c);
    c.image_size = h.image_size; c.patch_size = h.patch_size; c.channels = h.channels; c.dim = h.dim; c.heads = h.heads;
    c.mlp_dim = h.mlp_dim; c.layers = h.layers; c.classes = h.classes; c.ln_eps = h.ln_eps;
    c.dtype = VH_DTYPE_BF16; c.max_batch = 1;
    // check_config bounds every field, so the size arithmetic below cannot overflow and nothing is allocated from the header
    if (const char* why = check_config(c)) return fail(nullptr, VH_ERR_INVALID, "%s: header describes an unsupported model (%s)", path, why);
    const size_t need = blob_bytes_of(c);
    if (fsize != (long long)need) return fail(nullptr, VH_ERR_INVALID, "%s: %lld bytes, the header implies %zu", path, fsize, need);
    *cfg = c;
    if (need_out) *need_out = need;
    return VH_OK;
}

int vh_blob_file_config(const char* path, vh_config* cfg) {
    if (!path || !cfg) return fail(nullptr, VH_ERR_INVALID, "null argument");
    return blob_file_header(path, cfg, nullptr);
}

int vh_blob_file_read(const char* path, void* host_blob, size_t nbytes) {
    if (!path || !host_blob) return fail(nullptr, VH_ERR_INVALID, "null argument");
    vh_config c;
    size_t need = 0;
    int rc = blob_file_header(path, &c, &need);
    if (rc) return rc;
    if (nbytes != need) return fail(nullptr, VH_ERR_INVALID, "%s is %zu bytes, the buffer %zu", path, need, nbytes);
    FILE* f = fopen(path, "rb");
    if (!f) return fail(nullptr, VH_ERR_INVALID, "cannot open %s", path);
    const size_t got = fread(host_blob, 1, need, f);
    fclose(f);
    if (got != need) return fail(nullptr, VH_ERR_INVALID, "%s: short read", path);
    BlobHeader h;
    memcpy(&h, host_blob, sizeof h);
    if (h.flags & 1) {
        const uint64_t sum = fnv1a64((const char*)host_blob + sizeof h, need - sizeof h);
        if ((uint32_t)sum != h.sum_lo || (uint32_t)(sum >> 32) != h.sum_hi)
            return fail(nullptr, VH_ERR_INVALID, "%s: checksum mismatch (file damaged)", path);
    }
    h.sum_lo = h.sum_hi = h.flags = 0;   // memory form
    h.pad[0] = h.pad[1] = 0;
    memcpy(host_blob, &h, sizeof h);
    return VH_OK;
}

int vh_save_weights_file(vh_ctx* c, const char* path) {
    if (!c || !path) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (!c->weights_ready) return fail(&c->err, VH_ERR_STATE, "no weights loaded");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    std::vector<char> buf(need);
    int rc = vh_export_weights(c, buf.data(), need);
    if (rc) return rc;
    BlobHeader h;
    memcpy(&h, buf.data(), sizeof h);
    const uint64_t sum = fnv1a64(buf.data() + sizeof h, need - sizeof h);
    h.sum_lo = (uint32_t)sum; h.sum_hi = (uint32_t)(sum >> 32); h.flags = 1; h.pad[0] = h.pad[1] = 0;
    memcpy(buf.data(), &h, sizeof h);
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(&c->err, VH_ERR_INVALID, "cannot create %s", tmp.c_str());
    const size_t put = fwrite(buf.data(), 1, need, f);
    const int bad = fclose(f);
    if (put != need || bad) { remove(tmp.c_str()); return fail(&c->err, VH_ERR_INVALID, "short write to %s", tmp.c_str()); }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return fail(&c->err, VH_ERR_INVALID, "cannot rename %s to %s", tmp.c_str(), path); }
    return VH_OK;
}

int vh_load_weights_file(vh_ctx* c, const char* path) {
    if (!c || !path) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    FILE* f = fopen(path, "rb");
    if (!f) return fail(&c->err, VH_ERR_INVALID, "cannot open %s", path);
    std::vector<char> buf(need + 1);
    const size_t got = fread(buf.data(), 1, need + 1, f);
    fclose(f);
    if (got != need) return fail(&c->err, VH_ERR_INVALID, "%s holds %s%zu bytes, this model's blob is %zu", path, got > need ? "more than " : "", got > need ? need : got, need);
    BlobHeader h;
    memcpy(&h, buf.data(), sizeof h);
    int rc = check_blob_header(c, h);
    if (rc) return rc;
    if (h.flags & 1) {
        const uint64_t sum = fnv1a64(buf.data() + sizeof h, need - sizeof h);
        if ((uint32_t)sum != h.sum_lo || (uint32_t)(sum >> 32) != h.sum_hi)
            return fail(&c->err, VH_ERR_INVALID, "%s: checksum mismatch (file damaged)", path);
    }
    // the resident blob is the memory form: checksum words cleared, so export == what make_blob-style writers produce
    h.sum_lo = h.sum_hi = h.flags = 0;
    memcpy(buf.data(), &h, sizeof h);
    return vh_load_weights(c, buf.data(), need);
}

int vh_export_weights_device(vh_ctx* c, void* dev_blob, size_t nbytes) {
    if (!c || !dev_blob) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (!c->weights_ready) return fail(&c->err, VH_ERR_STATE, "no weights loaded");
    const size_t need = sizeof(BlobHeader) + 4 * c->L.total;
    if (nbytes != need) return fail(&c->err, VH_ERR_INVALID, "buffer is %zu bytes, blob is %zu", nbytes, need);
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipMemcpy(dev_blob, c->blob, need, hipMemcpyDeviceToDevice));
    return VH_OK;
}

int vh_forward_device_async(vh_ctx* c, const float* in, int batch, float* logits, int steps) {
    int rc = check_forward_args(c, in, batch, logits);
    if (rc) return rc;
    if (steps <= 0) return fail(&c->err, VH_ERR_INVALID, "steps must be positive");
    HIPCHK(&c->err, hipSetDevice(c->device));
    c->tev_used = 0;
    c->sev_used = 0;
    auto step_mark = [&]() -> int {
        if (!c->step_timing) return VH_OK;
        if (c->sev_used == c->sev.size()) {
            hipEvent_t e;
            HIPCHK(&c->err, hipEventCreate(&e));
            c->sev.push_back(e);
        }
        HIPCHK(&c->err, hipEventRecord(c->sev[c->sev_used++], c->stream));
        return VH_OK;
    };
    HIPCHK(&c->err, hipEventRecord(c->ev0, c->stream));
    if ((rc = step_mark())) return rc;
    for (int i = 0; i < steps; ++i) {
        if ((rc = run_step(c, in, batch, logits))) return rc;
        if ((rc = step_mark())) return rc;
    }
    HIPCHK(&c->err, hipEventRecord(c->ev1, c->stream));
    c->timed = true;
    return VH_OK;
}

int vh_synchronize(vh_ctx* c) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    // everything enqueued has completed: if one of those forwards tripped the fold's guard, switch NOW rather than at the
    // next forward's entry (vh_get_ln_guard reports it; the results already delivered came from the folded path)
    if (c->weights_ready) return guard_poll(c);
    return VH_OK;
}

int vh_forward_device(vh_ctx* c, const float* in, int batch, float* logits) {
    const auto t0 = std::chrono::high_resolution_clock::now();
    int rc = vh_forward_device_async(c, in, batch, logits, 1);
    if (rc) return rc;
    rc = vh_synchronize(c);
    if (rc) return rc;
    c->last_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return VH_OK;
}

int vh_forward(vh_ctx* c, const float* in_host, int batch, float* logits_host) {
    int rc = check_forward_args(c, in_host, batch, logits_host);
    if (rc) return rc;
    const vh_config& f = c->cfg;
    // same timing window as the reference: H2D + device work + blocking D2H (netFPGA.cpp:262-284)
    const auto t0 = std::chrono::high_resolution_clock::now();
    HIPCHK(&c->err, hipSetDevice(c->device));
    const size_t in_bytes = (size_t)batch * f.image_size * f.image_size * f.channels * 4;
    HIPCHK(&c->err, hipMemcpyAsync(c->in_dev, in_host, in_bytes, hipMemcpyHostToDevice, c->stream));
    rc = vh_forward_device_async(c, c->in_dev, batch, c->logits_dev, 1);
    if (rc) return rc;
    HIPCHK(&c->err, hipMemcpyAsync(logits_host, c->logits_dev, (size_t)batch * f.classes * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    if (c->guard_auto && guard_exceeded(c)) {
        // this (synchronous) forward itself ran folded on rows beyond the guard: run it again, now with the stand-alone LayerNorm
        rc = vh_forward_device_async(c, c->in_dev, batch, c->logits_dev, 1);   // (its run_step polls the guard and switches)
        if (rc) return rc;
        HIPCHK(&c->err, hipMemcpyAsync(logits_host, c->logits_dev, (size_t)batch * f.classes * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    }
    c->last_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return VH_OK;
}

int vh_fill_input_seeded(vh_ctx* c, uint64_t seed, int batch, float* in_dev) {
    if (!c || !in_dev || batch <= 0) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "bad argument");
    HIPCHK(&c->err, hipSetDevice(c->device));
    const int64_t n = (int64_t)batch * c->cfg.image_size * c->cfg.image_size * c->cfg.channels;
    HIPCHK(&c->err, launch_fill(in_dev, n, seed, TID_IMAGES, 0, 0.f, 0.f, c->stream));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    return VH_OK;
}

int vh_last_forward_us(const vh_ctx* c, int64_t* us) {
    if (!c || !us) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *us = c->last_us;
    return VH_OK;
}

int vh_last_kernel_ms(vh_ctx* c, double* ms) {
    if (!c || !ms) return fail(nullptr, VH_ERR_INVALID, "null argument");
    if (!c->timed) return fail(&c->err, VH_ERR_STATE, "no forward has been enqueued yet");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipEventSynchronize(c->ev1));
    float t = 0.f;
    HIPCHK(&c->err, hipEventElapsedTime(&t, c->ev0, c->ev1));
    *ms = t;
    return VH_OK;
}

const char* vh_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

int vh_profile_forward(vh_ctx* c, const float* in, int batch, float* logits, double* stage_ms, int n_slots, int* n_written) {
    int rc = check_forward_args(c, in, batch, logits);
    if (rc) return rc;
    if (!stage_ms || n_slots < 2 * ST_COUNT) return fail(&c->err, VH_ERR_INVALID, "need %d stage slots (ms then launch counts)", 2 * ST_COUNT);
    HIPCHK(&c->err, hipSetDevice(c->device));
    std::vector<std::pair<int, hipEvent_t>> ev;
    rc = enqueue_forward(c, in, batch, logits, &ev, c->stream, 0);
    if (!rc) { hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) rc = fail(&c->err, VH_ERR_HIP, "sync: %s", hipGetErrorString(e)); }
    for (int i = 0; i < 2 * ST_COUNT; ++i) stage_ms[i] = 0.0;
    if (!rc)
        for (size_t i = 1; i < ev.size(); ++i) {
            float t = 0.f;
            hipEventElapsedTime(&t, ev[i - 1].second, ev[i].second);
            stage_ms[ev[i].first] += t;
            stage_ms[ST_COUNT + ev[i].first] += 1.0;
        }
    for (auto& p : ev) hipEventDestroy(p.second);
    if (n_written) *n_written = ST_COUNT;
    return rc;
}

// ---- pipelined host path: a ring of in-flight batches ---------------------------------------------------------------
// Modelled on the one asynchronous pattern the reference has, the 24-slot image ring of filter_image /
// get_filtered_image (netFPGA.cpp:292-365, ring state :47-56): submit enqueues H2D copy -> forward -> D2H copy of
// one batch into the next free slot and returns; collect waits for the OLDEST slot.  The copies run on their own
// streams, so the upload of batch i+1 and the download of batch i-1 overlap the forward of batch i.  A full ring on
// submit / an empty ring on collect are reported as errors (the reference prints "PILA LLENA" / "PILA VACIA" and
// drops the frame, :330-333, :358-361).
int vh_ring_destroy(vh_ctx* c) {
    if (!c) return VH_OK;
    if (c->ring.empty() && !c->copy_in) return VH_OK;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->copy_in) hipStreamSynchronize(c->copy_in);
    if (c->copy_out) hipStreamSynchronize(c->copy_out);
    for (auto& s : c->ring) {
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        if (s.d_in) hipFree(s.d_in);
        if (s.d_out) hipFree(s.d_out);
        if (s.in_done) hipEventDestroy(s.in_done);
        if (s.fwd_done) hipEventDestroy(s.fwd_done);
        if (s.out_done) hipEventDestroy(s.out_done);
    }
    c->ring.clear();
    if (c->copy_in) hipStreamDestroy(c->copy_in);
    if (c->copy_out) hipStreamDestroy(c->copy_out);
    c->copy_in = c->copy_out = nullptr;
    c->ring_batch = c->ring_wr = c->ring_rd = c->ring_used = 0;
    return VH_OK;
}

int vh_ring_create(vh_ctx* c, int slots, int batch_per_slot) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    if (slots < 1 || slots > 64) return fail(&c->err, VH_ERR_INVALID, "slots must be 1..64");
    if (batch_per_slot < 1 || batch_per_slot > c->cfg.max_batch)
        return fail(&c->err, VH_ERR_INVALID, "batch_per_slot %d outside 1..max_batch=%d", batch_per_slot, c->cfg.max_batch);
    vh_ring_destroy(c);
    HIPCHK(&c->err, hipSetDevice(c->device));
    const size_t in_bytes = (size_t)batch_per_slot * c->cfg.image_size * c->cfg.image_size * c->cfg.channels * 4;
    const size_t out_bytes = (size_t)batch_per_slot * c->cfg.classes * 4;
    HIPCHK(&c->err, hipStreamCreateWithFlags(&c->copy_in, hipStreamNonBlocking));
    HIPCHK(&c->err, hipStreamCreateWithFlags(&c->copy_out, hipStreamNonBlocking));
    c->ring.resize(slots);
    for (auto& s : c->ring) {
        hipError_t e = hipHostMalloc((void**)&s.h_in, in_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_out, out_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_in, in_bytes);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_out, out_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.fwd_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming);
        if (e != hipSuccess) {
            vh_ring_destroy(c);
            return fail(&c->err, VH_ERR_HIP, "vh_ring_create: %s", hipGetErrorString(e));
        }
    }
    c->ring_batch = batch_per_slot;
    return VH_OK;
}

int vh_ring_free_slots(const vh_ctx* c, int* n) {
    if (!c || !n) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *n = (int)c->ring.size() - c->ring_used;
    return VH_OK;
}

int vh_ring_input(vh_ctx* c, float** pinned_in) {
    if (!c || !pinned_in) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (c->ring.empty()) return fail(&c->err, VH_ERR_STATE, "no ring: call vh_ring_create first");
    if (c->ring_used == (int)c->ring.size()) return fail(&c->err, VH_ERR_RING_FULL, "ring full (PILA LLENA)");
    *pinned_in = c->ring[c->ring_wr].h_in;
    return VH_OK;
}

int vh_ring_submit(vh_ctx* c, const float* in_host, int batch) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    if (c->ring.empty()) return fail(&c->err, VH_ERR_STATE, "no ring: call vh_ring_create first");
    if (!c->weights_ready) return fail(&c->err, VH_ERR_STATE, "submit before weights were loaded");
    if (batch < 1 || batch > c->ring_batch) return fail(&c->err, VH_ERR_INVALID, "batch %d outside 1..%d", batch, c->ring_batch);
    if (c->ring_used == (int)c->ring.size()) return fail(&c->err, VH_ERR_RING_FULL, "ring full (PILA LLENA)");
    HIPCHK(&c->err, hipSetDevice(c->device));
    vh_ctx::RingSlot& s = c->ring[c->ring_wr];
    const size_t in_bytes = (size_t)batch * c->cfg.image_size * c->cfg.image_size * c->cfg.channels * 4;
    if (in_host && in_host != s.h_in) memcpy(s.h_in, in_host, in_bytes);  // NULL / the slot's own buffer: already filled in place
    s.batch = batch;
    HIPCHK(&c->err, hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, c->copy_in));
    HIPCHK(&c->err, hipEventRecord(s.in_done, c->copy_in));
    HIPCHK(&c->err, hipStreamWaitEvent(c->stream, s.in_done, 0));
    int rc = run_step(c, s.d_in, batch, s.d_out);
    if (rc) return rc;
    HIPCHK(&c->err, hipEventRecord(s.fwd_done, c->stream));
    HIPCHK(&c->err, hipStreamWaitEvent(c->copy_out, s.fwd_done, 0));
    HIPCHK(&c->err, hipMemcpyAsync(s.h_out, s.d_out, (size_t)batch * c->cfg.classes * 4, hipMemcpyDeviceToHost, c->copy_out));
    HIPCHK(&c->err, hipEventRecord(s.out_done, c->copy_out));
    c->ring_wr = (c->ring_wr + 1) % (int)c->ring.size();
    ++c->ring_used;
    return VH_OK;
}

int vh_ring_collect(vh_ctx* c, float* logits_host, int* batch) {
    if (!c || !logits_host) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (c->ring.empty()) return fail(&c->err, VH_ERR_STATE, "no ring: call vh_ring_create first");
    if (c->ring_used == 0) return fail(&c->err, VH_ERR_RING_EMPTY, "ring empty (PILA VACIA)");
    HIPCHK(&c->err, hipSetDevice(c->device));
    vh_ctx::RingSlot& s = c->ring[c->ring_rd];
    HIPCHK(&c->err, hipEventSynchronize(s.out_done));
    memcpy(logits_host, s.h_out, (size_t)s.batch * c->cfg.classes * 4);
    if (batch) *batch = s.batch;
    c->ring_rd = (c->ring_rd + 1) % (int)c->ring.size();
    --c->ring_used;
    return VH_OK;
}

int vh_set_streams(vh_ctx* c, int n) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    if (n < 1 || n > vh_ctx::kMaxStreams) return fail(&c->err, VH_ERR_INVALID, "streams must be 1..%d", vh_ctx::kMaxStreams);
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    c->nstreams = n;
    drop_graphs(c);
    return VH_OK;
}

int vh_set_graph(vh_ctx* c, int enable) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    c->use_graph = enable != 0;
    if (!c->use_graph) drop_graphs(c);
    return VH_OK;
}

int vh_get_graph(const vh_ctx* c, int* enabled, int* cached) {
    if (!c || !enabled) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *enabled = c->use_graph ? 1 : 0;
    if (cached) *cached = (int)c->graphs.size();
    return VH_OK;
}

int vh_get_streams(const vh_ctx* c, int* n) {
    if (!c || !n) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *n = c->nstreams;
    return VH_OK;
}

int vh_set_stage_timing(vh_ctx* c, int stage) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    if (stage < -1 || stage >= ST_COUNT) return fail(&c->err, VH_ERR_INVALID, "unknown stage %d", stage);
    c->timing_stage = stage;
    c->tev_used = 0;
    return VH_OK;
}

int vh_get_stage_timing(vh_ctx* c, double* avg_ms, double* min_ms, int* launches) {
    if (!c || !avg_ms || !launches) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    double sum = 0.0, mn = 1e30;
    int n = 0;
    for (size_t i = 0; i + 1 < c->tev_used; i += 2) {
        float t = 0.f;
        HIPCHK(&c->err, hipEventElapsedTime(&t, c->tev[i], c->tev[i + 1]));
        sum += t; if (t < mn) mn = t; ++n;
    }
    *avg_ms = n ? sum / n : 0.0;
    if (min_ms) *min_ms = n ? mn : 0.0;
    *launches = n;
    return VH_OK;
}

int vh_set_step_timing(vh_ctx* c, int enable) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    c->step_timing = enable != 0;
    c->sev_used = 0;
    return VH_OK;
}

int vh_get_step_timing(vh_ctx* c, double* step_ms, int max_steps, int* steps) {
    if (!c || !steps || (max_steps > 0 && !step_ms)) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    int n = 0;
    for (size_t i = 0; i + 1 < c->sev_used; ++i) {
        float t = 0.f;
        HIPCHK(&c->err, hipEventElapsedTime(&t, c->sev[i], c->sev[i + 1]));
        if (n < max_steps) step_ms[n] = t;
        ++n;
    }
    *steps = n;
    return VH_OK;
}

int vh_debug_read(vh_ctx* c, int what, float* host_out, size_t n_floats) {
    if (!c || !host_out) return fail(c ? &c->err : nullptr, VH_ERR_INVALID, "null argument");
    if (c->last_batch <= 0) return fail(&c->err, VH_ERR_STATE, "no forward has run");
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    const size_t D = c->cfg.dim;
    if (what == 0) {
        const size_t n = (size_t)c->last_batch * c->L.T * D;
        if (n_floats != n) return fail(&c->err, VH_ERR_INVALID, "expected %zu floats", n);
        const int nl = (c->run_layers < 0 || c->run_layers > c->cfg.layers) ? c->cfg.layers : c->run_layers;
        if (c->split && nl > 0) {   // the residual stream lives as two planes: x = hi (16 bit) + lo (one scaled e4m3 byte)
            auto f = [&](uint16_t b, int dt) {
                if (dt == VH_DTYPE_BF16) { uint32_t u = (uint32_t)b << 16; float v; memcpy(&v, &u, 4); return v; }
                _Float16 hv; memcpy(&hv, &b, 2); return (float)hv;
            };
            auto g = [&](uint8_t b) {   // OCP e4m3fn -> float
                const int e = (b >> 3) & 15, m = b & 7;
                float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
                return (b & 0x80) ? -v : v;
            };
            if (c->fp8) {   // e4m3 hi plane (the GEMM operand) + bf16 lo plane
                std::vector<uint8_t> hi(n);
                std::vector<uint16_t> lo(n);
                HIPCHK(&c->err, hipMemcpy(hi.data(), c->xn16, n, hipMemcpyDeviceToHost));
                HIPCHK(&c->err, hipMemcpy(lo.data(), c->xlo16, n * 2, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n; ++i) host_out[i] = g(hi[i]) + f(lo[i], VH_DTYPE_BF16);
                return VH_OK;
            }
            std::vector<uint16_t> hi(n);
            std::vector<uint8_t> lo(n);
            HIPCHK(&c->err, hipMemcpy(hi.data(), c->xn16, n * 2, hipMemcpyDeviceToHost));
            HIPCHK(&c->err, hipMemcpy(lo.data(), c->xlo16, n, hipMemcpyDeviceToHost));
            const float inv = c->dt16 == VH_DTYPE_BF16 ? 1.f / 32.f : 1.f / 256.f;   // the byte plane's scale (vh_common.h Lo8<T>)
            for (size_t i = 0; i < n; ++i) host_out[i] = f(hi[i], c->dt16) + g(lo[i]) * inv;
            return VH_OK;
        }
        HIPCHK(&c->err, hipMemcpy(host_out, c->x, n * 4, hipMemcpyDeviceToHost));
        return VH_OK;
    }
    if (what == 1) {
        const size_t n = (size_t)c->last_batch * D;
        if (n_floats != n) return fail(&c->err, VH_ERR_INVALID, "expected %zu floats", n);
        HIPCHK(&c->err, hipMemcpy(host_out, c->clsn32, n * 4, hipMemcpyDeviceToHost));
        return VH_OK;
    }
    if (what == 2) {   // how many residual GEMMs of the last forward ran as a split launch (VH_TAIL_OVERLAP)
        if (n_floats != 1) return fail(&c->err, VH_ERR_INVALID, "expected 1 float");
        host_out[0] = (float)c->tail_splits;
        return VH_OK;
    }
    if (what == 3) {   // 1 when the last forward kept the MLP hidden activation in its tiled layout (h_tiled)
        if (n_floats != 1) return fail(&c->err, VH_ERR_INVALID, "expected 1 float");
        host_out[0] = c->last_h_tiled ? 1.f : 0.f;
        return VH_OK;
    }
    if (what == 4) {   // 1 when the last forward kept q|k|v head-major (qkv_hm)
        if (n_floats != 1) return fail(&c->err, VH_ERR_INVALID, "expected 1 float");
        host_out[0] = c->last_qkv_hm ? 1.f : 0.f;
        return VH_OK;
    }
    return fail(&c->err, VH_ERR_INVALID, "unknown tap %d", what);
}

int vh_debug_set_layers(vh_ctx* c, int n_layers) {
    if (!c) return fail(nullptr, VH_ERR_INVALID, "null context");
    c->run_layers = n_layers;
    if (c->stream) hipStreamSynchronize(c->stream);
    drop_graphs(c);
    return VH_OK;
}

// ---- operator-level entry points ------------------------------------------------------------------
#define OPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(nullptr, VH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

int vh_op_gemm(const void* a, const void* w, const float* bias, void* out, int64_t M, int N, int K, int epi,
               const float* aux, int aux_i, int dtype, int variant, void* stream) {
    GemmArgs g{a, w, bias, out, M, N, K, epi, aux, aux_i, dtype, variant};
    if (const char* why = gemm_check(g)) return fail(nullptr, VH_ERR_INVALID, "%s", why);
    OPCHK(launch_gemm(g, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_gemm_fp8(const void* a8, const void* w8, const float* w_scale, const float* bias, void* out, int64_t M, int N, int K,
                   int epi, int variant, void* stream) {
    if (variant != 0 && variant != 5 && variant != 6 && variant != 7) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8: variant must be 0, 5, 6 or 7");
    if (!a8 || !w8 || !w_scale || !bias || !out) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 4) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8: need K %% 128 == 0 and N %% 4 == 0");
    if (M > 0x7FFFFFFF / 2) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8: M too large");
    if (epi != VH_EPI_BIAS && epi != VH_EPI_BIAS_GELU && epi != VH_EPI_BIAS_RESID && epi != VH_EPI_BIAS_F32)
        return fail(nullptr, VH_ERR_UNSUPPORTED, "gemm_fp8: epilogue %d not available", epi);
    GemmArgs g{a8, w8, bias, out, M, N, K, epi, w_scale, 0, VH_DTYPE_FP8, variant};
    OPCHK(launch_gemm_fp8(g, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_gemm_fp8_ex(const void* a8, const void* w8, const float* w_scale, const float* bias, void* out, int64_t M, int N, int K,
                      int epi, const float* c_dev, const float* stats, void* out16, float* partials, int variant, void* stream) {
    if (variant != 0 && variant != 5 && variant != 6 && variant != 7) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: variant must be 0, 5, 6 or 7");
    if (!a8 || !w8 || !w_scale || !bias || !out) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: null pointer");
    if (M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 4) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: need K %% 128 == 0 and N %% 4 == 0");
    if (M > 0x7FFFFFFF / 2) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: M too large");
    const bool fold = epi == VH_EPI_LNFOLD || epi == VH_EPI_LNFOLD_GELU, resid = epi == VH_EPI_RESID_LN || epi == VH_EPI_RESID_SPLIT;
    if (!fold && !resid) return fail(nullptr, VH_ERR_UNSUPPORTED, "gemm_fp8_ex: epilogue %d not available", epi);
    if (fold && (!c_dev || !stats)) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: LNFOLD needs c and stats");
    if (resid && (!out16 || !partials)) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: RESID_* needs out16 and partials");
    if ((resid || epi == VH_EPI_LNFOLD_GELU) && N % 256) return fail(nullptr, VH_ERR_INVALID, "gemm_fp8_ex: N %% 256 != 0");
    GemmArgs g{a8, w8, bias, out, M, N, K, epi, fold ? c_dev : w_scale, 0, VH_DTYPE_FP8, variant};
    if (fold) g.wscale = w_scale;
    g.stats = stats; g.out16 = out16; g.partials = partials;
    OPCHK(launch_gemm_fp8(g, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_quantize_rows(const float* w, int rows, int cols, float post_scale, void* w8, float* scale, void* stream) {
    if (!w || !w8 || !scale || rows <= 0 || cols <= 0 || cols % 4) return fail(nullptr, VH_ERR_INVALID, "quantize_rows: bad argument");
    OPCHK(launch_quantize_rows(w, rows, cols, post_scale, w8, scale, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_gemm_ex(const void* a, const void* w, const float* bias, void* out, int64_t M, int N, int K, int epi,
                  const float* aux, int aux_i, const float* stats, void* out16, float* partials, int dtype, int variant,
                  void* stream) {
    GemmArgs g{a, w, bias, out, M, N, K, epi, aux, aux_i, dtype, variant};
    g.stats = stats; g.out16 = out16; g.partials = partials;
    if (const char* why = gemm_check(g)) return fail(nullptr, VH_ERR_INVALID, "%s", why);
    OPCHK(launch_gemm(g, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_rowstats_cast(const float* x, int64_t rows, int dim, float eps, void* x16, float* stats, int dtype, void* stream) {
    if (!x || !x16 || !stats || rows <= 0 || dim <= 0 || dim % 4 || dim > 2048) return fail(nullptr, VH_ERR_INVALID, "rowstats_cast: bad argument");
    OPCHK(launch_rowstats_cast(x, rows, dim, eps, x16, stats, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_rowstats_split(const float* x, int64_t rows, int dim, float eps, void* hi, void* lo, float* stats, int dtype, void* stream) {
    if (!x || !hi || !lo || !stats || rows <= 0 || dim <= 0 || dim % 4 || dim > 2048) return fail(nullptr, VH_ERR_INVALID, "rowstats_split: bad argument");
    OPCHK(launch_rowstats_split(x, rows, dim, eps, hi, lo, stats, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_finalize_stats(const float* partials, int nblk, int64_t rows, int dim, float eps, float* stats, void* stream) {
    if (!partials || !stats || nblk <= 0 || rows <= 0 || dim <= 0) return fail(nullptr, VH_ERR_INVALID, "finalize_stats: bad argument");
    OPCHK(launch_finalize_stats(partials, nblk, rows, dim, eps, stats, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_fold_ln(const float* w, const float* b, const float* gamma, const float* beta, int rows, int dim, float scale,
                  void* w16, float* c, float* d, int dtype, void* stream) {
    if (!w || !b || !gamma || !beta || !w16 || !c || !d || rows <= 0 || dim <= 0) return fail(nullptr, VH_ERR_INVALID, "fold_ln: bad argument");
    OPCHK(launch_fold_ln(w, b, gamma, beta, rows, dim, scale, w16, c, d, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_layernorm(const float* x, int64_t rows, int dim, int64_t row_stride, const float* gamma, const float* beta,
                    float eps, void* out16, int dtype, void* stream) {
    if (!x || !gamma || !beta || !out16) return fail(nullptr, VH_ERR_INVALID, "null pointer");
    if (dim <= 0 || dim % 4 || dim > 2048 || rows <= 0 || row_stride % 4) return fail(nullptr, VH_ERR_INVALID, "layernorm: unsupported shape");
    OPCHK(launch_layernorm(x, rows, dim, row_stride, gamma, beta, eps, out16, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_attention(const void* qkv16, int batch, int tokens, int heads, void* out16, int dtype, void* stream) {
    if (!qkv16 || !out16) return fail(nullptr, VH_ERR_INVALID, "null pointer");
    if (batch <= 0 || tokens <= 0 || heads <= 0 || attention_lds_bytes(tokens) > 160 * 1024)
        return fail(nullptr, VH_ERR_INVALID, "attention: unsupported shape");
    // The work-queue counter is owned by THIS call (allocated, used, freed): taps may run concurrently from several host
    // threads, and a context carries its own counters in its arena.
    unsigned int* ticket = nullptr;
    OPCHK(hipMalloc((void**)&ticket, 256));
    hipError_t e = launch_attention(qkv16, batch, tokens, heads, out16, dtype, ticket, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    hipFree(ticket);
    if (e != hipSuccess) return fail(nullptr, VH_ERR_HIP, "attention failed: %s", hipGetErrorString(e));
    return VH_OK;
}
int vh_op_im2col(const float* in, int batch, int image, int patch, int channels, void* out16, int dtype, void* stream) {
    if (!in || !out16) return fail(nullptr, VH_ERR_INVALID, "null pointer");
    if (batch <= 0 || patch <= 0 || image <= 0 || image % patch || (patch * channels) % 4)
        return fail(nullptr, VH_ERR_INVALID, "im2col: unsupported shape");
    OPCHK(launch_im2col(in, batch, image, patch, channels, out16, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_cast(const float* in, void* out16, int64_t n, int dtype, void* stream) {
    if (!in || !out16 || n <= 0 || n % 4) return fail(nullptr, VH_ERR_INVALID, "cast: bad argument");
    OPCHK(launch_cast(in, out16, n, dtype, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}
int vh_op_fill(float* out, int64_t n, uint64_t seed, uint32_t tensor_id, int kind, float sigma, void* stream) {
    if (!out || n <= 0 || kind < 0 || kind > 2) return fail(nullptr, VH_ERR_INVALID, "fill: bad argument");
    OPCHK(launch_fill(out, n, seed, tensor_id, kind, kind == 2 ? 0.f : sigma, kind == 2 ? sigma : 0.f, (hipStream_t)stream));
    OPCHK(hipStreamSynchronize((hipStream_t)stream));
    return VH_OK;
}

// micro-benchmark of one GEMM shape: synthetic operands generated in HBM, `iters` back-to-back
// launches between two hip events; returns the average launch time
int vh_bench_gemm(int device, int64_t M, int N, int K, int epilogue, int dtype, int variant, int iters, double* avg_ms) {
    if (!avg_ms || iters <= 0) return fail(nullptr, VH_ERR_INVALID, "vh_bench_gemm: bad argument");
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    float *a32 = nullptr, *w32 = nullptr, *bias = nullptr, *aux = nullptr, *stats = nullptr, *partials = nullptr;
    void *a16 = nullptr, *w16 = nullptr, *out = nullptr, *out16 = nullptr;
    const int aux_i = 196;
    // the layer epilogues of the folded path: LNFOLD* read per-row (mean, rstd) and c_n (`aux`), RESID_LN / RESID_SPLIT
    // write a second 16-bit plane and the per-64-column row sums
    const bool fold = epilogue == VH_EPI_LNFOLD || epilogue == VH_EPI_LNFOLD_GELU;
    const bool resid2 = epilogue == VH_EPI_RESID_LN || epilogue == VH_EPI_RESID_SPLIT || epilogue == VH_EPI_PATCH_SPLIT;
    const size_t out_rows = (epilogue == VH_EPI_PATCH || epilogue == VH_EPI_PATCH_SPLIT) ? (size_t)(M / aux_i + 1) * (aux_i + 1) : (size_t)M;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&]() {
        hipFree(a32); hipFree(w32); hipFree(bias); hipFree(aux); hipFree(a16); hipFree(w16); hipFree(out);
        hipFree(stats); hipFree(partials); hipFree(out16);
        if (e0) hipEventDestroy(e0);
        if (e1) hipEventDestroy(e1);
    };
#define BCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(nullptr, VH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    BCHK(hipMalloc((void**)&a32, (size_t)M * K * 4));
    BCHK(hipMalloc((void**)&w32, (size_t)N * K * 4));
    BCHK(hipMalloc((void**)&bias, (size_t)N * 4));
    BCHK(hipMalloc((void**)&aux, (size_t)(aux_i + 1) * N * 4));
    BCHK(hipMalloc(&a16, (size_t)M * K * 2));
    BCHK(hipMalloc(&w16, (size_t)N * K * 2));
    BCHK(hipMalloc(&out, out_rows * N * 4));
    BCHK(hipMemset(out, 0, out_rows * N * 4));
    BCHK(launch_fill(a32, (int64_t)M * K, 1, 1, 0, 0.f, 0.f, nullptr));
    BCHK(launch_fill(w32, (int64_t)N * K, 1, 2, 1, 0.02f, 0.f, nullptr));
    BCHK(launch_fill(bias, N, 1, 3, 1, 0.02f, 0.f, nullptr));
    BCHK(launch_fill(aux, (int64_t)(aux_i + 1) * N, 1, 4, 1, 0.02f, 0.f, nullptr));
    BCHK(launch_cast(a32, a16, (int64_t)M * K, dtype, nullptr));
    BCHK(launch_cast(w32, w16, (int64_t)N * K, dtype, nullptr));
    GemmArgs g{a16, w16, bias, out, M, N, K, epilogue, aux, aux_i, dtype, variant};
    if (fold) {   // (mean, rstd) ~ (0.02 sigma, 1 + 0.02 sigma): the values do not matter for the time
        BCHK(hipMalloc((void**)&stats, (size_t)M * 2 * 4));
        BCHK(launch_fill(stats, (int64_t)M * 2, 1, 5, 1, 0.02f, 0.f, nullptr));
        g.stats = stats;
    }
    if (resid2) {
        BCHK(hipMalloc(&out16, out_rows * N * 2));
        BCHK(hipMemset(out16, 0, out_rows * N * 2));
        BCHK(hipMalloc((void**)&partials, (size_t)((N + 63) / 64) * out_rows * 2 * 4));
        if (epilogue == VH_EPI_PATCH_SPLIT) g.prow = (int64_t)out_rows;
        g.out16 = out16;
        g.partials = partials;
    }
    if (const char* why = gemm_check(g)) { cleanup(); return fail(nullptr, VH_ERR_INVALID, "%s", why); }
    BCHK(hipEventCreate(&e0));
    BCHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) BCHK(launch_gemm(g, nullptr));
    BCHK(hipDeviceSynchronize());
    BCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) BCHK(launch_gemm(g, nullptr));
    BCHK(hipEventRecord(e1, nullptr));
    BCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    BCHK(hipEventElapsedTime(&ms, e0, e1));
#undef BCHK
    *avg_ms = ms / iters;
    cleanup();
    return VH_OK;
}

// ---- MLP mode ------------------------------------------------------------------------------------
// ---- filter_image pipeline --------------------------------------------------------------------------------------
// submit: H2D of one frame -> 3x3 filter -> D2H, all asynchronous, chained by events (the reference's
// clEnqueueWriteBuffer -> clEnqueueTask -> clEnqueueReadBuffer chain, netFPGA.cpp:323-329); collect: wait for the
// OLDEST frame (clWaitForEvents on g_im_read_event[rd], :350).  Full / empty ring = VH_ERR_RING_FULL / _EMPTY.
int vh_filter_destroy(vh_filter* f) {
    if (!f) return VH_OK;
    hipSetDevice(f->device);
    if (f->copy_in) hipStreamSynchronize(f->copy_in);
    if (f->compute) hipStreamSynchronize(f->compute);
    if (f->copy_out) hipStreamSynchronize(f->copy_out);
    for (auto& s : f->slot) {
        if (s.h_in) hipHostFree(s.h_in);
        if (s.h_out) hipHostFree(s.h_out);
        if (s.d_in) hipFree(s.d_in);
        if (s.d_out) hipFree(s.d_out);
        if (s.in_done) hipEventDestroy(s.in_done);
        if (s.k_done) hipEventDestroy(s.k_done);
        if (s.out_done) hipEventDestroy(s.out_done);
    }
    if (f->copy_in) hipStreamDestroy(f->copy_in);
    if (f->compute) hipStreamDestroy(f->compute);
    if (f->copy_out) hipStreamDestroy(f->copy_out);
    delete f;
    return VH_OK;
}

int vh_filter_create(int device, int height, int width, int slots, int kind, vh_filter** out) {
    if (!out) return fail(nullptr, VH_ERR_INVALID, "vh_filter_create: null argument");
    *out = nullptr;
    if (height <= 0 || width <= 0 || height > 16384 || width > 16384) return fail(nullptr, VH_ERR_INVALID, "vh_filter_create: frame size outside 1..16384");
    if (slots < 1 || slots > 64) return fail(nullptr, VH_ERR_INVALID, "vh_filter_create: slots must be 1..64");
    if (kind != VH_FILTER_BLUR3 && kind != VH_FILTER_SOBEL3) return fail(nullptr, VH_ERR_INVALID, "vh_filter_create: unknown filter kind %d", kind);
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    vh_filter* f = new vh_filter();
    f->device = device; f->height = height; f->width = width; f->kind = kind;
    const size_t n = (size_t)height * width;
    hipError_t e = hipStreamCreateWithFlags(&f->copy_in, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->compute, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&f->copy_out, hipStreamNonBlocking);
    f->slot.resize(slots);
    for (auto& s : f->slot) {
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_in, n, hipHostMallocDefault);
        if (e == hipSuccess) e = hipHostMalloc((void**)&s.h_out, n, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_in, n);
        if (e == hipSuccess) e = hipMalloc((void**)&s.d_out, n);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.in_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.k_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.out_done, hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        vh_filter_destroy(f);
        return fail(nullptr, VH_ERR_HIP, "vh_filter_create: %s", hipGetErrorString(e));
    }
    *out = f;
    return VH_OK;
}

const char* vh_filter_last_error(const vh_filter* f) { return f ? f->err.c_str() : g_err.c_str(); }

int vh_filter_free_slots(const vh_filter* f, int* n) {
    if (!f || !n) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *n = (int)f->slot.size() - f->used;
    return VH_OK;
}

int vh_filter_submit(vh_filter* f, const uint8_t* frame) {
    if (!f || !frame) return fail(f ? &f->err : nullptr, VH_ERR_INVALID, "null argument");
    if (f->used == (int)f->slot.size()) return fail(&f->err, VH_ERR_RING_FULL, "ring full (PILA LLENA)");
    HIPCHK(&f->err, hipSetDevice(f->device));
    vh_filter::Slot& s = f->slot[f->wr];
    const size_t n = (size_t)f->height * f->width;
    memcpy(s.h_in, frame, n);
    HIPCHK(&f->err, hipMemcpyAsync(s.d_in, s.h_in, n, hipMemcpyHostToDevice, f->copy_in));
    HIPCHK(&f->err, hipEventRecord(s.in_done, f->copy_in));
    HIPCHK(&f->err, hipStreamWaitEvent(f->compute, s.in_done, 0));
    HIPCHK(&f->err, launch_filter3x3(s.d_in, s.d_out, f->height, f->width, f->kind, f->compute));
    HIPCHK(&f->err, hipEventRecord(s.k_done, f->compute));
    HIPCHK(&f->err, hipStreamWaitEvent(f->copy_out, s.k_done, 0));
    HIPCHK(&f->err, hipMemcpyAsync(s.h_out, s.d_out, n, hipMemcpyDeviceToHost, f->copy_out));
    HIPCHK(&f->err, hipEventRecord(s.out_done, f->copy_out));
    f->wr = (f->wr + 1) % (int)f->slot.size();
    ++f->used;
    return VH_OK;
}

int vh_filter_collect(vh_filter* f, uint8_t* frame) {
    if (!f || !frame) return fail(f ? &f->err : nullptr, VH_ERR_INVALID, "null argument");
    if (f->used == 0) return fail(&f->err, VH_ERR_RING_EMPTY, "ring empty (PILA VACIA)");
    HIPCHK(&f->err, hipSetDevice(f->device));
    vh_filter::Slot& s = f->slot[f->rd];
    HIPCHK(&f->err, hipEventSynchronize(s.out_done));
    memcpy(frame, s.h_out, (size_t)f->height * f->width);
    f->rd = (f->rd + 1) % (int)f->slot.size();
    --f->used;
    return VH_OK;
}

int vh_mlp_create(int device, int n_ins, int n_layers, const int* n_p_l, int activation, vh_mlp** out) {
    if (!out || !n_p_l || n_ins <= 0 || n_layers <= 0) return fail(nullptr, VH_ERR_INVALID, "vh_mlp_create: bad argument");
    if (activation < VH_ACT_IDENTITY || activation > VH_ACT_GELU) return fail(nullptr, VH_ERR_INVALID, "unknown activation %d", activation);
    *out = nullptr;
    int rc = set_device(nullptr, device);
    if (rc) return rc;
    vh_mlp* m = new vh_mlp();
    m->device = device; m->n_ins = n_ins; m->n_layers = n_layers; m->activation = activation;
    m->widest = n_ins;
    int fan = n_ins;
    for (int l = 0; l < n_layers; ++l) {
        if (n_p_l[l] <= 0) { delete m; return fail(nullptr, VH_ERR_INVALID, "n_p_l[%d] must be positive", l); }
        m->npl.push_back(n_p_l[l]);
        m->n_params += (size_t)n_p_l[l] * fan;   // netFPGA.cpp:68-76
        m->n_neurons += (size_t)n_p_l[l];
        fan = n_p_l[l];
        if (fan > m->widest) m->widest = fan;
    }
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&m->params, m->n_params * 4);
    if (e == hipSuccess) e = hipMalloc((void**)&m->bias, m->n_neurons * 4);
    if (e != hipSuccess) { vh_mlp_destroy(m); return fail(nullptr, VH_ERR_HIP, "vh_mlp_create: %s", hipGetErrorString(e)); }
    *out = m;
    return VH_OK;
}

int vh_mlp_load_params(vh_mlp* m, const float* params, size_t n_params, const float* bias, size_t n_neurons) {
    if (!m || !params || !bias) return fail(m ? &m->err : nullptr, VH_ERR_INVALID, "null argument");
    if (n_params != m->n_params || n_neurons != m->n_neurons)
        return fail(&m->err, VH_ERR_INVALID, "expected %zu params / %zu neurons, got %zu / %zu", m->n_params, m->n_neurons, n_params, n_neurons);
    HIPCHK(&m->err, hipSetDevice(m->device));
    // same order as the reference: params, then bias (netFPGA.cpp:506-509)
    HIPCHK(&m->err, hipMemcpyAsync(m->params, params, n_params * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(&m->err, hipMemcpyAsync(m->bias, bias, n_neurons * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(&m->err, hipStreamSynchronize(m->stream));
    m->loaded = true;
    return VH_OK;
}

int vh_mlp_forward(vh_mlp* m, const float* in, int n_vec, float* outp) {
    if (!m || !in || !outp || n_vec <= 0) return fail(m ? &m->err : nullptr, VH_ERR_INVALID, "bad argument");
    if (!m->loaded) return fail(&m->err, VH_ERR_STATE, "forward before params were loaded");
    const auto t0 = std::chrono::high_resolution_clock::now();
    HIPCHK(&m->err, hipSetDevice(m->device));
    if (n_vec > m->max_vec) {
        if (m->buf0) hipFree(m->buf0);
        if (m->buf1) hipFree(m->buf1);
        m->buf0 = m->buf1 = nullptr; m->max_vec = 0;
        HIPCHK(&m->err, hipMalloc((void**)&m->buf0, (size_t)n_vec * m->widest * 4));
        HIPCHK(&m->err, hipMalloc((void**)&m->buf1, (size_t)n_vec * m->widest * 4));
        m->max_vec = n_vec;
    }
    HIPCHK(&m->err, hipMemcpyAsync(m->buf0, in, (size_t)n_vec * m->n_ins * 4, hipMemcpyHostToDevice, m->stream));
    float *cur = m->buf0, *nxt = m->buf1;
    size_t woff = 0, boff = 0;
    int fan = m->n_ins;
    for (int l = 0; l < m->n_layers; ++l) {
        HIPCHK(&m->err, launch_dense_layer(m->params + woff, m->bias + boff, cur, nxt, fan, m->npl[l], n_vec, m->activation, m->stream));
        woff += (size_t)m->npl[l] * fan; boff += (size_t)m->npl[l]; fan = m->npl[l];
        std::swap(cur, nxt);
    }
    HIPCHK(&m->err, hipMemcpyAsync(outp, cur, (size_t)n_vec * fan * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(&m->err, hipStreamSynchronize(m->stream));
    m->last_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return VH_OK;
}

// ---- MLP-mode training (netFPGA.cpp:518-580 is commented-out code: the definitions are in include/vithip.h)
int vh_mlp_init_gradient(vh_mlp* m, const float* set_ins, const float* set_outs, int n_sets) {
    if (!m || !set_ins || !set_outs || n_sets <= 0) return fail(m ? &m->err : nullptr, VH_ERR_INVALID, "vh_mlp_init_gradient: bad argument");
    if (!m->loaded) return fail(&m->err, VH_ERR_STATE, "init_gradient before params were loaded");
    if (n_sets > 65535) return fail(&m->err, VH_ERR_INVALID, "at most 65535 sets");
    for (int n : m->npl) if (n > 65535) return fail(&m->err, VH_ERR_INVALID, "training supports layers of at most 65535 neurons");
    HIPCHK(&m->err, hipSetDevice(m->device));
    for (float** p : {&m->set_ins, &m->set_outs, &m->tz, &m->ta, &m->td, &m->terr}) { if (*p) hipFree(*p); *p = nullptr; }
    m->n_sets = 0;
    const size_t n_out = (size_t)m->npl.back();
    HIPCHK(&m->err, hipMalloc((void**)&m->set_ins, (size_t)n_sets * m->n_ins * 4));
    HIPCHK(&m->err, hipMalloc((void**)&m->set_outs, (size_t)n_sets * n_out * 4));
    HIPCHK(&m->err, hipMalloc((void**)&m->tz, m->n_neurons * n_sets * 4));
    HIPCHK(&m->err, hipMalloc((void**)&m->ta, m->n_neurons * n_sets * 4));
    HIPCHK(&m->err, hipMalloc((void**)&m->td, m->n_neurons * n_sets * 4));
    HIPCHK(&m->err, hipMalloc((void**)&m->terr, 4));
    HIPCHK(&m->err, hipMemcpyAsync(m->set_ins, set_ins, (size_t)n_sets * m->n_ins * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(&m->err, hipMemcpyAsync(m->set_outs, set_outs, (size_t)n_sets * n_out * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(&m->err, hipStreamSynchronize(m->stream));
    m->n_sets = n_sets;
    return VH_OK;
}

int vh_mlp_launch_gradient(vh_mlp* m, int iterations, float error_threshold, float multiplier, float* errors) {
    if (!m || iterations < 0 || (iterations > 0 && !errors)) return fail(m ? &m->err : nullptr, VH_ERR_INVALID, "vh_mlp_launch_gradient: bad argument");
    if (m->n_sets <= 0) return fail(&m->err, VH_ERR_STATE, "launch_gradient before init_gradient");
    const auto t0 = std::chrono::high_resolution_clock::now();
    HIPCHK(&m->err, hipSetDevice(m->device));
    const int S = m->n_sets, L = m->n_layers;
    const float scale = multiplier / (float)S;
    // layer l's slices of the per-set buffers start at noff[l] * S floats
    std::vector<size_t> woff(L), noff(L);
    { size_t w = 0, n = 0; int fan = m->n_ins; for (int l = 0; l < L; ++l) { woff[l] = w; noff[l] = n; w += (size_t)m->npl[l] * fan; n += (size_t)m->npl[l]; fan = m->npl[l]; } }
    auto fan_in = [&](int l) { return l == 0 ? m->n_ins : m->npl[l - 1]; };
    auto act_in = [&](int l) -> const float* { return l == 0 ? m->set_ins : m->ta + noff[l - 1] * S; };
    for (int it = 0; it < iterations; ++it) errors[it] = 0.f;
    for (int it = 0; it < iterations; ++it) {
        for (int l = 0; l < L; ++l)
            HIPCHK(&m->err, launch_dense_layer(m->params + woff[l], m->bias + noff[l], act_in(l), m->ta + noff[l] * S, fan_in(l), m->npl[l], S,
                                               m->activation, m->stream, m->tz + noff[l] * S));
        const int n_out = m->npl[L - 1];
        HIPCHK(&m->err, launch_mlp_out_delta(m->ta + noff[L - 1] * S, m->tz + noff[L - 1] * S, m->set_outs, m->td + noff[L - 1] * S,
                                             (int64_t)S * n_out, m->activation, m->terr, m->stream));
        // the error decides whether this iteration updates: one 4-byte read back per iteration (the reference's loop is
        // host-driven as well, netFPGA.cpp:552-565)
        HIPCHK(&m->err, hipMemcpyAsync(&errors[it], m->terr, 4, hipMemcpyDeviceToHost, m->stream));
        HIPCHK(&m->err, hipStreamSynchronize(m->stream));
        if (errors[it] <= error_threshold) break;
        for (int l = L - 1; l >= 1; --l)   // deltas back to front, all with the parameters of THIS iteration
            HIPCHK(&m->err, launch_mlp_back_delta(m->params + woff[l], m->td + noff[l] * S, m->tz + noff[l - 1] * S, m->td + noff[l - 1] * S,
                                                  m->npl[l - 1], m->npl[l], S, m->activation, m->stream));
        for (int l = 0; l < L; ++l)
            HIPCHK(&m->err, launch_mlp_update(m->params + woff[l], m->bias + noff[l], m->td + noff[l] * S, act_in(l), fan_in(l), m->npl[l], S,
                                              scale, m->stream));
    }
    HIPCHK(&m->err, hipStreamSynchronize(m->stream));
    m->last_gradient_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
    return VH_OK;
}

int vh_mlp_read_params(vh_mlp* m, float* params, size_t n_params, float* bias, size_t n_neurons) {
    if (!m || !params || !bias) return fail(m ? &m->err : nullptr, VH_ERR_INVALID, "null argument");
    if (n_params != m->n_params || n_neurons != m->n_neurons)
        return fail(&m->err, VH_ERR_INVALID, "expected %zu params / %zu neurons, got %zu / %zu", m->n_params, m->n_neurons, n_params, n_neurons);
    if (!m->loaded) return fail(&m->err, VH_ERR_STATE, "no params loaded");
    HIPCHK(&m->err, hipSetDevice(m->device));
    HIPCHK(&m->err, hipMemcpyAsync(params, m->params, n_params * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(&m->err, hipMemcpyAsync(bias, m->bias, n_neurons * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(&m->err, hipStreamSynchronize(m->stream));
    return VH_OK;
}

int vh_mlp_last_gradient_us(const vh_mlp* m, int64_t* us) {
    if (!m || !us) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *us = m->last_gradient_us;
    return VH_OK;
}

int vh_mlp_last_forward_us(const vh_mlp* m, int64_t* us) {
    if (!m || !us) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *us = m->last_us;
    return VH_OK;
}
const char* vh_mlp_last_error(const vh_mlp* m) { return m ? m->err.c_str() : g_err.c_str(); }

int vh_mlp_destroy(vh_mlp* m) {
    if (!m) return VH_OK;
    hipSetDevice(m->device);
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->params) hipFree(m->params);
    if (m->bias) hipFree(m->bias);
    if (m->buf0) hipFree(m->buf0);
    if (m->buf1) hipFree(m->buf1);
    for (float* p : {m->set_ins, m->set_outs, m->tz, m->ta, m->td, m->terr}) if (p) hipFree(p);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
    return VH_OK;
}

}  // extern "C"

// ---- device group: N GPUs of one node from ONE process ---------------------------------------------
// The reference drives exactly one device (clGetDeviceIDs(ACCELERATOR), netFPGA.cpp:376) with one input per call
// (:266-277); nothing to mirror, so the contract is the survey's: single process, one host thread + one context (own
// stream) per device, ncclCommInitAll, ONE ncclBroadcast of the canonical weight blob (what _load_params uploads per
// device, netFPGA.cpp:484-515) from member 0, contiguous image ranges per member, per-member D2H straight into the
// caller's logits; no collective on the data path.  RCCL is bound at run time (dlopen) and only when a group has more
// than one distinct device, so libvithip.so itself carries no RCCL dependency.
struct vh_group {
    struct Member {
        int device = 0;
        vh_ctx* ctx = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;   // pending command (empty = none)
        bool quit = false, busy = false;
        int rc = VH_OK;
        float *in_dev = nullptr, *out_dev = nullptr;   // device-resident benchmark buffers
        int resident_batch = 0;
    };
    vh_config cfg;
    std::vector<std::unique_ptr<Member>> m;
    bool same_device = false;   // rehearsal: duplicate ordinals, broadcast by D2D copy instead of RCCL
    void* rccl = nullptr;       // dlopen handle
    std::vector<void*> comms;   // ncclComm_t per member
    std::string err;
};

namespace {

struct RcclApi {
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;

int bind_rccl(vh_group* g) {
    if (g->rccl) return VH_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g->rccl = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g->rccl) break; }
    if (!g->rccl) return fail(&g->err, VH_ERR_UNSUPPORTED, "vh_group: cannot load librccl.so (%s)", dlerror());
    auto sym = [&](const char* n) { return dlsym(g->rccl, n); };
    g_rccl.CommInitAll = (int (*)(void**, int, const int*))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    g_rccl.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))sym("ncclBroadcast");
    g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
    g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    if (!g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.Broadcast || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.GetErrorString)
        return fail(&g->err, VH_ERR_UNSUPPORTED, "vh_group: librccl.so lacks a required symbol");
    return VH_OK;
}

void member_loop(vh_group::Member* m) {
    hipSetDevice(m->device);
    std::unique_lock<std::mutex> lk(m->mu);
    for (;;) {
        m->cv.wait(lk, [&] { return m->quit || (bool)m->job; });
        if (m->quit) return;
        std::function<int()> job = std::move(m->job);
        m->job = nullptr;
        lk.unlock();
        const int rc = job();
        lk.lock();
        m->rc = rc;
        m->busy = false;
        m->cv.notify_all();
    }
}

// run one command per member concurrently, wait for all; returns the first failure
int run_all(vh_group* g, const std::function<int(int)>& cmd) {
    for (size_t i = 0; i < g->m.size(); ++i) {
        vh_group::Member* m = g->m[i].get();
        std::lock_guard<std::mutex> lk(m->mu);
        m->busy = true;
        m->job = [cmd, i]() { return cmd((int)i); };
        m->cv.notify_all();
    }
    int rc = VH_OK;
    for (size_t i = 0; i < g->m.size(); ++i) {
        vh_group::Member* m = g->m[i].get();
        std::unique_lock<std::mutex> lk(m->mu);
        m->cv.wait(lk, [&] { return !m->busy; });
        if (m->rc != VH_OK && rc == VH_OK) {
            rc = m->rc;
            g->err = std::string("member ") + std::to_string(i) + " (device " + std::to_string(m->device) + "): " + m->ctx->err;
        }
    }
    if (rc != VH_OK) g_err = g->err;
    return rc;
}

}  // namespace

extern "C" {

void vh_group_shard_bounds(int batch, int n, int r, int* lo, int* hi) {
    // contiguous image range of member r; the first (batch % n) members take one extra image
    const int base = batch / n, extra = batch % n;
    const int l = r * base + (r < extra ? r : extra);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (r < extra ? 1 : 0);
}

const char* vh_group_last_error(const vh_group* g) { return g ? g->err.c_str() : g_err.c_str(); }

int vh_group_destroy(vh_group* g) {
    if (!g) return VH_OK;
    for (auto& mp : g->m) {
        vh_group::Member* m = mp.get();
        if (m->th.joinable()) {
            { std::lock_guard<std::mutex> lk(m->mu); m->quit = true; m->cv.notify_all(); }
            m->th.join();
        }
    }
    for (size_t i = 0; i < g->comms.size(); ++i)
        if (g->comms[i] && g_rccl.CommDestroy) { hipSetDevice(g->m[i]->device); g_rccl.CommDestroy(g->comms[i]); }
    for (auto& mp : g->m) {
        hipSetDevice(mp->device);
        if (mp->in_dev) hipFree(mp->in_dev);
        if (mp->out_dev) hipFree(mp->out_dev);
        if (mp->ctx) vh_destroy(mp->ctx);
    }
    delete g;
    return VH_OK;
}

int vh_group_create(const vh_config* cfg, const int* devices, int n, vh_group** out) {
    if (!cfg || !devices || !out || n < 1 || n > 64) return fail(nullptr, VH_ERR_INVALID, "vh_group_create: bad argument");
    *out = nullptr;
    vh_group* g = new vh_group();
    g->cfg = *cfg;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) g->same_device |= devices[i] == devices[j];
    for (int i = 0; i < n; ++i) {
        std::unique_ptr<vh_group::Member> m(new vh_group::Member());
        m->device = devices[i];
        const int rc = vh_create(cfg, devices[i], &m->ctx);
        g->m.push_back(std::move(m));
        if (rc != VH_OK) { vh_group_destroy(g); return rc; }
    }
    // RCCL serves groups of DISTINCT devices only (a communicator cannot hold one device twice): a group that lists an
    // ordinal more than once -- the one-GPU rehearsal of the N > 1 code path -- always broadcasts by device-to-device copy.
    // VH_GROUP_FORCE_RCCL=1 makes a group of ONE go through ncclCommInitAll / ncclBroadcast as well (a smoke test of the
    // run-time binding); it has no effect on a rehearsal group.
    const char* force = getenv("VH_GROUP_FORCE_RCCL");
    if ((n > 1 && !g->same_device) || (force && force[0] == '1' && !g->same_device)) {
        int rc = bind_rccl(g);
        if (rc == VH_OK) {
            g->comms.assign(n, nullptr);
            const int r = g_rccl.CommInitAll(g->comms.data(), n, devices);
            if (r != 0) rc = fail(&g->err, VH_ERR_HIP, "ncclCommInitAll: %s", g_rccl.GetErrorString(r));
        }
        if (rc != VH_OK) { const std::string e = g->err; vh_group_destroy(g); g_err = e; return rc; }
    }
    for (auto& mp : g->m) mp->th = std::thread(member_loop, mp.get());
    *out = g;
    return VH_OK;
}

int vh_group_size(const vh_group* g, int* n) {
    if (!g || !n) return fail(nullptr, VH_ERR_INVALID, "null argument");
    *n = (int)g->m.size();
    return VH_OK;
}

int vh_group_member(vh_group* g, int i, vh_ctx** ctx, int* device) {
    if (!g || i < 0 || i >= (int)g->m.size()) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "vh_group_member: bad index");
    if (ctx) *ctx = g->m[i]->ctx;
    if (device) *device = g->m[i]->device;
    return VH_OK;
}

// member 0's resident canonical blob -> every other member's blob buffer (one ncclBroadcast over xGMI), then every other
// member converts it to its compute layout (vh_load_weights_device, the same code path a blob from an RCCL broadcast of
// the multi-process form takes)
int vh_group_broadcast_weights(vh_group* g) {
    if (!g) return fail(nullptr, VH_ERR_INVALID, "null group");
    vh_ctx* root = g->m[0]->ctx;
    if (!root->weights_ready) return fail(&g->err, VH_ERR_STATE, "vh_group_broadcast_weights: member 0 has no weights");
    const size_t nbytes = sizeof(BlobHeader) + 4 * root->L.total;
    const int n = (int)g->m.size();
    if (!g->comms.empty()) {
        HIPCHK(&g->err, hipSetDevice(g->m[0]->device));
        HIPCHK(&g->err, hipStreamSynchronize(root->stream));
        // Whatever fails between GroupStart and GroupEnd, the group is still ENDED before this function returns: leaving it
        // open would leave the other members' streams with a half-enqueued collective.
        int r = g_rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (int i = 0; i < n && r == 0 && he == hipSuccess; ++i) {
            he = hipSetDevice(g->m[i]->device);
            if (he != hipSuccess) break;
            vh_ctx* c = g->m[i]->ctx;
            r = g_rccl.Broadcast(root->blob, c->blob, nbytes, /*ncclUint8*/ 1, 0, g->comms[i], c->stream);
        }
        const int r2 = g_rccl.GroupEnd();
        if (he != hipSuccess) return fail(&g->err, VH_ERR_HIP, "vh_group_broadcast_weights: hipSetDevice: %s", hipGetErrorString(he));
        if (r == 0) r = r2;
        if (r != 0) return fail(&g->err, VH_ERR_HIP, "ncclBroadcast: %s", g_rccl.GetErrorString(r));
        for (int i = 0; i < n; ++i) {
            HIPCHK(&g->err, hipSetDevice(g->m[i]->device));
            HIPCHK(&g->err, hipStreamSynchronize(g->m[i]->ctx->stream));
        }
    } else if (n > 1) {   // rehearsal group on one device: the broadcast is a device-to-device copy
        HIPCHK(&g->err, hipSetDevice(g->m[0]->device));
        HIPCHK(&g->err, hipStreamSynchronize(root->stream));
        for (int i = 1; i < n; ++i) HIPCHK(&g->err, hipMemcpy(g->m[i]->ctx->blob, root->blob, nbytes, hipMemcpyDeviceToDevice));
    }
    return run_all(g, [g, nbytes](int i) { return i == 0 ? VH_OK : vh_load_weights_device(g->m[i]->ctx, g->m[i]->ctx->blob, nbytes); });
}

int vh_group_load_weights(vh_group* g, const void* host_blob, size_t nbytes) {
    if (!g || !host_blob) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "null argument");
    int rc = vh_load_weights(g->m[0]->ctx, host_blob, nbytes);
    if (rc != VH_OK) { g->err = g->m[0]->ctx->err; return rc; }
    return vh_group_broadcast_weights(g);
}

int vh_group_init_weights_seeded(vh_group* g, uint64_t seed) {
    if (!g) return fail(nullptr, VH_ERR_INVALID, "null group");
    int rc = vh_init_weights_seeded(g->m[0]->ctx, seed);
    if (rc != VH_OK) { g->err = g->m[0]->ctx->err; return rc; }
    return vh_group_broadcast_weights(g);
}

// The hot path over the group: `batch` images in host memory, member r takes the contiguous range vh_group_shard_bounds
// gives it and writes its logits straight into the caller's buffer.  Every member runs its vh_forward (H2D, kernels,
// blocking D2H: the reference's launch_forward window, netFPGA.cpp:262-284) on its own host thread and stream.
int vh_group_forward(vh_group* g, const float* in_host, int batch, float* logits_host) {
    if (!g || !in_host || !logits_host) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "null argument");
    const int n = (int)g->m.size();
    if (batch < 1) return fail(&g->err, VH_ERR_INVALID, "batch must be positive");
    if ((batch + n - 1) / n > g->cfg.max_batch)
        return fail(&g->err, VH_ERR_INVALID, "batch %d over %d devices exceeds max_batch=%d per device", batch, n, g->cfg.max_batch);
    const size_t img = (size_t)g->cfg.image_size * g->cfg.image_size * g->cfg.channels, cls = (size_t)g->cfg.classes;
    return run_all(g, [=](int i) {
        int lo, hi;
        vh_group_shard_bounds(batch, n, i, &lo, &hi);
        if (hi == lo) return VH_OK;   // fewer images than members
        return vh_forward(g->m[i]->ctx, in_host + (size_t)lo * img, hi - lo, logits_host + (size_t)lo * cls);
    });
}

// ---- device-resident measurement path (bench.py --group): inputs generated in each member's HBM -----------------------
int vh_group_fill_inputs_seeded(vh_group* g, uint64_t seed, int batch_per_device) {
    if (!g || batch_per_device < 1 || batch_per_device > g->cfg.max_batch) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "bad argument");
    const size_t in_bytes = (size_t)batch_per_device * g->cfg.image_size * g->cfg.image_size * g->cfg.channels * 4;
    const size_t out_bytes = (size_t)batch_per_device * g->cfg.classes * 4;
    return run_all(g, [=](int i) {
        vh_group::Member* m = g->m[i].get();
        if (m->resident_batch < batch_per_device) {
            if (m->in_dev) hipFree(m->in_dev);
            if (m->out_dev) hipFree(m->out_dev);
            m->in_dev = m->out_dev = nullptr; m->resident_batch = 0;
            if (hipMalloc((void**)&m->in_dev, in_bytes) != hipSuccess || hipMalloc((void**)&m->out_dev, out_bytes) != hipSuccess)
                return fail(&m->ctx->err, VH_ERR_HIP, "vh_group_fill_inputs_seeded: out of device memory");
            m->resident_batch = batch_per_device;
        }
        return vh_fill_input_seeded(m->ctx, seed + (uint64_t)i, batch_per_device, m->in_dev);   // member i = rank i's shard
    });
}

int vh_group_forward_resident(vh_group* g, int batch_per_device, int steps) {
    if (!g || steps < 1) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "bad argument");
    for (auto& mp : g->m)
        if (mp->resident_batch < batch_per_device) return fail(&g->err, VH_ERR_STATE, "call vh_group_fill_inputs_seeded first");
    return run_all(g, [=](int i) {
        vh_group::Member* m = g->m[i].get();
        int rc = vh_forward_device_async(m->ctx, m->in_dev, batch_per_device, m->out_dev, steps);
        return rc != VH_OK ? rc : vh_synchronize(m->ctx);
    });
}

int vh_group_read_logits(vh_group* g, int batch_per_device, float* logits_host) {
    if (!g || !logits_host) return fail(g ? &g->err : nullptr, VH_ERR_INVALID, "null argument");
    const size_t per = (size_t)batch_per_device * g->cfg.classes;
    return run_all(g, [=](int i) {
        vh_group::Member* m = g->m[i].get();
        if (m->resident_batch < batch_per_device) return fail(&m->ctx->err, VH_ERR_STATE, "no resident batch");
        hipError_t e = hipMemcpy(logits_host + (size_t)i * per, m->out_dev, per * 4, hipMemcpyDeviceToHost);
        return e == hipSuccess ? VH_OK : fail(&m->ctx->err, VH_ERR_HIP, "hipMemcpy: %s", hipGetErrorString(e));
    });
}

}  // extern "C"
