// probe_fp8.hip — facts the fp8 path relies on, checked with exact data on the device:
//   (1) v_cvt_pk_fp8_f32 == round-to-nearest-even to OCP e4m3fn, and what it does beyond +-448 / with tiny inputs
//   (2) lane map of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8, unit scales): lane l carries row/col l&15 and
//       the 32 k-bytes of group l>>4; C/D as the bf16 16x16 forms
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probe_fp8.hip -o tools/probe_fp8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static float e4m3_to_float(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) v = NAN;
    else if (e == 0) v = std::ldexp((float)m, -9);
    else v = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
// RNE to e4m3fn, saturating at +-448
static uint8_t float_to_e4m3_sat(float f) {
    if (std::isnan(f)) return 0x7F;
    const uint8_t s = std::signbit(f) ? 0x80 : 0;
    float a = std::fabs(f);
    if (a >= 448.f) return s | 0x7E;
    if (a < std::ldexp(1.0f, -10)) return s;  // below half the smallest subnormal (2^-9): 0 (tie at 2^-10 -> even = 0)
    int e;
    std::frexp(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int ex = e - 1;     // a = 1.x * 2^ex
    if (ex < -6) ex = -6;
    const float q = std::ldexp(1.0f, ex - 3);  // quantum
    float r = std::nearbyint(a / q);           // RNE (default rounding mode)
    float v = r * q;
    if (v >= 448.f) return s | 0x7E;
    // encode
    if (v < std::ldexp(1.0f, -6)) return s | (uint8_t)(int)(v / std::ldexp(1.0f, -9));
    std::frexp(v, &e);
    ex = e - 1;
    const int m = (int)((v / std::ldexp(1.0f, ex) - 1.0f) * 8.0f);
    return s | (uint8_t)(((ex + 7) << 3) | m);
}

__global__ void cvt_k(const float* in, uint8_t* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int p = __builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.f, 0, false);
        out[i] = (uint8_t)(p & 0xFF);
    }
}

__global__ void mfma_k(const uint8_t* A, const uint8_t* Bt, float* D) {  // A[16][128], Bt[16][128] (Bt[n][k]), D[16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = *(const int*)(A + r * 128 + g * 32 + j * 4);
        b[j] = *(const int*)(Bt + r * 128 + g * 32 + j * 4);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    // bf16-form C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg  with A as the ROW operand
    for (int i = 0; i < 4; ++i) D[(g * 4 + i) * 16 + r] = c[i];
}

#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

int main() {
    // ---- (1) conversion ----
    std::vector<float> in;
    for (int b = 0; b < 256; ++b) { float v = e4m3_to_float((uint8_t)b); if (!std::isnan(v)) in.push_back(v); }
    for (int b = 0; b < 255; ++b) {  // midpoints and just off midpoints between neighbours
        float lo = e4m3_to_float((uint8_t)(b & 0x7F)), hi = e4m3_to_float((uint8_t)((b & 0x7F) + 1));
        if (std::isnan(lo) || std::isnan(hi)) continue;
        float mid = 0.5f * (lo + hi);
        for (float v : {mid, std::nextafterf(mid, 0.f), std::nextafterf(mid, 1e9f)}) { in.push_back(v); in.push_back(-v); }
    }
    for (float v : {448.f, 449.f, 464.f, 465.f, 480.f, 1000.f, 1e30f, INFINITY, 1e-3f, 9.765625e-4f, 9.7656256e-4f, 1.5e-3f, 0.f, -0.f}) { in.push_back(v); in.push_back(-v); }
    unsigned s = 12345;
    for (int i = 0; i < 200000; ++i) { s = s * 1664525u + 1013904223u; float u = ((s >> 8) & 0xFFFF) / 65536.f; s = s * 1664525u + 1013904223u; float e = (float)((s >> 8) % 20) - 12.f; in.push_back((i & 1 ? -1.f : 1.f) * (1.f + u) * std::ldexp(1.f, (int)e)); }
    const int n = (int)in.size();
    float* din; uint8_t* dout;
    CK(hipMalloc(&din, n * 4)); CK(hipMalloc(&dout, n));
    CK(hipMemcpy(din, in.data(), n * 4, hipMemcpyHostToDevice));
    cvt_k<<<(n + 255) / 256, 256>>>(din, dout, n);
    std::vector<uint8_t> out(n);
    CK(hipMemcpy(out.data(), dout, n, hipMemcpyDeviceToHost));
    int bad = 0, bad_in_range = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t want = float_to_e4m3_sat(in[i]);
        if (out[i] != want) {
            ++bad;
            if (std::fabs(in[i]) <= 448.f) ++bad_in_range;
            if (bad <= 24) printf("cvt mismatch: in %.9g (0x%08x) hw 0x%02x (%g) sw 0x%02x (%g)\n", in[i], *(uint32_t*)&in[i], out[i], e4m3_to_float(out[i]), want, e4m3_to_float(want));
        }
    }
    printf("cvt: %d values, %d mismatches vs RNE-saturating e4m3fn (%d of them with |x| <= 448)\n", n, bad, bad_in_range);

    // ---- (2) MFMA lane map ----
    std::vector<uint8_t> A(16 * 128), Bt(16 * 128);
    std::vector<float> Af(16 * 128), Bf(16 * 128);
    for (int i = 0; i < 16 * 128; ++i) {
        s = s * 1664525u + 1013904223u; int va = (int)((s >> 10) % 9) - 4;   // exact small integers
        s = s * 1664525u + 1013904223u; int vb = (int)((s >> 10) % 7) - 3;
        Af[i] = (float)va; Bf[i] = (float)vb * 0.5f;
        A[i] = float_to_e4m3_sat(Af[i]); Bt[i] = float_to_e4m3_sat(Bf[i]);
    }
    uint8_t *dA, *dB; float* dD;
    CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bt.data(), 2048, hipMemcpyHostToDevice));
    mfma_k<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(256);
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int ok_rowA = 0, ok_colA = 0;
    for (int m = 0; m < 16; ++m)
        for (int nn = 0; nn < 16; ++nn) {
            float ref = 0.f;
            for (int k = 0; k < 128; ++k) ref += Af[m * 128 + k] * Bf[nn * 128 + k];
            if (D[m * 16 + nn] == ref) ++ok_rowA;
            if (D[nn * 16 + m] == ref) ++ok_colA;
        }
    printf("mfma 16x16x128 fp8: D[row=(l>>4)*4+i][col=l&15] == A(row) x Bt(col): %d/256 exact;  transposed reading: %d/256\n", ok_rowA, ok_colA);
    return 0;
}
