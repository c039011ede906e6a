// diag_sp.hip -- feasibility probe for DESIGN.md section 9 item 8 (NOT product code): the K loop of a GEMM with ONE wave per
// SIMD (4 waves, 512-register budget, wave tile 128 x 64, workgroup tile 256 x 128, three 48 KiB LDS stages), every fragment
// read and DMA piece issued by the same wave between its own MFMAs, one counted wait + one barrier per K-tile.  Question:
// what does that instruction stream deliver, compiler-scheduled from HIP source, against the ping-pong kernel's K loop
// (1.41 PF on the fc1 shape without an epilogue, profiles/r03_i_mainloop_ablation.txt)?  Only then is the second
// accumulator set such a kernel has room for worth building an overlapped epilogue on.
// Measured (profiles/r03_s_single_wave_probe.txt; results checked against a scalar reference):
//   * operands by LDS-DMA, first form of this file (332 registers, no spill, prefetch one K-tile ahead): fc1 550 TF/s, q|k|v 539,
//     fc2 421, out-proj 486 -- 39 % of the ping-pong K loop.  An in-order wave cannot issue its own MFMAs while one of its twelve
//     DMA pieces per K-tile is being issued (100+ cycles each, MI355X_MICROARCH.md); with a partner wave that cost hides, alone
//     it is the K-tile.  Prefetching two K-tiles ahead (this form) does not help: 455 TF/s.
//   * operands through registers (global_load_dwordx4, ds_write_b128 a K-tile later): 396 TF/s AS COMPILED -- hipcc keeps the
//     non-accumulator state in the 256 architectural VGPRs (accumulators in AGPRs), spills a few 64-bit values and reloads them
//     inside the K loop with scratch loads, each behind an s_waitcnt vmcnt(0) that also drains the operand loads in flight.
// Reading: the arrangement of DESIGN.md section 9 item 8 is not reachable from HIP source with this compiler; it needs a
// hand-allocated, hand-scheduled instruction stream.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_sp.hip -o tools/diag_sp      Run: tools/diag_sp [check]
#include "../vit-fpga_amd/csrc/vh_common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <tuple>
#include <type_traits>
#include <cstring>
#include <cmath>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
using namespace vh;
using T = BF16;
using vec8 = T::vec8;

constexpr int BM = 256, BN = 128, KTB = 128;          // K-tile: 64 elements = 128 bytes per row
constexpr int A_ST = BM * KTB, W_ST = BN * KTB, STAGE = A_ST + W_ST, NST = 3;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void bar() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// STORE: write the tile (bf16, direct stores: slow, for the correctness check only)
template <bool STORE, bool REGST>
__global__ void __launch_bounds__(256)
sp_kernel(const char* __restrict__ A, const char* __restrict__ W, T::elem* __restrict__ out, int M, int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t rb = (int64_t)K * 2;
    const int nk = K / 64;
    const int ntiles = tiles_m * tiles_n, stride = (int)gridDim.x;
    // DMA: this wave moves A rows wave*64 .. +63 (8 pieces of 8 rows) and W rows wave*32 .. +31 (4 pieces) of every K-tile
    const int lr = lane >> 3, lc = (lane & 7) ^ lr;
    uint32_t oa[8], ow[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) oa[i] = (uint32_t)((wave * 64 + i * 8 + lr) * (int)rb + lc * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) ow[i] = (uint32_t)((wave * 32 + i * 8 + lr) * (int)rb + lc * 16);
    auto dma = [&](const char* src, char* dst) __attribute__((always_inline)) {
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src, (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
    };
    // the stream of K-tiles this workgroup walks: tile t0, t0 + stride, ...; n fastest inside a row of tiles
    int t_issue = blockIdx.x, kt_issue = 0, g_issue = 0;      // next K-tile to FETCH
    const char *ia = nullptr, *iw = nullptr;
    auto issue_setup = [&]() __attribute__((always_inline)) {
        const int tm = t_issue / tiles_n, tn = t_issue - tm * tiles_n;
        ia = A + (int64_t)tm * BM * rb;
        iw = W + (int64_t)tn * BN * rb;
    };
    auto issue_piece = [&](int i) __attribute__((always_inline)) {   // piece i of 12 of the K-tile being fetched (0-7: A, 8-11: W)
        char* st = smem + (g_issue % NST) * STAGE;
        if (i < 8) dma(ia + (int64_t)kt_issue * KTB + oa[i], st + wave * 8192 + i * 1024);
        else dma(iw + (int64_t)kt_issue * KTB + ow[i - 8], st + A_ST + wave * 4096 + (i - 8) * 1024);
    };
    auto issue_advance = [&]() __attribute__((always_inline)) {
        ++g_issue;
        if (++kt_issue == nk) { kt_issue = 0; t_issue += stride; if (t_issue < ntiles) issue_setup(); }
    };
    // REGST: operands staged through registers (global_load_dwordx4 now, ds_write_b128 a K-tile later) instead of LDS-DMA, whose
    // issue blocks an in-order wave for 100+ cycles per piece (MI355X_MICROARCH.md constants) -- fatal with no partner wave
    u32x4 rg[1][12];
    auto load_piece = [&](u32x4 (&r)[12], int i) __attribute__((always_inline)) {
        if (i < 8) r[i] = *(const u32x4*)(ia + (int64_t)kt_issue * KTB + oa[i]);
        else r[i] = *(const u32x4*)(iw + (int64_t)kt_issue * KTB + ow[i - 8]);
    };
    auto store_piece = [&](const u32x4 (&r)[12], int i, int gdst) __attribute__((always_inline)) {
        char* st = smem + (gdst % NST) * STAGE;
        if (i < 8) *(u32x4*)(st + wave * 8192 + i * 1024 + lane * 16) = r[i];
        else *(u32x4*)(st + A_ST + wave * 4096 + (i - 8) * 1024 + lane * 16) = r[i];
    };
    const int frow = lane & 15, fq = lane >> 4;
    const int off0 = frow * 128 + ((fq ^ (frow & 7)) << 4), off1 = frow * 128 + (((4 | fq) ^ (frow & 7)) << 4);
    const int xbase = wm * 128 * 128, wbase = A_ST + wn * 64 * 128;

    if (blockIdx.x >= ntiles) return;
    issue_setup();
    if constexpr (REGST) {
        // prologue: K-tiles 0 and 1 through the registers into stages 0 and 1; K-tile 2 left in flight in the registers
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int i = 0; i < 12; ++i) load_piece(rg[0], i);
#pragma unroll
            for (int i = 0; i < 12; ++i) store_piece(rg[0], i, p);
            issue_advance();
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) load_piece(rg[0], i);
        issue_advance();
        __builtin_amdgcn_s_waitcnt(0xC07F);
        bar();
    } else {
    // prologue: K-tiles 0, 1 and 2 of the stream (all three stages)
    for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 12; ++i) issue_piece(i);
        issue_advance();
    }
    wait_vmcnt<24>();
    bar();
    }
    vec8 xf[2][8], wf[2][4];
    {
        const char* st = smem;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) xf[0][mi] = *(const vec8*)(st + xbase + mi * 2048 + off0);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) wf[0][ni] = *(const vec8*)(st + wbase + ni * 2048 + off0);
    }
    int g = 0;   // K-tile of the stream being COMPUTED
    for (int t = blockIdx.x; t < ntiles; t += stride) {
        f32x4 acc[8][4];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto ktile = [&](auto FETCH, auto PAR) __attribute__((always_inline)) {
            constexpr int par = decltype(PAR)::value;   // parity of g + 1: the register set K-tile g + 1 sits in (REGST)   // FETCH: is there a K-tile g + 3 in the stream?  (compile-time: no branch inside the MFMA stream)
            constexpr bool fetch = decltype(FETCH)::value;
            const char* st = smem + (g % NST) * STAGE;
            const char* sn = smem + ((g + 1) % NST) * STAGE;
            __builtin_amdgcn_s_waitcnt(0xC07F);    // lgkmcnt(0): the fragments of (g, k-step 0) are in registers
            // ---- first half: MFMAs of k-step 0; between them the 12 fragment reads of k-step 1 and the 12 DMA pieces of K-tile g + 2
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int mi = i >> 2, ni = i & 3;
                acc[mi][ni] = T::mfma16(wf[0][ni], xf[0][mi], acc[mi][ni]);
                if (i < 8) xf[1][i] = *(const vec8*)(st + xbase + i * 2048 + off1);
                else if (i < 12) wf[1][i - 8] = *(const vec8*)(st + wbase + (i - 8) * 2048 + off1);
                else if (i < 24) {
                    if constexpr (REGST) store_piece(rg[0], i - 12, g + 2);   // K-tile g + 2 (loaded during K-tile g - 1) into the stage bar(g - 1) freed
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (REGST) { }
            else if constexpr (fetch) wait_vmcnt<12>();   // all but K-tile g + 2: K-tile g + 1 has landed
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            bar();   // K-tile g + 1 visible to everybody; nobody reads stage g any more
            // ---- second half: MFMAs of k-step 1; between them the 12 fragment reads of (g + 1, k-step 0)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int mi = i >> 2, ni = i & 3;
                acc[mi][ni] = T::mfma16(wf[1][ni], xf[1][mi], acc[mi][ni]);
                if (i < 8) xf[0][i] = *(const vec8*)(sn + xbase + i * 2048 + off0);
                else if (i < 12) wf[0][i - 8] = *(const vec8*)(sn + wbase + (i - 8) * 2048 + off0);
                else if (i < 24) {   // K-tile g + 3: by DMA into the stage the barrier just freed, or into the staging registers
                    if constexpr (fetch && !REGST) issue_piece(i - 12);
                    if constexpr (fetch && REGST) load_piece(rg[0], i - 12);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (fetch) issue_advance();
        };
        for (int kt = 0; kt < nk; ++kt, ++g) {
            if (t_issue < ntiles) ktile(std::true_type{}, std::integral_constant<int, 0>{});
            else ktile(std::false_type{}, std::integral_constant<int, 0>{});
        }
        if constexpr (STORE) {
            const int tm = t / tiles_n, tn = t - tm * tiles_n;
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    // accumulator layout of T::mfma16(w, x, acc): lane (frow, fq) holds row frow of the x block, 4 consecutive columns fq * 4 ..
                    const int m = tm * BM + wm * 128 + mi * 16 + frow, n = tn * BN + wn * 64 + ni * 16 + fq * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) out[(int64_t)m * N + n + j] = (T::elem)acc[mi][ni][j];
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep the stores out of the next tile's counted waits
        } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(acc[mi][ni]));
        }
    }
}

__global__ void ref_kernel(const T::elem* A, const T::elem* W, float* out, int N, int K, const int* rows, const int* cols, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(int64_t)rows[i] * K + k] * (float)W[(int64_t)cols[i] * K + k];
    out[i] = s;
}
__global__ void fill(T::elem* p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (T::elem)(((float)(x & 0xFFFF) / 32768.f - 1.f) * scale);
    }
}

int main(int argc, char** argv) {
    const bool check = argc > 1;
    const int M = 100864;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto [N, K, name] : {std::tuple<int, int, const char*>{3072, 768, "fc1"}, {2304, 768, "qkv"}, {768, 3072, "fc2"}, {768, 768, "proj"}}) {
        T::elem *A, *W, *out;
        CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&out, (size_t)M * N * 2));
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, A, (int64_t)M * K, 1u, 1.0f);
        hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, W, (int64_t)N * K, 2u, 0.05f);
        const int tiles_m = M / BM, tiles_n = N / BN, ntiles = tiles_m * tiles_n;
        const int grid = ntiles < 256 ? ntiles : 256;
        const size_t lds = (size_t)NST * STAGE;
        for (const void* f : {(const void*)sp_kernel<false, false>, (const void*)sp_kernel<true, false>, (const void*)sp_kernel<false, true>, (const void*)sp_kernel<true, true>})
            CK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int regst = 0; regst < 2; ++regst) {
        if (check) {
            if (regst) hipLaunchKernelGGL((sp_kernel<true, true>), dim3(grid), dim3(256), lds, 0, (const char*)A, (const char*)W, out, M, N, K, tiles_m, tiles_n);
            else hipLaunchKernelGGL((sp_kernel<true, false>), dim3(grid), dim3(256), lds, 0, (const char*)A, (const char*)W, out, M, N, K, tiles_m, tiles_n);
            CK(hipDeviceSynchronize());
            const int n = 4096;
            std::vector<int> rows(n), cols(n);
            for (int i = 0; i < n; ++i) { rows[i] = (int)((uint64_t)i * 2654435761u % M); cols[i] = (int)((uint64_t)i * 40503u % N); }
            rows[0] = M - 1; cols[0] = N - 1; rows[1] = 0; cols[1] = 0; rows[2] = 255; cols[2] = 127; rows[3] = 256; cols[3] = 128;
            int *dr, *dc; float* dref;
            CK(hipMalloc(&dr, n * 4)); CK(hipMalloc(&dc, n * 4)); CK(hipMalloc(&dref, n * 4));
            CK(hipMemcpy(dr, rows.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cols.data(), n * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(ref_kernel, dim3(n / 256), dim3(256), 0, 0, A, W, dref, N, K, dr, dc, n);
            std::vector<float> ref(n);
            CK(hipMemcpy(ref.data(), dref, n * 4, hipMemcpyDeviceToHost));
            std::vector<uint16_t> got(n);
            double worst = 0, big = 0;
            for (int i = 0; i < n; ++i) {
                uint16_t b;
                CK(hipMemcpy(&b, (const char*)out + ((size_t)rows[i] * N + cols[i]) * 2, 2, hipMemcpyDeviceToHost));
                uint32_t u = (uint32_t)b << 16; float v; memcpy(&v, &u, 4);
                worst = fmax(worst, fabs(v - ref[i])); big = fmax(big, fabs(ref[i]));
            }
            printf("%-5s check: max |d| %.4g of max |ref| %.4g (%s)\n", name, worst, big, worst <= 0.01 * big ? "ok" : "WRONG");
        }
        auto launch = [&]() {
            if (regst) hipLaunchKernelGGL((sp_kernel<false, true>), dim3(grid), dim3(256), lds, 0, (const char*)A, (const char*)W, out, M, N, K, tiles_m, tiles_n);
            else hipLaunchKernelGGL((sp_kernel<false, false>), dim3(grid), dim3(256), lds, 0, (const char*)A, (const char*)W, out, M, N, K, tiles_m, tiles_n);
        };
        for (int i = 0; i < 20; ++i) launch();
        CK(hipDeviceSynchronize());
        const int iters = 600;
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("%-5s N %4d K %4d: one wave per SIMD, 256 x 128 tile, K loop only, operands by %s: %7.1f us  %7.1f TF/s\n", name, N, K,
               regst ? "registers (global_load + ds_write)" : "LDS-DMA", us, 2.0 * M * N * K / us / 1e6);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(out));
    }
    return 0;
}
