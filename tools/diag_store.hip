// diag_store.hip — how fast can workgroups push 128 KiB tiles of stores out?  (gfx950 micro-benchmark)
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/diag_store.hip -o tools/diag_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned u4;

// each workgroup (512 threads) writes `reps` tiles of 256 rows x 512 B at row stride `ld` bytes
template <bool NT>
__global__ void __launch_bounds__(512) store_tiles(char* out, size_t ld, int tiles_n, int ntiles, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < reps; ++r) {
        const int t = (blockIdx.x + r * gridDim.x) % ntiles;
        const int tm = t / tiles_n, tn = t % tiles_n;
        u4 v = {(unsigned)t, (unsigned)lane, (unsigned)r, 1u};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = tm * 256 + wave * 32 + i * 2 + (lane >> 5);
            char* p = out + (size_t)row * ld + (size_t)tn * 512 + (lane & 31) * 16;
            if (NT) __builtin_nontemporal_store(v, (u4*)p);
            else *(u4*)p = v;
        }
    }
}

int main() {
    const int M = 100864, N = 2304;
    const size_t ld = (size_t)N * 2;
    char* out;
    CK(hipMalloc(&out, (size_t)M * ld));
    const int tiles_n = N / 256, tiles_m = M / 256, ntiles = tiles_m * tiles_n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; ++nt)
        for (int grid : {8, 32, 64, 256, 1024, 3546}) {
            const int reps = grid >= 3546 ? 1 : (grid >= 256 ? 14 : 64);
            auto launch = [&]() {
                if (nt) hipLaunchKernelGGL(store_tiles<true>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_n, ntiles, reps);
                else hipLaunchKernelGGL(store_tiles<false>, dim3(grid), dim3(512), 0, 0, out, ld, tiles_n, ntiles, reps);
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 100.0, bytes = (double)grid * reps * 131072.0;
            printf("%s grid %5d x %2d tiles: %8.1f us  %7.2f TB/s  (%6.1f GB/s per workgroup, %5.2f us per tile)\n", nt ? "nt   " : "plain",
                   grid, reps, us, bytes / us / 1e6, bytes / grid / us / 1e3, us / reps);
        }
    return 0;
}
