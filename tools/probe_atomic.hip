// probe_atomic.hip — the inline-asm ticket draw of the attention kernel in isolation (one lane of one wave per
// workgroup draws from a device counter; the wave waits with vmcnt and publishes the value).
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_atomic.hip -o tools/probe_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned int* __restrict__ ticket, unsigned int* __restrict__ out, int rounds) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int tkv = 0;
    for (int r = 0; r < rounds; ++r) {
        if (wave == 1 && lane == 0) {
            const unsigned int one = 1u, zero = 0u;
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=&v"(tkv) : "v"(zero), "v"(one), "s"(ticket) : "memory");
        }
        if (wave == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(tkv));
            if (lane == 0) out[(blockIdx.x * rounds + r)] = tkv;
        }
        __syncthreads();
    }
}
int main() {
    const int grid = 512, rounds = 10;
    unsigned int *t, *o;
    hipMalloc(&t, 256); hipMalloc(&o, grid * rounds * 4);
    hipMemsetAsync(t, 0, 4, nullptr);
    hipLaunchKernelGGL(k, dim3(grid), dim3(448), 0, nullptr, t, o, rounds);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    std::vector<unsigned int> h(grid * rounds);
    hipMemcpy(h.data(), o, h.size() * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool ok = true;
    for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == i;
    printf("tickets 0..%zu drawn exactly once: %s (min %u max %u)\n", h.size() - 1, ok ? "yes" : "NO", h.front(), h.back());
    return ok ? 0 : 1;
}
