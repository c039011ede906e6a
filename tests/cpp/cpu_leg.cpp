// cpu_leg.cpp — times the CPU leg (cpu::net_cpu behind net::net_abstract*) the way BASELINE.md section 3 prescribes:
// through launch_forward, read back with get_forward_performance() (the reference's chrono window, netFPGA.cpp:262-284),
// 3 warm-ups + >= 5 timed runs, median.  TEST INFRASTRUCTURE; prints one JSON object.
//   cpu_leg <image> <patch> <channels> <dim> <heads> <mlp> <layers> <classes> <batch> <threads> <max_seconds>
#include "net_cpu.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

int main(int argc, char **argv)
{
    if (argc != 12) { std::fprintf(stderr, "usage: cpu_leg image patch channels dim heads mlp layers classes batch threads max_seconds\n"); return 2; }
    oracle_vit_config c;
    c.image_size = std::atoi(argv[1]); c.patch_size = std::atoi(argv[2]); c.channels = std::atoi(argv[3]); c.dim = std::atoi(argv[4]);
    c.heads = std::atoi(argv[5]); c.mlp_dim = std::atoi(argv[6]); c.layers = std::atoi(argv[7]); c.classes = std::atoi(argv[8]);
    c.ln_eps = 1e-6f;
    const int batch = std::atoi(argv[9]), threads = std::atoi(argv[10]);
    const double max_s = std::atof(argv[11]);
    std::unique_ptr<net::net_abstract> net(new cpu::net_cpu(c, (uint64_t)0, threads));   // weights seed 0
    std::vector<DATA_TYPE> in((size_t)batch * c.image_size * c.image_size * c.channels);
    oracle_fill(in.data(), (int64_t)in.size(), 1, 0x100, 0, 0.f, 0.f);                     // images seed 1, uniform[-1,1)
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<DATA_TYPE> out;
    int warm = 0;
    for (; warm < 3 && (warm == 0 || elapsed() < 0.2 * max_s); ++warm) out = net->launch_forward(in);
    std::vector<double> us;
    while (us.size() < 5 || (us.size() < 50 && elapsed() < max_s)) {
        out = net->launch_forward(in);
        us.push_back((double)net->get_forward_performance());
        if (elapsed() > 3.0 * max_s) break;   // hard stop for very slow configurations
    }
    std::sort(us.begin(), us.end());
    const double med = us[us.size() / 2];
    double chk = 0.0;
    for (float v : out) chk += v;
    std::printf("{\"images_per_s\": %.4f, \"median_us\": %.1f, \"runs\": %zu, \"warmups\": %d, \"batch\": %d, \"threads\": %d, \"logit_sum\": %.6e}\n",
                batch / (med * 1e-6), med, us.size(), warm, batch, threads, chk);
    return 0;
}
