// net_cpu.cpp — see net_cpu.h.  TEST INFRASTRUCTURE (links oracle/liboracle.so).
#include "net_cpu.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace cpu
{
    // MLP mode: sizes and flatten order exactly as the reference's constructor (netFPGA.cpp:58-109)
    net_cpu::net_cpu(const net::net_data &data, bool /*derivate*/, bool random)
        : n_ins((int)data.n_ins), n_layers((int)data.n_p_l.size()), n_neurons(0), n_params(0), activations(1),
          gradient_performance(0), forward_performance(0), vit_mode(false), threads(0)
    {
        std::memset(&vcfg, 0, sizeof vcfg);
        if (n_ins <= 0 || n_layers <= 0) throw std::runtime_error("net_cpu: empty network");
        int fan = n_ins;
        for (int l = 0; l < n_layers; ++l) {
            n_p_l.push_back((int)data.n_p_l[l]);
            n_params += n_p_l[l] * fan; // netFPGA.cpp:68-76
            n_neurons += n_p_l[l];
            fan = n_p_l[l];
        }
        params.resize((size_t)n_params);
        bias.resize((size_t)n_neurons);
        if (random) {
            oracle_mlp_random_params(params.data(), params.size(), bias.data(), bias.size(), 1u); // netFPGA.cpp:82-88
        } else {
            size_t p = 0, b = 0;
            for (int l = 0; l < n_layers; ++l)
                for (int j = 0; j < n_p_l[l]; ++j) {
                    for (size_t k = 0; k < data.params[l][j].size(); ++k) params[p++] = data.params[l][j][k];
                    bias[b++] = data.bias[l][j];
                }
        }
    }

    net_cpu::net_cpu(const oracle_vit_config &cfg, const void *blob, size_t blob_bytes, int nthreads)
        : n_ins(cfg.image_size * cfg.image_size * cfg.channels), n_layers(cfg.layers), n_neurons(0), n_params(0), activations(4),
          gradient_performance(0), forward_performance(0), vit_mode(true), vcfg(cfg), threads(nthreads)
    {
        if (blob_bytes != oracle_vit_blob_bytes(&cfg)) throw std::runtime_error("net_cpu: weight blob has the wrong size");
        vblob.assign((const char *)blob, blob_bytes);
    }

    net_cpu::net_cpu(const oracle_vit_config &cfg, uint64_t seed, int nthreads)
        : n_ins(cfg.image_size * cfg.image_size * cfg.channels), n_layers(cfg.layers), n_neurons(0), n_params(0), activations(4),
          gradient_performance(0), forward_performance(0), vit_mode(true), vcfg(cfg), threads(nthreads)
    {
        vblob.resize(oracle_vit_blob_bytes(&cfg));
        if (oracle_vit_make_blob(&cfg, seed, &vblob[0], vblob.size()) != 0) throw std::runtime_error("net_cpu: make_blob failed");
    }

    net::net_data net_cpu::get_net_data()
    {
        net::net_data d;
        d.n_ins = (size_t)n_ins;
        d.n_layers = vit_mode ? 0 : (size_t)n_layers;
        if (vit_mode) return d; // ViT weights travel as the canonical blob, not as net_data
        size_t p = 0, b = 0;
        int fan = n_ins;
        for (int l = 0; l < n_layers; ++l) {
            d.n_p_l.push_back((size_t)n_p_l[l]);
            d.params.emplace_back();
            d.bias.emplace_back();
            for (int j = 0; j < n_p_l[l]; ++j) {
                d.params[l].emplace_back(params.begin() + p, params.begin() + p + fan);
                p += (size_t)fan;
                d.bias[l].push_back(bias[b++]);
            }
            fan = n_p_l[l];
        }
        return d;
    }

    // the hot path on the CPU, inside the reference's timing window (netFPGA.cpp:262-264, 280-284)
    std::vector<DATA_TYPE> net_cpu::launch_forward(const std::vector<DATA_TYPE> &inputs)
    {
        if (inputs.empty() || inputs.size() % (size_t)n_ins) throw std::runtime_error("net_cpu::launch_forward: input size is not a multiple of n_ins");
        const int nvec = (int)(inputs.size() / (size_t)n_ins);
        const auto t0 = std::chrono::high_resolution_clock::now();
        std::vector<DATA_TYPE> out;
        if (vit_mode) {
            out.resize((size_t)nvec * vcfg.classes);
            if (oracle_vit_forward(&vcfg, vblob.data(), inputs.data(), nvec, out.data(), nullptr, -1, threads) != 0)
                throw std::runtime_error("net_cpu: oracle_vit_forward failed");
        } else {
            const int n_out = n_p_l[n_layers - 1];
            out.resize((size_t)nvec * n_out);
            for (int v = 0; v < nvec; ++v)
                if (oracle_mlp_forward(n_ins, n_layers, n_p_l.data(), params.data(), bias.data(), activations,
                                       inputs.data() + (size_t)v * n_ins, out.data() + (size_t)v * n_out) != 0)
                    throw std::runtime_error("net_cpu: oracle_mlp_forward failed");
        }
        forward_performance = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
        return out;
    }

    // training: bodies commented out in the reference (netFPGA.cpp:518-591); same observable behaviour
    void net_cpu::init_gradient(const net::net_sets &) {}
    std::vector<DATA_TYPE> net_cpu::launch_gradient(size_t iterations, DATA_TYPE, DATA_TYPE) { return std::vector<DATA_TYPE>(iterations, 0); }
    void net_cpu::print_inner_vals() {}
    signed long net_cpu::get_gradient_performance() { return (signed long)gradient_performance; }
    signed long net_cpu::get_forward_performance() { return (signed long)forward_performance; }

    // filter_image: the build's documented 3x3 blur (oracle_filter3x3), FIFO of at most 24 frames (netFPGA.cpp:12, 292-365)
    void net_cpu::filter_image(const net::image_set &set)
    {
        if (frames.size() >= 24) { std::printf("PILA LLENA\n"); return; }
        net::image_set o = set;
        oracle_filter3x3(set.resized_image_data.data(), o.resized_image_data.data(), (int)set.original_h, (int)set.original_w, 0);
        frames.push_back(o);
    }
    net::image_set net_cpu::get_filtered_image()
    {
        if (frames.empty()) { std::printf("PILA VACIA\n"); return net::image_set(); }
        net::image_set o = frames.front();
        frames.erase(frames.begin());
        return o;
    }
}
