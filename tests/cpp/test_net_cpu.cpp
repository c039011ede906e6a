// test_net_cpu.cpp — cpu::net_cpu driven through net::net_abstract* (tests/cpp/net_cpu.h): the CPU leg of BASELINE
// config 1.  TEST INFRASTRUCTURE.  Checks: MLP mode = the reference's semantics (flatten order, get_net_data inverse),
// ViT mode = the oracle bit for bit, the timing getter, the reference's observable behaviour of the stubs and of the
// 24-slot frame FIFO.
#include "net_cpu.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main()
{
    // ---- MLP mode --------------------------------------------------------------------------------------------------
    net::net_data d;
    d.n_ins = 5; d.n_p_l = {4, 3}; d.n_layers = 2;
    d.params = {std::vector<std::vector<DATA_TYPE>>(4, std::vector<DATA_TYPE>(5)), std::vector<std::vector<DATA_TYPE>>(3, std::vector<DATA_TYPE>(4))};
    d.bias = {std::vector<DATA_TYPE>(4), std::vector<DATA_TYPE>(3)};
    float v = 0.01f;
    for (auto &l : d.params) for (auto &n : l) for (auto &w : n) { w = v; v = -v * 1.07f; }
    for (auto &l : d.bias) for (auto &b : l) { b = v; v = -v * 0.9f; }
    std::unique_ptr<net::net_abstract> mlp(new cpu::net_cpu(d, false, false));
    net::net_data back = mlp->get_net_data();
    CHECK(back.n_ins == d.n_ins && back.n_p_l == d.n_p_l && back.params == d.params && back.bias == d.bias);
    std::vector<DATA_TYPE> x = {0.5f, -0.25f, 1.0f, 0.0f, -1.0f};
    std::vector<DATA_TYPE> y = mlp->launch_forward(x);
    CHECK(y.size() == 3);
    // by hand: RELU2 = min(max(.,0),1) per layer (include/vithip.h VH_ACT_RELU2)
    std::vector<double> h(4), o(3);
    for (int j = 0; j < 4; ++j) { double s = d.bias[0][j]; for (int k = 0; k < 5; ++k) s += (double)d.params[0][j][k] * x[k]; h[j] = std::fmin(std::fmax(s, 0.0), 1.0); }
    for (int j = 0; j < 3; ++j) { double s = d.bias[1][j]; for (int k = 0; k < 4; ++k) s += (double)d.params[1][j][k] * h[k]; o[j] = std::fmin(std::fmax(s, 0.0), 1.0); }
    for (int j = 0; j < 3; ++j) CHECK(std::fabs(y[j] - o[j]) < 1e-6);
    CHECK(mlp->get_forward_performance() >= 0);
    CHECK(mlp->launch_gradient(7, 0.1f, 0.5f) == std::vector<DATA_TYPE>(7, 0));   // netFPGA.cpp:579
    CHECK(mlp->get_gradient_performance() == 0);

    // ---- ViT mode: bit for bit the oracle ----------------------------------------------------------------------------
    oracle_vit_config c;
    c.image_size = 32; c.patch_size = 8; c.channels = 4; c.dim = 64; c.heads = 1; c.mlp_dim = 128; c.layers = 2; c.classes = 12; c.ln_eps = 1e-6f;
    std::unique_ptr<net::net_abstract> vit(new cpu::net_cpu(c, (uint64_t)11, 2));
    std::vector<DATA_TYPE> img((size_t)3 * 32 * 32 * 4);
    oracle_fill(img.data(), (int64_t)img.size(), 12, 0x100, 0, 0.f, 0.f);
    std::vector<DATA_TYPE> logits = vit->launch_forward(img);
    CHECK(logits.size() == 3 * 12);
    std::vector<char> blob(oracle_vit_blob_bytes(&c));
    CHECK(oracle_vit_make_blob(&c, 11, blob.data(), blob.size()) == 0);
    std::vector<float> ref(3 * 12);
    CHECK(oracle_vit_forward(&c, blob.data(), img.data(), 3, ref.data(), nullptr, -1, 2) == 0);
    CHECK(std::memcmp(ref.data(), logits.data(), ref.size() * 4) == 0);
    CHECK(vit->get_forward_performance() > 0);
    bool threw = false;
    try { vit->launch_forward(std::vector<DATA_TYPE>(17)); } catch (const std::exception &) { threw = true; }
    CHECK(threw);

    // ---- frame FIFO: 24 in flight, then "PILA LLENA"; empty -> "PILA VACIA" + empty image (netFPGA.cpp:330-333, 358-361)
    net::image_set im;
    im.original_h = 6; im.original_w = 8; im.original_x_pos = im.original_y_pos = 0;
    im.resized_image_data.assign(48, 10);
    im.resized_image_data[20] = 250;
    for (int i = 0; i < 25; ++i) vit->filter_image(im);
    int got = 0;
    for (int i = 0; i < 26; ++i) { net::image_set o = vit->get_filtered_image(); if (!o.resized_image_data.empty()) ++got; }
    CHECK(got == 24);

    std::printf("net_cpu: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
