// test_net_hip.cpp — drives hip::net_hip exactly the way an application of the reference drives
// fpga::net_fpga: through a net::net_abstract*.  `cpu` runs the host-only checks (no device is
// touched: construction, flatten order, get_net_data round trip, rule-of-five, stubs); `gpu` adds
// launch_forward in MLP and ViT mode against the CPU oracle (linked here, in the TEST only).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <netHIP.h>
#include <string>
#include <vector>

#include "../../oracle/oracle.h"

static int failures = 0;
#define CHECK(cond)                                                             \
    do {                                                                        \
        if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static net::net_data make_net(size_t n_ins, const std::vector<size_t> &npl)
{
    net::net_data d;
    d.n_ins = n_ins;
    d.n_layers = npl.size();
    d.n_p_l = npl;
    float v = 0.f;
    size_t fan = n_ins;
    for (size_t l = 0; l < npl.size(); ++l)
    {
        d.params.emplace_back(npl[l], std::vector<float>(fan));
        d.bias.emplace_back(npl[l]);
        for (size_t j = 0; j < npl[l]; ++j)
        {
            for (size_t k = 0; k < fan; ++k) { d.params[l][j][k] = std::sin(v) * 0.3f; v += 1.f; }
            d.bias[l][j] = std::cos(v) * 0.1f;
        }
        fan = npl[l];
    }
    return d;
}

static void host_checks()
{
    const net::net_data d = make_net(5, {4, 3, 2});
    std::unique_ptr<net::net_abstract> net(new hip::net_hip(d, false, false));
    hip::net_hip *h = static_cast<hip::net_hip *>(net.get());
    // bookkeeping of netFPGA.cpp:68-76
    CHECK(h->n_ins == 5 && h->n_layers == 3 && h->n_neurons == 9 && h->n_params == 5 * 4 + 4 * 3 + 3 * 2);
    CHECK(h->n_p_l[0] == 4 && h->n_p_l[1] == 3 && h->n_p_l[2] == 2);
    CHECK(h->activations == 1 && h->n_sets == 0 && !h->gradient_init);
    CHECK(h->forward_performance == 0 && h->gradient_performance == 0 && !h->device_init);
    // flatten order of netFPGA.cpp:91-106: layer-major, neuron-major, input-minor
    CHECK(h->params[0] == d.params[0][0][0] && h->params[5] == d.params[0][1][0] && h->params[20] == d.params[1][0][0]);
    CHECK(h->params[20 + 12 + 3] == d.params[2][1][0] && h->bias[4] == d.bias[1][0] && h->bias[8] == d.bias[2][1]);
    // get_net_data is the exact inverse
    const net::net_data r = net->get_net_data();
    CHECK(r.n_ins == d.n_ins && r.n_layers == d.n_layers && r.n_p_l == d.n_p_l && r.params == d.params && r.bias == d.bias);
    // stubs behave like the reference's
    net->init_gradient(net::net_sets());
    const std::vector<float> g = net->launch_gradient(7, 0.1f, 0.5f);
    CHECK(g.size() == 7 && g[0] == 0.f && g[6] == 0.f);
    net->print_inner_vals();
    CHECK(net->get_forward_performance() == 0 && net->get_gradient_performance() == 0);
    const net::image_set im = net->get_filtered_image();
    CHECK(im.original_h == 1080 && im.original_w == 1920 && im.resized_image_data.empty());
    // rule of five
    hip::net_hip moved(std::move(*h));
    CHECK(moved.n_params == 38 && moved.params[5] == d.params[0][1][0] && h->params == nullptr);
    hip::net_hip other(make_net(2, {2}), false, false);
    other = moved; // copy-assign: deep copy
    CHECK(other.n_params == 38 && other.params != moved.params && other.params[20] == d.params[1][0][0] && other.n_p_l[2] == 2);
    other = hip::net_hip(make_net(3, {1}), false, false); // move-assign
    CHECK(other.n_params == 3 && other.n_layers == 1);
    // random branch: the reference's formula and draw order on libc rand()
    srand(1);
    hip::net_hip rnd(make_net(5, {4, 3, 2}), false, true);
    srand(1);
    bool same = true;
    for (int i = 0; i < rnd.n_params; ++i) same &= rnd.params[i] == float(rand() % 200 - 100) / 100;
    for (int i = 0; i < rnd.n_neurons; ++i) same &= rnd.bias[i] == float(rand() % 200 - 100) / 100;
    CHECK(same);
    // bad input is reported, not UB
    bool threw = false;
    try { moved.launch_forward(std::vector<float>(4)); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
    // ViT-mode construction is host-only too
    vh_config c = {64, 16, 3, 128, 2, 256, 2, 40, VH_DTYPE_FP16, 2, 1e-6f, 0};
    hip::net_hip vit(c, 11);
    CHECK(vit.is_vit() && vit.n_ins == 64 * 64 * 3 && !vit.device_init && vit.vit_param_count() > 0);
    threw = false;
    c.dim = 100;
    try { hip::net_hip bad(c, 1); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
}

static void gpu_checks()
{
    // ---- MLP mode vs oracle --------------------------------------------------------------
    const net::net_data d = make_net(37, {64, 130, 5});
    std::unique_ptr<net::net_abstract> net(new hip::net_hip(d, false, false));
    hip::net_hip *h = static_cast<hip::net_hip *>(net.get());
    std::vector<float> x(37);
    for (int i = 0; i < 37; ++i) x[i] = std::sin(0.37f * i);
    const std::vector<float> y = net->launch_forward(x);
    std::vector<float> ref(5);
    oracle_mlp_forward(h->n_ins, h->n_layers, h->n_p_l, h->params, h->bias, h->activations, x.data(), ref.data());
    CHECK(y.size() == 5);
    for (int i = 0; i < 5; ++i) CHECK(std::fabs(y[i] - ref[i]) <= 1e-5f * (1.f + std::fabs(ref[i])));
    CHECK(net->get_forward_performance() > 0 && h->device_init);

    // ---- MLP-mode training through the abstract interface vs the oracle's restatement (parity unpinned: f4) ------
    {
        const net::net_data td = make_net(12, {20, 9, 3});
        std::unique_ptr<net::net_abstract> tn(new hip::net_hip(td, false, false));
        hip::net_hip *th = static_cast<hip::net_hip *>(tn.get());
        net::net_sets sets;
        std::vector<float> fi, fo;
        for (int j = 0; j < 6; ++j) {
            std::vector<float> in(12), out(3);
            for (int i = 0; i < 12; ++i) in[i] = std::sin(0.3f * i + j);
            for (int i = 0; i < 3; ++i) out[i] = 0.5f + 0.4f * std::cos(1.1f * i + 0.7f * j);
            fi.insert(fi.end(), in.begin(), in.end()); fo.insert(fo.end(), out.begin(), out.end());
            sets.set_ins.push_back(in); sets.set_outs.push_back(out);
        }
        std::vector<float> rp(th->params, th->params + th->n_params), rb(th->bias, th->bias + th->n_neurons), rerr(15);
        CHECK(tn->launch_gradient(4, -1.f, 0.1f) == std::vector<float>(4, 0.f));   // before init_gradient: the reference's zeros
        tn->init_gradient(sets);
        CHECK(th->gradient_init && th->n_sets == 6);
        const std::vector<float> err = tn->launch_gradient(15, -1.f, 0.1f);
        oracle_mlp_train(12, 3, th->n_p_l, rp.data(), rb.data(), th->activations, fi.data(), fo.data(), 6, 15, -1.f, 0.1f, rerr.data());
        CHECK(err.size() == 15 && err[14] < err[0] && tn->get_gradient_performance() > 0);
        for (int i = 0; i < 15; ++i) CHECK(std::fabs(err[i] - rerr[i]) <= 2e-4f * (1.f + rerr[i]));
        float dmax = 0.f;
        for (int i = 0; i < th->n_params; ++i) dmax = std::fmax(dmax, std::fabs(th->params[i] - rp[i]));
        CHECK(dmax <= 2e-4f);                                    // the host copy follows the device: get_net_data() is the trained net
        const net::net_data trained = tn->get_net_data();
        CHECK(trained.params[0][0][0] == th->params[0] && trained.bias[2][2] == th->bias[th->n_neurons - 1]);
    }

    // ---- ViT mode vs oracle ------------------------------------------------------------------
    oracle_vit_config oc = {64, 16, 3, 128, 2, 256, 2, 40, 1e-6f};
    vh_config c = {64, 16, 3, 128, 2, 256, 2, 40, VH_DTYPE_FP16, 1, 1e-6f, 0};
    std::vector<char> blob(oracle_vit_blob_bytes(&oc));
    CHECK(blob.size() == vh_weight_blob_bytes(&c));
    oracle_vit_make_blob(&oc, 5, blob.data(), blob.size());
    const int B = 3;
    std::vector<float> img((size_t)B * 64 * 64 * 3), want((size_t)B * 40);
    oracle_fill(img.data(), (int64_t)img.size(), 6, 0x100, 0, 0.f, 0.f);
    oracle_vit_forward(&oc, blob.data(), img.data(), B, want.data(), nullptr, -1, 0);
    std::unique_ptr<net::net_abstract> v(new hip::net_hip(c, blob.data(), blob.size()));
    const std::vector<float> got = v->launch_forward(img); // batch 3 > max_batch 1: workspace grows
    CHECK(got.size() == want.size());
    float mx = 0.f, err = 0.f;
    for (size_t i = 0; i < want.size(); ++i) { mx = std::fmax(mx, std::fabs(want[i])); err = std::fmax(err, std::fabs(got[i] - want[i])); }
    printf("net_hip ViT fp16: max|d|/max|ref| = %.3e\n", err / mx);
    CHECK(err / mx <= 1e-3f);
    // seeded ViT: device generator == oracle generator, so logits must match the oracle's too
    oracle_vit_make_blob(&oc, 77, blob.data(), blob.size());
    oracle_vit_forward(&oc, blob.data(), img.data(), B, want.data(), nullptr, -1, 0);
    hip::net_hip seeded(c, 77);
    const std::vector<float> got2 = seeded.launch_forward(img);
    err = 0.f; mx = 0.f;
    for (size_t i = 0; i < want.size(); ++i) { mx = std::fmax(mx, std::fabs(want[i])); err = std::fmax(err, std::fabs(got2[i] - want[i])); }
    CHECK(err / mx <= 1e-3f);
    CHECK(seeded.get_forward_performance() > 0 && seeded.last_kernel_ms() > 0.0);

    // ---- pipelined submit/collect: FIFO, same logits as launch_forward, reference-style full/empty reports ----
    seeded.set_pipeline(2, B);
    CHECK(seeded.collect_forward().empty());                 // "PILA VACIA"
    const std::vector<float> one(img.begin(), img.begin() + 64 * 64 * 3);
    CHECK(seeded.submit_forward(img) && seeded.submit_forward(one));
    CHECK(!seeded.submit_forward(one));                      // "PILA LLENA": two slots, both in flight
    const std::vector<float> r0 = seeded.collect_forward(), r1 = seeded.collect_forward();
    CHECK(r0 == got2);
    CHECK(r1.size() == 40 && std::equal(r1.begin(), r1.end(), got2.begin()));
    CHECK(seeded.collect_forward().empty());
    CHECK(seeded.launch_forward(img) == got2);               // the synchronous call still works next to the pipeline
    // a larger batch would re-create the context: refused while a batch is still in the ring, fine once it is collected
    CHECK(seeded.submit_forward(one));
    std::vector<float> big(img);
    big.insert(big.end(), img.begin(), img.end());
    bool refused = false;
    try { seeded.launch_forward(big); } catch (const std::exception &) { refused = true; }
    CHECK(refused);
    CHECK(seeded.collect_forward().size() == 40);
    const std::vector<float> rb = seeded.launch_forward(big);
    CHECK(rb.size() == 2 * got2.size() && std::equal(got2.begin(), got2.end(), rb.begin()) && std::equal(got2.begin(), got2.end(), rb.begin() + got2.size()));

    // ---- device list: the same net_abstract* shards its batch over a device group (here a rehearsal group: GPU 0 listed
    //      three times = three contexts, three host threads, blob broadcast by device copy); same logits, bit for bit ----
    {
        hip::net_hip many(c, 77);
        many.set_devices(std::vector<int>{0, 0, 0});
        CHECK(many.device_count() == 3);
        CHECK(many.launch_forward(img) == got2);
        CHECK(many.launch_forward(one) == std::vector<float>(got2.begin(), got2.begin() + 40));   // fewer images than members
        std::vector<float> seven;
        for (int r = 0; r < 7; ++r) seven.insert(seven.end(), one.begin(), one.end());
        const std::vector<float> r7 = many.launch_forward(seven);                                 // ragged shards 3 | 2 | 2
        CHECK(r7.size() == 7 * 40);
        for (int r = 0; r < 7; ++r) CHECK(std::equal(r7.begin() + 40 * r, r7.begin() + 40 * (r + 1), got2.begin()));
    }

    // ---- weights on disk: save, re-create from the file alone, same logits ----
    const char *path = "/tmp/test_net_hip.vhblob";
    seeded.save_weights(path);
    hip::net_hip loaded = hip::net_hip::from_file(path, VH_DTYPE_FP16);
    CHECK(loaded.is_vit() && loaded.n_ins == 64 * 64 * 3 && !loaded.device_init);
    CHECK(loaded.launch_forward(img) == got2);
    bool threw = false;
    try { hip::net_hip::from_file("/tmp/does_not_exist.vhblob", VH_DTYPE_FP16); } catch (const std::exception &) { threw = true; }
    CHECK(threw);
    // a damaged payload byte: from_file verifies the checksum (as vh_load_weights_file does) instead of loading it
    {
        FILE *f = fopen(path, "r+b");
        CHECK(f != nullptr);
        if (f) { fseek(f, 4096, SEEK_SET); int ch = fgetc(f); fseek(f, 4096, SEEK_SET); fputc(ch ^ 1, f); fclose(f); }
        threw = false;
        try { hip::net_hip::from_file(path, VH_DTYPE_FP16); } catch (const std::exception &) { threw = true; }
        CHECK(threw);
    }
    remove(path);

    // ---- filter_image / get_filtered_image: 24 frames in flight, the 25th is dropped, FIFO results ----
    net::image_set fs;
    fs.original_x_pos = fs.original_y_pos = 0;
    fs.original_h = 48; fs.original_w = 100;
    std::vector<std::vector<unsigned char>> frames(25, std::vector<unsigned char>(48 * 100));
    for (int i = 0; i < 25; ++i)
        for (size_t j = 0; j < frames[i].size(); ++j) frames[i][j] = (unsigned char)((j * 7 + i * 13 + (j >> 5)) & 0xFF);
    for (int i = 0; i < 25; ++i) { fs.resized_image_data = frames[i]; seeded.filter_image(fs); }   // prints PILA LLENA once
    std::vector<unsigned char> want8(48 * 100);
    for (int i = 0; i < 24; ++i)
    {
        const net::image_set r = seeded.get_filtered_image();
        oracle_filter3x3(frames[i].data(), want8.data(), 48, 100, 0);
        CHECK(r.original_h == 48 && r.original_w == 100 && r.resized_image_data == want8);
    }
    CHECK(seeded.get_filtered_image().resized_image_data.empty());     // PILA VACIA
}

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "cpu";
    host_checks();
    if (mode == "gpu") gpu_checks();
    printf("%s: %d failure(s)\n", mode.c_str(), failures);
    return failures ? 1 : 0;
}
