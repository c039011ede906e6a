// netHIP.cpp — hip::net_hip: the host half of the drop-in for the reference's src/netFPGA.cpp.
//
// Compiled by plain g++ into libnetHIP.a (the counterpart of libnetFPGA.a, reference
// Makefile:75); links against libvithip.so.  Reference behaviour each member mirrors is cited
// inline as netFPGA.cpp:<line>.
#include <netHIP.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <utility>

namespace hip
{
    using namespace std;

    namespace
    {
        const int IMAGE_HEIGHT = 1080; // netFPGA.h:14-15, only reported by get_filtered_image()
        const int IMAGE_WIDTH = 1920;
        const int FILTER_SLOTS = 24; // BATCH_SIZE of the reference's ring (netFPGA.cpp:47)

        size_t floats_per_image(const vh_config &c)
        {
            return (size_t)c.image_size * c.image_size * c.channels;
        }
    }

    void net_hip::die(const char *what, const char *detail) const
    {
        string msg = string("net_hip: ") + what + ": " + (detail ? detail : "?");
        const char *fatal = getenv("VH_FATAL");
        if (fatal && fatal[0] == '1')
        {
            // the reference's convention: print, release, exit (checkError -> cleanup -> exit)
            cerr << msg << "\n";
            exit(1);
        }
        throw runtime_error(msg);
    }

    // ---- MLP-mode constructor: same bookkeeping and flatten order as netFPGA.cpp:58-109 ----
    net_hip::net_hip(const net::net_data &data, bool derivate, bool random)
        : n_ins((int)data.n_ins), n_layers((int)data.n_p_l.size()), n_p_l(nullptr), n_neurons(0), n_params(0),
          params(nullptr), activations(VH_ACT_RELU2), bias(nullptr), n_sets(0), gradient_init(false),
          gradient_performance(0), forward_performance(0), device_init(false), device(0), vit_mode(false),
          vcfg(), vit_seed(0), ring_slots(0), ring_batch(0), filter_kind(VH_FILTER_BLUR3), filt_h(0), filt_w(0), mlp(nullptr), vit(nullptr), filt(nullptr), grp(nullptr)
    {
        (void)derivate; // ignored by the reference as well
        // every check comes BEFORE the first allocation: a constructor that throws does not run its destructor
        if (n_layers <= 0 || n_ins <= 0)
            die("constructor", "net_data needs n_ins > 0 and at least one layer");
        for (int l = 0; l < n_layers; l++)
        {
            const long long fan_in = (l == 0 ? (long long)data.n_ins : (long long)data.n_p_l[l - 1]);
            if ((long long)data.n_p_l[l] <= 0 || (long long)data.n_p_l[l] > 0x7fffffff / 4)
                die("constructor", "net_data.n_p_l entries must be positive");
            if (!random)
            {
                if (data.params.size() <= (size_t)l || data.params[l].size() != data.n_p_l[l] || data.bias.size() <= (size_t)l ||
                    data.bias[l].size() != data.n_p_l[l])
                    die("constructor", "net_data.params / bias do not match n_p_l");
                for (size_t j = 0; j < data.n_p_l[l]; j++)
                    if ((long long)data.params[l][j].size() != fan_in)
                        die("constructor", "net_data.params row length differs from the layer's fan-in");
            }
        }
        n_p_l = new int[n_layers];
        for (int l = 0; l < n_layers; l++)
        {
            n_p_l[l] = (int)data.n_p_l[l];
            n_neurons += n_p_l[l];
            n_params += n_p_l[l] * (l == 0 ? n_ins : n_p_l[l - 1]);
        }
        params = new DATA_TYPE[n_params];
        bias = new DATA_TYPE[n_neurons];

        if (random)
        {
            // value formula and draw order of the reference (netFPGA.cpp:82-88): libc rand(), no srand
            for (int i = 0; i < n_params; i++)
                params[i] = DATA_TYPE(rand() % 200 - 100) / 100;
            for (int i = 0; i < n_neurons; i++)
                bias[i] = DATA_TYPE(rand() % 200 - 100) / 100;
            return;
        }
        int p = 0, q = 0;
        for (int l = 0; l < n_layers; l++)
        {
            const int fan_in = (l == 0 ? n_ins : n_p_l[l - 1]);
            for (int j = 0; j < n_p_l[l]; j++)
            {
                memcpy(params + p, data.params[l][j].data(), sizeof(DATA_TYPE) * fan_in);
                p += fan_in;
                bias[q++] = data.bias[l][j];
            }
        }
    }

    // ---- ViT-mode constructors (host only; the device is touched in launch_forward) ----
    net_hip::net_hip(const vh_config &cfg, uint64_t seed, int device_index)
        : n_ins((int)floats_per_image(cfg)), n_layers(cfg.layers), n_p_l(nullptr), n_neurons(0), n_params(0), params(nullptr),
          activations(VH_ACT_GELU), bias(nullptr), n_sets(0), gradient_init(false), gradient_performance(0),
          forward_performance(0), device_init(false), device(device_index), vit_mode(true), vcfg(cfg), vit_seed(seed),
          ring_slots(0), ring_batch(0), filter_kind(VH_FILTER_BLUR3), filt_h(0), filt_w(0), mlp(nullptr), vit(nullptr), filt(nullptr), grp(nullptr)
    {
        if (vh_weight_blob_bytes(&vcfg) == 0)
            die("constructor", "unsupported vh_config");
    }

    net_hip::net_hip(const vh_config &cfg, const void *blob, size_t blob_bytes, int device_index)
        : net_hip(cfg, 0, device_index)
    {
        if (!blob || blob_bytes != vh_weight_blob_bytes(&vcfg))
            die("constructor", "weight blob size does not match vh_config");
        vit_blob.assign((const char *)blob, blob_bytes);
    }

    void net_hip::release()
    {
        if (mlp)
            vh_mlp_destroy(mlp);
        if (vit)
            vh_destroy(vit);
        if (grp)
            vh_group_destroy(grp);
        if (filt)
            vh_filter_destroy(filt);
        mlp = nullptr;
        vit = nullptr;
        grp = nullptr;
        filt = nullptr;
        device_init = false;
        delete[] n_p_l;
        delete[] params;
        delete[] bias;
        n_p_l = nullptr;
        params = bias = nullptr;
    }

    net_hip::~net_hip() { release(); }

    void net_hip::steal(net_hip &rh)
    {
        n_ins = rh.n_ins; n_layers = rh.n_layers; n_p_l = rh.n_p_l; n_neurons = rh.n_neurons; n_params = rh.n_params;
        params = rh.params; activations = rh.activations; bias = rh.bias; n_sets = rh.n_sets;
        gradient_init = rh.gradient_init; gradient_performance = rh.gradient_performance;
        forward_performance = rh.forward_performance; device_init = rh.device_init; device = rh.device;
        vit_mode = rh.vit_mode; vcfg = rh.vcfg; vit_seed = rh.vit_seed; vit_blob = std::move(rh.vit_blob);
        ring_slots = rh.ring_slots; ring_batch = rh.ring_batch;
        filter_kind = rh.filter_kind; filt_h = rh.filt_h; filt_w = rh.filt_w;
        mlp = rh.mlp; vit = rh.vit; filt = rh.filt; grp = rh.grp; devices = std::move(rh.devices);
        rh.n_p_l = nullptr; rh.params = nullptr; rh.bias = nullptr; rh.mlp = nullptr; rh.vit = nullptr; rh.filt = nullptr; rh.grp = nullptr;
        rh.device_init = false;
    }

    void net_hip::copy_from(const net_hip &rh)
    {
        n_ins = rh.n_ins; n_layers = rh.n_layers; n_neurons = rh.n_neurons; n_params = rh.n_params;
        activations = rh.activations; n_sets = rh.n_sets; gradient_init = rh.gradient_init;
        gradient_performance = rh.gradient_performance; forward_performance = rh.forward_performance;
        device = rh.device; vit_mode = rh.vit_mode; vcfg = rh.vcfg; vit_seed = rh.vit_seed; vit_blob = rh.vit_blob;
        ring_slots = rh.ring_slots; ring_batch = rh.ring_batch; devices = rh.devices;
        filter_kind = rh.filter_kind; filt_h = filt_w = 0; // the copy builds its own pipeline on first use
        if (!vit_mode)
        {
            n_p_l = new int[n_layers];
            params = new DATA_TYPE[n_params];
            bias = new DATA_TYPE[n_neurons];
            memcpy(n_p_l, rh.n_p_l, sizeof(int) * n_layers);
            memcpy(params, rh.params, sizeof(DATA_TYPE) * n_params);
            memcpy(bias, rh.bias, sizeof(DATA_TYPE) * n_neurons);
        }
        // device objects are re-created lazily by the copy's first launch_forward
    }

    net_hip::net_hip(net_hip &&rh)
        : n_p_l(nullptr), params(nullptr), bias(nullptr), device_init(false), mlp(nullptr), vit(nullptr), filt(nullptr), grp(nullptr)
    {
        steal(rh);
    }

    net_hip &net_hip::operator=(net_hip &&rh)
    {
        if (this != &rh)
        {
            release();
            steal(rh);
        }
        return *this;
    }

    net_hip &net_hip::operator=(const net_hip &rh)
    {
        if (this != &rh)
        {
            release();
            copy_from(rh);
        }
        return *this;
    }

    // exact inverse of the MLP constructor (the reference's version is a broken TODO, netFPGA.cpp:206-237)
    net::net_data net_hip::get_net_data()
    {
        net::net_data d;
        d.n_ins = (size_t)n_ins;
        d.n_layers = (size_t)n_layers;
        if (vit_mode)
            return d; // a ViT is not expressible as net_data; weights travel as the canonical blob
        int p = 0, q = 0;
        for (int l = 0; l < n_layers; l++)
        {
            const int fan_in = (l == 0 ? n_ins : n_p_l[l - 1]);
            d.n_p_l.push_back((size_t)n_p_l[l]);
            d.params.emplace_back();
            d.bias.emplace_back();
            for (int j = 0; j < n_p_l[l]; j++)
            {
                d.params[l].emplace_back(params + p, params + p + fan_in);
                p += fan_in;
                d.bias[l].push_back(bias[q++]);
            }
        }
        return d;
    }

    void net_hip::set_activation(int code)
    {
        if (device_init)
            die("set_activation", "must be called before the first launch_forward");
        activations = code;
    }

    size_t net_hip::vit_param_count() const
    {
        return vit_mode ? (vh_weight_blob_bytes(&vcfg) - 64) / 4 : 0;
    }

    // lazy device init + weight upload: the roles of _init_program/_init_kernel/_load_params
    // (netFPGA.cpp:242-260), done once per instance instead of via shared globals
    void net_hip::ensure_device(int batch)
    {
        if (!vit_mode)
        {
            if (device_init)
                return;
            if (vh_mlp_create(device, n_ins, n_layers, n_p_l, activations, &mlp) != VH_OK)
                die("vh_mlp_create", vh_last_error(nullptr));
            if (vh_mlp_load_params(mlp, params, (size_t)n_params, bias, (size_t)n_neurons) != VH_OK)
                die("vh_mlp_load_params", vh_mlp_last_error(mlp));
            device_init = true;
            return;
        }
        if (!device_init && devices.empty())
        {
            // VH_DEVICES="0,1,2,3": device list from the environment (consulted once, at the first forward)
            const char *e = getenv("VH_DEVICES");
            for (const char *p = e; p && *p;)
            {
                char *end = nullptr;
                const long v = strtol(p, &end, 10);
                if (end == p)
                    break;
                devices.push_back((int)v);
                p = (*end == ',') ? end + 1 : end;
            }
        }
        if (devices.size() == 1)
            device = devices[0];
        if (devices.size() > 1)
        {
            const int n = (int)devices.size(), per = (batch + n - 1) / n;
            if (device_init && per <= vcfg.max_batch)
                return;
            if (ring_slots > 0)
                die("launch_forward", "the submit/collect pipeline is a single-device feature (set_devices with one device)");
            if (grp)
                vh_group_destroy(grp);
            grp = nullptr;
            device_init = false;
            if (per > vcfg.max_batch)
                vcfg.max_batch = per;
            if (vh_group_create(&vcfg, devices.data(), n, &grp) != VH_OK)
                die("vh_group_create", vh_last_error(nullptr));
            const int rc = vit_blob.empty() ? vh_group_init_weights_seeded(grp, vit_seed)
                                            : vh_group_load_weights(grp, vit_blob.data(), vit_blob.size());
            if (rc != VH_OK)
                die("group weights", vh_group_last_error(grp));
            device_init = true;
            return;
        }
        if (device_init && batch <= vcfg.max_batch)
            return;
        if (vit)
        {
            // growing the workspace re-creates the context: batches still in the pipeline ring would be dropped silently
            int free_slots = 0;
            if (ring_slots > 0 && vh_ring_free_slots(vit, &free_slots) == VH_OK && free_slots != ring_slots)
                die("launch_forward", "batches still in flight in the pipeline: collect them before a larger batch re-creates the context");
            vh_destroy(vit); // grow the workspace for a larger batch
            vit = nullptr;
            device_init = false;
        }
        if (batch > vcfg.max_batch)
            vcfg.max_batch = batch;
        if (vh_create(&vcfg, device, &vit) != VH_OK)
            die("vh_create", vh_last_error(nullptr));
        const int rc = vit_blob.empty() ? vh_init_weights_seeded(vit, vit_seed)
                                        : vh_load_weights(vit, vit_blob.data(), vit_blob.size());
        if (rc != VH_OK)
            die("weights", vh_last_error(vit));
        if (ring_slots > 0 && vh_ring_create(vit, ring_slots, ring_batch) != VH_OK)
            die("vh_ring_create", vh_last_error(vit));
        device_init = true;
    }

    // ---- weights on disk (ViT mode) ----
    net_hip net_hip::from_file(const char *blob_path, int vh_dtype, int device_index)
    {
        vh_config c;
        if (vh_blob_file_config(blob_path, &c) != VH_OK)
            throw runtime_error(string("net_hip::from_file: ") + vh_last_error(nullptr));
        c.dtype = vh_dtype;
        // the file as a memory blob: length, header AND checksum verified by the library (same check as vh_load_weights_file)
        string bytes(vh_weight_blob_bytes(&c), '\0');
        if (vh_blob_file_read(blob_path, &bytes[0], bytes.size()) != VH_OK)
            throw runtime_error(string("net_hip::from_file: ") + vh_last_error(nullptr));
        return net_hip(c, bytes.data(), bytes.size(), device_index);
    }

    void net_hip::save_weights(const char *blob_path)
    {
        if (!vit_mode)
            die("save_weights", "only available in ViT mode (MLP mode: get_net_data())");
        ensure_device(1);
        vh_ctx *c0 = vit;
        if (grp && vh_group_member(grp, 0, &c0, nullptr) != VH_OK)
            die("vh_group_member", vh_group_last_error(grp));
        if (vh_save_weights_file(c0, blob_path) != VH_OK)
            die("vh_save_weights_file", vh_last_error(c0));
    }

    // ---- pipelined forward (ViT mode) ----
    void net_hip::set_pipeline(int slots, int max_batch_per_slot)
    {
        if (!vit_mode)
            die("set_pipeline", "only available in ViT mode");
        if (slots < 1 || slots > 64 || max_batch_per_slot < 1)
            die("set_pipeline", "slots must be 1..64 and max_batch_per_slot positive");
        if (vit)
        {
            int free_slots = 0;
            if (ring_slots > 0 && vh_ring_free_slots(vit, &free_slots) == VH_OK && free_slots != ring_slots)
                die("set_pipeline", "batches still in flight: collect them first");
            vh_destroy(vit);
            vit = nullptr;
            device_init = false;
        }
        ring_slots = slots;
        ring_batch = max_batch_per_slot;
    }

    bool net_hip::submit_forward(const vector<DATA_TYPE> &inputs)
    {
        if (!vit_mode || ring_slots == 0)
            die("submit_forward", "call set_pipeline first (ViT mode)");
        if (inputs.empty() || inputs.size() % (size_t)n_ins != 0)
            die("submit_forward", "inputs.size() must be a positive multiple of n_ins");
        const int count = (int)(inputs.size() / (size_t)n_ins);
        if (count > ring_batch)
            die("submit_forward", "more images than a pipeline slot holds");
        ensure_device(ring_batch);
        const int rc = vh_ring_submit(vit, inputs.data(), count);
        if (rc == VH_ERR_RING_FULL)
        {
            cout << "PILA LLENA\n"; // the reference's own overflow report (netFPGA.cpp:358-361); the batch is not queued
            return false;
        }
        if (rc != VH_OK)
            die("vh_ring_submit", vh_last_error(vit));
        return true;
    }

    vector<DATA_TYPE> net_hip::collect_forward()
    {
        if (!vit_mode || ring_slots == 0)
            die("collect_forward", "call set_pipeline first (ViT mode)");
        vector<DATA_TYPE> out;
        if (!vit)
        {
            cout << "PILA VACIA\n";
            return out;
        }
        out.resize((size_t)ring_batch * vcfg.classes);
        int count = 0;
        const int rc = vh_ring_collect(vit, out.data(), &count);
        if (rc == VH_ERR_RING_EMPTY)
        {
            cout << "PILA VACIA\n"; // netFPGA.cpp:330-333
            out.clear();
            return out;
        }
        if (rc != VH_OK)
            die("vh_ring_collect", vh_last_error(vit));
        out.resize((size_t)count * vcfg.classes);
        return out;
    }

    vector<DATA_TYPE> net_hip::launch_forward(const vector<DATA_TYPE> &inputs)
    {
        if (inputs.empty() || inputs.size() % (size_t)n_ins != 0)
            die("launch_forward", "inputs.size() must be a positive multiple of n_ins");
        const int count = (int)(inputs.size() / (size_t)n_ins);
        ensure_device(count);
#ifdef PERFORMANCE
        const auto start = chrono::high_resolution_clock::now(); // same window as netFPGA.cpp:262-284
#endif
        vector<DATA_TYPE> out;
        if (vit_mode)
        {
            out.resize((size_t)count * vcfg.classes);
            if (grp)
            {
                if (vh_group_forward(grp, inputs.data(), count, out.data()) != VH_OK)
                    die("vh_group_forward", vh_group_last_error(grp));
            }
            else if (vh_forward(vit, inputs.data(), count, out.data()) != VH_OK)
                die("vh_forward", vh_last_error(vit));
        }
        else
        {
            out.resize((size_t)count * n_p_l[n_layers - 1]);
            if (vh_mlp_forward(mlp, inputs.data(), count, out.data()) != VH_OK)
                die("vh_mlp_forward", vh_mlp_last_error(mlp));
        }
#ifdef PERFORMANCE
        forward_performance = chrono::duration_cast<chrono::microseconds>(chrono::high_resolution_clock::now() - start).count();
#endif
        return out;
    }

    void net_hip::set_devices(const vector<int> &device_ordinals)
    {
        if (!vit_mode)
            die("set_devices", "only available in ViT mode");
        if (device_init)
            die("set_devices", "call it before the first forward");
        if (device_ordinals.empty())
            die("set_devices", "empty device list");
        devices = device_ordinals;
    }

    double net_hip::last_kernel_ms()
    {
        double ms = 0.0;
        if (vit && vh_last_kernel_ms(vit, &ms) == VH_OK)
            return ms;
        return 0.0;
    }

    // ---- training ----
    // The reference's bodies are commented-out code (netFPGA.cpp:518-580); their observable behaviour -- init_gradient
    // does nothing, launch_gradient returns `iterations` zeros -- is kept wherever this build has nothing better: in ViT
    // mode, and in MLP mode before init_gradient.  In MLP mode the loop the comments sketch is implemented on the
    // device (vh_mlp_init_gradient / vh_mlp_launch_gradient; the definitions the reference leaves open are stated in
    // include/vithip.h and restated on the CPU in oracle/mlp_oracle.c -- PARITY UNPINNED).
    void net_hip::init_gradient(const net::net_sets &sets)
    {
        if (vit_mode || gradient_init) // a second call is ignored, as in the reference's sketch (:520, :541)
            return;
        const size_t n = sets.set_ins.size();
        if (n == 0 && sets.set_outs.empty())
            return; // nothing to train on: launch_gradient keeps returning zeros (no device is touched)
        if (sets.set_outs.size() != n)
            die("init_gradient", "set_ins and set_outs must hold the same number of sets");
        const size_t n_out = (size_t)n_p_l[n_layers - 1];
        vector<DATA_TYPE> ins, outs;
        ins.reserve(n * (size_t)n_ins);
        outs.reserve(n * n_out);
        for (size_t j = 0; j < n; j++)
        {
            if (sets.set_ins[j].size() != (size_t)n_ins || sets.set_outs[j].size() != n_out)
                die("init_gradient", "every set needs n_ins inputs and n_p_l[n_layers-1] outputs");
            ins.insert(ins.end(), sets.set_ins[j].begin(), sets.set_ins[j].end());
            outs.insert(outs.end(), sets.set_outs[j].begin(), sets.set_outs[j].end());
        }
        ensure_device(1);
        if (vh_mlp_init_gradient(mlp, ins.data(), outs.data(), (int)n) != VH_OK)
            die("vh_mlp_init_gradient", vh_mlp_last_error(mlp));
        n_sets = (int)n;
        gradient_init = true;
    }

    vector<DATA_TYPE> net_hip::launch_gradient(size_t iterations, DATA_TYPE error_threshold, DATA_TYPE multiplier)
    {
        vector<DATA_TYPE> errors(iterations, 0); // netFPGA.cpp:550, :579
        if (vit_mode || !gradient_init || iterations == 0)
            return errors;
#ifdef PERFORMANCE
        const auto start = chrono::high_resolution_clock::now(); // the window the reference sketches, :546-547, :566-568
#endif
        if (vh_mlp_launch_gradient(mlp, (int)iterations, error_threshold, multiplier, errors.data()) != VH_OK)
            die("vh_mlp_launch_gradient", vh_mlp_last_error(mlp));
        // the host copy follows the device, so that get_net_data() returns the trained net
        if (vh_mlp_read_params(mlp, params, (size_t)n_params, bias, (size_t)n_neurons) != VH_OK)
            die("vh_mlp_read_params", vh_mlp_last_error(mlp));
#ifdef PERFORMANCE
        gradient_performance = chrono::duration_cast<chrono::microseconds>(chrono::high_resolution_clock::now() - start).count();
#endif
        return errors;
    }

    void net_hip::print_inner_vals() {} // netFPGA.cpp:582-591

    signed long net_hip::get_gradient_performance()
    {
#ifdef PERFORMANCE
        return gradient_performance;
#else
        return 0;
#endif
    }

    signed long net_hip::get_forward_performance()
    {
#ifdef PERFORMANCE
        return forward_performance;
#else
        return 0;
#endif
    }

    // ---- image filter ----
    // The reference pushes single-channel 8-bit frames through a kernel `image_process` whose source is absent, with a
    // 24-slot ring of in-flight frames (netFPGA.cpp:292-365).  The ring behaviour is reproduced (non-blocking submit,
    // FIFO collect, "PILA LLENA" / "PILA VACIA" and a dropped frame / an empty result on overflow / underflow); the
    // arithmetic is this build's documented choice, a 3x3 filter (include/vithip.h, VH_FILTER_*).  The pipeline is
    // created lazily for the size of the first frame, as the reference sizes its buffers from the first set (:305).
    void net_hip::set_filter(int vh_filter_kind)
    {
        if (filt)
            die("set_filter", "must be called before the first filter_image");
        if (vh_filter_kind != VH_FILTER_BLUR3 && vh_filter_kind != VH_FILTER_SOBEL3)
            die("set_filter", "unknown filter kind");
        filter_kind = vh_filter_kind;
    }

    void net_hip::filter_image(const net::image_set &set)
    {
        if (set.original_h == 0 || set.original_w == 0 || set.resized_image_data.size() < set.original_h * set.original_w)
            die("filter_image", "resized_image_data holds fewer than original_h * original_w bytes");
        if (filt && ((size_t)filt_h != set.original_h || (size_t)filt_w != set.original_w))
            die("filter_image", "frame size differs from the first frame's");
        if (!filt)
        {
            if (vh_filter_create(device, (int)set.original_h, (int)set.original_w, FILTER_SLOTS, filter_kind, &filt) != VH_OK)
                die("vh_filter_create", vh_last_error(nullptr));
            filt_h = (int)set.original_h;
            filt_w = (int)set.original_w;
        }
        const int rc = vh_filter_submit(filt, set.resized_image_data.data());
        if (rc == VH_ERR_RING_FULL)
        {
            cout << "PILA LLENA\n"; // netFPGA.cpp:330-333: the frame is dropped
            return;
        }
        if (rc != VH_OK)
            die("vh_filter_submit", vh_filter_last_error(filt));
    }

    net::image_set net_hip::get_filtered_image()
    {
        net::image_set out;
        out.original_x_pos = 0;
        out.original_y_pos = 0;
        out.original_h = filt ? (size_t)filt_h : (size_t)IMAGE_HEIGHT; // the reference always reports 1080 x 1920 (:340-343)
        out.original_w = filt ? (size_t)filt_w : (size_t)IMAGE_WIDTH;
        if (!filt)
        {
            cout << "PILA VACIA\n";
            return out;
        }
        out.resized_image_data.resize((size_t)filt_h * filt_w);
        const int rc = vh_filter_collect(filt, out.resized_image_data.data());
        if (rc == VH_ERR_RING_EMPTY)
        {
            cout << "PILA VACIA\n"; // netFPGA.cpp:358-361
            out.resized_image_data.clear();
            return out;
        }
        if (rc != VH_OK)
            die("vh_filter_collect", vh_filter_last_error(filt));
        return out;
    }
}
