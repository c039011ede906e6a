// netHIP.h — hip::net_hip, the MI355X implementer of net::net_abstract.
//
// Mirrors the shape of the reference's fpga::net_fpga (include/netFPGA.h:17-71): same
// constructor signature, same nine overrides, same public field names, lazy device
// initialisation in launch_forward (netFPGA.cpp:242-253).  Plain C++ (gnu++14): no HIP headers,
// everything device-side goes through the C ABI in include/vithip.h.
//
// Two modes:
//   * MLP mode  — the reference's real semantics: net_hip(net_data, derivate, random); one or more
//                 input vectors of n_ins floats -> output vectors of n_p_l[last] floats.
//   * ViT mode  — additive: net_hip(vh_config, seed) / net_hip(vh_config, blob, bytes);
//                 launch_forward takes B x image x image x channels floats (NHWC) and returns
//                 B x classes logits, B = inputs.size() / floats-per-image.
//
// Deliberate deviations from the reference (all listed in DESIGN.md):
//   - errors are reported by throwing std::runtime_error (the reference prints and calls exit via
//     aocl_utils::checkError); set the environment variable VH_FATAL=1 to get print + exit(1);
//   - per-instance device state (the reference keeps namespace globals shared by all instances);
//   - members are initialised, the rule-of-five members really copy/move, get_net_data() is the
//     exact inverse of the constructor (netFPGA.cpp:122-124, 176-199, 206-237 are UB / TODO there);
//   - launch_forward's input length: the reference reads the first n_ins values of whatever it is given
//     (netFPGA.cpp:265-267); here inputs.size() must be a positive MULTIPLE of n_ins -- k * n_ins values are a batch of
//     k vectors / images (the interface has no batch argument) -- and any other length is an error, not a truncation.
#ifndef NETHIP_H
#define NETHIP_H

#include <cstdint>
#include <netAbstract.h>
#include <string>
#include <vector>
#include <vithip.h>

namespace hip
{
    class net_hip : public net::net_abstract
    {
    public:
        // ---- same public fields as fpga::net_fpga (netFPGA.h:22-36) ----
        int n_ins;       // MLP: inputs per vector; ViT: floats per image
        int n_layers;
        int *n_p_l;      // neurons per layer (MLP mode), nullptr in ViT mode
        int n_neurons;
        int n_params;    // MLP: total weights; ViT: 0 (see vit_param_count())

        DATA_TYPE *params; // flattened weights, layer-major / neuron-major / input-minor
        int activations;   // VH_ACT_* (1 = "RELU2", the reference's stored value, netFPGA.cpp:79)
        DATA_TYPE *bias;

        int n_sets;
        bool gradient_init;

        int64_t gradient_performance;
        int64_t forward_performance;

        // ---- device state (per instance, created lazily) ----
        bool device_init;
        int device;

    private:
        net_hip() = delete;
        bool vit_mode;
        vh_config vcfg;
        uint64_t vit_seed;
        int ring_slots, ring_batch; // 0 = no pipeline
        int filter_kind, filt_h, filt_w;
        std::string vit_blob;  // host copy of the canonical blob when constructed from one
        vh_mlp *mlp;
        vh_ctx *vit;
        vh_filter *filt;
        std::vector<int> devices; // ViT mode: more than one entry = device group (vh_group_*), images sharded across them
        vh_group *grp;
        void release();
        void copy_from(const net_hip &rh);
        void steal(net_hip &rh);
        void ensure_device(int batch);
        [[noreturn]] void die(const char *what, const char *detail) const;

    public:
        ~net_hip();
        net_hip(const net::net_data &data, bool derivate, bool random);
        net_hip(const vh_config &cfg, uint64_t seed, int device_index = 0);
        net_hip(const vh_config &cfg, const void *blob, size_t blob_bytes, int device_index = 0);
        net_hip(net_hip &&rh);
        net_hip &operator=(net_hip &&rh);
        net_hip &operator=(const net_hip &rh);

        net::net_data get_net_data() override;
        std::vector<DATA_TYPE> launch_forward(const std::vector<DATA_TYPE> &inputs) override;
        void init_gradient(const net::net_sets &sets) override;
        std::vector<DATA_TYPE> launch_gradient(size_t iterations, DATA_TYPE error_threshold, DATA_TYPE multiplier) override;
        void print_inner_vals() override;
        signed long get_gradient_performance() override;
        signed long get_forward_performance() override;
        void filter_image(const net::image_set &set) override;
        net::image_set get_filtered_image() override;

        // ---- additions ----
        bool is_vit() const { return vit_mode; }
        void set_activation(int vh_act_code);       // MLP mode, before the first forward
        size_t vit_param_count() const;             // ViT mode: number of fp32 parameters
        double last_kernel_ms();                    // ViT mode: device time of the last forward
        // ViT mode, pipelined: the shape of filter_image/get_filtered_image (netFPGA.cpp:292-365) applied to
        // launch_forward.  submit_forward returns false when every slot is in flight ("PILA LLENA"),
        // collect_forward returns an empty vector when nothing is ("PILA VACIA"); results come back in FIFO order.
        // ViT mode, weights on disk: the canonical blob as a file (vithip.h, vh_*_weights_file).  from_file reads the
        // model shape from the file's header (host only; the device is still touched lazily by the first forward).
        static net_hip from_file(const char *blob_path, int vh_dtype, int device_index = 0);
        void save_weights(const char *blob_path);
        void set_filter(int vh_filter_kind);        // VH_FILTER_*; before the first filter_image
        void set_pipeline(int slots, int max_batch_per_slot);
        // ViT mode, several GPUs of one node behind ONE net_abstract*: launch_forward shards the batch's images over the
        // listed devices (contiguous ranges, weights broadcast once over xGMI, no data-path collective; vithip.h
        // vh_group_*).  Call before the first forward.  Without a call the environment variable VH_DEVICES (e.g.
        // "0,1,2,3") is consulted at the first forward; a single ordinal just selects that device.  vcfg.max_batch is the
        // per-device capacity.  The pipelined submit/collect ring is a single-device feature.
        void set_devices(const std::vector<int> &device_ordinals);
        size_t device_count() const { return devices.empty() ? 1 : devices.size(); }
        bool submit_forward(const std::vector<DATA_TYPE> &inputs);
        std::vector<DATA_TYPE> collect_forward();
    };
}

#endif
