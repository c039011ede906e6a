// netAbstract.h — the plugin interface every backend implements.
//
// Re-declared from scratch, signature-for-signature compatible with the reference's
// include/netAbstract.h:8-21 (nine pure virtuals + virtual destructor), so that an application
// holding a `net::net_abstract*` can swap `fpga::net_fpga` for `hip::net_hip` without edits.
#ifndef NETABSTRACT_H
#define NETABSTRACT_H

#include <defines.h>

namespace net
{
    class net_abstract
    {
    public:
        virtual ~net_abstract() {}

        // network description held by the backend (inverse of the constructor's flatten)
        virtual net_data get_net_data() = 0;

        // THE HOT PATH: one forward pass, inputs in, result out (returned by value)
        virtual std::vector<DATA_TYPE> launch_forward(const std::vector<DATA_TYPE> &inputs) = 0;

        // training entry points (bodies are commented out in the reference backend)
        virtual void init_gradient(const net_sets &sets) = 0;
        virtual std::vector<DATA_TYPE> launch_gradient(size_t iterations, DATA_TYPE error_threshold,
                                                       DATA_TYPE multiplier) = 0; // per-iteration errors
        virtual void print_inner_vals() = 0;

        // wall time in microseconds of the last gradient / forward call (0 without PERFORMANCE)
        virtual signed long get_gradient_performance() = 0;
        virtual signed long get_forward_performance() = 0;

        // 8-bit image filter pipeline (separate workload, out of this backend's scope)
        virtual void filter_image(const image_set &set) = 0;
        virtual image_set get_filtered_image() = 0;
    };
}
#endif
