// defines.h — plugin-boundary PODs, re-declared from scratch for the MI355X backend.
//
// API-compatible with LimpBunion22/VIT-FPGA's def/defines.h (reference file:line cited per
// item): a caller written against the reference's header compiles unchanged against this one.
// Differences: self-contained (<cstddef>; the reference relies on a transitive size_t), plain
// `struct` declarations, documented fields.
#ifndef DEFINES_H
#define DEFINES_H

#include <cstddef>
#include <vector>

// compile-time switches of the reference (def/defines.h:8-10)
#define ASSERT
#define PERFORMANCE      // get_*_performance() report microseconds instead of 0
#define DATA_TYPE float  // element type of every interface vector

namespace net
{
    // declared value range of inputs/activations (def/defines.h:11-12)
    constexpr DATA_TYPE MAX_RANGE = 1;
    constexpr DATA_TYPE MIN_RANGE = -1;

    // A fully-connected network (def/defines.h:14-23).
    //   params[l][j][k] : weight of input k of neuron j of layer l  (fan-in of layer 0 = n_ins,
    //                     of layer l = n_p_l[l-1])
    //   bias[l][j]      : bias of neuron j of layer l
    //   activations     : per-neuron activation selector; declared but unused by the reference
    //                     ("TODO: IMPLEMENTAR ACTIVATIONS", def/defines.h:21-22)
    struct net_data
    {
        size_t n_ins;
        size_t n_layers;
        std::vector<size_t> n_p_l;
        std::vector<std::vector<std::vector<DATA_TYPE>>> params;
        std::vector<std::vector<DATA_TYPE>> bias;
        std::vector<std::vector<DATA_TYPE>> activations;
    };

    // training sets (def/defines.h:25-29); only consumed by the gradient stubs
    struct net_sets
    {
        std::vector<std::vector<DATA_TYPE>> set_ins;
        std::vector<std::vector<DATA_TYPE>> set_outs;
    };

    // 8-bit single-channel image, row-major (def/defines.h:31-38)
    struct image_set
    {
        std::vector<unsigned char> resized_image_data;
        size_t original_x_pos;
        size_t original_y_pos;
        size_t original_h;
        size_t original_w;
    };
}
#endif
