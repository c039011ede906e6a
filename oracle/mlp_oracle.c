/*
 * mlp_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * CPU restatement of what the reference's `network_v1` FPGA kernel must compute, as far as
 * the reference pins it.  The kernel source/bitstream is absent (named only at
 * /root/reference/src/netFPGA.cpp:250), so the arithmetic is inferred from its argument
 * list and the host-side layout:
 *   arg0 inputs[n_ins], arg1 params[n_params], arg2 bias[n_neurons], arg3 outs[n_p_l[L-1]],
 *   arg4 npl[n_layers]                                   (netFPGA.cpp:427-436)
 *   arg5 n_layers, arg6 n_ins                            (netFPGA.cpp:499-502)
 *   params: layer-major, neuron-major, input-minor       (netFPGA.cpp:91-106)
 *   fan-in of layer 0 = n_ins, of layer i = n_p_l[i-1]   (netFPGA.cpp:68-76)
 * i.e. a chain of dense layers a_l[j] = act(bias[j] + sum_k W_l[j][k] a_{l-1}[k]).
 * PARITY UNPINNED: `activations = 1 // RELU2` (netFPGA.cpp:79) is never defined in the
 * reference; the codes are those of include/vithip.h.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static float act_apply(int act, float v) {
    switch (act) {
    case 1: return v < 0.f ? 0.f : (v > 1.f ? 1.f : v);           /* RELU2 (this build) */
    case 2: return v < 0.f ? 0.f : v;                             /* RELU */
    case 3: return v < -1.f ? -1.f : (v > 1.f ? 1.f : v);         /* HARDTANH */
    case 4: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    default: return v;
    }
}

int oracle_mlp_forward(int n_ins, int n_layers, const int* n_p_l, const float* params,
                       const float* bias, int activation, const float* inputs, float* outputs) {
    int widest = n_ins;
    for (int l = 0; l < n_layers; ++l) if (n_p_l[l] > widest) widest = n_p_l[l];
    float* cur = (float*)malloc(sizeof(float) * (size_t)widest);
    float* nxt = (float*)malloc(sizeof(float) * (size_t)widest);
    if (!cur || !nxt) return 1;
    memcpy(cur, inputs, sizeof(float) * (size_t)n_ins);
    size_t woff = 0, boff = 0;
    int fan_in = n_ins;
    for (int l = 0; l < n_layers; ++l) {
        const int n_out = n_p_l[l];
        for (int j = 0; j < n_out; ++j) {
            const float* wrow = params + woff + (size_t)j * fan_in;
            double s = 0.0; /* sequential k order, double accumulate: the tightest statement */
            for (int k = 0; k < fan_in; ++k) s += (double)wrow[k] * (double)cur[k];
            nxt[j] = act_apply(activation, (float)(s + (double)bias[boff + j]));
        }
        woff += (size_t)n_out * fan_in;
        boff += (size_t)n_out;
        fan_in = n_out;
        float* t = cur; cur = nxt; nxt = t;
    }
    memcpy(outputs, cur, sizeof(float) * (size_t)n_p_l[n_layers - 1]);
    free(cur); free(nxt);
    return 0;
}

/* value formula of the reference ctor's random branch (netFPGA.cpp:82-88):
 * float(r % 200 - 100) / 100 with r a non-negative pseudo-random int.  The reference draws r
 * from libc rand() without srand; a fixed LCG stands in so tests are reproducible. */
void oracle_mlp_random_params(float* params, size_t n_params, float* bias, size_t n_neurons,
                              uint32_t seed) {
    uint32_t s = seed ? seed : 1u;
    for (size_t i = 0; i < n_params + n_neurons; ++i) {
        s = s * 1103515245u + 12345u;
        const int r = (int)((s >> 16) & 0x7FFF);
        const float v = (float)(r % 200 - 100) / 100;
        if (i < n_params) params[i] = v; else bias[i - n_params] = v;
    }
}
