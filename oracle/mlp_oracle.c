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

/* ---- training (init_gradient / launch_gradient) -------------------------------------------------------
 * PARITY UNPINNED.  The reference's bodies are commented-out code (netFPGA.cpp:518-580) built on a vector library that
 * is not in the repository; what they show is the SHAPE of the loop, which is kept:
 *     per iteration:  for every set j: back-propagate set j; error_j = sum |.| of its output error   (:556-559)
 *                     accumulate the sets' gradients (:558), normalize_1() (:561), update the parameters (:562),
 *                     reset the accumulator (:563); errors[it] = sum_j error_j                       (:564)
 * What they do not define is chosen here and documented in include/vithip.h: the loss is 1/2 |a_L - t|^2 per set,
 * normalize_1 = mean over the sets, the update is  p -= multiplier * mean gradient, the error of an iteration is taken
 * BEFORE its update, and an iteration whose error is <= error_threshold ends the loop (later entries stay 0, the
 * value the reference initialises its result vector with, :550).  Plain fp32 arithmetic in a fixed order. */
static float act_deriv(int act, float z) {
    switch (act) {
    case 1: return (z > 0.f && z < 1.f) ? 1.f : 0.f;              /* RELU2 */
    case 2: return z > 0.f ? 1.f : 0.f;                            /* RELU */
    case 3: return (z > -1.f && z < 1.f) ? 1.f : 0.f;             /* HARDTANH */
    case 4: {                                                      /* GELU: Phi(z) + z phi(z) */
        const float cdf = 0.5f * (1.0f + erff(z * 0.70710678118654752440f));
        const float pdf = 0.39894228040143267794f * expf(-0.5f * z * z);
        return cdf + z * pdf;
    }
    default: return 1.f;
    }
}

int oracle_mlp_train(int n_ins, int n_layers, const int* n_p_l, float* params, float* bias, int activation,
                     const float* set_ins, const float* set_outs, int n_sets, int iterations, float error_threshold,
                     float multiplier, float* errors) {
    if (n_sets <= 0 || iterations < 0) return 1;
    size_t n_neurons = 0, n_params = 0;
    int fan = n_ins;
    for (int l = 0; l < n_layers; ++l) { n_neurons += (size_t)n_p_l[l]; n_params += (size_t)n_p_l[l] * fan; fan = n_p_l[l]; }
    const int n_out = n_p_l[n_layers - 1];
    /* per set and neuron: pre-activation z, activation a, delta d */
    float* z = (float*)malloc(sizeof(float) * n_neurons * (size_t)n_sets);
    float* a = (float*)malloc(sizeof(float) * n_neurons * (size_t)n_sets);
    float* d = (float*)malloc(sizeof(float) * n_neurons * (size_t)n_sets);
    if (!z || !a || !d) { free(z); free(a); free(d); return 1; }
    for (int it = 0; it < iterations; ++it) errors[it] = 0.f;
    const float scale = multiplier / (float)n_sets;
    for (int it = 0; it < iterations; ++it) {
        /* forward of every set with the current parameters */
        size_t woff = 0, noff = 0;
        fan = n_ins;
        for (int l = 0; l < n_layers; ++l) {
            const int no = n_p_l[l];
            for (int j = 0; j < n_sets; ++j) {
                const float* x = l == 0 ? set_ins + (size_t)j * n_ins : a + (noff - (size_t)fan) * n_sets + (size_t)j * fan;
                for (int o = 0; o < no; ++o) {
                    const float* w = params + woff + (size_t)o * fan;
                    float s = 0.f;
                    for (int k = 0; k < fan; ++k) s = fmaf(w[k], x[k], s);
                    s += bias[noff + o];
                    z[noff * n_sets + (size_t)j * no + o] = s;
                    a[noff * n_sets + (size_t)j * no + o] = act_apply(activation, s);
                }
            }
            woff += (size_t)no * fan; noff += (size_t)no; fan = no;
        }
        /* output error and delta of the last layer */
        const size_t last = (noff - (size_t)n_out) * n_sets;
        float err = 0.f;
        for (int j = 0; j < n_sets; ++j)
            for (int o = 0; o < n_out; ++o) {
                const float e = a[last + (size_t)j * n_out + o] - set_outs[(size_t)j * n_out + o];
                err += fabsf(e);
                d[last + (size_t)j * n_out + o] = e * act_deriv(activation, z[last + (size_t)j * n_out + o]);
            }
        errors[it] = err;
        if (err <= error_threshold) break;
        /* deltas of the hidden layers, back to front, with the OLD parameters */
        size_t woff_l = n_params, noff_l = n_neurons;
        for (int l = n_layers - 1; l >= 1; --l) {
            const int no = n_p_l[l], ni = n_p_l[l - 1];
            woff_l -= (size_t)no * ni; noff_l -= (size_t)no;
            const size_t prev = (noff_l - (size_t)ni) * n_sets, cur = noff_l * n_sets;
            for (int j = 0; j < n_sets; ++j)
                for (int i = 0; i < ni; ++i) {
                    float s = 0.f;
                    for (int o = 0; o < no; ++o) s = fmaf(params[woff_l + (size_t)o * ni + i], d[cur + (size_t)j * no + o], s);
                    d[prev + (size_t)j * ni + i] = s * act_deriv(activation, z[prev + (size_t)j * ni + i]);
                }
        }
        /* update: p -= multiplier * mean over the sets of the gradient */
        woff = 0; noff = 0; fan = n_ins;
        for (int l = 0; l < n_layers; ++l) {
            const int no = n_p_l[l];
            for (int o = 0; o < no; ++o) {
                for (int k = 0; k < fan; ++k) {
                    float g = 0.f;
                    for (int j = 0; j < n_sets; ++j) {
                        const float xin = l == 0 ? set_ins[(size_t)j * n_ins + k] : a[(noff - (size_t)fan) * n_sets + (size_t)j * fan + k];
                        g = fmaf(d[noff * n_sets + (size_t)j * no + o], xin, g);
                    }
                    params[woff + (size_t)o * fan + k] -= scale * g;
                }
                float gb = 0.f;
                for (int j = 0; j < n_sets; ++j) gb += d[noff * n_sets + (size_t)j * no + o];
                bias[noff + o] -= scale * gb;
            }
            woff += (size_t)no * fan; noff += (size_t)no; fan = no;
        }
    }
    free(z); free(a); free(d);
    return 0;
}
