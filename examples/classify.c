/* classify.c — the C ABI from plain C: create a ViT-B/16 context, load (or seed) weights, run one batch.
 *
 *   gcc -std=c99 -O2 -I include examples/classify.c -L vit-fpga_amd -lvithip -Wl,-rpath,$PWD/vit-fpga_amd -o classify
 *   ./classify [weights.vhblob | -] [batch]
 *
 * Without a file the weights are the seeded synthetic ones (seed 0); the input is the seeded synthetic batch the
 * benchmark uses.  Prints the arg-max class and logit of every image and the device time of the forward. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vithip.h"

#define CHECK(call, ctx)                                                        \
    do {                                                                        \
        if ((call) != VH_OK) {                                                  \
            fprintf(stderr, "%s: %s\n", #call, vh_last_error(ctx));             \
            return 1;                                                           \
        }                                                                       \
    } while (0)

int main(int argc, char** argv) {
    const char* path = (argc > 1 && strcmp(argv[1], "-") != 0) ? argv[1] : NULL;
    const int batch = argc > 2 ? atoi(argv[2]) : 8;
    vh_config cfg = {224, 16, 3, 768, 12, 3072, 12, 1000, VH_DTYPE_BF16, 0, 1e-6f, 0};
    if (path) CHECK(vh_blob_file_config(path, &cfg), NULL);   /* model shape from the file's header */
    cfg.max_batch = batch;
    vh_ctx* ctx = NULL;
    CHECK(vh_create(&cfg, 0, &ctx), NULL);
    if (path) CHECK(vh_load_weights_file(ctx, path), ctx);
    else CHECK(vh_init_weights_seeded(ctx, 0), ctx);

    const size_t in_floats = (size_t)batch * cfg.image_size * cfg.image_size * cfg.channels;
    float *d_in = NULL, *d_out = NULL;
    CHECK(vh_malloc(0, in_floats * sizeof(float), (void**)&d_in), ctx);
    CHECK(vh_malloc(0, (size_t)batch * cfg.classes * sizeof(float), (void**)&d_out), ctx);
    CHECK(vh_fill_input_seeded(ctx, 1, batch, d_in), ctx);
    CHECK(vh_forward_device(ctx, d_in, batch, d_out), ctx);

    float* logits = (float*)malloc((size_t)batch * cfg.classes * sizeof(float));
    CHECK(vh_memcpy_d2h(0, logits, d_out, (size_t)batch * cfg.classes * sizeof(float)), ctx);
    for (int b = 0; b < batch; ++b) {
        int best = 0;
        for (int c = 1; c < cfg.classes; ++c)
            if (logits[(size_t)b * cfg.classes + c] > logits[(size_t)b * cfg.classes + best]) best = c;
        printf("image %d: class %d (logit %.4f)\n", b, best, logits[(size_t)b * cfg.classes + best]);
    }
    double ms = 0.0;
    CHECK(vh_last_kernel_ms(ctx, &ms), ctx);
    printf("forward of %d images: %.3f ms on the device\n", batch, ms);
    free(logits);
    vh_free(0, d_in);
    vh_free(0, d_out);
    vh_destroy(ctx);
    return 0;
}
