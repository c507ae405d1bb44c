/* Drives the multi-device entry points of include/rocoder_hip.h from plain C, the way a host binding (Rust over the
 * C-ABI: INTEGRATION.md) would: a stereo job cut over a device list, against the same job on one engine.
 * usage: multi_driver <n_listed_devices> <window_len> <factor> <pitch> <in_len>   (all entries = device 0 on a one-GPU box)
 * prints "OK ..." and returns 0 when every output sample agrees bit for bit. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rocoder_hip.h"

#define CHECK(x)                                                               \
    do {                                                                       \
        int rc_ = (x);                                                         \
        if (rc_ < 0) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, rc_last_error());       \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 6) return 2;
    const unsigned n_dev = (unsigned)atoi(argv[1]);
    rc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.window_len = (uint32_t)atoi(argv[2]);
    cfg.factor = (float)atof(argv[3]);
    cfg.amplitude = 1.0f;
    cfg.pitch_multiple = atoi(argv[4]);
    cfg.sample_rate = 44100;
    cfg.channels = 2;
    cfg.buffer_secs = 1.0f;
    cfg.seed = 0x5EED;
    const size_t L = (size_t)atol(argv[5]);
    float *x[2], *a[2], *b[2];
    const size_t n_out = rc_offline_output_len(&cfg, L);
    if (!n_out) return 3;
    for (int c = 0; c < 2; ++c) {
        x[c] = (float *)malloc(L * sizeof(float));
        a[c] = (float *)malloc(n_out * sizeof(float));
        b[c] = (float *)malloc(n_out * sizeof(float));
        unsigned s = 12345u + (unsigned)c;
        for (size_t i = 0; i < L; ++i) {
            s = s * 1664525u + 1013904223u;
            x[c][i] = 0.5f * sinf(0.03f * (float)(c + 1) * (float)i) + 0.05f * ((float)(s >> 8) / 8388608.0f - 1.0f);
        }
    }
    rc_engine *e = NULL;
    CHECK(rc_engine_create(&cfg, &e));
    size_t got = 0;
    CHECK(rc_engine_stretch_host(e, (const float *const *)x, L, a, n_out, &got));
    rc_engine_destroy(e);
    int32_t ids[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rc_multi *m = NULL;
    CHECK(rc_multi_create(&cfg, ids, n_dev, &m));
    if (rc_multi_device_count(m) != n_dev) return 4;
    size_t got2 = 0;
    CHECK(rc_multi_stretch_host(m, (const float *const *)x, L, b, n_out, &got2));
    rc_multi_destroy(m);
    if (got != n_out || got2 != n_out) return 5;
    for (int c = 0; c < 2; ++c)
        if (memcmp(a[c], b[c], n_out * sizeof(float)) != 0) {
            size_t i = 0;
            while (a[c][i] == b[c][i]) ++i;
            fprintf(stderr, "channel %d differs first at %zu: %g vs %g\n", c, i, a[c][i], b[c][i]);
            return 6;
        }
    rc_shard plan[24];
    const size_t ns = rc_shard_plan(2, n_out / (cfg.window_len / (cfg.pitch_multiple > 0 ? 1 : 1)), n_dev, plan, 24);
    printf("OK devices=%u shards=%zu samples=%zu\n", n_dev, ns, n_out);
    return 0;
}
