/* dev: rate of the streaming seam measured from C (no interpreter in the loop): a closed mono channel of 600 s,
 * window 16384, factor 8; rc_engine_next_window (copy) and rc_engine_next_window_view (pointer).
 *   gcc -O2 -I include -o /tmp/stream_rate tests/c/stream_rate.c -Lrocoder_amd -lrocoder_hip -Wl,-rpath,$PWD/rocoder_amd -lm */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rocoder_hip.h"

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
int main(void) {
    const size_t L = 44100u * 600u;
    float *x = malloc(L * sizeof(float));
    for (size_t i = 0; i < L; ++i) x[i] = (float)((i * 2654435761u) >> 8 & 0xFFFF) / 65536.0f - 0.5f;
    for (int mode = 0; mode < 4; ++mode) {
        rc_config c;
        memset(&c, 0, sizeof c);
        c.struct_size = sizeof c;
        c.window_len = 16384;
        c.factor = 8.0f;
        c.amplitude = 1.0f;
        c.pitch_multiple = 1;
        c.sample_rate = 44100;
        c.channels = 1;
        c.buffer_secs = 1.0f;
        c.seed = 1;
        rc_engine *e = NULL;
        if (rc_engine_create(&c, &e) != RC_OK) {
            fprintf(stderr, "create: %s\n", rc_last_error());
            return 1;
        }
        const double t0 = now();
        rc_engine_push_input(e, 0, x, L);
        rc_engine_close_input(e, 0);
        const double t1 = now();
        float *out = malloc(16384 * sizeof(float));
        size_t n = 0, total = 0;
        double acc = 0;
        while (rc_engine_is_done(e, 0) != 1) {
            if (mode & 1) {
                const float *w = NULL;
                if (rc_engine_next_window_view(e, 0, &w, &n) != RC_OK) return 2;
                acc += w[0] + w[n - 1];
            } else {
                if (rc_engine_next_window(e, 0, out, 16384, &n) != RC_OK) return 2;
                acc += out[0];
            }
            total += n;
        }
        const double t2 = now();
        printf("%s: %zu samples, push %.1f ms, windows %.1f ms = %.2f Gsamples/s (%.2f with the push) [%g]\n",
               (mode & 1) ? "view" : "copy", total, (t1 - t0) * 1e3, (t2 - t1) * 1e3, total / (t2 - t1) / 1e9,
               total / (t2 - t0) / 1e9, acc);
        free(out);
        rc_engine_destroy(e);
    }
    free(x);
    return 0;
}
