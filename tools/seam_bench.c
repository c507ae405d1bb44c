/* Measurement helper for bench.py (config.streaming_seam, config.live_latency_us): the literal drop-in seam -
 * rc_engine_push_input / rc_engine_close_input / rc_engine_next_window(_view), what INTEGRATION.md tells a maintainer to
 * bind - driven from C the way src/stretcher_processor.rs:56-71 drives Stretcher::next_window (windows outer, channels
 * inner), with no interpreter in the loop. Built by rocoder_amd.build.build() into rocoder_amd/bin/libseam_bench.so; the
 * engine library is resolved at run time from the path the caller gives (the product library or an A/B variant), so this
 * file links against nothing.
 *   standalone: gcc -O2 -DSEAM_BENCH_MAIN -I include -o /tmp/seam_bench tools/seam_bench.c -ldl -lm && /tmp/seam_bench rocoder_amd/librocoder_hip.so */
#define _GNU_SOURCE
#include <ctype.h>
#include <dlfcn.h>
#include <sched.h>
#include <math.h>
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

typedef struct api {
    int (*create)(const rc_config *, rc_engine **);
    void (*destroy)(rc_engine *);
    int (*get_params)(const rc_engine *, rc_params *);
    int (*push)(rc_engine *, uint32_t, const float *, size_t);
    int (*close)(rc_engine *, uint32_t);
    int (*next)(rc_engine *, uint32_t, float *, size_t, size_t *);
    int (*view)(rc_engine *, uint32_t, const float **, size_t *);
    int (*is_done)(const rc_engine *, uint32_t);
    const char *(*last_error)(void);
} api;

static int load(const char *path, api *a) {
    void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return -1;
#define SYM(field, name)                      \
    *(void **)(&a->field) = dlsym(h, name);   \
    if (!a->field) return -1
    SYM(create, "rc_engine_create");
    SYM(destroy, "rc_engine_destroy");
    SYM(get_params, "rc_engine_get_params");
    SYM(push, "rc_engine_push_input");
    SYM(close, "rc_engine_close_input");
    SYM(next, "rc_engine_next_window");
    SYM(view, "rc_engine_next_window_view");
    SYM(is_done, "rc_engine_is_done");
    SYM(last_error, "rc_last_error");
#undef SYM
    return 0;
}

/* What `numactl --cpunodebind=<the GPU's node>` does for a host program: the calling thread runs on the CPUs next to
 * device 0 (sysfs local_cpulist of its PCI address, which the HIP runtime the engine library brought in reports). Returns
 * the NUMA node, or -1 when anything is missing (the thread then stays where the scheduler put it). */
int seam_bench_pin_near_gpu(void) {
    int (*bus_id)(char *, int, int) = NULL;
    *(void **)(&bus_id) = dlsym(RTLD_DEFAULT, "hipDeviceGetPCIBusId");
    if (!bus_id) {  /* (a host that loaded the engine library RTLD_LOCAL, e.g. through ctypes: ask for the runtime by name) */
        const char *names[] = {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", NULL};
        for (int i = 0; names[i] && !bus_id; ++i) {
            void *h = dlopen(names[i], RTLD_LAZY | RTLD_NOLOAD);
            if (h) *(void **)(&bus_id) = dlsym(h, "hipDeviceGetPCIBusId");
        }
    }
    char bdf[64] = {0}, path[160], buf[4096];
    if (!bus_id || bus_id(bdf, (int)sizeof bdf - 1, 0) != 0) return -1;
    for (char *q = bdf; *q; ++q) *q = (char)tolower((unsigned char)*q);
    int node = -1;
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (f) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    f = fopen(path, "r");
    if (!f) return -1;
    if (!fgets(buf, sizeof buf, f)) buf[0] = 0;
    fclose(f);
    cpu_set_t set;
    CPU_ZERO(&set);
    int any = 0;
    for (char *q = buf; *q;) {  /* "0-63,128-191" */
        if (!isdigit((unsigned char)*q)) {
            ++q;
            continue;
        }
        long a = strtol(q, &q, 10), b = a;
        if (*q == '-') b = strtol(q + 1, &q, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) {
            CPU_SET((int)c, &set);
            any = 1;
        }
    }
    if (!any || sched_setaffinity(0, sizeof set, &set) != 0) return -1;
    return node;
}

static rc_config config(uint32_t window, float factor, uint32_t channels, float buffer_secs) {
    rc_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.window_len = window;
    c.factor = factor;
    c.amplitude = 1.0f;
    c.pitch_multiple = 1;
    c.sample_rate = 44100;
    c.channels = channels;
    c.buffer_secs = buffer_secs;
    c.seed = 0x5EED;
    return c;
}

static float *synth(uint32_t channels, size_t L) {  /* BASELINE.md's shape: a sine per channel + a little noise */
    float *x = malloc((size_t)channels * L * sizeof(float));
    if (!x) return NULL;
    for (uint32_t c = 0; c < channels; ++c) {
        uint32_t s = 0xC0DEC0DEu + c;
        const double w = 2.0 * 3.14159265358979323846 * 220.0 * (c + 1) / 44100.0;
        for (size_t i = 0; i < L; ++i) {
            s = s * 1664525u + 1013904223u;
            x[(size_t)c * L + i] = 0.5f * (float)sin(w * (double)(i % 44100)) + 0.05f * ((float)(s >> 8) / 8388608.0f - 1.0f);
        }
    }
    return x;
}

/* A closed job (`-o` mode: every channel sent whole, then the sender dropped - src/main.rs:148), handed out in the
 * processor's order. view != 0: rc_engine_next_window_view (a pointer into the engine's pinned block), else the copying
 * rc_engine_next_window into one caller buffer. Returns 0 and the output samples per second of the hand-out loop (the
 * push of the input is timed separately: push_ms). */
int seam_bench_closed(const char *libpath, uint32_t window, float factor, uint32_t channels, size_t L, int view,
                      double *samples_per_s, double *push_ms, uint64_t *samples_out) {
    api a;
    if (load(libpath, &a)) return -100;
    rc_config c = config(window, factor, channels, 1.0f);
    rc_engine *e = NULL;
    int rc = a.create(&c, &e);
    if (rc != RC_OK) return rc;
    rc_params P;
    a.get_params(e, &P);
    float *x = synth(channels, L);
    float *out = malloc((size_t)P.window_out_len * sizeof(float));
    if (!x || !out) return RC_ENOMEM;
    const double t0 = now();
    for (uint32_t ch = 0; ch < channels; ++ch) {
        if ((rc = a.push(e, ch, x + (size_t)ch * L, L)) != RC_OK) goto done;
        if ((rc = a.close(e, ch)) != RC_OK) goto done;
    }
    const double t1 = now();
    uint64_t total = 0;
    double acc = 0;
    const int trace = getenv("SEAM_BENCH_TRACE") != NULL;  /* dev: every hand-out that took longer than 50 us */
    uint64_t call_i = 0;
    for (;;) {  /* src/stretcher_processor.rs:63-70: stop at the first channel that is done */
        int stop = 0;
        for (uint32_t ch = 0; ch < channels; ++ch) {
            if (a.is_done(e, ch) == 1) {
                stop = 1;
                break;
            }
            size_t n = 0;
            const double tc0 = trace ? now() : 0;
            if (view) {
                const float *w = NULL;
                if ((rc = a.view(e, ch, &w, &n)) != RC_OK) goto done;
                acc += w[0] + w[n - 1];
            } else {
                if ((rc = a.next(e, ch, out, P.window_out_len, &n)) != RC_OK) goto done;
                acc += out[0];
            }
            if (trace) {
                const double tc1 = now();
                if (tc1 - tc0 > 50e-6)
                    fprintf(stderr, "call %llu ch %u at %.2f ms: %.3f ms\n", (unsigned long long)call_i, ch, (tc0 - t1) * 1e3, (tc1 - tc0) * 1e3);
            }
            call_i++;
            total += n;
        }
        if (stop) break;
    }
    {
        const double t2 = now();
        *samples_per_s = (double)total / (t2 - t1);
        *push_ms = (t1 - t0) * 1e3;
        *samples_out = total;
        if (acc != acc) rc = -101;  /* (keeps the reads alive; a NaN in the output would also be news) */
    }
done:
    free(out);
    free(x);
    a.destroy(e);
    return rc;
}

static int cmp_double(const void *p, const void *q) {
    const double a = *(const double *)p, b = *(const double *)q;
    return (a > b) - (a < b);
}

/* Open channels (live mode: the one the hot-swap exists for, src/stretcher.rs:82-85, src/hotswapper.rs:12): per
 * iteration every channel is pushed the input one output window consumes (hops_per_window x step samples) and then
 * asked for its next window, as the processor would. Latency = from the first push of the iteration until the FIRST
 * rc_engine_next_window returns (out[0..3]: median, p99, mean, max in microseconds); round = until every channel has
 * its window (out[4..5]: median, p99). */
int seam_bench_live(const char *libpath, uint32_t window, float factor, uint32_t channels, float buffer_secs,
                    uint32_t n_windows, double *out6) {
    api a;
    if (load(libpath, &a)) return -100;
    rc_config c = config(window, factor, channels, buffer_secs);
    rc_engine *e = NULL;
    int rc = a.create(&c, &e);
    if (rc != RC_OK) return rc;
    rc_params P;
    a.get_params(e, &P);
    const size_t adv = (size_t)P.hops_per_window * P.sample_step_len;
    const size_t L = (size_t)window + adv * ((size_t)n_windows + 8);
    float *x = synth(channels, L);
    float *w = malloc((size_t)P.window_out_len * sizeof(float));
    double *first = malloc(sizeof(double) * n_windows), *round = malloc(sizeof(double) * n_windows);
    if (!x || !w || !first || !round) return RC_ENOMEM;
    size_t pos = 0, n = 0;
    /* prime: a window's worth of history, and a few untimed rounds (first-touch allocations, the first launches) */
    for (uint32_t ch = 0; ch < channels; ++ch)
        if ((rc = a.push(e, ch, x + (size_t)ch * L, window - P.sample_step_len)) != RC_OK) goto done;
    pos = window - P.sample_step_len;  /* window w needs w * adv + step + N samples: one more push of adv completes it */
    for (uint32_t it = 0; it < n_windows + 8; ++it) {
        const double t0 = now();
        double t_first = 0;
        for (uint32_t ch = 0; ch < channels; ++ch)
            if ((rc = a.push(e, ch, x + (size_t)ch * L + pos, adv)) != RC_OK) goto done;
        for (uint32_t ch = 0; ch < channels; ++ch) {
            if ((rc = a.next(e, ch, w, P.window_out_len, &n)) != RC_OK) goto done;
            if (ch == 0) t_first = now();
        }
        const double t1 = now();
        pos += adv;
        if (it >= 8) {
            first[it - 8] = (t_first - t0) * 1e6;
            round[it - 8] = (t1 - t0) * 1e6;
        }
    }
    {
        double mean = 0;
        for (uint32_t i = 0; i < n_windows; ++i) mean += first[i];
        qsort(first, n_windows, sizeof(double), cmp_double);
        qsort(round, n_windows, sizeof(double), cmp_double);
        out6[0] = first[n_windows / 2];
        out6[1] = first[(size_t)((double)n_windows * 0.99)];
        out6[2] = mean / n_windows;
        out6[3] = first[n_windows - 1];
        out6[4] = round[n_windows / 2];
        out6[5] = round[(size_t)((double)n_windows * 0.99)];
    }
done:
    free(first);
    free(round);
    free(w);
    free(x);
    a.destroy(e);
    return rc;
}

#ifdef SEAM_BENCH_MAIN
int main(int argc, char **argv) {
    const char *lib = argc > 1 ? argv[1] : "rocoder_amd/librocoder_hip.so";
    const uint32_t channels = argc > 2 ? (uint32_t)atoi(argv[2]) : 2;
    {
        api a;
        if (load(lib, &a)) return 1;
        if (!(argc > 3 && !strcmp(argv[3], "nopin"))) printf("thread on the CPUs of the GPU's NUMA node %d\n", seam_bench_pin_near_gpu());
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int view = 0; view < 2; ++view) {
            double sps = 0, push = 0;
            uint64_t n = 0;
            const int rc = seam_bench_closed(lib, 16384, 8.0f, channels, 26460000u, view, &sps, &push, &n);
            printf("%s: rc %d, %llu samples, push %.1f ms, %.2f Gsamples/s\n", view ? "view" : "copy", rc,
                   (unsigned long long)n, push, sps / 1e9);
        }
    for (int b = 0; b < 2; ++b) {
        double o[6] = {0};
        const int rc = seam_bench_live(lib, 16384, 8.0f, 2, b ? 0.1f : 1.0f, 1000, o);
        printf("live buffer %.1f s: rc %d first window median %.1f us p99 %.1f mean %.1f max %.1f; round median %.1f p99 %.1f\n",
               b ? 0.1 : 1.0, rc, o[0], o[1], o[2], o[3], o[4], o[5]);
    }
    return 0;
}
#endif
