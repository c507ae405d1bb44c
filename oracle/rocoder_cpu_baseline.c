/*
 * rocoder_cpu_baseline.c — TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests/).
 *
 * The CPU path of rocoder's stretch (src/stretcher.rs:87-121 over src/fft.rs:42-74) written for
 * SPEED, as SURVEY.md §8 d5 asks for the reported baseline: the same per-hop work as the reference —
 * a full N-point complex FFT (src/fft.rs:59), scalar libm hypotf / sincosf per bin (src/fft.rs:65-68;
 * num-complex's norm() and exp()), a full N-point inverse FFT (src/fft.rs:69), /N * window
 * (src/fft.rs:70-73), the overlap-add in the reference's operation order (src/stretcher.rs:96-103) —
 * but with an optimised FFT (radix-4 Stockham autosort, precomputed twiddles, contiguous inner loops the
 * compiler vectorises; rustfft 6.3.0 is an AVX mixed-radix FFT) in place of the oracle's naive radix-2,
 * and optionally all cores: one OpenMP task per (channel, hop range), the hop before a range recomputed
 * (phases are a pure function of (seed, channel, hop, bin), exactly as the GPU shards the job).
 * One thread for all channels is the reference's own threading (src/stretcher_processor.rs:55-71).
 *
 * It is NOT the oracle: tests/test_oracle.py checks it against oracle/rocoder_oracle.c, never the other
 * way round, and the product never links or loads it. Pitch multiples >= 1, default hanning window,
 * power-of-two window lengths >= 4.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    float re, im;
} cpx;

/* ---- phase source: the frozen spec of oracle/rocoder_oracle.c (rco_phase_*) ---------------- */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint32_t phash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}

/* ---- radix-4 Stockham autosort FFT (decimation in frequency), n a power of two ------------- */
typedef struct {
    size_t n;
    cpx *w;    /* w[k] = exp(-2 pi i k / n), k < n */
    cpx *work; /* n */
} plan_t;

/* one radix-4 pass: n = current sub-length, s = stride (number of interleaved sequences) */
static void pass4(size_t n, size_t s, const cpx *restrict x, cpx *restrict y, const cpx *restrict w,
                  size_t wstride, int inverse) {
    const size_t n1 = n / 4;
    for (size_t p = 0; p < n1; p++) {
        cpx w1 = w[p * wstride], w2 = w[2 * p * wstride], w3 = w[3 * p * wstride];
        if (inverse) {
            w1.im = -w1.im;
            w2.im = -w2.im;
            w3.im = -w3.im;
        }
        const cpx *xa = x + s * p, *xb = x + s * (p + n1), *xc = x + s * (p + 2 * n1), *xd = x + s * (p + 3 * n1);
        cpx *y0 = y + s * (4 * p), *y1 = y0 + s, *y2 = y1 + s, *y3 = y2 + s;
        for (size_t q = 0; q < s; q++) {
            const float ar = xa[q].re, ai = xa[q].im, br = xb[q].re, bi = xb[q].im;
            const float cr = xc[q].re, ci = xc[q].im, dr = xd[q].re, di = xd[q].im;
            const float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
            const float bpdr = br + dr, bpdi = bi + di, bmdr = br - dr, bmdi = bi - di;
            /* j (b - d) with j = -i forward, +i inverse */
            const float jr = inverse ? -bmdi : bmdi, ji = inverse ? bmdr : -bmdr;
            y0[q].re = apcr + bpdr;
            y0[q].im = apci + bpdi;
            const float t1r = amcr + jr, t1i = amci + ji;
            const float t2r = apcr - bpdr, t2i = apci - bpdi;
            const float t3r = amcr - jr, t3i = amci - ji;
            y1[q].re = t1r * w1.re - t1i * w1.im;
            y1[q].im = t1r * w1.im + t1i * w1.re;
            y2[q].re = t2r * w2.re - t2i * w2.im;
            y2[q].im = t2r * w2.im + t2i * w2.re;
            y3[q].re = t3r * w3.re - t3i * w3.im;
            y3[q].im = t3r * w3.im + t3i * w3.re;
        }
    }
}
static void pass2(size_t n, size_t s, const cpx *restrict x, cpx *restrict y, const cpx *restrict w,
                  size_t wstride, int inverse) {
    const size_t n1 = n / 2;
    for (size_t p = 0; p < n1; p++) {
        cpx w1 = w[p * wstride];
        if (inverse) w1.im = -w1.im;
        const cpx *xa = x + s * p, *xb = x + s * (p + n1);
        cpx *y0 = y + s * (2 * p), *y1 = y0 + s;
        for (size_t q = 0; q < s; q++) {
            const float ar = xa[q].re, ai = xa[q].im, br = xb[q].re, bi = xb[q].im;
            y0[q].re = ar + br;
            y0[q].im = ai + bi;
            const float tr = ar - br, ti = ai - bi;
            y1[q].re = tr * w1.re - ti * w1.im;
            y1[q].im = tr * w1.im + ti * w1.re;
        }
    }
}
/* in place on buf (result in buf), unnormalised both ways */
static void fft_run(const plan_t *pl, cpx *buf, int inverse) {
    cpx *x = buf, *y = pl->work;
    size_t n = pl->n, s = 1;
    while (n > 1) {
        if (n % 4 == 0) {
            pass4(n, s, x, y, pl->w, s, inverse);
            n /= 4;
            s *= 4;
        } else {
            pass2(n, s, x, y, pl->w, s, inverse);
            n /= 2;
            s *= 2;
        }
        cpx *t = x;
        x = y;
        y = t;
    }
    if (x != buf) memcpy(buf, x, sizeof(cpx) * pl->n);
}

typedef struct {
    uint32_t N, H, step, pitch;
    float amp;
    uint64_t seed_mixed;
    const float *window, *env;
    const cpx *w;
} job_t;

/* y_k = resynth(x[k step .. k step + N)) for channel c (src/fft.rs:42-74) into y[N] */
static void one_hop(const job_t *j, const plan_t *pl, const float *x, size_t L, uint32_t c, int64_t k,
                    cpx *buf, float *y) {
    const uint32_t N = j->N;
    const size_t off = (size_t)k * j->step;
    for (uint32_t n = 0; n < N; n++) {
        const size_t i = off + n;
        buf[n].re = (i < L ? x[i] : 0.0f) * j->window[n]; /* src/stretcher.rs:129-132, src/fft.rs:51-55 */
        buf[n].im = 0.0f;
    }
    fft_run(pl, buf, 0);
    const uint64_t key = mix64(j->seed_mixed ^ (((uint64_t)c << 40) | ((uint64_t)k & 0xFFFFFFFFFFull)));
    const uint32_t k0 = (uint32_t)key, mul = (uint32_t)(key >> 32) | 1u, M = N / 2;
    const float pi = 3.14159274101257324219f;
    for (uint32_t b = 0; b < M; b++) { /* bins b and b + M share one hash (frozen spec) */
        const uint32_t h = phash(b * mul + k0);
        const float t0 = (float)(h >> 9) * (1.0f / 8388608.0f) * pi;
        const float t1 = (float)(h & 0xFFFFu) * (1.0f / 65536.0f) * pi;
        float s0, c0, s1, c1;
        sincosf(t0, &s0, &c0);
        sincosf(t1, &s1, &c1);
        const float m0 = hypotf(buf[b].re, buf[b].im), m1 = hypotf(buf[b + M].re, buf[b + M].im);
        buf[b].re = c0 * m0;
        buf[b].im = s0 * m0;
        buf[b + M].re = c1 * m1;
        buf[b + M].im = s1 * m1;
    }
    fft_run(pl, buf, 1);
    const float fn = (float)N;
    for (uint32_t n = 0; n < N; n++) y[n] = buf[n].re / fn * j->window[n]; /* src/fft.rs:70-73 */
}

/* hops [k0, k1) of channel c -> F[(k H + i) / p] for every k H + i divisible by p */
static void hop_range(const job_t *j, const plan_t *pl, const float *x, size_t L, uint32_t c, int64_t k0,
                      int64_t k1, float *out, cpx *buf, float *y, float *tail) {
    const uint32_t H = j->H, p = j->pitch;
    if (k0 > 0) {
        one_hop(j, pl, x, L, c, k0 - 1, buf, y);
        memcpy(tail, y + H, sizeof(float) * H);
    } else {
        memset(tail, 0, sizeof(float) * H); /* src/stretcher.rs:58-59 */
    }
    for (int64_t k = k0; k < k1; k++) {
        one_hop(j, pl, x, L, c, k, buf, y);
        const int64_t g0 = k * (int64_t)H;
        if (p == 1) {
            float *o = out + g0;
            for (uint32_t i = 0; i < H; i++) o[i] = (y[i] + tail[i]) * j->env[i] * j->amp; /* stretcher.rs:97-100 */
        } else {
            for (uint32_t i = 0; i < H; i++) {
                const int64_t g = g0 + i;
                if (g % p == 0) out[g / p] = (y[i] + tail[i]) * j->env[i] * j->amp; /* resampler.rs:15-18 */
            }
        }
        memcpy(tail, y + H, sizeof(float) * H);
    }
}

/* Output length per channel, 0 for unsupported parameters. */
size_t rcb_output_len(size_t L, uint32_t N, float factor, int pitch) {
    if (pitch < 1 || N < 4 || (N & (N - 1))) return 0;
    const float psf = factor * (float)pitch;
    const float stepf = (float)N / (psf * 2.0f);
    if (!(stepf >= 1.0f)) return 0;
    const uint32_t step = (uint32_t)stepf;
    const uint64_t kd = L >= N ? (uint64_t)(L - N) / step + 1 : 0;
    const uint64_t hpw = 2ull * (uint64_t)pitch;
    const uint64_t windows = kd / hpw + 1;
    return (size_t)(windows * N);
}

/* in: [C][L] (stride L), out: [C][rcb_output_len] ; threads <= 1: one DSP thread for all channels.
 * Returns 0, or -1 for unsupported parameters. */
int rcb_stretch_from(const float *in, size_t L, uint32_t C, uint32_t ch_first, uint32_t N, float factor,
                     float amplitude, int pitch, uint64_t seed, float *out, int threads);
int rcb_stretch(const float *in, size_t L, uint32_t C, uint32_t N, float factor, float amplitude, int pitch,
                uint64_t seed, float *out, int threads) {
    return rcb_stretch_from(in, L, C, 0, N, factor, amplitude, pitch, seed, out, threads);
}
/* The same for rows that are channels ch_first .. ch_first + C - 1 of a larger job (the phase key takes the
 * job's channel index): lets a test check one channel of a BASELINE-size job at a time. */
int rcb_stretch_from(const float *in, size_t L, uint32_t C, uint32_t ch_first, uint32_t N, float factor,
                     float amplitude, int pitch, uint64_t seed, float *out, int threads) {
    const size_t n_out = rcb_output_len(L, N, factor, pitch);
    if (!n_out) return -1;
    const uint32_t H = N / 2;
    const float psf = factor * (float)pitch;
    job_t j;
    j.N = N;
    j.H = H;
    j.step = (uint32_t)((float)N / (psf * 2.0f));
    j.pitch = (uint32_t)pitch;
    j.amp = fmaxf(4.0f, psf / 4.0f) * amplitude; /* src/stretcher.rs:52 */
    j.seed_mixed = mix64(seed);
    float *window = (float *)malloc(sizeof(float) * N), *env = (float *)malloc(sizeof(float) * H);
    cpx *w = (cpx *)malloc(sizeof(cpx) * N);
    const float two_pi = 3.14159274101257324219f * 2.0f;
    for (uint32_t i = 0; i < N; i++) window[i] = 0.5f - (cosf(((float)i * two_pi) / (float)(N - 1)) * 0.5f);
    const float hh = (1.0f + sqrtf(sqrtf(0.5f))) * 0.5f;
    for (uint32_t i = 0; i < H; i++) env[i] = 0.5f - ((1.0f - hh) * cosf(((float)i * two_pi) / (float)(H - 1)));
    for (uint32_t k = 0; k < N; k++) {
        const double a = -2.0 * M_PI * (double)k / (double)N;
        w[k].re = (float)cos(a);
        w[k].im = (float)sin(a);
    }
    j.window = window;
    j.env = env;
    j.w = w;
    const int64_t K = (int64_t)(n_out * (size_t)pitch / H); /* hops per channel */
    int nt = threads > 1 ? threads : 1;
    /* segments: channel-major, about 4 per thread so the tail of the job balances */
    int64_t seg_per_ch = nt > 1 ? ((int64_t)nt * 4 + C - 1) / C : 1;
    if (seg_per_ch > K) seg_per_ch = K > 0 ? K : 1;
    const int64_t seg_len = (K + seg_per_ch - 1) / seg_per_ch;
    const int64_t n_seg = (int64_t)C * seg_per_ch;
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
    {
        plan_t pl;
        pl.n = N;
        pl.w = w;
        pl.work = (cpx *)malloc(sizeof(cpx) * N);
        cpx *buf = (cpx *)malloc(sizeof(cpx) * N);
        float *y = (float *)malloc(sizeof(float) * N), *tail = (float *)malloc(sizeof(float) * H);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t sgi = 0; sgi < n_seg; sgi++) {
            const uint32_t c = (uint32_t)(sgi / seg_per_ch);
            const int64_t k0 = (sgi % seg_per_ch) * seg_len;
            int64_t k1 = k0 + seg_len;
            if (k1 > K) k1 = K;
            if (k0 < k1) hop_range(&j, &pl, in + (size_t)c * L, L, ch_first + c, k0, k1, out + (size_t)c * n_out, buf, y, tail);
        }
        free(pl.work);
        free(buf);
        free(y);
        free(tail);
    }
    free(window);
    free(env);
    free(w);
    return 0;
}
