/*
 * rocoder_oracle.c — TEST INFRASTRUCTURE ONLY (see rocoder_oracle.h for the rules and the
 * "parity unpinned" statement). Plain-C restatement of the reference hot path; every
 * function cites the reference file:line (under /root/reference) it follows.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (no FMA contraction: rustc does not
 * contract f32 expressions either).
 */
#include "rocoder_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define RCO_PI_F32 3.14159274101257324219f /* f32::consts::PI */

/* ------------------------------------------------------------------------- */
/* src/windows.rs                                                             */
/* ------------------------------------------------------------------------- */

/* windows.rs:4-9: 0.5 - (cos((i as f32 * two_pi) / (len - 1) as f32) * 0.5), all f32. */
void rco_hanning(size_t len, float *out) {
    const float two_pi = RCO_PI_F32 * 2.0f;
    for (size_t i = 0; i < len; i++) {
        float arg = ((float)i * two_pi) / (float)(len - 1);
        out[i] = 0.5f - (cosf(arg) * 0.5f);
    }
}

/* windows.rs:12-14 */
void rco_rectangular(size_t len, float *out) {
    for (size_t i = 0; i < len; i++) out[i] = 1.0f;
}

/* windows.rs:19-21 */
void rco_inverse(const float *in, size_t len, float *out) {
    for (size_t i = 0; i < len; i++) out[i] = 1.0f / in[i];
}

/* ------------------------------------------------------------------------- */
/* src/crossfade.rs:4-10                                                      */
/* ------------------------------------------------------------------------- */
void rco_hanning_crossfade_compensation(size_t len, float *out) {
    const float two_pi = RCO_PI_F32 * 2.0f;
    const float hinv_sqrt2 = (1.0f + sqrtf(sqrtf(0.5f))) * 0.5f;
    for (size_t i = 0; i < len; i++) {
        float arg = ((float)i * two_pi) / (float)(len - 1);
        out[i] = 0.5f - ((1.0f - hinv_sqrt2) * cosf(arg));
    }
}

/* ------------------------------------------------------------------------- */
/* src/math.rs:28-30, src/resampler.rs:3-35                                   */
/* ------------------------------------------------------------------------- */
float rco_lerp(float start, float end, float ratio) { return start + (end - start) * ratio; }

size_t rco_resample_len(size_t n, int factor) {
    if (factor == 1) return n;
    if (factor > 1) return (n + (size_t)factor - 1) / (size_t)factor; /* step_by */
    if (factor < -1) return n == 0 ? 0 : (n - 1) * (size_t)(-factor);
    return (size_t)-1; /* 0 and -1 panic in the reference (resampler.rs:11) */
}

size_t rco_resample(const float *in, size_t n, int factor, float *out) {
    if (factor == 1) { /* resampler.rs:4-5 */
        memcpy(out, in, n * sizeof(float));
        return n;
    } else if (factor > 1) { /* resample_faster, resampler.rs:15-18 */
        size_t m = 0;
        for (size_t i = 0; i < n; i += (size_t)factor) out[m++] = in[i];
        return m;
    } else if (factor < -1) { /* resample_slower, resampler.rs:20-35 */
        size_t f = (size_t)(-factor), m = 0;
        if (n == 0) return 0;
        for (size_t i = 0; i + 1 < n; i++) {
            for (size_t j = 0; j < f; j++)
                out[m++] = rco_lerp(in[i], in[i + 1], (float)j / (float)f);
        }
        return m;
    }
    return (size_t)-1;
}

/* ------------------------------------------------------------------------- */
/* Phase source. The reference draws theta = thread_rng().gen_range(0.0..PI)   */
/* per bin per hop (fft.rs:13,64-67); rand 0.8.5 UniformFloat::sample_single   */
/* makes that (u32 >> 9) * 2^-23 * (high - low) + low. Here the u32 comes from */
/* a counter-based hash of (seed, channel, hop, bin) so CPU and GPU agree:     */
/*   key  = mix64(mix64(seed) ^ ((channel << 40) | hop))   (splitmix64 steps)  */
/*   h(c) = lowbias32(c * (hi32(key) | 1) + lo32(key))                         */
/* One hash serves two bins, b and b + N/2 (half the hashing on the GPU):      */
/*   b <  N/2: theta = fl32((h(b) >> 9) * 2^-23) * PI_f32   (rand's 23 bits)   */
/*   b >= N/2: theta = fl32((h(b - N/2) & 0xFFFF) * 2^-16) * PI_f32            */
/* ------------------------------------------------------------------------- */
static uint64_t rco_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t rco_phase_key(uint64_t seed, uint32_t channel, uint64_t hop) {
    uint64_t ctr = ((uint64_t)channel << 40) | (hop & 0xFFFFFFFFFFull);
    return rco_mix64(rco_mix64(seed) ^ ctr);
}

uint32_t rco_phase_hash(uint64_t key, uint32_t bin) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    uint32_t x = bin * (k1 | 1u) + k0;
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}

float rco_phase_theta(uint64_t key, uint32_t bin, uint32_t n_bins) {
    uint32_t half = n_bins / 2;
    float u;
    if (bin < half) {
        uint32_t h = rco_phase_hash(key, bin);
        u = (float)(h >> 9) * (1.0f / 8388608.0f); /* exact: 23-bit mantissa */
    } else {
        uint32_t h = rco_phase_hash(key, bin - half);
        u = (float)(h & 0xFFFFu) * (1.0f / 65536.0f); /* exact */
    }
    return u * RCO_PI_F32; /* * (high - low) + low, low = 0 */
}

/* ------------------------------------------------------------------------- */
/* Complex FFT, unnormalised both ways, forward = e^{-i...} (the rustfft       */
/* conventions relied on at fft.rs:59,69,72). Twiddles computed in f64 and     */
/* rounded to f32 as rustfft does; butterflies in f32.                         */
/* ------------------------------------------------------------------------- */
typedef struct {
    size_t n;
    int pow2;
    float *tw;     /* pow2: n/2 forward twiddles (re,im) */
    uint32_t *rev; /* pow2: bit reversal */
    double *dtw;   /* !pow2: n roots of unity (re,im) in f64 */
    float *scratch;
} rco_fft_plan;

static int rco_is_pow2(size_t n) { return n && !(n & (n - 1)); }

static void rco_fft_plan_init(rco_fft_plan *p, size_t n) {
    memset(p, 0, sizeof *p);
    p->n = n;
    p->pow2 = rco_is_pow2(n);
    if (p->pow2) {
        p->tw = (float *)malloc(sizeof(float) * (n ? n : 1));
        p->rev = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
        for (size_t k = 0; k < n / 2; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)n;
            p->tw[2 * k] = (float)cos(ang);
            p->tw[2 * k + 1] = (float)sin(ang);
        }
        unsigned bits = 0;
        while (((size_t)1 << bits) < n) bits++;
        for (size_t i = 0; i < n; i++) {
            uint32_t r = 0;
            for (unsigned b = 0; b < bits; b++)
                if (i & ((size_t)1 << b)) r |= 1u << (bits - 1 - b);
            p->rev[i] = r;
        }
    } else {
        p->dtw = (double *)malloc(sizeof(double) * 2 * (n ? n : 1));
        for (size_t k = 0; k < n; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)n;
            p->dtw[2 * k] = cos(ang);
            p->dtw[2 * k + 1] = sin(ang);
        }
        p->scratch = (float *)malloc(sizeof(float) * 2 * (n ? n : 1));
    }
}

static void rco_fft_plan_free(rco_fft_plan *p) {
    free(p->tw);
    free(p->rev);
    free(p->dtw);
    free(p->scratch);
}

/* in-place on interleaved (re,im); inverse != 0 conjugates the twiddles. */
static void rco_fft_process(const rco_fft_plan *p, float *buf, int inverse) {
    const size_t n = p->n;
    if (n <= 1) return;
    if (p->pow2) {
        for (size_t i = 0; i < n; i++) {
            size_t r = p->rev[i];
            if (r > i) {
                float tr = buf[2 * i], ti = buf[2 * i + 1];
                buf[2 * i] = buf[2 * r];
                buf[2 * i + 1] = buf[2 * r + 1];
                buf[2 * r] = tr;
                buf[2 * r + 1] = ti;
            }
        }
        for (size_t len = 2; len <= n; len <<= 1) {
            size_t half = len >> 1, tstep = n / len;
            for (size_t base = 0; base < n; base += len) {
                for (size_t k = 0; k < half; k++) {
                    float wr = p->tw[2 * k * tstep];
                    float wi = p->tw[2 * k * tstep + 1];
                    if (inverse) wi = -wi;
                    size_t a = base + k, b = a + half;
                    float br = buf[2 * b], bi = buf[2 * b + 1];
                    float tr = br * wr - bi * wi;
                    float ti = br * wi + bi * wr;
                    float ar = buf[2 * a], ai = buf[2 * a + 1];
                    buf[2 * a] = ar + tr;
                    buf[2 * a + 1] = ai + ti;
                    buf[2 * b] = ar - tr;
                    buf[2 * b + 1] = ai - ti;
                }
            }
        }
    } else {
        /* direct O(n^2) DFT with f64 accumulation (only for small non-power-of-two n) */
        float *out = p->scratch;
        for (size_t k = 0; k < n; k++) {
            double sr = 0.0, si = 0.0;
            for (size_t t = 0; t < n; t++) {
                size_t idx = (k * t) % n;
                double wr = p->dtw[2 * idx], wi = p->dtw[2 * idx + 1];
                if (inverse) wi = -wi;
                double xr = buf[2 * t], xi = buf[2 * t + 1];
                sr += xr * wr - xi * wi;
                si += xr * wi + xi * wr;
            }
            out[2 * k] = (float)sr;
            out[2 * k + 1] = (float)si;
        }
        memcpy(buf, out, sizeof(float) * 2 * n);
    }
}

/* ------------------------------------------------------------------------- */
/* src/fft.rs : ReFFT                                                          */
/* ------------------------------------------------------------------------- */
struct rco_refft {
    size_t window_len;
    float *window;
    rco_fft_plan plan; /* fft.rs:27-29: forward and inverse plans of window_len */
    float *buf;        /* window_len complex */
    float *kbuf;
};

rco_refft *rco_refft_new(const float *window, size_t window_len) {
    rco_refft *r = (rco_refft *)calloc(1, sizeof *r);
    r->window_len = window_len;
    r->window = (float *)malloc(sizeof(float) * (window_len ? window_len : 1));
    memcpy(r->window, window, sizeof(float) * window_len);
    rco_fft_plan_init(&r->plan, window_len);
    r->buf = (float *)malloc(sizeof(float) * 2 * (window_len ? window_len : 1));
    r->kbuf = (float *)malloc(sizeof(float) * 2 * (window_len ? window_len : 1));
    return r;
}

void rco_refft_free(rco_refft *r) {
    if (!r) return;
    rco_fft_plan_free(&r->plan);
    free(r->window);
    free(r->buf);
    free(r->kbuf);
    free(r);
}

/* fft.rs:50-61 */
void rco_refft_forward(rco_refft *r, const float *samples, size_t n_samples, float *out_reim) {
    const size_t n = r->window_len;
    size_t m = n_samples < n ? n_samples : n; /* zip stops at the shorter (fft.rs:51-55) */
    for (size_t i = 0; i < m; i++) {
        out_reim[2 * i] = samples[i] * r->window[i];
        out_reim[2 * i + 1] = 0.0f;
    }
    for (size_t i = m; i < n; i++) { /* fft.rs:56-58 zero extension */
        out_reim[2 * i] = 0.0f;
        out_reim[2 * i + 1] = 0.0f;
    }
    rco_fft_process(&r->plan, out_reim, 0); /* fft.rs:59 */
}

/* fft.rs:63-74 */
void rco_refft_resynth_from_spectrum(rco_refft *r, const float *spec_reim, uint64_t phase_key,
                                     float *out) {
    const size_t n = r->window_len;
    float *buf = r->buf;
    for (size_t j = 0; j < n; j++) {
        float theta = rco_phase_theta(phase_key, (uint32_t)j, (uint32_t)n);
        /* Complex32::new(0.0, theta).exp() == from_polar(exp(0.0), theta)
         * == (1.0 * cos theta, 1.0 * sin theta)   (num-complex 0.4.6) */
        float er = 1.0f * cosf(theta);
        float ei = 1.0f * sinf(theta);
        float norm = hypotf(spec_reim[2 * j], spec_reim[2 * j + 1]); /* c.norm() */
        buf[2 * j] = er * norm;
        buf[2 * j + 1] = ei * norm;
    }
    rco_fft_process(&r->plan, buf, 1); /* fft.rs:69 */
    for (size_t i = 0; i < n; i++)     /* fft.rs:70-73 */
        out[i] = (buf[2 * i] / (float)n) * r->window[i];
}

/* fft.rs:42-48 (+ the kernel call of fft.rs:76-108 in its C-ABI shape) */
void rco_refft_resynth(rco_refft *r, const float *samples, size_t n_samples, uint64_t phase_key,
                       rco_freq_kernel kernel, void *user, uint64_t time_ms, float *out) {
    const size_t n = r->window_len;
    float *spec = r->kbuf;
    rco_refft_forward(r, samples, n_samples, spec);
    if (kernel) {
        float *tmp = (float *)malloc(sizeof(float) * 2 * (n ? n : 1));
        int rc = kernel(time_ms, spec, tmp, n, user);
        if (rc == 0) memcpy(spec, tmp, sizeof(float) * 2 * n);
        /* rc != 0 == panic: "retrying with last or noop" (fft.rs:100-106) -> noop */
        free(tmp);
    }
    rco_refft_resynth_from_spectrum(r, spec, phase_key, out);
}

/* ------------------------------------------------------------------------- */
/* src/stretcher.rs : Stretcher                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
    float *p;
    size_t off, len, cap;
} rco_deque;

static void dq_reserve(rco_deque *d, size_t extra) {
    if (d->off + d->len + extra <= d->cap) return;
    if (d->off) {
        memmove(d->p, d->p + d->off, d->len * sizeof(float));
        d->off = 0;
    }
    if (d->len + extra > d->cap) {
        size_t nc = d->cap ? d->cap : 1024;
        while (nc < d->len + extra) nc *= 2;
        d->p = (float *)realloc(d->p, nc * sizeof(float));
        d->cap = nc;
    }
}
static void dq_extend(rco_deque *d, const float *src, size_t n) {
    dq_reserve(d, n);
    if (src)
        memcpy(d->p + d->off + d->len, src, n * sizeof(float));
    else
        memset(d->p + d->off + d->len, 0, n * sizeof(float));
    d->len += n;
}
/* slice-deque 0.3.0 truncate_front(len): keep the LAST `len` elements; no-op if len >= self.len */
static void dq_truncate_front(rco_deque *d, size_t keep) {
    if (keep >= d->len) return;
    d->off += d->len - keep;
    d->len = keep;
}
static float *dq_ptr(rco_deque *d) { return d->p + d->off; }

struct rco_stretcher {
    size_t skip_debt; /* samples of future input to discard (step > buffered length) */
    uint32_t sample_rate;
    uint16_t channels;
    float buffer_secs;
    float corrected_amp_factor;
    int pitch_multiple;
    float *amp_correction_envelope;
    rco_refft *re_fft;
    size_t window_len, half_window_len, samples_needed_per_window, sample_step_len;
    int done;
    rco_deque input_buf, output_buf;
    /* the crossbeam Receiver<Vec<f32>> (stretcher.rs:14): a queue of pending samples + closed */
    rco_deque pending;
    int input_closed;
    /* resumable next_window state */
    size_t iter_output_buf_pos;
    int in_window;
    /* phase source + kernel */
    uint64_t seed;
    uint32_t channel_index;
    uint64_t hop;
    rco_freq_kernel kernel;
    void *user;
    uint64_t time_ms;
    float *resynth_out;
};

rco_stretcher *rco_stretcher_new(uint32_t sample_rate, uint16_t channels, float factor,
                                 float amplitude, int pitch_multiple, const float *window,
                                 size_t window_len, float buffer_secs, uint64_t seed,
                                 uint32_t channel_index, rco_freq_kernel kernel, void *user) {
    if (pitch_multiple == 0 || pitch_multiple < -128 || pitch_multiple > 127)
        return NULL; /* stretcher.rs:40 assert!(pitch_multiple != 0); i8 */
    if (window_len < 2) return NULL;
    float abs_p = (float)abs(pitch_multiple);
    /* stretcher.rs:42-46 */
    float pitch_shifted_factor = pitch_multiple < 0 ? factor / abs_p : factor * abs_p;
    /* stretcher.rs:47-51 */
    size_t samples_needed = pitch_multiple < 0 ? (size_t)ceilf((float)window_len / abs_p)
                                               : window_len * (size_t)abs(pitch_multiple);
    /* stretcher.rs:53 */
    float corrected = fmaxf(4.0f, pitch_shifted_factor / 4.0f) * amplitude;
    size_t half = window_len / 2; /* stretcher.rs:54 */
    /* stretcher.rs:55: (window_len as f32 / (psf * 2.0)) as usize — saturating cast */
    float stepf = (float)window_len / (pitch_shifted_factor * 2.0f);
    size_t step;
    if (!(stepf >= 1.0f)) return NULL; /* step 0 / NaN: the reference never terminates */
    if (stepf >= 1.8446744e19f)
        step = (size_t)-1;
    else
        step = (size_t)stepf;
    /* step > window_len (psf < 0.5, README "-f 0.2 to speed up 5x") is supported: see the
     * deliberate deviation at the truncate in rco_stretcher_next_window. */

    rco_stretcher *s = (rco_stretcher *)calloc(1, sizeof *s);
    s->sample_rate = sample_rate;
    s->channels = channels;
    s->buffer_secs = buffer_secs;
    s->corrected_amp_factor = corrected;
    s->pitch_multiple = pitch_multiple;
    s->window_len = window_len;
    s->half_window_len = half;
    s->samples_needed_per_window = samples_needed;
    s->sample_step_len = step;
    s->amp_correction_envelope = (float *)malloc(sizeof(float) * (half ? half : 1));
    rco_hanning_crossfade_compensation(half, s->amp_correction_envelope); /* stretcher.rs:56 */
    s->re_fft = rco_refft_new(window, window_len);                        /* stretcher.rs:57 */
    dq_extend(&s->output_buf, NULL, half);                                /* stretcher.rs:58-59 */
    s->seed = seed;
    s->channel_index = channel_index;
    s->kernel = kernel;
    s->user = user;
    s->resynth_out = (float *)malloc(sizeof(float) * window_len);
    return s;
}

void rco_stretcher_free(rco_stretcher *s) {
    if (!s) return;
    rco_refft_free(s->re_fft);
    free(s->amp_correction_envelope);
    free(s->input_buf.p);
    free(s->output_buf.p);
    free(s->pending.p);
    free(s->resynth_out);
    free(s);
}

void rco_stretcher_send(rco_stretcher *s, const float *chunk, size_t n) {
    if (!s->input_closed) dq_extend(&s->pending, chunk, n);
}
void rco_stretcher_close_input(rco_stretcher *s) { s->input_closed = 1; }
int rco_stretcher_is_done(const rco_stretcher *s) { return s->done; }

/* stretcher.rs:82-85 */
size_t rco_stretcher_channel_bound(const rco_stretcher *s) {
    float v = ((float)s->window_len / (float)s->sample_rate) / s->buffer_secs;
    return (size_t)ceilf(v);
}

size_t rco_stretcher_max_window_out(const rco_stretcher *s) {
    size_t r = rco_resample_len(s->samples_needed_per_window, s->pitch_multiple);
    return r == (size_t)-1 ? s->samples_needed_per_window : r;
}

/* stretcher.rs:123-135. The pending queue is drained chunk-wise like recv(); on a closed,
 * empty channel the buffer is zero-padded to n and done is set. */
int rco_stretcher_ensure_input(rco_stretcher *s, size_t n) {
    while (s->input_buf.len < n) {
        if (s->pending.len) {
            size_t skip = s->skip_debt < s->pending.len ? s->skip_debt : s->pending.len;
            s->skip_debt -= skip; /* samples a step larger than the buffer still owes (see next_window) */
            dq_extend(&s->input_buf, dq_ptr(&s->pending) + skip, s->pending.len - skip);
            s->pending.len = 0;
            s->pending.off = 0;
        } else if (s->input_closed) {
            dq_extend(&s->input_buf, NULL, n - s->input_buf.len); /* resize(n, 0.0) */
            s->done = 1;
        } else {
            return RCO_WOULD_BLOCK;
        }
    }
    return RCO_OK;
}

size_t rco_stretcher_input_len(const rco_stretcher *s) { return s->input_buf.len; }
const float *rco_stretcher_input_ptr(const rco_stretcher *s) {
    return s->input_buf.p + s->input_buf.off;
}
size_t rco_stretcher_step(const rco_stretcher *s) { return s->sample_step_len; }
size_t rco_stretcher_samples_needed(const rco_stretcher *s) {
    return s->samples_needed_per_window;
}
float rco_stretcher_amp(const rco_stretcher *s) { return s->corrected_amp_factor; }
uint64_t rco_stretcher_hops_done(const rco_stretcher *s) { return s->hop; }
void rco_stretcher_set_time_ms(rco_stretcher *s, uint64_t t) { s->time_ms = t; }

/* stretcher.rs:87-121 */
int rco_stretcher_next_window(rco_stretcher *s, float *out, size_t *n_out) {
    const size_t N = s->window_len, H = s->half_window_len, S = s->samples_needed_per_window;
    if (!s->in_window) {
        s->iter_output_buf_pos = 0; /* stretcher.rs:90 */
        s->in_window = 1;
    }
    while (s->output_buf.len < S + H) { /* stretcher.rs:91 */
        int rc = rco_stretcher_ensure_input(s, N); /* stretcher.rs:94 */
        if (rc != RCO_OK) return rc;
        uint64_t key = rco_phase_key(s->seed, s->channel_index, s->hop);
        rco_refft_resynth(s->re_fft, dq_ptr(&s->input_buf), N, key, s->kernel, s->user,
                          s->time_ms, s->resynth_out); /* stretcher.rs:95 */
        s->hop++;
        float *ob = dq_ptr(&s->output_buf);
        const float *fr = s->resynth_out;
        const size_t pos = s->iter_output_buf_pos;
        for (size_t i = 0; i < H; i++) { /* stretcher.rs:96-101 */
            ob[pos + i] = (fr[i] + ob[pos + i]) * s->amp_correction_envelope[i] *
                          s->corrected_amp_factor;
        }
        dq_extend(&s->output_buf, fr + H, N - H); /* stretcher.rs:102-103 */
        s->iter_output_buf_pos += H;              /* stretcher.rs:104 */
        /* stretcher.rs:105-106: truncate_front(len - step). DELIBERATE DEVIATION for step > len
         * (only possible when step > window_len, i.e. speed-up factors below 0.5): the reference's
         * `len - step` underflows there (panic in debug; in release the wrapped value makes the
         * truncate a no-op and the loop never ends). Here the hop grid stays x[k step .. k step + N):
         * everything buffered is dropped and the samples still owed are skipped from the input that
         * arrives later (on a closed channel they are the zero padding of stretcher.rs:129-132). */
        if (s->sample_step_len > s->input_buf.len) {
            s->skip_debt += s->sample_step_len - s->input_buf.len;
            dq_truncate_front(&s->input_buf, 0);
        } else {
            dq_truncate_front(&s->input_buf, s->input_buf.len - s->sample_step_len);
        }
    }
    /* stretcher.rs:108-111 */
    size_t m = rco_resample(dq_ptr(&s->output_buf), S, s->pitch_multiple, out);
    if (m == (size_t)-1) return RCO_EINVAL; /* pitch_multiple == -1 panics: resampler.rs:11 */
    dq_truncate_front(&s->output_buf, H); /* stretcher.rs:112 */
    s->in_window = 0;
    *n_out = m;
    return RCO_OK;
}

/* ------------------------------------------------------------------------- */
/* src/main.rs:131-155 + src/stretcher_processor.rs:56-71                      */
/* ------------------------------------------------------------------------- */
int rco_stretch_offline(uint16_t channels, const float *const *in, size_t len,
                        uint32_t sample_rate, size_t window_len, float factor, float amplitude,
                        int pitch_multiple, uint64_t seed, rco_freq_kernel kernel, void *user,
                        float *const *out, size_t out_cap, size_t *out_len) {
    if (channels == 0) return RCO_EINVAL;
    float *window = (float *)malloc(sizeof(float) * (window_len ? window_len : 1));
    rco_hanning(window_len, window); /* main.rs:131 */
    rco_stretcher **st = (rco_stretcher **)calloc(channels, sizeof *st);
    int rc = RCO_OK;
    for (uint16_t c = 0; c < channels; c++) { /* main.rs:133-153 */
        st[c] = rco_stretcher_new(sample_rate, channels, factor, amplitude, pitch_multiple, window,
                                  window_len, 1.0f, seed, c, kernel, user);
        if (!st[c]) {
            rc = RCO_EINVAL;
            goto cleanup;
        }
        rco_stretcher_send(st[c], in[c], len); /* main.rs:148: whole channel, one chunk */
        rco_stretcher_close_input(st[c]);      /* tx dropped at end of closure */
    }
    {
        size_t *pos = (size_t *)calloc(channels, sizeof(size_t));
        size_t cap_w = rco_stretcher_max_window_out(st[0]);
        float *wbuf = (float *)malloc(sizeof(float) * (cap_w ? cap_w : 1));
        int running = 1;
        while (running) { /* stretcher_processor.rs:56-71 */
            for (uint16_t c = 0; c < channels; c++) {
                if (rco_stretcher_is_done(st[c])) { /* :64-68 */
                    running = 0;
                    break;
                }
                size_t m = 0;
                rc = rco_stretcher_next_window(st[c], wbuf, &m); /* :69 */
                if (rc != RCO_OK) {
                    running = 0;
                    break;
                }
                if (pos[c] + m > out_cap) {
                    rc = -2;
                    running = 0;
                    break;
                }
                memcpy(out[c] + pos[c], wbuf, m * sizeof(float));
                pos[c] += m;
            }
        }
        if (out_len) *out_len = pos[0];
        for (uint16_t c = 1; c < channels; c++)
            if (out_len && pos[c] < *out_len) *out_len = pos[c];
        free(pos);
        free(wbuf);
    }
cleanup:
    for (uint16_t c = 0; c < channels; c++) rco_stretcher_free(st[c]);
    free(st);
    free(window);
    return rc;
}

size_t rco_offline_output_len(size_t len, size_t window_len, float factor, int pitch_multiple) {
    float w[2] = {1.0f, 1.0f};
    (void)w;
    if (pitch_multiple == 0 || pitch_multiple == -1 || window_len < 2) return 0;
    float abs_p = (float)abs(pitch_multiple);
    float psf = pitch_multiple < 0 ? factor / abs_p : factor * abs_p;
    float stepf = (float)window_len / (psf * 2.0f);
    if (!(stepf >= 1.0f)) return 0;
    size_t step = (size_t)stepf;
    size_t H = window_len / 2;
    size_t S = pitch_multiple < 0 ? (size_t)ceilf((float)window_len / abs_p)
                                  : window_len * (size_t)abs(pitch_multiple);
    /* hops per window: smallest h with H + h*(N-H) >= S + H */
    size_t tail = window_len - H;
    size_t hpw = (S + tail - 1) / tail;
    /* first hop index k_d at which the closed channel runs short */
    size_t kd = len >= window_len ? (len - window_len) / step + 1 : 0;
    size_t windows = kd / hpw + 1;
    size_t per = rco_resample_len(S, pitch_multiple);
    return windows * per;
}
