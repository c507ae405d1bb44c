/*
 * rocoder_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, f32 arithmetic at the reference's rounding points) of
 * rocoder's analysis -> kernel -> resynthesis -> overlap-add hot path. It exists to
 * CHECK the HIP engine; it is never linked into, imported by, or called from the
 * product path (rocoder_amd/). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY UNPINNED for FFT / phase / overlap-add: the reference (Rust, /root/reference)
 * cannot be built here (no rustc/cargo, rustfft 6.3.0 / rand 0.8.5 / num-complex 0.4.6
 * not vendored), has no tests on src/fft.rs or Stretcher::next_window, and draws its
 * phases from an unseedable thread_rng (src/fft.rs:64). What IS pinned against the
 * reference's own known-answer tests: hanning (src/windows.rs:28-43), rectangular,
 * inverse, resample (src/resampler.rs:42-54), lerp (src/math.rs:67-80) and the
 * end-of-stream behaviour of ensure_input_samples_available (src/stretcher.rs:144-161).
 * The FFT arithmetic is additionally cross-checked against numpy's f64 pocketfft
 * (oracle/oracle_np.py) — an independent implementation of the same unnormalised DFT.
 *
 * The random phase of the reference is replaced by an explicit counter-based phase
 * source theta(seed, channel, hop, bin) (see rco_phase_*), shared as a SPEC (not as
 * code) with the HIP engine.
 */
#ifndef ROCODER_ORACLE_H
#define ROCODER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* C-ABI shape of the reference's `apply(elapsed_ms, Vec<(f32,f32)>) -> Vec<(f32,f32)>`
 * (README.md:106-112, src/fft.rs:93-94). Non-zero return == "kernel panicked". */
typedef int (*rco_freq_kernel)(uint64_t time_ms, const float *in_reim, float *out_reim,
                               size_t n_bins, void *user);

/* ---- src/windows.rs ------------------------------------------------------- */
void rco_hanning(size_t len, float *out);                   /* windows.rs:4-9   */
void rco_rectangular(size_t len, float *out);               /* windows.rs:12-14 */
void rco_inverse(const float *in, size_t len, float *out);  /* windows.rs:19-21 */
/* ---- src/crossfade.rs ----------------------------------------------------- */
void rco_hanning_crossfade_compensation(size_t len, float *out); /* crossfade.rs:4-10 */
/* ---- src/math.rs, src/resampler.rs ---------------------------------------- */
float rco_lerp(float start, float end, float ratio);        /* math.rs:28-30 */
/* returns number of samples written, or (size_t)-1 for an invalid factor
 * (the reference panics: resampler.rs:11). out must hold rco_resample_len(). */
size_t rco_resample_len(size_t n, int factor);
size_t rco_resample(const float *in, size_t n, int factor, float *out); /* resampler.rs:3-35 */

/* ---- phase source (replaces rand::thread_rng, src/fft.rs:64-67) ------------ */
uint64_t rco_phase_key(uint64_t seed, uint32_t channel, uint64_t hop);
uint32_t rco_phase_hash(uint64_t key, uint32_t bin);
/* in [0, pi): fft.rs:13 TWO_PI == PI. Bins b and b + n_bins/2 share one hash (see the .c file) */
float rco_phase_theta(uint64_t key, uint32_t bin, uint32_t n_bins);

/* ---- src/fft.rs : ReFFT ---------------------------------------------------- */
typedef struct rco_refft rco_refft;
rco_refft *rco_refft_new(const float *window, size_t window_len); /* fft.rs:25-40 */
void rco_refft_free(rco_refft *r);
/* forward_fft (fft.rs:50-61): out_reim holds window_len interleaved (re,im). */
void rco_refft_forward(rco_refft *r, const float *samples, size_t n_samples, float *out_reim);
/* resynth_from_fft_result (fft.rs:63-74) with the phase source. */
void rco_refft_resynth_from_spectrum(rco_refft *r, const float *spec_reim, uint64_t phase_key,
                                     float *out);
/* resynth (fft.rs:42-48): forward -> optional kernel -> resynth. */
void rco_refft_resynth(rco_refft *r, const float *samples, size_t n_samples, uint64_t phase_key,
                       rco_freq_kernel kernel, void *user, uint64_t time_ms, float *out);

/* ---- src/stretcher.rs : Stretcher ----------------------------------------- */
typedef struct rco_stretcher rco_stretcher;

#define RCO_OK 0
#define RCO_WOULD_BLOCK 1 /* recv() would block: push more input or close the channel */
#define RCO_EINVAL (-1)

/* Stretcher::new (stretcher.rs:30-76). window is copied. Returns NULL when the reference
 * would assert (pitch_multiple == 0) or never terminate (sample_step_len == 0). */
rco_stretcher *rco_stretcher_new(uint32_t sample_rate, uint16_t channels, float factor,
                                 float amplitude, int pitch_multiple, const float *window,
                                 size_t window_len, float buffer_secs, uint64_t seed,
                                 uint32_t channel_index, rco_freq_kernel kernel, void *user);
void rco_stretcher_free(rco_stretcher *s);
/* The Receiver<Vec<f32>> side (stretcher.rs:14,125): send a chunk / drop the sender. */
void rco_stretcher_send(rco_stretcher *s, const float *chunk, size_t n);
void rco_stretcher_close_input(rco_stretcher *s);
int rco_stretcher_is_done(const rco_stretcher *s);               /* stretcher.rs:78-80 */
size_t rco_stretcher_channel_bound(const rco_stretcher *s);      /* stretcher.rs:82-85 */
size_t rco_stretcher_max_window_out(const rco_stretcher *s);     /* capacity for next_window */
/* ensure_input_samples_available (stretcher.rs:123-135). */
int rco_stretcher_ensure_input(rco_stretcher *s, size_t n);
size_t rco_stretcher_input_len(const rco_stretcher *s);
const float *rco_stretcher_input_ptr(const rco_stretcher *s);
/* next_window (stretcher.rs:87-121). Writes *n_out samples. */
int rco_stretcher_next_window(rco_stretcher *s, float *out, size_t *n_out);
/* derived parameters, for tests */
size_t rco_stretcher_step(const rco_stretcher *s);
size_t rco_stretcher_samples_needed(const rco_stretcher *s);
float rco_stretcher_amp(const rco_stretcher *s);
uint64_t rco_stretcher_hops_done(const rco_stretcher *s);
void rco_stretcher_set_time_ms(rco_stretcher *s, uint64_t t); /* deterministic kernel time */

/* ---- src/main.rs:131-155 + src/stretcher_processor.rs:56-71 ----------------
 * Offline (`-o`) run: hanning window, one Stretcher per channel fed the whole channel
 * as one chunk, round-robin next_window until the first channel reports done.
 * out[c] must hold out_cap samples; *out_len = samples written per channel.
 * Returns RCO_OK, RCO_EINVAL (bad params) or -2 (out_cap too small). */
int rco_stretch_offline(uint16_t channels, const float *const *in, size_t len,
                        uint32_t sample_rate, size_t window_len, float factor, float amplitude,
                        int pitch_multiple, uint64_t seed, rco_freq_kernel kernel, void *user,
                        float *const *out, size_t out_cap, size_t *out_len);
/* Number of samples rco_stretch_offline will emit per channel (0 on invalid params). */
size_t rco_offline_output_len(size_t len, size_t window_len, float factor, int pitch_multiple);

#ifdef __cplusplus
}
#endif
#endif
