/*
 * rocoder_hip.h — C-ABI of the MI355X (gfx950) stretch engine: the drop-in boundary for
 * rocoder's analysis -> kernel -> resynthesis -> overlap-add hot path.
 *
 * Plain C: opaque handle, plain pointers and sizes, int status codes, no C++/torch types.
 * Every entry point names the reference interface (file:line under the rocoder source
 * tree, v0.4.0) that it replaces. A Rust host binds this with one `extern "C"` block
 * (INTEGRATION.md shows the exact stub and where `Stretcher` calls into it).
 *
 * Threading: one thread at a time per handle (the reference calls every Stretcher from
 * the single StretcherProcessor thread: src/stretcher_processor.rs:55-71). HIP streams,
 * events and scratch buffers are private to the handle.
 *
 * The library has NO CPU fallback: every compute entry point fails with RC_ENODEVICE when
 * no gfx950 device is usable.
 */
#ifndef ROCODER_HIP_H
#define ROCODER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_ABI_VERSION 5 /* 3: + rc_shard_plan / rc_multi_*; 4: + rc_engine_next_window_view, rc_multi_set_staging,
                          * rc_calib_valu; 5: + rc_host_alloc / rc_host_free (struct layouts unchanged since 2) */

/* status codes */
#define RC_OK 0
#define RC_WOULD_BLOCK 1   /* the reference would block in Receiver::recv (src/stretcher.rs:125) */
#define RC_EINVAL (-1)     /* the reference would assert/panic/never terminate */
#define RC_ENODEVICE (-2)  /* no usable HIP device */
#define RC_EUNSUPPORTED (-3) /* valid for the reference, not (yet) on the GPU path */
#define RC_ENOMEM (-4)
#define RC_EHIP (-5)       /* HIP runtime error; see rc_last_error() */
#define RC_ECAPACITY (-6)  /* caller buffer too small */

/*
 * User frequency kernel. C-ABI shape of the reference's hot-swapped
 *   #[no_mangle] pub fn apply(elapsed_ms: usize, input: Vec<(f32,f32)>) -> Vec<(f32,f32)>
 * (README.md:106-112; resolved and called at src/fft.rs:93-95). `in_reim`/`out_reim` hold
 * n_bins == window_len interleaved (re,im) pairs in natural DFT order (ALL N bins).
 * `time_ms` is Unix-epoch milliseconds as in src/fft.rs:89-92. A non-zero return is the
 * equivalent of a panic (src/fft.rs:100-106): the hop falls back to the unmodified spectrum.
 * Called on the engine's calling thread, per channel in hop order.
 */
typedef int (*rc_freq_kernel)(uint64_t time_ms, const float *in_reim, float *out_reim,
                              size_t n_bins, void *user);

/* Stretcher::new arguments (src/stretcher.rs:30-39) + main.rs:131-147 call-site values. */
typedef struct rc_config {
    uint32_t struct_size;    /* = sizeof(rc_config), ABI guard */
    uint32_t window_len;     /* -w/--window (src/main.rs:34); even, 4..65536 (powers of two: the fast kernels) */
    float factor;            /* -f/--factor (src/main.rs:46-52) */
    float amplitude;         /* -a/--amplitude */
    int32_t pitch_multiple;  /* -p/--pitch_multiple, i8 in the reference, != 0 */
    uint32_t sample_rate;    /* AudioSpec.sample_rate (src/audio.rs:31-37) */
    uint32_t channels;       /* AudioSpec.channels: one Stretcher per channel (main.rs:133) */
    float buffer_secs;       /* -b/--buffer (Duration, default 1 s) */
    uint64_t seed;           /* phase-source seed (replaces thread_rng, src/fft.rs:64) */
    int32_t device;          /* HIP device ordinal */
    uint32_t max_batch_hops; /* streaming look-ahead cap per launch and channel; 0 = default (8 MiB of windows) */
    const float *window;     /* host, window_len floats; NULL = windows::hanning (main.rs:131) */
    rc_freq_kernel kernel;   /* --freq-kernel; NULL = none */
    void *kernel_user;
    uint64_t kernel_time_ms; /* 0 = wall clock (src/fft.rs:89-92); non-zero = fixed (tests) */
    /* Host threads that call `kernel`: 0 or 1 = one thread and the reference's call order (windows outer,
     * channels inner: src/stretcher_processor.rs:63-70). n > 1 = the channels are dealt to up to n threads;
     * each channel still sees its hops in order, but calls of different channels interleave, so `kernel`
     * must be re-entrant and must not carry state across channels. */
    uint32_t kernel_threads;
    /* Curated frequency kernels that run on the GPU: the spectrum never leaves the device.
     * They take the place of `kernel` (setting both is RC_EINVAL). All act on the N-bin spectrum X of a hop
     * like an apply() would, Y = K(X), before |Y| is taken (src/fft.rs:42-48,67):
     *   RC_DK_GAIN   Y[j] = dk_gain X[j]                      (README.md:121-128 with any factor)
     *   RC_DK_BAND   Y[j] = (dk_lo_bin <= min(j, N - j) <= dk_hi_bin ? dk_gain : dk_gain_outside) X[j]
     *   RC_DK_SHIFT  Y[j] = X[j - dk_shift_bins] for 0 <= j <= N/2 (0 where j - shift leaves [0, N/2]),
     *                Y[N - j] = conj(Y[j]) : the spectrum of a real signal moved up or down by whole bins
     * Cost: GAIN rides on the amplitude of the fused kernels (free); BAND is applied inside the fused kernel of the
     * default 16384-sample window (a few instructions per bin pair) and otherwise, like SHIFT, runs as forward
     * transform -> kernel -> resynthesis on device scratch. */
    uint32_t device_kernel;
    float dk_gain, dk_gain_outside;
    uint32_t dk_lo_bin, dk_hi_bin;
    int32_t dk_shift_bins;
} rc_config;
#define RC_DK_NONE 0
#define RC_DK_GAIN 1
#define RC_DK_BAND 2
#define RC_DK_SHIFT 3

/* Values derived in Stretcher::new (src/stretcher.rs:40-56). */
typedef struct rc_params {
    uint32_t window_len, half_window_len;
    uint64_t samples_needed_per_window;
    uint32_t sample_step_len;
    uint32_t hops_per_window; /* iterations of the loop at src/stretcher.rs:91 */
    uint32_t window_out_len;  /* samples returned per next_window() */
    float corrected_amp_factor;
    float pitch_shifted_factor;
} rc_params;

typedef struct rc_engine rc_engine;

/* Thread-local description of the last failure on this thread. */
const char *rc_last_error(void);
int rc_abi_version(void);
/* Identifies the kernel generation inside the library (e.g. "hop4/r02b"): measurement files under
 * profiles/ carry it, so a counter summary is only ever quoted next to the kernels it was taken on. */
const char *rc_kernel_id(void);
/* Number of usable gfx950 devices (0 when none; never fails). */
int rc_device_count(void);

/* Pure host helpers (no device): parameter derivation of Stretcher::new
 * (src/stretcher.rs:40-56) and the output length of an offline `-o` run
 * (src/stretcher.rs:91,123-134 + src/stretcher_processor.rs:64-69). */
int rc_derive_params(const rc_config *cfg, rc_params *out);
size_t rc_offline_output_len(const rc_config *cfg, size_t in_len);
/* Phase-source spec (replaces rand::thread_rng at src/fft.rs:64-67), exposed for tests. */
uint64_t rc_phase_key(uint64_t seed, uint32_t channel, uint64_t hop);
uint32_t rc_phase_hash(uint64_t key, uint32_t counter);
/* theta in [0, pi) of bin `bin` of an n_bins-point spectrum: bins b < n_bins/2 take the top 23 bits
 * of rc_phase_hash(key, b) (rand 0.8.5's f32 draw), bins b >= n_bins/2 the low 16 bits of
 * rc_phase_hash(key, b - n_bins/2). */
float rc_phase_theta(uint64_t key, uint32_t bin, uint32_t n_bins);

/* Stretcher::new for all channels (src/main.rs:133-153, src/stretcher.rs:30-76) +
 * ReFFT::new (src/fft.rs:25-40): builds window/envelope/twiddle tables on the device. */
int rc_engine_create(const rc_config *cfg, rc_engine **out);
void rc_engine_destroy(rc_engine *e);
int rc_engine_get_params(const rc_engine *e, rc_params *out);

/* ---- streaming seam: one call per reference call ------------------------------------- */
/* Sender<Vec<f32>>::send on the channel's input (src/main.rs:148; src/stretcher.rs:125-127) */
int rc_engine_push_input(rc_engine *e, uint32_t channel, const float *samples, size_t n);
/* dropping the Sender: recv() then fails and the tail is zero-padded (src/stretcher.rs:129-132) */
int rc_engine_close_input(rc_engine *e, uint32_t channel);
/* Stretcher::next_window (src/stretcher.rs:87-121). Writes rc_params.window_out_len samples.
 * RC_WOULD_BLOCK when the reference would block waiting for input. */
int rc_engine_next_window(rc_engine *e, uint32_t channel, float *out, size_t out_cap,
                          size_t *n_out);
/* The same hand-out without the copy: *window points at the window's window_out_len samples inside the engine's
 * pinned host block, valid until the next rc_engine_next_window / _view call on the SAME channel. The reference moves
 * a freshly allocated Vec<f32> into the bounded queue (src/stretcher.rs:112-120, src/stretcher_processor.rs:69); a
 * host that writes the window straight to its sink (src/main.rs:197-203: the WAV writer) needs no Vec at all. On a
 * closed channel (whole input known) the batches after the one being handed out - every closed channel's together, up to
 * two in flight - are computed and copied meanwhile. */
int rc_engine_next_window_view(rc_engine *e, uint32_t channel, const float **window, size_t *n_out);
/* Stretcher::is_done (src/stretcher.rs:78-80): 1 / 0, or <0 on error. */
int rc_engine_is_done(const rc_engine *e, uint32_t channel);
/* Stretcher::channel_bound (src/stretcher.rs:82-85) */
size_t rc_engine_channel_bound(const rc_engine *e);

/* ---- offline fast path (`-o`, all hops known up-front) ------------------------------- */
/* Whole-job stretch of `channels` host arrays of `in_len` samples: what main.rs:133-155 +
 * StretcherProcessor::start (src/stretcher_processor.rs:56-71) + AudioBus::into_audio
 * (src/audio.rs:152-172) produce. out[c] must hold rc_offline_output_len() samples. */
/* Upload, compute and download run as a pipeline over window chunks (the download of chunk i under the kernel of
 * chunk i + 1 and the upload of chunk i + 2). Rows that came from rc_host_alloc (or that the caller registered with
 * hipHostRegister) are the DMA's source / target themselves; pageable rows are staged through pinned slots by up to
 * eight copy threads. Blocking: out[c] is complete on return. */
int rc_engine_stretch_host(rc_engine *e, const float *const *in, size_t in_len, float *const *out,
                           size_t out_cap, size_t *out_len);
/* Page-locked host memory for the host-form calls (the `Vec<f32>` a Rust host would otherwise hand over, src/main.rs:
 * 148, src/audio.rs:152-172): rows allocated here cross PCIe without a staging copy. rc_host_free(NULL) is a no-op.
 * RC_ENODEVICE without a GPU, RC_ENOMEM when the pages cannot be locked. */
int rc_host_alloc(size_t bytes, void **out);
int rc_host_free(void *p);
/* Same job on DEVICE-resident buffers (channel c at base + c*stride, strides in floats).
 * `hip_stream` is a hipStream_t (NULL = the engine's own stream); the call is asynchronous
 * on that stream unless a user kernel is configured. The engine's scratch (tail copy, seam stash,
 * pipeline buffers) is per handle: a call on a different stream than the previous one first waits
 * (on the device) for that call's last enqueue, so back-to-back calls never overlap on the GPU. */
int rc_engine_stretch_device(rc_engine *e, const float *d_in, size_t in_stride, size_t in_len,
                             float *d_out, size_t out_stride, size_t out_cap, size_t *out_len,
                             void *hip_stream);
/* Sharded form for multi-GPU runs: only channels [ch_first, ch_first+ch_count) and output
 * windows [win_first, win_first+win_count) of the same job; written at d_out offset 0.
 * Hops are independent given the phase source, the one-hop overlap is recomputed locally. */
int rc_engine_stretch_device_range(rc_engine *e, const float *d_in, size_t in_stride,
                                   size_t in_len, uint32_t ch_first, uint32_t ch_count,
                                   uint64_t win_first, uint64_t win_count, float *d_out,
                                   size_t out_stride, size_t out_cap, void *hip_stream);

/* Blocks until everything the engine queued on ITS OWN stream has finished (calls that were given
 * a caller stream are ordered by that stream instead). Like every entry point that touches the
 * device it returns RC_EHIP, once, if an earlier launch left a device error word (a run-seam wait of
 * the window-16384 kernel that expired: the affected output samples were not written). */
int rc_engine_synchronize(rc_engine *e);

/* ---- measurement ---------------------------------------------------------------------- */
/* HIP-event time (ms) and hop count of the hop kernel launches of the last offline call;
 * synchronises on the events. */
int rc_engine_last_kernel_stats(rc_engine *e, float *kernel_ms, uint64_t *hops,
                                uint32_t *launches);
/* The same event time for each of the last min(cap, 64) offline calls, oldest first (a bench runs K
 * calls back to back and reads K kernel durations afterwards); synchronises on the newest. */
int rc_engine_kernel_times(rc_engine *e, float *ms, size_t cap, size_t *n_out);

/* ---- single-hop entry points (ReFFT seam, used by parity tests) ----------------------- */
/* ReFFT::forward_fft (src/fft.rs:50-61): host samples[window_len] -> host spectrum (re,im)*N */
int rc_engine_forward_fft(rc_engine *e, const float *samples, float *out_reim);
/* ReFFT::resynth (src/fft.rs:42-48) for hop `hop` of `channel` (phase key), no overlap-add:
 * host samples[window_len] -> host out[window_len]. Applies the user kernel if configured. */
int rc_engine_resynth(rc_engine *e, uint32_t channel, uint64_t hop, const float *samples,
                      float *out);

/* ---- one process, several GPUs of one node ---------------------------------------------------
 * The reference builds every channel's Stretcher in one process (src/main.rs:133-155) and one thread walks them
 * (src/stretcher_processor.rs:56-71). rc_multi is that job cut over a LIST of devices behind the same C-ABI: one
 * engine and one host thread per listed device, the channel-major sequence of output windows cut into one contiguous
 * piece per device (rc_shard_plan: SURVEY 8 e1's (channel, window range) units; the hop before a piece is
 * recomputed locally, no data-path collective), and every shard's output copied ONCE, straight into its place in the
 * caller's layout (device form: hipMemcpyPeerAsync into the root device's tensor; host form: device -> the caller's
 * arrays). A host frequency kernel (rc_config::kernel) is RC_EUNSUPPORTED here: a stateful apply() sees its
 * channel's hops in order, which a cut job cannot promise; the curated device kernels work. rc_config::device is
 * ignored. A device may be listed more than once (then its engines share it). Threading as for an engine: one thread
 * at a time per rc_multi handle; the handle owns one persistent worker thread per listed device beyond the first (created
 * by rc_multi_create, joined by rc_multi_destroy), and the calling thread's current HIP device is restored on return. */
typedef struct rc_shard {
    uint32_t device_index;        /* index into the device list */
    uint32_t ch_first, ch_count;  /* part of ONE channel, or a block of WHOLE channels */
    uint64_t win_first, win_count;
} rc_shard;
/* Pure host: the plan for `channels` x `total_windows` windows over `n_devices`; writes up to `cap` shards, returns
 * how many the plan has (at most 3 per device). */
size_t rc_shard_plan(uint32_t channels, uint64_t total_windows, uint32_t n_devices, rc_shard *out, size_t cap);

typedef struct rc_multi rc_multi;
int rc_multi_create(const rc_config *cfg, const int32_t *device_ids, uint32_t n_devices, rc_multi **out);
void rc_multi_destroy(rc_multi *m);
uint32_t rc_multi_device_count(const rc_multi *m);
/* Diagnostic: in the device form a share that runs on the root's own device (the root itself, or the root device
 * listed again) reads and writes the caller's tensors in place. force != 0 makes such shares take the span-copy /
 * shard-copy path of a remote device as well, so that a one-GPU box exercises it. Default 0. */
int rc_multi_set_staging(rc_multi *m, int force);
/* rc_engine_stretch_host over all listed devices: each uploads the span of input its shards read and downloads
 * its shards into out[c]. Blocking. */
int rc_multi_stretch_host(rc_multi *m, const float *const *in, size_t in_len, float *const *out, size_t out_cap,
                          size_t *out_len);
/* rc_engine_stretch_device with input and output resident on device_ids[root] (channel c at base + c * stride,
 * strides in floats): the other devices fetch the input span they need from the root and deliver their shards into
 * d_out by peer copies. `hip_stream` (a hipStream_t of the root device, or NULL) is what produced d_in: it is
 * synchronised on entry. Blocking: d_out is complete on return. */
int rc_multi_stretch_device(rc_multi *m, uint32_t root, const float *d_in, size_t in_stride, size_t in_len,
                            float *d_out, size_t out_stride, size_t out_cap, size_t *out_len, void *hip_stream);

/* ---- measurement support -------------------------------------------------------------------
 * Box calibration for bench.py (boxes of one pool differ by several per cent on the same binary): runs a fixed
 * pure-VALU kernel (independent v_pk_fma_f32 chains, eight waves per SIMD, no memory traffic) `launches` times on
 * `hip_stream` of `device` and returns the mean duration of one launch in *ms_per_launch and the time one packed-FMA
 * wave instruction takes on one SIMD in *ns_per_inst. Not on the reference's path. */
int rc_calib_valu(int device, void *hip_stream, uint32_t launches, float *ms_per_launch, float *ns_per_inst);

#ifdef __cplusplus
}
#endif
#endif
