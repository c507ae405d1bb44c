// Internal interface between the host engine (rc_engine.cpp) and the gfx950 kernels
// (rc_*.hip). Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rc {

// One launch = channels [0, n_channels) x hops [hop_first, hop_first + hop_count), cut into
// `runs_per_channel` contiguous runs of `run_len` hops; one workgroup walks one run.
// 0: variant builds only (tools/build_variant.sh x -DRC_BLUESTEIN=0) - window lengths that are not a power of two run
// the first implementation's O(N^2) DFT kernels instead of the chirp-z transforms, as an independent A/B partner
#ifndef RC_BLUESTEIN
#define RC_BLUESTEIN 1
#endif

struct HopParams {
    const float *x;        // channel c at x + c * in_stride; x[0] is absolute sample `in_origin`
    size_t in_stride;
    int64_t in_origin;
    int64_t in_len;        // samples valid at x (absolute index in_origin + in_len is the end)
    // hops k >= tail_hop_first read a zero-padded copy instead (windows running past the end of
    // the closed input): channel c at xtail + c * tail_stride, xtail[0] is absolute tail_origin
    const float *xtail;
    size_t tail_stride;
    int64_t tail_origin;
    int64_t tail_hop_first;
    float *out;            // F (decimated overlap-add), channel c at out + c * out_stride
    size_t out_stride;
    int64_t out_origin;    // absolute F index of out[0]
    const float *window;   // [N]
    const float *env;      // [N/2] hanning_crossfade_compensation
    // default-window fast path of the N = 16384 kernel (null: load window/env from the tables):
    // [2][256][4] = thread t's {cos, sin}(2 pi (2t + e) / (len - 1)), e = 0, 1; len = N, then N/2
    const float *hann_rot;
    const float2 *wtab;    // [M/2]   exp(-2 pi i k / M), M = N/2
    const float2 *rtab;    // [M/4+1] exp(-2 pi i j / N)
    float amp;             // corrected_amp_factor
    uint32_t step;         // sample_step_len
    uint32_t pitch;        // pitch_multiple >= 1
    uint64_t seed_mixed;   // mix64(seed)
    uint32_t ch_first;     // absolute channel index of local channel 0 (phase source)
    uint32_t n_channels;
    int64_t hop_first;
    int64_t hop_count;
    uint32_t runs_per_channel;
    uint32_t run_len;
    // seam hand-over of hop3_kernel (null: every run recomputes the hop before it instead). The first
    // hop of run g > 0 of a channel stashes its windowed head [H] at seam_head + g * H and publishes
    // seam_flag[g] = seam_epoch; run g - 1 adds its last tail to it. Runs are numbered in the order
    // their workgroups start (run_counter, zeroed per launch), so a run only ever waits for a
    // workgroup that started after it and is already resident or next in line.
    float *seam_head;
    uint32_t *seam_flag;
    uint32_t *run_counter;  // RC_RUN_COUNTERS words: hop4_kernel keeps one ticket counter per XCD
    uint32_t seam_epoch;
    // RC_DK_BAND fused into hop4_kernel's pair stage (band_on): bins lo..lo+span of the real spectrum (0..N/2) are
    // scaled by band_gin, the others by band_gout (both already |.|: the resynthesis takes magnitudes)
    uint32_t band_on, band_lo, band_span;
    float band_gin, band_gout;
    // A run that never sees its successor's flag within seam_spin_limit polls gives up, leaves its seam
    // samples unwritten and reports through *err_word (host-visible pinned memory; RC_ERR_SEAM_TIMEOUT):
    // the engine turns that into RC_EHIP. diag_flags is test-only (RC_DIAG_*).
    uint32_t seam_spin_limit;
    uint32_t diag_flags;
    uint32_t *err_word;
    // spectrum modes (user-kernel path)
    float2 *spec;          // [n_channels][hop_count][N] natural-order spectrum
    float *ybuf;           // [n_channels][hop_count][N] windowed resynthesis output y_k
    // window lengths that are not a power of two (launch_gen): N and exp(-2 pi i k / N), k < N
    uint32_t n_generic;
    const float2 *tw_generic;
    // ... of them, the lengths N <= 16384 run chirp-z (Bluestein) transforms of the packed N/2-point sequence in LDS:
    // bl_log2l = log2 of the convolution length L >= N - 1 (0: the O(N^2) DFT kernels), bl_tab = [N/2] chirp
    // exp(-i pi n^2 / (N/2)) | [L/2] exp(-2 pi i k / L) | [L] FFT_L of the conjugate chirp / L in bit-reversed order
    uint32_t bl_log2l;
    const float2 *bl_tab;
    float2 *bl_wk;         // L > 16384 only: [n_channels][hop_count][L] work buffer (the FFT_L then runs in three kernels)
};

struct OlaParams {
    const float *ybuf;     // [n_channels][hop_count][N]
    float *tail;           // [n_channels][H] carried y_{k-1}[H..] (read, then updated)
    float *out;
    size_t out_stride;
    int64_t out_origin;
    const float *env;
    float amp;
    int32_t pitch;            // pitch_multiple: >= 1 decimates, <= -2 interpolates (one hop per window)
    uint32_t samples_needed;  // samples_needed_per_window (pitch < 0)
    uint32_t window_out_len;  // samples per next_window() (pitch < 0)
    uint32_t n_channels;
    int64_t hop_first;
    int64_t hop_count;
    uint32_t log2n;           // 0: the window length is `n` (not a power of two)
    uint32_t n;
};

// Windows larger than one workgroup's LDS (N = 32768, 65536): the N/2-point FFT is split into
// four interleaved quarter-length FFTs that DO fit (stage A / C), glued by a radix-4 step fused
// with the per-bin middle stage (stage B); everything goes through HBM scratch.
struct BigParams {
    const float *x;
    size_t in_stride;
    int64_t in_origin;
    const float *xtail;
    size_t tail_stride;
    int64_t tail_origin;
    int64_t tail_hop_first;
    const float *window;     // [N]
    const float2 *wtab_sub;  // [Ms/2]   exp(-2 pi i k / Ms), Ms = N/8
    const float2 *t1;        // [Ms+1]   exp(-2 pi i j / (N/2))
    const float2 *rtab;      // [Ms/2+1] exp(-2 pi i j / N)
    float2 *ysub;            // [n_channels][hop_count][4][Ms] scratch: Y_s, then U_s (natural order)
    float *ybuf;             // [n_channels][hop_count][N] windowed resynthesis y_k
    float2 *spec;            // [n_channels][hop_count][N] natural-order spectrum (user-kernel path)
    uint32_t step;
    uint64_t seed_mixed;
    uint32_t ch_first;
    uint32_t n_channels;
    int64_t hop_first;
    int64_t hop_count;
    uint32_t log2n;          // 15 or 16
};

// Stage C of the large-window pipeline with the overlap-add fused (pitch >= 1, no user kernel): a
// workgroup walks a run of hops of one quarter and carries y_{k-1}[H..] in registers, so y never
// goes to HBM and the gather-form ola_kernel is not needed.
struct BigOlaParams {
    BigParams b;
    float *out;             // F, channel c at out + c * out_stride
    size_t out_stride;
    int64_t out_origin;
    const float *env;       // [N/2]
    float amp;
    uint32_t pitch;
    const float *tail_in;   // [n_channels][N/2]: y_{hop_first-1}[H..] (read by the first run)
    float *tail_out;        // [n_channels][N/2]: y_{last}[H..]       (written by the last run)
    uint32_t run_len, runs; // hops [r * run_len, ...) of the chunk per run
    uint32_t tail_only;     // 1: compute tails only, store nothing to out
};

// rc_kernel_id(): "hop4=<hash> big4=<hash> hopw=<hash> generic=<hash> spectrum=<hash>", one hash per kernel family
// over its sources, generated at build time into $(OBJDIR)/rc_kernel_id.h (tools/kernel_id.py, csrc/Makefile) - a
// changed kernel can never be quoted next to counters taken with another one

enum HopMode { MODE_FUSED = 0, MODE_FORWARD = 1, MODE_RESYNTH = 2 };
// values a kernel may leave in *HopParams::err_word
constexpr uint32_t RC_ERR_SEAM_TIMEOUT = 1;
// HopParams::diag_flags (ROCODER_DIAG, tests only): the producer of a seam never publishes its flag
constexpr uint32_t RC_DIAG_SKIP_SEAM_PUBLISH = 1;
// run the previous kernel generation of the N = 16384 path (hop3) instead of hop4: A/B timing and the
// bit-exactness test between the two. Only the test-hook library (-DRC_TEST_HOOKS=1, `make hooks`) contains it.
constexpr uint32_t RC_DIAG_PREV_KERNEL = 2;
// test-hook library only: hop2_kernel's computed-window variant (two workgroups per CU) instead of hop4
constexpr uint32_t RC_DIAG_HOP2_HANN = 4;
// test-hook library only: N = 65536 through big4_kernel<64> (round 4's kernel) instead of big5_kernel: A/B partner
constexpr uint32_t RC_DIAG_BIG4_64 = 8;
#ifndef RC_TEST_HOOKS
#define RC_TEST_HOOKS 0
#endif

// Geometry chosen by the kernels for a window length (threads per workgroup, LDS bytes).
bool hop_geometry(int log2n, int *threads, size_t *lds_bytes);
// Workgroups of the fused kernel that share a CU when the kernel variant fixes it (0: derive it
// from hop_geometry's LDS size). The run planner sizes a launch to two rounds of them.
int hop_workgroups_per_cu(int log2n, bool default_window, bool pitch1 = false);
// Workgroups per CU of the wave-local kernels (rc_hopw.hip: N = 4096 / 8192 with the default window), 0 otherwise:
// all runs are equally long, so the planner launches whole multiples of what is resident at once.
int hop_resident_workgroups(int log2n, bool default_window);
// Runs that share one workgroup (= one wave) of the fused generic kernel: 64 / T below N = 512, else 1.
int hop_slots(int log2n);
// Launchers. Return hipSuccess or the launch error. log2n in [5, 14].
hipError_t launch_hop(int log2n, HopMode mode, const HopParams &p, hipStream_t s);
// (between translation units) the N = 16384 fused path: rc_hop16k.hip, and rc_hop16k_prev.hip in the test-hook library
hipError_t launch_hop16k(const HopParams &p, hipStream_t s);
hipError_t launch_hop16k_prev(const HopParams &p, hipStream_t s);
// N = 4096, fused path, default window: one wave per hop (rc_hopw.hip)
hipError_t launch_hopw(const HopParams &p, hipStream_t s);
// N = 512 with the default window: two hops per wave, 8 points per lane (rc_hopw.hip)
hipError_t launch_hopw9(const HopParams &p, hipStream_t s);
// N = 1024 with the default window: two hops per wave (rc_hopw.hip)
hipError_t launch_hopw10(const HopParams &p, hipStream_t s);
// N = 2048 with the default window: one wave per hop, 16 points per lane (rc_hopw.hip)
hipError_t launch_hopw11(const HopParams &p, hipStream_t s);
// N = 8192 with the default window: two waves per hop (rc_hopw.hip)
hipError_t launch_hopw2(const HopParams &p, hipStream_t s);
// tail_only: just save y_{last}[H..] of the chunk as the carried tail (no output written)
hipError_t launch_ola(const OlaParams &p, hipStream_t s, bool tail_only = false);
// resample_slower (src/resampler.rs:20-35) after a FUSED launch at pitch 1: hop k of channel c has its H overlap-added
// samples at obuf + c * o_stride + (k - hop_first) * H; its window is the (S - 1) f samples lerp(O[i], O[i + 1], j / f)
// at out + c * out_stride + (k * window_out_len - out_origin) (one hop per window; the rest of the half window is
// dropped as in the reference, src/stretcher.rs:108-111)
struct ResampleParams {
    const float *obuf;
    size_t o_stride;
    float *out;
    size_t out_stride;
    int64_t out_origin;
    int64_t hop_first, hop_count;
    uint32_t n_channels, half, samples_needed, window_out_len, f;
};
hipError_t launch_resample_slower(const ResampleParams &p, hipStream_t s);
// box calibration (rc_calib_valu): CALIB_ITERS x 16 packed FMAs per wave, eight waves per SIMD
constexpr int CALIB_ITERS = 4096;
hipError_t launch_calib_valu(float *d_out, int n_cu, hipStream_t s);
// stage 0 = A (forward quarter FFTs), 1 = B (radix-4 + middle + radix-4), 2 = C (inverse quarter FFTs)
// mode selects stage B's variant (user-kernel path: MODE_FORWARD, host apply(), MODE_RESYNTH)
hipError_t launch_big(int stage, const BigParams &p, hipStream_t s, HopMode mode = MODE_FUSED);
hipError_t launch_big_cr(const BigOlaParams &p, hipStream_t s);
// Fused large-window kernel (log2n 15 / 16, pitch >= 1, no spectrum kernel): HopParams with
// wtab = exp(-2 pi i k / M) [>= RES/2 + 1], rtab = exp(-2 pi i j / N) [>= RES/2 + 1], RES = N / 32, and for
// builds whose carried tail does not fit the registers (big4_tail_scratch_floats) ybuf = tail scratch of
// runs * n_channels * N/2 floats. One workgroup walks one run and
// recomputes the hop before it for its tail.
hipError_t launch_big4(int log2n, const HopParams &p, hipStream_t s);
// N = 65536 (BASELINE C5): big4_kernel<64>'s arithmetic with wave-local E2 / E3 exchanges (rc_big5.hip); same HopParams
hipError_t launch_big5(const HopParams &p, hipStream_t s);
// N = 32768: the same thread mapping with single-round exchanges (big5s_kernel), four barriers per hop instead of eight
hipError_t launch_big5s(const HopParams &p, hipStream_t s);
size_t big5_lds_bytes();
// floats of per-workgroup tail scratch (HopParams::ybuf) big4_kernel needs per run for this window length: 0 when
// the carried tail y_{k-1}[H..] lives in registers (the default build, both lengths)
size_t big4_tail_scratch_floats(int log2n);

// Curated on-GPU frequency kernels (rc_config::device_kernel, RC_DK_BAND / RC_DK_SHIFT): Y = K(X) on the
// natural-order N-bin spectra [hops_total][N] between the forward and the resynthesis kernels.
struct DevKernelParams {
    const float2 *in;    // [hops_total][N]
    float2 *out;         // may equal `in` for BAND; must differ for SHIFT
    uint32_t log2n;      // 0: the window length is `n`
    uint32_t n;
    uint32_t kind;       // 2 = band, 3 = shift (values of RC_DK_*)
    float gain_in, gain_out;
    uint32_t lo_bin, hi_bin;
    int32_t shift;
    uint64_t hops_total;
};
hipError_t launch_dev_kernel(const DevKernelParams &p, hipStream_t s);

// Window lengths that are not a power of two (any even N): the reference accepts them through rustfft
// (src/main.rs:34, src/fft.rs:27-29). They run as plain O(N^2) DFTs on the device - correct, not fast:
//   stage 0: X[k] = sum_n x[k_hop step + n] w[n] e^{-2 pi i n k / N}   -> p.spec (natural order, all N bins)
//   stage 1: Z[k] = |X[k]| e^{i theta(seed, c, hop, k)}                 (in place)
//   stage 2: y[n] = Re(sum_k Z[k] e^{+2 pi i n k / N}) / N * w[n]       -> p.ybuf
// followed by the gather-form overlap-add (launch_ola).
// One small launch in front of a job: the zero-padded copy of the input tail that the hops past the end of the
// input read (src/stretcher.rs:129-132) and the reset of the run counter of the seam hand-over.
constexpr int RC_RUN_COUNTERS = 8;
struct PrepParams {
    float *xtail;         // [n_channels][tail_len] or nullptr (no hop of the job runs past the input)
    size_t tail_len;
    const float *src;     // first sample of the tail, channel 0
    size_t src_stride;
    size_t real;          // samples that exist (the rest is zero)
    uint32_t n_channels;
    uint32_t *run_counter;  // RC_RUN_COUNTERS words, or nullptr
};
hipError_t launch_prep(const PrepParams &p, hipStream_t s);
hipError_t launch_gen(int stage, const HopParams &p, hipStream_t s);

}  // namespace rc
