// Host side of the C-ABI in include/rocoder_hip.h: parameter derivation, device tables,
// launch planning, the streaming (Stretcher) seam and the user-kernel round trip.
// There is deliberately no CPU compute path here: without a gfx950 device every compute
// entry point fails with RC_ENODEVICE.
#include "../../include/rocoder_hip.h"
#include "rc_kernels.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <complex>
#include <condition_variable>
#include <mutex>
#include <vector>

#include <sys/mman.h>  // madvise(MADV_POPULATE_WRITE): host_pipeline

namespace {
// Worker threads that are always joined: if a later std::thread constructor throws (std::system_error), the ones
// already started must not be destroyed joinable (std::terminate) - the C ABI's catch (...) turns the failure into
// RC_E* only if unwinding gets that far.
struct JoinedThreads {
    std::vector<std::thread> th;
    template <class F, class... A>
    bool start(F &&f, A &&...a) {  // false: could not start (the caller runs that share itself)
        try {
            th.emplace_back(std::forward<F>(f), std::forward<A>(a)...);
            return true;
        } catch (const std::system_error &) {
            return false;
        }
    }
    void join() {
        for (auto &t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
    ~JoinedThreads() { join(); }
};

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define RC_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return fail(RC_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                        __FILE__, __LINE__);                                               \
    } while (0)

constexpr float kPiF32 = 3.14159274101257324219f;

uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// windows::hanning (src/windows.rs:4-9), f32 operation order
void hanning(size_t len, float *out) {
    const float two_pi = kPiF32 * 2.0f;
    for (size_t i = 0; i < len; i++)
        out[i] = 0.5f - (cosf(((float)i * two_pi) / (float)(len - 1)) * 0.5f);
}
// crossfade::hanning_crossfade_compensation (src/crossfade.rs:4-10)
void crossfade_comp(size_t len, float *out) {
    const float two_pi = kPiF32 * 2.0f;
    const float h = (1.0f + sqrtf(sqrtf(0.5f))) * 0.5f;
    for (size_t i = 0; i < len; i++)
        out[i] = 0.5f - ((1.0f - h) * cosf(((float)i * two_pi) / (float)(len - 1)));
}

int ilog2_exact(uint32_t v) {
    if (!v || (v & (v - 1))) return -1;
    int l = 0;
    while ((1u << l) < v) l++;
    return l;
}

// Stretcher::new derivation (src/stretcher.rs:40-56), f32 arithmetic as in the reference.
int derive(const rc_config *c, rc_params *o) {
    if (!c || c->struct_size != sizeof(rc_config)) return fail(RC_EINVAL, "bad rc_config size");
    if (c->pitch_multiple == 0) return fail(RC_EINVAL, "pitch_multiple must be non-zero (src/stretcher.rs:40)");
    if (c->pitch_multiple < -128 || c->pitch_multiple > 127)
        return fail(RC_EINVAL, "pitch_multiple is an i8 in the reference");
    if (c->pitch_multiple == -1)
        return fail(RC_EINVAL, "pitch_multiple -1 panics in resampler::resample (src/resampler.rs:11)");
    if (c->window_len < 2) return fail(RC_EINVAL, "window_len < 2");
    if (c->device_kernel > RC_DK_SHIFT) return fail(RC_EINVAL, "unknown device_kernel %u", c->device_kernel);
    if (c->device_kernel != RC_DK_NONE && c->kernel)
        return fail(RC_EINVAL, "a device kernel takes the place of the host frequency kernel: set one of them");
    if (c->channels == 0) return fail(RC_EINVAL, "channels == 0");
    const float abs_p = (float)std::abs(c->pitch_multiple);
    const float psf = c->pitch_multiple < 0 ? c->factor / abs_p : c->factor * abs_p;
    const uint64_t S = c->pitch_multiple < 0
                           ? (uint64_t)ceilf((float)c->window_len / abs_p)
                           : (uint64_t)c->window_len * (uint64_t)std::abs(c->pitch_multiple);
    const float amp = fmaxf(4.0f, psf / 4.0f) * c->amplitude;
    const uint32_t H = c->window_len / 2;
    const float stepf = (float)c->window_len / (psf * 2.0f);
    if (!(stepf >= 1.0f))
        return fail(RC_EINVAL, "sample_step_len == 0: the reference never terminates (src/stretcher.rs:55,105-106)");
    // stepf > window_len (factors below 0.5: README "-f 0.2 to speed up 5x") is supported: hop k still reads
    // x[k step .. k step + N) and the stream ends at the first hop whose window runs past the input, the
    // samples between two windows are skipped. (The reference's `len - step` at src/stretcher.rs:105-106
    // underflows once fewer than `step` samples are buffered - a panic or an endless loop at the end of
    // every such file; a deliberate fix, mirrored by the oracle.)
    o->window_len = c->window_len;
    o->half_window_len = H;
    o->samples_needed_per_window = S;
    o->sample_step_len = (uint32_t)stepf;
    const uint64_t tail = c->window_len - H;
    o->hops_per_window = (uint32_t)((S + tail - 1) / tail);
    if (c->pitch_multiple >= 1)
        o->window_out_len = (uint32_t)((S + c->pitch_multiple - 1) / c->pitch_multiple);
    else
        o->window_out_len = (uint32_t)((S - 1) * (uint64_t)(-c->pitch_multiple));
    o->corrected_amp_factor = amp;
    o->pitch_shifted_factor = psf;
    return RC_OK;
}

// windows emitted by an offline run on a closed channel of in_len samples
uint64_t offline_windows(const rc_params &p, size_t in_len) {
    const uint64_t kd = in_len >= p.window_len
                            ? (uint64_t)(in_len - p.window_len) / p.sample_step_len + 1
                            : 0;  // first hop whose window runs past the input
    return kd / p.hops_per_window + 1;
}

// samples [lo, hi) of the input that the hops of windows [w0, w1) read, incl. the recomputed hop before them
void input_span(const rc_params &p, uint64_t w0, uint64_t w1, size_t in_len, size_t *lo, size_t *hi) {
    const uint64_t h0 = w0 * p.hops_per_window, h1 = w1 * p.hops_per_window;
    const uint64_t first = h0 > 0 ? h0 - 1 : 0;
    const uint64_t a = first * p.sample_step_len, b = (h1 - 1) * p.sample_step_len + p.window_len;
    *lo = (size_t)std::min<uint64_t>(a, in_len);
    *hi = (size_t)std::min<uint64_t>(b, in_len);
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return RC_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(RC_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        }
        cap = want;
        return RC_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Channel {
    std::vector<float> fifo;   // host samples from absolute index fifo_base
    uint64_t fifo_base = 0;
    bool closed = false;
    uint64_t total_in = 0;     // samples ever pushed
    uint64_t next_window = 0;  // next window index to COMPUTE
    uint64_t windows_out = 0;  // windows handed to the caller
    int64_t done_window = -1;  // window index whose hand-out sets is_done
    // computed windows not handed out yet: two pinned blocks per channel. The D2H of a batch lands in one at link
    // speed and next_window hands its windows out (a copy, or a pointer: rc_engine_next_window_view); on a closed
    // channel the NEXT batch is already being computed and copied into the other block meanwhile (look-ahead).
    // Windows [ready_pos, ready_n) of block `cur` are pending; `ahead_n` windows are in flight into block cur ^ 1.
    float *blk_p[2] = {nullptr, nullptr};
    size_t blk_cap[2] = {0, 0};  // floats
    hipEvent_t blk_ev[2] = {nullptr, nullptr};
    int cur = 0;
    uint64_t ready_pos = 0, ready_n = 0;
    uint64_t ahead_n = 0;
};

}  // namespace

struct rc_engine {
    rc_config cfg{};
    rc_params par{};
    int log2n = 0;     // 0: window_len is not a power of two (gen)
    bool gen = false;  // window length that is not a power of two: chirp-z transforms (launch_gen)
    float2 *d_tw_gen = nullptr;  // gen: exp(-2 pi i k / N), k < N
    float2 *d_bl_tab = nullptr;  // gen: chirp-z tables (HopParams::bl_tab)
    uint32_t bl_log2l = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    // event pairs around the kernel launches of the last RC_TIMING_RING offline calls (measurement)
    static constexpr int kRing = 64;
    hipEvent_t ev0[kRing] = {}, ev1[kRing] = {};
    uint64_t timed_calls = 0;  // slot of call i is i % kRing
    uint64_t stats_hops = 0;
    uint32_t stats_launches = 0;
    float *d_window = nullptr, *d_env = nullptr;
    float *d_hann_rot = nullptr;  // N >= 16384 with the default window only (HopParams::hann_rot)
    float2 *d_wtab = nullptr, *d_rtab = nullptr;
    float2 *d_t1 = nullptr;  // large windows only: exp(-2 pi i j / (N/2)), j <= N/8
    float2 *d_wtab_m = nullptr;  // large windows, fused kernel: exp(-2 pi i k / (N/2)), k <= N/64
    uint64_t seed_mixed = 0;
    int n_cu = 256;
    std::vector<Channel> ch;
    // scratch
    DevBuf d_in, d_out, d_spec, d_spec2, d_ybuf, d_ysub, d_tail, d_hop_in, d_hop_out, d_xtail;
    // seam hand-over between the runs of hop3_kernel (HopParams::seam_*): stash [runs][H], one flag
    // per run (compared with seam_epoch, so it is zeroed only when it is (re)allocated), run counter
    DevBuf d_seam_head, d_seam_flag, d_run_counter;
    DevBuf d_tail_stage;  // large windows: tails written by big_cr_kernel, copied to d_tail after the launch
    uint32_t seam_epoch = 0;
    // device -> host error word (pinned, mapped): a kernel that gives up (seam wait expired) leaves a
    // code here; the next API call reports it as RC_EHIP
    uint32_t *h_err = nullptr, *d_err = nullptr;
    uint32_t diag_flags = 0;  // ROCODER_DIAG: read only by the test-hook library (RC_TEST_HOOKS), else always 0
    // run-planner overrides (tuning tools; test-hook library only, read once at create): 0 = the defaults
    int tune_rounds = 0, tune_min_run = 0, tune_b4_rounds = 0;
    std::vector<float> h_spec, h_spec2;
    bool tail_zeroed = false;
    bool as_pitch1 = false;  // run_hops is computing O (pitch 1) for the negative-pitch path's resampler
    DevBuf d_obuf;           // ... into this scratch
    // user-kernel path: a stateful apply() forbids recomputing hops, so the overlap tail is carried
    // on the device and ranges must continue where the previous one ended (or restart at hop 0)
    std::vector<int64_t> kernel_next_hop;
    // scratch (tail copy, seam stash, run counter, pipe buffers) is per engine: a call on another
    // stream first waits for the previous call's last enqueue
    hipEvent_t ev_last = nullptr;
    hipStream_t last_stream = nullptr;
    bool last_valid = false;
    // user-kernel path: two chunk-sized resource sets so that the forward transform + D2H of chunk
    // i, the host apply() calls of chunk i - 1 and the H2D + resynthesis + overlap-add of chunk i - 2
    // overlap (streams kf / kb beside the caller's stream; pinned host buffers)
    struct KernelPipe {
        // THREE resource sets: with two, the forward transform + D2H of chunk i + 1 had to wait for the set of chunk
        // i - 1, i.e. for its H2D + resynthesis - the two PCIe directions never ran together. With three, D2H(i + 1)
        // overlaps H2D(i - 1) (separate copy engines) while the host calls apply() on chunk i.
        static constexpr int kSets = 3;
        DevBuf d_spec[kSets], d_ybuf[kSets], d_ysub[kSets];
        float *h_in[kSets] = {}, *h_out[kSets] = {};
        size_t h_cap = 0;
        hipStream_t kf = nullptr, kb = nullptr;
        hipEvent_t ev_fwd[kSets] = {}, ev_back[kSets] = {};
        hipEvent_t ev_in = nullptr, ev_done = nullptr;
    } kp;
    // host-buffer calls (rc_engine_stretch_host): pinned staging slots, one per copy worker (host_copy)
    struct HostPipe {
        static constexpr int kWorkers = 8;
        static constexpr size_t kSlotBytes = (size_t)16 << 20;
        void *slot[kWorkers] = {};
        hipStream_t st[kWorkers] = {};
        hipEvent_t ev_start = nullptr;
        int max_workers = kWorkers;  // (rc_multi: the listed devices' engines copy at the same time and share the host's cores)
        std::vector<hipEvent_t> ev_chunk;  // host_pipeline: one "chunk computed" event per window chunk (pooled)
    } hp;
};

namespace {

uint64_t now_ms(const rc_engine *e) {
    if (e->cfg.kernel_time_ms) return e->cfg.kernel_time_ms;
    using namespace std::chrono;
    return (uint64_t)duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count();
}

// Split hop_count hops per channel into runs so the launch fills the chip (>= ~2 workgroups
// of 256 threads per CU) while keeping the one-hop recompute overhead of each run small.
#ifndef RC_ROUNDS
#define RC_ROUNDS 2
#endif
#ifndef RC_ROUNDS_WAVE
#define RC_ROUNDS_WAVE 2  // the wave-local kernels (tools/ab_rounds.py, ROCODER_AB_N=8192: 1 / 2 / 3 / 4 rounds = 0.741 / 0.733 / 0.751 / 0.752 ms)
#endif
// independent_hops: the launch is one of the two halves of the spectrum paths (MODE_FORWARD / MODE_RESYNTH): no tail
// is carried and no hop recomputed, so a run may be a single hop - a chunk of 128 hop indices then fills the chip
// instead of 32 workgroups walking 8 hops each.
void plan_runs(const rc_engine *e, uint32_t n_channels, int64_t hop_count, uint32_t *runs,
               uint32_t *run_len, bool independent_hops = false) {
    int threads = 64;
    size_t lds = 0;
    rc::hop_geometry(e->log2n, &threads, &lds);
    const size_t lds_cap = 160 * 1024;
    uint32_t wg_per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(lds_cap / std::max<size_t>(lds, 1), 2048 / threads));
    wg_per_cu = std::min<uint32_t>(wg_per_cu, 8);
    uint32_t rounds = RC_ROUNDS;
    if (const int fixed = rc::hop_workgroups_per_cu(e->log2n, e->d_hann_rot != nullptr)) {  // (N = 16384 only)
        wg_per_cu = (uint32_t)fixed;
        // 8 rounds of 768 workgroups = runs of 9 hops at C2. With hop4's per-XCD run tickets the 96 runs resident on
        // an XCD are neighbours, 9 hops = 36 KB apart, so their 64 KB windows overlap and the XCD's working set (3.5 MB)
        // fits its L2: FETCH_SIZE 1.12e6 -> 0.44e6 KiB per launch, L2 hit rate 42 -> 61 %, and in an interleaved A/B
        // (tools/ab_rounds.py) 4 / 6 / 8 / 12 rounds = 1.434 / 1.422 / 1.423 / 1.421 ms; shorter runs (min_run 4) lose
        // to the seam hand-overs again
        rounds = 4 * RC_ROUNDS;
        if (e->tune_rounds > 0) rounds = (uint32_t)e->tune_rounds;
    }
    if (const int res = rc::hop_resident_workgroups(e->log2n, e->d_hann_rot != nullptr)) {
        wg_per_cu = (uint32_t)res;
        rounds = RC_ROUNDS_WAVE;
        if (e->tune_rounds > 0) rounds = (uint32_t)e->tune_rounds;
    }
    // below N = 512 one workgroup (= one wave) of the fused generic kernel holds several runs (hop slots)
    if (!independent_hops && !e->gen && e->log2n >= 5 && e->log2n <= 8) wg_per_cu *= (uint32_t)rc::hop_slots(e->log2n);
#ifdef RC_WG_PER_CU
    wg_per_cu = RC_WG_PER_CU;  // tuning builds
#endif
    const uint64_t target = (uint64_t)e->n_cu * wg_per_cu * rounds;  // rounds of resident workgroups
    uint64_t r = std::max<uint64_t>(1, target / std::max<uint32_t>(1, n_channels));
    const int64_t min_run = independent_hops ? 1 : (e->tune_min_run > 0 ? e->tune_min_run : 8);
    r = std::min<uint64_t>(r, (uint64_t)std::max<int64_t>(1, hop_count / min_run));
    r = std::max<uint64_t>(r, 1);
    uint64_t len = ((uint64_t)hop_count + r - 1) / r;
    r = ((uint64_t)hop_count + len - 1) / len;
    *runs = (uint32_t)r;
    *run_len = (uint32_t)len;
}

// corrected_amp_factor (src/stretcher.rs:52), times |g| under RC_DK_GAIN: |g X| = |g| |X| and everything after
// the magnitude is linear, so the gain rides on the amplitude factor of every overlap-add and the fused
// kernels run unchanged (the README's x2.0 kernel costs nothing)
float amp_of(const rc_engine *e) {
    return e->par.corrected_amp_factor * (e->cfg.device_kernel == RC_DK_GAIN ? fabsf(e->cfg.dk_gain) : 1.0f);
}

rc::HopParams base_params(const rc_engine *e) {
    rc::HopParams p{};
    p.window = e->d_window;
    p.env = e->d_env;
    p.hann_rot = e->d_hann_rot;
    p.wtab = e->d_wtab;
    p.rtab = e->d_rtab;
    p.amp = amp_of(e);
    p.step = e->par.sample_step_len;
    p.pitch = (uint32_t)std::max(1, e->cfg.pitch_multiple);
    p.seed_mixed = e->seed_mixed;
    p.n_generic = e->gen ? e->par.window_len : 0;
    p.tw_generic = e->d_tw_gen;
    p.bl_log2l = e->bl_log2l;
    p.bl_tab = e->d_bl_tab;
    p.err_word = e->d_err;
    p.diag_flags = e->diag_flags;
    // 2^22 polls of ~2 000 cycles each (seconds); the diagnostic build of the protocol gives up at once
    p.seam_spin_limit = (e->diag_flags & rc::RC_DIAG_SKIP_SEAM_PUBLISH) ? 64u : (1u << 22);
    return p;
}

// the large-window kernels take the same description of the input as the hop kernels
rc::BigParams big_params(const rc_engine *e, const rc::HopParams &p) {
    rc::BigParams b{};
    b.x = p.x;
    b.in_stride = p.in_stride;
    b.in_origin = p.in_origin;
    b.xtail = p.xtail;
    b.tail_stride = p.tail_stride;
    b.tail_origin = p.tail_origin;
    b.tail_hop_first = p.tail_hop_first;
    b.window = e->d_window;
    b.wtab_sub = e->d_wtab;
    b.t1 = e->d_t1;
    b.rtab = e->d_rtab;
    b.step = p.step;
    b.seed_mixed = p.seed_mixed;
    b.ch_first = p.ch_first;
    b.n_channels = p.n_channels;
    b.hop_first = p.hop_first;
    b.hop_count = p.hop_count;
    b.log2n = (uint32_t)e->log2n;
    return b;
}


// A kernel that could not finish its work (HopParams::err_word) fails the NEXT call on the handle, once.
int check_device_error(rc_engine *e) {
    if (!e->h_err) return RC_OK;
    const uint32_t code = __atomic_exchange_n(e->h_err, 0u, __ATOMIC_ACQ_REL);
    if (code == rc::RC_ERR_SEAM_TIMEOUT)
        return fail(RC_EHIP, "an earlier launch on this engine gave up waiting for a run seam hand-over: "
                             "its output is incomplete");
    if (code) return fail(RC_EHIP, "an earlier launch on this engine reported device error %u", code);
    return RC_OK;
}

// User frequency kernel (src/fft.rs:76-108), hops [hop_first, hop_first + hop_count): forward transform
// -> D2H -> apply() per hop on this host thread, in the reference's order (windows outer, channels
// inner, hops of a window innermost: src/stretcher_processor.rs:63-70) -> H2D -> resynthesis ->
// gather-form overlap-add with the tail carried in d_tail. apply() may be stateful, so no hop is
// ever recomputed. Chunks of <= 32 MiB of spectrum run as a three-stage pipeline over two resource
// sets: while the host works through chunk i - 1, stream kf produces chunk i and stream kb finishes
// chunk i - 2.
int run_hops_kernel(rc_engine *e, const rc::HopParams &p, uint32_t ch_first, uint32_t n_channels,
                    int64_t hop_first, int64_t hop_count, float *d_out, size_t out_stride,
                    int64_t out_origin, hipStream_t s, uint32_t *launches, bool *started) {
    rc_engine::KernelPipe &kp = e->kp;
    const uint32_t N = e->par.window_len, H = N / 2;
    const bool big = e->log2n > 14;
    const int64_t hpw = e->par.hops_per_window;
    if (!kp.kf) {
        RC_HIP(hipStreamCreateWithFlags(&kp.kf, hipStreamNonBlocking));
        RC_HIP(hipStreamCreateWithFlags(&kp.kb, hipStreamNonBlocking));
        for (int i = 0; i < rc_engine::KernelPipe::kSets; ++i) {
            RC_HIP(hipEventCreateWithFlags(&kp.ev_fwd[i], hipEventDisableTiming));
            RC_HIP(hipEventCreateWithFlags(&kp.ev_back[i], hipEventDisableTiming));
        }
        RC_HIP(hipEventCreateWithFlags(&kp.ev_in, hipEventDisableTiming));
        RC_HIP(hipEventCreateWithFlags(&kp.ev_done, hipEventDisableTiming));
    }
    const size_t hop_bytes = (size_t)N * 2 * sizeof(float) * n_channels;  // one hop index, all channels
#ifndef RC_KCHUNK_MB
#define RC_KCHUNK_MB 8  // 32 / 16 / 8 / 4 / 2 MiB measured at C4 (two kernel threads, device-resident job): 28.8 / 27.7 / 24.6 / 28.4 / 35 ms
#endif
    int64_t kc_max = (int64_t)(((size_t)RC_KCHUNK_MB << 20) / hop_bytes) / hpw * hpw;
    kc_max = std::max<int64_t>(hpw, std::min<int64_t>(kc_max, (hop_count + hpw - 1) / hpw * hpw));
    const size_t chunk_bytes = (size_t)kc_max * hop_bytes;
    constexpr int kSets = rc_engine::KernelPipe::kSets;
    for (int i = 0; i < kSets; ++i) {
        if (int rc = kp.d_spec[i].reserve(chunk_bytes)) return rc;
        if (int rc = kp.d_ybuf[i].reserve(chunk_bytes / 2)) return rc;
        if (big)
            if (int rc = kp.d_ysub[i].reserve(chunk_bytes / 2)) return rc;
        if (e->bl_log2l > 14)  // chirp-z work buffer: L points per hop
            if (int rc = kp.d_ysub[i].reserve(((size_t)kc_max * n_channels * sizeof(float2)) << e->bl_log2l)) return rc;
    }
    if (kp.h_cap < chunk_bytes) {
        for (int i = 0; i < kSets; ++i) {
            if (kp.h_in[i]) (void)hipHostFree(kp.h_in[i]);
            if (kp.h_out[i]) (void)hipHostFree(kp.h_out[i]);
            kp.h_in[i] = kp.h_out[i] = nullptr;
        }
        kp.h_cap = 0;
        for (int i = 0; i < kSets; ++i) {
            RC_HIP(hipHostMalloc((void **)&kp.h_in[i], chunk_bytes, hipHostMallocDefault));
            RC_HIP(hipHostMalloc((void **)&kp.h_out[i], chunk_bytes, hipHostMallocDefault));
        }
        kp.h_cap = chunk_bytes;
    }
    // kf and kb start after everything already queued on the caller's stream
    RC_HIP(hipEventRecord(kp.ev_in, s));
    RC_HIP(hipStreamWaitEvent(kp.kf, kp.ev_in, 0));
    RC_HIP(hipStreamWaitEvent(kp.kb, kp.ev_in, 0));

    auto params_of = [&](int set, int64_t k0, int64_t kc, rc::BigParams *b) {
        rc::HopParams q = p;
        q.spec = (float2 *)kp.d_spec[set].p;
        q.ybuf = (float *)kp.d_ybuf[set].p;
        q.ch_first = ch_first;
        q.n_channels = n_channels;
        q.hop_first = k0;
        q.hop_count = kc;
        if (e->bl_log2l > 14) q.bl_wk = (float2 *)kp.d_ysub[set].p;
        if (big) {
            *b = big_params(e, q);
            b->ysub = (float2 *)kp.d_ysub[set].p;
            b->ybuf = q.ybuf;
            b->spec = q.spec;
        } else {
            plan_runs(e, n_channels, kc, &q.runs_per_channel, &q.run_len, true);
        }
        return q;
    };
    auto front = [&](int i, int64_t k0, int64_t kc) -> int {  // forward transform + D2H on kf
        const int set = i % kSets;
        if (i >= kSets) RC_HIP(hipStreamWaitEvent(kp.kf, kp.ev_back[set], 0));  // set free again
        rc::BigParams b{};
        const rc::HopParams q = params_of(set, k0, kc, &b);
        if (big) {
            RC_HIP(rc::launch_big(0, b, kp.kf));
            RC_HIP(rc::launch_big(1, b, kp.kf, rc::MODE_FORWARD));
            *launches += 2;
        } else if (e->gen) {
            RC_HIP(rc::launch_gen(0, q, kp.kf));
            *launches += 1;
        } else {
            RC_HIP(rc::launch_hop(e->log2n, rc::MODE_FORWARD, q, kp.kf));
            *launches += 1;
        }
        RC_HIP(hipMemcpyAsync(kp.h_in[set], kp.d_spec[set].p, (size_t)kc * hop_bytes,
                              hipMemcpyDeviceToHost, kp.kf));
        RC_HIP(hipEventRecord(kp.ev_fwd[set], kp.kf));
        return RC_OK;
    };
    auto back = [&](int i, int64_t k0, int64_t kc) -> int {  // apply() here, then H2D .. OLA on kb
        const int set = i % kSets;
        RC_HIP(hipEventSynchronize(kp.ev_fwd[set]));
        // channels [c0, c1): windows outer, channels inner, hops of a window innermost
        auto apply_range = [&](uint32_t c0, uint32_t c1) {
            for (int64_t w0 = 0; w0 < kc; w0 += hpw) {
                for (uint32_t c = c0; c < c1; ++c) {
                    for (int64_t h = w0; h < std::min<int64_t>(w0 + hpw, kc); ++h) {
                        const size_t off = ((size_t)c * kc + h) * N * 2;
                        // src/fft.rs:86-99: apply(now_ms, bins) -> bins
                        const int krc = e->cfg.kernel(now_ms(e), kp.h_in[set] + off, kp.h_out[set] + off, N,
                                                      e->cfg.kernel_user);
                        // non-zero == panic: keep the unmodified spectrum (src/fft.rs:100-106)
                        if (krc != 0) memcpy(kp.h_out[set] + off, kp.h_in[set] + off, (size_t)N * 2 * sizeof(float));
                    }
                }
            }
        };
        const uint32_t nthr = std::min<uint32_t>(std::max<uint32_t>(1, e->cfg.kernel_threads), n_channels);
        if (nthr <= 1) {
            apply_range(0, n_channels);  // the reference's single DSP thread and call order
        } else {  // rc_config::kernel_threads: channels dealt to threads, per-channel hop order kept
            JoinedThreads pool;
            for (uint32_t t = 1; t < nthr; ++t) {
                const uint32_t c0 = n_channels * t / nthr, c1 = n_channels * (t + 1) / nthr;
                if (!pool.start(apply_range, c0, c1)) apply_range(c0, c1);  // (no thread to be had: this one does it)
            }
            apply_range(0, n_channels / nthr);
            pool.join();
        }
        RC_HIP(hipMemcpyAsync(kp.d_spec[set].p, kp.h_out[set], (size_t)kc * hop_bytes,
                              hipMemcpyHostToDevice, kp.kb));
        rc::BigParams b{};
        const rc::HopParams q = params_of(set, k0, kc, &b);
        if (big) {
            RC_HIP(rc::launch_big(1, b, kp.kb, rc::MODE_RESYNTH));
            RC_HIP(rc::launch_big(2, b, kp.kb));
            *launches += 2;
        } else if (e->gen) {
            RC_HIP(rc::launch_gen(1, q, kp.kb));
            RC_HIP(rc::launch_gen(2, q, kp.kb));
            *launches += 2;
        } else {
            RC_HIP(rc::launch_hop(e->log2n, rc::MODE_RESYNTH, q, kp.kb));
            *launches += 1;
        }
        rc::OlaParams o{};
        o.ybuf = (const float *)kp.d_ybuf[set].p;
        o.tail = (float *)e->d_tail.p + (size_t)ch_first * H;
        o.out = d_out;
        o.out_stride = out_stride;
        o.out_origin = out_origin;
        o.env = e->d_env;
        o.amp = amp_of(e);
        o.pitch = e->cfg.pitch_multiple;
        o.samples_needed = (uint32_t)e->par.samples_needed_per_window;
        o.window_out_len = e->par.window_out_len;
        o.n_channels = n_channels;
        o.hop_first = k0;
        o.hop_count = kc;
        o.log2n = (uint32_t)e->log2n;
        o.n = N;
        RC_HIP(rc::launch_ola(o, kp.kb, false));
        *launches += 2;
        RC_HIP(hipEventRecord(kp.ev_back[set], kp.kb));
        return RC_OK;
    };
    int i = 0;
    int64_t prev_k0 = 0, prev_kc = 0;
    *started = true;  // from here on chunks advance d_tail and consume apply() calls: a failure cannot be retried in place
    for (int64_t k0 = hop_first; k0 < hop_first + hop_count; k0 += kc_max, ++i) {
        const int64_t kc = std::min<int64_t>(kc_max, hop_first + hop_count - k0);
        if (int rc = front(i, k0, kc)) return rc;
        if (i > 0)
            if (int rc = back(i - 1, prev_k0, prev_kc)) return rc;
        prev_k0 = k0;
        prev_kc = kc;
    }
    if (i > 0)
        if (int rc = back(i - 1, prev_k0, prev_kc)) return rc;
    // the caller's stream continues after the last overlap-add
    RC_HIP(hipEventRecord(kp.ev_done, kp.kb));
    RC_HIP(hipStreamWaitEvent(s, kp.ev_done, 0));
    return RC_OK;
}

#ifdef RC_STAMP_DUMP
// diagnostic builds only: per-wave phase cycle totals of the last launch -> $ROCODER_STAMPS (text)
static int dump_stamps(rc_engine *e, size_t n_dbg, hipStream_t s, int64_t hop_count, uint32_t run_len) {
    const char *path = getenv("ROCODER_STAMPS");
    if (!path) return RC_OK;
    std::vector<unsigned> h(n_dbg);
    RC_HIP(hipStreamSynchronize(s));
    RC_HIP(hipMemcpy(h.data(), e->d_spec.p, n_dbg * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (FILE *f = fopen(path, "w")) {
        double sum[32] = {0};
        size_t nw = 0;
        for (size_t w = 0; w < n_dbg / 32; ++w) {
            bool any = false;
            for (int i = 0; i < 32; ++i) any |= h[w * 32 + i] != 0;
            if (!any) continue;
            nw++;
            for (int i = 0; i < 32; ++i) sum[i] += h[w * 32 + i];
        }
        fprintf(f, "waves %zu hops %lld run_len %u\n", nw, (long long)hop_count, run_len);
        for (int i = 0; i < 32; ++i) fprintf(f, "phase %2d mean_cycles_per_wave %.0f\n", i, nw ? sum[i] / nw : 0.0);
        for (size_t w = 0; w < 16 && w < n_dbg / 32; ++w) {  // the waves of the first workgroup(s), one per line
            fprintf(f, "wave %2zu:", w);
            for (int i = 0; i < 16; ++i) fprintf(f, " %u", h[w * 32 + i]);
            fprintf(f, "\n");
        }
        fclose(f);
    }
    return RC_OK;
}
#endif

// The core: compute hops [hop_first, hop_first+hop_count) of n_channels channels whose samples
// live on the device, writing the decimated overlap-add at d_out (absolute F index out_origin).
int run_hops(rc_engine *e, const float *d_in, size_t in_stride, int64_t in_origin, int64_t in_len,
             uint32_t ch_first, uint32_t n_channels, int64_t hop_first, int64_t hop_count,
             float *d_out, size_t out_stride, int64_t out_origin, hipStream_t s, bool timed) {
    if (hop_count <= 0 || n_channels == 0) return RC_OK;
    if (e->cfg.kernel) {
        for (uint32_t c = ch_first; c < ch_first + n_channels; ++c)
            if (hop_first != 0 && hop_first != e->kernel_next_hop[c])
                return fail(RC_EINVAL, "a user kernel carries the overlap tail: channel %u continues at hop %lld "
                                       "(or restarts at 0), not at %lld", c, (long long)e->kernel_next_hop[c],
                            (long long)hop_first);
    }
    if (e->last_valid && e->last_stream != s) RC_HIP(hipStreamWaitEvent(s, e->ev_last, 0));
    struct MarkLast {  // whatever path returns: the next call orders itself behind this one
        rc_engine *e;
        hipStream_t s;
        ~MarkLast() {
            if (hipEventRecord(e->ev_last, s) == hipSuccess) {
                e->last_stream = s;
                e->last_valid = true;
            }
        }
    } mark_last{e, s};
    rc::HopParams p = base_params(e);
    p.x = d_in;
    p.in_stride = in_stride;
    p.in_origin = in_origin;
    p.in_len = in_len;
    p.out = d_out;
    p.out_stride = out_stride;
    p.out_origin = out_origin;
    p.ch_first = ch_first;
    p.n_channels = n_channels;
    const uint32_t N = e->par.window_len, H = N / 2;
    rc::PrepParams prep{};  // ONE small launch in front of the job: the padded tail + the seam's run counter
    {
        // Hops whose window runs past the end of the input read a zero-padded copy of the input
        // tail (src/stretcher.rs:129-132: resize(n, 0.0)); the kernels never bounds-check.
        const int64_t end_abs = in_origin + in_len;
        const int64_t step = e->par.sample_step_len;
        const int64_t k_last = hop_first + hop_count - 1;
        int64_t k_t = end_abs >= (int64_t)N ? (end_abs - N) / step + 1 : 0;  // first short hop
        const bool carries_tail = e->cfg.kernel != nullptr;  // a stateful apply(): no hop is recomputed
        const int64_t k_lo = (!carries_tail && hop_first > 0) ? hop_first - 1 : hop_first;
        if (k_t < k_lo) k_t = k_lo;
        p.xtail = d_in;
        p.tail_stride = 0;
        p.tail_origin = 0;
        p.tail_hop_first = INT64_MAX;
        if (k_t <= k_last) {
            const int64_t t0 = k_t * step;
            const size_t tail_len = (size_t)((k_last - k_t) * step + N);
            const int64_t real = std::max<int64_t>(0, std::min<int64_t>(end_abs - t0, (int64_t)tail_len));
            int rc = e->d_xtail.reserve((size_t)n_channels * tail_len * sizeof(float));
            if (rc) return rc;
            prep.xtail = (float *)e->d_xtail.p;
            prep.tail_len = tail_len;
            prep.src = d_in + (t0 - in_origin);
            prep.src_stride = in_stride;
            prep.real = (size_t)real;
            prep.n_channels = n_channels;
            p.xtail = (const float *)e->d_xtail.p;
            p.tail_stride = tail_len;
            p.tail_origin = t0;
            p.tail_hop_first = k_t;
        }
    }
    // RC_DK_BAND / RC_DK_SHIFT act on the spectrum between analysis and resynthesis: unfused, but on the device —
    // except the band mask on the default 16384-sample window, which hop4_kernel applies in its pair stage
    // (a per-bin gain on the magnitudes: the fused kernel's speed instead of three kernels through HBM)
    const bool band_fused = e->cfg.device_kernel == RC_DK_BAND && e->cfg.pitch_multiple >= 1 &&
                            rc::hop_workgroups_per_cu(e->log2n, e->d_hann_rot != nullptr) != 0 &&
                            !(e->diag_flags & rc::RC_DIAG_PREV_KERNEL);
    const bool devk = (e->cfg.device_kernel == RC_DK_BAND && !band_fused) || e->cfg.device_kernel == RC_DK_SHIFT;
    const bool pos_pitch = e->cfg.pitch_multiple >= 1 || e->as_pitch1;
    const bool fused = !e->gen && !e->cfg.kernel && !devk && e->log2n <= 14 && pos_pitch;
#ifndef RC_NEGFUSED
#define RC_NEGFUSED 1  // negative pitch multiples: the fused kernels at pitch 1 into scratch + resample_slower_kernel (0: unfused, for A/B)
#endif
    if (RC_NEGFUSED && e->cfg.pitch_multiple < 0 && !e->as_pitch1 && !e->gen && !e->cfg.kernel && !devk &&
        !(e->diag_flags & rc::RC_DIAG_PREV_KERNEL)) {
        // pitch <= -2 (one hop per window, src/stretcher.rs:47-49): O = the pitch-1 overlap-add of the hop range, computed by
        // the fused kernels into scratch in pieces of <= 256 MiB, then each window = resample_slower(O_k[0..S))
        const int64_t piece = std::max<int64_t>(1, (int64_t)(((size_t)256 << 20) / ((size_t)n_channels * H * sizeof(float))));
        for (int64_t k0 = hop_first; k0 < hop_first + hop_count; k0 += piece) {
            const int64_t kc = std::min<int64_t>(piece, hop_first + hop_count - k0);
            if (int rcb = e->d_obuf.reserve((size_t)n_channels * kc * H * sizeof(float))) return rcb;
            int rcr;
            {
                struct AsPitch1 {  // cleared on every way out of the inner pass, exceptions included
                    rc_engine *e;
                    explicit AsPitch1(rc_engine *e_) : e(e_) { e->as_pitch1 = true; }
                    ~AsPitch1() { e->as_pitch1 = false; }
                } guard(e);
                rcr = run_hops(e, d_in, in_stride, in_origin, in_len, ch_first, n_channels, k0, kc, (float *)e->d_obuf.p,
                               (size_t)kc * H, k0 * (int64_t)H, s, timed && k0 == hop_first);
            }
            if (rcr) return rcr;
            rc::ResampleParams r{};
            r.obuf = (const float *)e->d_obuf.p;
            r.o_stride = (size_t)kc * H;
            r.out = d_out;
            r.out_stride = out_stride;
            r.out_origin = out_origin;
            r.hop_first = k0;
            r.hop_count = kc;
            r.n_channels = n_channels;
            r.half = H;
            r.samples_needed = (uint32_t)e->par.samples_needed_per_window;
            r.window_out_len = e->par.window_out_len;
            r.f = (uint32_t)(-e->cfg.pitch_multiple);
            RC_HIP(rc::launch_resample_slower(r, s));
        }
        return RC_OK;
    }
    if (band_fused) {
        const uint32_t half = e->par.window_len / 2;
        const uint32_t lo = e->cfg.dk_lo_bin, hi = std::min<uint32_t>(e->cfg.dk_hi_bin, half);
        p.band_on = 1;
        p.band_lo = lo;
        p.band_span = hi >= lo ? hi - lo : 0;
        p.band_gout = fabsf(e->cfg.dk_gain_outside);
        p.band_gin = hi >= lo ? fabsf(e->cfg.dk_gain) : p.band_gout;  // (an empty band: everything is outside)
    }
    if (fused) {
        p.hop_first = hop_first;
        p.hop_count = hop_count;
        plan_runs(e, n_channels, hop_count, &p.runs_per_channel, &p.run_len);
#ifdef RC_STAMP_DUMP
        // diagnostic builds only: per-wave phase cycle totals -> $ROCODER_STAMPS (text)
        const size_t n_dbg = (size_t)p.runs_per_channel * n_channels * 8 * 32;
        if (int rcd = e->d_spec.reserve(n_dbg * sizeof(unsigned))) return rcd;
        RC_HIP(hipMemsetAsync(e->d_spec.p, 0, n_dbg * sizeof(unsigned), s));
        p.spec = (float2 *)e->d_spec.p;
#endif
#ifndef RC_SEAM
#define RC_SEAM 1
#endif
        if (RC_SEAM && rc::hop_workgroups_per_cu(e->log2n, e->d_hann_rot != nullptr) && p.runs_per_channel > 1) {
            // (no memory for the stash is not an error: the kernel then recomputes one hop per run)
            const size_t runs_total = (size_t)p.runs_per_channel * n_channels;
            const size_t flag_cap = e->d_seam_flag.cap;
            if (e->d_seam_head.reserve(runs_total * H * sizeof(float)) == RC_OK &&
                e->d_seam_flag.reserve(runs_total * sizeof(uint32_t)) == RC_OK &&
                e->d_run_counter.reserve(rc::RC_RUN_COUNTERS * sizeof(uint32_t)) == RC_OK) {
                if (e->d_seam_flag.cap != flag_cap) {  // fresh allocation: no flag equals any epoch yet
                    RC_HIP(hipMemsetAsync(e->d_seam_flag.p, 0, e->d_seam_flag.cap, s));
                    e->seam_epoch = 0;
                }
                prep.run_counter = (uint32_t *)e->d_run_counter.p;
                if (++e->seam_epoch == 0) {  // wrapped: start over with clean flags
                    RC_HIP(hipMemsetAsync(e->d_seam_flag.p, 0, e->d_seam_flag.cap, s));
                    e->seam_epoch = 1;
                }
                p.seam_head = (float *)e->d_seam_head.p;
                p.seam_flag = (uint32_t *)e->d_seam_flag.p;
                p.run_counter = (uint32_t *)e->d_run_counter.p;
                p.seam_epoch = e->seam_epoch;
            }
        }
        RC_HIP(rc::launch_prep(prep, s));
        if (timed) RC_HIP(hipEventRecord(e->ev0[e->timed_calls % rc_engine::kRing], s));
        RC_HIP(rc::launch_hop(e->log2n, rc::MODE_FUSED, p, s));
#ifdef RC_STAMP_DUMP
        if (int rcd = dump_stamps(e, n_dbg, s, hop_count, p.run_len)) return rcd;
#endif
        if (timed) {
            RC_HIP(hipEventRecord(e->ev1[e->timed_calls % rc_engine::kRing], s));
            e->timed_calls++;
            e->stats_hops = (uint64_t)hop_count * n_channels;
            e->stats_launches = 1;
        }
        return RC_OK;
    }
#ifndef RC_BIG4
#define RC_BIG4 1
#endif
    if (RC_BIG4 && e->log2n > 14 && !e->cfg.kernel && !devk && pos_pitch &&
        !(e->diag_flags & rc::RC_DIAG_PREV_KERNEL)) {
        // fused large-window kernel: one workgroup (512 threads, ~150 KB of LDS: one per CU) per run of hops;
        // each run recomputes the hop before it for its tail
        p.hop_first = hop_first;
        p.hop_count = hop_count;
        p.wtab = e->d_wtab_m;
        // one workgroup per CU and launch (measured on C5: 1 / 2 / 3 / 4 / 6 rounds of workgroups = 6.45 / 6.50 / 6.61 /
        // 6.60 / 6.72 ms: every run recomputes one hop and reloads its tables); ROCODER_B4_ROUNDS overrides it in the test-hook build
        const int b4_rounds = std::max(1, e->tune_b4_rounds);
        uint64_t r = std::max<uint64_t>(1, (uint64_t)e->n_cu * b4_rounds / n_channels);
        r = std::max<uint64_t>(1, std::min<uint64_t>(r, (uint64_t)hop_count / 8));
        const uint64_t len = ((uint64_t)hop_count + r - 1) / r;
        p.run_len = (uint32_t)len;
        p.runs_per_channel = (uint32_t)(((uint64_t)hop_count + len - 1) / len);
        if (const size_t tsf = rc::big4_tail_scratch_floats(e->log2n)) {  // (builds with the tail outside the registers)
            if (int rcs = e->d_ybuf.reserve((size_t)p.runs_per_channel * n_channels * tsf * sizeof(float))) return rcs;
            p.ybuf = (float *)e->d_ybuf.p;
        }
#ifdef RC_STAMP_DUMP
        const size_t n_dbg = (size_t)p.runs_per_channel * n_channels * 8 * 32;
        if (int rcd = e->d_spec.reserve(n_dbg * sizeof(unsigned))) return rcd;
        RC_HIP(hipMemsetAsync(e->d_spec.p, 0, n_dbg * sizeof(unsigned), s));
        p.spec = (float2 *)e->d_spec.p;
#endif
        RC_HIP(rc::launch_prep(prep, s));
        if (timed) RC_HIP(hipEventRecord(e->ev0[e->timed_calls % rc_engine::kRing], s));
        RC_HIP(rc::launch_big4(e->log2n, p, s));
#ifdef RC_STAMP_DUMP
        if (int rcd = dump_stamps(e, n_dbg, s, hop_count, p.run_len)) return rcd;
#endif
        if (timed) {
            RC_HIP(hipEventRecord(e->ev1[e->timed_calls % rc_engine::kRing], s));
            e->timed_calls++;
            e->stats_hops = (uint64_t)hop_count * n_channels;
            e->stats_launches = 1;
        }
        return RC_OK;
    }
    // ---- unfused pipeline: y_k for a chunk of hops lands in HBM scratch, then a gather-form
    // overlap-add kernel writes the output. Used for
    //   * a user frequency kernel: forward -> host apply() per hop -> resynth. apply() may be
    //     stateful, so no hop is ever recomputed: the overlap tail is carried in d_tail across
    //     chunks and calls, and hops reach apply() in the reference's order (windows outer, channels
    //     inner: src/stretcher_processor.rs:63-70; hops innermost);
    //   * windows of 32768 / 65536 samples: four quarter FFTs through scratch (rc::launch_big);
    //   * negative pitch multiples: linear-interpolating overlap-add (src/resampler.rs:20-35).
    // Without a user kernel the hop before the range is recomputed to seed the tail, so ranges and
    // streaming batches do not depend on call history.
    RC_HIP(rc::launch_prep(prep, s));
    const bool big = e->log2n > 14;
    const uint32_t hpw = e->par.hops_per_window;
    const size_t per_hop = (size_t)N * (devk ? 28 : 12) + (e->bl_log2l > 14 ? sizeof(float2) << e->bl_log2l : 0);  // spectrum(s) / quarter-FFT scratch, y
    int64_t chunk_max = (int64_t)(((size_t)1024 << 20) / (per_hop * n_channels));
    chunk_max = std::max<int64_t>(hpw, std::min<int64_t>(chunk_max / hpw * hpw, 32768));
    int rc = e->d_tail.reserve((size_t)e->cfg.channels * H * sizeof(float));
    if (rc) return rc;
    if (!e->tail_zeroed) {
        RC_HIP(hipMemsetAsync(e->d_tail.p, 0, (size_t)e->cfg.channels * H * sizeof(float), s));
        e->tail_zeroed = true;
    }
    uint32_t launches = 0;
    auto run_chunk = [&](int64_t k0, int64_t kc, bool tail_only) -> int {
        int rcc;
        const size_t spec_floats = (size_t)n_channels * kc * N * 2;
        if ((!big || devk || e->gen) && (rcc = e->d_spec.reserve(spec_floats * sizeof(float)))) return rcc;
        if (e->cfg.device_kernel == RC_DK_SHIFT && (rcc = e->d_spec2.reserve(spec_floats * sizeof(float)))) return rcc;
        if ((rcc = e->d_ybuf.reserve((size_t)n_channels * kc * N * sizeof(float)))) return rcc;
        rc::HopParams q = p;
        q.spec = (float2 *)e->d_spec.p;
        q.ybuf = (float *)e->d_ybuf.p;
        if (e->bl_log2l > 14) {  // chirp-z transforms longer than one workgroup's LDS: work buffer of L points per hop
            if ((rcc = e->d_ysub.reserve(((size_t)n_channels * kc * sizeof(float2)) << e->bl_log2l))) return rcc;
            q.bl_wk = (float2 *)e->d_ysub.p;
        }
        q.ch_first = ch_first;
        q.n_channels = n_channels;
        q.hop_first = k0;
        q.hop_count = kc;
        rc::BigParams b{};
        if (big) {
            if ((rcc = e->d_ysub.reserve((size_t)n_channels * kc * N * sizeof(float)))) return rcc;
            b = big_params(e, q);
            b.ysub = (float2 *)e->d_ysub.p;
            b.ybuf = q.ybuf;
            b.spec = q.spec;
        } else if (!e->gen) {
            plan_runs(e, n_channels, kc, &q.runs_per_channel, &q.run_len, true);  // (FORWARD / RESYNTH halves)
        }
#ifndef RC_BIGCR
#define RC_BIGCR 1
#endif
        if (RC_BIGCR && big && !devk && e->cfg.pitch_multiple >= 1) {
            // stage C with the overlap-add fused: runs of hops per quarter, tail in registers; the
            // tail of the chunk's last hop goes to a staging buffer and replaces d_tail afterwards
            // (the first run of this very launch still reads the old one)
            if ((rcc = e->d_tail_stage.reserve((size_t)e->cfg.channels * H * sizeof(float)))) return rcc;
            RC_HIP(rc::launch_big(0, b, s));
            RC_HIP(rc::launch_big(1, b, s));
            rc::BigOlaParams c{};
            c.b = b;
            c.out = d_out;
            c.out_stride = out_stride;
            c.out_origin = out_origin;
            c.env = e->d_env;
            c.amp = amp_of(e);
            c.pitch = (uint32_t)e->cfg.pitch_multiple;
            c.tail_in = (const float *)e->d_tail.p + (size_t)ch_first * H;
            c.tail_out = (float *)e->d_tail_stage.p + (size_t)ch_first * H;
#ifndef RC_BIGRUN
#define RC_BIGRUN 12  // measured on BASELINE C5: 12 hops per run (8..16 within 5 %)
#endif
            c.run_len = RC_BIGRUN;
            c.runs = (uint32_t)((kc + c.run_len - 1) / c.run_len);
            c.tail_only = tail_only ? 1u : 0u;
            RC_HIP(rc::launch_big_cr(c, s));
            RC_HIP(hipMemcpyAsync((float *)e->d_tail.p + (size_t)ch_first * H, c.tail_out,
                                  (size_t)n_channels * H * sizeof(float), hipMemcpyDeviceToDevice, s));
            launches += 3;
            return RC_OK;
        } else if (e->gen) {
            // window length that is not a power of two: chirp-z transforms, optional device kernel in between
            RC_HIP(rc::launch_gen(0, q, s));
            if (devk) {
                rc::DevKernelParams d{};
                d.in = (const float2 *)e->d_spec.p;
                d.out = e->cfg.device_kernel == RC_DK_SHIFT ? (float2 *)e->d_spec2.p : (float2 *)e->d_spec.p;
                d.n = N;
                d.kind = e->cfg.device_kernel;
                d.gain_in = e->cfg.dk_gain;
                d.gain_out = e->cfg.dk_gain_outside;
                d.lo_bin = e->cfg.dk_lo_bin;
                d.hi_bin = e->cfg.dk_hi_bin;
                d.shift = e->cfg.dk_shift_bins;
                d.hops_total = (uint64_t)n_channels * (uint64_t)kc;
                RC_HIP(rc::launch_dev_kernel(d, s));
                q.spec = d.out;
            }
            RC_HIP(rc::launch_gen(1, q, s));
            RC_HIP(rc::launch_gen(2, q, s));
            launches += 3;
        } else if (devk) {
            // analysis -> curated device kernel on the N-bin spectra -> resynthesis; nothing leaves the GPU
            if (big) {
                RC_HIP(rc::launch_big(0, b, s));
                RC_HIP(rc::launch_big(1, b, s, rc::MODE_FORWARD));
            } else {
                RC_HIP(rc::launch_hop(e->log2n, rc::MODE_FORWARD, q, s));
            }
            rc::DevKernelParams d{};
            d.in = (const float2 *)e->d_spec.p;
            d.out = e->cfg.device_kernel == RC_DK_SHIFT ? (float2 *)e->d_spec2.p : (float2 *)e->d_spec.p;
            d.log2n = (uint32_t)e->log2n;
            d.n = N;
            d.kind = e->cfg.device_kernel;
            d.gain_in = e->cfg.dk_gain;
            d.gain_out = e->cfg.dk_gain_outside;
            d.lo_bin = e->cfg.dk_lo_bin;
            d.hi_bin = e->cfg.dk_hi_bin;
            d.shift = e->cfg.dk_shift_bins;
            d.hops_total = (uint64_t)n_channels * (uint64_t)kc;
            RC_HIP(rc::launch_dev_kernel(d, s));
            q.spec = d.out;
            b.spec = d.out;
            if (big) {
                RC_HIP(rc::launch_big(1, b, s, rc::MODE_RESYNTH));
                RC_HIP(rc::launch_big(2, b, s));
                launches += 5;
            } else {
                RC_HIP(rc::launch_hop(e->log2n, rc::MODE_RESYNTH, q, s));
                launches += 3;
            }
        } else if (big) {
            for (int stage = 0; stage < 3; ++stage) RC_HIP(rc::launch_big(stage, b, s));
            launches += 3;
        } else {  // negative pitch multiples, N <= 16384: spectrum -> y_k through scratch
            RC_HIP(rc::launch_hop(e->log2n, rc::MODE_FORWARD, q, s));
            RC_HIP(rc::launch_hop(e->log2n, rc::MODE_RESYNTH, q, s));
            launches += 2;
        }
        rc::OlaParams o{};
        o.ybuf = (const float *)e->d_ybuf.p;
        o.tail = (float *)e->d_tail.p + (size_t)ch_first * H;
        o.out = d_out;
        o.out_stride = out_stride;
        o.out_origin = out_origin;
        o.env = e->d_env;
        o.amp = amp_of(e);
        o.pitch = e->cfg.pitch_multiple;
        o.samples_needed = (uint32_t)e->par.samples_needed_per_window;
        o.window_out_len = e->par.window_out_len;
        o.n_channels = n_channels;
        o.hop_first = k0;
        o.hop_count = kc;
        o.log2n = (uint32_t)e->log2n;
        o.n = N;
        RC_HIP(rc::launch_ola(o, s, tail_only));
        launches += 2;
        return RC_OK;
    };
    if (timed) RC_HIP(hipEventRecord(e->ev0[e->timed_calls % rc_engine::kRing], s));
    if (hop_first == 0)  // a fresh stream starts from H zeros (src/stretcher.rs:58-59)
        RC_HIP(hipMemsetAsync((float *)e->d_tail.p + (size_t)ch_first * H, 0,
                              (size_t)n_channels * H * sizeof(float), s));
    else if (!e->cfg.kernel && (rc = run_chunk(hop_first - 1, 1, true)))
        return rc;
    if (e->cfg.kernel) {
        bool started = false;
        if ((rc = run_hops_kernel(e, p, ch_first, n_channels, hop_first, hop_count, d_out, out_stride,
                                  out_origin, s, &launches, &started))) {
            // Failed before any chunk ran (reserve / pinned allocation): nothing moved, the same range may be retried.
            // Failed later: completed chunks have advanced d_tail and consumed stateful apply() calls, so a retry at
            // hop_first would overlap-add the wrong tail - only a restart at hop 0 is accepted from here.
            if (started)
                for (uint32_t c = ch_first; c < ch_first + n_channels; ++c) e->kernel_next_hop[c] = -1;
            return rc;
        }
        for (uint32_t c = ch_first; c < ch_first + n_channels; ++c) e->kernel_next_hop[c] = hop_first + hop_count;
    } else
    for (int64_t k0 = hop_first; k0 < hop_first + hop_count; k0 += chunk_max) {
        const int64_t kc = std::min<int64_t>(chunk_max, hop_first + hop_count - k0);
        if ((rc = run_chunk(k0, kc, false))) return rc;
    }
    if (timed) {
        RC_HIP(hipEventRecord(e->ev1[e->timed_calls % rc_engine::kRing], s));
        e->timed_calls++;
        e->stats_hops = (uint64_t)hop_count * n_channels;
        e->stats_launches = launches;
    }
    return RC_OK;
}

// one hop described by p (spec / ybuf set) on the engine's stream, any supported window length
int single_hop_forward(rc_engine *e, const rc::HopParams &p) {
    if (e->gen) {
        rc::HopParams g = p;
        if (e->bl_log2l > 14) {
            if (int rcw = e->d_ysub.reserve(sizeof(float2) << e->bl_log2l)) return rcw;
            g.bl_wk = (float2 *)e->d_ysub.p;
        }
        RC_HIP(rc::launch_gen(0, g, e->stream));
        return RC_OK;
    }
    if (e->log2n <= 14) {
        RC_HIP(rc::launch_hop(e->log2n, rc::MODE_FORWARD, p, e->stream));
        return RC_OK;
    }
    if (int rc = e->d_ysub.reserve((size_t)e->par.window_len * sizeof(float))) return rc;
    rc::BigParams b = big_params(e, p);
    b.ysub = (float2 *)e->d_ysub.p;
    b.spec = p.spec;
    b.ybuf = p.ybuf;
    RC_HIP(rc::launch_big(0, b, e->stream));
    RC_HIP(rc::launch_big(1, b, e->stream, rc::MODE_FORWARD));
    return RC_OK;
}
int single_hop_resynth(rc_engine *e, const rc::HopParams &p) {
    if (e->gen) {  // (p.spec holds the spectrum: magnitudes x phasors in place, then the inverse DFT)
        rc::HopParams g = p;
        if (e->bl_log2l > 14) {
            if (int rcw = e->d_ysub.reserve(sizeof(float2) << e->bl_log2l)) return rcw;
            g.bl_wk = (float2 *)e->d_ysub.p;
        }
        RC_HIP(rc::launch_gen(1, g, e->stream));
        RC_HIP(rc::launch_gen(2, g, e->stream));
        return RC_OK;
    }
    if (e->log2n <= 14) {
        RC_HIP(rc::launch_hop(e->log2n, rc::MODE_RESYNTH, p, e->stream));
        return RC_OK;
    }
    if (int rc = e->d_ysub.reserve((size_t)e->par.window_len * sizeof(float))) return rc;
    rc::BigParams b = big_params(e, p);
    b.ysub = (float2 *)e->d_ysub.p;
    b.spec = p.spec;
    b.ybuf = p.ybuf;
    RC_HIP(rc::launch_big(1, b, e->stream, rc::MODE_RESYNTH));
    RC_HIP(rc::launch_big(2, b, e->stream));
    return RC_OK;
}

// No C++ exception may cross the C-ABI (a Rust or C host cannot unwind through it): every entry
// point that allocates is a function-try-block that maps what it catches to a status code.
int rc_catch() noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(RC_ENOMEM, "host allocation failed");
    } catch (const std::exception &ex) {
        return fail(RC_EHIP, "internal error: %s", ex.what());
    } catch (...) {
        return fail(RC_EHIP, "internal error");
    }
}

}  // namespace

extern "C" {

const char *rc_last_error(void) { return g_err.c_str(); }
int rc_abi_version(void) { return RC_ABI_VERSION; }
const char *rc_kernel_id(void) { return RC_KERNEL_ID; }

int rc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int rc_derive_params(const rc_config *cfg, rc_params *out) {
    if (!out) return fail(RC_EINVAL, "null out");
    return derive(cfg, out);
}

size_t rc_offline_output_len(const rc_config *cfg, size_t in_len) {
    rc_params p;
    if (derive(cfg, &p) != RC_OK) return 0;
    return (size_t)(offline_windows(p, in_len) * p.window_out_len);
}

uint64_t rc_phase_key(uint64_t seed, uint32_t channel, uint64_t hop) {
    const uint64_t ctr = ((uint64_t)channel << 40) | (hop & 0xFFFFFFFFFFull);
    return mix64(mix64(seed) ^ ctr);
}

uint32_t rc_phase_hash(uint64_t key, uint32_t bin) {
    uint32_t x = bin * ((uint32_t)(key >> 32) | 1u) + (uint32_t)key;
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
float rc_phase_theta(uint64_t key, uint32_t bin, uint32_t n_bins) {
    const uint32_t half = n_bins / 2;
    const bool upper = bin >= half;
    const uint32_t h = rc_phase_hash(key, upper ? bin - half : bin);
    const float u = upper ? (float)(h & 0xFFFFu) * (1.0f / 65536.0f) : (float)(h >> 9) * (1.0f / 8388608.0f);
    return u * 3.14159265358979323846f;
}

int rc_engine_create(const rc_config *cfg, rc_engine **out) try {
    if (!out) return fail(RC_EINVAL, "null out");
    *out = nullptr;
    rc_params par;
    int rc = derive(cfg, &par);
    if (rc) return rc;
    int log2n = ilog2_exact(cfg->window_len);
    // Powers of two 32...65536 run the FFT kernels. Any other EVEN length up to 65536 (the reference takes any
    // -w through rustfft: src/main.rs:34, src/fft.rs:27-29) runs as plain O(N^2) DFTs on the device: correct,
    // not fast, and without a host frequency kernel.
    const bool gen = log2n < 0 && cfg->window_len >= 4 && cfg->window_len <= 65536 && cfg->window_len % 2 == 0;
    if (!gen && (log2n < 5 || log2n > 16))
        return fail(RC_EUNSUPPORTED, "window_len %u: the GPU path supports powers of two in [32, 65536] and even lengths in [4, 65536]", cfg->window_len);
    if (gen) log2n = 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        return fail(RC_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
    }
    if (cfg->device < 0 || cfg->device >= ndev) return fail(RC_EINVAL, "device %d out of range", cfg->device);
    RC_HIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    RC_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(RC_ENODEVICE, "device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);
    rc_engine *e = new (std::nothrow) rc_engine();
    if (!e) return fail(RC_ENOMEM, "out of host memory");
    e->cfg = *cfg;
    e->cfg.window = nullptr;
    e->par = par;
    e->log2n = log2n;
    e->gen = gen;
    e->device = cfg->device;
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->seed_mixed = mix64(cfg->seed);
    e->ch.resize(cfg->channels);
    e->kernel_next_hop.assign(cfg->channels, 0);
    const uint32_t N = cfg->window_len, M = N / 2, H = N / 2;
    std::vector<float> w(N), env(H);
    if (cfg->window) memcpy(w.data(), cfg->window, N * sizeof(float));
    else hanning(N, w.data());                    // src/main.rs:131
    crossfade_comp(H, env.data());                // src/stretcher.rs:56
    // the N = 16384 kernel computes the default window and the envelope in registers instead of
    // loading them (rc_hop16k.hip / rc_big4.hip, HANN); a caller-supplied window gets that path only when it
    // is the default one bit for bit
    bool default_window = true;
    if (cfg->window) {
        std::vector<float> d(N);
        hanning(N, d.data());
        default_window = memcmp(d.data(), w.data(), N * sizeof(float)) == 0;
    }
    std::vector<float> hann_rot;
    if (default_window && log2n >= 5 && !gen) {
        // thread t of the fused kernels touches samples 2 T q + 2 t + e: {cos, sin}(2 pi (2 t + e) / (len - 1)) for
        // the window (len = N) and the envelope (len = N / 2); T = 256 threads at N = 16384, 512 above, the generic
        // kernel's geometry below
        int threads = log2n == 14 ? 256 : 512;
        if (log2n < 14) rc::hop_geometry(log2n, &threads, nullptr);
        hann_rot.resize((size_t)2 * threads * 4);
        for (int part = 0; part < 2; ++part) {
            const double len1 = (double)((part ? H : N) - 1);
            for (int t = 0; t < threads; ++t)
                for (int b = 0; b < 2; ++b) {
                    const double beta = 2.0 * M_PI * (double)(2 * t + b) / len1;
                    hann_rot[((size_t)part * threads + t) * 4 + 2 * b] = (float)cos(beta);
                    hann_rot[((size_t)part * threads + t) * 4 + 2 * b + 1] = (float)sin(beta);
                }
        }
    }
    // twiddles in f64, rounded to f32 (as rustfft does). Windows that fit one workgroup:
    // wtab = exp(-2 pi i k / M) [M/2], rtab = exp(-2 pi i j / N) [M/4+1]. Larger windows run four
    // quarter FFTs of Ms = M/4 points: wtab is for Ms, rtab covers j <= Ms/2, t1 = exp(-2 pi i j / M).
    const bool big = log2n > 14;
    std::vector<float2> twg;
    if (gen) {
        twg.resize(N);
        for (uint32_t k = 0; k < N; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)N;
            twg[k] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    // chirp-z tables for the lengths whose packed half fits the LDS (bluestein_kernel), all in f64 first
    std::vector<float2> blt;
    uint32_t bl_log2l = 0;
    if (gen && RC_BLUESTEIN) {
        const uint32_t Mh = N / 2;
        while ((1u << bl_log2l) < 2 * Mh - 1 || bl_log2l < 1) ++bl_log2l;
        const uint32_t L = 1u << bl_log2l;
        blt.resize((size_t)Mh + L / 2 + L);
        std::vector<std::complex<double>> c(Mh), b(L, 0.0), w(L / 2);
        for (uint32_t n = 0; n < Mh; ++n) {
            const uint64_t r = ((uint64_t)n * n) % (2ull * Mh);  // n^2 mod 2M: the angle stays exact
            c[n] = std::polar(1.0, -M_PI * (double)r / (double)Mh);
            blt[n] = make_float2((float)c[n].real(), (float)c[n].imag());
            b[n] = std::conj(c[n]);
            if (n) b[L - n] = std::conj(c[n]);
        }
        for (uint32_t k = 0; k < L / 2; ++k) {
            w[k] = std::polar(1.0, -2.0 * M_PI * (double)k / (double)L);
            blt[Mh + k] = make_float2((float)w[k].real(), (float)w[k].imag());
        }
        for (int s = (int)bl_log2l - 1; s >= 0; --s) {  // the kernel's DIF (natural -> bit-reversed order)
            const uint32_t half = 1u << s;
            for (uint32_t q = 0; q < L / 2; ++q) {
                const uint32_t lo = q & (half - 1), i = ((q >> s) << (s + 1)) | lo, j = i + half;
                const std::complex<double> u = b[i], v = b[j];
                b[i] = u + v;
                b[j] = (u - v) * w[(size_t)lo << (bl_log2l - 1 - s)];
            }
        }
        for (uint32_t i = 0; i < L; ++i)
            blt[(size_t)Mh + L / 2 + i] = make_float2((float)(b[i].real() / L), (float)(b[i].imag() / L));
    }
    const uint32_t Mf = big ? M / 4 : M;  // length of the in-LDS FFT
    std::vector<float2> wtab(std::max<uint32_t>(1, Mf / 2)), rtab(big ? Mf / 2 + 1 : M / 4 + 1), t1;
    for (uint32_t k = 0; k < Mf / 2; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)Mf;
        wtab[k] = make_float2((float)cos(a), (float)sin(a));
    }
    for (uint32_t j = 0; j < rtab.size(); ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)N;
        rtab[j] = make_float2((float)cos(a), (float)sin(a));
    }
    if (big) {
        t1.resize(Mf + 1);
        for (uint32_t j = 0; j <= Mf; ++j) {
            const double a = -2.0 * M_PI * (double)j / (double)M;
            t1[j] = make_float2((float)cos(a), (float)sin(a));
        }
    }
    auto cleanup = [&](int code) {
        rc_engine_destroy(e);
        return code;
    };
#define RC_HIP_C(expr)                                                                          \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return cleanup(fail(RC_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)));       \
    } while (0)
    RC_HIP_C(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (int i = 0; i < rc_engine::kRing; ++i) {
        RC_HIP_C(hipEventCreate(&e->ev0[i]));
        RC_HIP_C(hipEventCreate(&e->ev1[i]));
    }
    RC_HIP_C(hipEventCreateWithFlags(&e->ev_last, hipEventDisableTiming));
    RC_HIP_C(hipHostMalloc((void **)&e->h_err, sizeof(uint32_t), hipHostMallocMapped));
    *e->h_err = 0;
    RC_HIP_C(hipHostGetDevicePointer((void **)&e->d_err, e->h_err, 0));
#if RC_TEST_HOOKS
    // test-hook library only (make hooks): the product library reads no diagnostic or tuning variable
    if (const char *diag = getenv("ROCODER_DIAG")) e->diag_flags = (uint32_t)strtoul(diag, nullptr, 0);
    if (const char *v = getenv("ROCODER_ROUNDS")) e->tune_rounds = atoi(v);
    if (const char *v = getenv("ROCODER_MIN_RUN")) e->tune_min_run = atoi(v);
    if (const char *v = getenv("ROCODER_B4_ROUNDS")) e->tune_b4_rounds = atoi(v);
#endif
    RC_HIP_C(hipMalloc((void **)&e->d_window, N * sizeof(float)));
    RC_HIP_C(hipMalloc((void **)&e->d_env, H * sizeof(float)));
    RC_HIP_C(hipMalloc((void **)&e->d_wtab, wtab.size() * sizeof(float2)));
    RC_HIP_C(hipMalloc((void **)&e->d_rtab, rtab.size() * sizeof(float2)));
    RC_HIP_C(hipMemcpy(e->d_window, w.data(), N * sizeof(float), hipMemcpyHostToDevice));
    RC_HIP_C(hipMemcpy(e->d_env, env.data(), H * sizeof(float), hipMemcpyHostToDevice));
    if (!hann_rot.empty()) {
        RC_HIP_C(hipMalloc((void **)&e->d_hann_rot, hann_rot.size() * sizeof(float)));
        RC_HIP_C(hipMemcpy(e->d_hann_rot, hann_rot.data(), hann_rot.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    RC_HIP_C(hipMemcpy(e->d_wtab, wtab.data(), wtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    RC_HIP_C(hipMemcpy(e->d_rtab, rtab.data(), rtab.size() * sizeof(float2), hipMemcpyHostToDevice));
    if (gen) {
        RC_HIP_C(hipMalloc((void **)&e->d_tw_gen, twg.size() * sizeof(float2)));
        RC_HIP_C(hipMemcpy(e->d_tw_gen, twg.data(), twg.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
    if (!blt.empty()) {
        RC_HIP_C(hipMalloc((void **)&e->d_bl_tab, blt.size() * sizeof(float2)));
        RC_HIP_C(hipMemcpy(e->d_bl_tab, blt.data(), blt.size() * sizeof(float2), hipMemcpyHostToDevice));
        e->bl_log2l = bl_log2l;
    }
    if (big) {
        RC_HIP_C(hipMalloc((void **)&e->d_t1, t1.size() * sizeof(float2)));
        RC_HIP_C(hipMemcpy(e->d_t1, t1.data(), t1.size() * sizeof(float2), hipMemcpyHostToDevice));
        std::vector<float2> wm(N / 64 + 1);  // big4_kernel: W_M^k for k <= RES/2, RES = N/32
        for (uint32_t k = 0; k < wm.size(); ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)M;
            wm[k] = make_float2((float)cos(a), (float)sin(a));
        }
        RC_HIP_C(hipMalloc((void **)&e->d_wtab_m, wm.size() * sizeof(float2)));
        RC_HIP_C(hipMemcpy(e->d_wtab_m, wm.data(), wm.size() * sizeof(float2), hipMemcpyHostToDevice));
    }
#undef RC_HIP_C
    *out = e;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

void rc_engine_destroy(rc_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->d_window) (void)hipFree(e->d_window);
    if (e->d_env) (void)hipFree(e->d_env);
    if (e->d_hann_rot) (void)hipFree(e->d_hann_rot);
    if (e->d_wtab) (void)hipFree(e->d_wtab);
    if (e->d_rtab) (void)hipFree(e->d_rtab);
    if (e->d_t1) (void)hipFree(e->d_t1);
    if (e->d_wtab_m) (void)hipFree(e->d_wtab_m);
    if (e->d_tw_gen) (void)hipFree(e->d_tw_gen);
    if (e->d_bl_tab) (void)hipFree(e->d_bl_tab);
    e->d_in.release();
    e->d_out.release();
    e->d_spec.release();
    e->d_spec2.release();
    e->d_ybuf.release();
    e->d_ysub.release();
    e->d_obuf.release();
    e->d_tail.release();
    e->d_hop_in.release();
    e->d_hop_out.release();
    e->d_xtail.release();
    e->d_seam_head.release();
    e->d_seam_flag.release();
    e->d_run_counter.release();
    e->d_tail_stage.release();
    for (int i = 0; i < rc_engine::KernelPipe::kSets; ++i) {
        e->kp.d_spec[i].release();
        e->kp.d_ybuf[i].release();
        e->kp.d_ysub[i].release();
        if (e->kp.h_in[i]) (void)hipHostFree(e->kp.h_in[i]);
        if (e->kp.h_out[i]) (void)hipHostFree(e->kp.h_out[i]);
        if (e->kp.ev_fwd[i]) (void)hipEventDestroy(e->kp.ev_fwd[i]);
        if (e->kp.ev_back[i]) (void)hipEventDestroy(e->kp.ev_back[i]);
    }
    if (e->kp.ev_in) (void)hipEventDestroy(e->kp.ev_in);
    if (e->kp.ev_done) (void)hipEventDestroy(e->kp.ev_done);
    if (e->kp.kf) (void)hipStreamDestroy(e->kp.kf);
    if (e->kp.kb) (void)hipStreamDestroy(e->kp.kb);
    for (int i = 0; i < rc_engine::kRing; ++i) {
        if (e->ev0[i]) (void)hipEventDestroy(e->ev0[i]);
        if (e->ev1[i]) (void)hipEventDestroy(e->ev1[i]);
    }
    for (auto &c : e->ch)
        for (int b = 0; b < 2; ++b) {
            if (c.blk_p[b]) (void)hipHostFree(c.blk_p[b]);
            if (c.blk_ev[b]) (void)hipEventDestroy(c.blk_ev[b]);
        }
    for (int i = 0; i < rc_engine::HostPipe::kWorkers; ++i) {
        if (e->hp.slot[i]) (void)hipHostFree(e->hp.slot[i]);
        if (e->hp.st[i]) (void)hipStreamDestroy(e->hp.st[i]);
    }
    if (e->hp.ev_start) (void)hipEventDestroy(e->hp.ev_start);
    for (hipEvent_t ev : e->hp.ev_chunk) (void)hipEventDestroy(ev);
    if (e->ev_last) (void)hipEventDestroy(e->ev_last);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->h_err) (void)hipHostFree(e->h_err);
    delete e;
}

int rc_engine_get_params(const rc_engine *e, rc_params *out) {
    if (!e || !out) return fail(RC_EINVAL, "null argument");
    *out = e->par;
    return RC_OK;
}

size_t rc_engine_channel_bound(const rc_engine *e) {
    if (!e) return 0;
    // src/stretcher.rs:82-85
    const float v = ((float)e->par.window_len / (float)e->cfg.sample_rate) / e->cfg.buffer_secs;
    return (size_t)ceilf(v);
}

int rc_engine_push_input(rc_engine *e, uint32_t channel, const float *samples, size_t n) try {
    if (!e || channel >= e->ch.size() || (!samples && n)) return fail(RC_EINVAL, "bad argument");
    Channel &c = e->ch[channel];
    if (c.closed) return fail(RC_EINVAL, "channel %u is closed", channel);
    c.fifo.insert(c.fifo.end(), samples, samples + n);
    c.total_in += n;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_engine_close_input(rc_engine *e, uint32_t channel) {
    if (!e || channel >= e->ch.size()) return fail(RC_EINVAL, "bad argument");
    e->ch[channel].closed = true;
    return RC_OK;
}

int rc_engine_is_done(const rc_engine *e, uint32_t channel) {
    if (!e || channel >= e->ch.size()) return fail(RC_EINVAL, "bad argument");
    const Channel &c = e->ch[channel];
    return (c.done_window >= 0 && (int64_t)c.windows_out > c.done_window) ? 1 : 0;
}

namespace {
// How many whole windows of the channel can be computed now: RC_OK with *nwin > 0, or RC_WOULD_BLOCK (more input
// needed), or RC_EINVAL (called after is_done).
//  hop h needs samples [h*step, h*step+N); on a closed channel the shortfall is zero padded and `done` is raised at
//  the first short hop (src/stretcher.rs:123-135).
int stream_plan(rc_engine *e, Channel &c, uint64_t *nwin_out) {
    const rc_params &P = e->par;
    const uint32_t N = P.window_len, hpw = P.hops_per_window, step = P.sample_step_len, wout = P.window_out_len;
    const uint64_t k0 = c.next_window * hpw;
    uint64_t nwin = 0;
    if (c.closed) {
        if (c.done_window >= 0 && (int64_t)c.next_window > c.done_window)
            return fail(RC_EINVAL, "next_window called after is_done (src/stretcher_processor.rs:64-68 never does)");
        const uint64_t kd = c.total_in >= N ? (c.total_in - N) / step + 1 : 0;
        const uint64_t last_win = kd / hpw;  // window containing the first short hop
        nwin = last_win + 1 - c.next_window;
        c.done_window = (int64_t)last_win;
    } else {
        if (c.total_in < N) return RC_WOULD_BLOCK;
        const uint64_t full_hops = (c.total_in - N) / step + 1;  // hops 0..full_hops-1 are complete
        if (full_hops < k0 + hpw) return RC_WOULD_BLOCK;
        nwin = (full_hops - k0) / hpw;
    }
    uint64_t max_hops = e->cfg.max_batch_hops ? e->cfg.max_batch_hops : 2048;
    // live mode: never run further ahead than the bounded output queue would
    // (src/stretcher_processor.rs:34, src/stretcher.rs:82-85) unless the whole input is known
    uint64_t max_win = std::max<uint64_t>(1, max_hops / hpw);
    if (!c.closed) max_win = std::min<uint64_t>(max_win, std::max<size_t>(1, rc_engine_channel_bound(e)));
    // a host frequency kernel may be stateful: no look-ahead, so that the processor's round-robin over the
    // channels (src/stretcher_processor.rs:63-70: windows outer, channels inner) is also the order in
    // which apply() sees the hops
    if (e->cfg.kernel) max_win = 1;
    // one batch fills at most 16 MiB of a pinned block
    max_win = std::min<uint64_t>(max_win, std::max<uint64_t>(1, (((size_t)16 << 20) / sizeof(float)) / wout));
    *nwin_out = std::min<uint64_t>(nwin, max_win);
    return RC_OK;
}
// Enqueues (no wait) the next `nwin` windows of the channel on the engine's stream: input span up, the hops, the
// batch down into pinned block `blk`, an event behind it. Advances c.next_window and trims the input history.
int stream_enqueue(rc_engine *e, uint32_t channel, uint64_t nwin, int blk) {
    Channel &c = e->ch[channel];
    const rc_params &P = e->par;
    const uint32_t N = P.window_len, hpw = P.hops_per_window, step = P.sample_step_len, wout = P.window_out_len;
    const uint64_t k0 = c.next_window * hpw, hop_count = nwin * hpw;
    // input span: from the hop before k0 (its tail is recomputed) unless a user kernel
    // carries the tail on the device
    const bool recompute = !e->cfg.kernel && k0 > 0;
    const uint64_t span_lo = (recompute ? k0 - 1 : k0) * (uint64_t)step;
    const uint64_t span_hi = std::min<uint64_t>(c.total_in, (k0 + hop_count - 1) * (uint64_t)step + N);
    if (span_lo < c.fifo_base) return fail(RC_EINVAL, "internal: input history dropped");
    const size_t span = span_hi > span_lo ? (size_t)(span_hi - span_lo) : 0;
    int rc;
    RC_HIP(hipSetDevice(e->device));
    // (the scratch is shared by the channels' batches: they follow one another on the engine's stream, and a
    // reserve() that has to grow it frees the old block, which waits for the device)
    if ((rc = e->d_in.reserve(std::max<size_t>(span, 1) * sizeof(float)))) return rc;
    if ((rc = e->d_out.reserve((size_t)nwin * wout * sizeof(float)))) return rc;
    if (c.blk_cap[blk] < (size_t)nwin * wout) {
        if (c.blk_p[blk]) RC_HIP(hipHostFree(c.blk_p[blk]));
        c.blk_p[blk] = nullptr;
        c.blk_cap[blk] = 0;
        RC_HIP(hipHostMalloc((void **)&c.blk_p[blk], (size_t)nwin * wout * sizeof(float), hipHostMallocDefault));
        c.blk_cap[blk] = (size_t)nwin * wout;
    }
    if (!c.blk_ev[blk]) RC_HIP(hipEventCreateWithFlags(&c.blk_ev[blk], hipEventDisableTiming));
    if (span)  // (pageable source: the runtime has consumed it when the call returns)
        RC_HIP(hipMemcpyAsync(e->d_in.p, c.fifo.data() + (span_lo - c.fifo_base), span * sizeof(float),
                              hipMemcpyHostToDevice, e->stream));
    rc = run_hops(e, (const float *)e->d_in.p, 0, (int64_t)span_lo, (int64_t)span, channel, 1,
                  (int64_t)k0, (int64_t)hop_count, (float *)e->d_out.p, 0,
                  (int64_t)(c.next_window * wout), e->stream, false);
    if (rc) return rc;
    RC_HIP(hipMemcpyAsync(c.blk_p[blk], e->d_out.p, (size_t)nwin * wout * sizeof(float), hipMemcpyDeviceToHost,
                          e->stream));
    RC_HIP(hipEventRecord(c.blk_ev[blk], e->stream));
    c.next_window += nwin;
    // drop input no later hop needs (keep from (next hop - 1) * step) once it is at least half of what is
    // held: erasing the front of the vector every batch would move the whole remainder each time
    const uint64_t keep_from = (c.next_window * hpw > 0 ? c.next_window * hpw - 1 : 0) * (uint64_t)step;
    if (keep_from > c.fifo_base) {
        const uint64_t drop = std::min<uint64_t>(keep_from - c.fifo_base, c.fifo.size());
        if (drop * 2 >= c.fifo.size()) {
            c.fifo.erase(c.fifo.begin(), c.fifo.begin() + drop);
            c.fifo_base += drop;
        }
    }
    return RC_OK;
}
// The window the next hand-out of the channel delivers, computing / waiting as needed: *win points into the
// channel's pinned block (valid until the next hand-out of the same channel).
int stream_next(rc_engine *e, uint32_t channel, const float **win) {
    int rc;
    Channel &c = e->ch[channel];
    const uint32_t wout = e->par.window_out_len;
    if (c.ready_pos == c.ready_n) {
        // (the device error word is looked at once per batch, not per window: it lives in mapped host memory, and an
        // atomic exchange there costs about a microsecond - more than handing a window out)
        if ((rc = check_device_error(e))) return rc;
        if (c.ahead_n) {  // the look-ahead batch: wait for its copy, make its block the current one
            RC_HIP(hipEventSynchronize(c.blk_ev[c.cur ^ 1]));
            c.cur ^= 1;
            c.ready_n = c.ahead_n;
            c.ahead_n = 0;
        } else {
            uint64_t nwin = 0;
            if ((rc = stream_plan(e, c, &nwin))) return rc;
            if ((rc = stream_enqueue(e, channel, nwin, c.cur))) return rc;
            RC_HIP(hipEventSynchronize(c.blk_ev[c.cur]));
            c.ready_n = nwin;
        }
        c.ready_pos = 0;
        if ((rc = check_device_error(e))) return rc;
        // Look-ahead: with the whole input known (closed channel) and no host kernel (whose apply() order is the
        // processor's), the following batch is enqueued now and overlaps the hand-out of this one. A live channel
        // keeps to the bounded queue's look-ahead (stream_plan) and computes when asked.
        if (c.closed && !e->cfg.kernel && !(c.done_window >= 0 && (int64_t)c.next_window > c.done_window)) {
            uint64_t nwin = 0;
            if (stream_plan(e, c, &nwin) == RC_OK && nwin) {
                if ((rc = stream_enqueue(e, channel, nwin, c.cur ^ 1))) return rc;
                c.ahead_n = nwin;
            }
        }
    }
    *win = c.blk_p[c.cur] + (size_t)c.ready_pos * wout;
    c.ready_pos++;
    c.windows_out++;
    return RC_OK;
}
}  // namespace

int rc_engine_next_window(rc_engine *e, uint32_t channel, float *out, size_t out_cap, size_t *n_out) try {
    if (!e || channel >= e->ch.size() || !out) return fail(RC_EINVAL, "bad argument");
    const uint32_t wout = e->par.window_out_len;
    if (out_cap < wout) return fail(RC_ECAPACITY, "out_cap %zu < window_out_len %u", out_cap, wout);
    const float *win = nullptr;
    if (int rc = stream_next(e, channel, &win)) return rc;
    memcpy(out, win, (size_t)wout * sizeof(float));
    if (n_out) *n_out = wout;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_engine_next_window_view(rc_engine *e, uint32_t channel, const float **window, size_t *n_out) try {
    if (!e || channel >= e->ch.size() || !window) return fail(RC_EINVAL, "bad argument");
    *window = nullptr;
    if (int rc = stream_next(e, channel, window)) return rc;
    if (n_out) *n_out = e->par.window_out_len;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_engine_stretch_device_range(rc_engine *e, const float *d_in, size_t in_stride, size_t in_len,
                                   uint32_t ch_first, uint32_t ch_count, uint64_t win_first,
                                   uint64_t win_count, float *d_out, size_t out_stride,
                                   size_t out_cap, void *hip_stream) try {
    if (!e || !d_out || (!d_in && in_len)) return fail(RC_EINVAL, "null argument");
    int rc = check_device_error(e);
    if (rc) return rc;
    if (ch_first + ch_count > e->cfg.channels) return fail(RC_EINVAL, "channel range out of bounds");
    const uint64_t total_win = offline_windows(e->par, in_len);
    if (win_first + win_count > total_win) return fail(RC_EINVAL, "window range out of bounds");
    const uint64_t wout = e->par.window_out_len;
    if (out_cap < win_count * wout) return fail(RC_ECAPACITY, "out_cap %zu < %llu", out_cap, (unsigned long long)(win_count * wout));
    RC_HIP(hipSetDevice(e->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
    const uint32_t hpw = e->par.hops_per_window;
    return run_hops(e, d_in + (size_t)ch_first * in_stride, in_stride, 0, (int64_t)in_len, ch_first,
                    ch_count, (int64_t)(win_first * hpw), (int64_t)(win_count * hpw), d_out, out_stride,
                    (int64_t)(win_first * wout), s, true);
} catch (...) {
    return rc_catch();
}

int rc_engine_stretch_device(rc_engine *e, const float *d_in, size_t in_stride, size_t in_len,
                             float *d_out, size_t out_stride, size_t out_cap, size_t *out_len,
                             void *hip_stream) try {
    if (!e) return fail(RC_EINVAL, "null engine");
    const uint64_t total_win = offline_windows(e->par, in_len);
    int rc = rc_engine_stretch_device_range(e, d_in, in_stride, in_len, 0, e->cfg.channels, 0,
                                            total_win, d_out, out_stride, out_cap, hip_stream);
    if (rc == RC_OK && out_len) *out_len = (size_t)(total_win * e->par.window_out_len);
    return rc;
} catch (...) {
    return rc_catch();
}

namespace {
// Host <-> device copies of a host-buffer call. Pageable memory cannot be the target of a DMA, so the runtime's own
// hipMemcpyAsync stages it on one thread (measured 14 GB/s on 1.7 GB of output). Here up to eight workers each own a
// 16 MiB pinned slot and a stream: DMA into / out of the slot, memcpy between slot and the caller's buffer; the
// workers' DMAs and memcpys overlap each other. `to_device`: rows host[c][0..n) -> dev + c * dev_stride, else the
// reverse. D2H starts after everything enqueued on e->stream so far; on return all copies are complete.
int host_copy(rc_engine *e, bool to_device, float *const *host, float *dev, size_t dev_stride, size_t n, uint32_t C) {
    using HP = rc_engine::HostPipe;
    const size_t total = (size_t)C * n * sizeof(float);
    if (total == 0) return RC_OK;
    const size_t slot_floats = HP::kSlotBytes / sizeof(float);
    const size_t per_row = (n + slot_floats - 1) / slot_floats, n_chunks = per_row * C;
    int workers = (int)std::min<size_t>((size_t)std::max(1, std::min(e->hp.max_workers, (int)HP::kWorkers)), n_chunks);
    workers = std::max(1, std::min<int>(workers, (int)std::max(1u, std::thread::hardware_concurrency())));
    if (total < ((size_t)4 << 20)) {  // small: the plain path (no threads)
        for (uint32_t c = 0; c < C; ++c) {
            if (to_device) RC_HIP(hipMemcpyAsync(dev + (size_t)c * dev_stride, host[c], n * sizeof(float), hipMemcpyHostToDevice, e->stream));
            else RC_HIP(hipMemcpyAsync(host[c], dev + (size_t)c * dev_stride, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        }
        if (!to_device) RC_HIP(hipStreamSynchronize(e->stream));
        return RC_OK;
    }
    for (int w = 0; w < workers; ++w) {
        if (!e->hp.slot[w]) RC_HIP(hipHostMalloc(&e->hp.slot[w], HP::kSlotBytes, hipHostMallocDefault));
        if (!e->hp.st[w]) RC_HIP(hipStreamCreateWithFlags(&e->hp.st[w], hipStreamNonBlocking));
    }
    if (!e->hp.ev_start) RC_HIP(hipEventCreateWithFlags(&e->hp.ev_start, hipEventDisableTiming));
    // the workers' streams start behind what the engine's stream holds (the compute of a D2H; earlier users of d_in)
    RC_HIP(hipEventRecord(e->hp.ev_start, e->stream));
    for (int w = 0; w < workers; ++w) RC_HIP(hipStreamWaitEvent(e->hp.st[w], e->hp.ev_start, 0));
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};
    auto work = [&](int w) {
        if (hipSetDevice(e->device) != hipSuccess) { err = 1; return; }
        float *slot = (float *)e->hp.slot[w];
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks || err.load()) return;
            const uint32_t c = (uint32_t)(i / per_row);
            const size_t off = (i % per_row) * slot_floats, cnt = std::min(slot_floats, n - off);
            float *d = dev + (size_t)c * dev_stride + off;
            float *h = host[c] + off;
            if (to_device) {
                std::memcpy(slot, h, cnt * sizeof(float));
                if (hipMemcpyAsync(d, slot, cnt * sizeof(float), hipMemcpyHostToDevice, e->hp.st[w]) != hipSuccess ||
                    hipStreamSynchronize(e->hp.st[w]) != hipSuccess) { err = 1; return; }
            } else {
                if (hipMemcpyAsync(slot, d, cnt * sizeof(float), hipMemcpyDeviceToHost, e->hp.st[w]) != hipSuccess ||
                    hipStreamSynchronize(e->hp.st[w]) != hipSuccess) { err = 1; return; }
                std::memcpy(h, slot, cnt * sizeof(float));
            }
        }
    };
    JoinedThreads th;
    std::vector<int> mine{0};  // worker slots this thread serves itself (slot 0 + any whose thread did not start)
    for (int w = 1; w < workers; ++w)
        if (!th.start(work, w)) mine.push_back(w);
    for (int w : mine) work(w);
    th.join();
    if (err.load()) return fail(RC_EHIP, "host copy: %s", hipGetErrorString(hipGetLastError()));
    // (H2D: every worker synchronised its stream, so whatever is launched on e->stream next sees the data)
    return RC_OK;
}

// ---- host-buffer jobs as a pipeline (round 5) --------------------------------------------------------------------
// A piece = channels [ch_first, ch_first + ch_count) x windows [w0, w1) of the job, with its own device regions: d_in
// holds samples [in_lo, in_lo + span) of each of its channels (row stride span), d_out its n_sh output samples per
// channel. The piece is cut into window chunks of about one staging slot of output per channel; per chunk:
//   upload the part of the input span the chunk reads beyond what earlier chunks brought  (copy workers)
//   run_hops of the chunk on the engine's stream, an event behind it                      (calling thread)
//   download the chunk, one task per channel and slot-sized part                          (copy workers)
// so that the download of chunk i runs under the kernel of chunk i + 1 and the upload of chunk i + 2 (PCIe is full
// duplex). Rows the caller allocated with rc_host_alloc (or registered with hipHostRegister) are the DMA's source /
// target themselves; pageable rows are staged through the workers' pinned slots, the destination pages of a download
// populated (MADV_POPULATE_WRITE) while its DMA is in flight.
struct HostPiece {
    uint32_t ch_first, ch_count;
    uint64_t w0, w1;
    float *d_in;
    size_t in_lo, span;
    float *d_out;
    size_t n_sh;
};

bool row_is_pinned(const float *p, size_t n) {
    if (!p || n == 0) return true;
    for (const float *q : {p, p + (n - 1)}) {
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, q) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

#ifndef RC_POPULATE_MODE
#define RC_POPULATE_MODE 3  // bit 0: MADV_POPULATE_WRITE, bit 1: MADV_HUGEPAGE first (A/B: tests/dev/e2e_host.py, profiles/README.md)
#endif
// A download into pageable memory the caller has never touched (a fresh output array) takes a page fault per 4 KiB
// while it is copied. Ahead of the copy, while the DMA of the piece is in flight: ask for huge pages on the part of the
// destination that can take them and have the kernel populate the range in one call.
void populate_pages(void *p, size_t bytes) {
#if defined(__linux__) && defined(MADV_POPULATE_WRITE)
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    if (RC_POPULATE_MODE & 2) {
        const uintptr_t huge = (uintptr_t)2 << 20, a = (lo + huge - 1) & ~(huge - 1), b = hi & ~(huge - 1);
        if (b > a) (void)madvise((void *)a, b - a, MADV_HUGEPAGE);
    }
    if (RC_POPULATE_MODE & 1) {
        const uintptr_t page = 4096, a = (lo + page - 1) & ~(page - 1), b = hi & ~(page - 1);
        if (b > a) (void)madvise((void *)a, b - a, MADV_POPULATE_WRITE);  // (older kernels: EINVAL, the memcpy faults them in)
    }
#else
    (void)p;
    (void)bytes;
#endif
}

int host_pipeline(rc_engine *e, const std::vector<HostPiece> &pieces, const float *const *in_rows, size_t in_len,
                  float *const *out_rows) {
    using HP = rc_engine::HostPipe;
    const rc_params &par = e->par;
    const uint64_t wout = par.window_out_len;
    const uint32_t hpw = par.hops_per_window;
    const size_t slot_floats = HP::kSlotBytes / sizeof(float);
    struct Task {
        bool up;
        uint32_t c;       // absolute channel
        size_t host_off;  // offset into the caller's row
        float *dev;
        size_t cnt;
        int gate;         // download: the chunk whose event it waits for
    };
    struct Chunk {
        size_t piece;
        uint64_t w0, w1;
        size_t need_up;  // upload tasks [0, need_up) must be on the device before the chunk is launched
    };
    std::vector<Task> ups, downs_of_chunk_flat;
    std::vector<Chunk> chunks;
    std::vector<std::pair<size_t, size_t>> down_range;  // per chunk: [first, last) into downs_of_chunk_flat
    std::vector<size_t> up_until;                        // per chunk: its upload tasks end here (exclusive) in `ups`
    bool in_pinned = true, out_pinned = true;
    for (size_t pi = 0; pi < pieces.size(); ++pi) {
        const HostPiece &pc = pieces[pi];
        if (pc.w1 <= pc.w0 || pc.ch_count == 0) continue;
        for (uint32_t c = 0; c < pc.ch_count; ++c) {
            in_pinned = in_pinned && row_is_pinned(in_rows[pc.ch_first + c] + pc.in_lo, pc.span);
            out_pinned = out_pinned && row_is_pinned(out_rows[pc.ch_first + c] + (size_t)(pc.w0 * wout), pc.n_sh);
        }
        const uint64_t wc = std::max<uint64_t>(1, slot_floats / std::max<uint64_t>(1, wout));
        size_t brought = 0;  // samples of the piece's span already scheduled for upload
        for (uint64_t w = pc.w0; w < pc.w1; w += wc) {
            const uint64_t we = std::min<uint64_t>(pc.w1, w + wc);
            size_t lo, hi;
            input_span(par, w, we, in_len, &lo, &hi);
            const size_t upto = we == pc.w1 ? pc.span : std::min(pc.span, hi > pc.in_lo ? hi - pc.in_lo : 0);
            for (size_t off = brought; off < upto; off += slot_floats) {
                const size_t cnt = std::min(slot_floats, upto - off);
                for (uint32_t c = 0; c < pc.ch_count; ++c)
                    ups.push_back(Task{true, pc.ch_first + c, pc.in_lo + off, pc.d_in + (size_t)c * pc.span + off, cnt, -1});
            }
            brought = std::max(brought, upto);
            chunks.push_back(Chunk{pi, w, we, ups.size()});
            up_until.push_back(ups.size());
            const size_t first = downs_of_chunk_flat.size();
            const size_t o0 = (size_t)((w - pc.w0) * wout), o1 = (size_t)((we - pc.w0) * wout);
            for (uint32_t c = 0; c < pc.ch_count; ++c)
                for (size_t off = o0; off < o1; off += slot_floats)
                    downs_of_chunk_flat.push_back(Task{false, pc.ch_first + c, (size_t)(pc.w0 * wout) + off,
                                                       pc.d_out + (size_t)c * pc.n_sh + off, std::min(slot_floats, o1 - off),
                                                       (int)(chunks.size() - 1)});
            down_range.emplace_back(first, downs_of_chunk_flat.size());
        }
    }
    if (chunks.empty()) return RC_OK;
    // one list in an order that is a valid schedule: uploads of chunk k + 1 in front of the downloads of chunk k
    std::vector<Task> tasks;
    std::vector<long> up_index;  // per task: its index among the uploads, or -1
    tasks.reserve(ups.size() + downs_of_chunk_flat.size());
    {
        size_t u = 0;
        auto push_ups = [&](size_t until) {
            for (; u < until; ++u) {
                tasks.push_back(ups[u]);
                up_index.push_back((long)u);
            }
        };
        push_ups(up_until[0]);
        for (size_t k = 0; k < chunks.size(); ++k) {
            if (k + 1 < chunks.size()) push_ups(up_until[k + 1]);
            for (size_t d = down_range[k].first; d < down_range[k].second; ++d) {
                tasks.push_back(downs_of_chunk_flat[d]);
                up_index.push_back(-1);
            }
        }
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    int workers = std::max(1, std::min(e->hp.max_workers, (int)HP::kWorkers));
    workers = (int)std::min<size_t>((size_t)std::min<unsigned>((unsigned)workers, hw), tasks.size());
    if (in_pinned && out_pinned) workers = std::min(workers, 4);  // nothing to memcpy: the workers only keep the DMA queues fed
    for (int w = 0; w < workers; ++w) {
        if (!e->hp.slot[w] && !(in_pinned && out_pinned))
            RC_HIP(hipHostMalloc(&e->hp.slot[w], HP::kSlotBytes, hipHostMallocDefault));
        if (!e->hp.st[w]) RC_HIP(hipStreamCreateWithFlags(&e->hp.st[w], hipStreamNonBlocking));
    }
    while (e->hp.ev_chunk.size() < chunks.size()) {
        hipEvent_t ev = nullptr;
        RC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        e->hp.ev_chunk.push_back(ev);
    }
    if (!e->hp.ev_start) RC_HIP(hipEventCreateWithFlags(&e->hp.ev_start, hipEventDisableTiming));
    // the uploads overwrite d_in: behind whatever the engine's stream still holds
    RC_HIP(hipEventRecord(e->hp.ev_start, e->stream));
    for (int w = 0; w < workers; ++w) RC_HIP(hipStreamWaitEvent(e->hp.st[w], e->hp.ev_start, 0));

    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> up_done(ups.size(), 0);
    size_t up_prefix = 0;  // uploads [0, up_prefix) are on the device
    long launched = 0;     // chunks [0, launched) have their event recorded
    bool failed = false;
    std::atomic<size_t> next{0};
    auto fail_all = [&] {
        {
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
        }
        cv.notify_all();
    };
    auto work = [&](int w) {
        if (hipSetDevice(e->device) != hipSuccess) return fail_all();
        float *slot = (float *)e->hp.slot[w];
        hipStream_t st = e->hp.st[w];
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= tasks.size()) return;
            const Task &t = tasks[i];
            const size_t bytes = t.cnt * sizeof(float);
            if (t.up) {
                const float *h = in_rows[t.c] + t.host_off;
                const void *src = h;
                if (!in_pinned) {
                    std::memcpy(slot, h, bytes);
                    src = slot;
                }
                if (hipMemcpyAsync(t.dev, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipStreamSynchronize(st) != hipSuccess)
                    return fail_all();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    up_done[(size_t)up_index[i]] = 1;
                    while (up_prefix < up_done.size() && up_done[up_prefix]) ++up_prefix;
                }
                cv.notify_all();
            } else {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return failed || launched > t.gate; });
                    if (failed) return;
                }
                float *h = out_rows[t.c] + t.host_off;
                if (hipStreamWaitEvent(st, e->hp.ev_chunk[(size_t)t.gate], 0) != hipSuccess ||
                    hipMemcpyAsync(out_pinned ? (void *)h : (void *)slot, t.dev, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
                    return fail_all();
                if (!out_pinned) populate_pages(h, bytes);  // page faults of a fresh output array, under the DMA
                if (hipStreamSynchronize(st) != hipSuccess) return fail_all();
                if (!out_pinned) std::memcpy(h, slot, bytes);
            }
        }
    };
    JoinedThreads th;
    int started = 0;
    for (int w = 0; w < workers; ++w) started += th.start(work, w) ? 1 : 0;
    int rc = RC_OK;
    if (started == 0) {
        fail_all();
        rc = fail(RC_ENOMEM, "host pipeline: no copy worker thread could be started");
    }
    for (size_t k = 0; k < chunks.size() && rc == RC_OK; ++k) {
        const Chunk &ck = chunks[k];
        const HostPiece &pc = pieces[ck.piece];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return failed || up_prefix >= ck.need_up; });
            if (failed) break;
        }
        rc = run_hops(e, pc.d_in, pc.span, (int64_t)pc.in_lo, (int64_t)in_len - (int64_t)pc.in_lo, pc.ch_first, pc.ch_count,
                      (int64_t)(ck.w0 * hpw), (int64_t)((ck.w1 - ck.w0) * hpw), pc.d_out, pc.n_sh, (int64_t)(pc.w0 * wout),
                      e->stream, k == 0);
        if (rc == RC_OK && hipEventRecord(e->hp.ev_chunk[k], e->stream) != hipSuccess) rc = fail(RC_EHIP, "hipEventRecord failed");
        if (rc != RC_OK) {
            fail_all();
            break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            launched = (long)k + 1;
        }
        cv.notify_all();
    }
    th.join();
    if (rc != RC_OK) return rc;
    if (failed) return fail(RC_EHIP, "host pipeline: a copy failed: %s", hipGetErrorString(hipGetLastError()));
    RC_HIP(hipStreamSynchronize(e->stream));
    return check_device_error(e);
}
}  // namespace

int rc_engine_stretch_host(rc_engine *e, const float *const *in, size_t in_len, float *const *out,
                           size_t out_cap, size_t *out_len) try {
    if (!e || !in || !out) return fail(RC_EINVAL, "null argument");
    int rc = check_device_error(e);
    if (rc) return rc;
    const uint32_t C = e->cfg.channels;
    const uint64_t total_win = offline_windows(e->par, in_len);
    const size_t n_out = (size_t)(total_win * e->par.window_out_len);
    if (out_cap < n_out) return fail(RC_ECAPACITY, "out_cap %zu < %zu", out_cap, n_out);
    RC_HIP(hipSetDevice(e->device));
    const size_t in_stride = std::max<size_t>(in_len, 1);
    if ((rc = e->d_in.reserve((size_t)C * in_stride * sizeof(float)))) return rc;
    if ((rc = e->d_out.reserve((size_t)C * std::max<size_t>(n_out, 1) * sizeof(float)))) return rc;
    if (!e->cfg.kernel) {  // upload, compute and download overlapped chunk by chunk (host_pipeline)
        std::vector<HostPiece> piece{HostPiece{0, C, 0, total_win, (float *)e->d_in.p, 0, in_len, (float *)e->d_out.p, n_out}};
        if ((rc = host_pipeline(e, piece, in, in_len, out))) return rc;
        if (out_len) *out_len = n_out;
        return RC_OK;
    }
    // a host frequency kernel has its own three-stage pipeline over the spectra: whole input up, whole output down
    if ((rc = host_copy(e, true, const_cast<float *const *>(in), (float *)e->d_in.p, in_stride, in_len, C))) return rc;
    rc = rc_engine_stretch_device_range(e, (const float *)e->d_in.p, in_stride, in_len, 0, C, 0, total_win,
                                        (float *)e->d_out.p, n_out, n_out, e->stream);
    if (rc) return rc;
    if ((rc = host_copy(e, false, out, (float *)e->d_out.p, n_out, n_out, C))) return rc;
    RC_HIP(hipStreamSynchronize(e->stream));
    if ((rc = check_device_error(e))) return rc;
    if (out_len) *out_len = n_out;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_host_alloc(size_t bytes, void **out) try {
    if (!out) return fail(RC_EINVAL, "null argument");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(RC_ENODEVICE, "no HIP device: page-locked memory is allocated through the HIP runtime");
    }
    void *p = nullptr;
    const hipError_t err = hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocPortable);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_ENOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    }
    *out = p;
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_host_free(void *p) {
    if (!p) return RC_OK;
    const hipError_t err = hipHostFree(p);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EINVAL, "hipHostFree: %s (not a pointer from rc_host_alloc?)", hipGetErrorString(err));
    }
    return RC_OK;
}

int rc_engine_synchronize(rc_engine *e) {
    if (!e) return fail(RC_EINVAL, "null engine");
    RC_HIP(hipSetDevice(e->device));
    RC_HIP(hipStreamSynchronize(e->stream));
    return check_device_error(e);
}

int rc_engine_kernel_times(rc_engine *e, float *ms, size_t cap, size_t *n_out) {
    if (!e || (!ms && cap)) return fail(RC_EINVAL, "null argument");
    const uint64_t have = std::min<uint64_t>(e->timed_calls, rc_engine::kRing);
    const size_t n = (size_t)std::min<uint64_t>(have, cap);
    if (n) RC_HIP(hipEventSynchronize(e->ev1[(e->timed_calls - 1) % rc_engine::kRing]));
    if (int rc = check_device_error(e)) return rc;
    for (size_t i = 0; i < n; ++i) {  // oldest of the last n first
        const uint64_t call = e->timed_calls - n + i;
        RC_HIP(hipEventElapsedTime(&ms[i], e->ev0[call % rc_engine::kRing], e->ev1[call % rc_engine::kRing]));
    }
    if (n_out) *n_out = n;
    return RC_OK;
}

int rc_engine_last_kernel_stats(rc_engine *e, float *kernel_ms, uint64_t *hops, uint32_t *launches) {
    if (!e) return fail(RC_EINVAL, "null engine");
    if (!e->timed_calls) return fail(RC_EINVAL, "no timed launch recorded yet");
    float ms = 0.f;
    if (int rc = rc_engine_kernel_times(e, &ms, 1, nullptr)) return rc;
    if (kernel_ms) *kernel_ms = ms;
    if (hops) *hops = e->stats_hops;
    if (launches) *launches = e->stats_launches;
    return RC_OK;
}

int rc_engine_forward_fft(rc_engine *e, const float *samples, float *out_reim) try {
    if (!e || !samples || !out_reim) return fail(RC_EINVAL, "null argument");
    const uint32_t N = e->par.window_len;
    RC_HIP(hipSetDevice(e->device));
    int rc;
    if ((rc = e->d_hop_in.reserve(N * sizeof(float)))) return rc;
    if ((rc = e->d_hop_out.reserve((size_t)N * 2 * sizeof(float)))) return rc;
    RC_HIP(hipMemcpyAsync(e->d_hop_in.p, samples, N * sizeof(float), hipMemcpyHostToDevice, e->stream));
    rc::HopParams p = base_params(e);
    p.x = (const float *)e->d_hop_in.p;
    p.in_len = N;
    p.xtail = p.x;
    p.tail_hop_first = INT64_MAX;
    p.n_channels = 1;
    p.hop_first = 0;
    p.hop_count = 1;
    p.runs_per_channel = 1;
    p.run_len = 1;
    p.step = 1;
    p.spec = (float2 *)e->d_hop_out.p;
    if ((rc = single_hop_forward(e, p))) return rc;
    RC_HIP(hipMemcpyAsync(out_reim, e->d_hop_out.p, (size_t)N * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    RC_HIP(hipStreamSynchronize(e->stream));
    return RC_OK;
} catch (...) {
    return rc_catch();
}

int rc_engine_resynth(rc_engine *e, uint32_t channel, uint64_t hop, const float *samples, float *out) try {
    if (!e || !samples || !out) return fail(RC_EINVAL, "null argument");
    const uint32_t N = e->par.window_len;
    RC_HIP(hipSetDevice(e->device));
    int rc;
    if ((rc = e->d_hop_in.reserve(N * sizeof(float)))) return rc;
    if ((rc = e->d_hop_out.reserve((size_t)N * 2 * sizeof(float)))) return rc;
    if ((rc = e->d_ybuf.reserve((size_t)N * sizeof(float)))) return rc;
    RC_HIP(hipMemcpyAsync(e->d_hop_in.p, samples, N * sizeof(float), hipMemcpyHostToDevice, e->stream));
    rc::HopParams p = base_params(e);
    p.x = (const float *)e->d_hop_in.p;
    // present the window as hop `hop` of a stream whose sample 0 is hop*step
    p.in_origin = (int64_t)hop * p.step;
    p.in_len = N;
    p.xtail = p.x;
    p.tail_hop_first = INT64_MAX;
    p.ch_first = channel;
    p.n_channels = 1;
    p.hop_first = (int64_t)hop;
    p.hop_count = 1;
    p.runs_per_channel = 1;
    p.run_len = 1;
    p.spec = (float2 *)e->d_hop_out.p;
    p.ybuf = (float *)e->d_ybuf.p;
    if ((rc = single_hop_forward(e, p))) return rc;
    if (e->cfg.kernel) {
        e->h_spec.resize((size_t)N * 2);
        e->h_spec2.resize((size_t)N * 2);
        RC_HIP(hipMemcpyAsync(e->h_spec.data(), e->d_hop_out.p, (size_t)N * 2 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        RC_HIP(hipStreamSynchronize(e->stream));
        if (e->cfg.kernel(now_ms(e), e->h_spec.data(), e->h_spec2.data(), N, e->cfg.kernel_user) == 0)
            RC_HIP(hipMemcpyAsync(e->d_hop_out.p, e->h_spec2.data(), (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice, e->stream));
    }
    if (e->cfg.device_kernel != RC_DK_NONE) {
        if ((rc = e->d_spec2.reserve((size_t)N * 2 * sizeof(float)))) return rc;
        rc::DevKernelParams d{};
        d.in = (const float2 *)e->d_hop_out.p;
        d.out = e->cfg.device_kernel == RC_DK_SHIFT ? (float2 *)e->d_spec2.p : (float2 *)e->d_hop_out.p;
        d.log2n = (uint32_t)e->log2n;
        d.n = N;
        const bool gain = e->cfg.device_kernel == RC_DK_GAIN;  // (the offline paths fold it into amp)
        d.kind = gain ? (uint32_t)RC_DK_BAND : e->cfg.device_kernel;
        d.gain_in = e->cfg.dk_gain;
        d.gain_out = gain ? e->cfg.dk_gain : e->cfg.dk_gain_outside;
        d.lo_bin = gain ? 0u : e->cfg.dk_lo_bin;
        d.hi_bin = gain ? N : e->cfg.dk_hi_bin;
        d.shift = e->cfg.dk_shift_bins;
        d.hops_total = 1;
        RC_HIP(rc::launch_dev_kernel(d, e->stream));
        p.spec = d.out;
    }
    if ((rc = single_hop_resynth(e, p))) return rc;
    RC_HIP(hipMemcpyAsync(out, e->d_ybuf.p, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    RC_HIP(hipStreamSynchronize(e->stream));
    return RC_OK;
} catch (...) {
    return rc_catch();
}

// ================================ several devices, one process ====================================
// (include/rocoder_hip.h, rc_multi). The plan is rocoder_amd/distributed.py::shard_plan in C++; the multi-process
// layer (one process per GPU over torch.distributed / RCCL) and this one cut a job identically.
size_t rc_shard_plan(uint32_t channels, uint64_t total_windows, uint32_t n_devices, rc_shard *out, size_t cap) {
    if (channels == 0 || n_devices == 0) return 0;
    size_t n = 0;
    const unsigned __int128 total = (unsigned __int128)channels * total_windows;
    for (uint32_t r = 0; r < n_devices; ++r) {
        uint64_t lo = (uint64_t)(total * r / n_devices), hi = (uint64_t)(total * (r + 1) / n_devices);
        while (lo < hi) {
            const uint32_t c = (uint32_t)(lo / total_windows);
            const uint64_t w0 = lo % total_windows;
            uint64_t w1 = std::min<uint64_t>(total_windows, w0 + (hi - lo));
            uint32_t cc = 1;
            lo += w1 - w0;
            if (w0 == 0 && w1 == total_windows)  // merge the whole channels that follow
                while (hi - lo >= total_windows) {
                    ++cc;
                    lo += total_windows;
                }
            if (out && n < cap) out[n] = rc_shard{r, c, cc, w0, w1 - w0};
            ++n;
        }
    }
    return n;
}

namespace {
struct WorkerResult {
    int rc = RC_OK;
    std::string msg;
};
// One persistent host thread per listed device beyond the first (the caller's thread takes index 0 and the share of
// any thread that could not be started). A call posts ONE function for all indices, bumps the generation and waits
// for the count of outstanding shares to reach zero: no thread is created or joined per call.
struct MultiWorkers {
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    size_t pending = 0;
    bool quit = false;
    const std::function<int(size_t)> *fn = nullptr;
    std::vector<WorkerResult> res;
    std::vector<std::thread> th;  // th[i - 1] serves list index i (not joinable: that index has no thread)
    void run_one(size_t i) {
        int rc;
        try {
            rc = (*fn)(i);
        } catch (...) {
            rc = rc_catch();
        }
        res[i].rc = rc;
        res[i].msg = rc != RC_OK ? g_err : std::string();
    }
    void loop(size_t i) {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
            }
            run_one(i);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
    void start(size_t n) {
        res.assign(n, WorkerResult{});
        th.resize(n > 0 ? n - 1 : 0);
        for (size_t i = 1; i < n; ++i) {
            try {
                th[i - 1] = std::thread([this, i] { loop(i); });
            } catch (const std::system_error &) {  // the caller's thread runs that share
            }
        }
    }
    int run(const std::function<int(size_t)> &f) {
        const size_t n = res.size();
        size_t live = 0;
        for (auto &t : th) live += t.joinable() ? 1 : 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            fn = &f;
            pending = live;
            ++gen;
        }
        cv_go.notify_all();
        run_one(0);
        for (size_t i = 1; i < n; ++i)
            if (!th[i - 1].joinable()) run_one(i);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
        fn = nullptr;
        return RC_OK;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_go.notify_all();
        for (auto &t : th)
            if (t.joinable()) t.join();
        th.clear();
    }
};
}  // namespace

struct rc_multi {
    rc_config cfg{};
    rc_params par{};
    std::vector<int> dev;
    std::vector<rc_engine *> eng;
    std::vector<DevBuf> d_in, d_out;  // per listed device: the input spans and the shards of ALL its pieces of a job
    bool force_staging = false;       // rc_multi_set_staging
    MultiWorkers workers;
};

namespace {
// Runs fn(i) for every list index (persistent threads, MultiWorkers) and folds the results: the first failure, with
// its thread's rc_last_error text re-posted on the calling thread.
int for_each_device(rc_multi *m, const std::function<int(size_t)> &fn) {
    m->workers.run(fn);
    for (size_t i = 0; i < m->eng.size(); ++i)
        if (m->workers.res[i].rc != RC_OK)
            return fail(m->workers.res[i].rc, "device %d (list index %zu): %s", m->dev[i], i, m->workers.res[i].msg.c_str());
    return RC_OK;
}
// the calling thread's current device is put back on every way out of an rc_multi_* call
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};
// a device pointer handed to rc_multi_stretch_device must live on the root device of the list
int check_on_device(const void *ptr, int dev, const char *what) {
    if (!ptr) return RC_OK;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return fail(RC_EINVAL, "%s is not a device pointer", what);
    }
    if (at.type != hipMemoryTypeDevice || at.device != dev)
        return fail(RC_EINVAL, "%s lives on device %d (memory type %d), not on the root device %d of the list", what,
                    at.device, (int)at.type, dev);
    return RC_OK;
}
}  // namespace

int rc_multi_create(const rc_config *cfg, const int32_t *device_ids, uint32_t n_devices, rc_multi **out) try {
    if (!cfg || !out || !device_ids || n_devices == 0) return fail(RC_EINVAL, "null argument or empty device list");
    *out = nullptr;
    if (cfg->kernel)
        return fail(RC_EUNSUPPORTED, "a host frequency kernel sees its channel's hops in order (src/fft.rs:76-108): "
                                     "not on a job cut over devices; use one engine or a device kernel");
    DeviceRestore restore;
    std::unique_ptr<rc_multi, void (*)(rc_multi *)> m(new rc_multi, rc_multi_destroy);
    m->cfg = *cfg;
    if (int rc = rc_derive_params(cfg, &m->par)) return rc;
    for (uint32_t i = 0; i < n_devices; ++i) {
        rc_config c = *cfg;
        c.device = device_ids[i];
        rc_engine *e = nullptr;
        if (int rc = rc_engine_create(&c, &e)) return rc;
        m->dev.push_back(device_ids[i]);
        m->eng.push_back(e);
        // host form: every listed device's engine runs its copy workers at the same time
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        e->hp.max_workers = (int)std::max(1u, std::min<unsigned>(rc_engine::HostPipe::kWorkers, hw / n_devices));
    }
    m->d_in.resize(n_devices);
    m->d_out.resize(n_devices);
    for (uint32_t i = 0; i < n_devices; ++i)  // peer access where the hardware has it (copies work without)
        for (uint32_t j = 0; j < n_devices; ++j)
            if (m->dev[i] != m->dev[j]) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, m->dev[i], m->dev[j]) == hipSuccess && can) {
                    (void)hipSetDevice(m->dev[i]);
                    (void)hipDeviceEnablePeerAccess(m->dev[j], 0);  // (already enabled: an error we ignore)
                    (void)hipGetLastError();
                }
            }
    m->workers.start(n_devices);
    *out = m.release();
    return RC_OK;
} catch (...) {
    return rc_catch();
}

void rc_multi_destroy(rc_multi *m) {
    if (!m) return;
    m->workers.stop();
    DeviceRestore restore;
    for (size_t i = 0; i < m->eng.size(); ++i) {
        (void)hipSetDevice(m->dev[i]);
        (void)rc_engine_synchronize(m->eng[i]);
        if (i < m->d_in.size()) m->d_in[i].release();
        if (i < m->d_out.size()) m->d_out[i].release();
        rc_engine_destroy(m->eng[i]);
    }
    delete m;
}

uint32_t rc_multi_device_count(const rc_multi *m) { return m ? (uint32_t)m->eng.size() : 0; }

int rc_multi_set_staging(rc_multi *m, int force) {
    if (!m) return fail(RC_EINVAL, "null argument");
    m->force_staging = force != 0;
    return RC_OK;
}

namespace {
// one device's share of the job. src: where channel c's sample 0 lives (root device or host), with its stride;
// dst likewise for the output. host = both are host memory. Every piece of the share has its own region of the
// device's span / shard buffers, so nothing waits between pieces: copies in, compute and copies out are enqueued on
// the engine's stream back to back and the thread synchronises once at the end.
int multi_share(rc_multi *m, size_t i, const std::vector<rc_shard> &plan, bool host, int root_dev,
                const float *src, size_t src_stride, const float *const *src_rows, size_t in_len, float *dst,
                size_t dst_stride, float *const *dst_rows) {
    const int dev = m->dev[i];
    rc_engine *e = m->eng[i];
    const uint64_t wout = m->par.window_out_len;
    const uint32_t hpw = m->par.hops_per_window;
    RC_HIP(hipSetDevice(dev));
    hipStream_t s = e->stream;
    // a share on the root's own device reads and writes the caller's tensors in place (also when the root device is
    // listed more than once): nothing to copy. rc_multi_set_staging(m, 1) takes the copy path regardless.
    const bool in_place = !host && dev == root_dev && !m->force_staging;
    size_t need_in = 0, need_out = 0;
    if (!in_place)
        for (const rc_shard &sh : plan) {
            if (sh.device_index != i || sh.win_count == 0) continue;
            size_t lo, hi;
            input_span(m->par, sh.win_first, sh.win_first + sh.win_count, in_len, &lo, &hi);
            need_in += (size_t)sh.ch_count * (hi - lo);
            need_out += (size_t)sh.ch_count * (size_t)(sh.win_count * wout);
        }
    if (need_in || need_out) {
        if (int rc = m->d_in[i].reserve(std::max<size_t>(1, need_in) * sizeof(float))) return rc;
        if (int rc = m->d_out[i].reserve(std::max<size_t>(1, need_out) * sizeof(float))) return rc;
    }
    size_t off_in = 0, off_out = 0;
    std::vector<HostPiece> host_pieces;  // host form: all pieces of the share go through ONE upload / compute / download pipeline
    for (const rc_shard &sh : plan) {
        if (sh.device_index != i || sh.win_count == 0) continue;
        const size_t n_sh = (size_t)(sh.win_count * wout);
        if (in_place) {
            int rc = run_hops(e, src + (size_t)sh.ch_first * src_stride, src_stride, 0, (int64_t)in_len, sh.ch_first,
                              sh.ch_count, (int64_t)(sh.win_first * hpw), (int64_t)(sh.win_count * hpw),
                              dst + (size_t)sh.ch_first * dst_stride + (size_t)(sh.win_first * wout), dst_stride,
                              (int64_t)(sh.win_first * wout), s, true);
            if (rc) return rc;
            continue;
        }
        size_t lo, hi;
        input_span(m->par, sh.win_first, sh.win_first + sh.win_count, in_len, &lo, &hi);
        const size_t span = hi - lo;
        float *li = (float *)m->d_in[i].p + off_in, *lo_ = (float *)m->d_out[i].p + off_out;
        off_in += (size_t)sh.ch_count * span;
        off_out += (size_t)sh.ch_count * n_sh;
        if (host) {  // pinned staging slots + copy workers, not the runtime's pageable path (host_pipeline)
            host_pieces.push_back(HostPiece{sh.ch_first, sh.ch_count, sh.win_first, sh.win_first + sh.win_count, li, lo, span, lo_, n_sh});
            continue;
        }
        for (uint32_t c = 0; span && c < sh.ch_count; ++c)
            RC_HIP(hipMemcpyPeerAsync(li + (size_t)c * span, dev, src + (size_t)(sh.ch_first + c) * src_stride + lo,
                                      root_dev, span * sizeof(float), s));
        // the local buffer holds samples [lo, hi) of each channel: in_origin = lo, and the samples that exist end at
        // in_len (a window running past it is zero-padded by the engine as in the whole job)
        int rc = run_hops(e, li, span, (int64_t)lo, (int64_t)in_len - (int64_t)lo, sh.ch_first, sh.ch_count,
                          (int64_t)(sh.win_first * hpw), (int64_t)(sh.win_count * hpw), lo_, n_sh,
                          (int64_t)(sh.win_first * wout), s, true);
        if (rc) return rc;
        for (uint32_t c = 0; c < sh.ch_count; ++c)
            RC_HIP(hipMemcpyPeerAsync(dst + (size_t)(sh.ch_first + c) * dst_stride + (size_t)(sh.win_first * wout),
                                      root_dev, lo_ + (size_t)c * n_sh, dev, n_sh * sizeof(float), s));
    }
    if (host) return host_pipeline(e, host_pieces, src_rows, in_len, dst_rows);
    RC_HIP(hipStreamSynchronize(s));
    return check_device_error(e);
}
}  // namespace

int rc_multi_stretch_host(rc_multi *m, const float *const *in, size_t in_len, float *const *out, size_t out_cap,
                          size_t *out_len) try {
    if (!m || !in || !out) return fail(RC_EINVAL, "null argument");
    const uint64_t total_win = offline_windows(m->par, in_len);
    const size_t n_out = (size_t)(total_win * m->par.window_out_len);
    if (out_cap < n_out) return fail(RC_ECAPACITY, "out_cap %zu < %zu", out_cap, n_out);
    DeviceRestore restore;
    std::vector<rc_shard> plan(3 * m->eng.size());
    plan.resize(rc_shard_plan(m->cfg.channels, total_win, (uint32_t)m->eng.size(), plan.data(), plan.size()));
    int rc = for_each_device(m, [&](size_t i) {
        return multi_share(m, i, plan, true, 0, nullptr, 0, in, in_len, nullptr, 0, out);
    });
    if (rc == RC_OK && out_len) *out_len = n_out;
    return rc;
} catch (...) {
    return rc_catch();
}

int rc_multi_stretch_device(rc_multi *m, uint32_t root, const float *d_in, size_t in_stride, size_t in_len,
                            float *d_out, size_t out_stride, size_t out_cap, size_t *out_len, void *hip_stream) try {
    if (!m || !d_out || (!d_in && in_len)) return fail(RC_EINVAL, "null argument");
    if (root >= m->eng.size()) return fail(RC_EINVAL, "root %u is not an index of the device list", root);
    const uint64_t total_win = offline_windows(m->par, in_len);
    const size_t n_out = (size_t)(total_win * m->par.window_out_len);
    if (out_cap < n_out) return fail(RC_ECAPACITY, "out_cap %zu < %zu", out_cap, n_out);
    if (m->cfg.channels > 1 && (in_stride < in_len || out_stride < n_out))
        return fail(RC_EINVAL, "row strides (%zu, %zu) shorter than the rows (%zu, %zu)", in_stride, out_stride, in_len, n_out);
    DeviceRestore restore;
    const int root_dev = m->dev[root];
    if (int rc = check_on_device(in_len ? d_in : nullptr, root_dev, "d_in")) return rc;
    if (int rc = check_on_device(d_out, root_dev, "d_out")) return rc;
    RC_HIP(hipSetDevice(root_dev));
    if (hip_stream) RC_HIP(hipStreamSynchronize((hipStream_t)hip_stream));
    std::vector<rc_shard> plan(3 * m->eng.size());
    plan.resize(rc_shard_plan(m->cfg.channels, total_win, (uint32_t)m->eng.size(), plan.data(), plan.size()));
    int rc = for_each_device(m, [&](size_t i) {
        return multi_share(m, i, plan, false, root_dev, d_in, in_stride, nullptr, in_len, d_out, out_stride, nullptr);
    });
    if (rc == RC_OK && out_len) *out_len = n_out;
    return rc;
} catch (...) {
    return rc_catch();
}

int rc_calib_valu(int device, void *hip_stream, uint32_t launches, float *ms_per_launch, float *ns_per_inst) try {
    if (launches == 0 || (!ms_per_launch && !ns_per_inst)) return fail(RC_EINVAL, "bad argument");
    DeviceRestore restore;
    RC_HIP(hipSetDevice(device));
    hipDeviceProp_t pr{};
    RC_HIP(hipGetDeviceProperties(&pr, device));
    const int n_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    hipStream_t s = (hipStream_t)hip_stream;
    float *d = nullptr;
    RC_HIP(hipMalloc((void **)&d, sizeof(float)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Cleanup {
        float *&d;
        hipEvent_t &e0, &e1;
        ~Cleanup() {
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
            if (d) (void)hipFree(d);
        }
    } cleanup{d, e0, e1};
    RC_HIP(hipEventCreate(&e0));
    RC_HIP(hipEventCreate(&e1));
    RC_HIP(rc::launch_calib_valu(d, n_cu, s));  // (code object load, clocks)
    RC_HIP(hipEventRecord(e0, s));
    for (uint32_t i = 0; i < launches; ++i) RC_HIP(rc::launch_calib_valu(d, n_cu, s));
    RC_HIP(hipEventRecord(e1, s));
    RC_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    RC_HIP(hipEventElapsedTime(&ms, e0, e1));
    const float per = ms / (float)launches;
    if (ms_per_launch) *ms_per_launch = per;
    // per SIMD: 8 workgroups x 4 waves per CU over 4 SIMDs = 8 waves, each CALIB_ITERS x 16 instructions
    if (ns_per_inst) *ns_per_inst = per * 1.0e6f / (8.0f * (float)rc::CALIB_ITERS * 16.0f);
    return RC_OK;
} catch (...) {
    return rc_catch();
}

}  // extern "C"
