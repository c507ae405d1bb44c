// TEST INFRASTRUCTURE: drives the engine's host code (rocoder_amd/csrc/rc_engine.cpp built host-only over
// tests/c/hip_stub.cpp) through the C-ABI under AddressSanitizer + UBSan and, separately, ThreadSanitizer
// (rocoder_amd/csrc/host/sanitize.mk: engine_asan / engine_tsan; tools/run_sanitizers.sh). The stub's kernels compute
// nothing, so only status codes, lengths and the sanitizers' verdict are checked - numerics are the GPU tests' job.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rocoder_hip.h"

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, rc_last_error()); \
            exit(2);                                                             \
        }                                                                        \
    } while (0)

// the stub runtime's allocator stands in for "a tensor on the root device" (tests/c/hip_stub.cpp)
extern "C" int hipSetDevice(int);
extern "C" int hipMalloc(void **, size_t);
extern "C" int hipFree(void *);

static std::atomic<long> g_calls{0};
static int gain_kernel(uint64_t, const float *in, float *out, size_t n, void *user) {
    g_calls++;
    const float g = user ? *(const float *)user : 1.0f;
    for (size_t i = 0; i < 2 * n; ++i) out[i] = in[i] * g;
    return 0;
}
static int panicking_kernel(uint64_t, const float *, float *, size_t, void *) { return 1; }

static rc_config config(uint32_t N, float f, int p, uint32_t ch) {
    rc_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.window_len = N;
    c.factor = f;
    c.amplitude = 1.0f;
    c.pitch_multiple = p;
    c.sample_rate = 44100;
    c.channels = (uint16_t)ch;
    c.buffer_secs = 1.0f;
    c.seed = 7;
    return c;
}
static std::vector<std::vector<float>> input(uint32_t ch, size_t L) {
    std::vector<std::vector<float>> x(ch, std::vector<float>(L));
    for (uint32_t c = 0; c < ch; ++c)
        for (size_t i = 0; i < L; ++i) x[c][i] = 0.5f * sinf(0.01f * (float)(i + 100 * c));
    return x;
}

static void offline(uint32_t N, float f, int p, uint32_t ch, size_t L, rc_freq_kernel k, uint32_t kthreads) {
    rc_config c = config(N, f, p, ch);
    float gain = 2.0f;
    c.kernel = k;
    c.kernel_user = &gain;
    c.kernel_threads = kthreads;
    rc_engine *e = nullptr;
    CHECK(rc_engine_create(&c, &e) == RC_OK);
    auto x = input(ch, L);
    const size_t n_out = rc_offline_output_len(&c, L);
    std::vector<std::vector<float>> y(ch, std::vector<float>(n_out));
    std::vector<const float *> in;
    std::vector<float *> out;
    for (uint32_t i = 0; i < ch; ++i) {
        in.push_back(x[i].data());
        out.push_back(y[i].data());
    }
    size_t got = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(rc_engine_stretch_host(e, in.data(), L, out.data(), n_out, &got) == RC_OK);
        CHECK(got == n_out);
    }
    CHECK(rc_engine_stretch_host(e, in.data(), L, out.data(), n_out ? n_out - 1 : 0, &got) == RC_ECAPACITY || n_out == 0);
    {  // rows from rc_host_alloc: the pipeline's direct-DMA form (no staging slot), then one side pinned, one pageable
        std::vector<void *> blocks;
        std::vector<const float *> pin;
        std::vector<float *> pout;
        for (uint32_t i = 0; i < ch; ++i) {
            void *a = nullptr, *b = nullptr;
            CHECK(rc_host_alloc(std::max<size_t>(L, 1) * sizeof(float), &a) == RC_OK && a);
            CHECK(rc_host_alloc(std::max<size_t>(n_out, 1) * sizeof(float), &b) == RC_OK && b);
            if (L) memcpy(a, x[i].data(), L * sizeof(float));
            blocks.push_back(a);
            blocks.push_back(b);
            pin.push_back((const float *)a);
            pout.push_back((float *)b);
        }
        CHECK(rc_engine_stretch_host(e, pin.data(), L, pout.data(), n_out, &got) == RC_OK && got == n_out);
        CHECK(rc_engine_stretch_host(e, pin.data(), L, out.data(), n_out, &got) == RC_OK && got == n_out);
        CHECK(rc_engine_stretch_host(e, in.data(), L, pout.data(), n_out, &got) == RC_OK && got == n_out);
        for (void *b : blocks) CHECK(rc_host_free(b) == RC_OK);
        CHECK(rc_host_free(nullptr) == RC_OK);
    }
    CHECK(rc_engine_synchronize(e) == RC_OK);
    float ms[8];
    size_t nms = 0;
    CHECK(rc_engine_kernel_times(e, ms, 8, &nms) == RC_OK);
    rc_engine_destroy(e);
}

static void seam(uint32_t N, float f, int p, uint32_t ch, size_t L, uint32_t batch, bool view) {
    rc_config c = config(N, f, p, ch);
    c.max_batch_hops = batch;
    rc_engine *e = nullptr;
    CHECK(rc_engine_create(&c, &e) == RC_OK);
    rc_params P;
    CHECK(rc_engine_get_params(e, &P) == RC_OK);
    auto x = input(ch, L);
    std::vector<float> w(P.window_out_len);
    size_t total = 0;
    auto pull = [&](bool until_block) {
        for (;;) {
            bool any = false;
            for (uint32_t i = 0; i < ch; ++i) {
                if (rc_engine_is_done(e, i) == 1) return;
                size_t n = 0;
                const float *pv = nullptr;
                const int rc = view ? rc_engine_next_window_view(e, i, &pv, &n)
                                    : rc_engine_next_window(e, i, w.data(), w.size(), &n);
                if (rc == RC_WOULD_BLOCK) {
                    CHECK(until_block);
                    continue;
                }
                CHECK(rc == RC_OK && n == P.window_out_len);
                if (view) {
                    volatile float sink = pv[0] + pv[n - 1];  // the block is readable
                    (void)sink;
                }
                total += n;
                any = true;
            }
            if (!any) return;
        }
    };
    // producer and consumer are different threads, one after the other (one thread at a time per handle)
    std::thread([&] {
        for (uint32_t i = 0; i < ch; ++i) CHECK(rc_engine_push_input(e, i, x[i].data(), L / 2) == RC_OK);
    }).join();
    std::thread([&] { pull(true); }).join();
    std::thread([&] {
        for (uint32_t i = 0; i < ch; ++i) {
            CHECK(rc_engine_push_input(e, i, x[i].data() + L / 2, L - L / 2) == RC_OK);
            CHECK(rc_engine_close_input(e, i) == RC_OK);
            CHECK(rc_engine_push_input(e, i, x[i].data(), 1) == RC_EINVAL);  // closed
        }
    }).join();
    std::thread([&] { pull(false); }).join();
    CHECK(total == (size_t)ch * rc_offline_output_len(&c, L));
    size_t n = 0;
    CHECK(rc_engine_next_window(e, 0, w.data(), w.size(), &n) == RC_EINVAL);  // after is_done
    rc_engine_destroy(e);
}

// A closed job consumed in other orders than the processor's round-robin: one channel wholly before the next (the group
// batches of stream_enqueue_group give way to per-channel batches once the channels part), channels of different
// lengths (never a group), and a consumer that stops half way (blocks in flight at destroy).
static void seam_orders(uint32_t N, float f, uint32_t ch, size_t L, uint32_t batch, int order) {
    rc_config c = config(N, f, 1, ch);
    c.max_batch_hops = batch;
    rc_engine *e = nullptr;
    CHECK(rc_engine_create(&c, &e) == RC_OK);
    rc_params P;
    CHECK(rc_engine_get_params(e, &P) == RC_OK);
    auto x = input(ch, L);
    std::vector<size_t> len(ch, L);
    if (order == 1)
        for (uint32_t i = 0; i < ch; ++i) len[i] = L - (size_t)i * (L / 4);  // unequal channels
    for (uint32_t i = 0; i < ch; ++i) {
        CHECK(rc_engine_push_input(e, i, x[i].data(), len[i]) == RC_OK);
        CHECK(rc_engine_close_input(e, i) == RC_OK);
    }
    std::vector<size_t> total(ch, 0);
    size_t n = 0;
    const float *pv = nullptr;
    if (order == 2) {  // round-robin, abandoned half way
        const size_t half = rc_offline_output_len(&c, L) / 2;
        while (total[0] < half)
            for (uint32_t i = 0; i < ch; ++i) {
                CHECK(rc_engine_next_window_view(e, i, &pv, &n) == RC_OK);
                total[i] += n;
            }
        rc_engine_destroy(e);
        return;
    }
    // a few windows round-robin first (the job starts as a group), then channel after channel
    for (int w = 0; w < 3; ++w)
        for (uint32_t i = 0; i < ch; ++i)
            if (rc_engine_is_done(e, i) != 1) {
                CHECK(rc_engine_next_window_view(e, i, &pv, &n) == RC_OK);
                volatile float sink = pv[0] + pv[n - 1];
                (void)sink;
                total[i] += n;
            }
    for (uint32_t i = 0; i < ch; ++i)
        while (rc_engine_is_done(e, i) != 1) {
            CHECK(rc_engine_next_window_view(e, i, &pv, &n) == RC_OK);
            volatile float sink = pv[0] + pv[n - 1];
            (void)sink;
            total[i] += n;
        }
    for (uint32_t i = 0; i < ch; ++i) CHECK(total[i] == rc_offline_output_len(&c, len[i]));
    rc_engine_destroy(e);
}

static void multi(const std::vector<int32_t> &devs, uint32_t N, float f, int p, uint32_t ch, size_t L) {
    rc_config c = config(N, f, p, ch);
    rc_multi *m = nullptr;
    CHECK(rc_multi_create(&c, devs.data(), (uint32_t)devs.size(), &m) == RC_OK);
    CHECK(rc_multi_device_count(m) == devs.size());
    auto x = input(ch, L);
    const size_t n_out = rc_offline_output_len(&c, L);
    std::vector<std::vector<float>> y(ch, std::vector<float>(n_out));
    std::vector<const float *> in;
    std::vector<float *> out;
    for (uint32_t i = 0; i < ch; ++i) {
        in.push_back(x[i].data());
        out.push_back(y[i].data());
    }
    size_t got = 0;
    for (int rep = 0; rep < 3; ++rep) CHECK(rc_multi_stretch_host(m, in.data(), L, out.data(), n_out, &got) == RC_OK && got == n_out);
    // device form: "device" tensors on the list's root (hipMalloc of the stub)
    rc_engine *probe = nullptr;  // (only to make the root device current for the allocations below)
    rc_config c1 = c;
    c1.device = devs[devs.size() - 1];
    CHECK(rc_engine_create(&c1, &probe) == RC_OK);
    hipSetDevice(devs[devs.size() - 1]);
    float *d_in = nullptr, *d_out = nullptr;
    CHECK(hipMalloc((void **)&d_in, (size_t)ch * L * sizeof(float)) == 0);
    CHECK(hipMalloc((void **)&d_out, (size_t)ch * n_out * sizeof(float)) == 0);
    for (int staged = 0; staged < 2; ++staged) {
        CHECK(rc_multi_set_staging(m, staged) == RC_OK);
        CHECK(rc_multi_stretch_device(m, (uint32_t)devs.size() - 1, d_in, L, L, d_out, n_out, n_out, &got, nullptr) == RC_OK);
        CHECK(got == n_out);
    }
    CHECK(rc_multi_stretch_device(m, (uint32_t)devs.size(), d_in, L, L, d_out, n_out, n_out, &got, nullptr) == RC_EINVAL);
    if (L)  // (an empty input is never read: its pointer is not looked at)
        CHECK(rc_multi_stretch_device(m, 0, x[0].data(), L, L, d_out, n_out, n_out, &got, nullptr) == RC_EINVAL);  // host pointer
    hipFree(d_in);
    hipFree(d_out);
    rc_engine_destroy(probe);
    rc_multi_destroy(m);
}

int main() {
    CHECK(rc_abi_version() == RC_ABI_VERSION);
    // offline: fused paths, large windows, a window length that is not a power of two, negative pitch
    offline(1024, 4.0f, 1, 2, 30000, nullptr, 0);
    offline(16384, 8.0f, 3, 2, 200000, nullptr, 0);
    offline(65536, 32.0f, 1, 3, 300000, nullptr, 0);
    offline(3000, 2.0f, 1, 1, 20000, nullptr, 0);
    offline(2048, 2.0f, -2, 2, 30000, nullptr, 0);
    offline(256, 0.3f, 1, 1, 5000, nullptr, 0);
    offline(16384, 8.0f, 1, 2, 1500000, nullptr, 0);  // 12 M output samples per channel: three chunks in the host pipeline
    offline(4096, 0.3f, 1, 2, 9000000, nullptr, 0);   // more input than output: the uploads dominate
    offline(1024, 2.0f, 1, 1, 0, nullptr, 0);    // empty input
    offline(1024, 2.0f, 1, 1, 1023, nullptr, 0); // shorter than a window
    // host frequency kernel: the three-set pinned pipeline, 1 and 4 kernel threads, chunks smaller than the job
    offline(16384, 8.0f, 1, 4, 400000, gain_kernel, 1);
    offline(16384, 8.0f, 1, 4, 400000, gain_kernel, 4);
    offline(32768, 8.0f, 1, 3, 300000, gain_kernel, 4);
    offline(1000, 4.0f, 2, 2, 30000, gain_kernel, 2);
    offline(4096, 2.0f, -3, 2, 60000, gain_kernel, 2);
    offline(1024, 4.0f, 1, 2, 30000, panicking_kernel, 2);
    CHECK(g_calls.load() > 0);
    // the streaming seam: copies and views, small batches (look-ahead changes blocks often), one and several channels
    for (int view = 0; view < 2; ++view) {
        seam(1024, 4.0f, 1, 2, 90000, 8, view);
        seam(16384, 8.0f, 1, 2, 300000, 6, view);
        seam(4096, 2.0f, 3, 1, 90000, 12, view);
        seam(2048, 2.0f, -2, 2, 50000, 4, view);
        seam(256, 0.3f, 1, 1, 20000, 0, view);
        seam(16384, 8.0f, 1, 2, 2500000, 0, view);  // the default batch size: the ramp of the group batches (1/16, 1/4, 1)
        seam(1024, 8.0f, 1, 8, 400000, 0, view);    // eight channels in one group
    }
    for (int order = 0; order < 3; ++order) {
        seam_orders(4096, 4.0f, 2, 300000, 16, order);
        seam_orders(16384, 8.0f, 3, 600000, 0, order);
    }
    // several devices in one process: the persistent workers, host and device form, in place and staged
    multi({0, 0, 0}, 16384, 8.0f, 1, 2, 300000);
    multi({0, 1}, 16384, 8.0f, 1, 3, 1500000);  // shards of three chunks each through the host pipeline
    multi({0, 1, 2}, 16384, 8.0f, 3, 3, 200000);
    multi({2, 0, 0, 1, 3, 3, 1, 2}, 1024, 2.0f, 2, 1, 50000);
    multi({1}, 65536, 32.0f, 1, 8, 150000);
    multi({0, 0}, 1024, 2.0f, 1, 2, 0);       // empty input: one window of silence per channel
    multi({0, 1, 0}, 1024, 2.0f, 1, 1, 700);  // shorter than a window, more devices than windows
    // two engines driven from two threads at once (separate handles are independent)
    std::thread a([] { offline(4096, 4.0f, 1, 2, 120000, gain_kernel, 2); });
    std::thread b([] { seam(8192, 4.0f, 1, 2, 150000, 16, 1); });
    a.join();
    b.join();
    // argument errors come back as status codes
    rc_engine *e = nullptr;
    rc_config bad = config(1024, 4096.0f, 1, 1);  // step == 0
    CHECK(rc_engine_create(&bad, &e) == RC_EINVAL && e == nullptr);
    bad = config(1001, 2.0f, 1, 1);  // odd length
    CHECK(rc_engine_create(&bad, &e) != RC_OK);
    float ms = 0, ns = 0;
    CHECK(rc_calib_valu(0, nullptr, 2, &ms, &ns) == RC_OK);
    printf("OK engine host driver (%ld kernel calls)\n", g_calls.load());
    return 0;
}
