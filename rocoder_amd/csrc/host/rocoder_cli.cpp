// rocoder CLI twin over the gfx950 engine (the callers and data formats either
// side of the hot path). Same flags, defaults and semantics as the reference's src/main.rs:27-122,
// WAV reading as src/audio_files.rs:86-188 (hound: 8/16/24/32-bit int + 32-bit float; int->f32
// scaling of src/audio.rs:16-29), 32-bit-float WAV writing (src/audio_files.rs:203-226),
// --start/--duration clipping (src/audio.rs:57-63,115-130), --rotate-channels (:73-75), the
// hh:mm:ss.ss duration grammar (src/duration_parser.rs:5-25), the Stretcher / StretcherProcessor /
// AudioBus plumbing (src/stretcher.rs, src/stretcher_processor.rs, src/audio.rs:141-224) and the
// kernel hot-swapper (src/hotswapper.rs) for C-ABI kernels. Playback/recording (cpal) and MP3 are
// out of scope: -o is required and the input must be WAV.
#include "../../../include/rocoder_hip.h"

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ src/duration_parser.rs:5-25
// "hh:mm:ss.ss" -> milliseconds. Seconds parse as f32 (truncated to ms), minutes/hours as u64.
bool parse_u64(const std::string &s, uint64_t *out) {
    if (s.empty()) return false;
    size_t i = 0;
    if (s[0] == '+') i = 1;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (uint64_t)(s[i] - '0');
    }
    *out = v;
    return true;
}
bool parse_f32(const std::string &s, float *out) {
    if (s.empty()) return false;
    char *end = nullptr;
    const float v = strtof(s.c_str(), &end);
    if (end != s.c_str() + s.size()) return false;
    *out = v;
    return true;
}
bool parse_duration_ms(const std::string &str, uint64_t *ms) {
    std::vector<std::string> parts;  // duration_str.rsplit(":"): seconds first
    size_t end = str.size();
    for (;;) {
        const size_t pos = end == 0 ? std::string::npos : str.rfind(':', end - 1);
        if (pos == std::string::npos) {
            parts.push_back(str.substr(0, end));
            break;
        }
        parts.push_back(str.substr(pos + 1, end - pos - 1));
        end = pos;
    }
    if (parts.size() > 3 || parts.empty()) return false;
    float secs;
    if (!parse_f32(parts[0], &secs)) return false;
    const float msf = secs * 1000.0f;  // `as u64` saturates: negative / NaN -> 0
    uint64_t total = (msf > 0.0f) ? (msf >= 1.8446744e19f ? UINT64_MAX : (uint64_t)msf) : 0;
    if (parts.size() > 1) {
        uint64_t m;
        if (!parse_u64(parts[1], &m)) return false;
        total += m * 60 * 1000;
    }
    if (parts.size() > 2) {
        uint64_t h;
        if (!parse_u64(parts[2], &h)) return false;
        total += h * 3600 * 1000;
    }
    *ms = total;
    return true;
}

// ------------------------------------------------------------------ src/audio.rs
struct AudioSpec {
    uint16_t channels = 2;
    uint32_t sample_rate = 44100;
};
struct Audio {
    std::vector<std::vector<float>> data;
    AudioSpec spec;
    // src/audio.rs:57-63,115-130
    void clip_in_place(std::optional<uint64_t> start_ms, std::optional<uint64_t> dur_ms) {
        const size_t len = data.empty() ? 0 : data[0].size();
        size_t start = 0;
        if (start_ms) start = (size_t)((double)*start_ms / 1000.0 * (double)spec.sample_rate);
        size_t endp = len;
        if (dur_ms) endp = start + (size_t)((double)*dur_ms / 1000.0 * (double)spec.sample_rate);
        if (start > len || endp > len || start > endp)
            throw std::runtime_error("clip range out of bounds (the reference panics on the slice)");
        for (auto &c : data) c = std::vector<float>(c.begin() + start, c.begin() + endp);
    }
    void rotate_channels() {  // src/audio.rs:73-75: rotate_right(1)
        if (data.size() < 2) return;
        auto last = std::move(data.back());
        data.pop_back();
        data.insert(data.begin(), std::move(last));
    }
};

// ------------------------------------------------------------------ WAV (hound semantics)
struct ByteReader {
    FILE *f;
    bool read(void *dst, size_t n) { return fread(dst, 1, n, f) == n; }
    bool skip(size_t n) {
        std::vector<char> tmp(4096);
        while (n) {
            const size_t k = std::min(n, tmp.size());
            if (fread(tmp.data(), 1, k, f) != k) return false;
            n -= k;
        }
        return true;
    }
};
uint32_t le32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t le16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

Audio read_wav(FILE *f) {
    ByteReader r{f};
    unsigned char hdr[12];
    if (!r.read(hdr, 12) || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4))
        throw std::runtime_error("not a RIFF/WAVE stream");
    uint16_t fmt_tag = 0, channels = 0, bits = 0, block_align = 0;
    uint32_t rate = 0;
    bool have_fmt = false;
    std::vector<unsigned char> raw;
    for (;;) {
        unsigned char ch[8];
        if (!r.read(ch, 8)) break;
        const uint32_t len = le32(ch + 4);
        if (!memcmp(ch, "fmt ", 4)) {
            if (len < 16 || len > 65536) throw std::runtime_error("bad fmt chunk");  // (a hostile length is not an allocation)
            std::vector<unsigned char> b(len);
            if (!r.read(b.data(), len)) throw std::runtime_error("bad fmt chunk");
            fmt_tag = le16(&b[0]);
            channels = le16(&b[2]);
            rate = le32(&b[4]);
            block_align = le16(&b[12]);
            bits = le16(&b[14]);
            if (fmt_tag == 0xFFFE && len >= 26) fmt_tag = le16(&b[24]);  // extensible: subformat
            have_fmt = true;
            if (len & 1) r.skip(1);
        } else if (!memcmp(ch, "data", 4)) {
            if (!have_fmt) throw std::runtime_error("data chunk before fmt chunk");
            // streamed (stdin) WAV (length 0 or 0xFFFFFFFF): read to EOF; otherwise at most `len` bytes - in pieces, so
            // that a truncated file with a huge declared length costs the bytes it has, not the bytes it claims
            const bool to_eof = len == 0xFFFFFFFFu || len == 0;
            std::vector<unsigned char> buf(1 << 20);
            size_t left = to_eof ? SIZE_MAX : (size_t)len, k;
            while (left && (k = fread(buf.data(), 1, std::min(left, buf.size()), f)) > 0) {
                raw.insert(raw.end(), buf.begin(), buf.begin() + k);
                left -= k;
            }
            break;
        } else {
            if (!r.skip(len + (len & 1))) break;
        }
    }
    if (!have_fmt || channels == 0) throw std::runtime_error("no fmt chunk");
    (void)block_align;
    const size_t bps = bits / 8;
    if (bps == 0) throw std::runtime_error("unsupported bits per sample");
    // whole frames only: a file cut inside a frame would leave the channels ragged (the reference de-interleaves by
    // i % channels, audio_files.rs:39-43, and then indexes every channel up to the first one's length)
    const size_t n = raw.size() / bps / channels * channels;
    Audio a;
    a.spec.channels = channels;
    a.spec.sample_rate = rate;
    a.data.assign(channels, {});
    for (auto &c : a.data) c.reserve(n / channels + 1);
    const bool is_float = fmt_tag == 3, is_int = fmt_tag == 1;
    if (!((is_float && bits == 32) || (is_int && (bits == 8 || bits == 16 || bits == 24 || bits == 32))))
        throw std::runtime_error("Cannot read unsupported .wav format");  // audio_files.rs:184-186
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *p = raw.data() + i * bps;
        float v;
        if (is_float) {
            memcpy(&v, p, 4);
        } else if (bits == 8) {  // hound: u8 -> i8 by subtracting 128; from_i8: n / 127
            v = (float)((int)p[0] - 128) / 127.0f;
        } else if (bits == 16) {  // from_i16: n / 32767
            v = (float)(int16_t)le16(p) / 32767.0f;
        } else if (bits == 24) {  // from_i24: n / 8388608
            int32_t s = p[0] | (p[1] << 8) | (p[2] << 16);
            if (s & 0x800000) s |= ~0xFFFFFF;
            v = (float)s / 8388608.0f;
        } else {  // from_i32: n / 2147483647
            v = (float)(int32_t)le32(p) / 2147483647.0f;
        }
        a.data[i % channels].push_back(v);  // de-interleave, audio_files.rs:39-43
    }
    return a;
}

// 32-bit float WAV as hound writes it (audio_files.rs:68-81,203-226): WAVE_FORMAT_EXTENSIBLE header + interleaved frames
void write_wav_header(FILE *f, const AudioSpec &spec, uint64_t frames) {
    const uint32_t ch = spec.channels, rate = spec.sample_rate;
    const uint64_t data_bytes = frames * ch * 4;
    auto w32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    auto w16 = [&](uint16_t v) { fwrite(&v, 2, 1, f); };
    fwrite("RIFF", 1, 4, f);
    w32((uint32_t)std::min<uint64_t>(0xFFFFFFFFu, 4 + 8 + 40 + 8 + data_bytes));
    fwrite("WAVE", 1, 4, f);
    fwrite("fmt ", 1, 4, f);
    w32(40);  // WAVE_FORMAT_EXTENSIBLE, as hound writes 32-bit float
    w16(0xFFFE);
    w16((uint16_t)ch);
    w32(rate);
    w32(rate * ch * 4);
    w16((uint16_t)(ch * 4));
    w16(32);
    w16(22);
    w16(32);
    w32(0);  // channel mask
    const unsigned char guid_float[16] = {0x03, 0x00, 0x00, 0x00, 0x00, 0x00, 0x10, 0x00,
                                          0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71};
    fwrite(guid_float, 1, 16, f);
    fwrite("data", 1, 4, f);
    w32((uint32_t)std::min<uint64_t>(0xFFFFFFFFu, data_bytes));
}

void write_wav_f32(const std::string &path, const Audio &a) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    const uint32_t ch = a.spec.channels;
    const uint64_t frames = a.data.empty() ? 0 : a.data[0].size();
    write_wav_header(f, a.spec, frames);
    std::vector<float> row((size_t)ch * 4096);
    for (uint64_t i0 = 0; i0 < frames; i0 += 4096) {
        const uint64_t k = std::min<uint64_t>(4096, frames - i0);
        for (uint64_t i = 0; i < k; ++i)
            for (uint32_t c = 0; c < ch; ++c) row[i * ch + c] = a.data[c][i0 + i];  // interleave
        fwrite(row.data(), 4, (size_t)(k * ch), f);
    }
    fclose(f);
}

// The same file written as the windows arrive (README.md:74 "stream output to disk instead of RAM", src/main.rs:197-203
// collects the whole AudioBus first): the header goes out with the sizes still zero, every batch of windows that all
// channels have delivered is interleaved and appended, and the two size fields are patched at the end - byte for
// byte the file write_wav_f32 writes, with the resident set bounded by the queue depth instead of the output length.
struct WavStreamWriter {
    FILE *f = nullptr;
    AudioSpec spec;
    uint64_t frames = 0;
    std::vector<float> row;
    std::string path, tmp;
    // The samples go to "<path>.part" and the file takes its name in finish(): a run that fails half way (a device
    // error, a full disk) leaves no output file with empty size fields behind (the reference writes nothing before the
    // whole output exists, src/main.rs:197-203)
    WavStreamWriter(const std::string &path_, const AudioSpec &sp) : spec(sp), path(path_), tmp(path_ + ".part") {
        f = fopen(tmp.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot create " + tmp);
        write_wav_header(f, spec, 0);
    }
    ~WavStreamWriter() {
        if (f) {  // not finished: drop the partial file
            fclose(f);
            (void)std::remove(tmp.c_str());
        }
    }
    // chans[c]: the next samples of channel c (equal lengths)
    void append(const std::vector<const float *> &chans, size_t n) {
        const uint32_t ch = spec.channels;
        row.resize((size_t)ch * std::min<size_t>(n, 4096));
        for (size_t i0 = 0; i0 < n; i0 += 4096) {
            const size_t k = std::min<size_t>(4096, n - i0);
            for (size_t i = 0; i < k; ++i)
                for (uint32_t c = 0; c < ch; ++c) row[i * ch + c] = chans[c][i0 + i];  // interleave
            if (fwrite(row.data(), 4, k * ch, f) != k * ch) throw std::runtime_error("write failed (disk full?)");
        }
        frames += n;
    }
    void finish() {
        if (fseek(f, 0, SEEK_SET) != 0) throw std::runtime_error("cannot seek in the output file");
        write_wav_header(f, spec, frames);
        if (fclose(f) != 0) {
            f = nullptr;
            (void)std::remove(tmp.c_str());
            throw std::runtime_error("closing the output file failed");
        }
        f = nullptr;
        if (std::rename(tmp.c_str(), path.c_str()) != 0) {
            (void)std::remove(tmp.c_str());
            throw std::runtime_error("cannot move " + tmp + " to " + path);
        }
    }
};

// ------------------------------------------------------------------ bounded(cap) channel of windows
struct WindowQueue {
    explicit WindowQueue(size_t cap) : cap(std::max<size_t>(1, cap)) {}
    void send(std::vector<float> v) {
        std::unique_lock<std::mutex> lk(m);
        cv_space.wait(lk, [&] { return q.size() < cap || abandoned; });
        if (abandoned) return;
        q.push_back(std::move(v));
        cv_item.notify_one();
    }
    void close() {
        std::lock_guard<std::mutex> lk(m);
        closed = true;
        cv_item.notify_all();
    }
    void abandon() {  // the Receiver was dropped: pending and later sends are discarded (never block the sender)
        std::lock_guard<std::mutex> lk(m);
        abandoned = true;
        q.clear();
        cv_space.notify_all();
    }
    // 0 = got item, 1 = timeout, 2 = disconnected
    int recv_timeout(std::vector<float> *out, std::chrono::milliseconds to) {
        std::unique_lock<std::mutex> lk(m);
#ifdef __SANITIZE_THREAD__
        // gcc 11's libtsan does not intercept pthread_cond_clockwait (what wait_for on the steady clock calls): it then
        // believes the mutex stays locked across the wait and reports every access behind it. The system-clock form
        // goes through pthread_cond_timedwait, which it knows (sanitizer builds only: host/sanitize.mk).
        if (!cv_item.wait_until(lk, std::chrono::system_clock::now() + to, [&] { return !q.empty() || closed; })) return 1;
#else
        if (!cv_item.wait_for(lk, to, [&] { return !q.empty() || closed; })) return 1;
#endif
        if (q.empty()) return 2;
        *out = std::move(q.front());
        q.pop_front();
        cv_space.notify_one();
        return 0;
    }
    size_t cap;
    std::mutex m;
    std::condition_variable cv_item, cv_space;
    std::deque<std::vector<float>> q;
    bool closed = false, abandoned = false;
};

// ------------------------------------------------------------------ src/hotswapper.rs + fft.rs:76-108
// A stack of loaded kernel libraries; the newest is called, a failing one is popped and the call
// retried with the previous one (or no-op). C-ABI kernels: `int apply(uint64_t, const float*,
// float*, size_t, void*)`. A source file (.c/.cc/.cpp) is compiled with the system compiler and
// watched every 100 ms; a prebuilt .so is loaded as is.
struct KernelStack {
    std::mutex m;
    std::vector<std::pair<void *, rc_freq_kernel>> libs;  // kept loaded for fallback (fft.rs:21)
    std::string src;
    std::thread watcher;
    std::atomic<bool> stop{false};
    struct timespec last_mtime {};  // nanosecond mtime: an edit in the same second as the last build counts

    static bool is_source(const std::string &p) {
        auto ends = [&](const char *s) { const size_t n = strlen(s); return p.size() >= n && p.compare(p.size() - n, n, s) == 0; };
        return ends(".c") || ends(".cc") || ends(".cpp") || ends(".cxx");
    }
    bool load_so(const std::string &so) {
        void *h = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            fprintf(stderr, "WARN failed to load kernel library %s: %s\n", so.c_str(), dlerror());
            return false;
        }
        auto fn = (rc_freq_kernel)dlsym(h, "apply");
        if (!fn) {
            fprintf(stderr, "WARN kernel library %s has no `apply` symbol\n", so.c_str());
            dlclose(h);
            return false;
        }
        std::lock_guard<std::mutex> lk(m);
        libs.emplace_back(h, fn);
        fprintf(stderr, "INFO Got new kernel\n");  // fft.rs:79
        return true;
    }
    bool compile_and_load() {  // hotswapper.rs:52-86 with cc instead of rustc
        char tmpl[] = "/tmp/rocoder_kernel_XXXXXX.so";
        const int fd = mkstemps(tmpl, 3);
        if (fd >= 0) close(fd);
        const bool cxx = src.size() > 2 && src.substr(src.size() - 2) != ".c";
        const std::string cmd = std::string(cxx ? "c++" : "cc") + " -O3 -shared -fPIC -w -o " + tmpl + " '" + src + "' -lm 2>&1";
        FILE *p = popen(cmd.c_str(), "r");
        std::string outp;
        char buf[512];
        while (p && fgets(buf, sizeof buf, p)) outp += buf;
        const int rc = p ? pclose(p) : -1;
        if (rc != 0) {
            fprintf(stderr, "================ kernel compilation failed ================\n%s", outp.c_str());
            fprintf(stderr, "WARN Failed to compile library for file %s\n", src.c_str());  // hotswapper.rs:39
            return false;
        }
        const bool ok = load_so(tmpl);
        unlink(tmpl);
        return ok;
    }
    void start(const std::string &path) {
        src = path;
        if (!is_source(path)) {
            load_so(path);
            return;
        }
        struct stat st;
        if (stat(path.c_str(), &st) == 0) last_mtime = st.st_mtim;
        compile_and_load();  // hotswapper.rs:17 (synchronous first attempt)
        watcher = std::thread([this] {
            while (!stop.load()) {
                std::this_thread::sleep_for(std::chrono::milliseconds(100));  // hotswapper.rs:12,29
                struct stat st2;
                if (stat(src.c_str(), &st2) == 0 && (st2.st_mtim.tv_sec != last_mtime.tv_sec ||
                                                     st2.st_mtim.tv_nsec != last_mtime.tv_nsec)) {
                    last_mtime = st2.st_mtim;
                    compile_and_load();
                }
            }
        });
    }
    ~KernelStack() {
        stop.store(true);
        if (watcher.joinable()) watcher.join();
    }
    static int trampoline(uint64_t t, const float *in, float *out, size_t n, void *user) {
        auto *self = (KernelStack *)user;
        for (;;) {
            rc_freq_kernel fn = nullptr;
            void *handle = nullptr;
            {
                std::lock_guard<std::mutex> lk(self->m);
                if (self->libs.empty()) return 1;  // no-op
                handle = self->libs.back().first;
                fn = self->libs.back().second;
            }
            if (fn(t, in, out, n, nullptr) == 0) return 0;
            fprintf(stderr, "WARN kernel panicked, retrying with last or noop.\n");  // fft.rs:101
            // pop the library that FAILED, not whatever is on top now: the watcher thread may have pushed a newer one
            // between the call and this line (the reference cannot race here: same thread, fft.rs:78-106)
            std::lock_guard<std::mutex> lk(self->m);
            for (size_t i = self->libs.size(); i-- > 0;)
                if (self->libs[i].first == handle) {
                    self->libs.erase(self->libs.begin() + (long)i);
                    break;
                }
        }
    }
};

static bool g_timing = false;  // ROCODER_CLI_TIMING

// ------------------------------------------------------------------ src/stretcher.rs over the engine
struct Engine {
    rc_engine *h = nullptr;
    std::mutex m;  // one thread at a time per handle
    ~Engine() { rc_engine_destroy(h); }
};
struct Stretcher {
    AudioSpec spec;
    std::shared_ptr<Engine> eng;
    uint32_t channel;
    rc_params par;
    // the Receiver<Vec<f32>> side: main sends the whole channel as one chunk (main.rs:148)
    std::deque<std::vector<float>> input;
    bool input_closed = false;

    bool is_done() {  // stretcher.rs:78-80
        std::lock_guard<std::mutex> lk(eng->m);
        return rc_engine_is_done(eng->h, channel) == 1;
    }
    size_t channel_bound() { return rc_engine_channel_bound(eng->h); }  // stretcher.rs:82-85
    std::vector<float> next_window() {                                    // stretcher.rs:87-121
        std::vector<float> out(par.window_out_len);
        for (;;) {
            size_t n = 0;
            int rc;
            {
                std::lock_guard<std::mutex> lk(eng->m);
                rc = rc_engine_next_window(eng->h, channel, out.data(), out.size(), &n);
            }
            if (rc == RC_OK) {
                out.resize(n);
                return out;
            }
            if (rc != RC_WOULD_BLOCK) throw std::runtime_error(std::string("rocoder_hip: ") + rc_last_error());
            std::lock_guard<std::mutex> lk(eng->m);  // self.input.recv(), stretcher.rs:125
            if (!input.empty()) {
                rc_engine_push_input(eng->h, channel, input.front().data(), input.front().size());
                input.pop_front();
                // file input: main queued the whole channel and dropped its sender before the processor started
                // (main.rs:148-150), so an empty queue IS the disconnect. Telling the engine now instead of at the
                // next shortfall lets it batch ahead (an open channel is computed one queue-bound at a time);
                // the windows are the same either way
                if (input.empty() && input_closed) rc_engine_close_input(eng->h, channel);
            } else {
                rc_engine_close_input(eng->h, channel);  // Err(_): Sender dropped, stretcher.rs:129-132
            }
        }
    }
};

// src/stretcher_processor.rs:26-89
struct StretcherProcessor {
    std::vector<std::pair<std::shared_ptr<WindowQueue>, Stretcher>> channels;
    std::atomic<bool> shutdown{false}, finished{false};
    std::thread th;
    std::string error;
    double t_next = 0, t_send = 0;
    // min_depth: file-to-file runs deepen the queues (the results do not depend on the depth). The reference's
    // bound is ceil(window seconds / buffer seconds) = 1 at the defaults: two thread hand-offs per window, which
    // its ~1 ms of CPU work per window hides and the engine's ~1 us per window does not (2.0 of 2.3 s on a
    // 600 s stereo file)
    std::vector<std::shared_ptr<WindowQueue>> make(std::vector<Stretcher> st, size_t min_depth = 1) {
        std::vector<std::shared_ptr<WindowQueue>> rx;
        for (auto &s : st) {
            auto q = std::make_shared<WindowQueue>(std::max(s.channel_bound(), min_depth));  // bounded(channel_bound())
            rx.push_back(q);
            channels.emplace_back(q, std::move(s));
        }
        return rx;
    }
    void start() {
        th = std::thread([this] {
            try {
                bool running = true;
                while (running) {
                    if (shutdown.load()) break;  // handle_control_messages (:57-62)
                    for (auto &c : channels) {
                        if (c.second.is_done()) {  // :64-68
                            fprintf(stderr, "INFO stretch process completed\n");
                            running = false;
                            break;
                        }
                        if (!g_timing) {
                            c.first->send(c.second.next_window());  // :69
                        } else {  // ROCODER_CLI_TIMING: where the processor thread's time goes
                            const auto t0 = std::chrono::steady_clock::now();
                            auto w = c.second.next_window();
                            const auto t1 = std::chrono::steady_clock::now();
                            c.first->send(std::move(w));
                            const auto t2 = std::chrono::steady_clock::now();
                            t_next += std::chrono::duration<double, std::milli>(t1 - t0).count();
                            t_send += std::chrono::duration<double, std::milli>(t2 - t1).count();
                        }
                    }
                }
            } catch (const std::exception &ex) {
                error = ex.what();
            }
            if (g_timing) fprintf(stderr, "[timing] processor: next_window %.1f ms, send %.1f ms\n", t_next, t_send);
            for (auto &c : channels) c.first->close();
            finished.store(true);  // :72
        });
    }
    void join() {
        if (th.joinable()) th.join();
    }
};

// AudioBus::into_audio (src/audio.rs:152-172): 5 ms recv_timeout polling drain
Audio into_audio(const AudioSpec &spec, std::vector<std::shared_ptr<WindowQueue>> &chs, size_t expected = 0) {
    Audio a;
    a.spec = spec;
    a.data.assign(chs.size(), {});
    for (auto &c : a.data) c.reserve(expected);
    std::vector<bool> closed(chs.size(), false);
    for (;;) {
        size_t disconnected = 0;
        for (size_t i = 0; i < chs.size(); ++i) {
            if (closed[i]) {
                disconnected++;
                continue;
            }
            std::vector<float> chunk;
            const int rc = chs[i]->recv_timeout(&chunk, std::chrono::milliseconds(5));
            if (rc == 0) a.data[i].insert(a.data[i].end(), chunk.begin(), chunk.end());
            else if (rc == 2) {
                closed[i] = true;
                disconnected++;
            }
        }
        if (disconnected == chs.size()) break;
    }
    return a;
}

// The same drain, written out as it arrives: a window goes to the file as soon as every channel has delivered its
// next one (the processor sends them round robin, src/stretcher_processor.rs:63-70). Returns the frames written.
uint64_t drain_to_wav(WavStreamWriter &w, std::vector<std::shared_ptr<WindowQueue>> &chs) {
    const size_t C = chs.size();
    std::vector<std::deque<std::vector<float>>> pend(C);
    std::vector<size_t> used(C, 0);  // samples of pend[c].front() already written
    std::vector<bool> closed(C, false);
    auto flush = [&] {
        for (;;) {
            size_t n = SIZE_MAX;
            for (size_t c = 0; c < C; ++c) n = std::min(n, pend[c].empty() ? 0 : pend[c].front().size() - used[c]);
            if (n == 0 || n == SIZE_MAX) return;
            std::vector<const float *> rows(C);
            for (size_t c = 0; c < C; ++c) rows[c] = pend[c].front().data() + used[c];
            w.append(rows, n);
            for (size_t c = 0; c < C; ++c) {
                used[c] += n;
                if (used[c] == pend[c].front().size()) {
                    pend[c].pop_front();
                    used[c] = 0;
                }
            }
        }
    };
    for (;;) {
        size_t disconnected = 0;
        for (size_t i = 0; i < C; ++i) {
            if (closed[i]) {
                disconnected++;
                continue;
            }
            if (pend[i].size() >= 4) continue;  // this channel is ahead: let the others catch up first
            std::vector<float> chunk;
            const int rc = chs[i]->recv_timeout(&chunk, std::chrono::milliseconds(5));
            if (rc == 0) pend[i].push_back(std::move(chunk));
            else if (rc == 2) {
                closed[i] = true;
                disconnected++;
            }
        }
        flush();
        if (disconnected == C) break;
        bool stuck = true;  // every open channel is waiting on a closed, empty one: nothing more can pair up
        for (size_t i = 0; i < C; ++i)
            if (!closed[i] && pend[i].size() < 4) stuck = false;
        if (stuck) break;
    }
    return w.frames;
}

struct Opt {  // src/main.rs:27-122
    size_t window_len = 16384;
    uint64_t buffer_ms = 1000;
    float factor = 1.0f;
    int pitch_multiple = 1;
    float amplitude = 1.0f;
    std::optional<std::string> input;
    bool rotate_channels = false;
    std::optional<std::string> freq_kernel;
    std::optional<std::string> device_kernel;  // not in the reference: curated on-GPU kernels
    int kernel_threads = 0;
    uint64_t fade_ms = 1000;
    std::optional<uint64_t> start_ms, duration_ms;
    std::optional<std::string> output;
    uint64_t seed = 0;  // not in the reference: thread_rng there
    int device = 0;
    std::vector<int32_t> devices;  // not in the reference: --devices a,b,... shards one job over several GPUs (rc_multi_*)
};

void usage() {
    fprintf(stderr,
            "rocoder (gfx950 engine)\nA live-codeable phase vocoder.\n\nUSAGE:\n    rocoder [FLAGS] [OPTIONS]\n\n"
            "FLAGS:\n        --rotate-channels    Rotate the input audio channels\n    -h, --help\n\nOPTIONS:\n"
            "    -a, --amplitude <amplitude>        Output amplitude [default: 1]\n"
            "    -b, --buffer <buffer-dur>          The maximum amount of audio to process ahead of time [default: 1]\n"
            "    -d, --duration <duration>          Duration to use from input audio (hh:mm:ss.ss)\n"
            "    -x, --fade <fade>                  Fade (playback only; accepted and ignored with -o) [default: 1]\n"
            "    -f, --factor <factor>              Stretch factor [default: 1]\n"
            "        --freq-kernel <freq-kernel>    Path to a frequency kernel (.c/.cpp source or .so exporting `apply`)\n"
            "    -i, --input <input>                A .wav file; '-' for stdin\n"
            "    -o, --output <output>              Output .wav file path. Uses 32-bit float.\n"
            "    -p, --pitch_multiple <n>           A non-zero integer pitch multiplier [default: 1]\n"
            "    -s, --start <start>                Start time in input audio (hh:mm:ss.ss)\n"
            "    -w, --window <window-len>          Processing window size [default: 16384]\n"
            "        --device-kernel <spec>         On-GPU frequency kernel: gain:<g> | band:<lo>:<hi>:<g_in>:<g_out> | shift:<bins>\n"
            "        --kernel-threads <n>           Host threads calling --freq-kernel (channels in parallel; needs a re-entrant kernel)\n"
            "        --seed <u64>                   Phase-source seed (the reference uses an unseeded thread_rng)\n"
            "        --device <n>                   HIP device ordinal [default: 0]\n"
            "        --devices <a,b,...>            Shard the job over several GPUs (offline; not with --freq-kernel)\n");
}

int run(int argc, char **argv) {
    Opt o;
    auto need = [&](int &i) -> std::string {
        if (i + 1 >= argc) throw std::runtime_error(std::string("missing value for ") + argv[i]);
        return argv[++i];
    };
    auto dur = [&](const std::string &s) -> uint64_t {
        uint64_t ms;
        if (!parse_duration_ms(s, &ms)) throw std::runtime_error("Invalid duration specification: " + s);
        return ms;
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-h" || a == "--help") { usage(); return 0; }
        else if (a == "--parse-duration") {  // test hook: the 8 cases of duration_parser.rs:32-39
            uint64_t ms;
            const std::string v = need(i);
            if (parse_duration_ms(v, &ms)) printf("%llu\n", (unsigned long long)ms);
            else printf("error\n");
            return 0;
        } else if (a == "--decode-wav") {  // test hook: WAV -> raw f32 planar [channels][frames]
            const std::string in = need(i), outp = need(i);
            FILE *f = in == "-" ? stdin : fopen(in.c_str(), "rb");
            if (!f) throw std::runtime_error("cannot open " + in);
            Audio au = read_wav(f);
            FILE *g = fopen(outp.c_str(), "wb");
            for (auto &c : au.data) fwrite(c.data(), 4, c.size(), g);
            fclose(g);
            printf("%u %u %zu\n", au.spec.channels, au.spec.sample_rate, au.data.empty() ? 0 : au.data[0].size());
            return 0;
        }
        else if (a == "-w" || a == "--window") o.window_len = (size_t)strtoull(need(i).c_str(), nullptr, 10);
        else if (a == "-b" || a == "--buffer") o.buffer_ms = dur(need(i));
        else if (a == "-f" || a == "--factor") o.factor = strtof(need(i).c_str(), nullptr);
        else if (a == "-p" || a == "--pitch_multiple" || a == "--pitch-multiple") o.pitch_multiple = atoi(need(i).c_str());
        else if (a == "-a" || a == "--amplitude") o.amplitude = strtof(need(i).c_str(), nullptr);
        else if (a == "-i" || a == "--input") o.input = need(i);
        else if (a == "--rotate-channels") o.rotate_channels = true;
        else if (a == "--freq-kernel") o.freq_kernel = need(i);
        else if (a == "--device-kernel") o.device_kernel = need(i);
        else if (a == "--kernel-threads") o.kernel_threads = atoi(need(i).c_str());
        else if (a == "-x" || a == "--fade") o.fade_ms = dur(need(i));
        else if (a == "-s" || a == "--start") o.start_ms = dur(need(i));
        else if (a == "-d" || a == "--duration") o.duration_ms = dur(need(i));
        else if (a == "-o" || a == "--output") o.output = need(i);
        else if (a == "--seed") o.seed = strtoull(need(i).c_str(), nullptr, 0);
        else if (a == "--device") o.device = atoi(need(i).c_str());
        else if (a == "--devices") {
            const std::string v = need(i);
            for (size_t b = 0; b <= v.size();) {
                const size_t c = std::min(v.find(',', b), v.size());
                if (c == b) throw std::runtime_error("bad --devices " + v);
                o.devices.push_back((int32_t)atoi(v.substr(b, c - b).c_str()));
                b = c + 1;
            }
        }
        else throw std::runtime_error("unknown argument " + a);
    }
    if (!o.input) throw std::runtime_error("recording from an input device (no -i) is not supported: pass -i <file.wav> or -i -");
    if (!o.output) throw std::runtime_error("live playback (no -o) is not supported: pass -o <file.wav>");
    if (o.pitch_multiple < -128 || o.pitch_multiple > 127) throw std::runtime_error("pitch_multiple must fit an i8");

    // ROCODER_CLI_TIMING=1: wall time of each phase on stderr (dev aid)
    const bool timing = g_timing = getenv("ROCODER_CLI_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[timing] %-14s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // load_audio (src/main.rs:162-188)
    FILE *f = *o.input == "-" ? stdin : fopen(o.input->c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + *o.input);
    Audio audio = read_wav(f);
    if (f != stdin) fclose(f);
    if (o.start_ms || o.duration_ms) audio.clip_in_place(o.start_ms, o.duration_ms);
    if (o.rotate_channels) audio.rotate_channels();
    const size_t total_samples_len = audio.data.empty() ? 0 : audio.data[0].size();
    const AudioSpec spec = audio.spec;
    lap("read input");

    KernelStack kernels;
    if (o.freq_kernel) kernels.start(*o.freq_kernel);

    rc_config cfg{};
    cfg.struct_size = sizeof cfg;
    cfg.window_len = (uint32_t)o.window_len;
    cfg.factor = o.factor;
    cfg.amplitude = o.amplitude;
    cfg.pitch_multiple = o.pitch_multiple;
    cfg.sample_rate = spec.sample_rate;
    cfg.channels = spec.channels;
    cfg.buffer_secs = (float)o.buffer_ms / 1000.0f;
    cfg.seed = o.seed;
    cfg.device = o.device;
    cfg.kernel = o.freq_kernel ? &KernelStack::trampoline : nullptr;
    cfg.kernel_user = &kernels;
    cfg.kernel_threads = (uint32_t)std::max(0, o.kernel_threads);
    if (o.device_kernel) {
        std::vector<std::string> f;
        size_t b = 0;
        for (;;) {
            const size_t c = o.device_kernel->find(':', b);
            f.push_back(o.device_kernel->substr(b, c == std::string::npos ? c : c - b));
            if (c == std::string::npos) break;
            b = c + 1;
        }
        if (f[0] == "gain" && f.size() == 2) {
            cfg.device_kernel = RC_DK_GAIN;
            cfg.dk_gain = strtof(f[1].c_str(), nullptr);
        } else if (f[0] == "band" && f.size() == 5) {
            cfg.device_kernel = RC_DK_BAND;
            cfg.dk_lo_bin = (uint32_t)strtoul(f[1].c_str(), nullptr, 10);
            cfg.dk_hi_bin = (uint32_t)strtoul(f[2].c_str(), nullptr, 10);
            cfg.dk_gain = strtof(f[3].c_str(), nullptr);
            cfg.dk_gain_outside = strtof(f[4].c_str(), nullptr);
        } else if (f[0] == "shift" && f.size() == 2) {
            cfg.device_kernel = RC_DK_SHIFT;
            cfg.dk_shift_bins = atoi(f[1].c_str());
        } else {
            throw std::runtime_error("bad --device-kernel " + *o.device_kernel);
        }
    }
    if (!o.devices.empty()) {
        // several GPUs: the whole job at once through the multi-device entry (windows of a channel are independent
        // given the phase source, so the devices share nothing but the input; include/rocoder_hip.h, rc_multi_*).
        // Same samples as the Stretcher / StretcherProcessor loop below produces (tests/test_gpu_cli.py).
        rc_multi *m = nullptr;
        if (rc_multi_create(&cfg, o.devices.data(), (uint32_t)o.devices.size(), &m) != RC_OK)
            throw std::runtime_error(std::string("rocoder_hip: ") + rc_last_error());
        lap("engine create");
        const size_t in_len = audio.data[0].size(), cap = rc_offline_output_len(&cfg, in_len);
        Audio out;
        out.spec = spec;
        out.data.assign(spec.channels, std::vector<float>(cap));
        std::vector<const float *> ins;
        std::vector<float *> outs;
        for (uint32_t c = 0; c < spec.channels; ++c) {
            ins.push_back(audio.data[c].data());
            outs.push_back(out.data[c].data());
        }
        size_t n = 0;
        const int rc = rc_multi_stretch_host(m, ins.data(), in_len, outs.data(), cap, &n);
        rc_multi_destroy(m);
        if (rc != RC_OK) throw std::runtime_error(std::string("rocoder_hip: ") + rc_last_error());
        for (auto &c : out.data) c.resize(n);
        lap("stretch");
        write_wav_f32(*o.output, out);
        lap("write output");
        return 0;
    }
    auto eng = std::make_shared<Engine>();
    if (rc_engine_create(&cfg, &eng->h) != RC_OK) throw std::runtime_error(std::string("rocoder_hip: ") + rc_last_error());
    lap("engine create");

    // one Stretcher per channel, fed the whole channel as one chunk (src/main.rs:133-153)
    std::vector<Stretcher> stretchers;
    for (uint32_t c = 0; c < spec.channels; ++c) {
        Stretcher s;
        s.spec = spec;
        s.eng = eng;
        s.channel = c;
        rc_engine_get_params(eng->h, &s.par);
        s.input.push_back(std::move(audio.data[c]));
        s.input_closed = true;  // nothing else will be sent
        stretchers.push_back(std::move(s));
    }
    const size_t expected_total_samples = (size_t)((float)total_samples_len * o.factor);  // main.rs:154
    StretcherProcessor proc;
    auto bus = proc.make(std::move(stretchers), 64);
    proc.start();
    // handle_result (src/main.rs:190-211). ROCODER_CLI_COLLECT=1 keeps the reference's order (collect the whole bus in
    // memory, then write): the A/B partner of the streamed file in tests/test_gpu_cli.py
    if (getenv("ROCODER_CLI_COLLECT")) {
        Audio out = into_audio(spec, bus, expected_total_samples + o.window_len);
        proc.join();
        if (!proc.error.empty()) throw std::runtime_error(proc.error);
        lap("stretch");
        write_wav_f32(*o.output, out);
        lap("write output");
        return 0;
    }
    WavStreamWriter wav(*o.output, spec);
    drain_to_wav(wav, bus);
    for (auto &q : bus) q->abandon();  // (only matters after a `stuck` exit: the processor must not block on a full queue)
    proc.join();
    if (!proc.error.empty()) throw std::runtime_error(proc.error);
    wav.finish();
    lap("stretch+write");
    return 0;
}

}  // namespace

// ROCODER_CLI_TIMING: the resident-set high-water mark of THIS program image (VmHWM of /proc/self/status: getrusage's
// ru_maxrss also carries the launching process's pages from between fork and exec)
static void report_peak_rss() {
    if (!getenv("ROCODER_CLI_TIMING")) return;
    FILE *f = fopen("/proc/self/status", "r");
    if (!f) return;
    char line[256];
    while (fgets(line, sizeof line, f))
        if (strncmp(line, "VmHWM:", 6) == 0) fprintf(stderr, "[timing] peak_rss_kib %ld\n", strtol(line + 6, nullptr, 10));
    fclose(f);
}

int main(int argc, char **argv) {
    try {
        const int rc = run(argc, argv);
        report_peak_rss();
        return rc;
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
