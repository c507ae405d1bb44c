// dev microbenchmark (round 6): what pinned host memory costs on the GPU box, and how fast one thread copies a window
// out of a pinned block the DMA engine wrote. hipcc -O2 -o tools/bin/hostbench tools/hostbench.cpp
#include <hip/hip_runtime.h>

#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x)                                                         \
    do {                                                              \
        hipError_t e_ = (x);                                          \
        if (e_ != hipSuccess) {                                       \
            printf("%s: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                 \
        }                                                             \
    } while (0)

static void copy_pf(float *dst, const float *src, size_t n) {
    const size_t chunk = 1024;
    const char *nxt = (const char *)(src + n);
    for (size_t i = 0; i < n; i += chunk) {
        for (size_t b = 0; b < chunk * 4; b += 64) __builtin_prefetch(nxt + i * 4 + b, 0, 1);
        memcpy(dst + i, src + i, chunk * 4);
    }
}

int main() {
    CK(hipSetDevice(0));
    void *warm = nullptr;
    CK(hipHostMalloc(&warm, 1 << 20, hipHostMallocDefault));
    for (int rep = 0; rep < 2; ++rep)
        for (size_t mb : {1, 4, 16, 64, 128}) {
            void *p = nullptr;
            double t0 = now();
            CK(hipHostMalloc(&p, mb << 20, hipHostMallocDefault));
            double t1 = now();
            CK(hipHostFree(p));
            double t2 = now();
            void *q = aligned_alloc(4096, mb << 20);
            memset(q, 0, mb << 20);
            double t3 = now();
            CK(hipHostRegister(q, mb << 20, hipHostRegisterDefault));
            double t4 = now();
            CK(hipHostUnregister(q));
            double t5 = now();
            free(q);
            printf("%4zu MiB: hipHostMalloc %.2f ms, hipHostFree %.2f ms, hipHostRegister (touched pages) %.2f ms, unregister %.2f ms\n", mb,
                   (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
        }
    // D2H into a pinned block, then one thread hands it out in 64 KiB windows
    const size_t bytes = (size_t)64 << 20, win = 16384;
    float *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 1, bytes));
    CK(hipHostMalloc((void **)&h, bytes + win * 4, hipHostMallocDefault));
    float *out = (float *)malloc(win * 4);
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
            double t1 = now();
            double acc = 0;
            for (size_t o = 0; o < bytes / 4; o += win) {
                if (mode == 0) memcpy(out, h + o, win * 4);
                else if (mode == 1) copy_pf(out, h + o, win);
                else acc += h[o] + h[o + win - 1];
                acc += out[0];
            }
            double t2 = now();
            printf("%s: D2H 64 MiB %.2f ms (%.1f GB/s), hand-out %.2f ms = %.2f Gsamples/s [%g]\n",
                   mode == 0 ? "memcpy" : mode == 1 ? "memcpy+prefetch" : "view", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9,
                   (t2 - t1) * 1e3, bytes / 4 / (t2 - t1) / 1e9, acc);
        }
    // the same block as anonymous memory with transparent huge pages, populated, then registered
    for (int huge = 0; huge < 2; ++huge)
        for (int rep = 0; rep < 2; ++rep) {
            const size_t len = bytes + ((size_t)2 << 20);
            double t0 = now();
            char *raw = (char *)mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (raw == MAP_FAILED) return 2;
            char *al = (char *)(((uintptr_t)raw + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1));
            if (huge) madvise(al, bytes, MADV_HUGEPAGE);
            double t1 = now();
            if (madvise(al, bytes, 23 /* MADV_POPULATE_WRITE */) != 0) memset(al, 0, bytes);
            double t2 = now();
            CK(hipHostRegister(al, bytes, hipHostRegisterDefault));
            double t3 = now();
            float *hh = (float *)al;
            CK(hipMemcpy(hh, d, bytes, hipMemcpyDeviceToHost));
            double t4 = now();
            CK(hipMemcpy(hh, d, bytes, hipMemcpyDeviceToHost));
            double t5 = now();
            double acc = 0;
            for (size_t o = 0; o < bytes / 4; o += win) {
                memcpy(out, hh + o, win * 4);
                acc += out[0];
            }
            double t6 = now();
            CK(hipHostUnregister(al));
            double t7 = now();
            munmap(raw, len);
            double t8 = now();
            printf("mmap%s 64 MiB: map %.2f ms, populate %.2f ms, register %.2f ms, first D2H %.2f ms, second D2H %.2f ms (%.1f GB/s), "
                   "memcpy hand-out %.2f ms = %.2f Gsamples/s, unregister %.2f ms, munmap %.2f ms [%g]\n", huge ? "+THP" : "", (t1 - t0) * 1e3,
                   (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, bytes / (t5 - t4) / 1e9, (t6 - t5) * 1e3,
                   bytes / 4 / (t6 - t5) / 1e9, (t7 - t6) * 1e3, (t8 - t7) * 1e3, acc);
        }
    // fresh destination per window (what a Vec<f32> per next_window costs): 64 KiB malloc + copy + free
    {
        double t1 = now();
        double acc = 0;
        for (size_t o = 0; o < bytes / 4; o += win) {
            float *v = (float *)malloc(win * 4);
            memcpy(v, h + o, win * 4);
            acc += v[5];
            free(v);
        }
        double t2 = now();
        printf("malloc+memcpy+free per window: %.2f ms = %.2f Gsamples/s [%g]\n", (t2 - t1) * 1e3, bytes / 4 / (t2 - t1) / 1e9, acc);
    }
    return 0;
}
