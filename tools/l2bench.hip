// L2-resident table read rate per CU: 8 B/lane vs 16 B/lane vs 4 B/lane loads (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int W>  // bytes per lane
__global__ __launch_bounds__(256) void rd(const float *tab, size_t tab_floats, float *sink, int iters) {
    float acc = 0.f;
    const unsigned lane = threadIdx.x;
    // each block walks the whole table (shared by all blocks -> L2 hits), 256 threads x W bytes per step
    for (int it = 0; it < iters; ++it) {
        size_t base = ((size_t)blockIdx.x * 7919 + it * 131) % 64 * (256 * W / 4);
#pragma unroll 8
        for (size_t off = base; off + 256 * W / 4 <= tab_floats; off += 64 * 256 * W / 4) {
            if constexpr (W == 4) acc += tab[off + lane];
            else if constexpr (W == 8) { float2 v = *reinterpret_cast<const float2 *>(tab + off + 2 * lane); acc += v.x + v.y; }
            else { float4 v = *reinterpret_cast<const float4 *>(tab + off + 4 * lane); acc += v.x + v.y + v.z + v.w; }
        }
    }
    if (acc == 1.2345f) sink[0] = acc;
}
template <int W> void run(const float *d, size_t n, float *sink, int ncu) {
    const int iters = 64;
    for (int bpc : {1, 2, 4}) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(rd<W>, dim3(ncu * bpc), dim3(256), 0, 0, d, n, sink, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(rd<W>, dim3(ncu * bpc), dim3(256), 0, 0, d, n, sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // bytes read per block per iteration: n*4/64 (every 64th chunk)
        double bytes = (double)ncu * bpc * iters * (double)(n * 4 / 64);
        printf("  W=%2d wg/CU=%d: %.2f TB/s total, %.1f B/ns/CU\n", W, bpc, bytes / ms / 1e9, bytes / ms / 1e6 / ncu);
    }
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    size_t n = (size_t)4 << 20;  // 16 MB table: L2 (32 MB aggregate) / MALL resident
    float *d, *sink; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemset(d, 0, n * 4)); CHECK(hipMalloc(&sink, 4));
    printf("table %zu MB\n", n * 4 >> 20);
    run<4>(d, n, sink, ncu); run<8>(d, n, sink, ncu); run<16>(d, n, sink, ncu);
    n = (size_t)64 << 10;  // 256 KB table: L2-hot in every XCD
    printf("table %zu KB\n", n * 4 >> 10);
    run<4>(d, n, sink, ncu); run<8>(d, n, sink, ncu); run<16>(d, n, sink, ncu);
    return 0;
}
