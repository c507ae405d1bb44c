// Microbenchmark of the in-register FFT passes of rc_kernels.hip (dev tool, not shipped):
// cycles per pass per wave and ns per pass per SIMD at 1 and 2 workgroups (of 4 waves) per CU.
#include "../rocoder_amd/csrc/rc_kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace rc;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int WHICH>
__global__ __launch_bounds__(256, 2) void pb(const float2 *wtab_, unsigned long long *cyc, float *sink, int iters) {
    using G = Geo<14>;
    GV2 wtab = (GV2)wtab_;
    float2 v[G::P];
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < G::P; ++q) v[q] = make_float2(1.0f + tid * 0.001f + q, 0.5f - q * 0.01f);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < iters; ++i) {
        if constexpr (WHICH == 0) run_pass<G, 8, 8, 12, false>(v, tid, wtab);       // fwd pass A (DIF)
        else if constexpr (WHICH == 1) run_pass<G, 8, 8, 12, true>(v, tid, wtab);   // inv pass A (DIT)
        else if constexpr (WHICH == 2) run_pass<G, 0, 0, 2, false>(v, tid, wtab);   // fwd pass C (const twiddles)
        else if constexpr (WHICH == 3) run_pass<G, 0, 0, 2, true>(v, tid, wtab);
#pragma unroll
        for (int q = 0; q < G::P; ++q) { asm volatile("" : "+v"(v[q].x), "+v"(v[q].y)); }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int q = 0; q < G::P; ++q) s += v[q].x + v[q].y;
    if (s == 1.2345f) sink[0] = s;
}

template <int WHICH>
void run(const char *name, const float2 *d_w, unsigned long long *d_cyc, float *d_sink, int ncu) {
    const int iters = 200;
    printf("%-28s", name);
    for (int bpc : {1, 2}) {
        int blocks = ncu * bpc;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(pb<WHICH>, dim3(blocks), dim3(256), 0, 0, d_w, d_cyc, d_sink, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(pb<WHICH>, dim3(blocks), dim3(256), 0, 0, d_w, d_cyc, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(blocks * 4);
        CHECK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        printf("  wg/CU=%d: %.0f cyc/pass/wave, %.1f ns/pass (wall/iters)", bpc, (double)c[c.size()/2] / iters, ms * 1e6 / iters);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    std::vector<float2> w(4096);
    for (int k = 0; k < 4096; ++k) { double a = -2.0 * M_PI * k / 8192.0; w[k] = make_float2((float)cos(a), (float)sin(a)); }
    float2 *d_w; unsigned long long *d_cyc; float *d_sink;
    CHECK(hipMalloc(&d_w, sizeof(float2) * 4096)); CHECK(hipMemcpy(d_w, w.data(), sizeof(float2) * 4096, hipMemcpyHostToDevice));
    CHECK(hipMalloc(&d_cyc, 8 * ncu * 2 * 4)); CHECK(hipMalloc(&d_sink, 4));
    printf("RC_PK=%d\n", RC_PK);
    run<0>("fwd pass A (DIF, 5 stages)", d_w, d_cyc, d_sink, ncu);
    run<1>("inv pass A (DIT, 5 stages)", d_w, d_cyc, d_sink, ncu);
    run<2>("fwd pass C (3 stages const)", d_w, d_cyc, d_sink, ncu);
    run<3>("inv pass C (3 stages const)", d_w, d_cyc, d_sink, ncu);
    return 0;
}
