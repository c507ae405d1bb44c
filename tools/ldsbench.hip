// Does an LDS store burst overlap with VALU work? (dev tool; hipcc --offload-arch=gfx950 -O3 -o ldsbench ldsbench.hip)
// modes: 0 = VALU only, 1 = ds_write_b64 only, 2 = both in every wave (stores issued first),
//        3 = waves 0-3 of a 512-thread workgroup VALU, waves 4-7 stores, 4 = ds_read_b64 only,
//        5 = reads + VALU in every wave, 6 = stores + reads(no valu)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
    extern __shared__ float2 lds[];
    const int tid = threadIdx.x;
    v2f a[16];
    for (int i = 0; i < 16; ++i) a[i] = v2f{(float)tid + i, 1.0f};
    const v2f m = {0.999f, 1.001f}, c = {0.5f, 0.25f};
    const bool do_valu = MODE == 0 || MODE == 2 || MODE == 5 || (MODE == 3 && tid < 256);
    const bool do_st = MODE == 1 || MODE == 2 || MODE == 6 || (MODE == 3 && tid >= 256);
    const bool do_ld = MODE == 4 || MODE == 5 || MODE == 6;
    float2 *base = lds + (tid & 255) + (tid >> 8) * 4224;  // conflict-free: consecutive lanes
    v2f acc = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (do_st) {
#pragma unroll
            for (int q = 0; q < 16; ++q) base[q * 264] = make_float2(a[q].x, a[q].y);
        }
        if (do_ld) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                typedef const volatile v2f __attribute__((address_space(3))) *LV2;
                acc += *(LV2)(base + q * 264);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = __builtin_elementwise_fma(a[q], m, c);
        }
        __builtin_amdgcn_s_waitcnt(0);  // drain LDS ops of this iteration
        asm volatile("" ::: "memory");
    }
    float s = acc.x + acc.y;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    if (s == 12345.678f) out[tid] = s;
}
template <int MODE>
void run(const char *name, int wgs, int iters, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 2 * 4224 * sizeof(float2);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), lds, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per iteration per wave: 16 stores / 16 loads / 96 pk_fma
    printf("%-34s %8.3f ms  %7.1f ns/iter  (%.2f ns per 16 LDS ops or 96 pk_fma)\n", name, ms, ms * 1e6 / iters, ms * 1e6 / iters);
}
int main() {
    float *d;
    hipMalloc(&d, 4096);
    const int wgs = 256, iters = 20000;
    run<0>("valu only (96 pk_fma/iter)", wgs, iters, d);
    run<1>("ds_write_b64 only (16/iter)", wgs, iters, d);
    run<2>("both, same wave", wgs, iters, d);
    run<3>("waves 0-3 valu, 4-7 stores", wgs, iters, d);
    run<4>("ds_read_b64 only (16/iter)", wgs, iters, d);
    run<5>("reads + valu, same wave", wgs, iters, d);
    run<6>("stores + reads", wgs, iters, d);
    return 0;
}
