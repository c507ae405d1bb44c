// LDS throughput per CU by access width (dev tool; hipcc --offload-arch=gfx950 -O3 -o ldsbench2 ldsbench2.hip)
// one 512-thread workgroup per CU, consecutive lanes -> consecutive units (conflict-free), every wave the same op
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    v4f val = {(float)tid, 1.f, 2.f, 3.f};
    typedef volatile v2f __attribute__((address_space(3))) *L2;
    typedef volatile v4f __attribute__((address_space(3))) *L4;
    typedef volatile float __attribute__((address_space(3))) *L1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (MODE == 0) *((L2)lds + tid + 512 * q) = v2f{val.x, val.y};
            if (MODE == 1) *((L4)lds + tid + 512 * q) = val;
            if (MODE == 2) { v2f r = *((L2)lds + tid + 512 * q); acc.x += r.x; acc.y += r.y; }
            if (MODE == 3) { v4f r = *((L4)lds + tid + 512 * q); acc += r; }
            if (MODE == 4) *((L1)lds + tid + 512 * q) = val.x;
            if (MODE == 5) { acc.x += *((L1)lds + tid + 512 * q); }
        }
        __builtin_amdgcn_s_waitcnt(0);
        asm volatile("" ::: "memory");
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 12345.678f) out[tid] = s;
}
template <int MODE>
void run(const char *name, int bytes, float *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, wgs = 256;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(512), 512 * 16 * 16, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double b = (double)iters * 16 * 512 * bytes;  // per CU
    printf("%-16s %8.3f ms  %7.1f B/ns/CU  (= B/clk at 1 GHz; divide by the clock in GHz)\n", name, ms, b / (ms * 1e6));
}
int main() {
    float *d;
    hipMalloc(&d, 4096);
    run<4>("ds_write_b32", 4, d);
    run<0>("ds_write_b64", 8, d);
    run<1>("ds_write_b128", 16, d);
    run<5>("ds_read_b32", 4, d);
    run<2>("ds_read_b64", 8, d);
    run<3>("ds_read_b128", 16, d);
    return 0;
}
