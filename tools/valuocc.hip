// tools/valuocc.hip: how many waves per SIMD does it take to keep the VALU issuing? Chains of v_pk_fma_f32 with DEP
// independent chains per wave (1 = every instruction depends on the one before), 1..4 waves per SIMD (occupancy capped
// through the LDS allocation). Prints time per wave-instruction per SIMD relative to the saturated rate.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valuocc tools/valuocc.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 1 << 16;

template <int DEP, int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    extern __shared__ float lds[];
    v2f p[DEP];
#pragma unroll
    for (int i = 0; i < DEP; ++i) p[i] = v2f{seed + (float)threadIdx.x, seed + (float)i};
    if (seed == 123.0f) lds[threadIdx.x] = seed;
    for (int it = 0; it < ITER / DEP / 8; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < DEP; ++i) {
                if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %0" : "+v"(p[i].x));
                if (KIND == 2) asm volatile("v_sin_f32 %0, %0" : "+v"(p[i].x));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < DEP; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int DEP, int KIND>
double run(float *d, int cus, int waves_per_simd) {
    // one workgroup = 4 waves = one wave per SIMD; `waves_per_simd` workgroups fit a CU (LDS = 160 KB / that, minus a bit)
    const size_t lds = (size_t)(160 * 1024 / waves_per_simd) - 1024;
    CK(hipFuncSetAttribute((const void *)k<DEP, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int wgs = cus * waves_per_simd;  // exactly one resident round
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<DEP, KIND>), dim3(wgs), dim3(256), lds, 0, d, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<DEP, KIND>), dim3(wgs), dim3(256), lds, 0, d, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e6 / ((double)ITER * waves_per_simd);  // ns per wave-instruction per SIMD
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    float *d;
    CK(hipMalloc(&d, 4));
    printf("# ns per wave-instruction per SIMD; columns = waves per SIMD 1 2 3 4 8\n");
    const int occ[5] = {1, 2, 3, 4, 8};
#define ROW(DEP, KIND, name) { printf("%-28s", name); for (int o = 0; o < 5; ++o) printf(" %7.3f", run<DEP, KIND>(d, cus, occ[o])); printf("\n"); }
    ROW(1, 0, "v_pk_fma_f32 dependent");
    ROW(2, 0, "v_pk_fma_f32 2 chains");
    ROW(4, 0, "v_pk_fma_f32 4 chains");
    ROW(16, 0, "v_pk_fma_f32 16 chains");
    ROW(1, 1, "v_add_f32 dependent");
    ROW(4, 1, "v_add_f32 4 chains");
    ROW(16, 1, "v_add_f32 16 chains");
    ROW(1, 2, "v_sin_f32 dependent");
    ROW(4, 2, "v_sin_f32 4 chains");
    return 0;
}
