// tools/valurate.hip: issue rate of the VALU instruction kinds the pair stage is made of (gfx950), so that the
// counter-backed ceiling in DESIGN.md prices transcendental and 32-bit-multiply instructions at what they cost.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valurate tools/valurate.hip && /tmp/valurate
// Every kernel runs ITER x 64 independent instructions of one kind per wave, 256 threads x (8 * CUs) workgroups
// (2 waves per SIMD resident at a time x 4 rounds); prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int ITER = 4096, UNR = 16;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed, float seed2) {
    float a[UNR];
    v2f p[UNR];
    unsigned u[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) {
        a[i] = seed + 0.001f * (float)(threadIdx.x + i);
        p[i] = v2f{a[i], a[i] + 1.0f};
        u[i] = (unsigned)threadIdx.x * 2654435761u + (unsigned)i;
    }
    unsigned long long mask = __ballot(threadIdx.x & 1);
    const unsigned vm = (threadIdx.x & 1) ? ~0u : 0u;
    const v2f sp = {seed, seed2};  // (kernel arguments: an SGPR pair)
    for (int it = 0; it < ITER / 4; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < UNR; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
                if (KIND == 2) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(u[i]));
                if (KIND == 5) asm volatile("v_xor_b32 %0, %0, %0" : "+v"(u[i]));
                if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(u[i]));
                if (KIND == 7) asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(u[i]));
                if (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i]));
                if (KIND == 9) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(u[i]));
                if (KIND == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNR]), "s"(mask));
                if (KIND == 11) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(vm), "v"(u[(i + 1) % UNR]));
                if (KIND == 12) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(p[i]));
                if (KIND == 13) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i]));
                if (KIND == 14) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(u[i]));
                if (KIND == 15) asm volatile("v_and_or_b32 %0, %0, %1, 0.5" : "+v"(u[i]) : "s"(0x7fffff));
                if (KIND == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % UNR]));
                if (KIND == 17) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[i]) : "v"(u[(i + 1) % UNR]), "v"(u[(i + 2) % UNR]), "s"(mask));
                if (KIND == 18) asm volatile("v_add_f32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 19) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(p[i]) : "v"(p[(i + 1) % UNR]));
                if (KIND == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "s"(sp));
                if (KIND == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sp), "v"(p[(i + 1) % UNR]));
                if (KIND == 23) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(sp));
                if (KIND == 24) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p[i]) : "v"(p[(i + 1) % UNR]), "v"(p[(i + 2) % UNR]), "v"(p[(i + 3) % UNR]));
                if (KIND == 25) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p[i]) : "v"(p[(i + 1) % UNR]));
                if (KIND == 27) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % UNR]));
                if (KIND == 28) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % UNR]));
                if (KIND == 29) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) % UNR]));
                if (KIND == 26) asm volatile("v_cos_f32 %0, %0\n v_pk_fma_f32 %1, %1, %1, %1" : "+v"(a[i]), "+v"(p[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < UNR; ++i) s += a[i] + p[i].x + p[i].y + (float)u[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, float *d, int cus, double mhz) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int wgs = cus * 8;  // 2 workgroups of 4 waves per CU resident x 4 rounds ... (occupancy is not limited: all resident)
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 0.25f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(wgs), dim3(256), 0, 0, d, 1.0f, 0.25f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // per SIMD: wgs * 4 waves / (cus * 4 SIMDs) = 8 waves, each ITER * UNR instructions
    const double inst_per_simd = 8.0 * ITER * UNR;
    const double cycles = ms * 1e-3 * mhz * 1e6;
    printf("%-14s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, ms, cycles / inst_per_simd, mhz);
}

int main() {
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    const double mhz = pr.clockRate / 1000.0;
    printf("# %s, %d CUs, clockRate %.0f MHz (the chip may run below it: read the RATIOS)\n", pr.gcnArchName, cus, mhz);
    float *d;
    CK(hipMalloc(&d, 4));
    for (int rep = 0; rep < 1; ++rep) {
        run<0>("v_fma_f32", d, cus, mhz);
        run<1>("v_pk_fma_f32", d, cus, mhz);
        run<8>("v_pk_mul_f32", d, cus, mhz);
        run<2>("v_sin_f32", d, cus, mhz);
        run<3>("v_sqrt_f32", d, cus, mhz);
        run<4>("v_mul_lo_u32", d, cus, mhz);
        run<7>("v_mul_u32_u24", d, cus, mhz);
        run<9>("v_mad_u32_u24", d, cus, mhz);
        run<5>("v_xor_b32", d, cus, mhz);
        run<6>("v_cndmask vcc", d, cus, mhz);
        run<10>("v_cndmask e64 s", d, cus, mhz);
        run<17>("v_cndmask 3op", d, cus, mhz);
        run<11>("v_bfi_b32", d, cus, mhz);
        run<12>("v_pk_add_f32", d, cus, mhz);
        run<20>("v_pk_fma opsel", d, cus, mhz);
        run<21>("v_pk_fma sgpr", d, cus, mhz);
        run<22>("v_pk_fma sgpr+2v", d, cus, mhz);
        run<23>("v_pk_mul sgpr", d, cus, mhz);
        run<24>("v_pk_fma 3 srcs", d, cus, mhz);
        run<25>("v_pk_fma opsel+neg", d, cus, mhz);
        run<26>("v_cos + v_pk_fma", d, cus, mhz);
        run<27>("v_permlane32_swap", d, cus, mhz);
        run<28>("v_permlane16_swap", d, cus, mhz);
        run<29>("v_mov_dpp quad_perm", d, cus, mhz);
        run<13>("v_add_u32", d, cus, mhz);
        run<14>("v_lshrrev_b32", d, cus, mhz);
        run<15>("v_and_or_b32", d, cus, mhz);
        run<16>("v_mov_b32", d, cus, mhz);
        run<18>("v_add_f32", d, cus, mhz);
        run<19>("v_mul_f32", d, cus, mhz);
    }
    return 0;
}
