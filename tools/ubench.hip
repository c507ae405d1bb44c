// Instruction-rate microbenchmarks for gfx950 (dev tool; not part of the product).
// For each op: cycles per wave-instruction as seen by one wave (s_memtime) at 1, 2, 4 waves/SIMD,
// and the aggregate rate per SIMD. Build: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 16;  // instructions per asm block, 8 independent registers

#define OPS8(OP) \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned long long *cyc, float *sink) {
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    float a = 1.0001f, b = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    f2 pa = {a, a}, pb = {b, b};
    unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
    unsigned ua = 0x9E3779B1u;
    __shared__ float2 lds[8192];
    unsigned laddr = threadIdx.x * 8;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    for (int i = 0; i < ITERS; ++i) {
        if constexpr (KIND == 0) {
#define OP(n) "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        } else if constexpr (KIND == 1) {
#define OP(n) "v_pk_fma_f32 %" #n ", %8, %9, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
#undef OP
        } else if constexpr (KIND == 2) {
#define OP(n) "v_add_f32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        } else if constexpr (KIND == 3) {
#define OP(n) "v_pk_add_f32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
#undef OP
        } else if constexpr (KIND == 4) {
#define OP(n) "v_pk_mul_f32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
#undef OP
        } else if constexpr (KIND == 5) {
#define OP(n) "v_mul_lo_u32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 6) {
#define OP(n) "v_mul_u32_u24 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 7) {
#define OP(n) "v_mad_u32_u24 %" #n ", %8, %" #n ", %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 8) {
#define OP(n) "v_xor_b32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 9) {
#define OP(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 10) {
#define OP(n) "v_alignbit_b32 %" #n ", %8, %" #n ", 9\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 11) {
#define OP(n) "v_sin_f32 %" #n ", %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        } else if constexpr (KIND == 12) {
#define OP(n) "v_sqrt_f32 %" #n ", %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        } else if constexpr (KIND == 13) {
#define OP(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n"
            unsigned long long w0 = u0, w1 = u1, w2 = u2, w3 = u3, w4 = u4, w5 = u5, w6 = u6, w7 = u7;
            asm volatile(OPS8(OP) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7) : "v"(ua), "v"(u0) : "vcc");
            u1 ^= (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^ (unsigned)w3 ^ (unsigned)w4 ^ (unsigned)w5 ^ (unsigned)w6 ^ (unsigned)w7;
#undef OP
        } else if constexpr (KIND == 14) {
#define OP(n) "v_xad_u32 %" #n ", %8, %" #n ", %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(ua));
#undef OP
        } else if constexpr (KIND == 15) {  // ds_write_b64 x16
#define OP(n) "ds_write_b64 %8, %" #n " offset:" #n "*2048\n"
            asm volatile(OPS8(OP) "s_waitcnt lgkmcnt(0)\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(laddr) : "memory");
#undef OP
        } else if constexpr (KIND == 16) {  // ds_read_b64 x16
#define OP(n) "ds_read_b64 %" #n ", %8 offset:" #n "*2048\n"
            asm volatile(OPS8(OP) "s_waitcnt lgkmcnt(0)\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(laddr) : "memory");
#undef OP
        } else if constexpr (KIND == 17) {  // v_mul_f32
#define OP(n) "v_mul_f32 %" #n ", %8, %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        } else if constexpr (KIND == 18) {  // dependent chain of v_fma_f32
            asm volatile("v_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\n"
                         "v_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\n"
                         "v_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\n"
                         "v_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\nv_fma_f32 %0, %1, %2, %0\n"
                         : "+v"(r0) : "v"(a), "v"(b));
        } else if constexpr (KIND == 19) {  // v_cos_f32
#define OP(n) "v_cos_f32 %" #n ", %" #n "\n"
            asm volatile(OPS8(OP) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
#undef OP
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)");
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y +
              (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + lds[threadIdx.x].x;
    if (s == 12345.678f) sink[0] = s;
}

template <int KIND>
void run(const char *name, int ncu, unsigned long long *d_cyc, float *d_sink) {
    printf("%-18s", name);
    for (int wps : {1, 2, 4}) {
        int blocks = ncu * wps;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_cyc, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> c(blocks * 4);
        CHECK(hipMemcpy(c.data(), d_cyc, c.size() * 8, hipMemcpyDeviceToHost));
        std::sort(c.begin(), c.end());
        double med = (double)c[c.size() / 2];
        double per_wave = med / (ITERS * (double)UNROLL);  // s_memtime ticks (100 MHz?) or cycles
        double instr_per_simd = (double)ITERS * UNROLL * wps;
        printf("  wps=%d: %.2f tick/instr/wave, %.3f ms, %.2f ns/instr/SIMD", wps, per_wave, ms,
               ms * 1e6 / instr_per_simd);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, ncu, prop.clockRate);
    unsigned long long *d_cyc;
    float *d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * ncu * 4 * 4));
    CHECK(hipMalloc(&d_sink, 4));
    run<0>("v_fma_f32", ncu, d_cyc, d_sink);
    run<18>("v_fma_f32 (dep)", ncu, d_cyc, d_sink);
    run<1>("v_pk_fma_f32", ncu, d_cyc, d_sink);
    run<2>("v_add_f32", ncu, d_cyc, d_sink);
    run<17>("v_mul_f32", ncu, d_cyc, d_sink);
    run<3>("v_pk_add_f32", ncu, d_cyc, d_sink);
    run<4>("v_pk_mul_f32", ncu, d_cyc, d_sink);
    run<5>("v_mul_lo_u32", ncu, d_cyc, d_sink);
    run<6>("v_mul_u32_u24", ncu, d_cyc, d_sink);
    run<7>("v_mad_u32_u24", ncu, d_cyc, d_sink);
    run<13>("v_mad_u64_u32", ncu, d_cyc, d_sink);
    run<8>("v_xor_b32", ncu, d_cyc, d_sink);
    run<14>("v_xad_u32", ncu, d_cyc, d_sink);
    run<9>("v_lshrrev_b32", ncu, d_cyc, d_sink);
    run<10>("v_alignbit_b32", ncu, d_cyc, d_sink);
    run<11>("v_sin_f32", ncu, d_cyc, d_sink);
    run<19>("v_cos_f32", ncu, d_cyc, d_sink);
    run<12>("v_sqrt_f32", ncu, d_cyc, d_sink);
    run<15>("ds_write_b64", ncu, d_cyc, d_sink);
    run<16>("ds_read_b64", ncu, d_cyc, d_sink);
    return 0;
}
