// N = 4096 with the default hanning window: hopw_kernel - ONE WAVE PER HOP. The M = N/2 = 2048 complex points are
// 64 lanes x 32 registers, so every one of the four exchanges of hop4_kernel's structure stays inside the wave: no
// s_barrier at all, the LDS executes a wave's instructions in order and only the compiler needs a fence. hop4's
// arithmetic otherwise (tests/dev/proto_w.py is the index model: it replays the four exchanges with the address
// expressions below, checks every hand-over and counts bank conflicts - none):
//   forward DIT on positions P = brev11(n), passes of (5, 2, 4) stages:
//     F1  stages 0..4  on the 32 registers (P0..P4; constants only), lane t = low sample bits (coalesced loads),
//         the analysis window fused into stage 0
//     F2  stages 5..6  on registers P4..P8, lane = (P0..P3, P9, P10), base twiddle W_128^(P0..P3)
//     F3  stages 7..10 on two sets of 16 registers (P7..P10): lane tau holds residues tau and 128 - tau, so every
//         (j, M - j) bin pair sits in one lane and the pair stage runs in registers (lane 0: residues 0 and 64, which
//         pair with themselves - hop4's re-deal)
//   inverse DIT on Q = brev11(bin), passes of (4, 2, 5) stages: I1 stages 0..3 on the sets (constants), I2 stages 4..5
//     on registers Q4..Q8, I3 stages 6..10 on registers Q6..Q10 with lane t = Q0..Q5 (coalesced stores)
//   exchanges E1..E4: two rounds each over a half-size buffer (1083 float2 per wave); the round is a position bit that
//     is a register bit on both sides (P4, P6 = the set, Q4 = the set, Q8), so a round is 16 stores + 16 loads per
//     lane; one weight per position bit and exchange (address = lane part + immediate), conflict-free for the 16-lane
//     ds_write_b64 groups and the 32-lane ds_read_b64 groups
// 168 VGPRs and 12 KB of LDS per wave: three waves per SIMD. The first hop of a run is recomputed for its tail (no
// seam hand-over); a caller-supplied window, pitch != 1 ... all take this kernel except a non-default window (generic).
// (v_mad_u32_u16 for the upper phase mantissa, rc_dev.hpp: -1.2 % on hop4_kernel, but +1..3 % on the 512 / 1024 / 2048
// kernels of this file and flat at 4096 / 8192 - same-box A/B, profiles/README.md round 4: not used here)
#define RC_MAD16 0
#include "rc_dit.hpp"

namespace rc {
namespace {

// rc_engine_create lays hann_rot out as [part][Geo<LOG2N>::T][4] (hop_geometry); the kernels below index it with the
// lane counts they are written for (64 threads; 128 at N = 8192). A build with another RC_PMAX must not pass silently.
static_assert(Geo<9>::T == 64 && Geo<10>::T == 64 && Geo<11>::T == 64 && Geo<12>::T == 64 && Geo<13>::T == 128,
              "wave-local kernels: hann_rot part strides assume RC_PMAX = 32");

constexpr int HW_BUF = 1088;                      // exchange buffer, float2 slots (1083 used)
constexpr int HW_TA = HW_BUF;                     // [65] W_2048^r, r <= 64
constexpr int HW_TR = HW_TA + 72;                 // [65] W_4096^r, r <= 64 (r = 64: lane 0's second residue, as i W)
constexpr int HW_TB = HW_TR + 72;                 // [16] W_128^l
constexpr int HW_TC = HW_TB + 16;                 // [16] W_64^l
constexpr int HW_TH = HW_TC + 16;                 // [256] window / envelope rotations: lane t at 2 t (+ 128: envelope)
constexpr int HOPW_LDS_FLOAT2 = HW_TH + 256;      // 12 160 B

struct HannK32 {
    float c[32], s[32];
};
// value(i) = 0.5 + c[q] cos(beta) + s[q] sin(beta) for sample i = stride q + 2 t + e, beta = 2 pi (2 t + e) / (len - 1)
constexpr HannK32 make_hann_w(double amp, int len, int count, int stride = 128) {
    HannK32 k{};
    for (int q = 0; q < 32; ++q) {
        const double a = q < count ? 2.0 * CX_PI * (double)stride * q / (double)(len - 1) : 0.0;
        k.c[q] = (float)(-amp * cx_cos(a));
        k.s[q] = (float)(amp * cx_sin(a));
    }
    return k;
}
__device__ constexpr HannK32 HANN_W12 = make_hann_w(0.5, 4096, 32);
__device__ constexpr HannK32 HANN_E12 = make_hann_w(HANN_ENV_AMP, 2048, 16);
constexpr double HANN_KAPPA12 = -0.25 / 4096.0;   // -1/(4N): the scale pair_regs_pk4 leaves out (a power of two)
__device__ constexpr HannK32 HANN_W12K = make_hann_w(0.5 * HANN_KAPPA12, 4096, 32);
// N = 8192 (hopw2_kernel: 128 threads, sample i = 256 q + 2 t + e)
__device__ constexpr HannK32 HANN_W13 = make_hann_w(0.5, 8192, 32, 256);
__device__ constexpr HannK32 HANN_E13 = make_hann_w(HANN_ENV_AMP, 4096, 16, 256);
constexpr double HANN_KAPPA13 = -0.25 / 8192.0;
__device__ constexpr HannK32 HANN_W13K = make_hann_w(0.5 * HANN_KAPPA13, 8192, 32, 256);

// N = 512 (hopw9_kernel: two hops per wave, 32 lanes x 8 points each, sample i = 64 q + 2 t + e)
__device__ constexpr HannK32 HANN_W9 = make_hann_w(0.5, 512, 8, 64);
__device__ constexpr HannK32 HANN_E9 = make_hann_w(HANN_ENV_AMP, 256, 4, 64);
constexpr double HANN_KAPPA9 = -0.25 / 512.0;
__device__ constexpr HannK32 HANN_W9K = make_hann_w(0.5 * HANN_KAPPA9, 512, 8, 64);
constexpr int H9_HB = 288;                        // hopw9: exchange buffer of ONE half-wave (284 used)
constexpr int H9_TA = 2 * H9_HB;                  // [33] W_256^r
constexpr int H9_TR = H9_TA + 40;                 // [33] W_512^r (r = 32: lane 0's second residue, as i W)
constexpr int H9_TB = H9_TR + 40;                 // [8] W_64^l
constexpr int H9_TC = H9_TB + 8;                  // [4] W_32^l
constexpr int H9_TH = H9_TC + 8;                  // [128] window / envelope rotations
constexpr int HOPW9_LDS_FLOAT2 = H9_TH + 128;     // 6 400 B

// N = 1024 (hopw10_kernel: two hops per wave, 32 lanes x 16 points each, sample i = 64 q + 2 t + e)
__device__ constexpr HannK32 HANN_W10 = make_hann_w(0.5, 1024, 16, 64);
__device__ constexpr HannK32 HANN_E10 = make_hann_w(HANN_ENV_AMP, 512, 8, 64);
constexpr double HANN_KAPPA10 = -0.25 / 1024.0;
__device__ constexpr HannK32 HANN_W10K = make_hann_w(0.5 * HANN_KAPPA10, 1024, 16, 64);
constexpr int H0_HB = 288;                        // hopw10: exchange buffer of ONE half-wave (288 used)
constexpr int H0_TA = 2 * H0_HB;                  // [33] W_512^r
constexpr int H0_TR = H0_TA + 40;                 // [33] W_1024^r (r = 32: lane 0's second residue, as i W)
constexpr int H0_TB = H0_TR + 40;                 // [8] W_64^l
constexpr int H0_TC = H0_TB + 8;                  // [8] W_32^l
constexpr int H0_TH = H0_TC + 8;                  // [128] window / envelope rotations: lane t at 2 t (+ 64: envelope)
constexpr int HOPW10_LDS_FLOAT2 = H0_TH + 128;    // 6 400 B

// N = 2048 (hopw11_kernel: 64 threads x 16 points, sample i = 128 q + 2 t + e)
__device__ constexpr HannK32 HANN_W11 = make_hann_w(0.5, 2048, 16, 128);
__device__ constexpr HannK32 HANN_E11 = make_hann_w(HANN_ENV_AMP, 1024, 8, 128);
constexpr double HANN_KAPPA11 = -0.25 / 2048.0;
__device__ constexpr HannK32 HANN_W11K = make_hann_w(0.5 * HANN_KAPPA11, 2048, 16, 128);
constexpr int H1_BUF = 552;                       // hopw11: exchange buffer (548 used)
constexpr int H1_TA = H1_BUF;                     // [65] W_1024^r
constexpr int H1_TR = H1_TA + 72;                 // [65] W_2048^r (r = 64: lane 0's second residue, as i W)
constexpr int H1_TB = H1_TR + 72;                 // [8] W_128^l
constexpr int H1_TC = H1_TB + 8;                  // [8] W_64^l
constexpr int H1_TH = H1_TC + 8;                  // [256] window / envelope rotations
constexpr int HOPW11_LDS_FLOAT2 = H1_TH + 256;    // 7 744 B

constexpr int H2_BUF = 2304;                      // hopw2: exchange buffer (2297 used by E1; a wave's own half: 1152)
constexpr int H2_TA = H2_BUF;                     // [128] W_4096^r
constexpr int H2_TR = H2_TA + 128;                // [129] W_8192^r, r <= 128 (r = 128: thread 0's second residue, as i W)
constexpr int H2_TB = H2_TR + 136;                // [16] W_256^l
constexpr int H2_TC = H2_TB + 16;                 // [16] W_128^l
constexpr int H2_TH = H2_TC + 16;                 // [512] window / envelope rotations: thread t at 2 t (+ 256: envelope)
constexpr int HOPW2_LDS_FLOAT2 = H2_TH + 512;     // 24 896 B

#ifndef RC_HOPW_PREFETCH
#define RC_HOPW_PREFETCH 1  // hopw11_kernel: next hop's loads before the last inverse pass (0: after the stores, for A/B)
#endif
// compiler-only ordering of one wave's LDS accesses (no instruction is emitted)
__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---- pieces shared by the one-wave (N = 4096) and two-wave (N = 8192) kernels -------------------------------------
// F1: register brev5(q) := z[q * T + t] * window, stages 0..4. Stage 0 pairs registers brev5(q) and brev5(q + 16) =
// brev5(q) + 1: a +- b with a = x_q w_q and b = x_{q+16} w_{q+16} is one multiply and two FMAs
// the hop's samples: row q of thread t = samples 2 T q + 2 t, + 1
// UNIFORM: `src` is the same for every lane (one hop per wave / per two waves): raw buffer loads - the hop's base in a
// resource descriptor, one 32-bit lane offset for all rows, the row in the scalar offset (round 5, as hop4_kernel:
// no 64-bit VALU address arithmetic). The two-hops-per-wave kernels pass per-lane pointers and keep global loads.
template <int T, int P, bool UNIFORM = false>
__device__ __forceinline__ void hopw_load(GF src, unsigned lane2, float (&xr0)[P], float (&xr1)[P]) {
    if constexpr (UNIFORM) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x40000000, 0x00020000);
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * lane2), 4 * 2 * T * q, 0);
            xr0[q] = __uint_as_float(x.x);
            xr1[q] = __uint_as_float(x.y);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < P; ++q) {
        xr0[q] = (src + 2 * T * q)[lane2];
        xr1[q] = (src + 2 * T * q)[lane2 + 1];
    }
}
// TABW (round 5): a caller-supplied window - its values come from the engine's table (L2-resident: N floats) instead of
// the per-thread rotation of the computed hanning window; `wt2` = window table + 2 x (the lane's index in its hop)
template <int T, int m, int P = 32, bool TABW = false>
__device__ __forceinline__ void hopw_f1x(const float (&xr0)[P], const float (&xr1)[P], v2f cb, v2f sb, const HannK32 &W, v2f (&v)[P],
                                         GF wt2 = nullptr) {
    constexpr int LP = P == 32 ? 5 : (P == 16 ? 4 : 3), HP = P / 2;
    static_assert(P == 32 || P == 16 || P == 8, "points per lane");
    const v2f half2 = {0.5f, 0.5f};
    if constexpr (TABW) {
        // in batches of 4 butterflies (16 table floats in flight): with all 2 P values requested at once the 32-point kernels spill
        constexpr int B = HP < 4 ? HP : 4;
#pragma unroll
        for (int q0 = 0; q0 < HP; q0 += B) {
            float l0[B], l1[B], h0[B], h1[B];
#pragma unroll
            for (int q = 0; q < B; ++q) {
                l0[q] = (wt2 + 2 * T * (q0 + q))[0];
                l1[q] = (wt2 + 2 * T * (q0 + q))[1];
                h0[q] = (wt2 + 2 * T * (q0 + q + HP))[0];
                h1[q] = (wt2 + 2 * T * (q0 + q + HP))[1];
            }
#pragma unroll
            for (int qq = 0; qq < B; ++qq) {
                const int q = q0 + qq;
                const v2f a = v2f{xr0[q], xr1[q]} * v2f{l0[qq], l1[qq]}, xh = v2f{xr0[q + HP], xr1[q + HP]};
                const v2f wh = v2f{h0[qq], h1[qq]};
                v[2 * brev_c(q, LP - 1)] = __builtin_elementwise_fma(xh, wh, a);
                v[2 * brev_c(q, LP - 1) + 1] = __builtin_elementwise_fma(-xh, wh, a);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        dit_stages<P, m, 1, LP - 1, 0, false, false>(v);
        return;
    }
#pragma unroll
    for (int q = 0; q < HP; ++q) {
        const v2f wl = __builtin_elementwise_fma(v2f{W.s[q], W.s[q]}, sb,
                       __builtin_elementwise_fma(v2f{W.c[q], W.c[q]}, cb, half2));
        const v2f wh = __builtin_elementwise_fma(v2f{W.s[q + HP], W.s[q + HP]}, sb,
                       __builtin_elementwise_fma(v2f{W.c[q + HP], W.c[q + HP]}, cb, half2));
        const v2f a = v2f{xr0[q], xr1[q]} * wl, xh = v2f{xr0[q + HP], xr1[q + HP]};
        v[2 * brev_c(q, LP - 1)] = __builtin_elementwise_fma(xh, wh, a);
        v[2 * brev_c(q, LP - 1) + 1] = __builtin_elementwise_fma(-xh, wh, a);
    }
    dit_stages<P, m, 1, LP - 1, 0, false, false>(v);
}
template <int T, int m, int P = 32, bool TABW = false>
__device__ __forceinline__ void hopw_f1(GF src, unsigned lane2, v2f cb, v2f sb, const HannK32 &W, v2f (&v)[P], GF wt2 = nullptr) {
    float xr0[P], xr1[P];
    hopw_load<T, P, true>(src, lane2, xr0, xr1);  // (hopw_f1's callers pass hop_src: uniform)
    hopw_f1x<T, m, P, TABW>(xr0, xr1, cb, sb, W, v, wt2);
}

// The middle stage in registers: pair (A[q], B[15 - q]) = bins (r + RES q, M - that), M = 16 RES. Thread 0 owns the
// two residues that pair with themselves (0 and RES / 2): its 32 bins form 17 pairs; its registers are re-dealt so that
// the same 16 slots compute 16 of them (slots 0..7 on residue 0 with bin 0 as slot 0, slots 8..15 on residue RES / 2
// through a second twiddle base wrh / hash counter) and bin M / 2 is one extra pair (hop4_kernel).
// wrl = W_N^r, wrh = the same (thread 0: i W_N^(RES / 2)); W_N^(RES q) = W_32^q at both sizes.
// has0 (wave-uniform): this wave contains thread 0 - the other wave of hopw2_kernel skips the re-deal and the extra pair
template <int LOG2N, int RES, int NS = 16>
__device__ __forceinline__ void hopw_middle(v2f (&va)[NS], v2f (&vb)[NS], const bool is0, const uint32_t r,
                                            const float2 wrl, const float2 wrh, const PhaseKey &key, const bool has0 = true) {
    static_assert((1 << LOG2N) == 2 * NS * RES, "N = 2 NS RES");
    constexpr int h = NS / 2;  // (NS = 16 registers per set at N = 4096 / 8192, 8 at N = 2048: read 16 / 8 / 15 below as NS / h / NS - 1)
    v2f s8 = va[h];
    if (has0) {
        const v2f va0 = va[0];
#pragma unroll
        for (int i = 0; i < h; ++i) {
            const v2f a = va[h + i], b0 = vb[i], b1 = vb[h + i];
            const v2f nx = i < h - 1 ? va[h + 1 + i] : va0;
            va[h + i] = vsel(is0, b0, a);
            vb[i] = vsel(is0, b1, b0);
            vb[h + i] = vsel(is0, nx, b1);
        }
    }
    {
        const uint32_t x0 = r * key.mul + key.k0;
        const uint32_t dx = (uint32_t)RES * key.mul;
        const uint32_t x0h = x0 - (is0 ? (uint32_t)(h * RES - RES / 2) * key.mul : 0u);  // thread 0: bins RES/2 + RES (q - h)
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const float2 wr = q < h ? wrl : wrh;
            const v2f wrv = to_v(wr);
            constexpr int WS = 16 / NS;  // W_N^(RES q) = W_(2 NS)^q = W_32^(q WS)
            const v2f wq = q == 0 ? wrv : (q == h ? v2f{wr.y, -wr.x}
                           : vcmul(wrv, v2f{W32_RE[(q * WS) & 15], W32_IM[(q * WS) & 15]}));
            v2f VA, VB;
            if (q == 0)
                pair_regs_pk4<LOG2N, true>(va[q], vb[NS - 1 - q], wq, x0, key, VA, VB, is0);
            else
                pair_regs_pk4<LOG2N>(va[q], vb[NS - 1 - q], wq, (q < h ? x0 : x0h) + (uint32_t)q * dx, key, VA, VB);
            va[q] = VA;
            vb[NS - 1 - q] = VB;
        }
    }
    if (has0) {  // bin M / 2 pairs with itself: exp(-2 pi i (M/2) / N) = -i; then un-deal thread 0's registers
        v2f V8, V8b;
        pair_regs_pk4<LOG2N>(s8, s8, v2f{0.0f, -1.0f}, (uint32_t)(h * RES) * key.mul + key.k0, key, V8, V8b);
        v2f na[h], nb0[h], nb1[h];
#pragma unroll
        for (int i = 0; i < h; ++i) {
            na[i] = vsel(is0, i == 0 ? V8 : vb[h - 1 + i], va[h + i]);
            nb0[i] = vsel(is0, va[h + i], vb[i]);
            nb1[i] = vsel(is0, vb[i], vb[h + i]);
        }
#pragma unroll
        for (int i = 0; i < h; ++i) {
            va[h + i] = na[i];
            vb[i] = nb0[i];
            vb[h + i] = nb1[i];
        }
    }
}

// Epilogue: synthesis window (times -1/(4N)), overlap-add with the carried tail, store. cbW.. = this thread's window /
// envelope rotations (cos, sin of beta(2 t), beta(2 t + 1)); t = the thread's index in the hop, 2 T samples per row
template <int P, bool TABW = false, int T = 64>
__device__ __forceinline__ void hopw_window(v2f (&y)[P], const v2f cbW, const v2f sbW, const HannK32 &WK, const float half_kappa,
                                            GF wt2 = nullptr) {
    const v2f half2k = {half_kappa, half_kappa};
    if constexpr (TABW) {  // the synthesis window from its table, times -1/(4N) = 2 half_kappa
        const v2f kap = half2k + half2k;
#pragma unroll
        for (int q0 = 0; q0 < P; q0 += 4) {
            float a0[4], a1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a0[q] = (wt2 + 2 * T * (q0 + q))[0];
                a1[q] = (wt2 + 2 * T * (q0 + q))[1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q0 + q] *= v2f{a0[q], a1[q]} * kap;
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
        y[q] *= __builtin_elementwise_fma(v2f{WK.s[q], WK.s[q]}, sbW,
                __builtin_elementwise_fma(v2f{WK.c[q], WK.c[q]}, cbW, half2k));
}
// (WINDOWED: the caller has applied hopw_window already - hopw10_kernel, which fetches the other half-wave's tail between)
// TABW: wt2 / et2 = window / envelope table + 2 x (the lane's index in its hop): the values of a caller's window
template <int PITCHC, int T, int P = 32, bool WINDOWED = false, bool TABW = false>
__device__ __forceinline__ void hopw_epilogue(const HopParams &p, GFW outc, const int64_t k, const bool emit, const int t,
                                              v2f (&y)[P], v2f (&tail)[P / 2], const v2f cbW, const v2f sbW, v2f cbE,
                                              v2f sbE, const HannK32 &WK, const HannK32 &E, const float half_kappa,
                                              const uint32_t pitch, GF wt2 = nullptr, GF et2 = nullptr) {
    constexpr int PH = P / 2, H = T * P;
    constexpr bool PITCH1 = PITCHC == 1;
    const v2f half2 = {0.5f, 0.5f};
    const unsigned lane2 = 2u * (unsigned)t;
    if (!WINDOWED) hopw_window<P, TABW, T>(y, cbW, sbW, WK, half_kappa, wt2);
    if (emit) {
        const v2f amp2 = {p.amp, p.amp};
        if constexpr (TABW) {
            // (y + tail) * env * amp with the envelope from its table, in place and in batches of 4 rows (all PH rows
            // requested at once spill at 32 points per lane); the store loops below then take y[q] as it is
#pragma unroll
            for (int q0 = 0; q0 < PH; q0 += 4) {
                float e0[4], e1[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e0[q] = (et2 + 2 * T * (q0 + q))[0];
                    e1[q] = (et2 + 2 * T * (q0 + q))[1];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q0 + q] = (y[q0 + q] + tail[q0 + q]) * v2f{e0[q], e1[q]} * amp2;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // env[i] * amp = amp/2 + c_q (amp cb) + s_q (amp sb): the amplitude rides on the per-thread rotation
        cbE *= amp2;
        sbE *= amp2;
        const v2f halfa = half2 * amp2;
        const int64_t g0 = k * (int64_t)H;
        if constexpr (PITCH1) {
            const unsigned long long da = (unsigned long long)(outc + (g0 - p.out_origin));
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);
            const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
            GFW dst = (GFW)(((unsigned long long)dhi << 32) | dlo);
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{E.s[q], E.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{E.c[q], E.c[q]}, cbE, halfa));
                const v2f o = TABW ? y[q] : (y[q] + tail[q]) * er;  // (y + tail) * (env * amp), stretcher.rs:97-100
                __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * q + lane2));  // (plain stores: +2 % at 2048 / 8192)
            }
        } else {
            // F[t] = O[t * pitch] (src/resampler.rs:3-18): branch-free raw buffer stores, a lane that keeps
            // nothing stores out of range (hop4_kernel)
            const int64_t kq = g0 / pitch;
            const uint32_t kr = (uint32_t)(g0 % pitch);
            const unsigned long long da = (unsigned long long)(outc + (kq - p.out_origin));
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);
            const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((unsigned long long)dhi << 32) | dlo), 0, 0x40000000, 0x00020000);
            if constexpr (PITCHC > 1) {  // the pitch at compile time: rc_dit.hpp, pitch_store_pair
                const PitchOffsets<PITCHC> po = pitch_offsets<PITCHC>(kr + 2u * (uint32_t)t);
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    const v2f er = __builtin_elementwise_fma(v2f{E.s[q], E.s[q]}, sbE,
                                   __builtin_elementwise_fma(v2f{E.c[q], E.c[q]}, cbE, halfa));
                    const v2f o = TABW ? y[q] : (y[q] + tail[q]) * er;
                    const float ox = o.x, oy = o.y;
                    pitch_store_pair<PITCHC, 2 * T>(rsrc, po, q, ox, oy);
                }
            } else {
            constexpr uint32_t DROP = 0xFFFFFFFCu;
            const uint32_t a00 = kr + 2u * (uint32_t)t;
            const uint32_t d0 = a00 / pitch;
            uint32_t rr = a00 - d0 * pitch, d4 = 4u * d0;
            const uint32_t qs4 = 4u * ((2u * T) / pitch), rs = (2u * T) - (qs4 / 4u) * pitch;
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{E.s[q], E.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{E.c[q], E.c[q]}, cbE, halfa));
                const v2f o = TABW ? y[q] : (y[q] + tail[q]) * er;
                const float ox = o.x, oy = o.y;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ox), rsrc, rr == 0 ? d4 : DROP, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(oy), rsrc, rr + 1 == pitch ? d4 + 4u : DROP, 0, 0);
                d4 += qs4;
                rr += rs;
                if (rr >= pitch) {
                    rr -= pitch;
                    d4 += 4u;
                }
            }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = y[q + PH];
}

template <int PITCHC, bool TABW = false>  // PITCHC 1: pitch 1; 2 / 3: that pitch at compile time; 0: any pitch > 1; TABW: a caller's window (tables)
__global__ __launch_bounds__(64, 3) void hopw_kernel(const HopParams p) {
    constexpr int LOG2N = 12, m = 11, T = 64, P = 32, PH = 16, RES = 128;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;
    {   // tables, once per run
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M), k < M / 2
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N), j <= M / 4
        lds[HW_TA + tid] = ldg2(wt + tid);
        lds[HW_TR + tid] = ldg2(rt + tid);
        if (tid == 0) {
            lds[HW_TA + 64] = ldg2(wt + 64);
            const float2 w64 = ldg2(rt + 64);            // lane 0's second residue: W_N^(64 - 1024) = i W_N^64
            lds[HW_TR + 64] = make_float2(-w64.y, w64.x);
        }
        if (tid < 16) {
            lds[HW_TB + tid] = ldg2(wt + 16 * tid);
            lds[HW_TC + tid] = ldg2(wt + 32 * tid);
        }
        if constexpr (!TABW) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // hann_rot: [part][lane][4] = {cos, sin}(beta(2t)), {cos, sin}(beta(2t + 1))
            const float2 a = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid);
            const float2 b = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid + 1);
            lds[HW_TH + 128 * i + 2 * tid] = make_float2(a.x, b.x);      // (cos beta_0, cos beta_1)
            lds[HW_TH + 128 * i + 2 * tid + 1] = make_float2(a.y, b.y);  // (sin beta_0, sin beta_1)
        }
        }
        __syncthreads();
    }
    // lane identities are re-derived from an opaque copy of the lane id where they are needed (hop4: kept across the hop
    // they cost ~20 VGPRs and were spilled)
    auto lane = [&]() {
        int t = tid;
        opaque(t);
        return t;
    };
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = tid == 0;

    for (int64_t k = (k_begin > 0 ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        hopw_f1<T, m, P, TABW>(hop_src(p, xc, xt, k), lane2, to_v(lds[HW_TH + 2 * tid]), to_v(lds[HW_TH + 2 * tid + 1]), HANN_W12, v,
                               TABW ? per_hop(p.window) + lane2 : nullptr);
        // ---- E1: registers P0..P4 -> P4..P8, round = P4. Weights: P0 16, P1 33, P2 66, P3 136, P5 272, P6 544,
        // P7 1, P8 2, P9 4, P10 8 (lane t: P5 = t5 ... P10 = t0)
        v2f w2[P];
        int l2;  // F2 lane identity: (P0..P3) = l2 & 15, P9 = bit 4, P10 = bit 5
        wfence();
        {
            const int t = lane();
            const int b1s = 272 * ((t >> 5) & 1) + 544 * ((t >> 4) & 1) + ((t >> 3) & 1) + 2 * ((t >> 2) & 1) +
                            4 * ((t >> 1) & 1) + 8 * (t & 1);
            l2 = t;
            const int b1l = 16 * (t & 1) + 33 * ((t >> 1) & 1) + 66 * ((t >> 2) & 1) + 136 * ((t >> 3) & 1) +
                            4 * ((t >> 4) & 1) + 8 * ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r)  // register 16 h + r: (P0..P3) = r
                    lds[b1s + 16 * (r & 1) + 33 * ((r >> 1) & 1) + 66 * ((r >> 2) & 1) + 136 * ((r >> 3) & 1)] = to_f2(v[16 * h + r]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 16; ++sg)  // register j = h | sg << 1: (P5, P6, P7, P8) = sg
                    w2[h | (sg << 1)] = to_v(lds[b1l + 272 * (sg & 1) + 544 * ((sg >> 1) & 1) + ((sg >> 2) & 1) + 2 * ((sg >> 3) & 1)]);
                wfence();
            }
        }
        dit_stages<32, m, 5, 6, 4, false, true>(w2, to_v(lds[HW_TB + (l2 & 15)]));
        // ---- E2: registers P4..P8 -> sets of P7..P10, round = P6 = the set. Identity weights on the reduced index
        v2f va[16], vb[16];
        wfence();
        {
            const int t = lane();
            const int b2s = (t & 15) + 256 * ((t >> 4) & 1) + 512 * ((t >> 5) & 1);
            const int tb = (64 - t) & 63;  // low residue bits of 128 - tau (tau = 0: residue 64)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {  // registers with P6 = 0: (P4, P5, P7, P8) = kk
                const int j = (kk & 3) | ((kk >> 2) << 3);
                lds[b2s + 16 * (kk & 1) + 32 * ((kk >> 1) & 1) + 64 * ((kk >> 2) & 1) + 128 * ((kk >> 3) & 1)] = to_f2(w2[j]);
            }
            wfence();
#pragma unroll
            for (int q = 0; q < 16; ++q) va[q] = to_v(lds[t + 64 * q]);
            wfence();
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int j = (kk & 3) | 4 | ((kk >> 2) << 3);
                lds[b2s + 16 * (kk & 1) + 32 * ((kk >> 1) & 1) + 64 * ((kk >> 2) & 1) + 128 * ((kk >> 3) & 1)] = to_f2(w2[j]);
            }
            wfence();
#pragma unroll
            for (int q = 0; q < 16; ++q) vb[q] = to_v(lds[tb + 64 * q]);
            wfence();
        }
        const int r = lane();  // residue of set A (set B: RES - r; lane 0: RES / 2)
        {
            const v2f wa = to_v(lds[HW_TA + r]);    // W_M^r
            const v2f k16 = {W32_RE[2], W32_IM[2]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
            if (is0) wb = v2f{W32_RE[1], W32_IM[1]};  // lane 0: W_M^64 = W_32
            dit_stages<16, m, 7, 10, 7, false, true>(va, wa);
            dit_stages<16, m, 7, 10, 7, false, true>(vb, wb);
        }
        // ---- middle stage in registers (hopw_middle)
        hopw_middle<LOG2N, RES>(va, vb, is0, (uint32_t)r, lds[HW_TR + r], lds[is0 ? HW_TR + 64 : HW_TR + r], key);
        // ---- inverse: I1 in registers (register index = brev4(q) = Q0..Q3)
        v2f pa[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            pa[brev_c(q, 4)] = va[q];
            pb[brev_c(q, 4)] = vb[q];
        }
        dit_stages<16, m, 0, 3, 0, true, false>(pa);
        dit_stages<16, m, 0, 3, 0, true, false>(pb);
        // ---- E3: sets of Q0..Q3 -> registers Q4..Q8, round = Q4 = the set. Weights: Q10 1, Q9 2, Q8 4, Q7 8, Q6 16,
        // Q5 32 (= the residue's low six bits as they stand), Q0 65, Q1 132, Q2 264, Q3 528
        int l5;  // I2 lane identity: (Q0..Q3) = l5 & 15, Q9 = bit 4, Q10 = bit 5
        wfence();
        {
            const int t = lane();
            const int tb = (64 - t) & 63;
            l5 = t;
            const int b3l = 65 * (t & 1) + 132 * ((t >> 1) & 1) + 264 * ((t >> 2) & 1) + 528 * ((t >> 3) & 1) +
                            2 * ((t >> 4) & 1) + ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bs = h ? tb : t;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    lds[bs + 65 * (q & 1) + 132 * ((q >> 1) & 1) + 264 * ((q >> 2) & 1) + 528 * ((q >> 3) & 1)] = to_f2(h ? pb[q] : pa[q]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 16; ++sg)  // register k = h | sg << 1: (Q5, Q6, Q7, Q8) = sg
                    v[h | (sg << 1)] = to_v(lds[b3l + 32 * (sg & 1) + 16 * ((sg >> 1) & 1) + 8 * ((sg >> 2) & 1) + 4 * ((sg >> 3) & 1)]);
                wfence();
            }
        }
        dit_stages<32, m, 4, 5, 4, true, true>(v, to_v(lds[HW_TC + (l5 & 15)]));
        // ---- E4: registers Q4..Q8 -> Q6..Q10, round = Q8. Identity weights on the reduced index
        v2f y[P];
        wfence();
        {
            const int t = lane();
            const int b4s = (t & 15) + 256 * ((t >> 4) & 1) + 512 * ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)  // register 16 h + kk: (Q4..Q7) = kk
                    lds[b4s + 16 * kk] = to_f2(v[16 * h + kk]);
                wfence();
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {  // register (Q6, Q7, Q9, Q10) = rr, Q8 = h
                    const int rg = (rr & 3) | (h << 2) | ((rr >> 2) << 3);
                    y[rg] = to_v(lds[t + 64 * (rr & 1) + 128 * ((rr >> 1) & 1) + 256 * ((rr >> 2) & 1) + 512 * ((rr >> 3) & 1)]);
                }
                wfence();
            }
        }
        dit_stages<32, m, 6, 10, 6, true, true>(y, to_v(lds[HW_TA + lane()]));

        {
            const int t = lane();
            hopw_epilogue<PITCHC, T, P, false, TABW>(p, outc, k, k >= k_begin, t, y, tail, to_v(lds[HW_TH + 2 * t]), to_v(lds[HW_TH + 2 * t + 1]),
                                     to_v(lds[HW_TH + 128 + 2 * t]), to_v(lds[HW_TH + 128 + 2 * t + 1]), HANN_W12K, HANN_E12,
                                     (float)(0.5 * HANN_KAPPA12), pitch, TABW ? per_hop(p.window) + 2 * t : nullptr,
                                     TABW ? per_hop(p.env) + 2 * t : nullptr);
        }
    }
}


// ---- N = 2048: hopw11_kernel - one wave per hop with 16 points per lane (M = 1024) ------------------------------
// hopw_kernel's structure with passes of (4, 3, 3) / (3, 3, 4) stages: F1 stages 0..3 on the 16 registers (P0..P3),
// F2 stages 4..6 on registers P3..P6 (lane = P0..P2, P7..P9), F3 stages 7..9 on two sets of 8 registers (lane tau holds
// residues tau and 128 - tau); I1 stages 0..2 on the sets, I2 stages 3..5 on registers Q3..Q6, I3 stages 6..9 on
// registers Q6..Q9 (lane = Q0..Q5). Exchange rounds: P3, P6 = the set, Q3 = the set, Q6; 8 stores + 8 loads per round
// through a buffer of 548 float2. tests/dev/proto_w11.py found and checks the weights. ~100 VGPRs and 7.6 KB of LDS per
// wave: four waves per SIMD.
#ifndef RC_HOPW11_WPS
#define RC_HOPW11_WPS 3  // register budget of three waves per SIMD: the allocator takes 112 VGPRs (four still fit; with the budget of four it takes 94 and the kernel is 6 % slower)
#endif
template <int PITCHC, bool TABW = false>
__global__ __launch_bounds__(64, RC_HOPW11_WPS) void hopw11_kernel(const HopParams p) {
    constexpr int LOG2N = 11, m = 10, T = 64, P = 16, PH = 8, RES = 128, NS = 8;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;
    {   // tables, once per run
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M), k < M / 2
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N), j <= M / 4
        lds[H1_TA + tid] = ldg2(wt + tid);
        lds[H1_TR + tid] = ldg2(rt + tid);
        if (tid == 0) {
            lds[H1_TA + 64] = ldg2(wt + 64);
            const float2 w64 = ldg2(rt + 64);            // lane 0's second residue: W_N^(64 - 512) = i W_N^64
            lds[H1_TR + 64] = make_float2(-w64.y, w64.x);
        }
        if (tid < 8) {
            lds[H1_TB + tid] = ldg2(wt + 8 * tid);        // W_128^l
            lds[H1_TC + tid] = ldg2(wt + 16 * tid);       // W_64^l
        }
        if constexpr (!TABW) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float2 a = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid);
            const float2 b = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid + 1);
            lds[H1_TH + 128 * i + 2 * tid] = make_float2(a.x, b.x);
            lds[H1_TH + 128 * i + 2 * tid + 1] = make_float2(a.y, b.y);
        }
        }
        __syncthreads();
    }
    auto lane = [&]() {
        int t = tid;
        opaque(t);
        return t;
    };
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = tid == 0;

    // The next hop's samples are loaded before the last inverse pass of this one: they are then ahead of this hop's
    // output stores in the (in-order) vector-memory queue, and the wait at the top of the loop does not include the
    // stores' round trips (timing-only build without stores: -8 %)
    float xr0[P], xr1[P];
    const int64_t k_first = k_begin > 0 ? k_begin - 1 : k_begin;
    hopw_load<T, P, true>(hop_src(p, xc, xt, k_first), lane2, xr0, xr1);
    for (int64_t k = k_first; k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        hopw_f1x<T, m, P, TABW>(xr0, xr1, to_v(lds[H1_TH + 2 * tid]), to_v(lds[H1_TH + 2 * tid + 1]), HANN_W11, v,
                                TABW ? per_hop(p.window) + lane2 : nullptr);
        // ---- E1: registers P0..P3 -> P3..P6, round = P3. Weights: P9 1, P8 2, P7 4, P6 8, P0 16, P1 33, P2 72, P4 137,
        // P5 274 (lane t: P4 = t5 ... P9 = t0)
        v2f w2[P];
        int l2;  // F2 lane identity: (P0, P1, P2) = l2 & 7, P7 = bit 3, P8 = bit 4, P9 = bit 5
        wfence();
        {
            const int t = lane();
            const int b1s = 137 * ((t >> 5) & 1) + 274 * ((t >> 4) & 1) + 8 * ((t >> 3) & 1) + 4 * ((t >> 2) & 1) +
                            2 * ((t >> 1) & 1) + (t & 1);
            l2 = t;
            const int b1l = 16 * (t & 1) + 33 * ((t >> 1) & 1) + 72 * ((t >> 2) & 1) + 4 * ((t >> 3) & 1) +
                            2 * ((t >> 4) & 1) + ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 8; ++r)  // register 8 h + r: (P0, P1, P2) = r
                    lds[b1s + 16 * (r & 1) + 33 * ((r >> 1) & 1) + 72 * ((r >> 2) & 1)] = to_f2(v[8 * h + r]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 8; ++sg)  // register j = h | sg << 1: (P4, P5, P6) = sg
                    w2[h | (sg << 1)] = to_v(lds[b1l + 137 * (sg & 1) + 274 * ((sg >> 1) & 1) + 8 * ((sg >> 2) & 1)]);
                wfence();
            }
        }
        dit_stages<16, m, 4, 6, 3, false, true>(w2, to_v(lds[H1_TB + (l2 & 7)]));
        // ---- E2: registers P3..P6 -> sets of P7..P9, round = P6 = the set. Weights: P0..P5 1..32, P7 72, P8 136, P9 272
        v2f va[NS], vb[NS];
        wfence();
        {
            const int t = lane();
            const int b2s = (t & 7) + 72 * ((t >> 3) & 1) + 136 * ((t >> 4) & 1) + 272 * ((t >> 5) & 1);
            const int tb = (64 - t) & 63;  // low residue bits of 128 - tau (tau = 0: residue 64)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) lds[b2s + 8 * kk] = to_f2(w2[kk | (h << 3)]);  // (P3, P4, P5) = kk
                wfence();
                const int bl = h ? tb : t;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v2f x = to_v(lds[bl + 72 * (q & 1) + 136 * ((q >> 1) & 1) + 272 * ((q >> 2) & 1)]);
                    if (h) vb[q] = x;
                    else va[q] = x;
                }
                wfence();
            }
        }
        const int r = lane();  // residue of set A (set B: RES - r; lane 0: RES / 2)
        {
            const v2f wa = to_v(lds[H1_TA + r]);     // W_M^r
            const v2f k8 = {W32_RE[4], W32_IM[4]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k8);    // W_M^(RES - r) = W_8 conj(W_M^r)
            if (is0) wb = v2f{W32_RE[2], W32_IM[2]};  // lane 0: W_M^64 = W_16
            dit_stages<8, m, 7, 9, 7, false, true>(va, wa);
            dit_stages<8, m, 7, 9, 7, false, true>(vb, wb);
        }
        hopw_middle<LOG2N, RES, NS>(va, vb, is0, (uint32_t)r, lds[H1_TR + r], lds[is0 ? H1_TR + 64 : H1_TR + r], key);
        // ---- inverse: I1 in registers (register index = brev3(q) = Q0..Q2)
        v2f pa[NS], pb[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            pa[brev_c(q, 3)] = va[q];
            pb[brev_c(q, 3)] = vb[q];
        }
        dit_stages<8, m, 0, 2, 0, true, false>(pa);
        dit_stages<8, m, 0, 2, 0, true, false>(pb);
        // ---- E3: sets of Q0..Q2 -> registers Q3..Q6, round = Q3 = the set. Weights: Q9 1, Q8 2, Q7 4, Q6 8, Q5 16, Q4 32
        // (= the residue's low six bits as they stand), Q0 65, Q1 136, Q2 272
        int l5;  // I2 lane identity: (Q0, Q1, Q2) = l5 & 7, Q7 = bit 3, Q8 = bit 4, Q9 = bit 5
        wfence();
        {
            const int t = lane();
            const int tb = (64 - t) & 63;
            l5 = t;
            const int b3l = 65 * (t & 1) + 136 * ((t >> 1) & 1) + 272 * ((t >> 2) & 1) + 4 * ((t >> 3) & 1) +
                            2 * ((t >> 4) & 1) + ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bs = h ? tb : t;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    lds[bs + 65 * (q & 1) + 136 * ((q >> 1) & 1) + 272 * ((q >> 2) & 1)] = to_f2(h ? pb[q] : pa[q]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 8; ++sg)  // register k = h | sg << 1: (Q4, Q5, Q6) = sg
                    v[h | (sg << 1)] = to_v(lds[b3l + 32 * (sg & 1) + 16 * ((sg >> 1) & 1) + 8 * ((sg >> 2) & 1)]);
                wfence();
            }
        }
        dit_stages<16, m, 3, 5, 3, true, true>(v, to_v(lds[H1_TC + (l5 & 7)]));
        // ---- E4: registers Q3..Q6 -> Q6..Q9, round = Q6. Weights: Q0..Q5 1..32, Q7 72, Q8 136, Q9 272
        v2f y[P];
        wfence();
        {
            const int t = lane();
            const int b4s = (t & 7) + 72 * ((t >> 3) & 1) + 136 * ((t >> 4) & 1) + 272 * ((t >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) lds[b4s + 8 * kk] = to_f2(v[kk | (h << 3)]);  // (Q3, Q4, Q5) = kk
                wfence();
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)  // register (Q7, Q8, Q9) = rr, Q6 = h
                    y[h | (rr << 1)] = to_v(lds[t + 72 * (rr & 1) + 136 * ((rr >> 1) & 1) + 272 * ((rr >> 2) & 1)]);
                wfence();
            }
        }
        if (RC_HOPW_PREFETCH) hopw_load<T, P, true>(hop_src(p, xc, xt, k + 1 < k_end ? k + 1 : k), lane2, xr0, xr1);
        dit_stages<16, m, 6, 9, 6, true, true>(y, to_v(lds[H1_TA + lane()]));
        {
            const int t = lane();
            hopw_epilogue<PITCHC, T, P, false, TABW>(p, outc, k, k >= k_begin, t, y, tail, to_v(lds[H1_TH + 2 * t]), to_v(lds[H1_TH + 2 * t + 1]),
                                        to_v(lds[H1_TH + 128 + 2 * t]), to_v(lds[H1_TH + 128 + 2 * t + 1]), HANN_W11K, HANN_E11,
                                        (float)(0.5 * HANN_KAPPA11), pitch, TABW ? per_hop(p.window) + 2 * t : nullptr,
                                        TABW ? per_hop(p.env) + 2 * t : nullptr);
        }
        if (!RC_HOPW_PREFETCH) hopw_load<T, P, true>(hop_src(p, xc, xt, k + 1 < k_end ? k + 1 : k), lane2, xr0, xr1);
    }
}

// ---- N = 1024: hopw10_kernel - TWO HOPS PER WAVE (M = 512 = 32 lanes x 16 points per hop) ---------------------------
// Lanes 0-31 run hop kk, lanes 32-63 hop kk + 1 of the same run, each half-wave with hopw11_kernel's structure minus one
// stage: passes (4, 2, 3) / (3, 2, 4), two sets of 8 registers around the pair stage (lane tau of a half holds residues
// tau and 64 - tau), exchange rounds P3, P5 = the set, Q3 = the set, Q5 through 288 float2 per half (ds_write_b64 banks
// in 16-lane groups, ds_read_b64 in 32-lane groups = one half; tests/dev/proto_w10.py searched and checks the weights).
// The hop index, the phase key, the source pointer and the store offsets are per lane. Overlap-add across the halves:
// the upper half's head takes the lower half's tail of the same iteration, the lower half's head the upper half's tail of
// the iteration before - one v_permlane32_swap per register moves both.
template <int PITCHC, bool TABW = false>
__global__ __launch_bounds__(64, 3) void hopw10_kernel(const HopParams p) {
    constexpr int LOG2N = 10, m = 9, T = 32, P = 16, PH = 8, RES = 64, NS = 8;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;
    {   // tables, once per run
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M), k < M / 2
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N), j <= M / 4
        if (tid < 33) {
            lds[H0_TA + tid] = ldg2(wt + tid);
            const float2 w = ldg2(rt + tid);
            lds[H0_TR + tid] = tid == 32 ? make_float2(-w.y, w.x) : w;  // [32]: W_N^(32 - 256) = i W_N^32
        }
        if (tid < 8) {
            lds[H0_TB + tid] = ldg2(wt + 8 * tid);    // W_64^l
            lds[H0_TC + tid] = ldg2(wt + 16 * tid);   // W_32^l
        }
        if (!TABW && tid < 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // hann_rot: [part][64 threads][4]; the 32 lanes of a hop are threads 0..31
                const float2 a = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid);
                const float2 b = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid + 1);
                lds[H0_TH + 64 * i + 2 * tid] = make_float2(a.x, b.x);
                lds[H0_TH + 64 * i + 2 * tid + 1] = make_float2(a.y, b.y);
            }
        }
        __syncthreads();
    }
    auto lane = [&]() {
        int t = tid;
        opaque(t);
        return t;
    };
    v2f tail[PH];  // the windowed second half of this lane's last hop
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = (tid & 31) == 0;
    const int64_t k_first = k_begin > 0 ? k_begin - 1 : k_begin;

    for (int64_t kk = k_first; kk < k_end; kk += 2) {
        const int hf = lane() >> 5;
        const int64_t k = kk + hf;                          // this lane's hop
        const bool valid = k < k_end;
        const int64_t kc = valid ? k : k_end - 1;           // (an odd tail: the upper half recomputes the last hop, unused)
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, kc);
        float2 *own = lds + H0_HB * hf;
        v2f v[P];
        {
            const int tl = tid & 31;
            float xr0[P], xr1[P];
            hopw_load<T, P>(hop_src_lane(p, xc, xt, kc), 2u * (unsigned)tl, xr0, xr1);
            hopw_f1x<T, m, P, TABW>(xr0, xr1, to_v(lds[H0_TH + 2 * tl]), to_v(lds[H0_TH + 2 * tl + 1]), HANN_W10, v,
                                    TABW ? per_hop(p.window) + 2 * tl : nullptr);
        }
        // ---- E1: registers P0..P3 -> P3..P6, round = P3. Weights: P8 1, P7 2, P6 4, P5 8, P0 16, P1 36, P2 72, P4 140
        // (lane t of the half: P4 = t4 ... P8 = t0)
        v2f w2[P];
        int l2;  // F2 lane identity: (P0, P1, P2) = l2 & 7, P7 = bit 3, P8 = bit 4
        wfence();
        {
            const int t = lane() & 31;
            const int b1s = 140 * ((t >> 4) & 1) + 8 * ((t >> 3) & 1) + 4 * ((t >> 2) & 1) + 2 * ((t >> 1) & 1) + (t & 1);
            l2 = t;
            const int b1l = 16 * (t & 1) + 36 * ((t >> 1) & 1) + 72 * ((t >> 2) & 1) + 2 * ((t >> 3) & 1) + ((t >> 4) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 8; ++r)  // register 8 h + r: (P0, P1, P2) = r
                    own[b1s + 16 * (r & 1) + 36 * ((r >> 1) & 1) + 72 * ((r >> 2) & 1)] = to_f2(v[8 * h + r]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 8; ++sg)  // register j = h | sg << 1: (P4, P5, P6) = sg
                    w2[h | (sg << 1)] = to_v(own[b1l + 140 * (sg & 1) + 8 * ((sg >> 1) & 1) + 4 * ((sg >> 2) & 1)]);
                wfence();
            }
        }
        dit_stages<16, m, 4, 5, 3, false, true>(w2, to_v(lds[H0_TB + (l2 & 7)]));
        // ---- E2: registers P3..P6 -> sets of P6..P8, round = P5 = the set. Weights: P0..P4 1..16, P6 32, P7 72, P8 136
        v2f va[NS], vb[NS];
        wfence();
        {
            const int t = lane() & 31;
            const int b2s = (t & 7) + 72 * ((t >> 3) & 1) + 136 * ((t >> 4) & 1);
            const int tb = (32 - t) & 31;  // low residue bits of 64 - tau (tau = 0: residue 32)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kq = 0; kq < 8; ++kq)  // registers with P5 = h: (P3, P4, P6) = kq
                    own[b2s + 8 * (kq & 1) + 16 * ((kq >> 1) & 1) + 32 * ((kq >> 2) & 1)] = to_f2(w2[(kq & 3) | (h << 2) | ((kq >> 2) << 3)]);
                wfence();
                const int bl = h ? tb : t;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const v2f x = to_v(own[bl + 32 * (q & 1) + 72 * ((q >> 1) & 1) + 136 * ((q >> 2) & 1)]);
                    if (h) vb[q] = x;
                    else va[q] = x;
                }
                wfence();
            }
        }
        const int r = lane() & 31;  // residue of set A (set B: RES - r; lane 0 of the half: RES / 2)
        {
            const v2f wa = to_v(lds[H0_TA + r]);     // W_M^r
            const v2f k8 = {W32_RE[4], W32_IM[4]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k8);    // W_M^(RES - r) = W_8 conj(W_M^r)
            if (is0) wb = v2f{W32_RE[2], W32_IM[2]};  // lane 0: W_M^32 = W_16
            dit_stages<8, m, 6, 8, 6, false, true>(va, wa);
            dit_stages<8, m, 6, 8, 6, false, true>(vb, wb);
        }
        hopw_middle<LOG2N, RES, NS>(va, vb, is0, (uint32_t)r, lds[H0_TR + r], lds[is0 ? H0_TR + 32 : H0_TR + r], key);
        // ---- inverse: I1 in registers (register index = brev3(q) = Q0..Q2)
        v2f pa[NS], pb[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            pa[brev_c(q, 3)] = va[q];
            pb[brev_c(q, 3)] = vb[q];
        }
        dit_stages<8, m, 0, 2, 0, true, false>(pa);
        dit_stages<8, m, 0, 2, 0, true, false>(pb);
        // ---- E3: sets of Q0..Q2 -> registers Q3..Q6, round = Q3 = the set. Weights: Q8 1, Q7 2, Q6 4, Q5 8, Q4 16 (= the
        // residue's low five bits as they stand), Q0 36, Q1 72, Q2 144
        int l5;  // I2 lane identity: (Q0, Q1, Q2) = l5 & 7, Q7 = bit 3, Q8 = bit 4
        wfence();
        {
            const int t = lane() & 31;
            const int tb = (32 - t) & 31;
            l5 = t;
            const int b3l = 36 * (t & 1) + 72 * ((t >> 1) & 1) + 144 * ((t >> 2) & 1) + 2 * ((t >> 3) & 1) + ((t >> 4) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bs = h ? tb : t;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    own[bs + 36 * (q & 1) + 72 * ((q >> 1) & 1) + 144 * ((q >> 2) & 1)] = to_f2(h ? pb[q] : pa[q]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 8; ++sg)  // register k = h | sg << 1: (Q4, Q5, Q6) = sg
                    v[h | (sg << 1)] = to_v(own[b3l + 16 * (sg & 1) + 8 * ((sg >> 1) & 1) + 4 * ((sg >> 2) & 1)]);
                wfence();
            }
        }
        dit_stages<16, m, 3, 4, 3, true, true>(v, to_v(lds[H0_TC + (l5 & 7)]));
        // ---- E4: registers Q3..Q6 -> Q5..Q8, round = Q5. Weights: Q0..Q4 1..16, Q7 40, Q6 72, Q8 144
        v2f y[P];
        wfence();
        {
            const int t = lane() & 31;
            const int b4s = (t & 7) + 40 * ((t >> 3) & 1) + 144 * ((t >> 4) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kq = 0; kq < 8; ++kq)  // registers with Q5 = h: (Q3, Q4, Q6) = kq
                    own[b4s + 8 * (kq & 1) + 16 * ((kq >> 1) & 1) + 72 * ((kq >> 2) & 1)] = to_f2(v[(kq & 3) | (h << 2) | ((kq >> 2) << 3)]);
                wfence();
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)  // register (Q6, Q7, Q8) = rr, Q5 = h
                    y[h | (rr << 1)] = to_v(own[t + 72 * (rr & 1) + 40 * ((rr >> 1) & 1) + 144 * ((rr >> 2) & 1)]);
                wfence();
            }
        }
        dit_stages<16, m, 5, 8, 5, true, true>(y, to_v(lds[H0_TA + (lane() & 31)]));
        {
            const int t = lane(), tl = t & 31, up = t >> 5;
            hopw_window<P, TABW, T>(y, to_v(lds[H0_TH + 2 * tl]), to_v(lds[H0_TH + 2 * tl + 1]), HANN_W10K, (float)(0.5 * HANN_KAPPA10),
                                    TABW ? per_hop(p.window) + 2 * tl : nullptr);
            // the tail this lane's head overlaps: lower half <- upper half's tail of the iteration before (carried in
            // `tail`), upper half <- lower half's tail of this iteration. v_permlane32_swap(a, b): a = (a.lo, b.lo),
            // b = (a.hi, b.hi) over the two half-waves
            v2f tin[PH];
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                float ax = tail[q].x, bx = y[PH + q].x, ay = tail[q].y, by = y[PH + q].y;
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
                tin[q] = up ? v2f{ax, ay} : v2f{bx, by};
                tail[q] = y[PH + q];
            }
            // stores: the iteration's base is hop kk (uniform); the upper half's samples lie H = 512 further on
            hopw_epilogue<PITCHC, T, P, true, TABW>(p, outc, kk, valid && k >= k_begin, tl + 256 * up, y, tin,
                                              v2f{0.f, 0.f}, v2f{0.f, 0.f}, to_v(lds[H0_TH + 64 + 2 * tl]),
                                              to_v(lds[H0_TH + 64 + 2 * tl + 1]), HANN_W10K, HANN_E10, 0.0f, pitch, nullptr,
                                              TABW ? per_hop(p.env) + 2 * tl : nullptr);
        }
    }
}

// ---- N = 512: hopw9_kernel - two hops per wave, 8 points per lane (M = 256 = 32 lanes x 8) ----------------------------
// hopw10_kernel's arrangement with passes of (3, 3, 2) / (2, 3, 3) stages: every pass uses all three register bits, so an
// exchange is ONE round of 8 stores + 8 loads per lane through 284 float2 per half (tests/dev/proto_w9.py); two sets of 4
// registers around the pair stage (lane tau of a half holds residues tau and 64 - tau).
template <int PITCHC, bool TABW = false>
__global__ __launch_bounds__(64, 4) void hopw9_kernel(const HopParams p) {
    constexpr int LOG2N = 9, m = 8, T = 32, P = 8, PH = 4, RES = 64, NS = 4;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;
    {   // tables, once per run
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M), k < M / 2
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N), j <= M / 4
        if (tid < 33) {
            lds[H9_TA + tid] = ldg2(wt + tid);
            const float2 w = ldg2(rt + tid);
            lds[H9_TR + tid] = tid == 32 ? make_float2(-w.y, w.x) : w;  // [32]: W_N^(32 - 128) = i W_N^32
        }
        if (tid < 8) lds[H9_TB + tid] = ldg2(wt + 4 * tid);   // W_64^l
        if (tid < 4) lds[H9_TC + tid] = ldg2(wt + 8 * tid);   // W_32^l
        if (!TABW && tid < 32) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // hann_rot: [part][64 threads][4]; the 32 lanes of a hop are threads 0..31
                const float2 a = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid);
                const float2 b = ldg2((GV2)p.hann_rot + 128 * i + 2 * tid + 1);
                lds[H9_TH + 64 * i + 2 * tid] = make_float2(a.x, b.x);
                lds[H9_TH + 64 * i + 2 * tid + 1] = make_float2(a.y, b.y);
            }
        }
        __syncthreads();
    }
    auto lane = [&]() {
        int t = tid;
        opaque(t);
        return t;
    };
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = (tid & 31) == 0;
    const int64_t k_first = k_begin > 0 ? k_begin - 1 : k_begin;

    for (int64_t kk = k_first; kk < k_end; kk += 2) {
        const int hf = lane() >> 5;
        const int64_t k = kk + hf;
        const bool valid = k < k_end;
        const int64_t kc = valid ? k : k_end - 1;
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, kc);
        float2 *own = lds + H9_HB * hf;
        v2f v[P];
        {
            const int tl = tid & 31;
            float xr0[P], xr1[P];
            hopw_load<T, P>(hop_src_lane(p, xc, xt, kc), 2u * (unsigned)tl, xr0, xr1);
            hopw_f1x<T, m, P, TABW>(xr0, xr1, to_v(lds[H9_TH + 2 * tl]), to_v(lds[H9_TH + 2 * tl + 1]), HANN_W9, v,
                                    TABW ? per_hop(p.window) + 2 * tl : nullptr);
        }
        // ---- E1: registers P0..P2 -> P3..P5. Weights: P7 1 ... P3 16 (= the lane as it stands), P0 36, P1 72, P2 144
        v2f w2[P];
        int l2;  // F2 lane identity: (P0, P1, P2) = l2 & 7, P6 = bit 3, P7 = bit 4
        wfence();
        {
            const int t = lane() & 31;
            l2 = t;
            const int b1l = 36 * (t & 7) + 2 * ((t >> 3) & 1) + ((t >> 4) & 1);
#pragma unroll
            for (int r = 0; r < 8; ++r) own[t + 36 * r] = to_f2(v[r]);
            wfence();
#pragma unroll
            for (int j = 0; j < 8; ++j) w2[j] = to_v(own[b1l + 16 * (j & 1) + 8 * ((j >> 1) & 1) + 4 * ((j >> 2) & 1)]);
            wfence();
        }
        dit_stages<8, m, 3, 5, 3, false, true>(w2, to_v(lds[H9_TB + (l2 & 7)]));
        // ---- E2: registers P3..P5 -> sets of P6, P7 (set = P5). Weights: P0..P5 1..32, P6 72, P7 136
        v2f va[NS], vb[NS];
        wfence();
        {
            const int t = lane() & 31;
            const int b2s = (t & 7) + 72 * ((t >> 3) & 1) + 136 * ((t >> 4) & 1);
            const int tb = ((32 - t) & 31) + 32;  // 64 - tau (tau = 0: residue 32): its low five bits, set bit on
#pragma unroll
            for (int j = 0; j < 8; ++j) own[b2s + 8 * j] = to_f2(w2[j]);
            wfence();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                va[q] = to_v(own[t + 72 * (q & 1) + 136 * ((q >> 1) & 1)]);
                vb[q] = to_v(own[tb + 72 * (q & 1) + 136 * ((q >> 1) & 1)]);
            }
            wfence();
        }
        const int r = lane() & 31;
        {
            const v2f wa = to_v(lds[H9_TA + r]);     // W_M^r
            v2f wb = v2f{-wa.y, -wa.x};              // W_M^(RES - r) = W_4 conj(W_M^r) = -i conj(W_M^r)
            if (is0) wb = v2f{W32_RE[4], W32_IM[4]};  // lane 0: W_M^32 = W_8
            dit_stages<4, m, 6, 7, 6, false, true>(va, wa);
            dit_stages<4, m, 6, 7, 6, false, true>(vb, wb);
        }
        hopw_middle<LOG2N, RES, NS>(va, vb, is0, (uint32_t)r, lds[H9_TR + r], lds[is0 ? H9_TR + 32 : H9_TR + r], key);
        // ---- inverse: I1 in registers (register index = brev2(q) = Q0, Q1)
        v2f pa[NS], pb[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            pa[brev_c(q, 2)] = va[q];
            pb[brev_c(q, 2)] = vb[q];
        }
        dit_stages<4, m, 0, 1, 0, true, false>(pa);
        dit_stages<4, m, 0, 1, 0, true, false>(pb);
        // ---- E3: sets of Q0, Q1 -> registers Q2..Q4. Weights: Q7 1 ... Q3 16 (= the residue's low five bits), Q2 32 (the
        // set), Q0 72, Q1 144
        int l5;  // I2 lane identity: (Q0, Q1) = l5 & 3, Q5 = bit 2, Q6 = bit 3, Q7 = bit 4
        wfence();
        {
            const int t = lane() & 31;
            const int tb = ((32 - t) & 31) + 32;
            l5 = t;
            const int b3l = 72 * (t & 1) + 144 * ((t >> 1) & 1) + 4 * ((t >> 2) & 1) + 2 * ((t >> 3) & 1) + ((t >> 4) & 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                own[t + 72 * (q & 1) + 144 * ((q >> 1) & 1)] = to_f2(pa[q]);
                own[tb + 72 * (q & 1) + 144 * ((q >> 1) & 1)] = to_f2(pb[q]);
            }
            wfence();
#pragma unroll
            for (int kq = 0; kq < 8; ++kq)  // register (Q2, Q3, Q4) = kq
                v[kq] = to_v(own[b3l + 32 * (kq & 1) + 16 * ((kq >> 1) & 1) + 8 * ((kq >> 2) & 1)]);
            wfence();
        }
        dit_stages<8, m, 2, 4, 2, true, true>(v, to_v(lds[H9_TC + (l5 & 3)]));
        // ---- E4: registers Q2..Q4 -> Q5..Q7. Weights: Q0..Q4 1..16, Q5 36, Q6 72, Q7 140
        v2f y[P];
        wfence();
        {
            const int t = lane() & 31;
            const int b4s = (t & 3) + 36 * ((t >> 2) & 1) + 72 * ((t >> 3) & 1) + 140 * ((t >> 4) & 1);
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) own[b4s + 4 * kq] = to_f2(v[kq]);
            wfence();
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) y[rr] = to_v(own[t + 36 * (rr & 1) + 72 * ((rr >> 1) & 1) + 140 * ((rr >> 2) & 1)]);
            wfence();
        }
        dit_stages<8, m, 5, 7, 5, true, true>(y, to_v(lds[H9_TA + (lane() & 31)]));
        {
            const int t = lane(), tl = t & 31, up = t >> 5;
            hopw_window<P, TABW, T>(y, to_v(lds[H9_TH + 2 * tl]), to_v(lds[H9_TH + 2 * tl + 1]), HANN_W9K, (float)(0.5 * HANN_KAPPA9),
                                    TABW ? per_hop(p.window) + 2 * tl : nullptr);
            v2f tin[PH];  // (hopw10_kernel: the other half-wave's tail)
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                float ax = tail[q].x, bx = y[PH + q].x, ay = tail[q].y, by = y[PH + q].y;
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
                tin[q] = up ? v2f{ax, ay} : v2f{bx, by};
                tail[q] = y[PH + q];
            }
            hopw_epilogue<PITCHC, T, P, true, TABW>(p, outc, kk, valid && k >= k_begin, tl + 128 * up, y, tin,
                                              v2f{0.f, 0.f}, v2f{0.f, 0.f}, to_v(lds[H9_TH + 64 + 2 * tl]),
                                              to_v(lds[H9_TH + 64 + 2 * tl + 1]), HANN_W9K, HANN_E9, 0.0f, pitch, nullptr,
                                              TABW ? per_hop(p.env) + 2 * tl : nullptr);
        }
    }
}

// ---- N = 8192: hopw2_kernel - TWO WAVES PER HOP (128 threads x 32 complex points, M = 4096) ---------------------
// hopw_kernel's structure with one more stage (passes of (5, 3, 4) / (4, 3, 5)); tests/dev/proto_w2.py is the index
// model. The wave is the LOWEST position bit P0 = lowest bin bit = Q11 wherever the data is in bin order: residue r and
// its partner 256 - r have the same parity, so both bins of a pair, F2 before and I2 after, live in one wave and the
// exchanges E2 / E3 are wave-local (each wave in its own half of the buffer, no barrier). E1 (sample order -> P0 waves)
// and E4 (back) cross the two waves: a workgroup barrier around each of their rounds, 9 per hop, between two waves only.
//   F1 stages 0..4 (thread t = low sample bits: lane = P11..P6, wave = P5), F2 stages 5..7 on registers P4..P8
//   (lane = P1..P3, P9..P11), F3 stages 8..11 on two sets of 16 (thread tau = 2 lane + wave holds residues tau and
//   256 - tau; thread 0: residues 0 and 128); I1 stages 0..3, I2 stages 4..6 on registers Q4..Q8, I3 stages 7..11 on
//   registers Q7..Q11 (thread = Q0..Q6).
template <int PITCHC, bool TABW = false>
__global__ __launch_bounds__(128, 3) void hopw2_kernel(const HopParams p) {
    constexpr int LOG2N = 13, m = 12, T = 128, P = 32, PH = 16, RES = 256, HALF = H2_BUF / 2;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;
    {   // tables, once per run
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M), k < M / 2
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N), j <= M / 4
        lds[H2_TA + tid] = ldg2(wt + tid);
        lds[H2_TR + tid] = ldg2(rt + tid);
        if (tid == 0) {
            const float2 wh = ldg2(rt + 128);             // thread 0's second residue: W_N^(128 - 2048) = i W_N^128
            lds[H2_TR + 128] = make_float2(-wh.y, wh.x);
        }
        if (tid < 16) {
            lds[H2_TB + tid] = ldg2(wt + 16 * tid);       // W_256^l
            lds[H2_TC + tid] = ldg2(wt + 32 * tid);       // W_128^l
        }
        if constexpr (!TABW) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // hann_rot: [part][thread][4] = {cos, sin}(beta(2t)), {cos, sin}(beta(2t + 1))
            const float2 a = ldg2((GV2)p.hann_rot + 256 * i + 2 * tid);
            const float2 b = ldg2((GV2)p.hann_rot + 256 * i + 2 * tid + 1);
            lds[H2_TH + 256 * i + 2 * tid] = make_float2(a.x, b.x);      // (cos beta_0, cos beta_1)
            lds[H2_TH + 256 * i + 2 * tid + 1] = make_float2(a.y, b.y);  // (sin beta_0, sin beta_1)
        }
        }
        __syncthreads();
    }
    auto thread = [&]() {  // (an opaque copy: identities derived from it are not kept live across the hop)
        int t = tid;
        opaque(t);
        return t;
    };
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = tid == 0;

    for (int64_t k = (k_begin > 0 ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        hopw_f1<T, m, P, TABW>(hop_src(p, xc, xt, k), lane2, to_v(lds[H2_TH + 2 * tid]), to_v(lds[H2_TH + 2 * tid + 1]), HANN_W13, v,
                               TABW ? per_hop(p.window) + lane2 : nullptr);
        // ---- E1 (cross-wave): registers P0..P4 -> P4..P8, round = P4. Weights: P8 1, P9 2, P10 4, P11 8, P1 16, P2 33,
        // P3 72, P0 144, P5 288, P6 576, P7 1152 (writer: wave = P5, lane: P6 = bit 5 ... P11 = bit 0)
        v2f w2[P];
        int l2, wv;  // F2 identity: wave = P0, lane bits (P1, P2, P3, P9, P10, P11)
        {
            const int t = thread();
            const int ln = t & 63;
            wv = t >> 6;
            l2 = ln;
            const int b1s = 288 * wv + 576 * ((ln >> 5) & 1) + 1152 * ((ln >> 4) & 1) + ((ln >> 3) & 1) +
                            2 * ((ln >> 2) & 1) + 4 * ((ln >> 1) & 1) + 8 * (ln & 1);
            const int b1l = 144 * wv + 16 * (ln & 1) + 33 * ((ln >> 1) & 1) + 72 * ((ln >> 2) & 1) +
                            2 * ((ln >> 3) & 1) + 4 * ((ln >> 4) & 1) + 8 * ((ln >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r)  // register 16 h + r: (P0..P3) = r
                    lds[b1s + 144 * (r & 1) + 16 * ((r >> 1) & 1) + 33 * ((r >> 2) & 1) + 72 * ((r >> 3) & 1)] = to_f2(v[16 * h + r]);
                __syncthreads();
#pragma unroll
                for (int sg = 0; sg < 16; ++sg)  // register j = h | sg << 1: (P5, P6, P7, P8) = sg
                    w2[h | (sg << 1)] = to_v(lds[b1l + 288 * (sg & 1) + 576 * ((sg >> 1) & 1) + 1152 * ((sg >> 2) & 1) + ((sg >> 3) & 1)]);
                __syncthreads();
            }
        }
        dit_stages<32, m, 5, 7, 4, false, true>(w2, to_v(lds[H2_TB + wv + 2 * (l2 & 7)]));
        // ---- E2 (wave-local, own half): registers P4..P8 -> sets of P8..P11, round = P7 = the set. Weights: P1 1, P2 2,
        // P3 4, P4 8, P5 16, P6 32, P8 64, P9 136, P10 264, P11 528
        v2f va[16], vb[16];
        wfence();
        {
            const int t = thread();
            const int ln = t & 63, w = t >> 6;
            float2 *own = lds + HALF * w;
            const int b2s = (ln & 7) + 136 * ((ln >> 3) & 1) + 264 * ((ln >> 4) & 1) + 528 * ((ln >> 5) & 1);
            const int tb = (64 - ln - w) & 63;  // bits 1..6 of residue 256 - tau (tau = 0: residue 128)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {  // registers with P7 = h: (P4, P5, P6, P8) = kk
                    const int j = (kk & 7) | (h << 3) | ((kk >> 3) << 4);
                    own[b2s + 8 * (kk & 1) + 16 * ((kk >> 1) & 1) + 32 * ((kk >> 2) & 1) + 64 * ((kk >> 3) & 1)] = to_f2(w2[j]);
                }
                wfence();
                const int bl = h ? tb : ln;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const v2f x = to_v(own[bl + 64 * (q & 1) + 136 * ((q >> 1) & 1) + 264 * ((q >> 2) & 1) + 528 * ((q >> 3) & 1)]);
                    if (h) vb[q] = x;
                    else va[q] = x;
                }
                wfence();
            }
        }
        const int r = [&]() {  // residue of set A: tau = 2 lane + wave (set B: RES - tau; thread 0: RES / 2)
            const int t = thread();
            return 2 * (t & 63) + (t >> 6);
        }();
        {
            const v2f wa = to_v(lds[H2_TA + r]);    // W_M^r
            const v2f k16 = {W32_RE[2], W32_IM[2]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
            if (is0) wb = v2f{W32_RE[1], W32_IM[1]};  // thread 0: W_M^128 = W_32
            dit_stages<16, m, 8, 11, 8, false, true>(va, wa);
            dit_stages<16, m, 8, 11, 8, false, true>(vb, wb);
        }
        hopw_middle<LOG2N, RES>(va, vb, is0, (uint32_t)r, lds[H2_TR + r], lds[is0 ? H2_TR + 128 : H2_TR + r], key, tid < 64);
        // ---- inverse: I1 in registers (register index = brev4(q) = Q0..Q3)
        v2f pa[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            pa[brev_c(q, 4)] = va[q];
            pb[brev_c(q, 4)] = vb[q];
        }
        dit_stages<16, m, 0, 3, 0, true, false>(pa);
        dit_stages<16, m, 0, 3, 0, true, false>(pb);
        // ---- E3 (wave-local): sets of Q0..Q3 -> registers Q4..Q8, round = Q4 = the set. Weights: Q10 1, Q9 2, Q8 4,
        // Q7 8, Q6 16, Q5 32 (= bits 1..6 of the residue as they stand), Q0 65, Q1 132, Q2 264, Q3 528
        int l5;  // I2 identity: wave = Q11, lane bits (Q0..Q3, Q9, Q10)
        wfence();
        {
            const int t = thread();
            const int ln = t & 63, w = t >> 6;
            float2 *own = lds + HALF * w;
            const int tb = (64 - ln - w) & 63;
            l5 = ln;
            const int b3l = 65 * (ln & 1) + 132 * ((ln >> 1) & 1) + 264 * ((ln >> 2) & 1) + 528 * ((ln >> 3) & 1) +
                            2 * ((ln >> 4) & 1) + ((ln >> 5) & 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int bs = h ? tb : ln;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    own[bs + 65 * (q & 1) + 132 * ((q >> 1) & 1) + 264 * ((q >> 2) & 1) + 528 * ((q >> 3) & 1)] = to_f2(h ? pb[q] : pa[q]);
                wfence();
#pragma unroll
                for (int sg = 0; sg < 16; ++sg)  // register k = h | sg << 1: (Q5, Q6, Q7, Q8) = sg
                    v[h | (sg << 1)] = to_v(own[b3l + 32 * (sg & 1) + 16 * ((sg >> 1) & 1) + 8 * ((sg >> 2) & 1) + 4 * ((sg >> 3) & 1)]);
                wfence();
            }
        }
        dit_stages<32, m, 4, 6, 4, true, true>(v, to_v(lds[H2_TC + (l5 & 15)]));
        // ---- E4 (cross-wave): registers Q4..Q8 -> Q7..Q11, round = Q8. Identity weights on the reduced index
        // (Q0..Q7, Q9, Q10, Q11); the writer's wave is Q11, the reader's thread Q0..Q6
        v2f y[P];
        {
            const int t = thread();
            const int ln = t & 63, w = t >> 6;
            const int b4s = (ln & 15) + 256 * ((ln >> 4) & 1) + 512 * ((ln >> 5) & 1) + 1024 * w;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                __syncthreads();  // (the other wave is done with this wave's half: E3, or the round before)
#pragma unroll
                for (int kk = 0; kk < 16; ++kk)  // register 16 h + kk: (Q4..Q7) = kk
                    lds[b4s + 16 * kk] = to_f2(v[16 * h + kk]);
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {  // register (Q7, Q9, Q10, Q11) = rr, Q8 = h
                    const int rg = (rr & 1) | (h << 1) | ((rr >> 1) << 2);
                    y[rg] = to_v(lds[t + 128 * (rr & 1) + 256 * ((rr >> 1) & 1) + 512 * ((rr >> 2) & 1) + 1024 * ((rr >> 3) & 1)]);
                }
            }
            __syncthreads();
        }
        dit_stages<32, m, 7, 11, 7, true, true>(y, to_v(lds[H2_TA + thread()]));
        {
            const int t = thread();
            hopw_epilogue<PITCHC, T, P, false, TABW>(p, outc, k, k >= k_begin, t, y, tail, to_v(lds[H2_TH + 2 * t]), to_v(lds[H2_TH + 2 * t + 1]),
                                     to_v(lds[H2_TH + 256 + 2 * t]), to_v(lds[H2_TH + 256 + 2 * t + 1]), HANN_W13K, HANN_E13,
                                     (float)(0.5 * HANN_KAPPA13), pitch, TABW ? per_hop(p.window) + 2 * t : nullptr,
                                     TABW ? per_hop(p.env) + 2 * t : nullptr);
        }
    }
}

}  // namespace

size_t hopw_lds_bytes() { return sizeof(float2) * (size_t)HOPW_LDS_FLOAT2; }

// N = 4096, fused path, default hanning window (HopParams::hann_rot set: [2][64][4]).
hipError_t launch_hopw(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(64);
    const size_t lds = sizeof(float2) * (size_t)HOPW_LDS_FLOAT2;
    if (!p.hann_rot) {  // a caller's window: its table values (pitch 1, or any pitch at run time)
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the TABW kernels dereference both tables)
        if (p.pitch == 1) hipLaunchKernelGGL((hopw_kernel<1, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((hopw_kernel<0, true>), grid, block, lds, s, p);
        return hipGetLastError();
    }
    if (p.pitch == 1) hipLaunchKernelGGL((hopw_kernel<1>), grid, block, lds, s, p);
    else if (p.pitch == 2) hipLaunchKernelGGL((hopw_kernel<2>), grid, block, lds, s, p);
    else if (p.pitch == 3) hipLaunchKernelGGL((hopw_kernel<3>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((hopw_kernel<0>), grid, block, lds, s, p);
    return hipGetLastError();
}

// N = 512, fused path, default hanning window (HopParams::hann_rot set: [2][64][4]).
hipError_t launch_hopw9(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(64);
    const size_t lds = sizeof(float2) * (size_t)HOPW9_LDS_FLOAT2;
    if (!p.hann_rot) {  // a caller's window: its table values (pitch 1, or any pitch at run time)
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the TABW kernels dereference both tables)
        if (p.pitch == 1) hipLaunchKernelGGL((hopw9_kernel<1, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((hopw9_kernel<0, true>), grid, block, lds, s, p);
        return hipGetLastError();
    }
    if (p.pitch == 1) hipLaunchKernelGGL((hopw9_kernel<1>), grid, block, lds, s, p);
    else if (p.pitch == 2) hipLaunchKernelGGL((hopw9_kernel<2>), grid, block, lds, s, p);
    else if (p.pitch == 3) hipLaunchKernelGGL((hopw9_kernel<3>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((hopw9_kernel<0>), grid, block, lds, s, p);
    return hipGetLastError();
}

// N = 1024, fused path, default hanning window (HopParams::hann_rot set: [2][64][4]).
hipError_t launch_hopw10(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(64);
    const size_t lds = sizeof(float2) * (size_t)HOPW10_LDS_FLOAT2;
    if (!p.hann_rot) {  // a caller's window: its table values (pitch 1, or any pitch at run time)
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the TABW kernels dereference both tables)
        if (p.pitch == 1) hipLaunchKernelGGL((hopw10_kernel<1, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((hopw10_kernel<0, true>), grid, block, lds, s, p);
        return hipGetLastError();
    }
    if (p.pitch == 1) hipLaunchKernelGGL((hopw10_kernel<1>), grid, block, lds, s, p);
    else if (p.pitch == 2) hipLaunchKernelGGL((hopw10_kernel<2>), grid, block, lds, s, p);
    else if (p.pitch == 3) hipLaunchKernelGGL((hopw10_kernel<3>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((hopw10_kernel<0>), grid, block, lds, s, p);
    return hipGetLastError();
}

// N = 2048, fused path, default hanning window (HopParams::hann_rot set: [2][64][4]).
hipError_t launch_hopw11(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(64);
    const size_t lds = sizeof(float2) * (size_t)HOPW11_LDS_FLOAT2;
    if (!p.hann_rot) {  // a caller's window: its table values (pitch 1, or any pitch at run time)
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the TABW kernels dereference both tables)
        if (p.pitch == 1) hipLaunchKernelGGL((hopw11_kernel<1, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((hopw11_kernel<0, true>), grid, block, lds, s, p);
        return hipGetLastError();
    }
    if (p.pitch == 1) hipLaunchKernelGGL((hopw11_kernel<1>), grid, block, lds, s, p);
    else if (p.pitch == 2) hipLaunchKernelGGL((hopw11_kernel<2>), grid, block, lds, s, p);
    else if (p.pitch == 3) hipLaunchKernelGGL((hopw11_kernel<3>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((hopw11_kernel<0>), grid, block, lds, s, p);
    return hipGetLastError();
}

// N = 8192, fused path, default hanning window (HopParams::hann_rot set: [2][128][4]).
hipError_t launch_hopw2(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(128);
    const size_t lds = sizeof(float2) * (size_t)HOPW2_LDS_FLOAT2;
    if (!p.hann_rot) {  // a caller's window: its table values (pitch 1, or any pitch at run time)
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the TABW kernels dereference both tables)
        if (p.pitch == 1) hipLaunchKernelGGL((hopw2_kernel<1, true>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((hopw2_kernel<0, true>), grid, block, lds, s, p);
        return hipGetLastError();
    }
    if (p.pitch == 1) hipLaunchKernelGGL((hopw2_kernel<1>), grid, block, lds, s, p);
    else if (p.pitch == 2) hipLaunchKernelGGL((hopw2_kernel<2>), grid, block, lds, s, p);
    else if (p.pitch == 3) hipLaunchKernelGGL((hopw2_kernel<3>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((hopw2_kernel<0>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace rc
