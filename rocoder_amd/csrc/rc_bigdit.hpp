// Shared by the fused large-window kernels (rc_big4.hip: N = 32768, and N = 65536 in the test-hook build;
// rc_big5.hip: N = 65536): 64th-root constants, the computed default window's per-row constants and the DIT stage
// groups for up to 64 registers.
#pragma once
#include "rc_dit.hpp"

namespace rc {
namespace {

struct W64Tab {
    float re[32], im[32];
};
constexpr W64Tab make_w64() {  // exp(-2 pi i c / 64), c < 32
    W64Tab t{};
    for (int c = 0; c < 32; ++c) {
        t.re[c] = (float)cx_cos(2.0 * CX_PI * c / 64.0);
        t.im[c] = (float)(-cx_sin(2.0 * CX_PI * c / 64.0));
    }
    return t;
}
__device__ constexpr W64Tab W64 = make_w64();
// default-window fast path of big4_kernel (as HANN_W14 for hop4): value(i) = base + c[q] cos(beta) + s[q] sin(beta)
// for sample i = 1024 q + 2 t + e, beta = 2 pi (2 t + e) / (len - 1)
struct HannK64 {
    float c[64], s[64];
};
constexpr HannK64 make_hann_k64(double amp, int len, int count) {
    HannK64 k{};
    for (int q = 0; q < 64; ++q) {
        const double a = q < count ? 2.0 * CX_PI * 1024.0 * q / (double)(len - 1) : 0.0;
        k.c[q] = (float)(-amp * cx_cos(a));
        k.s[q] = (float)(amp * cx_sin(a));
    }
    return k;
}
__device__ constexpr HannK64 HANN_W15 = make_hann_k64(0.5, 32768, 32);
__device__ constexpr HannK64 HANN_E15 = make_hann_k64(HANN_ENV_AMP, 16384, 16);
__device__ constexpr HannK64 HANN_W16 = make_hann_k64(0.5, 65536, 64);
__device__ constexpr HannK64 HANN_E16 = make_hann_k64(HANN_ENV_AMP, 32768, 32);

// dit_stages for up to 64 registers: 64th-root constants, otherwise the same arithmetic
// One DIT stage with a runtime base twiddle, twiddle by twiddle: tw = base W_64^kidx serves the butterflies with
// c = cc (as it is) and c = cc + nc (rotated by -i), then dies - one live twiddle instead of NREG / 4. Every loop
// bound is a compile-time constant of the template (the array indices must fold, or v[] ends up in scratch memory).
template <int NREG, int RB, bool CONJ>
__device__ __forceinline__ void lean_stage(v2f (&v)[NREG], v2f base) {
    constexpr int half = 1 << RB, nc = half > 1 ? half / 2 : 1, nblk = NREG / (2 * half);
#pragma unroll
    for (int cc = 0; cc < nc; ++cc) {
        const int kidx = cc * (32 >> RB);
        const v2f kc = {W64.re[kidx & 31], W64.im[kidx & 31]};
        const v2f tw = cc == 0 ? base : vcmul(base, kc);
#pragma unroll
        for (int blk = 0; blk < nblk; ++blk) {
            const int q0 = blk * 2 * half + cc, q1 = q0 + half;
            {
                const v2f a = v[q0], b = v[q1];
                vdit_m<CONJ>(a, b, tw, v[q0], v[q1]);
            }
            if constexpr (half > 1) {
                const v2f a = v[q0 + nc], b = v[q1 + nc];
                vdit_rot_m<CONJ>(a, b, tw, v[q0 + nc], v[q1 + nc]);
            }
        }
    }
}
template <int NREG, int S_LO, int S_HI, int REG_LO, bool CONJ, int S = S_LO>
__device__ __forceinline__ void lean_stages(v2f (&v)[NREG], const v2f (&bases)[S_HI - S_LO + 1]) {
    if constexpr (S <= S_HI) {
        lean_stage<NREG, S - REG_LO, CONJ>(v, bases[S - S_LO]);
        lean_stages<NREG, S_LO, S_HI, REG_LO, CONJ, S + 1>(v, bases);
    }
}

// LEAN: one live twiddle at a time (for the R = 64 kernel with its carried tail in registers: 192 of the 256
// registers are data)
template <int NREG, int S_LO, int S_HI, int REG_LO, bool CONJ, bool HAS_L, bool LEAN = false>
__device__ __forceinline__ void dit_g(v2f (&v)[NREG], v2f wfine = v2f{1.0f, 0.0f}) {
    const v2f sgn = CONJ ? v2f{1.0f, -1.0f} : v2f{-1.0f, 1.0f};
    v2f bases[S_HI - S_LO + 1];
    if (HAS_L) {
        bases[S_HI - S_LO] = wfine;
#pragma unroll
        for (int s = S_HI - 1; s >= S_LO; --s) bases[s - S_LO] = vcsq(bases[s + 1 - S_LO]);
        if constexpr (LEAN) {
            lean_stages<NREG, S_LO, S_HI, REG_LO, CONJ>(v, bases);
            return;
        }
    }
#pragma unroll
    for (int s = S_LO; s <= S_HI; ++s) {
        const int rb = s - REG_LO;
        const int half = 1 << rb;
        if (!HAS_L) {
#pragma unroll
            for (int q0 = 0; q0 < NREG; ++q0) {
                if (q0 & half) continue;
                const int q1 = q0 | half;
                const int c = q0 & (half - 1);
                const int kidx = c * (32 >> rb);  // exp(-2 pi i c / 2^(rb+1)) = W64^kidx
                const v2f a = v[q0], b = v[q1];
                const v2f kc = {W64.re[kidx & 31], W64.im[kidx & 31]};
                if (c == 0) {
                    v[q0] = a + b;
                    v[q1] = a - b;
                } else if (kidx == 16) {
                    const v2f ib = __builtin_shufflevector(b, b, 1, 0) * sgn;
                    v[q0] = a - ib;
                    v[q1] = a + ib;
                } else {
                    const v2f w2 = v2f{kc.y, kc.y} * sgn;
                    vdit(a, b, kc, w2, v[q0], v[q1]);
                }
            }
        } else {
            const v2f base = bases[s - S_LO];
            constexpr int NCMAX = NREG / 4 > 0 ? NREG / 4 : 1;
            const int nc = half > 1 ? half / 2 : 1;
            {
                v2f tw[NCMAX];
    #pragma unroll
                for (int c = 0; c < NCMAX; ++c) {
                    if (c >= nc) continue;
                    const int kidx = c * (32 >> rb);
                    const v2f kc = {W64.re[kidx & 31], W64.im[kidx & 31]};
                    tw[c] = c == 0 ? base : vcmul(base, kc);
                }
    #pragma unroll
                for (int q0 = 0; q0 < NREG; ++q0) {
                    if (q0 & half) continue;
                    const int q1 = q0 | half;
                    const int c = q0 & (half - 1);
                    const v2f a = v[q0], b = v[q1];
                    if (c < nc) vdit_m<CONJ>(a, b, tw[c], v[q0], v[q1]);
                    else vdit_rot_m<CONJ>(a, b, tw[c - nc], v[q0], v[q1]);
                }
            }
        }
    }
}

// schedule constants settled by the measurements of rounds 2-4 (docs/LAB_NOTES): one ds_write per
// BIG4_OVL_K VALU instructions where exchange stores are interleaved with butterflies; input rows as a two-deep pipeline
// of BIG4_PIPE_ROWS-row batches; epilogue batches of BIG4_EPI_BATCH output pairs; BIG4_TAIL_LDS of a thread's 32 tail
// pairs in LDS at R = 64
constexpr int BIG4_OVL_K = 4, BIG4_PIPE_ROWS = 16, BIG4_EPI_BATCH = 16, BIG4_TAIL_LDS = 5;  // (epilogue batches of 2 / 4 / 8 / 16 pairs: 5.37 / 5.355 / 5.325 / 5.313 ms on one box)  // (4 / 3 / 2 tail pairs in LDS: no spill either since round 4, and no faster: 5.37-5.39 vs 5.38-5.41 ms)

}  // namespace
}  // namespace rc
