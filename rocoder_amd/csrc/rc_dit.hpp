// Packed DIT butterflies, the in-register (j, M - j) pair stage and the computed hanning window / envelope
// constants shared by the N = 16384 kernels (rc_hop16k.hip) and the fused large-window kernel (rc_big4.hip).
#pragma once
#include "rc_dev.hpp"

namespace rc {
namespace {

// ---- asm-free packed butterflies: plain vector code, hipcc picks the op_sel / neg / inline-constant
// forms itself (no inline-asm boundary pads, free scheduling).
//   DIT: r = a + w b = fma(b.yx, w2, fma(b, w.xx, a)),  w2 = (-w.y, w.y)   [conj: w2 = (w.y, -w.y)]
//        o = a - w b = 2a - r
__device__ __forceinline__ void vdit(v2f a, v2f b, v2f w, v2f w2, v2f &r, v2f &o) {
    const v2f t = __builtin_elementwise_fma(b, __builtin_shufflevector(w, w, 0, 0), a);
    r = __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), w2, t);
    const v2f two = {2.0f, 2.0f};
    o = __builtin_elementwise_fma(a, two, -r);
}
__device__ __forceinline__ v2f vcmul(v2f a, v2f k) {  // a * k
    const v2f t = __builtin_shufflevector(a, a, 0, 0) * k;
    return __builtin_elementwise_fma(__builtin_shufflevector(a, a, 1, 1), v2f{-k.y, k.x}, t);
}
// a * a: with a runtime operand hipcc builds (-a.y, a.x) with a v_xor + v_mov in front of the FMA; as VOP3P source
// modifiers (op_sel + neg_lo) the square is two instructions instead of four
#ifndef RC_VCSQ
#define RC_VCSQ 1
#endif
__device__ __forceinline__ v2f vcsq(v2f a) {
    if (!RC_VCSQ) return vcmul(a, a);
    const v2f t = __builtin_shufflevector(a, a, 0, 0) * a;
    v2f r;  // t + a.yy * (-a.y, a.x)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(a), "v"(t));
    return r;
}

// DIT stages S_LO..S_HI on NREG registers: register bit (s - REG_LO) <-> position bit s; the
// position bits below REG_LO are the runtime value l (< 2^REG_LO; HAS_L = false means l == 0).
//   a' = a + w b, b' = a - w b, w = exp(-2 pi i (p mod 2^s) / 2^(s+1))   (conjugated when CONJ)
// butterfly with the twiddle w' = -i w (the second half of a stage's twiddles is the first half
// rotated by -i): alpha = w.y, beta = -w.x, so only w2r = w.xx * (-sgn) is needed, no complex product
__device__ __forceinline__ void vdit_rot(v2f a, v2f b, v2f w, v2f w2r, v2f &r, v2f &o) {
    const v2f t = __builtin_elementwise_fma(b, __builtin_shufflevector(w, w, 1, 1), a);
    r = __builtin_elementwise_fma(__builtin_shufflevector(b, b, 1, 0), w2r, t);
    const v2f two = {2.0f, 2.0f};
    o = __builtin_elementwise_fma(a, two, -r);
}

// wfine = W_{2^(S_HI+1)}^l, the base twiddle of the last stage; the base of stage s - 1 is the
// square of the base of stage s (no table loads inside the hop loop: a global load waited on in
// place costs its full latency, and vmcnt retires in order behind the output stores).
__device__ __forceinline__ v2f xld(const float2 *lds, int idx) { return to_v(lds[idx]); }
// Runtime-twiddle butterflies with the (-w.y, w.y) / (w.x, -w.x) operand expressed as VOP3P source
// modifiers (op_sel + neg_lo / neg_hi): hipcc does not fold a per-lane negation into the modifiers, so
// the plain-C++ form needs one v_pk_mul per twiddle and form (124 per hop) to build those operands.
#ifndef RC_ASMNEG
#define RC_ASMNEG 1
#endif
//   r = a + w b (CONJ: a + conj(w) b), o = 2a - r
template <bool CONJ>
__device__ __forceinline__ void vdit_m(v2f a, v2f b, v2f w, v2f &r, v2f &o) {
    const v2f t = __builtin_elementwise_fma(b, __builtin_shufflevector(w, w, 0, 0), a);
    if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(b), "v"(w), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(b), "v"(w), "v"(t));
    const v2f two = {2.0f, 2.0f};
    o = __builtin_elementwise_fma(a, two, -r);
}
//   the same with the twiddle -i w (CONJ: +i conj(w))
template <bool CONJ>
__device__ __forceinline__ void vdit_rot_m(v2f a, v2f b, v2f w, v2f &r, v2f &o) {
    const v2f t = __builtin_elementwise_fma(b, __builtin_shufflevector(w, w, 1, 1), a);
    if (CONJ) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(b), "v"(w), "v"(t));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(b), "v"(w), "v"(t));
    const v2f two = {2.0f, 2.0f};
    o = __builtin_elementwise_fma(a, two, -r);
}
//   twiddle -i (CONJ: +i): no multiply at all - the swap and the sign ride on the VOP3P modifiers of two packed adds
#ifndef RC_ASMROT
#define RC_ASMROT 1
#endif
template <bool CONJ>
__device__ __forceinline__ void vdit_i(v2f a, v2f b, v2f &r, v2f &o) {
    v2f p, m;  // p = (a.x + b.y, a.y - b.x) = a - i b ; m = (a.x - b.y, a.y + b.x) = a + i b
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(p) : "v"(a), "v"(b));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(m) : "v"(a), "v"(b));
    r = CONJ ? m : p;
    o = CONJ ? p : m;
}
// Two runtime-twiddle butterflies as ONE asm block, their three packed FMAs interleaved (t0 t1 r0 r1 o0 o1): no
// v_pk_fma_f32 is followed by its consumer (hipcc pads every such pair around an inline asm with an s_nop, and a
// dependent packed FMA issues 8 cycles after its producer: a lone wave runs the one-butterfly form at half rate).
// In place: r lands in b's registers, o in a's; ROT0 / ROT1: that butterfly's twiddle is -i w (CONJ: +i conj w).
#ifndef RC_BF2
#define RC_BF2 1
#endif
#define RC_BF_T_N "op_sel_hi:[1,0,1]"
#define RC_BF_T_R "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
#define RC_BF_R_N "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
#define RC_BF_R_NC "op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"
#define RC_BF_R_R "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]"
#define RC_BF_R_RC "op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]"
#define RC_BF2_ASM(T0, R0, T1, R1)                                                                           \
    asm("v_pk_fma_f32 %[t0], %[b0], %[w0], %[a0] " T0 "\n\t"                                                 \
        "v_pk_fma_f32 %[t1], %[b1], %[w1], %[a1] " T1 "\n\t"                                                 \
        "v_pk_fma_f32 %[b0], %[b0], %[w0], %[t0] " R0 "\n\t"                                                 \
        "v_pk_fma_f32 %[b1], %[b1], %[w1], %[t1] " R1 "\n\t"                                                 \
        "v_pk_fma_f32 %[a0], %[a0], 2.0, %[b0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"          \
        "v_pk_fma_f32 %[a1], %[a1], 2.0, %[b1] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"              \
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [a0] "+v"(a0), [b0] "+v"(b0), [a1] "+v"(a1), [b1] "+v"(b1)         \
        : [w0] "v"(w0), [w1] "v"(w1))
template <bool CONJ>
__device__ __forceinline__ void vdit2_m(v2f &a0, v2f &b0, v2f w0, bool rot0, v2f &a1, v2f &b1, v2f w1, bool rot1) {
    v2f t0, t1;
    if (CONJ) {
        if (!rot0 && !rot1) RC_BF2_ASM(RC_BF_T_N, RC_BF_R_NC, RC_BF_T_N, RC_BF_R_NC);
        else if (!rot0) RC_BF2_ASM(RC_BF_T_N, RC_BF_R_NC, RC_BF_T_R, RC_BF_R_RC);
        else if (!rot1) RC_BF2_ASM(RC_BF_T_R, RC_BF_R_RC, RC_BF_T_N, RC_BF_R_NC);
        else RC_BF2_ASM(RC_BF_T_R, RC_BF_R_RC, RC_BF_T_R, RC_BF_R_RC);
    } else {
        if (!rot0 && !rot1) RC_BF2_ASM(RC_BF_T_N, RC_BF_R_N, RC_BF_T_N, RC_BF_R_N);
        else if (!rot0) RC_BF2_ASM(RC_BF_T_N, RC_BF_R_N, RC_BF_T_R, RC_BF_R_R);
        else if (!rot1) RC_BF2_ASM(RC_BF_T_R, RC_BF_R_R, RC_BF_T_N, RC_BF_R_N);
        else RC_BF2_ASM(RC_BF_T_R, RC_BF_R_R, RC_BF_T_R, RC_BF_R_R);
    }
}
template <int NREG, int M_LOG, int S_LO, int S_HI, int REG_LO, bool CONJ, bool HAS_L>
__device__ __forceinline__ void dit_stages(v2f (&v)[NREG], v2f wfine = v2f{1.0f, 0.0f}) {
    const v2f sgn = CONJ ? v2f{1.0f, -1.0f} : v2f{-1.0f, 1.0f};
    v2f bases[S_HI - S_LO + 1];
    if (HAS_L) {
        bases[S_HI - S_LO] = wfine;
#pragma unroll
        for (int s = S_HI - 1; s >= S_LO; --s) bases[s - S_LO] = vcsq(bases[s + 1 - S_LO]);
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
                const int kidx = c * (16 >> rb);  // exp(-2 pi i c / 2^(rb+1)) = W32^kidx
                const v2f a = v[q0], b = v[q1];
                const v2f kc = {W32_RE[kidx & 15], W32_IM[kidx & 15]};
                if (c == 0) {
                    v[q0] = a + b;
                    v[q1] = a - b;
                } else if (kidx == 8 && RC_ASMROT) {  // w b = -i b (forward) / +i b (inverse)
                    vdit_i<CONJ>(a, b, v[q0], v[q1]);
                } else if (kidx == 8) {  // ... = -(b.yx * sgn)
                    const v2f ib = __builtin_shufflevector(b, b, 1, 0) * sgn;
                    v[q0] = a - ib;
                    v[q1] = a + ib;
                } else {
                    const v2f w2 = v2f{kc.y, kc.y} * sgn;
                    vdit(a, b, kc, w2, v[q0], v[q1]);
                }
            }
        } else {
            const v2f base = bases[s - S_LO];  // W_{2^(s+1)}^l
            // twiddles of the first half of the stage (c < half/2); the rest are these times -i
            constexpr int NCMAX = NREG / 4 > 0 ? NREG / 4 : 1;
            const int nc = half > 1 ? half / 2 : 1;
            v2f tw[NCMAX], tw2[NCMAX], twr[NCMAX];
#pragma unroll
            for (int c = 0; c < NCMAX; ++c) {
                if (c >= nc) continue;
                const int kidx = c * (16 >> rb);
                const v2f kc = {W32_RE[kidx & 15], W32_IM[kidx & 15]};
                tw[c] = c == 0 ? base : vcmul(base, kc);
                if (!RC_ASMNEG) {
                    tw2[c] = __builtin_shufflevector(tw[c], tw[c], 1, 1) * sgn;
                    twr[c] = __builtin_shufflevector(tw[c], tw[c], 0, 0) * (-sgn);
                }
            }
            if (RC_BF2 && RC_ASMNEG) {
                // butterfly i of the stage: q0 = the i-th register index with bit rb clear
#pragma unroll
                for (int i = 0; i < NREG / 2; i += 2) {
                    const int qa = ((i >> rb) << (rb + 1)) | (i & (half - 1)), qb = (((i + 1) >> rb) << (rb + 1)) | ((i + 1) & (half - 1));
                    const int ca_ = qa & (half - 1), cb_ = qb & (half - 1);
                    v2f a0 = v[qa], b0 = v[qa | half], a1 = v[qb], b1 = v[qb | half];
                    vdit2_m<CONJ>(a0, b0, tw[ca_ < nc ? ca_ : ca_ - nc], ca_ >= nc, a1, b1, tw[cb_ < nc ? cb_ : cb_ - nc], cb_ >= nc);
                    v[qa] = b0, v[qa | half] = a0, v[qb] = b1, v[qb | half] = a1;
                }
            } else
#pragma unroll
            for (int q0 = 0; q0 < NREG; ++q0) {
                if (q0 & half) continue;
                const int q1 = q0 | half;
                const int c = q0 & (half - 1);
                const v2f a = v[q0], b = v[q1];
                if (RC_ASMNEG) {
                    if (c < nc) vdit_m<CONJ>(a, b, tw[c], v[q0], v[q1]);
                    else vdit_rot_m<CONJ>(a, b, tw[c - nc], v[q0], v[q1]);
                } else {
                    if (c < nc) vdit(a, b, tw[c], tw2[c], v[q0], v[q1]);
                    else vdit_rot(a, b, tw[c - nc], twr[c - nc], v[q0], v[q1]);
                }
            }
        }
    }
}

// one (ja, M - ja) pair entirely in registers; x1 = ja * mul + k0 (phase counter of bin ja)
template <int LOG2N>
__device__ __forceinline__ void pair_regs(float2 A, float2 Bp, float2 w, uint32_t x1, PhaseKey key,
                                          float2 &VA, float2 &VB, bool dc = false) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    const uint32_t cM = M * key.mul + 2u * key.k0;
    const float nkappa = -0.25f / (float)N;
    float2 X1, X2c;
    pair_analyze(A, Bp, w, X1, X2c);
    const float m1 = cabs_fast(X1) * nkappa, m2 = cabs_fast(X2c) * nkappa;
    float c1, s1, c2, s2, c3, s3, c4, s4;
    phase_ncs2_x(x1, c1, s1, c4, s4);       // bins ja and M + ja
    phase_ncs2_x(cM - x1, c3, s3, c2, s2);  // bins M - ja and N - ja
    if (dc) {  // ja == 0 wraps: N - 0 is bin 0 again, M - 0 is bin M
        c2 = c1, s2 = s1;
        c3 = c4, s3 = s4;
    }
    const float px = m1 * (c1 + c2), py = m1 * (s1 - s2);
    const float qx = m2 * (c3 + c4), qy = m2 * (s4 - s3);
    const float sx = px + qx, sy = py + qy, rx = px - qx, ry = py - qy;
    const float ux = rx * w.x + ry * w.y, uy = ry * w.x - rx * w.y;
    VA = make_float2(sx - uy, sy + ux);
    VB = make_float2(sx + uy, ux - sy);
}

// The same pair in packed (re,im) arithmetic: 17 v_pk_* + 10 transcendental + the two hashes instead
// of ~50 scalar VALU ops. A wave issues one VALU instruction per ~4.75 cycles whatever it is, so the
// instruction count, not the flop count, sets the middle stage's time (profiles/r01e stamps).
#ifndef RC_PAIR_PK
#define RC_PAIR_PK 1
#endif
__device__ __forceinline__ void phase_cs2_x(uint32_t x, v2f &lo, v2f &up) {
    float a, b, c, d;
    phase_ncs2_x(x, a, b, c, d);
    lo = v2f{a, b};
    up = v2f{c, d};
}
__device__ __forceinline__ v2f vsel(bool c, v2f a, v2f b) { return v2f{c ? a.x : b.x, c ? a.y : b.y}; }
// dc (lane predicate): this lane's pair is bin 0 with itself - N - 0 is bin 0 again and M - 0 is bin M, so the
// phases of "N - ja" and "M - ja" are those of bins ja and M + ja (only ever true for one lane of one slot)
template <int LOG2N, bool DC = false>
__device__ __forceinline__ void pair_regs_pk(v2f A, v2f Bp, v2f w, uint32_t x1, PhaseKey key, v2f &VA,
                                             v2f &VB, bool dc = false) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    const uint32_t cM = M * key.mul + 2u * key.k0;
    const float nkappa = -0.25f / (float)N;
    const v2f cj = {1.0f, -1.0f}, jc = {-1.0f, 1.0f};
    const v2f Bc = Bp * cj;                                    // conj(Bp)
    const v2f E = A + Bc, D = A - Bc;                          // 2E, 2D
    const v2f T = vcmul(D, w);                                 // T = w D
    // U = (X1.x, X2c.x) = (ex + ty, ex - ty), V = (X1.y, X2c.y) = (ey - tx, ey + tx)
    const v2f U = __builtin_shufflevector(E, E, 0, 0) + __builtin_shufflevector(T, T, 1, 1) * cj;
    const v2f V = __builtin_shufflevector(E, E, 1, 1) + __builtin_shufflevector(T, T, 0, 0) * jc;
    const v2f q2 = __builtin_elementwise_fma(V, V, U * U);     // (|X1|^2, |X2c|^2)
    const v2f mm = v2f{__builtin_amdgcn_sqrtf(q2.x), __builtin_amdgcn_sqrtf(q2.y)} * v2f{nkappa, nkappa};
    v2f cs1, cs2, cs3, cs4;
    phase_cs2_x(x1, cs1, cs4);       // bins ja and M + ja
    phase_cs2_x(cM - x1, cs3, cs2);  // bins M - ja and N - ja
    if (DC) {
        cs2 = vsel(dc, cs1, cs2);
        cs3 = vsel(dc, cs4, cs3);
    }
    const v2f P0 = cs1 + cs2 * cj;   // (c1 + c2, s1 - s2)
    const v2f Q0 = cs4 + cs3 * cj;   // (c4 + c3, s4 - s3)
    const v2f m1 = __builtin_shufflevector(mm, mm, 0, 0), m2 = __builtin_shufflevector(mm, mm, 1, 1);
    const v2f Pz = P0 * m1;
    const v2f S = __builtin_elementwise_fma(Q0, m2, Pz);
    const v2f R = __builtin_elementwise_fma(Q0, -m2, Pz);
    // Uc = conj(w) R = (rx wx + ry wy, ry wx - rx wy)
    const v2f t0 = R * __builtin_shufflevector(w, w, 0, 0);
    const v2f Uc = __builtin_elementwise_fma(__builtin_shufflevector(R, R, 1, 0),
                                             __builtin_shufflevector(w, w, 1, 1) * cj, t0);
    const v2f Us = __builtin_shufflevector(Uc, Uc, 1, 0);      // (uy, ux)
    VA = S + Us * jc;                                          // (sx - uy, sy + ux)
    VB = Us + S * cj;                                          // (sx + uy, ux - sy)
}

// hop4's variant of the pair: the same algebra with (a) the 1/(4N) scale left out (hop4 folds it into the
// synthesis window constants - an exact power of two), (b) the two complex products written with VOP3P
// source modifiers instead of materialised (-w.y, w.x) / (w.y, -w.y) operands.
//   cmul_fma(a, w, t)   = t + a.yy * (-w.y, w.x)      -> with t = a.xx * w this is a * w
//   cmulc_fma(a, w, t)  = t + a.yx * (w.y, -w.y)      -> with t = a * w.xx this is a * conj(w)
__device__ __forceinline__ v2f cmul_fma(v2f a, v2f w, v2f t) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ v2f cmulc_fma(v2f a, v2f w, v2f t) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// BAND: gq = the gains of bins ja and M - ja (the curated band-mask kernel RC_DK_BAND, fused: |g X| = |g| |X|)
template <int LOG2N, bool DC = false, bool BAND = false>
__device__ __forceinline__ void pair_regs_pk4(v2f A, v2f Bp, v2f w, uint32_t x1, PhaseKey key, v2f &VA,
                                              v2f &VB, bool dc = false, v2f gq = v2f{1.0f, 1.0f}) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    const uint32_t cM = M * key.mul + 2u * key.k0;
    const v2f cj = {1.0f, -1.0f}, jc = {-1.0f, 1.0f};
    const v2f Bc = Bp * cj;                                    // conj(Bp) (fused into E, D by the compiler)
    const v2f E = A + Bc, D = A - Bc;                          // 2E, 2D
    const v2f T = cmul_fma(D, w, __builtin_shufflevector(D, D, 0, 0) * w);  // T = w D
    const v2f U = __builtin_shufflevector(E, E, 0, 0) + __builtin_shufflevector(T, T, 1, 1) * cj;
    const v2f V = __builtin_shufflevector(E, E, 1, 1) + __builtin_shufflevector(T, T, 0, 0) * jc;
    const v2f q2 = __builtin_elementwise_fma(V, V, U * U);     // (|X1|^2, |X2c|^2)
    v2f mm = v2f{__builtin_amdgcn_sqrtf(q2.x), __builtin_amdgcn_sqrtf(q2.y)};
    if constexpr (BAND) mm = mm * gq;
    v2f cs1, cs2, cs3, cs4;
    phase_cs2_x(x1, cs1, cs4);       // bins ja and M + ja
    phase_cs2_x(cM - x1, cs3, cs2);  // bins M - ja and N - ja
    if (DC) {
        cs2 = vsel(dc, cs1, cs2);
        cs3 = vsel(dc, cs4, cs3);
    }
    const v2f P0 = cs1 + cs2 * cj;   // (c1 + c2, s1 - s2)
    const v2f Q0 = cs4 + cs3 * cj;   // (c4 + c3, s4 - s3)
    const v2f m1 = __builtin_shufflevector(mm, mm, 0, 0), m2 = __builtin_shufflevector(mm, mm, 1, 1);
    const v2f Pz = P0 * m1;
    const v2f S = __builtin_elementwise_fma(Q0, m2, Pz);
    const v2f R = __builtin_elementwise_fma(Q0, -m2, Pz);
    const v2f Uc = cmulc_fma(R, w, R * __builtin_shufflevector(w, w, 0, 0));  // conj(w) R
    const v2f Us = __builtin_shufflevector(Uc, Uc, 1, 0);      // (uy, ux)
    VA = S + Us * jc;                                          // (sx - uy, sy + ux)
    VB = Us + S * cj;                                          // (sx + uy, ux - sy)
}

// The Hermitian fold as a PRODUCT (round 6). Without a frequency kernel |X[j]| = |X[N - j]| = m, so the folded bin is
//   Zs[j] = m (e^{i th_j} + e^{-i th_{N-j}}) / 2 = m cos((th_j + th_{N-j}) / 2) e^{i (th_j - th_{N-j}) / 2}
// (src/fft.rs:63-69 draws both phases; nothing about the draws changes): three transcendentals per folded bin instead
// of four - v_cos of the half sum, v_cos / v_sin of the half difference - and the packed add of the two phasors becomes
// a multiply. The half angles come straight from the draws (phase_g2_x): fa = 0.5 + (r1 + r2) / 2 is rounded to the
// 2^-24 grid of [0.5, 1) (<= 2^-25 of a revolution = 1.9e-7 rad off), the half difference is exact.
// v_cos(fa) = -cos(half sum): the result is HALF of pair_regs_pk4's (same sign), so callers scale by -1/(2N).
template <int LOG2N, bool DC = false, bool BAND = false>
__device__ __forceinline__ void pair_regs_pk5(v2f A, v2f Bp, v2f w, uint32_t x1, PhaseKey key, v2f &VA,
                                              v2f &VB, bool dc = false, v2f gq = v2f{1.0f, 1.0f}) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    const uint32_t cM = M * key.mul + 2u * key.k0;
    const v2f cj = {1.0f, -1.0f}, jc = {-1.0f, 1.0f};
    const v2f Bc = Bp * cj;                                    // conj(Bp) (fused into E, D by the compiler)
    const v2f E = A + Bc, D = A - Bc;                          // 2E, 2D
    const v2f T = cmul_fma(D, w, __builtin_shufflevector(D, D, 0, 0) * w);  // T = w D
    const v2f U = __builtin_shufflevector(E, E, 0, 0) + __builtin_shufflevector(T, T, 1, 1) * cj;
    const v2f V = __builtin_shufflevector(E, E, 1, 1) + __builtin_shufflevector(T, T, 0, 0) * jc;
    const v2f q2 = __builtin_elementwise_fma(V, V, U * U);     // (|X1|^2, |X2c|^2)
    v2f mm = v2f{__builtin_amdgcn_sqrtf(q2.x), __builtin_amdgcn_sqrtf(q2.y)};
    if constexpr (BAND) mm = mm * gq;
    float gl1, gu4, gl3, gu2;
    phase_g2_x(x1, gl1, gu4);        // bins ja and M + ja
    phase_g2_x(cM - x1, gl3, gu2);   // bins M - ja and N - ja
    const v2f GU = {gu2, gu4}, GL = {gl1, gl3};
    // pair 1 = bins (ja, N - ja): half sum / half difference of (th1, th2); pair 2 = bins (M + ja, M - ja): of (th4, th3)
    v2f FA = __builtin_elementwise_fma(GU, v2f{0.5f, 0.5f}, GL);
    v2f FB = __builtin_elementwise_fma(GU, v2f{-0.5f, 0.5f}, GL * cj);
    if (DC) {  // bin 0 with itself: th2 := th1, th3 := th4 (half sums th1, th4; half differences 0)
        FA = vsel(dc, v2f{gl1 + gl1, gu4}, FA);
        FB = vsel(dc, v2f{0.0f, 0.0f}, FB);
    }
    const v2f ca = {__builtin_amdgcn_cosf(FA.x), __builtin_amdgcn_cosf(FA.y)};
    const v2f E1 = {__builtin_amdgcn_cosf(FB.x), __builtin_amdgcn_sinf(FB.x)};
    const v2f E2 = {__builtin_amdgcn_cosf(FB.y), __builtin_amdgcn_sinf(FB.y)};
    const v2f km = ca * mm;
    const v2f k1 = __builtin_shufflevector(km, km, 0, 0), k2 = __builtin_shufflevector(km, km, 1, 1);
    const v2f Pz = E1 * k1;
    const v2f S = __builtin_elementwise_fma(E2, k2, Pz);
    const v2f R = __builtin_elementwise_fma(E2, -k2, Pz);
    const v2f Uc = cmulc_fma(R, w, R * __builtin_shufflevector(w, w, 0, 0));  // conj(w) R
    const v2f Us = __builtin_shufflevector(Uc, Uc, 1, 0);      // (uy, ux)
    VA = S + Us * jc;                                          // (sx - uy, sy + ux)
    VB = Us + S * cj;                                          // (sx + uy, ux - sy)
}

// ---- default-window fast path: windows::hanning (src/windows.rs:4-9) and the crossfade envelope
// (src/crossfade.rs:4-10) are both 0.5 - c cos(2 pi i / (len - 1)). Thread t touches samples
// i = 512 q + 2 t + e, so cos(alpha_q + beta_te) = cos alpha_q cos beta_te - sin alpha_q sin beta_te:
// the 32 (16) alpha terms are compile-time constants, the beta terms 4 (+4) floats per thread from
// HopParams::hann_rot. Two FMAs per sample replace a table load (the loads were 70 % of the
// kernel's vector-memory traffic).
constexpr double cx_sin_taylor(double x) {  // |x| <= pi/2
    double term = x, sum = x;
    for (int n = 1; n < 16; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double CX_PI = 3.14159265358979323846264338327950288;
constexpr double cx_sin(double x) {  // 0 <= x < 2 pi + eps
    while (x > CX_PI) x -= 2.0 * CX_PI;
    if (x > CX_PI / 2) x = CX_PI - x;
    if (x < -CX_PI / 2) x = -CX_PI - x;
    return cx_sin_taylor(x);
}
constexpr double cx_cos(double x) { return cx_sin(x + CX_PI / 2); }
struct HannK {
    float c[32], s[32];
};
// c[q] = -amp cos(2 pi 512 q / (len - 1)), s[q] = amp sin(...): value(i) = 0.5 + c[q] cb + s[q] sb
constexpr HannK make_hann_k(double amp, int len, int count) {
    HannK k{};
    for (int q = 0; q < 32; ++q) {
        const double a = q < count ? 2.0 * CX_PI * 512.0 * q / (double)(len - 1) : 0.0;
        k.c[q] = (float)(-amp * cx_cos(a));
        k.s[q] = (float)(amp * cx_sin(a));
    }
    return k;
}
constexpr double cx_sqrt(double x) {
    double r = x > 1 ? x : 1.0;
    for (int i = 0; i < 64; ++i) r = 0.5 * (r + x / r);
    return r;
}
constexpr double HANN_ENV_AMP = 1.0 - (1.0 + cx_sqrt(cx_sqrt(0.5))) * 0.5;  // crossfade.rs:5
__device__ constexpr HannK HANN_W14 = make_hann_k(0.5, 16384, 32);
__device__ constexpr HannK HANN_E14 = make_hann_k(HANN_ENV_AMP, 8192, 16);
// synthesis window times -1/(4N) = -2^-16 (hop4: the scale of the magnitudes, src/fft.rs:72's / N and the sign
// of the negated phasors, moved out of the per-bin stage; a power of two, so nothing rounds differently)
// (round 6: pair_regs_pk5 returns half of pair_regs_pk4's values, so with it the scale is -1/(2N) = -2^-15)
#ifndef RC_FOLDPROD
#define RC_FOLDPROD 1  // the Hermitian fold as a product (pair_regs_pk5); 0: as a sum of two phasors (pair_regs_pk4), for A/B
#endif
constexpr double HANN_KAPPA = (RC_FOLDPROD ? -0.5 : -0.25) / 16384.0;
__device__ constexpr HannK HANN_W14K = make_hann_k(0.5 * HANN_KAPPA, 16384, 32);

}  // namespace

// ---- decimating stores with the pitch known at compile time (F[t] = O[t * pitch], src/resampler.rs:3-18) ----------
// A thread owns samples a = a00 + ROW * q + e of the hop (q = register pair, e = 0 / 1, a00 = (g0 % PC) + 2 t, ROW = 2 T
// samples per register row). With r0 = a00 % PC and d0 = a00 / PC, sample (q, e) is kept iff r0 == (PC - (ROW q + e) % PC)
// % PC =: c, and then lands at d0 + (c + ROW q + e) / PC - a constant per (q, e). So the whole index arithmetic of a hop
// is PC selects (byte offset 4 d0 for the lanes of residue class c, an out-of-range offset for the others: the raw
// buffer store drops them) and every store takes one of those PC registers plus a constant scalar offset.
template <int PC>
struct PitchOffsets {
    uint32_t off[PC];
};
constexpr uint32_t PITCH_DROP = 0x80000000u;  // beyond the 1 GiB window of the buffer resource, no wrap with the constants
template <int PC>
__device__ __forceinline__ PitchOffsets<PC> pitch_offsets(uint32_t a00) {
    const uint32_t d0 = a00 / (uint32_t)PC, r0 = a00 - d0 * (uint32_t)PC;
    PitchOffsets<PC> o;
#pragma unroll
    for (int c = 0; c < PC; ++c) o.off[c] = r0 == (uint32_t)c ? 4u * d0 : PITCH_DROP;
    return o;
}
template <int PC, int ROW>
__device__ __forceinline__ void pitch_store_pair(const __amdgpu_buffer_rsrc_t rsrc, const PitchOffsets<PC> &o, const int q,
                                                 const float ox, const float oy) {
    const int ax = ROW * q, ay = ROW * q + 1;
    const int cx = (PC - ax % PC) % PC, cy = (PC - ay % PC) % PC;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ox), rsrc, o.off[cx], 4 * ((cx + ax) / PC), 0);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(oy), rsrc, o.off[cy], 4 * ((cy + ay) / PC), 0);
}

}  // namespace rc
