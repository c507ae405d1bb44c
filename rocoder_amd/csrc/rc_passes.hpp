// Bit-group passes of the first-generation fused kernel (hop_kernel, N <= 16384) and of the quarter transforms
// of the large-window pipeline (big_a / big_c / big_cr): in-register radix-2 stages + padded LDS exchanges.
#pragma once
#include "rc_dev.hpp"

#ifndef RC_PASS_SQ
#define RC_PASS_SQ 1
#endif
namespace rc {
namespace {

// Per-thread state that is worth keeping in registers across the hops of a run.
template <class G>
struct ThreadCtx {
    int tid;
    int lb[G::m + 1];  // padded LDS base index per register layout LO (unused entries fold away)
};
template <class G, int LO>
__device__ __forceinline__ void fill_lds_bases(ThreadCtx<G> &c) {
    if constexpr (LO >= 0) {
        c.lb[LO] = (LO + G::B <= G::m) ? pad_idx(pos_of<G::B, LO>(c.tid, 0)) : 0;
        fill_lds_bases<G, LO - 1>(c);
    }
}

// One in-register pass: radix-2 stages on absolute bits S_LO..S_HI, register bit r = s - LOR.
// Forward (DIF):  a' = a + b,           b' = (a - b) w
// Inverse (DIT):  a' = a + conj(w) b,   b' = a - conj(w) b      (conjugate transpose of DIF)
// w = exp(-2 pi i (n mod 2^s) / 2^(s+1)) = base_s(thread) * W32^(c * 16 >> r)
template <class G, int LOR, int S_LO, int S_HI, bool INV>
__device__ __forceinline__ void run_pass(float2 (&v)[G::P], int tid,
                                         GV2 wtab) {
    const int l = tid & ((1 << LOR) - 1);
    // the base twiddles of all stages of the pass are requested together and waited on once (each
    // used to be waited on in place, one memory latency per stage); the opaque copies keep them from
    // being hoisted out of the hop loop into live registers
    float2 bases[S_HI - S_LO + 1];
    if (LOR > 0) {
        if (RC_PASS_SQ) {
            // TWO table loads per pass (the bases of the finest stage and of the one two below it) instead of five; the
            // others are squares: base(s - 1) = base(s)^2, two packed instructions, at most two deep from a table value
            // (each squaring doubles the f32 rounding of its input: four deep cost 3e-6 of the output's RMS on the
            // large-window spectrum path, two deep stays under 1e-6). A global load inside the hop loop is waited on in
            // place and retires in order behind the previous hop's output stores.
            constexpr int NS = S_HI - S_LO + 1;
            float2 bh = ldg2(wtab + (l << (G::m - 1 - S_HI)));
            float2 bm = NS > 2 ? ldg2(wtab + (l << (G::m - 1 - (S_HI - 2)))) : bh;
            opaque(bh);
            opaque(bm);
            bases[NS - 1] = bh;
            if (NS > 2) bases[NS - 3] = bm;
#pragma unroll
            for (int si = NS - 2; si >= 0; --si) {
                if (NS > 2 && si == NS - 3) continue;  // (loaded)
                const v2f a = to_v(bases[si + 1]);
                const v2f t = __builtin_shufflevector(a, a, 0, 0) * a;
                v2f sq;  // t + a.yy * (-a.y, a.x) = a * a
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(sq) : "v"(a), "v"(a), "v"(t));
                bases[si] = to_f2(sq);
            }
        } else {
#pragma unroll
            for (int si = 0; si <= S_HI - S_LO; ++si) bases[si] = ldg2(wtab + (l << (G::m - 1 - (S_LO + si))));
#pragma unroll
            for (int si = 0; si <= S_HI - S_LO; ++si) opaque(bases[si]);
        }
    }
#pragma unroll
    for (int si = 0; si <= S_HI - S_LO; ++si) {
        const int s = INV ? (S_LO + si) : (S_HI - si);
        const int r = s - LOR;
        const int half = 1 << r;
        float2 base = make_float2(1.f, 0.f);
        if (LOR > 0) base = bases[s - S_LO];
#pragma unroll
        for (int q0 = 0; q0 < G::P; ++q0) {
            if (q0 & half) continue;
            const int q1 = q0 | half;
            const int c = q0 & (half - 1);
            const int kidx = c * (16 >> r);
            const float2 a = v[q0], b = v[q1];
#if RC_PK
            // packed (re,im) arithmetic as plain vector code: hipcc emits v_pk_* with op_sel / neg /
            // SGPR-constant operands itself (no inline-asm boundary pads)
            const v2f av = to_v(a), bv = to_v(b);
            const v2f kc = {W32_RE[kidx & 15], W32_IM[kidx & 15]};
            const v2f two = {2.0f, 2.0f};
            if (LOR == 0 && c == 0) {  // w = 1
                v[q0] = to_f2(av + bv);
                v[q1] = to_f2(av - bv);
            } else if (LOR == 0 && kidx == 8) {  // w = -i
                const v2f ibm = __builtin_shufflevector(bv, bv, 1, 0) * v2f{-1.0f, 1.0f};  // i b
                if (!INV) {  // (a - b)(-i) = -i a + i b
                    const v2f iam = __builtin_shufflevector(av, av, 1, 0) * v2f{-1.0f, 1.0f};
                    v[q0] = to_f2(av + bv);
                    v[q1] = to_f2(ibm - iam);
                } else {  // a +- i b
                    v[q0] = to_f2(av + ibm);
                    v[q1] = to_f2(av - ibm);
                }
            } else {
                v2f w;
                if (LOR == 0) w = kc;
                else if (c == 0) w = to_v(base);
                else if (kidx == 8) w = v2f{base.y, -base.x};
                else {
                    const v2f bs = to_v(base);
                    const v2f t0 = __builtin_shufflevector(bs, bs, 0, 0) * kc;
                    w = __builtin_elementwise_fma(__builtin_shufflevector(bs, bs, 1, 1),
                                                  v2f{-kc.y, kc.x}, t0);
                }
                if (!INV) {  // b' = (a - b) w = d.xx * w + d.yy * (-w.y, w.x)
                    const v2f d = av - bv;
                    const v2f wm = __builtin_shufflevector(w, w, 1, 0) * v2f{-1.0f, 1.0f};
                    const v2f t0 = __builtin_shufflevector(d, d, 0, 0) * w;
                    v[q0] = to_f2(av + bv);
                    v[q1] = to_f2(__builtin_elementwise_fma(__builtin_shufflevector(d, d, 1, 1), wm, t0));
                } else {  // a' = a + conj(w) b, b' = 2a - a'
                    const v2f w2 = __builtin_shufflevector(w, w, 1, 1) * v2f{1.0f, -1.0f};
                    const v2f t0 = __builtin_elementwise_fma(bv, __builtin_shufflevector(w, w, 0, 0), av);
                    const v2f rv = __builtin_elementwise_fma(__builtin_shufflevector(bv, bv, 1, 0), w2, t0);
                    v[q0] = to_f2(rv);
                    v[q1] = to_f2(__builtin_elementwise_fma(av, two, -rv));
                }
            }
#else
            if (LOR == 0 && c == 0) {  // w = 1
                v[q0] = make_float2(a.x + b.x, a.y + b.y);
                v[q1] = make_float2(a.x - b.x, a.y - b.y);
            } else if (LOR == 0 && kidx == 8) {  // w = -i
                if (!INV) {
                    v[q0] = make_float2(a.x + b.x, a.y + b.y);
                    const float dx = a.x - b.x, dy = a.y - b.y;
                    v[q1] = make_float2(dy, -dx);  // d * (-i)
                } else {
                    const float tx = -b.y, ty = b.x;  // (+i) * b
                    v[q0] = make_float2(a.x + tx, a.y + ty);
                    v[q1] = make_float2(a.x - tx, a.y - ty);
                }
            } else {
                float2 w;
                if (c == 0) w = base;
                else if (LOR == 0) w = make_float2(W32_RE[kidx], W32_IM[kidx]);
                else if (kidx == 8) w = make_float2(base.y, -base.x);
                else w = cmul(base, make_float2(W32_RE[kidx], W32_IM[kidx]));
                if (!INV) {
                    v[q0] = make_float2(a.x + b.x, a.y + b.y);
                    const float dx = a.x - b.x, dy = a.y - b.y;
                    v[q1] = make_float2(dx * w.x - dy * w.y, dx * w.y + dy * w.x);
                } else {
                    // a + conj(w) b in 4 FMAs, a - conj(w) b = 2a - (a + conj(w) b) in 2
                    const float rx = fmaf(w.y, b.y, fmaf(w.x, b.x, a.x));
                    const float ry = fmaf(-w.y, b.x, fmaf(w.x, b.y, a.y));
                    v[q0] = make_float2(rx, ry);
                    v[q1] = make_float2(fmaf(2.f, a.x, -rx), fmaf(2.f, a.y, -ry));
                }
            }
#endif
        }
    }
}

// forward passes, high bits first. On return v is in register layout last_lor.
template <class G, int PREV, int PREV_LOR, bool FIRST, int SID = 1>
__device__ __forceinline__ void forward_passes(float2 (&v)[G::P], float2 *lds,
                                               const ThreadCtx<G> &c,
                                               GV2 wtab, Stamps &st) {
    if constexpr (PREV > 0) {
        constexpr int lo = lo_of<G>(PREV);
        constexpr int LOR = lor_of<G>(PREV);
        if constexpr (!FIRST) {
            lds_store<G, PREV_LOR>(v, lds, c.lb[PREV_LOR]);
            __syncthreads();
            st.mark(SID);
            lds_load<G, LOR>(v, lds, c.lb[LOR]);
            __syncthreads();
            st.mark(SID + 1);
        }
        run_pass<G, LOR, lo, PREV - 1, false>(v, c.tid, wtab);
        st.mark(SID + 2);
        forward_passes<G, lo, LOR, false, SID + 3>(v, lds, c, wtab, st);
    }
}

// inverse passes, low bits first. Expects v loaded in layout last_lor; returns layout LO0.
template <class G, int PREV, int SID = 16>
__device__ __forceinline__ void inverse_passes(float2 (&v)[G::P], float2 *lds,
                                               const ThreadCtx<G> &c,
                                               GV2 wtab, Stamps &st) {
    if constexpr (PREV > 0) {
        constexpr int lo = lo_of<G>(PREV);
        constexpr int LOR = lor_of<G>(PREV);
        if constexpr (lo > 0) {
            inverse_passes<G, lo, SID + 3>(v, lds, c, wtab, st);
            constexpr int LOR_DEEPER = lor_of<G>(lo);
            lds_store<G, LOR_DEEPER>(v, lds, c.lb[LOR_DEEPER]);
            __syncthreads();
            st.mark(SID);
            lds_load<G, LOR>(v, lds, c.lb[LOR]);
            __syncthreads();
            st.mark(SID + 1);
        }
        run_pass<G, LOR, lo, PREV - 1, true>(v, c.tid, wtab);
        st.mark(SID + 2);
    }
}

// a_k[n] = x[k*step + n] * w[n] for this thread's 2P samples (n = tid + T q -> samples 2n, 2n+1).
// Hops whose window runs past the end of the closed input (zero padding, stretcher.rs:129-132)
// read from the engine's zero-padded tail copy instead, so there is no per-element bounds test.
// Addresses are (uniform pointer + 2 T q) + 32-bit lane offset: SGPR base + VGPR offset loads,
// no per-register 64-bit address VGPRs.
template <int LOG2N>
__device__ __forceinline__ void load_hop(float2 (&v)[Geo<LOG2N>::P], const HopParams &p,
                                         GF xc, GF xt, GF win, int64_t k, unsigned lane2) {
    using G = Geo<LOG2N>;
    // uniform source pointer: force it into SGPRs (the select may otherwise be done in VALU)
    const int64_t off = (k >= p.tail_hop_first) ? (k * (int64_t)p.step - p.tail_origin)
                                                : (k * (int64_t)p.step - p.in_origin);
    const unsigned long long sa =
        (unsigned long long)((k >= p.tail_hop_first) ? xt : xc) + (unsigned long long)off * 4ull;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sa);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
    GF src = (GF)(((unsigned long long)hi << 32) | lo);
    // issue every load of the hop (input + window), then one wait, then the multiplies
    float xr0[G::P], xr1[G::P], wr0[G::P], wr1[G::P];
#pragma unroll
    for (int q = 0; q < G::P; ++q) {
        GF sq = src + 2 * G::T * q;
        xr0[q] = sq[lane2];
        xr1[q] = sq[lane2 + 1];
    }
#pragma unroll
    for (int q = 0; q < G::P; ++q) {
        GF wq = win + 2 * G::T * q;
        wr0[q] = wq[lane2];
        wr1[q] = wq[lane2 + 1];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < G::P; ++q) v[q] = make_float2(xr0[q] * wr0[q], xr1[q] * wr1[q]);
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace
}  // namespace rc
