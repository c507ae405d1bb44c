// gfx950 (MI355X / CDNA4) kernels of the rocoder stretch hot path.
//
// One workgroup owns a contiguous run of hops of one channel and, per hop k, computes
//   a_k[n] = x[k*step+n] * w[n]                       (src/fft.rs:51-55)
//   X_k    = DFT_N(a_k)                                (src/fft.rs:59)      real->complex, N/2-pt
//   Z_k[j] = |X_k[j]| * e^{i theta(seed,c,k,j)}        (src/fft.rs:65-68)   all N bins drawn
//   y_k[n] = Re(IDFT_N(Z_k))[n] / N * w[n]             (src/fft.rs:69-73)   complex->real, N/2-pt
//   O[kH+i] = (y_k[i] + y_{k-1}[H+i]) * env[i] * amp   (src/stretcher.rs:96-103)
//   F[t]    = O[t*p]                                   (src/stretcher.rs:108-111, resampler.rs:15-18)
//
// Layout of one hop inside the workgroup (M = N/2 complex points, T threads, P = M/T
// register-resident points per thread):
//   * the N/2-point FFT runs as bit-group passes: each pass transforms up to log2(P) index bits
//     entirely in registers (radix-2 butterflies with compile-time 32nd roots x one per-thread
//     base twiddle per stage), passes exchange through LDS (padded 1 complex per 32: conflict-free
//     ds_read/write_b64 for every pass layout); forward is DIF (natural in, bit-reversed out),
//     inverse is the mirrored DIT, so no reordering pass exists;
//   * the real<->complex split, magnitude, random phasors and the Hermitian fold happen on the
//     bit-reversed spectrum in LDS, a "quad" {j, j+M/2, M/2-j, M-j} per slot;
//   * the window multiply and the two-term overlap-add stay in registers: a thread's tail samples
//     of hop k line up with its head samples of hop k+1, so the run carries y_{k-1}[H..] in VGPRs
//     and the first hop of a run is recomputed (phases are a pure function of (seed,c,k,j)).
// No MFMA: this is an FFT/SFU/LDS-bound path, not a contraction.
#include "rc_kernels.h"

#include <algorithm>

namespace rc {
namespace {

// exp(-2 pi i c / 32), c = 0..15
__device__ constexpr float W32_RE[16] = {
    1.f, 0.980785251f, 0.923879504f, 0.831469595f, 0.707106769f, 0.555570245f, 0.382683426f,
    0.195090324f, 0.f, -0.195090324f, -0.382683426f, -0.555570245f, -0.707106769f, -0.831469595f,
    -0.923879504f, -0.980785251f};
__device__ constexpr float W32_IM[16] = {
    -0.f, -0.195090324f, -0.382683426f, -0.555570245f, -0.707106769f, -0.831469595f,
    -0.923879504f, -0.980785251f, -1.f, -0.980785251f, -0.923879504f, -0.831469595f,
    -0.707106769f, -0.555570245f, -0.382683426f, -0.195090324f};

// Pointers that arrive inside the by-value HopParams struct are generic (flat) pointers to the
// compiler; flat loads tie up both memory counters and cannot use SGPR-base addressing. Everything
// the engine passes is device global memory, so the kernels cast once to address space 1.
typedef float v2f __attribute__((ext_vector_type(2)));
#define RC_AS1 __attribute__((address_space(1)))
using GF = const float RC_AS1 *;    // global const float*
using GFW = float RC_AS1 *;         // global float*
using GV2 = const v2f RC_AS1 *;     // global const float2*
using GV2W = v2f RC_AS1 *;          // global float2*
__device__ __forceinline__ float2 ldg2(GV2 p) {
    const v2f t = *p;
    return make_float2(t.x, t.y);
}
__device__ __forceinline__ void stg2(GV2W p, float2 v) {
    v2f t;
    t.x = v.x;
    t.y = v.y;
    *p = t;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int clog2(int v) { return v <= 1 ? 0 : 1 + clog2(v >> 1); }

// points per thread for the large windows (tunable: 32 -> 256 threads, 2 waves/SIMD, 3 passes;
// 16 -> 512 threads, 4 waves/SIMD, 4 passes)
// Timing-only diagnostic builds (results are wrong): bit 0 = no phase hash/sincos, bit 1 = no global
// loads/stores, bit 2 = no LDS exchanges/barriers, bit 3 = no butterflies, bit 4 = no middle stage.
#ifndef RC_ABLATE
#define RC_ABLATE 0
#endif
#ifndef RC_LOADCH
#define RC_LOADCH 32
#endif
#ifndef RC_PMAX
#define RC_PMAX 32
#endif

template <int LOG2N>
struct Geo {
    static constexpr int m = LOG2N - 1;        // log2 of complex length
    static constexpr int M = 1 << m;           // complex points
    static constexpr int N = 2 * M;            // window length
    static constexpr int T = cmax(M / RC_PMAX, cmin(64, M / 4));  // threads per workgroup
    static constexpr int WPS = T >= 512 ? 4 : 2;  // waves per SIMD the register budget targets
    static constexpr int P = M / T;            // points per thread
    static constexpr int B = clog2(P);         // index bits per pass
    static constexpr int LDS_FLOAT2 = M + (M >> 5) + 1;
    static constexpr int LO0 = m - B;          // register layout of the first/last (global) pass
    static constexpr int QN = cmax(1, (M / 4) / T);  // middle-stage quad slots per thread
};

// pass k transforms absolute index bits [lo_of(prev), prev-1]; its registers hold bits
// [lor_of(prev), lor_of(prev)+B-1]
template <class G> constexpr int lo_of(int prev) { return cmax(0, prev - G::B); }
template <class G> constexpr int lor_of(int prev) {
    return (prev - lo_of<G>(prev) == G::B) ? lo_of<G>(prev) : 0;
}
template <class G> constexpr int last_lor(int prev) {
    return lo_of<G>(prev) == 0 ? lor_of<G>(prev) : last_lor<G>(lo_of<G>(prev));
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---- packed (re,im) arithmetic --------------------------------------------------------------
// A lone wave issues one VALU op per ~4.6 cycles whether it is packed or not (profiles/
// r01_ubench_instruction_rates.txt), so v_pk_*_f32 on (re,im) pairs halves the time a wave needs per
// butterfly whenever its SIMD partner is waiting on LDS / memory. The butterflies are written as
// plain ext-vector code: hipcc selects v_pk_fma/mul/add with op_sel, neg and SGPR/inline constants
// by itself, so there are no inline-asm boundary pads and the scheduler is free.
#ifndef RC_PK
#define RC_PK 1
#endif
__device__ __forceinline__ v2f to_v(float2 a) {
    v2f r;
    r.x = a.x;
    r.y = a.y;
    return r;
}
__device__ __forceinline__ float2 to_f2(v2f a) { return make_float2(a.x, a.y); }

constexpr int pad_idx(int n) { return n + (n >> 5); }

template <int B, int LO>
constexpr int pos_of(int tid, int q) {
    return ((tid >> LO) << (LO + B)) | (q << LO) | (tid & ((1 << LO) - 1));
}
// The index fields (l, q, u) occupy disjoint bit ranges, so the padded LDS index splits into a
// per-thread base (one VGPR, live across the run) plus a compile-time offset per register q
// (folded into the ds_read/ds_write immediate): pad(pos(tid,q)) = pad(pos(tid,0)) + pad(pos(0,q)).
template <int B, int LO>
constexpr int lds_reg_off(int q) { return pad_idx(pos_of<B, LO>(0, q)); }

template <class G, int LO>
__device__ __forceinline__ void lds_store(const float2 (&v)[G::P], float2 *lds, int base) {
    if (RC_ABLATE & 4) return;
#pragma unroll
    for (int q = 0; q < G::P; ++q) lds[base + lds_reg_off<G::B, LO>(q)] = v[q];
}
template <class G, int LO>
__device__ __forceinline__ void lds_load(float2 (&v)[G::P], const float2 *lds, int base) {
    if (RC_ABLATE & 4) return;
#pragma unroll
    for (int q = 0; q < G::P; ++q) v[q] = lds[base + lds_reg_off<G::B, LO>(q)];
}

// Opaque copies: stop LICM from hoisting per-hop recomputable values (twiddle products, table
// loads, slot addresses) out of the hop loop into hundreds of live VGPRs.
__device__ __forceinline__ void opaque(float2 &x) { asm volatile("" : "+v"(x.x), "+v"(x.y)); }
__device__ __forceinline__ void opaque(int &x) { asm volatile("" : "+v"(x)); }

// Diagnostic phase stamps (RC_STAMP builds only; never in the product build): per-wave cycle totals
// per phase id, written to the debug buffer passed in HopParams::spec.
#ifndef RC_STAMP
#define RC_STAMP 0
#endif
#ifndef RC_SWP
#define RC_SWP 2
#endif
struct Stamps {
#if RC_STAMP
    unsigned long long last;
    unsigned acc[32];
    __device__ __forceinline__ void init() {
        for (int i = 0; i < 32; ++i) acc[i] = 0;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last)::"memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void mark(int id) {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        acc[id] += (unsigned)(t - last);
        last = t;
    }
#else
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void mark(int) {}
#endif
};

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
#pragma unroll
        for (int si = 0; si <= S_HI - S_LO; ++si) bases[si] = ldg2(wtab + (l << (G::m - 1 - (S_LO + si))));
#pragma unroll
        for (int si = 0; si <= S_HI - S_LO; ++si) opaque(bases[si]);
    }
#pragma unroll
    for (int si = 0; si <= S_HI - S_LO; ++si) {
        const int s = INV ? (S_LO + si) : (S_HI - si);
        const int r = s - LOR;
        const int half = 1 << r;
        if (RC_ABLATE & 8) continue;
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
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(SID);
            lds_load<G, LOR>(v, lds, c.lb[LOR]);
            if (!(RC_ABLATE & 4)) __syncthreads();
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
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(SID);
            lds_load<G, LOR>(v, lds, c.lb[LOR]);
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(SID + 1);
        }
        run_pass<G, LOR, lo, PREV - 1, true>(v, c.tid, wtab);
        st.mark(SID + 2);
    }
}

// ---- phase source (spec shared with oracle/rocoder_oracle.c: rco_phase_*) -----------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct PhaseKey {
    uint32_t k0, mul;
};
__device__ __forceinline__ PhaseKey make_phase_key(uint64_t seed_mixed, uint32_t ch, int64_t hop) {
    const uint64_t ctr = ((uint64_t)ch << 40) | ((uint64_t)hop & 0xFFFFFFFFFFull);
    const uint64_t key = mix64(seed_mixed ^ ctr);
    PhaseKey k;
    k.k0 = (uint32_t)key;
    k.mul = (uint32_t)(key >> 32) | 1u;
    return k;
}
// One 32-bit hash serves the two bins b < M (its top 23 bits) and b + M (its low 16 bits), M = N/2:
//   theta(b)     = pi * (h >> 9)     * 2^-23          theta(b + M) = pi * (h & 0xFFFF) * 2^-16
// The phasors come out negated: v_cos/v_sin take revolutions, and f = 0.5 + theta / (2 pi) in
// [0.5, 1) is assembled in the mantissa (2 pi f = pi + theta).
__device__ __forceinline__ uint32_t phase_hash_x(uint32_t x) {
    x ^= x >> 16;
    x *= 0x21F0AAADu;
    x ^= x >> 15;
    x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}
__device__ __forceinline__ float phase_rev_lower(uint32_t h) { return __uint_as_float(0x3F000000u | (h >> 9)); }
__device__ __forceinline__ float phase_rev_upper(uint32_t h) {
    return __uint_as_float(0x3F000000u | ((h & 0xFFFFu) << 7));
}
// counter x = c * mul + k0 of c < M: (-cos, -sin) of bin c (lo*) and of bin c + M (up*)
__device__ __forceinline__ void phase_ncs2_x(uint32_t x, float &lo_nc, float &lo_ns, float &up_nc,
                                             float &up_ns) {
    if (RC_ABLATE & 1) {
        lo_nc = __uint_as_float(0x3F000000u | (x & 0xFFFFu));
        lo_ns = lo_nc + 1.0f;
        up_nc = lo_nc + 2.0f;
        up_ns = lo_nc + 3.0f;
        return;
    }
    const uint32_t h = phase_hash_x(x);
    const float fl = phase_rev_lower(h), fu = phase_rev_upper(h);
    lo_nc = __builtin_amdgcn_cosf(fl);
    lo_ns = __builtin_amdgcn_sinf(fl);
    up_nc = __builtin_amdgcn_cosf(fu);
    up_ns = __builtin_amdgcn_sinf(fu);
}
// the four phases of the pair (ja, M - ja), ja < M: bins ja, N - ja, M - ja, M + ja from the two
// hashes of counters ja and M - ja. ja == 0 wraps: N - 0 is bin 0 again and M - 0 is bin M.
__device__ __forceinline__ void phase_quad(PhaseKey k, uint32_t ja, uint32_t M, float &c1, float &s1,
                                           float &c2, float &s2, float &c3, float &s3, float &c4,
                                           float &s4) {
    const uint32_t ha = phase_hash_x(ja * k.mul + k.k0);
    const uint32_t hb = phase_hash_x(((M - ja) & (M - 1)) * k.mul + k.k0);
    const float f1 = phase_rev_lower(ha), f4 = phase_rev_upper(ha);
    const float fbl = phase_rev_lower(hb), fbu = phase_rev_upper(hb);
    const float f2 = ja ? fbu : fbl, f3 = ja ? fbl : fbu;
    c1 = __builtin_amdgcn_cosf(f1);
    s1 = __builtin_amdgcn_sinf(f1);
    c2 = __builtin_amdgcn_cosf(f2);
    s2 = __builtin_amdgcn_sinf(f2);
    c3 = __builtin_amdgcn_cosf(f3);
    s3 = __builtin_amdgcn_sinf(f3);
    c4 = __builtin_amdgcn_cosf(f4);
    s4 = __builtin_amdgcn_sinf(f4);
}

// ---- one (ja, M - ja) bin pair -----------------------------------------------------------
// analysis: A = Zf[ja], Bp = Zf[M-ja], w = exp(-2 pi i ja / N)
//   X1 = 2 X[ja], X2c = 2 conj(X[M-ja])
__device__ __forceinline__ void pair_analyze(float2 A, float2 Bp, float2 w, float2 &X1,
                                             float2 &X2c) {
    const float ex = A.x + Bp.x, ey = A.y - Bp.y;  // 2E = A + conj(Bp)
    const float dx = A.x - Bp.x, dy = A.y + Bp.y;  // 2D = A - conj(Bp)
    const float tx = dx * w.x - dy * w.y, ty = dx * w.y + dy * w.x;  // T = w D
    X1 = make_float2(ex + ty, ey - tx);   // E - iT
    X2c = make_float2(ex - ty, ey + tx);  // E + iT
}
// synthesis: magnitudes of bins ja, N-ja, M-ja, M+ja -> V[ja], V[M-ja] of the N/2-point c2r
//   Zs[j] = (|X[j]| e^{i th_j} + |X[N-j]| e^{-i th_{N-j}}) / 2 ; nkappa = -(scale) because the
//   phasors come negated.
template <int LOG2N>
__device__ __forceinline__ void pair_synth(float m1a, float m1b, float m2a, float m2b, float2 w,
                                           uint32_t ja, PhaseKey key, float nkappa, float2 &VA,
                                           float2 &VB) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    float c1, s1, c2, s2, c3, s3, c4, s4;
    phase_quad(key, ja, M, c1, s1, c2, s2, c3, s3, c4, s4);
    m1a *= nkappa;
    m1b *= nkappa;
    m2a *= nkappa;
    m2b *= nkappa;
    const float px = m1a * c1 + m1b * c2, py = m1a * s1 - m1b * s2;  // Zs[ja]
    const float qx = m2a * c3 + m2b * c4, qy = m2b * s4 - m2a * s3;  // conj(Zs[M-ja])
    const float sx = px + qx, sy = py + qy;
    const float rx = px - qx, ry = py - qy;
    const float ux = rx * w.x + ry * w.y, uy = ry * w.x - rx * w.y;  // U = conj(w) R
    VA = make_float2(sx - uy, sy + ux);  // S + iU
    VB = make_float2(sx + uy, ux - sy);  // conj(S - iU)
}

__device__ __forceinline__ float cabs_fast(float2 z) {
    return __builtin_amdgcn_sqrtf(z.x * z.x + z.y * z.y);
}

// pA / pB: padded LDS indices of bins ja and M - ja
template <int LOG2N, int MODE>
__device__ __forceinline__ void do_pair(float2 *lds, int pA, int pB, float2 w, uint32_t ja,
                                        PhaseKey key, GV2W spec) {
    constexpr uint32_t N = 1u << LOG2N, M = N / 2;
    float2 VA, VB;
    if constexpr (MODE == MODE_RESYNTH) {
        const float m1a = cabs_fast(ldg2(spec + ja));
        const float m1b = cabs_fast(ldg2(spec + ((N - ja) & (N - 1))));
        const float m2a = cabs_fast(ldg2(spec + (M - ja)));
        const float m2b = cabs_fast(ldg2(spec + ((M + ja) & (N - 1))));
        pair_synth<LOG2N>(m1a, m1b, m2a, m2b, w, ja, key, -0.5f / (float)N, VA, VB);
        lds[pA] = VA;
        if (pB != pA) lds[pB] = VB;
    } else {
        const float2 A = lds[pA];
        const float2 Bp = lds[pB];
        float2 X1, X2c;
        pair_analyze(A, Bp, w, X1, X2c);
        if constexpr (MODE == MODE_FORWARD) {
            const float2 x1 = make_float2(0.5f * X1.x, 0.5f * X1.y);
            const float2 x2 = make_float2(0.5f * X2c.x, 0.5f * X2c.y);
            stg2(spec + ja, x1);                                                       // X[ja]
            stg2(spec + ((N - ja) & (N - 1)), make_float2(x1.x, ja ? -x1.y : x1.y));   // X[N-ja]
            stg2(spec + (M - ja), make_float2(x2.x, -x2.y));                           // X[M-ja]
            stg2(spec + ((M + ja) & (N - 1)), ja ? x2 : make_float2(x2.x, -x2.y));     // X[M+ja]
        } else {
            const float m1 = cabs_fast(X1), m2 = cabs_fast(X2c);
            pair_synth<LOG2N>(m1, m1, m2, m2, w, ja, key, -0.25f / (float)N, VA, VB);
            lds[pA] = VA;
            if (pB != pA) lds[pB] = VB;
        }
    }
}

// Middle stage on the bit-reversed spectrum in LDS (position p holds bin brev_m(p)).
template <int LOG2N, int MODE>
__device__ __forceinline__ void middle_stage(float2 *lds, int tid, PhaseKey key,
                                             GV2 rtab, GV2W spec) {
    using G = Geo<LOG2N>;
    constexpr int m = G::m, M = G::M;
    opaque(tid);  // slot addresses / twiddles are recomputed per hop instead of living in VGPRs
#pragma unroll
    for (int s = 0; s < G::QN; ++s) {
        const int c = tid + G::T * s;
        if (c == 0) continue;  // slot 0 is the special block below
        const int j = (int)(__brev((unsigned)(2 * c)) >> (32 - (m - 1)));  // bin in (0, M/4)
        const int j2 = M / 2 - j;
        const int p1 = 4 * c;                                              // brev_m(j)
        const int p2 = (int)(__brev((unsigned)j2) >> (32 - m));            // brev_m(M/2 - j)
        const float2 w = ldg2(rtab + j);
        // pair (j, M-j): positions p1, p2+1 ; pair (M/2-j, M/2+j): positions p2, p1+1
        do_pair<LOG2N, MODE>(lds, pad_idx(p1), pad_idx(p2 + 1), w, (uint32_t)j, key, spec);
        do_pair<LOG2N, MODE>(lds, pad_idx(p2), pad_idx(p1 + 1), make_float2(-w.y, -w.x),
                             (uint32_t)j2, key, spec);
    }
    if (tid == 0) {
        // bins 0 (+Nyquist) at position 0, M/2 at position 1, pair (M/4, 3M/4) at 2, 3
        do_pair<LOG2N, MODE>(lds, 0, 0, make_float2(1.f, 0.f), 0u, key, spec);
        do_pair<LOG2N, MODE>(lds, 1, 1, make_float2(0.f, -1.f), (uint32_t)(M / 2), key, spec);
        do_pair<LOG2N, MODE>(lds, 2, 3, ldg2(rtab + M / 4), (uint32_t)(M / 4), key, spec);
    }
}

// Fused-path middle stage, batched: all slot addresses, twiddle loads and LDS reads are issued up
// front (v[] is dead here, so there are registers to hold them), then the pairs are computed
// branch-free; thread 0's slot 0 is computed on a harmless stand-in and written to a spare LDS
// element, the three special pairs follow under one branch.
template <int LOG2N>
__device__ __forceinline__ void middle_fused(float2 *lds, int tid, PhaseKey key, GV2 rtab) {
    using G = Geo<LOG2N>;
    constexpr int m = G::m, M = G::M, QN = G::QN;
    constexpr uint32_t N = 2u * M;
    constexpr int DUMMY = G::LDS_FLOAT2 - 1;
    opaque(tid);
    int ia[QN], ib[QN], ic[QN], id[QN];
    uint32_t ja[QN];
    float2 w[QN], A1[QN], A2[QN], B1[QN], B2[QN];
#pragma unroll
    for (int s = 0; s < QN; ++s) {
        int c = tid + G::T * s;
        if (s == 0) c = c ? c : 1;  // thread 0 / slot 0: stand-in, results go to DUMMY
        const int j = (int)(__brev((unsigned)(2 * c)) >> (32 - (m - 1)));  // bin in (0, M/4)
        const int p1 = 4 * c;                                              // brev_m(j)
        const int p2 = (int)(__brev((unsigned)(M / 2 - j)) >> (32 - m));   // brev_m(M/2 - j)
        ja[s] = (uint32_t)j;
        w[s] = ldg2(rtab + j);
        ia[s] = pad_idx(p1);
        ib[s] = pad_idx(p1 + 1);
        ic[s] = pad_idx(p2);
        id[s] = pad_idx(p2 + 1);
    }
#pragma unroll
    for (int s = 0; s < QN; ++s) {
        A1[s] = lds[ia[s]];  // bin j
        A2[s] = lds[ib[s]];  // bin j + M/2
        B1[s] = lds[ic[s]];  // bin M/2 - j
        B2[s] = lds[id[s]];  // bin M - j
    }
    if (tid == 0) {  // redirect the stand-in's writes (uniform per wave except wave 0)
        ia[0] = ib[0] = ic[0] = id[0] = DUMMY;
    }
    // the two counters of a pair follow from one multiply: x(b) = b*mul + k0 serves bins b and
    // M + b, x(M-b) = (M*mul + 2 k0) - x(b) serves bins M - b and N - b
    const uint32_t cM = M * key.mul + 2u * key.k0;
    const float nkappa = -0.25f / (float)N;
#pragma unroll
    for (int s = 0; s < QN; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // h = 0: pair (j, M-j) = (A1, B2), twiddle w ; h = 1: pair (M/2-j, M/2+j) = (B1, A2),
            // twiddle -i conj(w)
            const float2 A = h ? B1[s] : A1[s];
            const float2 Bp = h ? A2[s] : B2[s];
            const float2 ww = h ? make_float2(-w[s].y, -w[s].x) : w[s];
            const uint32_t jb = h ? (uint32_t)(M / 2) - ja[s] : ja[s];
            float2 X1, X2c;
            pair_analyze(A, Bp, ww, X1, X2c);
            float m1 = cabs_fast(X1) * nkappa, m2 = cabs_fast(X2c) * nkappa;
            const uint32_t x1 = jb * key.mul + key.k0;
            float c1, s1, c2, s2, c3, s3, c4, s4;
            phase_ncs2_x(x1, c1, s1, c4, s4);       // bins jb and M + jb
            phase_ncs2_x(cM - x1, c3, s3, c2, s2);  // bins M - jb and N - jb
            const float px = m1 * (c1 + c2), py = m1 * (s1 - s2);  // Zs[jb]
            const float qx = m2 * (c3 + c4), qy = m2 * (s4 - s3);  // conj(Zs[M-jb])
            const float sx = px + qx, sy = py + qy;
            const float rx = px - qx, ry = py - qy;
            const float ux = rx * ww.x + ry * ww.y, uy = ry * ww.x - rx * ww.y;  // U = conj(w) R
            const float2 VA = make_float2(sx - uy, sy + ux);  // S + iU        -> bin jb
            const float2 VB = make_float2(sx + uy, ux - sy);  // conj(S - iU)  -> bin M - jb
            if (h == 0) {
                lds[ia[s]] = VA;
                lds[id[s]] = VB;
            } else {
                lds[ic[s]] = VA;
                lds[ib[s]] = VB;
            }
        }
    }
    if (tid == 0) {
        // bins 0 (+Nyquist) at position 0, M/2 at position 1, pair (M/4, 3M/4) at 2, 3
        do_pair<LOG2N, MODE_FUSED>(lds, 0, 0, make_float2(1.f, 0.f), 0u, key, (GV2W) nullptr);
        do_pair<LOG2N, MODE_FUSED>(lds, 1, 1, make_float2(0.f, -1.f), (uint32_t)(M / 2), key,
                                   (GV2W) nullptr);
        do_pair<LOG2N, MODE_FUSED>(lds, 2, 3, ldg2(rtab + M / 4), (uint32_t)(M / 4), key,
                                   (GV2W) nullptr);
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
    if (RC_ABLATE & 2) {
#pragma unroll
        for (int q = 0; q < G::P; ++q) v[q] = make_float2((float)(lane2 + q), (float)(k + q));
        return;
    }
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

// a per-hop opaque copy of a table pointer: keeps the compiler from hoisting 2P table loads
// out of the hop loop (they are L1/L2 hits; 64+ live VGPRs would halve occupancy)
__device__ __forceinline__ GF per_hop(const float *ptr) {
    GF g = (GF)ptr;
    asm volatile("" : "+s"(g));
    return g;
}

template <int LOG2N, int MODE, bool PITCH1>
__global__ __launch_bounds__(Geo<LOG2N>::T, Geo<LOG2N>::WPS) void hop_kernel(const HopParams p) {
    using G = Geo<LOG2N>;
    constexpr int P = G::P, T = G::T, M = G::M, N = G::N, H = M;
    constexpr int LL = last_lor<G>(G::m);
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    ThreadCtx<G> ctx;
    ctx.tid = threadIdx.x;
    fill_lds_bases<G, G::m>(ctx);
    const int tid = ctx.tid;
    const uint32_t run = blockIdx.x % p.runs_per_channel;
    const uint32_t ch = blockIdx.x / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    GV2 wtab = (GV2)p.wtab;
    GV2 rtab = (GV2)p.rtab;

    float2 v[P];
    Stamps st;
    st.init();
    if constexpr (MODE == MODE_FORWARD) {
        for (int64_t k = k_begin; k < k_end; ++k) {
            GV2W spec = (GV2W)p.spec + ((size_t)ch * p.hop_count + (size_t)(k - p.hop_first)) * N;
            load_hop<LOG2N>(v, p, xc, xt, per_hop(p.window), k, lane2);
            forward_passes<G, G::m, 0, true>(v, lds, ctx, wtab, st);
            lds_store<G, LL>(v, lds, ctx.lb[LL]);
            if (!(RC_ABLATE & 4)) __syncthreads();
            middle_stage<LOG2N, MODE_FORWARD>(lds, tid, PhaseKey{0u, 1u}, rtab, spec);
            if (!(RC_ABLATE & 4)) __syncthreads();
        }
    } else if constexpr (MODE == MODE_RESYNTH) {
        for (int64_t k = k_begin; k < k_end; ++k) {
            const size_t hop_idx = (size_t)ch * p.hop_count + (size_t)(k - p.hop_first);
            GV2W spec = (GV2W)p.spec + hop_idx * N;
            const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
            middle_stage<LOG2N, MODE_RESYNTH>(lds, tid, key, rtab, spec);
            if (!(RC_ABLATE & 4)) __syncthreads();
            lds_load<G, LL>(v, lds, ctx.lb[LL]);
            if (!(RC_ABLATE & 4)) __syncthreads();
            inverse_passes<G, G::m>(v, lds, ctx, wtab, st);
            GFW y = (GFW)p.ybuf + hop_idx * N;
            GF wsrc = per_hop(p.window);
            constexpr int CH = P < RC_LOADCH ? P : RC_LOADCH;
#pragma unroll
            for (int q0 = 0; q0 < P; q0 += CH) {
#pragma unroll
                for (int q = q0; q < q0 + CH; ++q)
                    stg2((GV2W)(y + 2 * T * q + lane2),
                         make_float2(v[q].x * (wsrc + 2 * T * q)[lane2],
                                     v[q].y * (wsrc + 2 * T * q)[lane2 + 1]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        // fused path: overlap-add in registers. head slot q' <-> tail slot q' + P/2.
        constexpr int PH = P / 2;
        float2 tail[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = make_float2(0.f, 0.f);
        GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
        const uint32_t pitch = PITCH1 ? 1u : p.pitch;
        // hop k_begin - 1 is recomputed only for its tail (global hop 0 has a zero predecessor:
        // src/stretcher.rs:58-59)
        for (int64_t k = (k_begin > 0 ? k_begin - 1 : k_begin); k < k_end; ++k) {
            const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
            load_hop<LOG2N>(v, p, xc, xt, per_hop(p.window), k, lane2);
            st.mark(0);
            forward_passes<G, G::m, 0, true>(v, lds, ctx, wtab, st);
            lds_store<G, LL>(v, lds, ctx.lb[LL]);
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(12);
            if (!(RC_ABLATE & 16)) middle_fused<LOG2N>(lds, tid, key, rtab);
            st.mark(13);
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(14);
            lds_load<G, LL>(v, lds, ctx.lb[LL]);
            if (!(RC_ABLATE & 4)) __syncthreads();
            st.mark(15);
            inverse_passes<G, G::m>(v, lds, ctx, wtab, st);
            {
                // window loads for the synthesis multiply: all issued, one wait
                GF wsrc = per_hop(p.window);
                float wr0[P], wr1[P];
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    wr0[q] = (wsrc + 2 * T * q)[lane2];
                    wr1[q] = (wsrc + 2 * T * q)[lane2 + 1];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < P; ++q) v[q] = make_float2(v[q].x * wr0[q], v[q].y * wr1[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
            st.mark(27);
            if ((RC_ABLATE & 2) ? (v[0].x == 1.2345f) : (k >= k_begin)) {
                const int64_t g0 = k * (int64_t)H;  // absolute O index of this hop's first sample
                GF esrc = per_hop(p.env);
                if constexpr (PITCH1) {
                    GFW dst = outc + (g0 - p.out_origin);
                    float er0[PH], er1[PH];
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        er0[q] = (esrc + 2 * T * q)[lane2];
                        er1[q] = (esrc + 2 * T * q)[lane2 + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        float2 o;  // stretcher.rs:97-100 operation order
                        o.x = (v[q].x + tail[q].x) * er0[q] * p.amp;
                        o.y = (v[q].y + tail[q].y) * er1[q] * p.amp;
                        stg2((GV2W)(dst + 2 * T * q + lane2), o);
                    }
                } else {
                    // F[t] = O[t p]: keep element g = g0 + i iff g % p == 0, at F[g / p]
                    const int64_t kq = g0 / pitch;
                    const uint32_t kr = (uint32_t)(g0 % pitch);
                    GFW dst = outc + (kq - p.out_origin);
                    int t2 = tid;
                    opaque(t2);
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        const uint32_t i0 = 2u * (uint32_t)(t2 + T * q);
                        const float o0 = (v[q].x + tail[q].x) * (esrc + 2 * T * q)[lane2] * p.amp;
                        const float o1 = (v[q].y + tail[q].y) * (esrc + 2 * T * q)[lane2 + 1] * p.amp;
                        const uint32_t a0 = kr + i0, a1 = a0 + 1;
                        const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                        if (d0 * pitch == a0) dst[d0] = o0;
                        if (d1 * pitch == a1) dst[d1] = o1;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < PH; ++q) tail[q] = v[q + PH];
            st.mark(28);
        }
#if RC_STAMP
        if ((tid & 63) == 0 && p.spec) {
            unsigned *dbg = (unsigned *)p.spec + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 32;
            for (int i = 0; i < 32; ++i) dbg[i] = st.acc[i];
        }
#endif
    }
}


// uniform pointer to sample k*step of the hop's input, forced into SGPRs. Hops whose window runs
// past the end of the closed input read the engine's zero-padded tail copy (stretcher.rs:129-132).
__device__ __forceinline__ GF hop_src(const HopParams &p, GF xc, GF xt, int64_t k) {
    const int64_t off = (k >= p.tail_hop_first) ? (k * (int64_t)p.step - p.tail_origin)
                                                : (k * (int64_t)p.step - p.in_origin);
    const unsigned long long sa =
        (unsigned long long)((k >= p.tail_hop_first) ? xt : xc) + (unsigned long long)off * 4ull;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sa);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
    return (GF)(((unsigned long long)hi << 32) | lo);
}

// =============================== v2 fused kernel (N = 16384) =================================
// Same math as hop_kernel<.., MODE_FUSED>, restructured so that (tests/dev/proto_v2.py is the index
// model):
//   * BOTH transforms are DIT (3 packed FMAs per butterfly). The forward transform's bit-reversed
//     input order costs nothing: it is the order in which the thread's registers are loaded.
//   * the last forward pass leaves thread t with the natural-order bins of residues r = t and
//     512 - t (mod 512), i.e. every (j, M - j) pair sits in one thread: the real split, |X|, the
//     four phasors and the Hermitian fold run in registers, and the first inverse pass (position
//     bits 0..3 = frequency bits 9..12) follows without touching LDS. Residues 0 and 256 pair
//     with themselves; thread 0 owns them and hands its 17 pairs to lanes 0..16 of wave 0 through
//     a 32-element LDS scratch.
//   * 4 LDS exchanges per hop instead of 6 + the middle-stage round trip, 6 workgroup barriers
//     instead of 13: the stores of exchanges 2 and 4 are in place (same layout and index map as the
//     preceding load), so they need no write-after-read barrier. The index map
//     f3(n) = n + (n >> 5) + (n >> 8) keeps every access pattern at most 2-way conflicted on a few
//     lanes; it is additive over disjoint bit fields, so each access is a per-thread base VGPR +
//     an immediate offset.
#ifndef RC_V2
#define RC_V2 1
#endif
constexpr int f1_idx(int n) { return n + (n >> 5); }
constexpr int HOP2_XBUF = 8192 + 256 + 32;
constexpr int HOP2_LDS_FLOAT2 = HOP2_XBUF + 8 + 32 + 256 + 256 + 16 + 24 + 1024;  // 79 232 B
// LDS index map of the exchange buffer: a weight per position bit, so the map is additive over
// disjoint bit fields (per-thread base VGPR + immediate offset per register). The weights were searched
// (banking model of MI355X_MICROARCH.md: ds_write_b64 = 16-lane groups on 32 banks, ds_read_b64 =
// 32-lane groups on 64 banks) so that the stores of all four exchanges are conflict-free (the old map
// n + (n >> 5) + (n >> 8) was built for the read groups only and 2-way conflicted on the stores of
// exchanges 1 and 3: SQ_LDS_DATA_FIFO_FULL for half of the SQ cycles).
#ifndef RC_WMAP
#define RC_WMAP 1
#endif
constexpr int F3_W[13] = {1, 2, 4, 8, 16, 32, 64, 131, 259, 520, 1038, 2079, 4156};
constexpr int f3_idx(int n) {
    if (!RC_WMAP) return n + (n >> 5) + (n >> 8);
    int r = 0;
    for (int i = 0; i < 13; ++i) r += ((n >> i) & 1) * F3_W[i];
    return r;
}
constexpr int brev_c(int x, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}

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
template <int NREG, int M_LOG, int S_LO, int S_HI, int REG_LO, bool CONJ, bool HAS_L>
__device__ __forceinline__ void dit_stages(v2f (&v)[NREG], v2f wfine = v2f{1.0f, 0.0f}) {
    if (RC_ABLATE & 8) return;
    const v2f sgn = CONJ ? v2f{1.0f, -1.0f} : v2f{-1.0f, 1.0f};
    v2f bases[S_HI - S_LO + 1];
    if (HAS_L) {
        bases[S_HI - S_LO] = wfine;
#pragma unroll
        for (int s = S_HI - 1; s >= S_LO; --s) bases[s - S_LO] = vcmul(bases[s + 1 - S_LO], bases[s + 1 - S_LO]);
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
                } else if (kidx == 8) {  // w b = -i b (forward) / +i b (inverse) = -(b.yx * sgn)
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
constexpr double HANN_KAPPA = -0.25 / 16384.0;
__device__ constexpr HannK HANN_W14K = make_hann_k(0.5 * HANN_KAPPA, 16384, 32);

template <bool PITCH1, bool HANN>
__global__ __launch_bounds__(256, 2) void hop2_kernel(const HopParams p) {
    constexpr int LOG2N = 14, m = 13, M = 1 << m, H = M, T = 256, P = 32, PH = 16;
    constexpr int RES = 512;                      // residues of the last forward pass
    constexpr int SCR = HOP2_XBUF + 8;            // 32-element scratch for thread 0's pairs
    // per-workgroup twiddle / window-rotation tables (filled once per run): the hop loop itself has
    // no table loads from global memory
    constexpr int T_A = SCR + 32;                 // [256] W_8192^t
    constexpr int T_R = T_A + 256;                // [256] W_16384^t
    constexpr int T_B = T_R + 256;                // [16]  W_512^l
    constexpr int T_C = T_B + 16;                 // [24]  W_64^k, k <= 16
    constexpr int T_H = T_C + 24;                 // [1024] HANN: hann_rot as float2 pairs
    static_assert(T_H + 1024 == HOP2_LDS_FLOAT2, "LDS layout");
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
    GV2 wtab = (GV2)p.wtab;
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;

    // residues of this thread and per-thread LDS bases (thread part of every access pattern)
    const int r = tid, rb = tid ? RES - tid : RES / 2;
    const int l4 = tid & 15, uu = tid >> 4;
    const int pos4 = (uu << 9) | l4;                                  // LOR = 4 layout, q = 0
    const int bE1s = f3_idx(((int)(__brev((unsigned)tid) >> 24) << 5));  // brev8(t) << 5
    const int b4f3 = f3_idx(pos4);
    const int bAr = f3_idx(r), bBr = f3_idx(rb);
    const int bE3a = f3_idx(((int)(__brev((unsigned)r) >> 23) << 4));    // brev9(r) << 4
    const int bE3b = f3_idx(((int)(__brev((unsigned)rb) >> 23) << 4));
    const int bE4l = f3_idx(tid);

    Stamps st;
    st.init();
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};

    {
        lds[T_A + tid] = ldg2(wtab + tid);
        lds[T_R + tid] = ldg2((GV2)p.rtab + tid);
        if (tid < 16) lds[T_B + tid] = ldg2(wtab + 16 * tid);
        if (tid <= 16) lds[T_C + tid] = ldg2(wtab + 128 * tid);
        if constexpr (HANN) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // (cos, sin) of e = 0, 1 -> (cos e0, cos e1), (sin e0, sin e1)
                const float2 a = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid);
                const float2 b = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid + 1);
                lds[T_H + 512 * i + 2 * tid] = make_float2(a.x, b.x);
                lds[T_H + 512 * i + 2 * tid + 1] = make_float2(a.y, b.y);
            }
        }
        __syncthreads();
    }
    // The hop loop is software-pipelined (RC_SWP): the LDS stores of an exchange drain for ~800
    // cycles during which the wave would only wait at the barrier, so the next hop's window multiply
    // and first pass F1 (registers only) run between the E3 store and its barrier; the next hop's
    // samples are requested before I1. vn carries F1's output into the next iteration.
    //   RC_SWP = 0: plain order
    constexpr bool SWP = RC_SWP != 0 && HANN;  // (the table-window variant has no registers to spare)
    float xr0[P], xr1[P];
    auto issue_x = [&](int64_t kk) {
        GF src = hop_src(p, xc, xt, kk);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            xr0[q] = (src + 2 * T * q)[lane2];
            xr1[q] = (src + 2 * T * q)[lane2 + 1];
        }
    };
    // register q of vn := z[brev5(q) * T + t] * window, then F1 (bits 0..4, constants only)
    auto win_f1 = [&](v2f (&vn)[P]) {
        if constexpr (HANN) {
            const v2f cb = to_v(lds[T_H + 2 * tid]), sb = to_v(lds[T_H + 2 * tid + 1]);
            const v2f half = {0.5f, 0.5f};
#pragma unroll
            for (int q = 0; q < P; ++q) {  // packed: 3 instructions per sample pair
                const v2f wq = __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cb, half));
                vn[brev_c(q, 5)] = v2f{xr0[q], xr1[q]} * wq;
            }
        } else {
            GF win = per_hop(p.window);
            float wr0[P], wr1[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                wr0[q] = (win + 2 * T * q)[lane2];
                wr1[q] = (win + 2 * T * q)[lane2 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < P; ++q) vn[brev_c(q, 5)] = v2f{xr0[q], xr1[q]} * v2f{wr0[q], wr1[q]};
        }
        __builtin_amdgcn_sched_barrier(0);
        st.mark(0);
        dit_stages<32, m, 0, 4, 0, false, false>(vn);
        st.mark(1);
    };
    // epilogue of hop kk (ve = its I3 output): synthesis window, overlap-add with the carried tail,
    // store. With RC_SWP >= 2 it runs one iteration late, under the next hop's E1 store drain.
    auto epilogue = [&](int64_t kk, v2f (&ve)[P]) {
        // ---- epilogue: synthesis window, overlap-add with the carried tail, store
        // HANN: (cos, cos) / (sin, sin) of this thread's beta for samples e = 0, 1 (window, envelope)
        v2f cbW = {0.f, 0.f}, sbW = cbW, cbE = cbW, sbE = cbW;
        const v2f half2 = {0.5f, 0.5f};
        if constexpr (HANN) {
            cbW = to_v(lds[T_H + 2 * tid]), sbW = to_v(lds[T_H + 2 * tid + 1]);
            cbE = to_v(lds[T_H + 2 * T + 2 * tid]), sbE = to_v(lds[T_H + 2 * T + 2 * tid + 1]);
#pragma unroll
            for (int q = 0; q < P; ++q)
                ve[q] *= __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sbW,
                        __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cbW, half2));
            __builtin_amdgcn_sched_barrier(0);
        } else {
            GF wsrc = per_hop(p.window);
            float wr0[P], wr1[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                wr0[q] = (wsrc + 2 * T * q)[lane2];
                wr1[q] = (wsrc + 2 * T * q)[lane2 + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < P; ++q) ve[q] *= v2f{wr0[q], wr1[q]};
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kk >= k_begin) {
            const int64_t g0 = kk * (int64_t)H;
            GF esrc = per_hop(p.env);
            if constexpr (PITCH1) {
                GFW dst = outc + (g0 - p.out_origin);
                float er0[PH], er1[PH];
                if constexpr (!HANN) {
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        er0[q] = (esrc + 2 * T * q)[lane2];
                        er1[q] = (esrc + 2 * T * q)[lane2 + 1];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const v2f amp2 = {p.amp, p.amp};
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    v2f er;
                    if constexpr (HANN)
                        er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                             __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, half2));
                    else
                        er = v2f{er0[q], er1[q]};
                    // stretcher.rs:97-100 operation order, both samples of the pair per instruction
                    const v2f o = (ve[q] + tail[q]) * er * amp2;
                    *(GV2W)(dst + 2 * T * q + lane2) = o;
                }
            } else {
                const int64_t kq = g0 / pitch;
                const uint32_t kr = (uint32_t)(g0 % pitch);
                GFW dst = outc + (kq - p.out_origin);
                int t2 = tid;
                opaque(t2);
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    const uint32_t i0 = 2u * (uint32_t)(t2 + T * q);
                    v2f er;
                    if constexpr (HANN)
                        er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                             __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, half2));
                    else
                        er = v2f{(esrc + 2 * T * q)[lane2], (esrc + 2 * T * q)[lane2 + 1]};
                    const v2f o = (ve[q] + tail[q]) * er * v2f{p.amp, p.amp};
                    const float o0 = o.x, o1 = o.y;
                    const uint32_t a0 = kr + i0, a1 = a0 + 1;
                    const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                    if (d0 * pitch == a0) dst[d0] = o0;
                    if (d1 * pitch == a1) dst[d1] = o1;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = ve[q + PH];
        st.mark(21);
    };
    const int64_t k_first = k_begin > 0 ? k_begin - 1 : k_begin;
    constexpr bool SWP2 = SWP && RC_SWP >= 2;
    v2f vo[P];  // SWP2: I3 output of the previous hop, its epilogue still to run
    v2f vn[P];
    if constexpr (SWP) issue_x(k_first);
    if constexpr (SWP) win_f1(vn);
    for (int64_t k = k_first; k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        if constexpr (!SWP) {
            issue_x(k);
            win_f1(vn);
        }
        // ---- forward: F1 (done), E1, F2 (bits 5..8), E2, F3 (bits 9..12)
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 128))) lds[bE1s + f3_idx(q)] = to_f2(vn[q]);
        if (RC_ABLATE & 4) {
#pragma unroll
            for (int q = 0; q < P; ++q) v[q] = vn[q];
        }
        if constexpr (SWP2) {
            if (k > k_first) epilogue(k - 1, vo);
        }
        st.mark(2);
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(3);
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 256))) v[q] = xld(lds, b4f3 + f3_idx(q << 4));
        dit_stages<32, m, 5, 8, 4, false, true>(v, to_v(lds[T_B + l4]));
        st.mark(4);
        // E2 store is IN PLACE (same layout, same index map as the E1 load): each thread overwrites
        // exactly the elements it read, so no barrier is needed between the two
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 128))) lds[b4f3 + f3_idx(q << 4)] = to_f2(v[q]);
        st.mark(5);
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(6);
        v2f va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!(RC_ABLATE & (4 | 256))) {
                va[q] = xld(lds, bAr + f3_idx((RES * q)));
                vb[q] = xld(lds, bBr + f3_idx((RES * q)));
            } else {
                va[q] = v[q];
                vb[q] = v[q + 16];
            }
        }
        st.mark(7);
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(8);
        {
            const v2f wa = to_v(lds[T_A + tid]);  // W_8192^r
            // W_8192^rb: rb = 512 - r -> W_16 conj(W_8192^r); thread 0: rb = 256 -> W_32
            const v2f k16 = {W32_RE[2], W32_IM[2]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);
            if (tid == 0) wb = v2f{W32_RE[1], W32_IM[1]};
            dit_stages<16, m, 9, 12, 9, false, true>(va, wa);
            dit_stages<16, m, 9, 12, 9, false, true>(vb, wb);
        }
        st.mark(9);

        // ---- middle stage in registers: pair (A[q], B[15-q]) = bins (r + 512 q, M - that)
        if (tid == 0) {  // thread 0 owns the self-paired residues 0 and 256: hand them to wave 0
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                lds[SCR + q] = to_f2(va[q]);
                lds[SCR + 16 + q] = to_f2(vb[q]);
            }
        }
        {
            const int rr = r;
            const float2 wr = lds[T_R + tid];                  // exp(-2 pi i r / N), r = tid
            const uint32_t x0 = (uint32_t)rr * key.mul + key.k0;  // counter of bin r
            const uint32_t dx = (uint32_t)RES * key.mul;          // + 512 bins
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (RC_ABLATE & 16) continue;
                // exp(-2 pi i (r + 512 q) / N) = wr * W32^q
#if RC_PAIR_PK
                const v2f wrv = to_v(wr);
                const v2f wq = q == 0 ? wrv : (q == 8 ? v2f{wr.y, -wr.x}
                               : vcmul(wrv, v2f{W32_RE[q & 15], W32_IM[q & 15]}));
                v2f VA, VB;
                pair_regs_pk<LOG2N>(va[q], vb[15 - q], wq, x0 + (uint32_t)q * dx, key, VA, VB);
                va[q] = VA;
                vb[15 - q] = VB;
#else
                const float2 wq = q == 0 ? wr : (q == 8 ? make_float2(wr.y, -wr.x)
                                  : cmul(wr, make_float2(W32_RE[q & 15], W32_IM[q & 15])));
                float2 VA, VB;
                pair_regs<LOG2N>(to_f2(va[q]), to_f2(vb[15 - q]), wq, x0 + (uint32_t)q * dx, key, VA, VB);
                va[q] = to_v(VA);
                vb[15 - q] = to_v(VB);
#endif
            }
        }
        st.mark(10);
        if (tid < 64 && !(RC_ABLATE & 64)) {  // wave 0: lanes 0..16 compute thread 0's 17 pairs from the scratch
            const int i = tid;
            if (i <= 16) {
                int ja, ia, ib;
                if (i == 0) { ja = 0; ia = 0; ib = 0; }
                else if (i <= 7) { ja = RES * i; ia = i; ib = 16 - i; }
                else if (i == 8) { ja = RES * 8; ia = 8; ib = 8; }
                else { ja = RES / 2 + RES * (i - 9); ia = 16 + (i - 9); ib = 16 + 15 - (i - 9); }
                const float2 A = lds[SCR + ia], Bp = lds[SCR + ib];
                const float2 w = lds[T_C + (ja >> 8)];  // exp(-2 pi i ja / N) = W_64^(ja/256)
                float2 VA, VB;
                pair_regs<LOG2N>(A, Bp, w, (uint32_t)ja * key.mul + key.k0, key, VA, VB, ja == 0);
                lds[SCR + ia] = VA;
                if (ib != ia) lds[SCR + ib] = VB;
            }
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    va[q] = to_v(lds[SCR + q]);
                    vb[q] = to_v(lds[SCR + 16 + q]);
                }
            }
        }
        st.mark(11);
        if constexpr (SWP) issue_x(k + 1 < k_end ? k + 1 : k);  // (the last hop re-reads itself)
        // ---- inverse: I1 in registers (position bits 0..3 = brev4 of the register index)
        v2f pa[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            pa[brev_c(q, 4)] = va[q];
            pb[brev_c(q, 4)] = vb[q];
        }
        dit_stages<16, m, 0, 3, 0, true, false>(pa);
        dit_stages<16, m, 0, 3, 0, true, false>(pb);
        st.mark(12);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!(RC_ABLATE & (4 | 128))) lds[bE3a + f3_idx(q)] = to_f2(pa[q]);
            if (!(RC_ABLATE & (4 | 128))) lds[bE3b + f3_idx(q)] = to_f2(pb[q]);
        }
        st.mark(13);
        if constexpr (SWP) win_f1(vn);  // next hop's window + F1 while the E3 stores drain
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(14);
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 256))) v[q] = xld(lds, b4f3 + f3_idx(q << 4));
        dit_stages<32, m, 4, 8, 4, true, true>(v, to_v(lds[T_B + l4]));
        st.mark(15);
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 128))) lds[b4f3 + f3_idx(q << 4)] = to_f2(v[q]);  // in place (see E2)
        st.mark(16);
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(17);
#pragma unroll
        for (int q = 0; q < P; ++q) if (!(RC_ABLATE & (4 | 256))) v[q] = xld(lds, bE4l + f3_idx((q << 8)));
        st.mark(18);
        if (!(RC_ABLATE & 32)) __syncthreads();
        st.mark(19);
        dit_stages<32, m, 9, 12, 8, true, true>(v, to_v(lds[T_A + tid]));
        st.mark(20);

        if constexpr (SWP2) {
#pragma unroll
            for (int q = 0; q < P; ++q) vo[q] = v[q];
        } else {
            epilogue(k, v);
        }
    }
    if constexpr (SWP2) epilogue(k_end - 1, vo);
#if RC_STAMP
    if ((tid & 63) == 0 && p.spec) {
        unsigned *dbg = (unsigned *)p.spec + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 32;
        for (int i = 0; i < 32; ++i) dbg[i] = st.acc[i];
    }
#endif
}

// =============== v3: hop2's math with two-round exchanges, three workgroups per CU ===============
// The exchange buffer of hop2_kernel (8192 complex = 66 KiB) limits a CU to two workgroups. Here
// every exchange runs in two rounds over a HALF buffer (4096 complex): round A moves the elements
// whose position has a chosen bit (4 for exchanges 1 and 3, 8 for 2 and 4) clear, round B the others;
// the first store of exchanges 2 and 4 is in place (the thread overwrites what it read last). 46 KiB
// of LDS and <= 168 VGPRs per workgroup: three workgroups = 3 waves per SIMD, at 14 barriers per hop
// instead of 6. Default (hanning) window only; no software pipelining (no registers for it).
#ifndef RC_V3
#define RC_V3 1
#endif
#ifndef RC_NTSTORE
#define RC_NTSTORE 1
#endif
constexpr int G12_W[12] = {1, 2, 4, 8, 16, 32, 66, 130, 263, 526, 1052, 2104};  // searched like F3_W
constexpr int g_idx(int n) {
    int r = 0;
    for (int i = 0; i < 12; ++i) r += ((n >> i) & 1) * G12_W[i];
    return r;
}
constexpr int HOP3_XBUF = 4208;
constexpr int HOP3_LDS_FLOAT2 = HOP3_XBUF + 8 + 32 + 256 + 256 + 16 + 24 + 1024;  // 46 592 B

template <bool PITCH1>
__global__ __launch_bounds__(256, 3) void hop3_kernel(const HopParams p) {
    constexpr int LOG2N = 14, m = 13, M = 1 << m, H = M, T = 256, P = 32, PH = 16;
    constexpr int RES = 512;
    constexpr int SCR = HOP3_XBUF + 8;
    constexpr int T_A = SCR + 32;                 // [256] W_8192^t
    constexpr int T_R = T_A + 256;                // [256] W_16384^t
    constexpr int T_B = T_R + 256;                // [16]  W_512^l
    constexpr int T_C = T_B + 16;                 // [24]  W_64^k, k <= 16
    constexpr int T_H = T_C + 24;                 // [1024] window / envelope rotations
    static_assert(T_H + 1024 == HOP3_LDS_FLOAT2, "LDS layout");
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    // global run index: the order in which workgroups START when the seam hand-over is on
    uint32_t gr = blockIdx.x;
    const bool seam = p.seam_head != nullptr;
    if (seam) {
        unsigned *slot = reinterpret_cast<unsigned *>(lds + SCR);
        if (tid == 0) *slot = atomicAdd(p.run_counter, 1u);
        __syncthreads();
        gr = *reinterpret_cast<volatile unsigned *>(slot);
        __syncthreads();
    }
    const uint32_t run = gr % p.runs_per_channel;
    const uint32_t ch = gr / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    const bool stash_first = seam && run > 0;                          // my first head goes to the stash
    const bool has_next = seam && run + 1 < p.runs_per_channel;        // I finish my successor's first head
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    GV2 wtab = (GV2)p.wtab;
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;

    // reduced positions n' (the split bit removed), thread part of every access pattern
    const int l4 = tid & 15, uu = tid >> 4;
    const int rbp = (256 - tid) & 255;                                  // residue rb - 256
    const int bE1w = g_idx((int)(__brev((unsigned)tid) >> 24) << 4);    // (brev8(t) << 4) | q
    const int b4 = g_idx((uu << 8) | l4);                               // (uu << 8) | (j << 4) | l4
    const int bA = g_idx(tid), bB = g_idx(rbp);                         // (q << 8) | residue
    const int bE3a = g_idx((int)(__brev((unsigned)tid) >> 24) << 4);    // (brev8(r) << 4) | q
    const int bE3b = g_idx((int)(__brev((unsigned)rbp) >> 24) << 4);
    const int bE4 = g_idx(tid);                                         // (j << 8) | t

    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    {
        lds[T_A + tid] = ldg2(wtab + tid);
        lds[T_R + tid] = ldg2((GV2)p.rtab + tid);
        if (tid < 16) lds[T_B + tid] = ldg2(wtab + 16 * tid);
        if (tid <= 16) lds[T_C + tid] = ldg2(wtab + 128 * tid);
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // (cos, sin) of e = 0, 1 -> (cos e0, cos e1), (sin e0, sin e1)
            const float2 a = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid);
            const float2 b = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid + 1);
            lds[T_H + 512 * i + 2 * tid] = make_float2(a.x, b.x);
            lds[T_H + 512 * i + 2 * tid + 1] = make_float2(a.y, b.y);
        }
        __syncthreads();
    }
    const v2f half2 = {0.5f, 0.5f};
    // O[kk H + i] = (head[i] + tail[i]) * env[i] * amp for this thread's 32 head samples, decimated by
    // the pitch multiple (src/stretcher.rs:96-112)
    auto store_head = [&](int64_t kk, const auto &head) {
        const v2f cbE = to_v(lds[T_H + 2 * T + 2 * tid]), sbE = to_v(lds[T_H + 2 * T + 2 * tid + 1]);
        const int64_t g0 = kk * (int64_t)H;
        if constexpr (PITCH1) {
            GFW dst = outc + (g0 - p.out_origin);
            const v2f amp2 = {p.amp, p.amp};
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, half2));
                // stretcher.rs:97-100 operation order, both samples of the pair per instruction
                const v2f o = (head[q] + tail[q]) * er * amp2;
#if RC_NTSTORE
                __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * q + lane2));  // written once, never re-read here
#else
                *(GV2W)(dst + 2 * T * q + lane2) = o;
#endif
            }
        } else {
            const int64_t kq = g0 / pitch;
            const uint32_t kr = (uint32_t)(g0 % pitch);
            GFW dst = outc + (kq - p.out_origin);
            int t2 = tid;
            opaque(t2);
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const uint32_t i0 = 2u * (uint32_t)(t2 + T * q);
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, half2));
                const v2f o = (head[q] + tail[q]) * er * v2f{p.amp, p.amp};
                const uint32_t a0 = kr + i0, a1 = a0 + 1;
                const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                if (d0 * pitch == a0) dst[d0] = o.x;
                if (d1 * pitch == a1) dst[d1] = o.y;
            }
        }
    };
    // hop k_begin - 1 is recomputed for its tail only where no other run hands the seam over
    for (int64_t k = ((k_begin > 0 && !stash_first) ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        {   // register brev5(q) := z[q * T + t] * window ; F1 = stages 0..4
            GF src = hop_src(p, xc, xt, k);
            float xr0[P], xr1[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                xr0[q] = (src + 2 * T * q)[lane2];
                xr1[q] = (src + 2 * T * q)[lane2 + 1];
            }
            const v2f cb = to_v(lds[T_H + 2 * tid]), sb = to_v(lds[T_H + 2 * tid + 1]);
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const v2f wq = __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cb, half2));
                v[brev_c(q, 5)] = v2f{xr0[q], xr1[q]} * wq;
            }
            dit_stages<32, m, 0, 4, 0, false, false>(v);
        }
        // ---- E1 (split on position bit 4): register q = position bits 0..4
        __syncthreads();  // the previous hop's last E4 reads are done
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[bE1w + g_idx(q)] = to_f2(v[q]);
        __syncthreads();
        v2f w2[P];  // register q' = position bits 4..8
#pragma unroll
        for (int j = 0; j < 16; ++j) w2[2 * j] = to_v(lds[b4 + g_idx(j << 4)]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[bE1w + g_idx(q)] = to_f2(v[16 + q]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) w2[2 * j + 1] = to_v(lds[b4 + g_idx(j << 4)]);
        dit_stages<32, m, 5, 8, 4, false, true>(w2, to_v(lds[T_B + l4]));
        // ---- E2 (split on position bit 8 = register bit 4); first store in place
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[b4 + g_idx(j << 4)] = to_f2(w2[j]);
        __syncthreads();
        v2f va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) va[q] = to_v(lds[bA + g_idx(q << 8)]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[b4 + g_idx(j << 4)] = to_f2(w2[16 + j]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) vb[q] = to_v(lds[bB + g_idx(q << 8)]);
        {
            const v2f wa = to_v(lds[T_A + tid]);  // W_8192^r
            const v2f k16 = {W32_RE[2], W32_IM[2]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_8192^(512 - r); thread 0: rb = 256 -> W_32
            if (tid == 0) wb = v2f{W32_RE[1], W32_IM[1]};
            dit_stages<16, m, 9, 12, 9, false, true>(va, wa);
            dit_stages<16, m, 9, 12, 9, false, true>(vb, wb);
        }
        // ---- middle stage in registers (as hop2_kernel)
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                lds[SCR + q] = to_f2(va[q]);
                lds[SCR + 16 + q] = to_f2(vb[q]);
            }
        }
        {
            const float2 wr = lds[T_R + tid];
            const uint32_t x0 = (uint32_t)tid * key.mul + key.k0;
            const uint32_t dx = (uint32_t)RES * key.mul;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2f wrv = to_v(wr);
                const v2f wq = q == 0 ? wrv : (q == 8 ? v2f{wr.y, -wr.x}
                               : vcmul(wrv, v2f{W32_RE[q & 15], W32_IM[q & 15]}));
                v2f VA, VB;
                pair_regs_pk<LOG2N>(va[q], vb[15 - q], wq, x0 + (uint32_t)q * dx, key, VA, VB);
                va[q] = VA;
                vb[15 - q] = VB;
            }
        }
        if (tid < 64) {  // wave 0: lanes 0..16 compute thread 0's 17 pairs from the scratch
            const int i = tid;
            if (i <= 16) {
                int ja, ia, ib;
                if (i == 0) { ja = 0; ia = 0; ib = 0; }
                else if (i <= 7) { ja = RES * i; ia = i; ib = 16 - i; }
                else if (i == 8) { ja = RES * 8; ia = 8; ib = 8; }
                else { ja = RES / 2 + RES * (i - 9); ia = 16 + (i - 9); ib = 16 + 15 - (i - 9); }
                const float2 A = lds[SCR + ia], Bp = lds[SCR + ib];
                const float2 w = lds[T_C + (ja >> 8)];
                float2 VA, VB;
                pair_regs<LOG2N>(A, Bp, w, (uint32_t)ja * key.mul + key.k0, key, VA, VB, ja == 0);
                lds[SCR + ia] = VA;
                if (ib != ia) lds[SCR + ib] = VB;
            }
            if (tid == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    va[q] = to_v(lds[SCR + q]);
                    vb[q] = to_v(lds[SCR + 16 + q]);
                }
            }
        }
        // ---- inverse: I1 in registers
        v2f pa[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            pa[brev_c(q, 4)] = va[q];
            pb[brev_c(q, 4)] = vb[q];
        }
        dit_stages<16, m, 0, 3, 0, true, false>(pa);
        dit_stages<16, m, 0, 3, 0, true, false>(pb);
        // ---- E3 (split on inverse position bit 4: residue r < 256 -> round A, rb >= 256 -> round B)
        __syncthreads();  // every thread has read its vb
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[bE3a + g_idx(q)] = to_f2(pa[q]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[2 * j] = to_v(lds[b4 + g_idx(j << 4)]);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[bE3b + g_idx(q)] = to_f2(pb[q]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) v[2 * j + 1] = to_v(lds[b4 + g_idx(j << 4)]);
        dit_stages<32, m, 4, 8, 4, true, true>(v, to_v(lds[T_B + l4]));
        // ---- E4 (split on position bit 8); first store in place
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[b4 + g_idx(j << 4)] = to_f2(v[j]);
        __syncthreads();
        v2f y[P];  // register q = position bits 8..12
#pragma unroll
        for (int j = 0; j < 16; ++j) y[2 * j] = to_v(lds[bE4 + g_idx(j << 8)]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) lds[b4 + g_idx(j << 4)] = to_f2(v[16 + j]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) y[2 * j + 1] = to_v(lds[bE4 + g_idx(j << 8)]);
        dit_stages<32, m, 9, 12, 8, true, true>(y, to_v(lds[T_A + tid]));

        // ---- epilogue: synthesis window, overlap-add with the carried tail, store
        const v2f cbW = to_v(lds[T_H + 2 * tid]), sbW = to_v(lds[T_H + 2 * tid + 1]);
#pragma unroll
        for (int q = 0; q < P; ++q)
            y[q] *= __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sbW,
                    __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cbW, half2));
        if (k >= k_begin) {
            if (stash_first && k == k_begin) {
                // the run before this one holds the tail that belongs to this head: stash the windowed
                // head for it and publish (release at agent scope: the reader may sit on another XCD)
                // agent-scope (write-through) stores and loads for the stash and its flag instead of
                // release / acquire fences: a fence writes back or invalidates the whole XCD L2, and
                // 6 000 of them per launch cost 9 %
                int t2 = tid;
                opaque(t2);  // keep the 16 store addresses out of the hop loop's live registers
                unsigned long long *hs = (unsigned long long *)(p.seam_head + (size_t)gr * H) + t2;
#pragma unroll
                for (int q = 0; q < PH; ++q)
                    __hip_atomic_store(hs + T * q, __builtin_bit_cast(unsigned long long, y[q]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // every storing wave drains its own write-through stores before the barrier; only then
                // may lane 0 publish (a workgroup-scope fence emits no vmcnt wait on gfx950, and inline
                // asm is the form the compiler cannot drop: MI355X_MICROARCH.md, valid hand-off forms)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && !(p.diag_flags & RC_DIAG_SKIP_SEAM_PUBLISH))
                    __hip_atomic_store(p.seam_flag + gr, p.seam_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                store_head(k, y);
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = y[q + PH];
    }
    if (has_next) {
        // hop k_end is the first hop of run gr + 1: its workgroup started after this one and stashed
        // the head one hop after its start. Bounded wait (never reached unless the launch is broken):
        // on expiry the seam samples stay unwritten and the host is told through *err_word.
        unsigned *okw = reinterpret_cast<unsigned *>(lds + SCR);
        if (tid == 0) {
            unsigned ok = 0;
            for (unsigned spin = 0; spin < p.seam_spin_limit; ++spin) {
                if (__hip_atomic_load(p.seam_flag + gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                    p.seam_epoch) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            if (!ok && p.err_word)
                __hip_atomic_store(p.err_word, RC_ERR_SEAM_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *okw = ok;
        }
        __syncthreads();
        if (*reinterpret_cast<volatile unsigned *>(okw) == 0) return;
        const unsigned long long *hs = (const unsigned long long *)(p.seam_head + (size_t)(gr + 1) * H) + tid;
        v2f head[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q)
            head[q] = __builtin_bit_cast(v2f, __hip_atomic_load(hs + T * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        store_head(k_end, head);
    }
}

// =============== v4: hop3's arithmetic, two of the four exchanges wave-local ====================
// hop3 synchronises the whole workgroup around every half-exchange (14 s_barrier per hop), and its own
// ablations price that lockstep - all four waves reach the LDS store path together - above the stores
// themselves. hop4 keeps every register layout and every floating-point operation of hop3 (the output
// is bit-identical) and changes only WHICH THREAD holds a layout's elements (tests/dev/proto_v4.py is
// the index model; it checks every hand-over and the bank conflicts of every wave instruction):
//   * the 16 low nibbles of a residue fall into four classes closed under negation mod 16,
//     phi(l) = l0 ? 1 + 2 (l1 ^ l2) : 2 l1. A wave owns one class in the F2 / F3 / I1 / I2 layouts, so a
//     residue r and its partner 512 - r (the two bins of every (j, M - j) pair) sit in one wave, and
//     the exchanges E2 (F2 -> F3) and E3 (I1 -> I2) never leave the wave: no s_barrier at all - the LDS
//     executes one wave's instructions in order, only the compiler needs a fence;
//   * E1 (F1 -> F2) and E4 (I2 -> I3) still cross waves (the global load / store order wants thread =
//     low sample bits). Each runs in two rounds over the four per-wave regions of the half-size buffer:
//     round A writes the OWN region and reads all four, round B writes all four and reads the OWN one, so
//     a region is only ever overwritten by the wave that read it last and the wave-local exchanges in
//     between need no workgroup barrier either: 3 barriers per cross exchange, 6 per hop.
// Register strides 64 (+ a lane ^ 16 swizzle on two patterns) and 65 make 15 of the 16 access patterns
// conflict-free and the last one 2-way on half a wave.
#ifndef RC_V4
#define RC_V4 1
#endif
constexpr int HOP4_REG = 1040;                    // float2 slots per wave region (16 x 65)
constexpr int HOP4_XBUF = 4 * HOP4_REG;
constexpr int HOP4_LDS_FLOAT2 = HOP4_XBUF + 8 + 32 + 256 + 256 + 16 + 24 + 1024;  // 46 048 B
constexpr int phi_c(int l) { return (l & 1) ? 1 + 2 * (((l >> 1) ^ (l >> 2)) & 1) : 2 * ((l >> 1) & 1); }
constexpr int cidx_c(int l) { return (((l >> 2) & 1) << 1) | ((l >> 3) & 1); }
// compiler-only ordering of one wave's LDS accesses (no instruction is emitted)
__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier of the hop loop. __syncthreads() also waits for vmcnt(0), i.e. for the previous hop's
// output stores to be acknowledged; the exchanges only need this wave's LDS operations to have completed.
#ifndef RC_LGKM_BARRIER
#define RC_LGKM_BARRIER 1
#endif
// number of output pairs of a hop (of 16) whose store is deferred into the next hop's first pass; 0 = none
#ifndef RC_DEFER_STORE
#define RC_DEFER_STORE 0
#endif
#define HOP4_PAIR pair_regs_pk4
#ifndef RC_XCD_RUNS
#define RC_XCD_RUNS 1
#endif
#define HOP4_BAR()                                                                    \
    do {                                                                              \
        if (RC_ABLATE & 32) break;                                                    \
        if (RC_LGKM_BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else __syncthreads();                                                         \
    } while (0)
// RC_DK_BAND fused into the pair stage: |gain| of real-spectrum bin f <= N/2 (lo <= f <= hi: inside)
__device__ __forceinline__ float band_gain(const HopParams &p, uint32_t f) {
    return (f - p.band_lo) <= p.band_span ? p.band_gin : p.band_gout;
}
template <bool PITCH1, bool BAND = false>
__global__ __launch_bounds__(256, 3) void hop4_kernel(const HopParams p) {
    constexpr int LOG2N = 14, m = 13, M = 1 << m, H = M, T = 256, P = 32, PH = 16;
    constexpr int RES = 512, REG = HOP4_REG;
    constexpr int SCR = HOP4_XBUF + 8;
    constexpr int T_A = SCR + 32;                 // [256] W_8192^r
    constexpr int T_R = T_A + 256;                // [256] W_16384^r
    constexpr int T_B = T_R + 256;                // [16]  W_512^l
    constexpr int T_C = T_B + 16;                 // [24]  W_64^k, k <= 16
    constexpr int T_H = T_C + 24;                 // [1024] window / envelope rotations
    static_assert(T_H + 1024 == HOP4_LDS_FLOAT2, "LDS layout");
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    uint32_t gr = blockIdx.x;
    const bool seam = p.seam_head != nullptr;
    if (seam) {
        // Runs are handed out in the order workgroups start, per XCD: XCD x walks the x-th eighth of the runs
        // (neighbouring runs read overlapping input and meet in one L2; a run waits at its end for the head its
        // successor stashed at its start: the successor is the next ticket of the same XCD, or the first run of the
        // next eighth, which started with the launch). An XCD that runs out takes from the next one's counter.
        unsigned *slot = reinterpret_cast<unsigned *>(lds + SCR);
        if (tid == 0) {
            const uint32_t total = p.runs_per_channel * p.n_channels;
            unsigned got = 0xFFFFFFFFu;
            if (RC_XCD_RUNS) {
                const uint32_t G = (total + 7u) / 8u;
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                for (uint32_t i = 0; i < 8u; ++i) {
                    const uint32_t xx = (xcc + i) & 7u, lo = xx * G;
                    if (lo >= total) continue;
                    const uint32_t hi = lo + G < total ? lo + G : total;
                    const uint32_t t = atomicAdd(p.run_counter + xx, 1u);
                    if (t < hi - lo) {
                        got = lo + t;
                        break;
                    }
                }
            } else {
                got = atomicAdd(p.run_counter, 1u);
            }
            *slot = got;
        }
        __syncthreads();
        gr = *reinterpret_cast<volatile unsigned *>(slot);
        __syncthreads();
        if (gr == 0xFFFFFFFFu) return;  // (more workgroups than runs: cannot happen with the engine's grid)
    }
    const uint32_t run = gr % p.runs_per_channel;
    const uint32_t ch = gr / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    const bool stash_first = seam && run > 0;
    const bool has_next = seam && run + 1 < p.runs_per_channel;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    GV2 wtab = (GV2)p.wtab;
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;

    // ---- who am I in each layout. Everything below is a few integer operations on tid; it is recomputed
    // from an opaque copy right before each exchange instead of living in ~20 VGPRs across the whole hop
    // (at 168 VGPRs the allocator spilled them, and a scratch reload waits in line behind every older
    // vector-memory operation of the wave).
    struct Who {
        int wv, lane, cc, lo4, nib, r, A0;
    };
    // two of them ARE kept (one VGPR each): the class nibble (a dozen operations) and the own-region base
    int nib_keep, a0_keep;
    {
        const int w_ = tid >> 6, c_ = (tid >> 4) & 3;
        const int nl3 = c_ & 1, nl2 = c_ >> 1, nl0 = w_ & 1;
        const int nl1 = nl0 ? ((w_ >> 1) ^ nl2) : (w_ >> 1);
        nib_keep = (nl3 << 3) | (nl2 << 2) | (nl1 << 1) | nl0;  // member(wv, cc), cc = (l2 << 1 | l3)
        a0_keep = w_ * REG + (tid & 63);
    }
    auto who = [&]() {
        int t = tid;
        opaque(t);
        Who w;
        w.wv = t >> 6;
        w.lane = t & 63;
        w.cc = (t >> 4) & 3;
        w.lo4 = t & 15;
        w.nib = nib_keep;
        // F2: l4 = nib, uu = lo4.  F3 / I1: residue r = lo4 << 4 | nib (and 512 - r).  I2: l4' = lo4, uu' = brev4(nib)
        w.r = (w.lo4 << 4) | w.nib;
        w.A0 = a0_keep;
        return w;
    };
    const int wv = tid >> 6;  // (wave-uniform branches only)
    Stamps st;
    st.init();
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    {
        lds[T_A + tid] = ldg2(wtab + tid);
        lds[T_R + tid] = ldg2((GV2)p.rtab + tid);
        if (tid < 16) lds[T_B + tid] = ldg2(wtab + 16 * tid);
        if (tid <= 16) lds[T_C + tid] = ldg2(wtab + 128 * tid);
        if (tid == 0) {  // W_N^(256 - 4096) = i W_N^256: thread 0's twiddle base for its residue-256 slots
            const float2 w256 = ldg2((GV2)p.rtab + 256);
            lds[SCR + 2] = make_float2(-w256.y, w256.x);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float2 a = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid);
            const float2 b = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid + 1);
            lds[T_H + 512 * i + 2 * tid] = make_float2(a.x, b.x);
            lds[T_H + 512 * i + 2 * tid + 1] = make_float2(a.y, b.y);
        }
        __syncthreads();
    }
    const v2f half2 = {0.5f, 0.5f};
    auto store_head = [&](int64_t kk, const auto &head) {
        const v2f amp2 = {p.amp, p.amp};
        // env[i] * amp = amp/2 + c_q (amp cb) + s_q (amp sb): the amplitude rides on the per-thread rotation
        const v2f cbE = to_v(lds[T_H + 2 * T + 2 * tid]) * amp2, sbE = to_v(lds[T_H + 2 * T + 2 * tid + 1]) * amp2;
        const v2f halfa = half2 * amp2;
        const int64_t g0 = kk * (int64_t)H;
        if constexpr (PITCH1) {
            // uniform destination in SGPRs + 32-bit lane offset (no 64-bit address arithmetic per store)
            const unsigned long long da = (unsigned long long)(outc + (g0 - p.out_origin));
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);  // (the builtin returns int:
            const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));  // widen as unsigned)
            GFW dst = (GFW)(((unsigned long long)dhi << 32) | dlo);
            if (RC_ABLATE & 4096) dst = outc + ((g0 - p.out_origin) & 0x3FFFF);  // timing only: 1 MiB target
            if (RC_ABLATE & 2048) {  // timing only: the same bytes as 8 x 16-byte stores (wrong places)
                typedef float v4f __attribute__((ext_vector_type(4)));
                v2f oo[PH];
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                                   __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                    oo[q] = (head[q] + tail[q]) * er;
                }
#pragma unroll
                for (int q = 0; q < PH; q += 2) {
                    const v4f o4 = {oo[q].x, oo[q].y, oo[q + 1].x, oo[q + 1].y};
                    __builtin_nontemporal_store(o4, (v4f RC_AS1 *)(dst + 4 * T * (q / 2) + 2 * lane2));
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                const v2f o = (head[q] + tail[q]) * er;  // (y + tail) * (env * amp), stretcher.rs:97-100
                if ((RC_ABLATE & 1024) && o.x != 1.2345e-30f) continue;
#if RC_NTSTORE
                __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * q + lane2));
#else
                *(GV2W)(dst + 2 * T * q + lane2) = o;
#endif
            }
        } else {
            const int64_t kq = g0 / pitch;
            const uint32_t kr = (uint32_t)(g0 % pitch);
            GFW dst = outc + (kq - p.out_origin);
            int t2 = tid;
            opaque(t2);
            // F[t] = O[t * pitch] (src/resampler.rs:3-18): sample a = kr + i of this hop is kept iff a % pitch == 0,
            // at dst[a / pitch]. One division per hop and lane (q = 0); every further register pair is 2T = 512
            // samples on: quotient and remainder advance by the uniform 512 / pitch and 512 % pitch
            const uint32_t a00 = kr + 2u * (uint32_t)t2;
            uint32_t d = a00 / pitch, r = a00 - d * pitch;
            const uint32_t qs = (2u * T) / pitch, rs = (2u * T) - qs * pitch;
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                const v2f o = (head[q] + tail[q]) * er;
                if (!(RC_ABLATE & 1024) || o.x == 1.2345e-30f) {  // (bit 1024, timing only: no output stores)
                    if (r == 0) dst[d] = o.x;                 // a0 = d * pitch
                    if (r + 1 == pitch) dst[d + 1] = o.y;     // a1 = a0 + 1 = (d + 1) * pitch
                }
                d += qs;
                r += rs;
                if (r >= pitch) {
                    r -= pitch;
                    d += 1;
                }
            }
        }
    };
    // Deferred output stores (pitch 1). A CU drains about 11 bytes per clock towards memory: the 8 KiB a wave
    // writes per hop take ~750 cycles, and the four waves of a workgroup reach their epilogues together, so 16
    // back-to-back stores stall a wave for ~2 500 cycles at issue (timing-only builds without the stores run
    // 9 % faster, with 16-byte stores or an L2-resident target no faster). The epilogue therefore only
    // computes the 16 output pairs; they are stored one or two at a time between the butterflies of the NEXT
    // hop's first pass, behind that hop's input loads in issue order (loads no longer queue behind stores).
    constexpr int DN = RC_DEFER_STORE;  // output pairs [PH - DN, PH) of a hop are stored during the next hop
    constexpr bool DEFER = PITCH1 && DN > 0;
    constexpr int D0 = PH - (DN > 0 ? DN : PH);
    v2f od[DN > 0 ? DN : 1];
    bool pend = false;
    int64_t pend_k = 0;
    auto emit = [&](int q0, int q1) {
        if (!DEFER || !pend) return;
        const unsigned long long da = (unsigned long long)(outc + (pend_k * (int64_t)H - p.out_origin));
        const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);
        const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
        GFW dst = (GFW)(((unsigned long long)dhi << 32) | dlo);
#pragma unroll
        for (int q = q0; q < q1; ++q)
            if (q >= D0) __builtin_nontemporal_store(od[q - D0], (GV2W)(dst + 2 * T * q + lane2));
    };
    for (int64_t k = ((k_begin > 0 && !stash_first) ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        st.mark(26);
        {   // register brev5(q) := z[q * T + t] * window ; F1 = stages 0..4
            GF src = hop_src(p, xc, xt, k);
            float xr0[P], xr1[P];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                if (RC_ABLATE & 512) {
                    xr0[q] = (float)(lane2 + q) + (float)k;
                    xr1[q] = xr0[q] * 0.5f;
                    continue;
                }
                xr0[q] = (src + 2 * T * q)[lane2];
                xr1[q] = (src + 2 * T * q)[lane2 + 1];
            }
            const v2f cb = to_v(lds[T_H + 2 * tid]), sb = to_v(lds[T_H + 2 * tid + 1]);
            // stage 0 pairs registers brev5(q) and brev5(q + 16) = brev5(q) + 1: a +- b with a = x_q w_q and
            // b = x_{q+16} w_{q+16} is one multiply and two FMAs
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2f wl = __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cb, half2));
                const v2f wh = __builtin_elementwise_fma(v2f{HANN_W14.s[q + 16], HANN_W14.s[q + 16]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q + 16], HANN_W14.c[q + 16]}, cb, half2));
                const v2f a = v2f{xr0[q], xr1[q]} * wl, xh = v2f{xr0[q + 16], xr1[q + 16]};
                v[2 * brev_c(q, 4)] = __builtin_elementwise_fma(xh, wh, a);
                v[2 * brev_c(q, 4) + 1] = __builtin_elementwise_fma(-xh, wh, a);
                if (q & 1) emit(q / 2, q / 2 + 1);
            }
            st.mark(0);
            if (DEFER) {
                dit_stages<32, m, 1, 1, 0, false, false>(v);
                emit(8, 10);
                dit_stages<32, m, 2, 2, 0, false, false>(v);
                emit(10, 12);
                dit_stages<32, m, 3, 3, 0, false, false>(v);
                emit(12, 14);
                dit_stages<32, m, 4, 4, 0, false, false>(v);
                emit(14, 16);
                pend = false;
            } else {
                dit_stages<32, m, 1, 4, 0, false, false>(v);
            }
            st.mark(1);
        }
        // ---- E1 (cross-wave), round A: position bit 4 clear. Own region (last read by this wave in
        // the previous hop's E4 round B), then everybody reads everywhere.
        wave_fence();
        {
            const Who w = who();
            const int A0 = w.A0, A1 = w.wv * REG + (w.lane ^ 16);
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[((q >> 3) & 1 ? A1 : A0) + q * 64] = to_f2(v[q]);
        }
        st.mark(2);
        HOP4_BAR();
        st.mark(3);
        v2f w2[P];  // register = position bits 4..8
        {
            const Who w = who();
            const int b4u = (int)(__brev((unsigned)w.lo4) >> 28), l3 = w.nib >> 3;
            const int bE1Ae = w.nib * 64 + b4u + 16 * l3, bE1Ao = w.nib * 64 + b4u - 16 * l3;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int bj = brev_c(j, 4), X = bj & 3;
                w2[2 * j] = to_v(lds[(bj >> 2) * REG + X * 16 + ((X & 1) ? bE1Ao : bE1Ae)]);
            }
        }
        st.mark(4);
        HOP4_BAR();
        st.mark(5);
        // round B: position bit 4 set. Written into the region of the wave that will read it.
        {
            int t = tid;
            opaque(t);
            // brev4(t >> 4) * 64 + brev4(t & 15)
            const int bE1Bs = (int)(__brev((unsigned)(t >> 4)) >> 28) * 64 + (int)(__brev((unsigned)(t & 15)) >> 28);
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[phi_c(q) * REG + cidx_c(q) * 16 + bE1Bs] = to_f2(v[16 + q]);
        }
        st.mark(6);
        HOP4_BAR();
        st.mark(7);
        Who w = who();
#pragma unroll
        for (int j = 0; j < 16; ++j) w2[2 * j + 1] = to_v(lds[w.A0 + j * 64]);
        st.mark(8);
        dit_stages<32, m, 5, 8, 4, false, true>(w2, to_v(lds[T_B + w.nib]));
        st.mark(9);
        // ---- E2 (wave-local): position bit 8 clear (residues r), then set (residues 512 - r)
        wave_fence();
        v2f va[16], vb[16];
        {
            w = who();
            const int rlow = (256 - w.r) & 255;                       // (512 - r) - 256
            const int nibb = rlow & 15;                               // its nibble (same class), rho = rlow >> 4
            const int cb = (((nibb >> 2) & 1) << 1) | (nibb >> 3);    // cidx(nibb)
            const int bE2A = w.wv * REG + w.lo4 * 65 + w.cc * 16;
            const int bE2B = w.wv * REG + (rlow >> 4) * 65 + cb * 16;
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[w.A0 + q * 65] = to_f2(w2[q]);
            wave_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) va[q] = to_v(lds[bE2A + q]);
            wave_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[w.A0 + q * 65] = to_f2(w2[16 + q]);
            wave_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) vb[q] = to_v(lds[bE2B + q]);
            wave_fence();
        }
        st.mark(10);
        w = who();
        const int r = w.r;
        {
            const v2f wa = to_v(lds[T_A + r]);  // W_8192^r
            const v2f k16 = {W32_RE[2], W32_IM[2]};
            v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_8192^(512 - r); thread 0: rb = 256 -> W_32
            if (tid == 0) wb = v2f{W32_RE[1], W32_IM[1]};
            dit_stages<16, m, 9, 12, 9, false, true>(va, wa);
            dit_stages<16, m, 9, 12, 9, false, true>(vb, wb);
        }
        st.mark(11);
        // ---- middle stage in registers: pair (A[q], B[15 - q]) = bins (r + 512 q, M - that).
        // Thread 0 owns the two residues that pair with themselves (0 and 256): its 32 bins form 17 pairs,
        // (512 q, 512 (16 - q)), (256 + 512 i, 256 + 512 (15 - i)) and the self-paired bins 0 and 4096.
        // Wave 0 re-deals lane 0's registers (v_cndmask, a uniform branch for the other waves) so that the
        // same 16 slots compute 16 of them - slots 0..7 on residue 0 with bin 0 as slot 0 (dc), slots 8..15
        // on residue 256 through a second per-lane twiddle base / hash counter - and computes bin 4096 as
        // one extra pair. (hop2 / hop3 hand these pairs to 17 lanes through an LDS scratch: four dependent
        // LDS round trips on wave 0 alone, ~4 000 cycles per hop that the other three waves then wait for
        // at the next barrier.)
        const bool is0 = tid == 0;
        v2f s8 = va[8];
        if (wv == 0) {
            const v2f va0 = va[0];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const v2f a = va[8 + i], b0 = vb[i], b1 = vb[8 + i];
                const v2f nx = i < 7 ? va[9 + i] : va0;
                va[8 + i] = vsel(is0, b0, a);
                vb[i] = vsel(is0, b1, b0);
                vb[8 + i] = vsel(is0, nx, b1);
            }
        }
        {
            const float2 wrl = lds[T_R + r];
            const float2 wrh = lds[is0 ? SCR + 2 : T_R + r];         // thread 0: W_N^(256 - 4096)
            const uint32_t x0 = (uint32_t)r * key.mul + key.k0;
            const uint32_t dx = (uint32_t)RES * key.mul;
            const uint32_t x0h = x0 - (is0 ? 3840u * key.mul : 0u);   // thread 0: bins 256 + 512 (q - 8)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float2 wr = q < 8 ? wrl : wrh;
                const v2f wrv = to_v(wr);
                const v2f wq = q == 0 ? wrv : (q == 8 ? v2f{wr.y, -wr.x}
                               : vcmul(wrv, v2f{W32_RE[q & 15], W32_IM[q & 15]}));
                v2f VA, VB;
                v2f gq = {1.0f, 1.0f};
                if constexpr (BAND) {  // bins ja and M - ja of this slot (thread 0's slots q >= 8: residue 256)
                    const uint32_t ja = (uint32_t)r + (uint32_t)RES * (uint32_t)q - ((q >= 8 && is0) ? 3840u : 0u);
                    gq = v2f{band_gain(p, ja), band_gain(p, (uint32_t)M - ja)};
                }
                if (q == 0)
                    HOP4_PAIR<LOG2N, true, BAND>(va[q], vb[15 - q], wq, x0, key, VA, VB, is0, gq);
                else
                    HOP4_PAIR<LOG2N, false, BAND>(va[q], vb[15 - q], wq, (q < 8 ? x0 : x0h) + (uint32_t)q * dx, key, VA, VB,
                                                  false, gq);
                va[q] = VA;
                vb[15 - q] = VB;
            }
        }
        st.mark(12);
        if (wv == 0) {
            // bin 4096 = M / 2 pairs with itself: exp(-2 pi i 4096 / N) = -i, counter of bin 4096
            v2f V8, V8b;
            const float g8 = BAND ? band_gain(p, 8u * (uint32_t)RES) : 1.0f;
            HOP4_PAIR<LOG2N, false, BAND>(s8, s8, v2f{0.0f, -1.0f}, 8u * (uint32_t)RES * key.mul + key.k0, key, V8, V8b,
                                          false, v2f{g8, g8});
            v2f na[8], nb0[8], nb1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                na[i] = vsel(is0, i == 0 ? V8 : vb[7 + i], va[8 + i]);  // va[8+i] <- vb'[8 + (i - 1)]
                nb0[i] = vsel(is0, va[8 + i], vb[i]);                  // vb[i]   <- va'[8 + i]
                nb1[i] = vsel(is0, vb[i], vb[8 + i]);                  // vb[8+i] <- vb'[i]
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                va[8 + i] = na[i];
                vb[i] = nb0[i];
                vb[8 + i] = nb1[i];
            }
        }
        st.mark(13);
        // ---- inverse: I1 in registers
        v2f pa[16], pb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            pa[brev_c(q, 4)] = va[q];
            pb[brev_c(q, 4)] = vb[q];
        }
        dit_stages<16, m, 0, 3, 0, true, false>(pa);
        dit_stages<16, m, 0, 3, 0, true, false>(pb);
        st.mark(14);
        // ---- E3 (wave-local): inverse position bit 4 clear (residue r), then set (512 - r)
        wave_fence();
        {
            w = who();
            const int bE3A = w.wv * REG + w.lo4 * 65 + w.cc * 16;        // l4' * 65 + c * 16, l4' = lo4
            // round B: element 256 + x, x = brev4(j) << 4 | nib, is held by thread (256 - x) & 255: rho_s =
            // 15 - brev4(j) and nibble 16 - nib when nib != 0; rho_s = (16 - brev4(j)) & 15, nibble 0 otherwise
            const int nn = (16 - w.nib) & 15;
            const int cs3 = (((nn >> 2) & 1) << 1) | (nn >> 3);
            const int bE3B = w.wv * REG + w.lo4 * 65 + cs3 * 16 + (w.nib == 0 ? 1 : 0);
            const int bE3B0 = bE3B - (w.nib == 0 ? 16 : 0);              // brev4(j) == 0
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[w.A0 + q * 65] = to_f2(pa[q]);
            wave_fence();
#pragma unroll
            for (int j = 0; j < 16; ++j) v[2 * j] = to_v(lds[bE3A + brev_c(j, 4)]);
            wave_fence();
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[w.A0 + q * 65] = to_f2(pb[q]);
            wave_fence();
#pragma unroll
            for (int j = 0; j < 16; ++j)
                v[2 * j + 1] = to_v(lds[(brev_c(j, 4) == 0 ? bE3B0 : bE3B) + 15 - brev_c(j, 4)]);
            wave_fence();
        }
        st.mark(15);
        w = who();
        dit_stages<32, m, 4, 8, 4, true, true>(v, to_v(lds[T_B + w.lo4]));
        st.mark(16);
        // ---- E4 (cross-wave), round A: inverse position bit 8 clear, own region
        wave_fence();
        {
            w = who();
            const int A0 = w.A0, A1 = w.wv * REG + (w.lane ^ 16);
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[((q & 1) ? A1 : A0) + q * 64] = to_f2(v[q]);
        }
        st.mark(17);
        HOP4_BAR();
        st.mark(18);
        v2f y[P];  // register = position bits 8..12
        {
            int t = tid;
            opaque(t);
            const int rho8 = t >> 4, l4p = t & 15;
            const int bE4Ae = rho8 * 64 + l4p + 16 * (rho8 & 1), bE4Ao = rho8 * 64 + l4p - 16 * (rho8 & 1);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int nj = brev_c(j, 4), cs = cidx_c(nj);
                y[2 * j] = to_v(lds[phi_c(nj) * REG + 32 * (cs >> 1) + ((cs & 1) ? bE4Ao + 16 : bE4Ae)]);
            }
        }
        st.mark(19);
        HOP4_BAR();
        st.mark(20);
        // round B: bit 8 set, written into the reader's region
        {
            w = who();
            const int bE4Bs = (int)(__brev((unsigned)w.nib) >> 28) * 64 + w.lo4;  // brev4(nib) * 64 + l4'
#pragma unroll
            for (int q = 0; q < 16; ++q) lds[(q >> 2) * REG + (q & 3) * 16 + bE4Bs] = to_f2(v[16 + q]);
        }
        st.mark(21);
        HOP4_BAR();
        st.mark(22);
        w = who();
#pragma unroll
        for (int j = 0; j < 16; ++j) y[2 * j + 1] = to_v(lds[w.A0 + j * 64]);
        st.mark(23);
        dit_stages<32, m, 9, 12, 8, true, true>(y, to_v(lds[T_A + tid]));
        st.mark(24);

        // ---- epilogue: synthesis window, overlap-add with the carried tail, store
        const v2f cbW = to_v(lds[T_H + 2 * tid]), sbW = to_v(lds[T_H + 2 * tid + 1]);
        const v2f half2k = {(float)(0.5 * HANN_KAPPA), (float)(0.5 * HANN_KAPPA)};
#pragma unroll
        for (int q = 0; q < P; ++q)
            y[q] *= __builtin_elementwise_fma(v2f{HANN_W14K.s[q], HANN_W14K.s[q]}, sbW,
                    __builtin_elementwise_fma(v2f{HANN_W14K.c[q], HANN_W14K.c[q]}, cbW, half2k));
        if (k >= k_begin) {
            if (stash_first && k == k_begin) {
                int t2 = tid;
                opaque(t2);
                unsigned long long *hs = (unsigned long long *)(p.seam_head + (size_t)gr * H) + t2;
#pragma unroll
                for (int q = 0; q < PH; ++q)
                    __hip_atomic_store(hs + T * q, __builtin_bit_cast(unsigned long long, y[q]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // every storing wave drains its own write-through stores before the barrier; only then
                // may lane 0 publish (MI355X_MICROARCH.md, valid hand-off forms)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0 && !(p.diag_flags & RC_DIAG_SKIP_SEAM_PUBLISH))
                    __hip_atomic_store(p.seam_flag + gr, p.seam_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (DEFER) {
                const v2f amp2 = {p.amp, p.amp};
                const v2f cbE = to_v(lds[T_H + 2 * T + 2 * tid]) * amp2, sbE = to_v(lds[T_H + 2 * T + 2 * tid + 1]) * amp2;
                const v2f halfa = half2 * amp2;
                const unsigned long long da = (unsigned long long)(outc + (k * (int64_t)H - p.out_origin));
                const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);
                const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
                GFW dst = (GFW)(((unsigned long long)dhi << 32) | dlo);
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                                   __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                    const v2f o = (y[q] + tail[q]) * er;  // (y + tail) * (env * amp), stretcher.rs:97-100
                    if (q < D0) __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * q + lane2));
                    else od[q - D0] = o;
                }
                pend = true;
                pend_k = k;
            } else {
                store_head(k, y);
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = y[q + PH];
        st.mark(25);
    }
    emit(0, PH);  // the last hop's outputs
#if RC_STAMP
    if ((tid & 63) == 0 && p.spec) {
        unsigned *dbg = (unsigned *)p.spec + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 32;
        for (int i = 0; i < 32; ++i) dbg[i] = st.acc[i];
    }
#endif
    if (has_next) {
        unsigned *okw = reinterpret_cast<unsigned *>(lds + SCR);
        if (tid == 0) {
            unsigned ok = 0;
            for (unsigned spin = 0; spin < p.seam_spin_limit; ++spin) {
                if (__hip_atomic_load(p.seam_flag + gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                    p.seam_epoch) {
                    ok = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            if (!ok && p.err_word)
                __hip_atomic_store(p.err_word, RC_ERR_SEAM_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            *okw = ok;
        }
        __syncthreads();
        if (*reinterpret_cast<volatile unsigned *>(okw) == 0) return;
        const unsigned long long *hs = (const unsigned long long *)(p.seam_head + (size_t)(gr + 1) * H) + tid;
        v2f head[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q)
            head[q] = __builtin_bit_cast(v2f, __hip_atomic_load(hs + T * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        store_head(k_end, head);
    }
}

// Overlap-add for the user-kernel path (gather form, two terms per output sample).
__global__ __launch_bounds__(256) void ola_kernel(const OlaParams p) {
    const uint32_t N = p.log2n ? 1u << p.log2n : p.n, H = N / 2;
    const int64_t hop_local = blockIdx.x;
    const uint32_t ch = blockIdx.y;
    const int64_t k = p.hop_first + hop_local;
    const float *yk = p.ybuf + ((size_t)ch * p.hop_count + (size_t)hop_local) * N;
    const float *prev = hop_local > 0 ? yk - N + H : p.tail + (size_t)ch * H;
    float *outc = p.out + (size_t)ch * p.out_stride;
    const int64_t g0 = k * (int64_t)H;
    if (p.pitch >= 1) {
        for (uint32_t i = threadIdx.x; i < H; i += blockDim.x) {
            const int64_t g = g0 + i;
            if (p.pitch == 1 || g % p.pitch == 0) {
                const float o = (yk[i] + prev[i]) * p.env[i] * p.amp;
                outc[g / p.pitch - p.out_origin] = o;
            }
        }
    } else {
        // pitch <= -2: one hop per window; resample_slower (src/resampler.rs:20-35) emits
        // (S-1)*f samples lerp(O[i], O[i+1], j/f) from the first S overlap-added samples
        // (src/stretcher.rs:108-111; the rest of the half window is dropped as in the reference)
        const uint32_t f = (uint32_t)(-p.pitch), S = p.samples_needed;
        float *dst = outc + (k * (int64_t)p.window_out_len - p.out_origin);
        for (uint32_t m = threadIdx.x; m < (S - 1) * f; m += blockDim.x) {
            const uint32_t i = m / f, j = m - i * f;
            const float cur = (yk[i] + prev[i]) * p.env[i] * p.amp;
            const float nxt = (yk[i + 1] + prev[i + 1]) * p.env[i + 1] * p.amp;
            dst[m] = cur + (nxt - cur) * ((float)j / (float)f);  // math::lerp, src/math.rs:28-30
        }
    }
}
__global__ __launch_bounds__(256) void ola_save_tail_kernel(const OlaParams p) {
    const uint32_t N = p.log2n ? 1u << p.log2n : p.n, H = N / 2;
    const uint32_t ch = blockIdx.y;
    const float *yl = p.ybuf + ((size_t)ch * p.hop_count + (size_t)(p.hop_count - 1)) * N + H;
    float *t = p.tail + (size_t)ch * H;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < H; i += gridDim.x * blockDim.x)
        t[i] = yl[i];
}

template <int LOG2N>
hipError_t launch_hop_n(HopMode mode, const HopParams &p, hipStream_t s) {
    using G = Geo<LOG2N>;
    const dim3 grid(p.runs_per_channel * p.n_channels), block(G::T);
    const size_t lds = sizeof(float2) * G::LDS_FLOAT2;
    switch (mode) {
        case MODE_FUSED:
            if (RC_V2 && LOG2N == 14) {
                const size_t lds2 = sizeof(float2) * (size_t)HOP2_LDS_FLOAT2;
                const bool hann = p.hann_rot != nullptr;
                if (RC_V4 && hann && !(p.diag_flags & RC_DIAG_PREV_KERNEL)) {
                    const size_t lds4 = sizeof(float2) * (size_t)HOP4_LDS_FLOAT2;
                    if (p.band_on) {
                        if (p.pitch == 1) hipLaunchKernelGGL((hop4_kernel<true, true>), grid, block, lds4, s, p);
                        else hipLaunchKernelGGL((hop4_kernel<false, true>), grid, block, lds4, s, p);
                    } else if (p.pitch == 1) hipLaunchKernelGGL((hop4_kernel<true>), grid, block, lds4, s, p);
                    else hipLaunchKernelGGL((hop4_kernel<false>), grid, block, lds4, s, p);
                } else if (RC_V3 && hann) {
                    const size_t lds3 = sizeof(float2) * (size_t)HOP3_LDS_FLOAT2;
                    if (p.pitch == 1) hipLaunchKernelGGL((hop3_kernel<true>), grid, block, lds3, s, p);
                    else hipLaunchKernelGGL((hop3_kernel<false>), grid, block, lds3, s, p);
                } else
                if (p.pitch == 1 && hann) hipLaunchKernelGGL((hop2_kernel<true, true>), grid, block, lds2, s, p);
                else if (p.pitch == 1) hipLaunchKernelGGL((hop2_kernel<true, false>), grid, block, lds2, s, p);
                else if (hann) hipLaunchKernelGGL((hop2_kernel<false, true>), grid, block, lds2, s, p);
                else hipLaunchKernelGGL((hop2_kernel<false, false>), grid, block, lds2, s, p);
            } else if (p.pitch == 1)
                hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, true>), grid, block, lds, s, p);
            else
                hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, false>), grid, block, lds, s, p);
            break;
        case MODE_FORWARD:
            hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FORWARD, true>), grid, block, lds, s, p);
            break;
        case MODE_RESYNTH:
            hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_RESYNTH, true>), grid, block, lds, s, p);
            break;
    }
    return hipGetLastError();
}


// ======================= large windows (N = 32768 / 65536) ===================================
// z[n] (M = N/2 complex points) = 4 interleaved sequences z_s[n'] = z[4n'+s] of Ms = M/4 points:
//   Z[r + Ms k1] = sum_s W_M^{s r} (-i)^{s k1} Y_s[r],  Y_s = FFT_Ms(z_s)          (forward)
//   y[4n'+s]     = IFFT_Ms(U_s)[n'],  U_s[r] = conj(W_M^{s r}) sum_k1 (+i)^{s k1} V[r + Ms k1]
// Stage A/C reuse the in-LDS passes of the fused kernel on one quarter; stage B is per-bin.
// The last forward pass leaves thread t with the bins brev(t) + T * brev5(q): written straight to
// global memory that is 8 B per lane at a 32-B stride. Thread t therefore takes over, through LDS, the
// 32 registers of thread brev(t) and stores bins t + T * brev5(q): 512 contiguous bytes per wave
// instruction (big_c mirrors it for its loads). Row r of the LDS image starts at (r & 31) + 33 (r >> 5)
// and register q adds 33 T / 32 * q: conflict-free 16-lane stores of row t and 32-lane loads of
// row brev(t).
template <class G>
__device__ __forceinline__ int big_row(int r) { return (r & 31) + 33 * (r >> 5); }
template <class G>
__device__ __forceinline__ int big_brev_tid(int t) {
    return (int)(__brev((unsigned)t) >> (32 - clog2(G::T)));
}
// blocks b and b + 8 share an XCD (MI355X_MICROARCH.md, workgroup dispatch): the four quarter
// transforms of one hop are mapped to one XCD so that their interleaved 8-byte accesses to the hop's
// samples meet in one L2. grid.x = 32 * ceil(hop_count / 8).
__device__ __forceinline__ void big_block(uint32_t b, uint32_t &sub, int64_t &hop_local) {
    const uint32_t g = b >> 5, r = b & 31u;
    sub = r >> 3;
    hop_local = (int64_t)g * 8 + (r & 7u);
}

template <int LOG2NS>  // Geo<LOG2NS>::M == Ms
__global__ __launch_bounds__(Geo<LOG2NS>::T, 2) void big_a_kernel(const BigParams p) {
    using G = Geo<LOG2NS>;
    constexpr int P = G::P, T = G::T, Ms = G::M;
    constexpr int LL = last_lor<G>(G::m);
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    ThreadCtx<G> ctx;
    ctx.tid = threadIdx.x;
    fill_lds_bases<G, G::m>(ctx);
    const int tid = ctx.tid;
    uint32_t sub;
    int64_t hop_local;
    big_block(blockIdx.x, sub, hop_local);
    if (hop_local >= p.hop_count) return;
    const uint32_t ch = blockIdx.y;
    const int64_t k = p.hop_first + hop_local;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GF win = (GF)p.window;
    GF src = (k >= p.tail_hop_first) ? xt + (k * (int64_t)p.step - p.tail_origin)
                                     : xc + (k * (int64_t)p.step - p.in_origin);
    float2 v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int i0 = 2 * (4 * (tid + T * q) + (int)sub);  // samples 2n, 2n+1 of z[n], n = 4n'+s
        v[q] = make_float2(src[i0] * win[i0], src[i0 + 1] * win[i0 + 1]);
    }
    Stamps st;
    st.init();
    forward_passes<G, G::m, 0, true>(v, lds, ctx, (GV2)p.wtab_sub, st);
    GV2W y = (GV2W)p.ysub + (((size_t)ch * p.hop_count + (size_t)hop_local) * 4 + sub) * Ms;
    static_assert(LL == 0 && G::B == 5 && P == 32, "thread t holds positions 32 t + q");
    constexpr int RS = 33 * T / 32;
    static_assert(RS * P <= G::LDS_FLOAT2, "transpose image fits the exchange buffer");
#pragma unroll
    for (int q = 0; q < P; ++q) lds[RS * q + big_row<G>(tid)] = v[q];
    __syncthreads();
    const int rrow = big_row<G>(big_brev_tid<G>(tid));
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = lds[RS * q + rrow];
#pragma unroll
    for (int q = 0; q < P; ++q) stg2(y + tid + T * brev_c(q, 5), v[q]);  // bin t + T brev5(q)
}

template <int LOG2NS>
__global__ __launch_bounds__(Geo<LOG2NS>::T, 2) void big_c_kernel(const BigParams p) {
    using G = Geo<LOG2NS>;
    constexpr int P = G::P, T = G::T, Ms = G::M;
    constexpr int LL = last_lor<G>(G::m);
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    ThreadCtx<G> ctx;
    ctx.tid = threadIdx.x;
    fill_lds_bases<G, G::m>(ctx);
    const int tid = ctx.tid;
    uint32_t sub;
    int64_t hop_local;
    big_block(blockIdx.x, sub, hop_local);
    if (hop_local >= p.hop_count) return;
    const uint32_t ch = blockIdx.y;
    const size_t hop_idx = (size_t)ch * p.hop_count + (size_t)hop_local;
    GV2 u = (GV2)p.ysub + (hop_idx * 4 + sub) * Ms;
    float2 v[P];
    static_assert(LL == 0 && G::B == 5 && P == 32, "thread t holds positions 32 t + q");
    constexpr int RS = 33 * T / 32;
    // coalesced load of bins t + T brev5(q) = the registers of thread brev(t); hand them over
#pragma unroll
    for (int q = 0; q < P; ++q) v[q] = ldg2(u + tid + T * brev_c(q, 5));
    {
        const int wrow = big_row<G>(big_brev_tid<G>(tid));
#pragma unroll
        for (int q = 0; q < P; ++q) lds[RS * q + wrow] = v[q];
        __syncthreads();
        const int rrow = big_row<G>(tid);
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = lds[RS * q + rrow];
        __syncthreads();
    }
    Stamps st;
    st.init();
    inverse_passes<G, G::m>(v, lds, ctx, (GV2)p.wtab_sub, st);
    GF win = (GF)p.window;
    GFW y = (GFW)p.ybuf + hop_idx * (size_t)(8 * Ms);
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int i0 = 2 * (4 * (tid + T * q) + (int)sub);
        stg2((GV2W)(y + i0), make_float2(v[q].x * win[i0], v[q].y * win[i0 + 1]));
    }
}

// Stage C + overlap-add (BigOlaParams): thread t's register q is complex sample n' = t + T q of
// quarter `sub`, i.e. floats i0 = 2 (4 n' + sub), i0 + 1 of y_k; q < 16 is the head, q + 16 the
// matching tail sample (i0 + N/2), so the two-term overlap-add runs in registers as in the fused
// kernels. The hop before a run is recomputed for its tail (its U_s is in the scratch of the chunk);
// the first run of a chunk reads the tail the previous chunk left.
template <int LOG2NS>
__global__ __launch_bounds__(Geo<LOG2NS>::T, 2) void big_cr_kernel(const BigOlaParams pp) {
    using G = Geo<LOG2NS>;
    constexpr int P = G::P, T = G::T, Ms = G::M, PH = P / 2;
    constexpr int LL = last_lor<G>(G::m);
    static_assert(LL == 0 && G::B == 5 && P == 32, "thread t holds positions 32 t + q");
    constexpr int RS = 33 * T / 32;
    constexpr int H = 4 * Ms;  // floats per half window
    const BigParams &p = pp.b;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    ThreadCtx<G> ctx;
    ctx.tid = threadIdx.x;
    fill_lds_bases<G, G::m>(ctx);
    const int tid = ctx.tid;
    uint32_t sub;
    int64_t run;
    big_block(blockIdx.x, sub, run);
    if (run >= (int64_t)pp.runs) return;
    const uint32_t ch = blockIdx.y;
    const int64_t k_first = run * (int64_t)pp.run_len;
    const int64_t k_last = k_first + pp.run_len < p.hop_count ? k_first + pp.run_len : p.hop_count;
    GF win = (GF)p.window;
    GF env = (GF)pp.env;
    GFW outc = (GFW)pp.out + (size_t)ch * pp.out_stride;
    float2 tail[PH];
    if (run == 0) {
        GF tin = (GF)pp.tail_in + (size_t)ch * H;
#pragma unroll
        for (int q = 0; q < PH; ++q) {
            const int i0 = 2 * (4 * (tid + T * q) + (int)sub);
            tail[q] = make_float2(tin[i0], tin[i0 + 1]);
        }
    } else {
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = make_float2(0.f, 0.f);
    }
    Stamps st;
    st.init();
    for (int64_t hl = run == 0 ? k_first : k_first - 1; hl < k_last; ++hl) {
        const size_t hop_idx = (size_t)ch * p.hop_count + (size_t)hl;
        GV2 u = (GV2)p.ysub + (hop_idx * 4 + sub) * Ms;
        float2 v[P];
        int t2 = tid;
        opaque(t2);  // addresses are recomputed per hop instead of being hoisted into live registers
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = ldg2(u + t2 + T * brev_c(q, 5));
        {
            const int wrow = big_row<G>(big_brev_tid<G>(tid));
#pragma unroll
            for (int q = 0; q < P; ++q) lds[RS * q + wrow] = v[q];
            __syncthreads();
            const int rrow = big_row<G>(tid);
#pragma unroll
            for (int q = 0; q < P; ++q) v[q] = lds[RS * q + rrow];
            __syncthreads();
        }
        inverse_passes<G, G::m>(v, lds, ctx, (GV2)p.wtab_sub, st);
        GF wsrc = per_hop(p.window);
#pragma unroll
        for (int q = 0; q < P; ++q) {  // table loads in groups of 8 (all 64 at once would spill)
            const int i0 = 2 * (4 * (t2 + T * q) + (int)sub);
            v[q] = make_float2(v[q].x * wsrc[i0], v[q].y * wsrc[i0 + 1]);
            if ((q & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        if (hl >= k_first && !pp.tail_only) {
            const int64_t g0 = (p.hop_first + hl) * (int64_t)H;
            GF esrc = per_hop(pp.env);
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                if ((q & 7) == 0) __builtin_amdgcn_sched_barrier(0);
                const int i0 = 2 * (4 * (t2 + T * q) + (int)sub);
                // same operation order as ola_kernel / src/stretcher.rs:97-100
                const float o0 = (v[q].x + tail[q].x) * esrc[i0] * pp.amp;
                const float o1 = (v[q].y + tail[q].y) * esrc[i0 + 1] * pp.amp;
                const int64_t g = g0 + i0;
                if (pp.pitch == 1) {
                    stg2((GV2W)(outc + (g - pp.out_origin)), make_float2(o0, o1));
                } else {
                    if (g % pp.pitch == 0) outc[g / pp.pitch - pp.out_origin] = o0;
                    if ((g + 1) % pp.pitch == 0) outc[(g + 1) / pp.pitch - pp.out_origin] = o1;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = v[q + PH];
        __syncthreads();  // the exchange buffer is free again
    }
    (void)win;
    (void)env;
    if (run + 1 == (int64_t)pp.runs) {
        GFW tout = (GFW)pp.tail_out + (size_t)ch * H;
#pragma unroll
        for (int q = 0; q < PH; ++q) {
            const int i0 = 2 * (4 * (tid + T * q) + (int)sub);
            stg2((GV2W)(tout + i0), tail[q]);
        }
    }
}

// ======================= fused large-window kernel (N = 32768 / 65536) ==========================
// big_a / big_b / big_cr above move ~8 N bytes of scratch per hop through HBM. big4_kernel keeps a whole
// hop inside one workgroup: T = 512 threads hold the M = N/2 = 512 R complex points, R = 32 (N = 32768) or
// 64 (N = 65536) per thread, b = log2 R (tests/dev/proto_big.py is the index model):
//   F1  stages 0..b-1     on the R registers (constants only), thread t = low 9 bits of the sample index
//   F2  stages b..b+4     on R/32 groups of 32 registers, thread = (lf = p0..p4, uu = the top 4 position bits)
//   F3  stages b+5..b+8   on R/16 sets of 16 registers: thread tau holds the residues tau and RES - tau
//                         (and tau + 512, RES - 512 - tau for R = 64), RES = 2^(b+5), so every (j, M - j) bin
//                         pair sits in one thread and the middle stage runs in registers as in hop4_kernel
//   I1 / I2 / I3 mirror them (4, 5 and b stages); the synthesis window and the two-term overlap-add follow
//   in registers (R = 32) - for R = 64 the carried tail y_{k-1}[H..] does not fit the register file next to
//   128 data registers and travels through a per-workgroup scratch of 128 KiB (written and re-read by the
//   same CU one hop later: L2 / Infinity-Cache traffic, not HBM).
// Exchanges go through one 16 400-element LDS buffer (131 KB: one workgroup = 8 waves per CU), a single
// round for R = 32, two rounds of 32 registers per thread for R = 64.
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
template <int NREG, int S_LO, int S_HI, int REG_LO, bool CONJ, bool HAS_L>
__device__ __forceinline__ void dit_g(v2f (&v)[NREG], v2f wfine = v2f{1.0f, 0.0f}) {
    const v2f sgn = CONJ ? v2f{1.0f, -1.0f} : v2f{-1.0f, 1.0f};
    v2f bases[S_HI - S_LO + 1];
    if (HAS_L) {
        bases[S_HI - S_LO] = wfine;
#pragma unroll
        for (int s = S_HI - 1; s >= S_LO; --s) bases[s - S_LO] = vcmul(bases[s + 1 - S_LO], bases[s + 1 - S_LO]);
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

#ifndef RC_B4_LB
#define RC_B4_LB 32
#endif
#ifndef RC_B4_TAILREG_MAX
#define RC_B4_TAILREG_MAX 32  // largest R whose carried tail lives in registers (above: per-workgroup scratch)
#endif
#ifndef RC_B4_LGKM
#define RC_B4_LGKM 1
#endif
// the exchange barriers order LDS traffic only: global stores of the epilogue (this thread's own output and tail
// addresses) may still be in flight when the next hop starts
#define BIG4_BAR()                                                                       \
    do {                                                                                 \
        if (RC_B4_LGKM) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  \
        else __syncthreads();                                                            \
    } while (0)
#ifndef RC_B4_TAILNT
#define RC_B4_TAILNT 0  // 1: non-temporal tail-scratch accesses (experiment)
#endif
__device__ __forceinline__ v2f tail_ld(GV2 p) { return RC_B4_TAILNT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void tail_st(GV2W p, v2f v) {
    if (RC_B4_TAILNT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#ifndef RC_B4_DMA
#define RC_B4_DMA 0
#endif
#ifndef RC_B4_ABL
#define RC_B4_ABL 0  // timing-only ablations of big4_kernel: 1 no input loads, 2 no tail scratch traffic, 4 no output stores
#endif
#ifndef RC_B4_EB
#define RC_B4_EB 4
#endif
#ifndef RC_B4_TPRE
#define RC_B4_TPRE 0
#endif
constexpr int BIG4_T = 512;
constexpr int BIG4_XBUF = 16400;  // exchange buffer, float2 slots (16384 + the 15 of the E1 / E3 index map)
// tables behind the buffer: W_M^r [TA], W_N^r [TR] for r <= RES/2, thread 0's second twiddle base
constexpr int big4_lds_float2(int R) { return BIG4_XBUF + 2 * (16 * R + 1) + 8; }

template <int R, bool PITCH1, bool HANN>
__global__ __launch_bounds__(BIG4_T, 2) void big4_kernel(const HopParams p) {
    constexpr int b = clog2(R), m = b + 9, LOG2N = m + 1, M = 1 << m, H = M, T = BIG4_T;
    constexpr int RES = 1 << (b + 5), G = R / 32, NS = R / 16, PH = R / 2;
    constexpr int T_A = BIG4_XBUF, T_R = T_A + RES / 2 + 1, SCR = T_R + RES / 2 + 1;
    constexpr bool TAIL_GLOBAL = R > RC_B4_TAILREG_MAX;
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
    GV2W tsc = (GV2W)p.ybuf + (size_t)blockIdx.x * (H / 2);  // TAIL_GLOBAL: this workgroup's tail scratch
    if constexpr (TAIL_GLOBAL) {  // uniform base in SGPRs + a 32-bit lane offset: no 64-bit address per access
        const unsigned long long ta = (unsigned long long)tsc;
        const unsigned tlo = __builtin_amdgcn_readfirstlane((unsigned)ta);
        const unsigned thi = __builtin_amdgcn_readfirstlane((unsigned)(ta >> 32));
        tsc = (GV2W)(((unsigned long long)thi << 32) | tlo);
    }
    const unsigned lane2 = 2u * (unsigned)tid;
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;
    const int wv = tid >> 6;
    {
        GV2 wt = (GV2)p.wtab;  // [RES/2 + 1] exp(-2 pi i k / M)
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N)
        for (int i = tid; i <= RES / 2; i += T) {
            lds[T_A + i] = ldg2(wt + i);
            lds[T_R + i] = ldg2(rt + i);
        }
        if (tid == 0) {  // W_N^(RES/2 - M/2) = i W_N^(RES/2): thread 0's twiddle base for its second residue
            const float2 wq = ldg2(rt + RES / 2);
            lds[SCR] = make_float2(-wq.y, wq.x);
        }
        __syncthreads();
    }
    v2f tail[TAIL_GLOBAL ? 1 : PH];
    if constexpr (!TAIL_GLOBAL) {
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    } else {
#pragma unroll
        for (int q = 0; q < PH; ++q) stg2(tsc + T * q + (unsigned)tid, make_float2(0.f, 0.f));
    }
    // thread identities
    const int lf = tid & 31, uu = tid >> 5;   // F2
    const int l4 = tid & 15, hi = tid >> 4;   // I2
    const bool is0 = tid == 0;
    Stamps stp;
    stp.init();
    // RC_B4_DMA (R = 32): the next hop's whole window (N floats = the exchange buffer's size) is fetched by LDS-DMA
    // (global_load_lds_dwordx4: no VGPRs) into the exchange buffer while it is idle - from the last E4 read to the
    // next E1 write - so that its latency runs under I3 and the epilogue instead of in front of F1
    constexpr bool DMA = RC_B4_DMA && R == 32;
    bool dma_ready = false;
    for (int64_t k = (k_begin > 0 ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        int tt = tid;  // per-hop opaque copy for the scratch addresses (hoisted, they would be 2 PH live VGPRs)
        opaque(tt);
        v2f v[R];
        {   // register brev_b(q) := z[q * T + t] * window
            GF src = hop_src(p, xc, xt, k);
            GF win = per_hop(p.window);
            v2f cbW = {0.f, 0.f}, sbW = cbW;
            if constexpr (HANN) {  // {cos, sin}(beta) of this thread's two samples, from the engine's table
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tid;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
            }
            const HannK64 &HW = R == 32 ? HANN_W15 : HANN_W16;
            // every load of the hop in flight at once when the window is computed (one memory latency per
            // hop); batches of 16 when the window comes from its table too (register budget)
            constexpr int LB = HANN ? (R > 32 ? RC_B4_LB : R) : 16;
            if (DMA && dma_ready) {  // every wave waits for its own DMAs, then all of them are visible to all.
                // The PH output stores of the previous hop were issued behind the DMAs and may stay in flight
                // (vector memory operations retire in order)
                if (PITCH1 && k - 1 >= k_begin) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
#pragma unroll
            for (int q0 = 0; q0 < R; q0 += LB) {
                float xr0[LB], xr1[LB], wr0[HANN ? 1 : LB], wr1[HANN ? 1 : LB];
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    if (DMA && dma_ready) {  // (uniform) z[n], n = tid + 512 q, sits at float2 slot n
                        const float2 zz = lds[tid + T * (q0 + q)];
                        xr0[q] = zz.x;
                        xr1[q] = zz.y;
                    } else
                    if (RC_B4_ABL & 1) {  // timing only: no input loads
                        xr0[q] = (float)(lane2 + q0 + q) + (float)k;
                        xr1[q] = xr0[q] * 0.5f;
                    } else {
                        xr0[q] = (src + 2 * T * (q0 + q))[lane2];
                        xr1[q] = (src + 2 * T * (q0 + q))[lane2 + 1];
                    }
                    if constexpr (!HANN) {
                        wr0[q] = (win + 2 * T * (q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * (q0 + q))[lane2 + 1];
                    }
                }
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    v2f wq;
                    if constexpr (HANN)
                        wq = __builtin_elementwise_fma(v2f{HW.s[q0 + q], HW.s[q0 + q]}, sbW,
                             __builtin_elementwise_fma(v2f{HW.c[q0 + q], HW.c[q0 + q]}, cbW, v2f{0.5f, 0.5f}));
                    else
                        wq = v2f{wr0[q], wr1[q]};
                    v[brev_c(q0 + q, b)] = v2f{xr0[q], xr1[q]} * wq;
                }
            }
            stp.mark(0);
            dit_g<R, 0, b - 1, 0, false, false>(v);
        }
        stp.mark(1);
        // ---- E1: F1 -> F2, round g moves the registers with position bit 5 = g
        v2f w[R];
        {
            const int bs = (int)(__brev((unsigned)tid) >> 23);           // brev9(t)
            const int b1s = (bs << 5) + (bs >> 5);                       // e1(q | brev9(t) << 5) = q + this
            const int b1l = lf + (uu << 10) + uu;                        // e1(lf | j << 5 | uu << 10) = (j << 5) + this
#pragma unroll
            for (int g = 0; g < G; ++g) {
                BIG4_BAR();
#pragma unroll
                for (int q = 0; q < 32; ++q) lds[b1s + q] = to_f2(v[32 * g + q]);
                BIG4_BAR();
#pragma unroll
                for (int j = 0; j < 32; ++j) w[32 * g + j] = to_v(lds[b1l + (j << 5)]);
            }
        }
        stp.mark(2);
        {   // F2: stages b..b+4 on each group; base W_RES^(lf | g << 5) = W_M^(16 lf) * (g ? W_64 : 1)
            const v2f wf0 = to_v(lds[T_A + 16 * lf]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = w[32 * g + j];
                dit_g<32, b, b + 4, b, false, true>(grp, g ? vcmul(wf0, v2f{W64.re[1], W64.im[1]}) : wf0);
#pragma unroll
                for (int j = 0; j < 32; ++j) w[32 * g + j] = grp[j];
            }
        }
        stp.mark(3);
        // ---- E2: F2 -> F3. slot = (residue mod 1024) | uu << 10; R = 64: round 0 = residues < 1024
        v2f st[NS][16];
        {
            const int b2s = lf | (uu << 10);
#pragma unroll
            for (int rnd = 0; rnd < G; ++rnd) {
                BIG4_BAR();
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    if (R == 32) lds[b2s + (kk << 5)] = to_f2(w[kk]);
                    else lds[b2s + ((kk >> 4) << 5) + ((kk & 15) << 6)] = to_f2(w[32 * (kk >> 4) + 16 * rnd + (kk & 15)]);
                }
                BIG4_BAR();
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (R == 64 && ((s & 1) != rnd)) continue;
                    const int r0 = tid + 512 * (s >> 1);
                    const int res = (s & 1) ? ((r0 == 0 ? RES / 2 : RES - r0) & 1023) : (r0 & 1023);
#pragma unroll
                    for (int q = 0; q < 16; ++q) st[s][q] = to_v(lds[res + (q << 10)]);
                }
            }
        }
        stp.mark(4);
        // ---- F3 on every set, middle stage on every (A, B) pair of sets, I1
#pragma unroll
        for (int gp = 0; gp < NS / 2; ++gp) {
            const int r = tid + 512 * gp;
            v2f(&va)[16] = st[2 * gp];
            v2f(&vb)[16] = st[2 * gp + 1];
            {
                const v2f wa = to_v(lds[T_A + r]);  // W_M^r
                const v2f k16 = {W32_RE[2], W32_IM[2]};
                v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
                if (gp == 0 && is0) wb = v2f{W32_RE[1], W32_IM[1]};  // thread 0: residue RES/2 -> W_32
                dit_g<16, b + 5, b + 8, b + 5, false, true>(va, wa);
                dit_g<16, b + 5, b + 8, b + 5, false, true>(vb, wb);
            }
            // thread 0, group 0: residues 0 and RES/2 pair with themselves (hop4_kernel's re-deal)
            const bool sp = gp == 0 && is0;
            v2f s8 = va[8];
            if (gp == 0 && wv == 0) {
                const v2f va0 = va[0];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const v2f a = va[8 + i], b0 = vb[i], b1 = vb[8 + i];
                    const v2f nx = i < 7 ? va[9 + i] : va0;
                    va[8 + i] = vsel(sp, b0, a);
                    vb[i] = vsel(sp, b1, b0);
                    vb[8 + i] = vsel(sp, nx, b1);
                }
            }
            {
                const float2 wrl = lds[T_R + r];
                const float2 wrh = lds[sp ? SCR : T_R + r];
                const uint32_t x0 = (uint32_t)r * key.mul + key.k0;
                const uint32_t dx = (uint32_t)RES * key.mul;
                const uint32_t x0h = x0 - (sp ? (uint32_t)(M / 2 - RES / 2) * key.mul : 0u);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float2 wr = q < 8 ? wrl : wrh;
                    const v2f wrv = to_v(wr);
                    const v2f wq = q == 0 ? wrv : (q == 8 ? v2f{wr.y, -wr.x}
                                   : vcmul(wrv, v2f{W32_RE[q & 15], W32_IM[q & 15]}));
                    v2f VA, VB;
                    if (q == 0 && gp == 0)
                        pair_regs_pk4<LOG2N, true>(va[q], vb[15 - q], wq, x0, key, VA, VB, sp);
                    else
                        pair_regs_pk4<LOG2N>(va[q], vb[15 - q], wq, (q < 8 ? x0 : x0h) + (uint32_t)q * dx, key, VA, VB);
                    va[q] = VA;
                    vb[15 - q] = VB;
                }
            }
            if (gp == 0 && wv == 0) {  // bin M/2 pairs with itself; un-deal thread 0's registers
                v2f V8, V8b;
                pair_regs_pk4<LOG2N>(s8, s8, v2f{0.0f, -1.0f}, 8u * (uint32_t)RES * key.mul + key.k0, key, V8, V8b);
                v2f na[8], nb0[8], nb1[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    na[i] = vsel(sp, i == 0 ? V8 : vb[7 + i], va[8 + i]);
                    nb0[i] = vsel(sp, va[8 + i], vb[i]);
                    nb1[i] = vsel(sp, vb[i], vb[8 + i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    va[8 + i] = na[i];
                    vb[i] = nb0[i];
                    vb[8 + i] = nb1[i];
                }
            }
            // I1: inverse stages 0..3, register index = brev4(q)
            v2f pa[16], pb[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                pa[brev_c(q, 4)] = va[q];
                pb[brev_c(q, 4)] = vb[q];
            }
            dit_g<16, 0, 3, 0, true, false>(pa);
            dit_g<16, 0, 3, 0, true, false>(pb);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                va[q] = pa[q];
                vb[q] = pb[q];
            }
        }
        stp.mark(5);
        // ---- E3: I1 -> I2. element P = q' | brev_{b+5}(residue) << 4; R = 64: P4 (= residue >= 1024) is the round
        {
#pragma unroll
            for (int rnd = 0; rnd < G; ++rnd) {
                BIG4_BAR();
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (R == 64 && ((s & 1) != rnd)) continue;
                    const int r0 = tid + 512 * (s >> 1);
                    const int res = (s & 1) ? (r0 == 0 ? RES / 2 : RES - r0) : r0;
                    const int br = (int)(__brev((unsigned)res) >> (32 - (b + 5)));  // brev_{b+5}(residue) = P4..
                    const int hiP = R == 32 ? br : (br >> 1);                     // drop P4 for R = 64
                    const int n0 = hiP << 4;                                     // reduced index, q' = 0
                    const int base = n0 + ((n0 >> 10) & 15);
#pragma unroll
                    for (int q = 0; q < 16; ++q) lds[base + q] = to_f2(st[s][q]);
                }
                BIG4_BAR();
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    // R = 32: register kk = P4..P8; R = 64: kk = (group g = P14) << 4 | (P5..P8), P4 = rnd
                    const int n = R == 32 ? (l4 | (kk << 4) | (hi << 9))
                                          : (l4 | ((kk & 15) << 4) | (hi << 8) | ((kk >> 4) << 13));
                    const v2f x = to_v(lds[n + ((n >> 10) & 15)]);
                    if (R == 32) v[kk] = x;
                    else v[32 * (kk >> 4) + 2 * (kk & 15) + rnd] = x;
                }
            }
        }
        stp.mark(6);
        {   // I2: inverse stages 4..8 on each group, base W_512^l4 = W_M^(l4 R)
            const v2f wf = to_v(lds[T_A + l4 * R]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = v[32 * g + j];
                dit_g<32, 4, 8, 4, true, true>(grp, wf);
#pragma unroll
                for (int j = 0; j < 32; ++j) v[32 * g + j] = grp[j];
            }
        }
        stp.mark(7);
        // ---- E4: I2 -> I3 (registers = P9.., thread = P0..P8), round g = P14
        v2f y[R];
        {
            const int b4s = l4 | (hi << 9);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                BIG4_BAR();
#pragma unroll
                for (int j = 0; j < 32; ++j) lds[b4s + (j << 4)] = to_f2(v[32 * g + j]);
                BIG4_BAR();
#pragma unroll
                for (int q = 0; q < 32; ++q) y[q + 32 * g] = to_v(lds[tid + (q << 9)]);
            }
        }
        if constexpr (DMA) {
            dma_ready = k + 1 < k_end;
            if (dma_ready) {
                BIG4_BAR();  // every wave has its E4 data: the buffer is free
                typedef __attribute__((address_space(3))) void *LP;
                typedef const __attribute__((address_space(1))) void *GP;
                GF s2 = hop_src(p, xc, xt, k + 1);
                float *ldsf = reinterpret_cast<float *>(lds);
                const int lane = tid & 63;
#pragma unroll
                for (int j = 0; j < 16; ++j) {  // 16 KiB per wave: 16 pieces of 64 lanes x 16 bytes
                    const int c = wv * 16 + j;
                    __builtin_amdgcn_global_load_lds((GP)(s2 + c * 256 + lane * 4), (LP)(ldsf + c * 256), 16, 0, 0);
                }
            }
        }
        stp.mark(8);
        // R = 64: the carried tail comes back from the scratch; requested here, behind the last exchange, so that
        // its latency hides under I3 (v is dead: there are registers for it)
        constexpr bool TPRE = TAIL_GLOBAL && RC_B4_TPRE;
        v2f tpre[TPRE ? PH : 1];
        if constexpr (TPRE) {
#pragma unroll
            for (int q = 0; q < PH; ++q) tpre[q] = tail_ld((GV2)tsc + T * q + (unsigned)tt);
        }
        dit_g<R, 9, m - 1, 9, true, true>(y, to_v(lds[T_A + tid]));
        stp.mark(9);
        // ---- epilogue: synthesis window, overlap-add, store (tail in registers or in the scratch)
        {
            GF win = per_hop(p.window);
            GF esrc = per_hop(p.env);
            v2f cbW = {0.f, 0.f}, sbW = cbW, cbE = cbW, sbE = cbW;
            if constexpr (HANN) {
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tid;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1), e0r = ldg2(hr + 2 * T), e1r = ldg2(hr + 2 * T + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
                cbE = v2f{e0r.x, e1r.x};
                sbE = v2f{e0r.y, e1r.y};
            }
            const HannK64 &HW = R == 32 ? HANN_W15 : HANN_W16;
            const HannK64 &HE = R == 32 ? HANN_E15 : HANN_E16;
            const v2f hf = {0.5f, 0.5f};
            // pair_regs_pk4 leaves the -1/(4N) of the magnitudes out (a power of two): it rides on the amplitude
            const float ak = p.amp * (-0.25f / (float)(1 << LOG2N));
            const v2f ampk = {ak, ak};
            const int64_t g0 = k * (int64_t)H;
            GFW dst = outc + (g0 / (int64_t)pitch - p.out_origin);
            const uint32_t kr = (uint32_t)(g0 % pitch);
            constexpr int EB = R > 32 ? RC_B4_EB : 8;  // batch of table / tail loads in flight (register budget)
            constexpr bool TPIPE = TAIL_GLOBAL && !TPRE;  // tail loads one batch ahead of their use
            v2f tnx[TPIPE ? EB : 1];
            if constexpr (TPIPE) {
#pragma unroll
                for (int q = 0; q < EB; ++q)
                    tnx[q] = (RC_B4_ABL & 2) ? v2f{0.f, 0.f} : tail_ld((GV2)tsc + T * q + (unsigned)tt);
            }
#pragma unroll
            for (int q0 = 0; q0 < PH; q0 += EB) {
                float wr0[EB], wr1[EB], wt0[EB], wt1[EB], e0[EB], e1[EB];
                v2f tq[EB];
                if constexpr (TPIPE) {
#pragma unroll
                    for (int q = 0; q < EB; ++q) tq[q] = tnx[q];
                    if (q0 + EB < PH) {
#pragma unroll
                        for (int q = 0; q < EB; ++q)
                            tnx[q] = (RC_B4_ABL & 2) ? v2f{0.f, 0.f} : tail_ld((GV2)tsc + T * (q0 + EB + q) + (unsigned)tt);
                    }
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    if constexpr (!HANN) {
                        wr0[q] = (win + 2 * T * (q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * (q0 + q))[lane2 + 1];
                        wt0[q] = (win + 2 * T * (q0 + q + PH))[lane2];
                        wt1[q] = (win + 2 * T * (q0 + q + PH))[lane2 + 1];
                        e0[q] = (esrc + 2 * T * (q0 + q))[lane2];
                        e1[q] = (esrc + 2 * T * (q0 + q))[lane2 + 1];
                    } else {
                        const v2f wh = __builtin_elementwise_fma(v2f{HW.s[q0 + q], HW.s[q0 + q]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[q0 + q], HW.c[q0 + q]}, cbW, hf));
                        const v2f wt = __builtin_elementwise_fma(v2f{HW.s[q0 + q + PH], HW.s[q0 + q + PH]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[q0 + q + PH], HW.c[q0 + q + PH]}, cbW, hf));
                        const v2f ev = __builtin_elementwise_fma(v2f{HE.s[q0 + q], HE.s[q0 + q]}, sbE,
                                       __builtin_elementwise_fma(v2f{HE.c[q0 + q], HE.c[q0 + q]}, cbE, hf));
                        wr0[q] = wh.x, wr1[q] = wh.y, wt0[q] = wt.x, wt1[q] = wt.y, e0[q] = ev.x, e1[q] = ev.y;
                    }
                    if constexpr (TPRE) tq[q] = tpre[q0 + q];
                    else if constexpr (TAIL_GLOBAL) {}
                    else tq[q] = tail[q0 + q];
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const v2f head = y[q0 + q] * v2f{wr0[q], wr1[q]};
                    const v2f nt = y[q0 + q + PH] * v2f{wt0[q], wt1[q]};
                    if (k >= k_begin) {
                        // stretcher.rs:97-100 operation order
                        const v2f o = (head + tq[q]) * v2f{e0[q], e1[q]} * ampk;
                        if constexpr (PITCH1) {
                            if (!(RC_B4_ABL & 4) || o.x == 1.2345f)  // (bit 4, timing only: no output stores)
                            __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * (q0 + q) + lane2));
                        } else {
                            const uint32_t a0 = kr + 2u * (uint32_t)(tid + T * (q0 + q)), a1 = a0 + 1;
                            const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                            if (d0 * pitch == a0) dst[d0] = o.x;
                            if (d1 * pitch == a1) dst[d1] = o.y;
                        }
                    }
                    if constexpr (TAIL_GLOBAL) {
                        if (!(RC_B4_ABL & 2)) tail_st(tsc + T * (q0 + q) + (unsigned)tt, nt);
                        else if (nt.x == 1.2345f) stg2(tsc, to_f2(nt));  // (keeps nt alive)
                    }
                    else tail[q0 + q] = nt;
                }
            }
        }
        stp.mark(10);
    }
#if RC_STAMP
    if ((tid & 63) == 0 && p.spec) {
        unsigned *dbg = (unsigned *)p.spec + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 32;
        for (int i = 0; i < 32; ++i) dbg[i] = stp.acc[i];
    }
#endif
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmuli(float2 a) { return make_float2(-a.y, a.x); }    // * (+i)
__device__ __forceinline__ float2 cmulmi(float2 a) { return make_float2(a.y, -a.x); }   // * (-i)
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
// X[k1] = sum_s (-i)^{s k1} a[s]  (SIGN = -1)  or  sum_s (+i)^{s k1} a[s]  (SIGN = +1)
template <int SIGN>
__device__ __forceinline__ void radix4(const float2 (&a)[4], float2 (&o)[4]) {
    const float2 s02 = cadd(a[0], a[2]), d02 = csub(a[0], a[2]);
    const float2 s13 = cadd(a[1], a[3]), d13 = csub(a[1], a[3]);
    o[0] = cadd(s02, s13);
    o[2] = csub(s02, s13);
    const float2 r = SIGN < 0 ? cmulmi(d13) : cmuli(d13);
    o[1] = cadd(d02, r);
    o[3] = csub(d02, r);
}

// Stage B: one thread per residue pair (k2, Ms - k2), k2 in [0, Ms/2].
// MODE_FUSED: analysis + random phases + synthesis; MODE_FORWARD: analysis only, natural-order
// spectrum to p.spec; MODE_RESYNTH: magnitudes from p.spec (after the user kernel), synthesis.
template <int MODE>
__global__ __launch_bounds__(256) void big_b_kernel(const BigParams p) {
    const uint32_t N = 1u << p.log2n, M = N / 2, Ms = M / 4;
    const uint32_t k2 = blockIdx.x * blockDim.x + threadIdx.x;
    if (k2 > Ms / 2) return;
    const int64_t hop_local = blockIdx.y;
    const uint32_t ch = blockIdx.z;
    const int64_t k = p.hop_first + hop_local;
    const uint32_t r = k2, rp = (Ms - k2) & (Ms - 1);
    const size_t hop_idx = (size_t)ch * p.hop_count + (size_t)hop_local;
    GV2W y = (GV2W)p.ysub + hop_idx * 4 * (size_t)Ms;
    GV2W spec = (GV2W)p.spec + hop_idx * (size_t)N;
    GV2 t1 = (GV2)p.t1;
    const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
    float2 wr[4], wp[4], Zr[4], Zp[4], Vr[4], Vp[4];
    wr[0] = wp[0] = make_float2(1.f, 0.f);
    wr[1] = ldg2(t1 + r);
    wr[2] = ldg2(t1 + 2 * r);   // 2r <= Ms
    wr[3] = cmul(wr[1], wr[2]);
    wp[1] = ldg2(t1 + rp);
    wp[2] = cmul(wp[1], wp[1]);
    wp[3] = cmul(wp[1], wp[2]);
    if constexpr (MODE != MODE_RESYNTH) {
        float2 a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[s] = cmul(wr[s], ldg2((GV2)y + (size_t)s * Ms + r));
            b[s] = cmul(wp[s], ldg2((GV2)y + (size_t)s * Ms + rp));
        }
        radix4<-1>(a, Zr);  // Z[r + Ms k1]
        radix4<-1>(b, Zp);  // Z[rp + Ms k1]
    }
    const float2 wbase = ldg2((GV2)p.rtab + k2);  // exp(-2 pi i k2 / N)
    const float c8 = 0.70710678118654752f;
    const float2 e8[4] = {make_float2(1.f, 0.f), make_float2(c8, -c8), make_float2(0.f, -1.f),
                          make_float2(-c8, -c8)};  // exp(-2 pi i k1 / 8)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        const uint32_t J = k2 + Ms * (uint32_t)k1;  // partner M - J
        const float2 w = cmul(wbase, e8[k1]);
        float m1a, m1b, m2a, m2b;  // scaled magnitudes of bins J, N - J, M - J, M + J
        if constexpr (MODE == MODE_RESYNTH) {
            const float nk = -0.5f / (float)N;
            m1a = cabs_fast(ldg2((GV2)spec + J)) * nk;
            m1b = cabs_fast(ldg2((GV2)spec + ((N - J) & (N - 1)))) * nk;
            m2a = cabs_fast(ldg2((GV2)spec + (M - J))) * nk;
            m2b = cabs_fast(ldg2((GV2)spec + ((M + J) & (N - 1)))) * nk;
        } else {
            const float2 A = Zr[k1];
            const float2 Bp = (k2 == 0) ? Zr[(4 - k1) & 3] : Zp[3 - k1];
            float2 X1, X2c;
            pair_analyze(A, Bp, w, X1, X2c);
            if constexpr (MODE == MODE_FORWARD) {  // same stores as do_pair<.., MODE_FORWARD>
                const float2 x1 = make_float2(0.5f * X1.x, 0.5f * X1.y);
                const float2 x2 = make_float2(0.5f * X2c.x, 0.5f * X2c.y);
                stg2(spec + J, x1);
                stg2(spec + ((N - J) & (N - 1)), make_float2(x1.x, J ? -x1.y : x1.y));
                stg2(spec + (M - J), make_float2(x2.x, -x2.y));
                stg2(spec + ((M + J) & (N - 1)), J ? x2 : make_float2(x2.x, -x2.y));
                continue;
            }
            const float nk = -0.25f / (float)N;
            m1a = m1b = cabs_fast(X1) * nk;
            m2a = m2b = cabs_fast(X2c) * nk;
        }
        float c1, s1, c2, s2, c3, s3, c4, s4;
        phase_quad(key, J, M, c1, s1, c2, s2, c3, s3, c4, s4);
        const float px = m1a * c1 + m1b * c2, py = m1a * s1 - m1b * s2;
        const float qx = m2a * c3 + m2b * c4, qy = m2b * s4 - m2a * s3;
        const float sx = px + qx, sy = py + qy, rx = px - qx, ry = py - qy;
        const float ux = rx * w.x + ry * w.y, uy = ry * w.x - rx * w.y;
        Vr[k1] = make_float2(sx - uy, sy + ux);      // V[J]
        Vp[3 - k1] = make_float2(sx + uy, ux - sy);  // V[M - J] = V[rp + Ms (3 - k1)]   (k2 > 0)
    }
    if constexpr (MODE == MODE_FORWARD) return;
    float2 u[4];
    radix4<+1>(Vr, u);
#pragma unroll
    for (int s = 0; s < 4; ++s) stg2(y + (size_t)s * Ms + r, cmul(cconj(wr[s]), u[s]));
    if (k2 != 0 && 2 * k2 != Ms) {
        radix4<+1>(Vp, u);
#pragma unroll
        for (int s = 0; s < 4; ++s) stg2(y + (size_t)s * Ms + rp, cmul(cconj(wp[s]), u[s]));
    }
}

template <int LOG2NS>
hipError_t launch_big_ac(int stage, const BigParams &p, hipStream_t s) {
    using G = Geo<LOG2NS>;
    const dim3 grid((unsigned)((p.hop_count + 7) / 8 * 32), p.n_channels), block(G::T);
    const size_t lds = sizeof(float2) * G::LDS_FLOAT2;
    if (stage == 0) hipLaunchKernelGGL((big_a_kernel<LOG2NS>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((big_c_kernel<LOG2NS>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace

int hop_workgroups_per_cu(int log2n, bool default_window) {
    return (RC_V3 && RC_V2 && log2n == 14 && default_window) ? 3 : 0;
}

bool hop_geometry(int log2n, int *threads, size_t *lds_bytes) {
    if (log2n < 5 || log2n > 14) return false;
    const int m = log2n - 1, M = 1 << m;
    const int T = cmax(M / RC_PMAX, cmin(64, M / 4));
    if (threads) *threads = T;
    if (lds_bytes) *lds_bytes = sizeof(float2) * (size_t)(M + (M >> 5) + 1);
    return true;
}

hipError_t launch_hop(int log2n, HopMode mode, const HopParams &p, hipStream_t s) {
    switch (log2n) {
        case 5: return launch_hop_n<5>(mode, p, s);
        case 6: return launch_hop_n<6>(mode, p, s);
        case 7: return launch_hop_n<7>(mode, p, s);
        case 8: return launch_hop_n<8>(mode, p, s);
        case 9: return launch_hop_n<9>(mode, p, s);
        case 10: return launch_hop_n<10>(mode, p, s);
        case 11: return launch_hop_n<11>(mode, p, s);
        case 12: return launch_hop_n<12>(mode, p, s);
        case 13: return launch_hop_n<13>(mode, p, s);
        case 14: return launch_hop_n<14>(mode, p, s);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_big(int stage, const BigParams &p, hipStream_t s, HopMode mode) {
    if (p.log2n != 15 && p.log2n != 16) return hipErrorInvalidValue;
    if (stage == 1) {
        const uint32_t Ms = (1u << p.log2n) / 8;
        const dim3 grid((Ms / 2 + 1 + 255) / 256, (unsigned)p.hop_count, p.n_channels), block(256);
        if (mode == MODE_FORWARD) hipLaunchKernelGGL(big_b_kernel<MODE_FORWARD>, grid, block, 0, s, p);
        else if (mode == MODE_RESYNTH) hipLaunchKernelGGL(big_b_kernel<MODE_RESYNTH>, grid, block, 0, s, p);
        else hipLaunchKernelGGL(big_b_kernel<MODE_FUSED>, grid, block, 0, s, p);
        return hipGetLastError();
    }
    // quarter FFT of N/8 complex points == the passes of window length N/4
    return p.log2n == 15 ? launch_big_ac<13>(stage, p, s) : launch_big_ac<14>(stage, p, s);
}

hipError_t launch_big_cr(const BigOlaParams &p, hipStream_t s) {
    if (p.b.log2n != 15 && p.b.log2n != 16) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((p.runs + 7) / 8 * 32), p.b.n_channels);
    if (p.b.log2n == 15) {
        using G = Geo<13>;
        hipLaunchKernelGGL((big_cr_kernel<13>), grid, dim3(G::T), sizeof(float2) * G::LDS_FLOAT2, s, p);
    } else {
        using G = Geo<14>;
        hipLaunchKernelGGL((big_cr_kernel<14>), grid, dim3(G::T), sizeof(float2) * G::LDS_FLOAT2, s, p);
    }
    return hipGetLastError();
}

// Y = K(X) of the curated device kernels, one thread per bin (natural order, all N bins).
__global__ __launch_bounds__(256) void dev_kernel(const DevKernelParams p) {
    const uint32_t N = p.log2n ? 1u << p.log2n : p.n, M = N / 2;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t hop = blockIdx.y;
    if (j >= N) return;
    const float2 *x = p.in + hop * N;
    float2 y;
    if (p.kind == 2) {  // band mask, symmetric in frequency
        const uint32_t f = j <= M ? j : N - j;
        const float g = (f >= p.lo_bin && f <= p.hi_bin) ? p.gain_in : p.gain_out;
        y = make_float2(x[j].x * g, x[j].y * g);
    } else {            // shift by whole bins: Y[f] = X[f - s] on 0..M, Y[N - f] = conj(Y[f])
        const uint32_t f = j <= M ? j : N - j;
        const int64_t src = (int64_t)f - p.shift;
        y = make_float2(0.f, 0.f);
        if (src >= 0 && src <= (int64_t)M) {
            y = x[src];
            if (j > M) y.y = -y.y;
        }
    }
    p.out[hop * N + j] = y;
}
hipError_t launch_dev_kernel(const DevKernelParams &p, hipStream_t s) {
    const uint32_t N = p.log2n ? 1u << p.log2n : p.n;
    const uint64_t per_launch = 32768;  // grid.y limit
    for (uint64_t h0 = 0; h0 < p.hops_total; h0 += per_launch) {
        DevKernelParams q = p;
        q.in = p.in + h0 * N;
        q.out = p.out + h0 * N;
        q.hops_total = p.hops_total - h0 < per_launch ? p.hops_total - h0 : per_launch;
        hipLaunchKernelGGL(dev_kernel, dim3((N + 255) / 256, (unsigned)q.hops_total), dim3(256), 0, s, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int R>
hipError_t launch_big4_r(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(BIG4_T);
    const size_t lds = sizeof(float2) * (size_t)big4_lds_float2(R);
    const bool hann = p.hann_rot != nullptr;
    if (p.pitch == 1 && hann) hipLaunchKernelGGL((big4_kernel<R, true, true>), grid, block, lds, s, p);
    else if (p.pitch == 1) hipLaunchKernelGGL((big4_kernel<R, true, false>), grid, block, lds, s, p);
    else if (hann) hipLaunchKernelGGL((big4_kernel<R, false, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((big4_kernel<R, false, false>), grid, block, lds, s, p);
    return hipGetLastError();
}
hipError_t launch_big4(int log2n, const HopParams &p, hipStream_t s) {
    if (log2n == 15) return launch_big4_r<32>(p, s);
    if (log2n == 16) return launch_big4_r<64>(p, s);
    return hipErrorInvalidValue;
}

// ---- window lengths that are not a power of two: O(N^2) DFTs (rc_kernels.h, launch_gen) -----------
__global__ __launch_bounds__(256) void gen_fwd_kernel(const HopParams p) {
    const uint32_t N = p.n_generic;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hl = blockIdx.y;
    const uint32_t ch = blockIdx.z;
    const int64_t hop = p.hop_first + hl;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GF src = (hop >= p.tail_hop_first) ? xt + (hop * (int64_t)p.step - p.tail_origin)
                                       : xc + (hop * (int64_t)p.step - p.in_origin);
    GF win = (GF)p.window;
    GV2 tw = (GV2)p.tw_generic;
    if (k >= N) return;
    float ax = 0.f, ay = 0.f;
    uint32_t idx = 0;
    for (uint32_t n = 0; n < N; ++n) {
        const float a = src[n] * win[n];  // (src/fft.rs:51-55)
        const float2 w = ldg2(tw + idx);
        ax = fmaf(a, w.x, ax);
        ay = fmaf(a, w.y, ay);
        idx += k;
        if (idx >= N) idx -= N;
    }
    stg2((GV2W)p.spec + ((size_t)ch * p.hop_count + (size_t)hl) * N + k, make_float2(ax, ay));
}
__global__ __launch_bounds__(256) void gen_phase_kernel(const HopParams p) {
    const uint32_t N = p.n_generic, half = N / 2;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hl = blockIdx.y;
    const uint32_t ch = blockIdx.z;
    if (k >= N) return;
    const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, p.hop_first + hl);
    // frozen phase spec (rc_phase_theta): bins b < N/2 take the top 23 bits of hash(b), bins b + N/2 its low 16
    const bool upper = k >= half;
    const uint32_t h = phase_hash_x((upper ? k - half : k) * key.mul + key.k0);
    const float u = upper ? (float)(h & 0xFFFFu) * (1.0f / 65536.0f) : (float)(h >> 9) * (1.0f / 8388608.0f);
    const float th = u * 3.14159274101257324219f;
    GV2W z = (GV2W)p.spec + ((size_t)ch * p.hop_count + (size_t)hl) * N + k;
    const float2 X = ldg2((GV2)z);
    const float m = sqrtf(X.x * X.x + X.y * X.y);
    float sn, cs;
    sincosf(th, &sn, &cs);
    stg2(z, make_float2(m * cs, m * sn));  // src/fft.rs:65-68
}
__global__ __launch_bounds__(256) void gen_inv_kernel(const HopParams p) {
    const uint32_t N = p.n_generic;
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t hl = blockIdx.y;
    const uint32_t ch = blockIdx.z;
    if (n >= N) return;
    GV2 z = (GV2)p.spec + ((size_t)ch * p.hop_count + (size_t)hl) * N;
    GV2 tw = (GV2)p.tw_generic;
    float acc = 0.f;
    uint32_t idx = 0;
    for (uint32_t k = 0; k < N; ++k) {
        const float2 Z = ldg2(z + k), w = ldg2(tw + idx);
        acc = fmaf(Z.x, w.x, fmaf(Z.y, w.y, acc));  // Re(Z conj(w)), w = (cos, -sin)
        idx += n;
        if (idx >= N) idx -= N;
    }
    ((GFW)p.ybuf)[((size_t)ch * p.hop_count + (size_t)hl) * N + n] = acc / (float)N * ((GF)p.window)[n];  // fft.rs:70-73
}
hipError_t launch_gen(int stage, const HopParams &p, hipStream_t s) {
    const uint32_t N = p.n_generic;
    const int64_t per = 32768;  // grid.y limit
    for (int64_t h0 = 0; h0 < p.hop_count; h0 += per) {
        HopParams q = p;
        q.hop_first = p.hop_first + h0;
        q.hop_count = p.hop_count;  // (row stride of spec / ybuf)
        const unsigned ny = (unsigned)std::min<int64_t>(per, p.hop_count - h0);
        // hop index inside the chunk = blockIdx.y + h0: shift the bases instead of the index
        q.spec = p.spec ? p.spec + (size_t)h0 * N : nullptr;
        q.ybuf = p.ybuf ? p.ybuf + (size_t)h0 * N : nullptr;
        const dim3 grid((N + 255) / 256, ny, p.n_channels), block(256);
        if (stage == 0) hipLaunchKernelGGL(gen_fwd_kernel, grid, block, 0, s, q);
        else if (stage == 1) hipLaunchKernelGGL(gen_phase_kernel, grid, block, 0, s, q);
        else hipLaunchKernelGGL(gen_inv_kernel, grid, block, 0, s, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

__global__ __launch_bounds__(256) void prep_kernel(const PrepParams q) {
    if (q.run_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < RC_RUN_COUNTERS) q.run_counter[threadIdx.x] = 0;
    if (!q.xtail) return;
    const uint32_t ch = blockIdx.y;
    GF src = (GF)q.src + (size_t)ch * q.src_stride;
    GFW dst = (GFW)q.xtail + (size_t)ch * q.tail_len;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q.tail_len; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = i < q.real ? src[i] : 0.0f;
}
hipError_t launch_prep(const PrepParams &p, hipStream_t s) {
    if (!p.xtail && !p.run_counter) return hipSuccess;
    const unsigned bx = p.xtail ? (unsigned)std::min<size_t>(64, (p.tail_len + 255) / 256) : 1u;
    hipLaunchKernelGGL(prep_kernel, dim3(bx, p.xtail ? p.n_channels : 1u), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_ola(const OlaParams &p, hipStream_t s, bool tail_only) {
    const dim3 grid((unsigned)p.hop_count, p.n_channels), block(256);
    if (!tail_only) {
        hipLaunchKernelGGL(ola_kernel, grid, block, 0, s, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const dim3 g2(4, p.n_channels);
    hipLaunchKernelGGL(ola_save_tail_kernel, g2, block, 0, s, p);
    return hipGetLastError();
}

}  // namespace rc
