// Device-side helpers shared by every kernel translation unit of the rocoder stretch hot path (gfx950):
// address-space typed pointers, packed (re, im) helpers, the phase source, the per-bin pair algebra.
// Everything lives in an anonymous namespace: each translation unit gets its own (inlined) copy.
#pragma once
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
    // threads per workgroup (= per hop slot below N = 512, where 8 points per thread mean three passes instead of four)
    static constexpr int T = M <= 128 ? cmax(2, M / 8) : cmax(M / RC_PMAX, cmin(64, M / 4));
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
#pragma unroll
    for (int q = 0; q < G::P; ++q) lds[base + lds_reg_off<G::B, LO>(q)] = v[q];
}
template <class G, int LO>
__device__ __forceinline__ void lds_load(float2 (&v)[G::P], const float2 *lds, int base) {
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
#ifndef RC_MAD16
#define RC_MAD16 1
#endif
__device__ __forceinline__ float phase_rev_upper(uint32_t h) {
#if RC_MAD16
    // (h & 0xFFFF) * 128 + 0x3F000000 in one instruction: v_mad_u32_u16 multiplies the LOW HALVES of its first two
    // operands (no mask, no shift), the addend is the inline constant 0.5; the fields do not overlap, so + is |
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, 0.5" : "=v"(r) : "v"(h), "s"(128u));
    return __uint_as_float(r);
#else
    return __uint_as_float(0x3F000000u | ((h & 0xFFFFu) << 7));
#endif
}
// counter x = c * mul + k0 of c < M: (-cos, -sin) of bin c (lo*) and of bin c + M (up*)
__device__ __forceinline__ void phase_ncs2_x(uint32_t x, float &lo_nc, float &lo_ns, float &up_nc,
                                             float &up_ns) {
    const uint32_t h = phase_hash_x(x);
    const float fl = phase_rev_lower(h), fu = phase_rev_upper(h);
    lo_nc = __builtin_amdgcn_cosf(fl);
    lo_ns = __builtin_amdgcn_sinf(fl);
    up_nc = __builtin_amdgcn_cosf(fu);
    up_ns = __builtin_amdgcn_sinf(fu);
}
// The same two draws as HALF-angle operands for the folded pair stage (pair_regs_pk5, rc_dit.hpp): with r = theta / 2 pi,
//   g_lo = 0.25 + r(c) / 2        (the 23-bit draw under the exponent of [0.25, 0.5): exact)
//   g_up = 0.5  + r(c + M)        (the 16-bit draw under the exponent of [0.5, 1), as phase_rev_upper)
// so that fma(g_up, +-0.5, +-g_lo) is 0.5 + (r_lo + r_up) / 2 and (r_lo - r_up) / 2 in one instruction each.
__device__ __forceinline__ void phase_g2_x(uint32_t x, float &g_lo, float &g_up) {
    const uint32_t h = phase_hash_x(x);
    g_lo = __uint_as_float(0x3E800000u | (h >> 9));
    g_up = phase_rev_upper(h);
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

// a per-hop opaque copy of a table pointer: keeps the compiler from hoisting 2P table loads
// out of the hop loop (they are L1/L2 hits; 64+ live VGPRs would halve occupancy)
__device__ __forceinline__ GF per_hop(const float *ptr) {
    GF g = (GF)ptr;
    asm volatile("" : "+s"(g));
    return g;
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

// the same pointer per lane (kernels whose lanes work on different hops: two / four hops per wave, hop slots)
__device__ __forceinline__ GF hop_src_lane(const HopParams &p, GF xc, GF xt, int64_t k) {
    const int64_t off = (k >= p.tail_hop_first) ? (k * (int64_t)p.step - p.tail_origin) : (k * (int64_t)p.step - p.in_origin);
    return ((k >= p.tail_hop_first) ? xt : xc) + off;
}

constexpr int brev_c(int x, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}

}  // namespace
}  // namespace rc
