#include "rc_bigdit.hpp"

#include <type_traits>

namespace rc {
namespace {

// ======================= fused kernel for N = 65536 (BASELINE C5), round 5 ==========================
// big4_kernel<64>'s arithmetic - the same butterflies, twiddles, pair stage and epilogue, element for element - with
// hop4_kernel's exchange scheme carried over (tests/dev/proto_big5.py is the index model; every map below is
// checked there for who-gets-what and for bank conflicts: none):
//   * a wave is a RESIDUE CLASS: the low five residue bits of everything a wave holds between E1 and E4 lie in
//     {k, k + 16, 32 - k, 16 - k} (k = 1..7; class 0 = {0, 16, 24, 8}), a set closed under negation. The thread that
//     holds residue r holds RES - r (the (j, M - j) pair stage stays in registers, as in big4), and now both live in
//     the same wave as every element of the passes before and after: E2 (F2 -> F3) and E3 (I1 -> I2) never leave the
//     wave. They have no s_barrier - the LDS executes one wave's instructions in order, only the compiler needs a
//     fence - and run in the wave's own region of the buffer;
//   * the buffer is eight regions of 2111 float2 (2048 + the skew of the two wave-local transposes: rows of 32 at a
//     stride of 33, so that both sides address as base(lane) + constant(register) - immediate offsets, no VALU);
//   * E1 (F1 -> F2, cross-wave): round 0 is writer-major - a wave stores into its OWN region and everybody reads
//     everywhere - round 1 reader-major - stores go everywhere, a wave reads its OWN region: no barrier at the entry
//     (a wave's own region was last read by itself), three barriers instead of four;
//   * E4 (I2 -> I3, cross-wave): the same two-sided scheme. Its round is the I2 group P'9, so I2 runs the inverse
//     stages 4..9 (4..8 on each group of 32 registers, then stage 9 across the groups) and I3 the stages 10..13 on the
//     registers P'10..13 of each round (14 in the epilogue, as in big4). Three barriers.
//   Six barriers per hop instead of sixteen.
// Threads: F1 / I3 / epilogue: tid = sample order (coalesced loads and stores). Between E1 and E4: wave k = tid >> 6
// = class, lane = low4 | a << 4 with a = member index; F2: lf = member(k, a), uu = low4; F3 / I1: tau = member(k, a)
// | low4 << 5; I2: l4 = low4, class bits = brev5(member(k, a)).
constexpr int BIG5_T = 512, BIG5_R = 64;
constexpr int BIG5_REGION = 2111;              // float2 slots per wave region
constexpr int BIG5_XBUF = 8 * BIG5_REGION;
#ifndef BIG5_TL
#define BIG5_TL 5
#endif
#ifndef BIG5_K
#define BIG5_K 4
#endif
constexpr int BIG5_TAIL_LDS = BIG5_TL;         // tail pairs kept in LDS (of 32 per thread), as big4_kernel<64>
constexpr int BIG5_OVL_K = BIG5_K;             // VALU instructions per interleaved exchange store
constexpr int BIG5_TA = 1024;                  // W_M^r, r < RES / 2
constexpr int big5_lds_float2() { return BIG5_XBUF + BIG5_TA + 1 + 512 * BIG5_TAIL_LDS; }
static_assert(sizeof(float2) * big5_lds_float2() <= 160 * 1024, "big5 LDS budget");

constexpr int b5_member(int k, int a) {
    const int low4 = (a & 2) ? (k ? 16 - k : 8) : k;
    return ((a == 1 || a == 2) ? 16 : 0) | (low4 & 15);
}
constexpr int b5_class_k(int lf) {
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 4; ++a)
            if (b5_member(k, a) == lf) return k;
    return -1;
}
constexpr int b5_class_a(int lf) {
    for (int k = 0; k < 8; ++k)
        for (int a = 0; a < 4; ++a)
            if (b5_member(k, a) == lf) return a;
    return -1;
}
constexpr bool b5_classes_ok() {
    for (int lf = 0; lf < 32; ++lf) {
        const int k = b5_class_k(lf), a = b5_class_a(lf);
        if (k < 0 || b5_member(k, a) != lf) return false;
        if (b5_class_k((32 - lf) & 31) != k) return false;  // closed under negation
        if ((lf & 1) != (k & 1)) return false;              // one parity per class
    }
    return true;
}
static_assert(b5_classes_ok(), "residue classes");
struct B5Tab {
    int k[32], a[32];
};
constexpr B5Tab make_b5tab() {
    B5Tab t{};
    for (int lf = 0; lf < 32; ++lf) {
        t.k[lf] = b5_class_k(lf);
        t.a[lf] = b5_class_a(lf);
    }
    return t;
}
__device__ constexpr B5Tab B5 = make_b5tab();

#ifndef BIG5_ST_AUX
#define BIG5_ST_AUX 2  // cache policy of the output stores: 2 = nt; 16 = sc1, 17 = sc0 sc1 (write-through: A/B, round 6)
#endif
#define BIG5_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// wave-local exchanges: the LDS runs one wave's instructions in order; the compiler must not move a load over a store
#define BIG5_FENCE() asm volatile("" ::: "memory")

// SEAM (round 6, pitch 1): hop4_kernel's run hand-over at 64 Ki. Runs are SHORT (16 hops) and handed out per XCD in the
// order workgroups start (HW_REG_XCC_ID tickets), so the 32 workgroups resident on an XCD walk 32 ADJACENT runs: their
// 256 KB windows lie within 2.3 MB of input, which stays in the XCD's 4 MB L2 - every input byte comes over the fabric
// about once instead of once per hop (round 5: 10.7 GB of window reads per C5 launch, L2 hit rate 30 %). A run does not
// recompute the hop before it: run g > 0 stashes the windowed head of its first hop (write-through stores + flag) and run
// g - 1 adds its last tail to it at its end (sc1 loads), as in rc_hop16k.hip.
template <bool PITCH1, bool HANN, bool SEAM = false>
__global__ __launch_bounds__(BIG5_T, 2) void big5_kernel(const HopParams p) {
    static_assert(!SEAM || PITCH1, "the seam hand-over stores whole heads: pitch 1");
    constexpr int R = BIG5_R, b = 6, m = 15, LOG2N = 16, M = 1 << m, H = M, T = BIG5_T;
    constexpr int RES = 2048, NS = 4, PH = R / 2, RG = BIG5_REGION;
    constexpr int T_A = BIG5_XBUF, SCR = T_A + BIG5_TA;
    constexpr int TL = BIG5_TAIL_LDS, TLB = SCR + 1, PHR = R / 2 - TL;  // tail pairs [PHR, R/2) live at lds[TLB + ...]
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int tid = threadIdx.x;
    uint32_t gr = blockIdx.x;
    if constexpr (SEAM) {
        // the ticket: XCD x walks the x-th eighth of the runs; an XCD that runs out takes from the next one's counter
        // (rc_hop16k.hip). Read back through readfirstlane: run, channel and hop counter then live in SGPRs.
        unsigned *slot = reinterpret_cast<unsigned *>(lds);
        if (tid == 0) {
            const uint32_t total = p.runs_per_channel * p.n_channels;
            unsigned got = 0xFFFFFFFFu;
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
            *slot = got;
        }
        __syncthreads();
        gr = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile unsigned *>(slot));
        __syncthreads();
        if (gr == 0xFFFFFFFFu) return;  // (more workgroups than runs: cannot happen with the engine's grid)
    }
    const uint32_t run = gr % p.runs_per_channel;
    const uint32_t ch = gr / p.runs_per_channel;
    const int64_t k_begin = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if (k_begin >= k_end) return;
    const bool stash_first = SEAM && run > 0;
    const bool has_next = SEAM && run + 1 < p.runs_per_channel;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = residue class (scalar: uniform branches below)
        {
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M)
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N)
        for (int i = tid; i < BIG5_TA; i += T) lds[T_A + i] = ldg2(wt + i);
        if (tid == 0) {  // W_N^(RES/2 - M/2) = i W_N^(RES/2): thread 0's twiddle base for its second residue
            const float2 wq = ldg2(rt + RES / 2);
            lds[SCR] = make_float2(-wq.y, wq.x);
        }
        __syncthreads();
    }
    v2f tail[PHR];
#pragma unroll
    for (int q = 0; q < PHR; ++q) tail[q] = v2f{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < TL; ++q) lds[TLB + 512 * q + tid] = make_float2(0.f, 0.f);  // (read back by this thread only)
    const bool is0 = tid == 0;  // tau == 0: class 0, member 0, low4 0
    Stamps stp;
    stp.init();
    // Thread identities are re-derived from an opaque copy of the thread id inside each phase (hoisted out of the hop
    // loop they are ~40 live values, and the allocator spills them)
    auto ptid = [&]() {
        int t = tid;
        opaque(t);
        return t;
    };
    struct Cls {
        int lane, a, low4, mem;
    };
    // member(k, a) of this thread's class
    auto cls = [&]() {
        const int t = ptid();
        Cls c;
        c.lane = t & 63;
        c.a = c.lane >> 4;
        c.low4 = c.lane & 15;
        const int k4 = wv ? 16 - wv : 8;
        c.mem = (((c.a ^ (c.a >> 1)) & 1) << 4) | ((c.a & 2) ? k4 : wv);
        return c;
    };
    // (a 32-bit trip count: the scalar unit has no ordered 64-bit compare, so `k < k_end` on int64 ran on the VALU with
    // k_end parked in a VGPR pair - this kernel's one scratch reload per hop)
    const int64_t k_first = (k_begin > 0 && !stash_first) ? k_begin - 1 : k_begin;
    const int n_it = (int)(k_end - k_first);
    for (int it = 0; it < n_it; ++it) {
        const int64_t k = k_first + it;
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        int tt = tid;  // per-hop opaque copy for the epilogue's addresses
        opaque(tt);
        v2f v[R];
        {   // register brev6(q) := z[q * T + t] * window, stage 0 inside the load loop (big4_kernel<64>'s pipeline)
            // hop_src with `k >= tail_hop_first` as the sign of a 64-bit difference (k >= 0, tail_hop_first a hop index or
            // INT64_MAX: no overflow): the scalar unit has no ordered 64-bit compare, the VALU form kept tail_hop_first in a
            // VGPR pair that was parked in scratch and reloaded every hop
            GF src;
            {
                const int dhi = __builtin_amdgcn_readfirstlane((int)((uint64_t)(k - p.tail_hop_first) >> 32));  // (opaque to the
                const bool past = dhi >= 0;                       // optimiser, which would fold the sign test back into the compare)
                const int64_t off = past ? (k * (int64_t)p.step - p.tail_origin) : (k * (int64_t)p.step - p.in_origin);
                const unsigned long long sa = (unsigned long long)(past ? xt : xc) + (unsigned long long)off * 4ull;
                const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sa);
                const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(sa >> 32));
                src = (GF)(((unsigned long long)hi << 32) | lo);
            }
            GF win = per_hop(p.window);
            v2f cbW = {0.f, 0.f}, sbW = cbW;
            if constexpr (HANN) {
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tt;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
            }
            const HannK64 &HW = HANN_W16;
#define ROW(i) (((i) >> 1) + ((i) & 1) * (R / 2))
            if constexpr (HANN) {
                constexpr int PB = BIG4_PIPE_ROWS, NB = R / PB;
                float xp0[2][PB], xp1[2][PB];
                // buffer loads: the hop's base in a resource descriptor, ONE 32-bit lane offset for all rows, the row in the
                // scalar offset - the global_load form spent 128 v_add_co / v_addc per hop on 64-bit lane addresses (hop4: -1 %)
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x40000000, 0x00020000);
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
                auto issue = [&](int i, float (&x0)[PB], float (&x1)[PB]) {
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * lane2), 4 * 2 * T * ROW(i * PB + q), 0);
                        x0[q] = __uint_as_float(x.x);
                        x1[q] = __uint_as_float(x.y);
                    }
                };
                issue(0, xp0[0], xp1[0]);
                issue(1, xp0[1], xp1[1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NB; ++i) {
#pragma unroll
                    for (int q = 0; q < PB; q += 2) {
                        const int r0 = ROW(i * PB + q), r1 = ROW(i * PB + q + 1);
                        const v2f w0 = __builtin_elementwise_fma(v2f{HW.s[r0], HW.s[r0]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[r0], HW.c[r0]}, cbW, v2f{0.5f, 0.5f}));
                        const v2f w1 = __builtin_elementwise_fma(v2f{HW.s[r1], HW.s[r1]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[r1], HW.c[r1]}, cbW, v2f{0.5f, 0.5f}));
                        const v2f a = v2f{xp0[i & 1][q], xp1[i & 1][q]} * w0;
                        const v2f xh = v2f{xp0[i & 1][q + 1], xp1[i & 1][q + 1]};
                        v[brev_c(r0, b)] = __builtin_elementwise_fma(xh, w1, a);
                        v[brev_c(r0, b) + 1] = __builtin_elementwise_fma(-xh, w1, a);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (i + 2 < NB) {
                        issue(i + 2, xp0[i & 1], xp1[i & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
                constexpr int LB = 16;
#pragma unroll
                for (int q0 = 0; q0 < R; q0 += LB) {
                    float xr0[LB], xr1[LB], wr0[LB], wr1[LB];
#pragma unroll
                    for (int q = 0; q < LB; ++q) {
                        xr0[q] = (src + 2 * T * ROW(q0 + q))[lane2];
                        xr1[q] = (src + 2 * T * ROW(q0 + q))[lane2 + 1];
                        wr0[q] = (win + 2 * T * ROW(q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * ROW(q0 + q))[lane2 + 1];
                    }
#pragma unroll
                    for (int q = 0; q < LB; q += 2) {
                        const v2f a = v2f{xr0[q], xr1[q]} * v2f{wr0[q], wr1[q]}, xh = v2f{xr0[q + 1], xr1[q + 1]};
                        v[brev_c(ROW(q0 + q), b)] = __builtin_elementwise_fma(xh, v2f{wr0[q + 1], wr1[q + 1]}, a);
                        v[brev_c(ROW(q0 + q), b) + 1] = __builtin_elementwise_fma(-xh, v2f{wr0[q + 1], wr1[q + 1]}, a);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#undef ROW
            stp.mark(0);
            dit_g<R, 1, b - 1, 0, false, false>(v);
        }
        stp.mark(1);
        // ---- E1: F1 -> F2 (cross-wave). Element P: P0..4 = register q, P5 = round, P6..14 = brev9(tid)
        v2f w[R];
        {
            const int t_ = ptid();
            const int bs = (int)(__brev((unsigned)t_) >> 23);  // brev9(t): bit i = P(6 + i)
            // round 0, writer-major: own region, index uu | (P4 | P0..3 << 1 | P9 << 5 | P10 << 6) << 4
            const int w0b = RG * (t_ >> 6) + (bs >> 5) + (((bs >> 3) & 3) << 9);
            // round 1, reader-major: region = class of q, index uu | a << 4 | (P6..10) << 6
            const int w1b = (bs >> 5) + ((bs & 31) << 6);
            const Cls c = cls();
            const int r0b = c.low4 + 16 * ((c.mem >> 4) | ((c.mem & 15) << 1));
            const int r1b = RG * wv + c.lane;
            // (no barrier here: a wave's own region was last read by the wave itself - E4's second round of the hop before)
#pragma unroll
            for (int q = 0; q < 32; ++q) lds[w0b + 16 * ((q >> 4) | ((q & 15) << 1))] = to_f2(v[q]);
            BIG5_BAR();
#pragma unroll
            for (int j = 0; j < 32; ++j) w[j] = to_v(lds[r0b + RG * brev_c(j & 7, 3) + 512 * (j >> 3)]);
            const v2f wf0 = to_v(lds[T_A + 16 * c.mem]);
            BIG5_BAR();
            __builtin_amdgcn_sched_barrier(0);
            // round 1's stores are issued one at a time between the butterflies of F2 on round 0's group
#pragma unroll
            for (int q = 0; q < 32; ++q) lds[w1b + RG * B5.k[q] + 16 * B5.a[q]] = to_f2(v[32 + q]);
            {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = w[j];
                dit_g<32, b, b + 4, b, false, true, true>(grp, wf0);
#pragma unroll
                for (int j = 0; j < 32; ++j) w[j] = grp[j];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // one DS write
                __builtin_amdgcn_sched_group_barrier(0x2, BIG5_OVL_K, 0);  // then K VALU
            }
            __builtin_amdgcn_sched_barrier(0);
            BIG5_BAR();
#pragma unroll
            for (int j = 0; j < 32; ++j) w[32 + j] = to_v(lds[r1b + 64 * j]);
            stp.mark(2);
            {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = w[32 + j];
                dit_g<32, b, b + 4, b, false, true, true>(grp, vcmul(wf0, v2f{W64.re[1], W64.im[1]}));
#pragma unroll
                for (int j = 0; j < 32; ++j) w[32 + j] = grp[j];
            }
        }
        stp.mark(3);
        // ---- E2: F2 -> F3, inside the wave. Element = (residue r, uu); round = residue bit 10. In the region:
        //   x + 16 (a & 1) + 33 uu + 528 ((a >> 1) | gp << 1),  x = residue bits 5..8, gp = bit 9, a = member index
        v2f st[NS][16];
        float2 wrg[NS / 2];  // W_N^r of this thread's residues, requested here for the pair stage
        // base of set s of this thread in its wave's region, for the E2 loads (skew = 1: x + 33 uu) and the E3 stores
        // (skew = 33: 33 x + q'): set s = (gp = s >> 1, partner = s & 1); the partner residue RES - r (mod 1024) has
        // member index an and bits 5..9 = 31 - y (member 0: -y)
        auto set_base = [&](const Cls &c, int s, int skew) {
            const int an = wv ? (c.a ^ 2) : (c.a < 2 ? c.a : (c.a ^ 1));
            const int y = c.low4 | ((s >> 1) << 4);
            const int ye = (s & 1) ? (c.mem ? 31 - y : ((32 - y) & 31)) : y;
            const int ae = (s & 1) ? an : c.a;
            return RG * wv + 16 * (ae & 1) + 528 * ((ae >> 1) | ((ye >> 4) << 1)) + skew * (ye & 15);
        };
        {
            const Cls c = cls();
            GV2 rt = (GV2)per_hop(reinterpret_cast<const float *>(p.rtab)) + (c.mem | (c.low4 << 5));
#pragma unroll
            for (int gp = 0; gp < NS / 2; ++gp) wrg[gp] = ldg2(rt + 512 * gp);
            const int e2w = RG * wv + 16 * (c.a & 1) + 528 * (c.a >> 1) + 33 * c.low4;
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                BIG5_FENCE();
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    const int g = kk >> 4, jl = kk & 15;  // y = g | jl << 1: x = g | (jl & 7) << 1, gp = jl >> 3
                    lds[e2w + (g | ((jl & 7) << 1)) + 1056 * (jl >> 3)] = to_f2(w[32 * g + 16 * rnd + jl]);
                }
                BIG5_FENCE();
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if ((s & 1) != rnd) continue;
                    const int e2r = set_base(c, s, 1);
#pragma unroll
                    for (int q = 0; q < 16; ++q) st[s][q] = to_v(lds[e2r + 33 * q]);
                }
            }
        }
        stp.mark(4);
        // ---- F3 on every set, middle stage on every (A, B) pair of sets, I1 (big4_kernel's, with r = tau + 512 gp)
#pragma unroll
        for (int gp = 0; gp < NS / 2; ++gp) {
            const Cls cr = cls();
            const int r = (cr.mem | (cr.low4 << 5)) + 512 * gp;  // tau + 512 gp
            v2f(&va)[16] = st[2 * gp];
            v2f(&vb)[16] = st[2 * gp + 1];
            {
                const v2f wa = to_v(lds[T_A + r]);  // W_M^r
                const v2f k16 = {W32_RE[2], W32_IM[2]};
                v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
                if (gp == 0 && is0) wb = v2f{W32_RE[1], W32_IM[1]};  // thread 0: residue RES/2 -> W_32
                dit_g<16, b + 5, b + 8, b + 5, false, true, true>(va, wa);
                dit_g<16, b + 5, b + 8, b + 5, false, true, true>(vb, wb);
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
                const float2 wrl = wrg[gp];
                const float2 w0 = lds[SCR];
                const float2 wrh = make_float2(sp ? w0.x : wrl.x, sp ? w0.y : wrl.y);
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
        // ---- E3: I1 -> I2, inside the wave. Element P' = q' | brev11(residue) << 4; in the region:
        //   q' + 16 (a & 1) + 33 x + 528 ((a >> 1) | gp << 1).  I2 registers: j = P'4..8 (P'4 = the round = residue bit
        //   10, P'5..8 = residue bits 9, 8, 7, 6), group = P'9 = residue bit 5; lane = (l4 = q', a)
        {
            const Cls c = cls();
            const int e3r = RG * wv + c.low4 + 16 * (c.a & 1) + 528 * (c.a >> 1);
#pragma unroll
            for (int rnd = 0; rnd < 2; ++rnd) {
                BIG5_FENCE();
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if ((s & 1) != rnd) continue;
                    const int e3w = set_base(c, s, 33);
#pragma unroll
                    for (int q = 0; q < 16; ++q) lds[e3w + q] = to_f2(st[s][q]);
                }
                BIG5_FENCE();
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    const int grp = kk >> 4, jl = kk & 15;
                    const int y = grp | (brev_c(jl, 4) << 1);  // residue bits 5..9
                    v[32 * grp + 2 * jl + rnd] = to_v(lds[e3r + 33 * (y & 15) + 1056 * (y >> 4)]);
                }
            }
        }
        stp.mark(6);
        // ---- I2: stages 4..8 on each group of 32 registers (j = P'4..8), then stage 9 across the groups (P'9): the round
        // of E4 is then the group, every thread stores 32 registers per round and v[0..31] are dead after the first
        // (with I2 = 4..8 the round would be P'14 = the parity of the class: half of the waves storing 64 registers per
        // round and the other half keeping theirs live through I3's first half - that spilled).
        // E4: I2 -> I3 (cross-wave), as E1: round 0 writer-major into the wave's own region (index l4 | j << 4 | a << 9),
        // round 1 reader-major (region P'6..8, index P'0..5 | (P'10..14) << 6): no barrier at the entry, three in all.
        // I3: stages 10..13 on the registers P'10..13 of each round (P'9 = the round, P'14 look on); 14 in the epilogue.
        v2f y[R];
        {
            const Cls c = cls();
            const v2f wf = to_v(lds[T_A + c.low4 * R]);        // stage 8's base W_512^l4 = W_M^(64 l4)
            const v2f w9 = to_v(lds[T_A + 32 * c.low4]);       // stage 9's base W_1024^l4 = W_M^(32 l4)
            const int bm = (int)(__brev((unsigned)c.mem) >> 27);  // brev5(member) = P'10..14
            const int w0b = RG * wv + c.low4 + (c.a << 9);
            const int w1b = c.low4 + 64 * bm;
            const int t_ = ptid();
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = v[32 * g + j];
                dit_g<32, 4, 8, 4, true, true, true>(grp, wf);
#pragma unroll
                for (int j = 0; j < 32; ++j) v[32 * g + j] = grp[j];
            }
            __builtin_amdgcn_sched_barrier(0);
            // round 0's stores are issued one at a time between the butterflies of stage 9 that produce them
            lean_stage<64, 5, true>(v, w9);
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[w0b + (j << 4)] = to_f2(v[j]);
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x2, BIG5_OVL_K, 0);  // K VALU (the butterfly that makes v[j] final)
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // then its DS write
            }
            __builtin_amdgcn_sched_barrier(0);
            stp.mark(7);
            BIG5_BAR();
            // round 0 loads: register r5 = P'10..14 sits in the region of the class of brev5(r5); y[q + 32 h], q = P'9..13
#pragma unroll
            for (int r5 = 0; r5 < 32; ++r5) {
                const int lf = brev_c(r5, 5);
                y[((r5 & 15) << 1) + 32 * (r5 >> 4)] = to_v(lds[t_ + RG * B5.k[lf] + (B5.a[lf] << 9)]);
            }
            const v2f wf2 = vcsq(to_v(lds[T_A + t_]));  // stage 13's base W_M^(2 tid) (round 1: times W_32)
            BIG5_BAR();
            __builtin_amdgcn_sched_barrier(0);
            // round 1's stores between the butterflies of I3 on round 0's registers
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[w1b + RG * (j >> 2) + 16 * (j & 3)] = to_f2(v[32 + j]);
            {
                v2f grp[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) grp[i] = y[((i & 15) << 1) + 32 * (i >> 4)];
                dit_g<32, 10, 13, 10, true, true, true>(grp, wf2);
#pragma unroll
                for (int i = 0; i < 32; ++i) y[((i & 15) << 1) + 32 * (i >> 4)] = grp[i];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, BIG5_OVL_K, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            BIG5_BAR();
#pragma unroll
            for (int r5 = 0; r5 < 32; ++r5)
                y[1 + ((r5 & 15) << 1) + 32 * (r5 >> 4)] = to_v(lds[RG * (t_ >> 6) + (t_ & 63) + 64 * r5]);
            stp.mark(8);
            {
                v2f grp[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) grp[i] = y[1 + ((i & 15) << 1) + 32 * (i >> 4)];
                dit_g<32, 10, 13, 10, true, true, true>(grp, vcmul(wf2, v2f{W32_RE[1], W32_IM[1]}));
#pragma unroll
                for (int i = 0; i < 32; ++i) y[1 + ((i & 15) << 1) + 32 * (i >> 4)] = grp[i];
            }
        }
        stp.mark(9);
        // ---- epilogue: inverse stage 14, synthesis window, overlap-add, store (big4_kernel<64>'s).
        // STASH (SEAM, the first hop of a run g > 0): the windowed head goes to the seam stash as it is (write-through
        // stores: the run before adds its last tail to it, maybe from another XCD), the tail is kept as always.
        auto epilogue = [&](auto stash_tag) {
            constexpr bool STASH = decltype(stash_tag)::value;
            GF win = per_hop(p.window);
            GF esrc = per_hop(p.env);
            v2f cbW = {0.f, 0.f}, sbW = cbW, cbE = cbW, sbE = cbW;
            if constexpr (HANN) {
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tt;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
                if constexpr (!STASH) {
                    const float2 e0r = ldg2(hr + 2 * T), e1r = ldg2(hr + 2 * T + 1);
                    cbE = v2f{e0r.x, e1r.x};
                    sbE = v2f{e0r.y, e1r.y};
                }
            }
            const HannK64 &HW = HANN_W16;
            const HannK64 &HE = HANN_E16;
            const v2f hf = {0.5f, 0.5f};
            // pair_regs_pk4 leaves the -1/(4N) of the magnitudes out (a power of two): it rides on the amplitude
            const float ak = p.amp * (-0.25f / (float)(1 << LOG2N));
            const v2f ampk = {ak, ak};
            v2f hfE = hf;
            if constexpr (HANN) {  // env[i] * amp = amp/2 + c_q (amp cb) + s_q (amp sb): the amplitude rides on the rotation
                cbE *= ampk;
                sbE *= ampk;
                hfE = hf * ampk;
            }
            const int64_t g0 = k * (int64_t)H;
            GFW dst = STASH ? (GFW)p.seam_head + (size_t)gr * H : outc + (g0 / (int64_t)pitch - p.out_origin);
            const uint32_t kr = (uint32_t)(g0 % pitch);
            const unsigned long long da = (unsigned long long)dst;  // (uniform: descriptor in SGPRs)
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(da >> 32)) << 32) |
                         (unsigned)__builtin_amdgcn_readfirstlane((int)da)), 0, 0x40000000, 0x00020000);
            constexpr int EB = (HANN && PITCH1) ? BIG4_EPI_BATCH : 4;
#pragma unroll
            for (int q0 = 0; q0 < PH; q0 += EB) {
                float wr0[EB], wr1[EB], wt0[EB], wt1[EB], e0[EB], e1[EB];
                v2f tq[EB];
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    if constexpr (!HANN) {
                        wr0[q] = (win + 2 * T * (q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * (q0 + q))[lane2 + 1];
                        wt0[q] = (win + 2 * T * (q0 + q + PH))[lane2];
                        wt1[q] = (win + 2 * T * (q0 + q + PH))[lane2 + 1];
                        if constexpr (!STASH) {
                            e0[q] = (esrc + 2 * T * (q0 + q))[lane2];
                            e1[q] = (esrc + 2 * T * (q0 + q))[lane2 + 1];
                        }
                    } else {
                        const v2f wh = __builtin_elementwise_fma(v2f{HW.s[q0 + q], HW.s[q0 + q]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[q0 + q], HW.c[q0 + q]}, cbW, hf));
                        const v2f wt = __builtin_elementwise_fma(v2f{HW.s[q0 + q + PH], HW.s[q0 + q + PH]}, sbW,
                                       __builtin_elementwise_fma(v2f{HW.c[q0 + q + PH], HW.c[q0 + q + PH]}, cbW, hf));
                        wr0[q] = wh.x, wr1[q] = wh.y, wt0[q] = wt.x, wt1[q] = wt.y;
                        if constexpr (!STASH) {
                            const v2f ev = __builtin_elementwise_fma(v2f{HE.s[q0 + q], HE.s[q0 + q]}, sbE,
                                           __builtin_elementwise_fma(v2f{HE.c[q0 + q], HE.c[q0 + q]}, cbE, hfE));
                            e0[q] = ev.x, e1[q] = ev.y;
                        }
                    }
                    if constexpr (!STASH) {
                        if (q0 + q < PHR) tq[q] = tail[q0 + q];
                        else tq[q] = to_v(lds[TLB + 512 * (q0 + q - PHR) + tt]);
                    }
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    v2f yh = y[q0 + q], yt = y[q0 + q + PH];
                    {   // inverse stage m-1: (yh, yt) = (a + conj(w) b, a - conj(w) b), w = W_M^tid W_64^c
                        const int c = q0 + q;  // < 32
                        const v2f wfl = to_v(lds[T_A + tt]);
                        const v2f k64 = {W64.re[c & 15], W64.im[c & 15]};
                        const v2f tw = (c & 15) == 0 ? wfl : vcmul(wfl, k64);
                        const v2f a = yh, bb = yt;
                        if (c < 16) vdit_m<true>(a, bb, tw, yh, yt);
                        else vdit_rot_m<true>(a, bb, tw, yh, yt);
                    }
                    v2f head = yh * v2f{wr0[q], wr1[q]};
                    // (the windowed head is ROUNDED before the tail is added, as the reference's Vec<f32> is
                    // (src/fft.rs:72-73, src/stretcher.rs:97-100) and as a head that went through the seam stash is: with
                    // -ffp-contract=fast the compiler made it fma(yh, w, tail), one rounding, and a run seam differed
                    // from the same hop inside a run by an ulp)
                    asm volatile("" : "+v"(head));
                    const v2f nt = yt * v2f{wt0[q], wt1[q]};
                    typedef unsigned v2u __attribute__((ext_vector_type(2)));
                    if constexpr (STASH) {  // (16 = sc1: write-through, as the atomic stores of rc_hop16k.hip's stash)
                        __builtin_amdgcn_raw_buffer_store_b64(v2u{__float_as_uint(head.x), __float_as_uint(head.y)}, rd,
                                                              (int)(4u * lane2), 4 * 2 * T * (q0 + q), 16);
                    } else if (k >= k_begin) {
                        // stretcher.rs:97-100; with the computed envelope the amplitude is already inside it
                        const v2f o = HANN ? (head + tq[q]) * v2f{e0[q], e1[q]} : (head + tq[q]) * v2f{e0[q], e1[q]} * ampk;
                        if constexpr (PITCH1) {
                            // (buffer store, aux 2 = nt, as the loads)
                            __builtin_amdgcn_raw_buffer_store_b64(v2u{__float_as_uint(o.x), __float_as_uint(o.y)}, rd,
                                                                  (int)(4u * lane2), 4 * 2 * T * (q0 + q), BIG5_ST_AUX);
                        } else {
                            const uint32_t a0 = kr + 2u * (uint32_t)(tid + T * (q0 + q)), a1 = a0 + 1;
                            const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                            if (d0 * pitch == a0) dst[d0] = o.x;
                            if (d1 * pitch == a1) dst[d1] = o.y;
                        }
                    }
                    if (q0 + q < PHR) tail[q0 + q] = nt;
                    else lds[TLB + 512 * (q0 + q - PHR) + tt] = to_f2(nt);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (SEAM && stash_first && it == 0) {
            epilogue(std::true_type{});
            // every storing wave drains its own write-through stores before the barrier; only then may lane 0 publish
            // (MI355X_MICROARCH.md, valid hand-off forms)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0 && !(p.diag_flags & RC_DIAG_SKIP_SEAM_PUBLISH))
                __hip_atomic_store(p.seam_flag + gr, p.seam_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            epilogue(std::false_type{});
        }
        stp.mark(10);
    }
#if RC_STAMP
    if ((tid & 63) == 0 && p.spec) {
        unsigned *dbg = (unsigned *)p.spec + ((size_t)blockIdx.x * (T / 64) + (tid >> 6)) * 32;
        for (int i = 0; i < 32; ++i) dbg[i] = stp.acc[i];
    }
#endif
    if constexpr (SEAM) {
        if (!has_next) return;
        // the run's seam: wait for the head the next run stashed at its start (it started long ago: it is the next ticket
        // of this XCD, or the first run of the next eighth), add this run's last tail, envelope, store hop k_end's head
        unsigned *okw = reinterpret_cast<unsigned *>(lds);  // (the exchange buffer: no wave reads it after its last E4 loads)
        if (tid == 0) {
            unsigned ok = 0;
            for (unsigned spin = 0; spin < p.seam_spin_limit; ++spin) {
                if (__hip_atomic_load(p.seam_flag + gr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.seam_epoch) {
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
        int tt = tid;
        opaque(tt);
        const unsigned long long sa = (unsigned long long)(p.seam_head + (size_t)(gr + 1) * H);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(sa >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)sa)), 0, 0x40000000, 0x00020000);
        const unsigned long long da = (unsigned long long)(outc + (k_end * (int64_t)H - p.out_origin));
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(da >> 32)) << 32) |
                     (unsigned)__builtin_amdgcn_readfirstlane((int)da)), 0, 0x40000000, 0x00020000);
        const float ak = p.amp * (-0.25f / (float)(1 << LOG2N));
        const v2f ampk = {ak, ak};
        v2f cbE = {0.f, 0.f}, sbE = cbE, hfE = v2f{0.5f, 0.5f} * ampk;
        if constexpr (HANN) {
            GV2 hr = (GV2)p.hann_rot + 2 * tt;
            const float2 e0r = ldg2(hr + 2 * T), e1r = ldg2(hr + 2 * T + 1);
            cbE = v2f{e0r.x, e1r.x} * ampk;
            sbE = v2f{e0r.y, e1r.y} * ampk;
        }
        const HannK64 &HE = HANN_E16;
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        constexpr int SB = 8;
#pragma unroll
        for (int q0 = 0; q0 < PH; q0 += SB) {
            v2u hd[SB];
#pragma unroll
            for (int q = 0; q < SB; ++q)  // (sc1 loads: every load of the handed-over bytes bypasses this CU's L1)
                hd[q] = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * lane2), 4 * 2 * T * (q0 + q), 16);
#pragma unroll
            for (int q = 0; q < SB; ++q) {
                const v2f head = {__uint_as_float(hd[q].x), __uint_as_float(hd[q].y)};
                const v2f tq = (q0 + q < PHR) ? tail[q0 + q] : to_v(lds[TLB + 512 * (q0 + q - PHR) + tt]);
                v2f o;
                if constexpr (HANN) {
                    const v2f ev = __builtin_elementwise_fma(v2f{HE.s[q0 + q], HE.s[q0 + q]}, sbE,
                                   __builtin_elementwise_fma(v2f{HE.c[q0 + q], HE.c[q0 + q]}, cbE, hfE));
                    o = (head + tq) * ev;
                } else {
                    GF esrc = (GF)p.env;
                    o = (head + tq) * v2f{(esrc + 2 * T * (q0 + q))[lane2], (esrc + 2 * T * (q0 + q))[lane2 + 1]} * ampk;
                }
                __builtin_amdgcn_raw_buffer_store_b64(v2u{__float_as_uint(o.x), __float_as_uint(o.y)}, rd, (int)(4u * lane2),
                                                      4 * 2 * T * (q0 + q), BIG5_ST_AUX);
            }
        }
    }
}


// ======================= N = 32768: the same scheme with single-round exchanges ==========================
// big4_kernel<32>'s arithmetic (M = 16384 points, 32 per thread: F1 stages 0..4, F2 5..9, F3 10..13 on two sets of 16,
// I1 0..3, I2 4..8, I3 9..13) under big5's thread mapping: a wave is a residue class, E2 and E3 run inside the wave
// without barriers in its own region. Every exchange moves all 32 registers at once, so the cross-wave ones cannot have
// a writer-major AND a reader-major round; E1 is writer-major (own region, no barrier at its entry: the region was last
// read by the wave itself in E4), E4 reader-major (a wave reads its own region: the next hop's E1 needs no entry
// barrier): store / BAR / load / BAR and BAR / store / BAR / load - four barriers per hop instead of big4's eight.
// The index maps are big5's round-0 (E1), wave-local (E2, E3: gp = register bit 4) and round-1 (E4) maps.
constexpr int big5s_lds_float2() { return BIG5_XBUF + 512 + 1; }
template <bool PITCH1, bool HANN>
__global__ __launch_bounds__(BIG5_T, 2) void big5s_kernel(const HopParams p) {
    constexpr int R = 32, b = 5, m = 14, LOG2N = 15, M = 1 << m, H = M, T = BIG5_T;
    constexpr int RES = 1024, PH = R / 2, RG = BIG5_REGION;
    constexpr int T_A = BIG5_XBUF, SCR = T_A + 512;
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
    const uint32_t pitch = PITCH1 ? 1u : p.pitch;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = residue class
    {
        GV2 wt = (GV2)p.wtab;  // exp(-2 pi i k / M)
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N)
        lds[T_A + tid] = ldg2(wt + tid);
        if (tid == 0) {  // W_N^(RES/2 - M/2) = i W_N^(RES/2): thread 0's twiddle base for its second residue
            const float2 wq = ldg2(rt + RES / 2);
            lds[SCR] = make_float2(-wq.y, wq.x);
        }
        __syncthreads();
    }
    v2f tail[PH];
#pragma unroll
    for (int q = 0; q < PH; ++q) tail[q] = v2f{0.f, 0.f};
    const bool is0 = tid == 0;
    struct Cls {
        int lane, a, low4, mem;
    };
    auto cls = [&]() {
        const int t = tid;
        Cls c;
        c.lane = t & 63;
        c.a = c.lane >> 4;
        c.low4 = c.lane & 15;
        const int k4 = wv ? 16 - wv : 8;
        c.mem = (((c.a ^ (c.a >> 1)) & 1) << 4) | ((c.a & 2) ? k4 : wv);
        return c;
    };
    const Cls c = cls();                       // (R = 32 has the registers: the identities stay live across the hop)
    const int tau = c.mem | (c.low4 << 5);     // F3 / I1: this thread's residues are tau and RES - tau
    auto set_base = [&](int s, int skew) {     // region base of set s (0: tau, 1: its partner) for the E2 loads / E3 stores
        const int an = wv ? (c.a ^ 2) : (c.a < 2 ? c.a : (c.a ^ 1));
        const int y = c.low4;
        // (thread 0: the partner of residue 0 is RES / 2 = 512, bits 5..9 = 16 - at N = 65536 it is 1024 = 0 mod 1024)
        const int ye = s ? (is0 ? 16 : (c.mem ? 31 - y : ((32 - y) & 31))) : y;
        const int ae = s ? an : c.a;
        return RG * wv + 16 * (ae & 1) + 528 * ((ae >> 1) | ((ye >> 4) << 1)) + skew * (ye & 15);
    };
    const int64_t k_first = k_begin > 0 ? k_begin - 1 : k_begin;
    const int n_it = (int)(k_end - k_first);
    for (int it = 0; it < n_it; ++it) {
        const int64_t k = k_first + it;
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[R];
        {   // register brev5(q) := z[q * T + t] * window, stage 0 inside the fold
            GF src = hop_src(p, xc, xt, k);
            GF win = per_hop(p.window);
            v2f cbW = {0.f, 0.f}, sbW = cbW;
            if constexpr (HANN) {
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tid;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
            }
            const HannK64 &HW = HANN_W15;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x40000000, 0x00020000);
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define ROW(i) (((i) >> 1) + ((i) & 1) * (R / 2))
            constexpr int LB = HANN ? R : 16;
#pragma unroll
            for (int q0 = 0; q0 < R; q0 += LB) {
                float xr0[LB], xr1[LB], wr0[HANN ? 1 : LB], wr1[HANN ? 1 : LB];
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * lane2), 4 * 2 * T * ROW(q0 + q), 0);
                    xr0[q] = __uint_as_float(x.x);
                    xr1[q] = __uint_as_float(x.y);
                    if constexpr (!HANN) {
                        wr0[q] = (win + 2 * T * ROW(q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * ROW(q0 + q))[lane2 + 1];
                    }
                }
#pragma unroll
                for (int q = 0; q < LB; q += 2) {
                    v2f w0, w1;
                    if constexpr (HANN) {
                        const int r0 = ROW(q0 + q), r1 = ROW(q0 + q + 1);
                        w0 = __builtin_elementwise_fma(v2f{HW.s[r0], HW.s[r0]}, sbW,
                             __builtin_elementwise_fma(v2f{HW.c[r0], HW.c[r0]}, cbW, v2f{0.5f, 0.5f}));
                        w1 = __builtin_elementwise_fma(v2f{HW.s[r1], HW.s[r1]}, sbW,
                             __builtin_elementwise_fma(v2f{HW.c[r1], HW.c[r1]}, cbW, v2f{0.5f, 0.5f}));
                    } else {
                        w0 = v2f{wr0[q], wr1[q]};
                        w1 = v2f{wr0[q + 1], wr1[q + 1]};
                    }
                    const v2f a = v2f{xr0[q], xr1[q]} * w0, xh = v2f{xr0[q + 1], xr1[q + 1]};
                    v[brev_c(ROW(q0 + q), b)] = __builtin_elementwise_fma(xh, w1, a);
                    v[brev_c(ROW(q0 + q), b) + 1] = __builtin_elementwise_fma(-xh, w1, a);
                }
            }
#undef ROW
            dit_g<R, 1, b - 1, 0, false, false>(v);
        }
        // ---- E1: F1 -> F2 (cross-wave, writer-major). Element P: P0..4 = register q, P5..13 = brev9(tid)
        v2f w[R];
        {
            const int bs = (int)(__brev((unsigned)tid) >> 23);  // brev9(t): bit i = P(5 + i)
            const int w0b = RG * wv + (bs >> 5) + (((bs >> 3) & 3) << 9);
            const int r0b = c.low4 + 16 * ((c.mem >> 4) | ((c.mem & 15) << 1));
#pragma unroll
            for (int q = 0; q < 32; ++q) lds[w0b + 16 * ((q >> 4) | ((q & 15) << 1))] = to_f2(v[q]);
            BIG5_BAR();
#pragma unroll
            for (int j = 0; j < 32; ++j) w[j] = to_v(lds[r0b + RG * brev_c(j & 7, 3) + 512 * (j >> 3)]);
            BIG5_BAR();  // (every wave has read every region: E2 may overwrite the own one)
        }
        dit_g<32, b, b + 4, b, false, true>(w, to_v(lds[T_A + 16 * c.mem]));
        // ---- E2: F2 -> F3 inside the wave: x + 16 (a & 1) + 33 uu + 528 ((a >> 1) | gp << 1), y = j = x | gp << 4
        v2f va[16], vb[16];
        const float2 wrl = ldg2((GV2)per_hop(reinterpret_cast<const float *>(p.rtab)) + tau);  // W_N^tau, for the pair stage
        {
            const int e2w = RG * wv + 16 * (c.a & 1) + 528 * (c.a >> 1) + 33 * c.low4;
            BIG5_FENCE();
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[e2w + (j & 15) + 1056 * (j >> 4)] = to_f2(w[j]);
            BIG5_FENCE();
            const int ra = set_base(0, 1), rb = set_base(1, 1);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                va[q] = to_v(lds[ra + 33 * q]);
                vb[q] = to_v(lds[rb + 33 * q]);
            }
        }
        // ---- F3, the pair stage in registers, I1 (big4_kernel's, with r = tau)
        {
            const int r = tau;
            {
                const v2f wa = to_v(lds[T_A + r]);  // W_M^r
                const v2f k16 = {W32_RE[2], W32_IM[2]};
                v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
                if (is0) wb = v2f{W32_RE[1], W32_IM[1]};  // thread 0: residue RES/2 -> W_32
                dit_g<16, b + 5, b + 8, b + 5, false, true>(va, wa);
                dit_g<16, b + 5, b + 8, b + 5, false, true>(vb, wb);
            }
            const bool sp = is0;
            v2f s8 = va[8];
            if (wv == 0) {
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
                const float2 w0 = lds[SCR];
                const float2 wrh = make_float2(sp ? w0.x : wrl.x, sp ? w0.y : wrl.y);
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
                    if (q == 0)
                        pair_regs_pk4<LOG2N, true>(va[q], vb[15 - q], wq, x0, key, VA, VB, sp);
                    else
                        pair_regs_pk4<LOG2N>(va[q], vb[15 - q], wq, (q < 8 ? x0 : x0h) + (uint32_t)q * dx, key, VA, VB);
                    va[q] = VA;
                    vb[15 - q] = VB;
                }
            }
            if (wv == 0) {  // bin M/2 pairs with itself; un-deal thread 0's registers
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
        // ---- E3: I1 -> I2 inside the wave: q' + 16 (a & 1) + 33 x + 528 ((a >> 1) | gp << 1); I2 register j = P'4..8 holds
        // the residue bits 9..5: y = brev5(j)
        {
            const int wa_ = set_base(0, 33), wb_ = set_base(1, 33);
            const int e3r = RG * wv + c.low4 + 16 * (c.a & 1) + 528 * (c.a >> 1);
            BIG5_FENCE();
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                lds[wa_ + q] = to_f2(va[q]);
                lds[wb_ + q] = to_f2(vb[q]);
            }
            BIG5_FENCE();
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int y = brev_c(j, 5);
                v[j] = to_v(lds[e3r + 33 * (y & 15) + 1056 * (y >> 4)]);
            }
        }
        dit_g<32, 4, 8, 4, true, true>(v, to_v(lds[T_A + c.low4 * R]));  // I2: stages 4..8, base W_512^l4 = W_M^(32 l4)
        // ---- E4: I2 -> I3 (cross-wave, reader-major): element l4 | j << 4 | brev5(member) << 9 goes to the region of the
        // wave that reads it (P'6..8 = (j >> 2) & 7) at P'0..5 | (P'9..13) << 6
        v2f y[R];
        {
            const int bm = (int)(__brev((unsigned)c.mem) >> 27);
            const int w1b = c.low4 + 64 * bm;
            BIG5_BAR();  // (the other waves are done with their regions: their E3 is behind them)
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[w1b + RG * (j >> 2) + 16 * (j & 3)] = to_f2(v[j]);
            BIG5_BAR();
#pragma unroll
            for (int q = 0; q < 32; ++q) y[q] = to_v(lds[RG * wv + (tid & 63) + 64 * q]);
        }
        dit_g<R, 9, m - 1, 9, true, true>(y, to_v(lds[T_A + tid]));  // I3: stages 9..13
        // ---- epilogue: synthesis window, overlap-add, store (big4_kernel<32>'s)
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
            const HannK64 &HW = HANN_W15;
            const HannK64 &HE = HANN_E15;
            const v2f hf = {0.5f, 0.5f};
            const float ak = p.amp * (-0.25f / (float)(1 << LOG2N));  // pair_regs_pk4 leaves the -1/(4N) out
            const v2f ampk = {ak, ak};
            v2f hfE = hf;
            if constexpr (HANN) {
                cbE *= ampk;
                sbE *= ampk;
                hfE = hf * ampk;
            }
            const int64_t g0 = k * (int64_t)H;
            GFW dst = outc + (g0 / (int64_t)pitch - p.out_origin);
            const uint32_t kr = (uint32_t)(g0 % pitch);
            constexpr int EB = 8;
#pragma unroll
            for (int q0 = 0; q0 < PH; q0 += EB) {
                float wr0[EB], wr1[EB], wt0[EB], wt1[EB], e0[EB], e1[EB];
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
                                       __builtin_elementwise_fma(v2f{HE.c[q0 + q], HE.c[q0 + q]}, cbE, hfE));
                        wr0[q] = wh.x, wr1[q] = wh.y, wt0[q] = wt.x, wt1[q] = wt.y, e0[q] = ev.x, e1[q] = ev.y;
                    }
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    const v2f head = y[q0 + q] * v2f{wr0[q], wr1[q]};
                    const v2f nt = y[q0 + q + PH] * v2f{wt0[q], wt1[q]};
                    if (k >= k_begin) {
                        const v2f o = HANN ? (head + tail[q0 + q]) * v2f{e0[q], e1[q]}
                                           : (head + tail[q0 + q]) * v2f{e0[q], e1[q]} * ampk;  // stretcher.rs:97-100
                        if constexpr (PITCH1) {
                            __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * (q0 + q) + lane2));
                        } else {
                            const uint32_t a0 = kr + 2u * (uint32_t)(tid + T * (q0 + q)), a1 = a0 + 1;
                            const uint32_t d0 = a0 / pitch, d1 = a1 / pitch;
                            if (d0 * pitch == a0) dst[d0] = o.x;
                            if (d1 * pitch == a1) dst[d1] = o.y;
                        }
                    }
                    tail[q0 + q] = nt;
                }
            }
        }
    }
}

}  // namespace

size_t big5_lds_bytes() { return sizeof(float2) * (size_t)big5_lds_float2(); }
hipError_t launch_big5(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(BIG5_T);
    const size_t lds = sizeof(float2) * (size_t)big5_lds_float2();
    const bool hann = p.hann_rot != nullptr;
    // (the engine sets the stash up for pitch 1 and the default window only: with a caller's window the seam
    // instantiation spills 45 registers)
    if (p.seam_head != nullptr && p.pitch == 1 && hann) hipLaunchKernelGGL((big5_kernel<true, true, true>), grid, block, lds, s, p);
    else if (p.pitch == 1 && hann) hipLaunchKernelGGL((big5_kernel<true, true>), grid, block, lds, s, p);
    else if (p.pitch == 1) hipLaunchKernelGGL((big5_kernel<true, false>), grid, block, lds, s, p);
    else if (hann) hipLaunchKernelGGL((big5_kernel<false, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((big5_kernel<false, false>), grid, block, lds, s, p);
    return hipGetLastError();
}

hipError_t launch_big5s(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(BIG5_T);
    const size_t lds = sizeof(float2) * (size_t)big5s_lds_float2();
    const bool hann = p.hann_rot != nullptr;
    if (p.pitch == 1 && hann) hipLaunchKernelGGL((big5s_kernel<true, true>), grid, block, lds, s, p);
    else if (p.pitch == 1) hipLaunchKernelGGL((big5s_kernel<true, false>), grid, block, lds, s, p);
    else if (hann) hipLaunchKernelGGL((big5s_kernel<false, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((big5s_kernel<false, false>), grid, block, lds, s, p);
    return hipGetLastError();
}

}  // namespace rc
