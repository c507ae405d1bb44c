#include "rc_bigdit.hpp"

namespace rc {
namespace {

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
//   in registers: the thread that owns head sample q owns tail sample q + H, so the carried tail y_{k-1}[H..]
//   never leaves the thread (R = 64: 27 of its 32 pairs in registers, the last 5 in LDS; no global scratch).
// Exchanges go through one 16 400-element LDS buffer (131 KB: one workgroup = 8 waves per CU), a single
// round for R = 32, two rounds of 32 registers per thread for R = 64.
// the exchange barriers order LDS traffic only: global stores of the epilogue (this thread's own output and tail
// addresses) may still be in flight when the next hop starts
#define BIG4_BAR()                                                                       \
    do {                                                                                 \
        if (RC_B4_ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
    } while (0)
#ifndef RC_B4_ABL
#define RC_B4_ABL 0  // timing-only ablations (`make variant`; results are wrong by construction): 1 no input loads,
                     // 4 no output stores, 8 the E2 and E3 barriers become fences (32: E2 only, 64: E3 only),
                     // 128 the entry barriers of E1 / E4 become fences, 16 no barrier at all - profiles/r04a_c5_ablations.txt
#endif
// (timing-only, RC_B4_ABL bit 8: the E2 / E3 barriers become compiler fences - what desynchronised waves would buy)
#define BIG4_BAR_MID()                                                          \
    do {                                                                        \
        if (RC_B4_ABL & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        else BIG4_BAR();                                                        \
    } while (0)
#define BIG4_BAR_ENTRY()  /* (bit 128: the entry barriers of E1 / E4 - what hop4's own-region scheme would drop) */ \
    do {                                                                        \
        if (RC_B4_ABL & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        else BIG4_BAR();                                                        \
    } while (0)
#define BIG4_BAR_E2()                                                           \
    do {                                                                        \
        if (RC_B4_ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        else BIG4_BAR_MID();                                                    \
    } while (0)
#define BIG4_BAR_E3()                                                           \
    do {                                                                        \
        if (RC_B4_ABL & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
        else BIG4_BAR_MID();                                                    \
    } while (0)
// R = 64 (BASELINE C5): what the measurements of rounds 2-4 settled (docs/LAB_NOTES, DESIGN.md 5.4, profiles/r04a_c5_ablations.txt)
//  * the last inverse stage (the one that pairs head sample q with tail sample q + PH) is computed inside the
//    epilogue, one register pair at a time with its twiddle rebuilt on the spot, instead of inside I3 with 16
//    twiddles (32 VGPRs) live next to the 128 data registers: that is what makes room for the carried tail
//  * E1's second round of stores is interleaved with F2 on the first group, E4's first round with I2 on the second
//    group and its second round with I3 on the first half: one ds_write per BIG4_OVL_K VALU instructions (3 / 4 / 5 / 6 /
//    8 measured: flat, 4 best); the same for E2 / E3 measured within noise and is not built
//  * input rows are fetched in butterfly-pair order (0, R/2, 1, R/2 + 1, ...) with stage 0 inside the load loop, as a
//    two-deep pipeline of 16-row batches (8 rows: flat, 32 rows = everything in flight: +1.5 %)
//  * thread identities are re-derived per phase from an opaque copy of the thread id (hoisted they are spilled)
constexpr int BIG4_T = 512;
constexpr int BIG4_XBUF = 16400;  // exchange buffer, float2 slots (16384 + the 15 of the E1 / E3 index map)
// tables behind the buffer: W_M^r [TA], W_N^r [TR] for r <= RES/2, thread 0's second twiddle base
// R = 64 with the carried tail in registers: the last BIG4_TAIL_LDS of a thread's 32 tail pairs live in LDS instead
// (behind the tables; 512 float2 per pair: what is left of the 160 KiB takes three) - 192 data registers plus the
// temporaries of the widest phases are ~8 more than the allocator places, and what it spills instead goes through
// scratch memory, in line behind the output stores (8 spilled dwords cost 2.3 %)
constexpr int big4_tail_lds(int R) { return R == 64 ? BIG4_TAIL_LDS : 0; }
// ... and to make room for them the W_N^r table (two reads per thread and hop, in the pair stage) stays in global
// memory for that kernel: its loads are issued in front of the E2 exchange, far from any store
constexpr bool big4_tr_global(int R) { return big4_tail_lds(R) > 3; }
constexpr int big4_lds_float2(int R) {
    return BIG4_XBUF + (big4_tr_global(R) ? 1 : 2) * (16 * R + 1) + 8 + 512 * big4_tail_lds(R);
}
static_assert(sizeof(float2) * big4_lds_float2(64) <= 160 * 1024, "big4 LDS budget");

template <int R, bool PITCH1, bool HANN>
__global__ __launch_bounds__(BIG4_T, 2) void big4_kernel(const HopParams p) {
    constexpr int b = clog2(R), m = b + 9, LOG2N = m + 1, M = 1 << m, H = M, T = BIG4_T;
    constexpr int RES = 1 << (b + 5), G = R / 32, NS = R / 16, PH = R / 2;
    constexpr bool TRG = big4_tr_global(R);
    constexpr int T_A = BIG4_XBUF, T_R = T_A + RES / 2 + 1, SCR = T_R + (TRG ? 0 : RES / 2 + 1);
    constexpr int TL = big4_tail_lds(R), TLB = SCR + 8, PHR = R / 2 - TL;  // tail pairs [PHR, R/2) live at lds[TLB + ...]
    constexpr bool OVL = R == 64;   // exchange stores interleaved with the next pass (E1, E4)
    constexpr bool LEAN = R > 32;   // 192 data registers: everything else is kept short-lived
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
    const int wv = tid >> 6;
    {
        GV2 wt = (GV2)p.wtab;  // [RES/2 + 1] exp(-2 pi i k / M)
        GV2 rt = (GV2)p.rtab;  // exp(-2 pi i j / N)
        for (int i = tid; i <= RES / 2; i += T) {
            lds[T_A + i] = ldg2(wt + i);
            if constexpr (!TRG) lds[T_R + i] = ldg2(rt + i);
        }
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
    const bool is0 = tid == 0;
    Stamps stp;
    stp.init();
    // Thread identities are re-derived from an opaque copy of the thread id inside each phase: left to itself the
    // compiler hoists the ~40 loop-invariant LDS bases / residues out of the hop loop and then spills them
    auto ptid = [&]() {
        int t = tid;
        if (R > 32) opaque(t);  // (R = 32 has the registers: there the hoisted values are 2 % faster)
        return t;
    };
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
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tt;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
            }
            const HannK64 &HW = R == 32 ? HANN_W15 : HANN_W16;
            // R = 32 with the computed window: every load of the hop in flight at once (one memory latency per hop);
            // batches of 16 when the window comes from its table too (register budget); R = 64 with the computed
            // window: the pipeline below
            constexpr int LB = HANN ? R : 16;
            // rows are fetched in the order 0, R/2, 1, R/2 + 1, ...: butterfly stage 0 pairs row q with row q + R/2 (registers
            // brev(q) and brev(q) + 1), and the compiler folds the second row's window multiply into that butterfly, so a
            // row loaded long before its partner would wait for it as two live values (raw samples and window)
#define ROW(i) (((i) >> 1) + ((i) & 1) * (R / 2))
            if constexpr (HANN && R == 64) {
                // Two batches of rows in flight: batch i + 2 is requested as soon as batch i has been folded into
                // v[] (stage 0), so a hop exposes about one memory latency instead of one per batch. The window
                // values are computed pair by pair inside the fold (hoisted over a whole batch they are LB more
                // live pairs, and the allocator spills)
                constexpr int PB = BIG4_PIPE_ROWS, NB = R / PB;
                float xp0[2][PB], xp1[2][PB];
                auto issue = [&](int i, float (&x0)[PB], float (&x1)[PB]) {
#pragma unroll
                    for (int q = 0; q < PB; ++q) {
                        if (RC_B4_ABL & 1) {
                            x0[q] = (float)(lane2 + ROW(i * PB + q)) + (float)k;
                            x1[q] = x0[q] * 0.5f;
                        } else {
                            x0[q] = (src + 2 * T * ROW(i * PB + q))[lane2];
                            x1[q] = (src + 2 * T * ROW(i * PB + q))[lane2 + 1];
                        }
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
            } else
#pragma unroll
            for (int q0 = 0; q0 < R; q0 += LB) {
                float xr0[LB], xr1[LB], wr0[HANN ? 1 : LB], wr1[HANN ? 1 : LB];
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    if (RC_B4_ABL & 1) {  // timing only: no input loads
                        xr0[q] = (float)(lane2 + ROW(q0 + q)) + (float)k;
                        xr1[q] = xr0[q] * 0.5f;
                    } else {
                        xr0[q] = (src + 2 * T * ROW(q0 + q))[lane2];
                        xr1[q] = (src + 2 * T * ROW(q0 + q))[lane2 + 1];
                    }
                    if constexpr (!HANN) {
                        wr0[q] = (win + 2 * T * ROW(q0 + q))[lane2];
                        wr1[q] = (win + 2 * T * ROW(q0 + q))[lane2 + 1];
                    }
                }
                v2f wq[LB];
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    if constexpr (HANN)
                        wq[q] = __builtin_elementwise_fma(v2f{HW.s[ROW(q0 + q)], HW.s[ROW(q0 + q)]}, sbW,
                                __builtin_elementwise_fma(v2f{HW.c[ROW(q0 + q)], HW.c[ROW(q0 + q)]}, cbW, v2f{0.5f, 0.5f}));
                    else
                        wq[q] = v2f{wr0[q], wr1[q]};
                }
                // butterfly stage 0 right here (registers brev(q) and brev(q) + 1 = rows q and q + R/2): a +- b with
                // a = x_q w_q and b = x_{q+R/2} w_{q+R/2} is one multiply and two FMAs
#pragma unroll
                for (int q = 0; q < LB; q += 2) {
                    const v2f a = v2f{xr0[q], xr1[q]} * wq[q], xh = v2f{xr0[q + 1], xr1[q + 1]};
                    v[brev_c(ROW(q0 + q), b)] = __builtin_elementwise_fma(xh, wq[q + 1], a);
                    v[brev_c(ROW(q0 + q), b) + 1] = __builtin_elementwise_fma(-xh, wq[q + 1], a);
                }
                // the register tail leaves no room for a second batch in flight: keep the batches apart
                if constexpr (R > 32) __builtin_amdgcn_sched_barrier(0);
            }
#undef ROW
            stp.mark(0);
            dit_g<R, 1, b - 1, 0, false, false>(v);
        }
        stp.mark(1);
        // ---- E1: F1 -> F2, round g moves the registers with position bit 5 = g
        v2f w[R];
        {
            const int t_ = ptid(), lf = t_ & 31, uu = t_ >> 5;
            const int bs = (int)(__brev((unsigned)t_) >> 23);            // brev9(t)
            const int b1s = (bs << 5) + (bs >> 5);                       // e1(q | brev9(t) << 5) = q + this
            const int b1l = lf + (uu << 10) + uu;                        // e1(lf | j << 5 | uu << 10) = (j << 5) + this
            if constexpr (OVL) {
                // R = 64, two rounds: the LDS takes the 32 stores of a wave at ~50 cycles apiece while all eight waves
                // store (80 B/clk per CU), so the second round's stores are issued one at a time between the butterflies
                // of F2 on the first round's group instead of in front of a barrier
                BIG4_BAR_ENTRY();
#pragma unroll
                for (int q = 0; q < 32; ++q) lds[b1s + q] = to_f2(v[q]);
                BIG4_BAR();
#pragma unroll
                for (int j = 0; j < 32; ++j) w[j] = to_v(lds[b1l + (j << 5)]);
                const v2f wf0 = to_v(lds[T_A + 16 * lf]);
                BIG4_BAR();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 32; ++q) lds[b1s + q] = to_f2(v[32 + q]);
                {
                    v2f grp[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) grp[j] = w[j];
                    dit_g<32, b, b + 4, b, false, true, LEAN>(grp, wf0);
#pragma unroll
                    for (int j = 0; j < 32; ++j) w[j] = grp[j];
                }
#pragma unroll
                for (int i = 0; i < 32; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // one DS write
                    __builtin_amdgcn_sched_group_barrier(0x2, BIG4_OVL_K, 0);  // then K VALU
                }
                __builtin_amdgcn_sched_barrier(0);
                BIG4_BAR();
#pragma unroll
                for (int j = 0; j < 32; ++j) w[32 + j] = to_v(lds[b1l + (j << 5)]);
                stp.mark(2);
                {
                    v2f grp[32];
#pragma unroll
                    for (int j = 0; j < 32; ++j) grp[j] = w[32 + j];
                    dit_g<32, b, b + 4, b, false, true, LEAN>(grp, vcmul(wf0, v2f{W64.re[1], W64.im[1]}));
#pragma unroll
                    for (int j = 0; j < 32; ++j) w[32 + j] = grp[j];
                }
            } else {
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
        }
        if constexpr (!OVL) {
        stp.mark(2);
        {   // F2: stages b..b+4 on each group; base W_RES^(lf | g << 5) = W_M^(16 lf) * (g ? W_64 : 1)
            const int lf = ptid() & 31;
            const v2f wf0 = to_v(lds[T_A + 16 * lf]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = w[32 * g + j];
                dit_g<32, b, b + 4, b, false, true, LEAN>(grp, g ? vcmul(wf0, v2f{W64.re[1], W64.im[1]}) : wf0);
#pragma unroll
                for (int j = 0; j < 32; ++j) w[32 * g + j] = grp[j];
            }
        }
        }
        stp.mark(3);
        // ---- E2: F2 -> F3. slot = (residue mod 1024) | uu << 10; R = 64: round 0 = residues < 1024
        v2f st[NS][16];
        float2 wrg[TRG ? NS / 2 : 1];  // W_N^r of this thread's residues, requested here for the pair stage
        if constexpr (TRG) {
            GV2 rt = (GV2)per_hop(reinterpret_cast<const float *>(p.rtab)) + ptid();
#pragma unroll
            for (int gp = 0; gp < NS / 2; ++gp) wrg[gp] = ldg2(rt + 512 * gp);
        }
        {
            const int tid = ptid(), lf = tid & 31, uu = tid >> 5;
            const int b2s = lf | (uu << 10);
#pragma unroll
            for (int rnd = 0; rnd < G; ++rnd) {
                BIG4_BAR_E2();
#pragma unroll
                for (int kk = 0; kk < 32; ++kk) {
                    if (R == 32) lds[b2s + (kk << 5)] = to_f2(w[kk]);
                    else lds[b2s + ((kk >> 4) << 5) + ((kk & 15) << 6)] = to_f2(w[32 * (kk >> 4) + 16 * rnd + (kk & 15)]);
                }
                BIG4_BAR_E2();
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
            const int r = ptid() + 512 * gp;
            v2f(&va)[16] = st[2 * gp];
            v2f(&vb)[16] = st[2 * gp + 1];
            {
                const v2f wa = to_v(lds[T_A + r]);  // W_M^r
                const v2f k16 = {W32_RE[2], W32_IM[2]};
                v2f wb = vcmul(v2f{wa.x, -wa.y}, k16);  // W_M^(RES - r) = W_16 conj(W_M^r)
                if (gp == 0 && is0) wb = v2f{W32_RE[1], W32_IM[1]};  // thread 0: residue RES/2 -> W_32
                dit_g<16, b + 5, b + 8, b + 5, false, true, LEAN>(va, wa);
                dit_g<16, b + 5, b + 8, b + 5, false, true, LEAN>(vb, wb);
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
                float2 wrl, wrh;
                if constexpr (TRG) {
                    wrl = wrg[gp];
                    const float2 w0 = lds[SCR];
                    wrh = make_float2(sp ? w0.x : wrl.x, sp ? w0.y : wrl.y);
                } else {
                    wrl = lds[T_R + r];
                    wrh = lds[sp ? SCR : T_R + r];
                }
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
            const int tid = ptid(), l4 = tid & 15, hi = tid >> 4;
#pragma unroll
            for (int rnd = 0; rnd < G; ++rnd) {
                BIG4_BAR_E3();
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
                BIG4_BAR_E3();
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
        constexpr bool FUSE = R == 64;  // the last inverse stage runs inside the epilogue
        v2f y[R];
        if constexpr (OVL) {
            // I2 on group 0; E4 round 0's stores between the butterflies of I2 on group 1; E4 round 1's stores between
            // those of I3 (stages 9..m-2) on the half that round 0 delivered
            const int tid = ptid(), l4 = tid & 15, hi = tid >> 4;
            const int b4s = l4 | (hi << 9);
            const v2f wf = to_v(lds[T_A + l4 * R]);
            {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = v[j];
                dit_g<32, 4, 8, 4, true, true, LEAN>(grp, wf);
#pragma unroll
                for (int j = 0; j < 32; ++j) v[j] = grp[j];
            }
            BIG4_BAR_ENTRY();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[b4s + (j << 4)] = to_f2(v[j]);
            {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = v[32 + j];
                dit_g<32, 4, 8, 4, true, true, LEAN>(grp, wf);
#pragma unroll
                for (int j = 0; j < 32; ++j) v[32 + j] = grp[j];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, BIG4_OVL_K, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            stp.mark(7);
            BIG4_BAR();
#pragma unroll
            for (int q = 0; q < 32; ++q) y[q] = to_v(lds[tid + (q << 9)]);
            const v2f wf2 = vcsq(to_v(lds[T_A + tid]));
            BIG4_BAR();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 32; ++j) lds[b4s + (j << 4)] = to_f2(v[32 + j]);
            {
                v2f grp[R / 2];
#pragma unroll
                for (int j = 0; j < R / 2; ++j) grp[j] = y[j];
                dit_g<R / 2, 9, m - 2, 9, true, true, LEAN>(grp, wf2);
#pragma unroll
                for (int j = 0; j < R / 2; ++j) y[j] = grp[j];
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, BIG4_OVL_K, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            BIG4_BAR();
#pragma unroll
            for (int q = 0; q < 32; ++q) y[q + 32] = to_v(lds[tid + (q << 9)]);
            stp.mark(8);
            {
                v2f grp[R / 2];
#pragma unroll
                for (int j = 0; j < R / 2; ++j) grp[j] = y[R / 2 + j];
                dit_g<R / 2, 9, m - 2, 9, true, true, LEAN>(grp, wf2);
#pragma unroll
                for (int j = 0; j < R / 2; ++j) y[R / 2 + j] = grp[j];
            }
        } else {
        {   // I2: inverse stages 4..8 on each group, base W_512^l4 = W_M^(l4 R)
            const int l4 = ptid() & 15;
            const v2f wf = to_v(lds[T_A + l4 * R]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                v2f grp[32];
#pragma unroll
                for (int j = 0; j < 32; ++j) grp[j] = v[32 * g + j];
                dit_g<32, 4, 8, 4, true, true, LEAN>(grp, wf);
#pragma unroll
                for (int j = 0; j < 32; ++j) v[32 * g + j] = grp[j];
            }
        }
        stp.mark(7);
        // ---- E4: I2 -> I3 (registers = P9.., thread = P0..P8), round g = P14
        {
            const int tid = ptid(), l4 = tid & 15, hi = tid >> 4;
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
        }
        stp.mark(8);
        if constexpr (!OVL) dit_g<R, 9, m - 1, 9, true, true>(y, to_v(lds[T_A + ptid()]));  // (R = 64: done above)
        stp.mark(9);
        // ---- epilogue: synthesis window, overlap-add, store (the carried tail in registers, R = 64: its last pairs in LDS)
        {
            GF win = per_hop(p.window);
            GF esrc = per_hop(p.env);
            v2f cbW = {0.f, 0.f}, sbW = cbW, cbE = cbW, sbE = cbW;
            if constexpr (HANN) {
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * tt;
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
            v2f hfE = hf;
            if constexpr (HANN) {  // env[i] * amp = amp/2 + c_q (amp cb) + s_q (amp sb): the amplitude rides on the rotation
                cbE *= ampk;
                sbE *= ampk;
                hfE = hf * ampk;
            }
            const int64_t g0 = k * (int64_t)H;
            GFW dst = outc + (g0 / (int64_t)pitch - p.out_origin);
            const uint32_t kr = (uint32_t)(g0 % pitch);
            // batch of window values / table loads in flight (register budget): 16 pairs for the BASELINE C5 instantiation,
            // 4 where the window comes from its table or the store decimates (those spill at 16)
            constexpr int EB = R > 32 ? ((HANN && PITCH1) ? BIG4_EPI_BATCH : 4) : 8;
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
                    if (q0 + q < PHR) tq[q] = tail[q0 + q];
                    else tq[q] = to_v(lds[TLB + 512 * (q0 + q - PHR) + tt]);
                }
#pragma unroll
                for (int q = 0; q < EB; ++q) {
                    v2f yh = y[q0 + q], yt = y[q0 + q + PH];
                    if constexpr (FUSE) {  // inverse stage m-1: (yh, yt) = (a + conj(w) b, a - conj(w) b), w = W_M^tid W_64^c
                        const int c = q0 + q;  // < 32
                        const v2f wfl = to_v(lds[T_A + tt]);
                        const v2f k64 = {W64.re[c & 15], W64.im[c & 15]};
                        const v2f tw = (c & 15) == 0 ? wfl : vcmul(wfl, k64);
                        const v2f a = yh, bb = yt;
                        if (c < 16) vdit_m<true>(a, bb, tw, yh, yt);
                        else vdit_rot_m<true>(a, bb, tw, yh, yt);
                    }
                    const v2f head = yh * v2f{wr0[q], wr1[q]};
                    const v2f nt = yt * v2f{wt0[q], wt1[q]};
                    if (k >= k_begin) {
                        // stretcher.rs:97-100 operation order
                        // stretcher.rs:97-100; with the computed envelope the amplitude is already inside it
                        const v2f o = HANN ? (head + tq[q]) * v2f{e0[q], e1[q]} : (head + tq[q]) * v2f{e0[q], e1[q]} * ampk;
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
                    if (q0 + q < PHR) tail[q0 + q] = nt;
                    else lds[TLB + 512 * (q0 + q - PHR) + tt] = to_f2(nt);
                }
                // (register tail at R = 64: 192 registers are taken; keep the batches' window / twiddle temporaries apart)
                if constexpr (R > 32) __builtin_amdgcn_sched_barrier(0);
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

}  // namespace

size_t big4_tail_scratch_floats(int log2n) {
    const int R = log2n == 16 ? 64 : 32;
    (void)R;
    return 0;  // (round 2's R = 64 kernel carried the tail through a per-workgroup scratch; since round 3 it lives in registers / LDS)
}
hipError_t launch_big4(int log2n, const HopParams &p, hipStream_t s) {
    if (log2n == 15) return launch_big4_r<32>(p, s);
    if (log2n == 16) return launch_big4_r<64>(p, s);
    return hipErrorInvalidValue;
}

}  // namespace rc
