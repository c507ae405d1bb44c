// N = 16384 (the reference's default window, src/main.rs:31-36): hop4_kernel (default hanning window; the kernel
// bench.py measures) and hop2_kernel's table variant (caller-supplied window). DESIGN.md 5.1b / 5.1d.
#include "rc_dit.hpp"

#ifndef RC_HOP4_BUFLOAD
#define RC_HOP4_BUFLOAD 1  // hop4's input rows through buffer loads (0: global loads with 64-bit lane addresses, for A/B)
#endif
namespace rc {
namespace {

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
constexpr int f1_idx(int n) { return n + (n >> 5); }
constexpr int HOP2_XBUF = 8192 + 256 + 32;
constexpr int HOP2_LDS_FLOAT2 = HOP2_XBUF + 8 + 32 + 256 + 256 + 16 + 24 + 1024;  // 79 232 B
// LDS index map of the exchange buffer: a weight per position bit, so the map is additive over
// disjoint bit fields (per-thread base VGPR + immediate offset per register). The weights were searched
// (banking model of MI355X_MICROARCH.md: ds_write_b64 = 16-lane groups on 32 banks, ds_read_b64 =
// 32-lane groups on 64 banks) so that the stores of all four exchanges are conflict-free (the old map
// n + (n >> 5) + (n >> 8) was built for the read groups only and 2-way conflicted on the stores of
// exchanges 1 and 3: SQ_LDS_DATA_FIFO_FULL for half of the SQ cycles).
constexpr int F3_W[13] = {1, 2, 4, 8, 16, 32, 64, 131, 259, 520, 1038, 2079, 4156};
constexpr int f3_idx(int n) {
    int r = 0;
    for (int i = 0; i < 13; ++i) r += ((n >> i) & 1) * F3_W[i];
    return r;
}

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
        for (int q = 0; q < P; ++q) lds[bE1s + f3_idx(q)] = to_f2(vn[q]);
        if constexpr (SWP2) {
            if (k > k_first) epilogue(k - 1, vo);
        }
        st.mark(2);
        __syncthreads();
        st.mark(3);
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = xld(lds, b4f3 + f3_idx(q << 4));
        dit_stages<32, m, 5, 8, 4, false, true>(v, to_v(lds[T_B + l4]));
        st.mark(4);
        // E2 store is IN PLACE (same layout, same index map as the E1 load): each thread overwrites
        // exactly the elements it read, so no barrier is needed between the two
#pragma unroll
        for (int q = 0; q < P; ++q) lds[b4f3 + f3_idx(q << 4)] = to_f2(v[q]);
        st.mark(5);
        __syncthreads();
        st.mark(6);
        v2f va[16], vb[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            va[q] = xld(lds, bAr + f3_idx((RES * q)));
            vb[q] = xld(lds, bBr + f3_idx((RES * q)));
        }
        st.mark(7);
        __syncthreads();
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
        if (tid < 64) {  // wave 0: lanes 0..16 compute thread 0's 17 pairs from the scratch
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
            lds[bE3a + f3_idx(q)] = to_f2(pa[q]);
            lds[bE3b + f3_idx(q)] = to_f2(pb[q]);
        }
        st.mark(13);
        if constexpr (SWP) win_f1(vn);  // next hop's window + F1 while the E3 stores drain
        __syncthreads();
        st.mark(14);
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = xld(lds, b4f3 + f3_idx(q << 4));
        dit_stages<32, m, 4, 8, 4, true, true>(v, to_v(lds[T_B + l4]));
        st.mark(15);
#pragma unroll
        for (int q = 0; q < P; ++q) lds[b4f3 + f3_idx(q << 4)] = to_f2(v[q]);  // in place (see E2)
        st.mark(16);
        __syncthreads();
        st.mark(17);
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = xld(lds, bE4l + f3_idx((q << 8)));
        st.mark(18);
        __syncthreads();
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
#if RC_FOLDPROD
#define HOP4_PAIR pair_regs_pk5
#else
#define HOP4_PAIR pair_regs_pk4
#endif
#define HOP4_BAR()                                                                    \
    do {                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
    } while (0)
// RC_DK_BAND fused into the pair stage: |gain| of real-spectrum bin f <= N/2 (lo <= f <= hi: inside)
__device__ __forceinline__ float band_gain(const HopParams &p, uint32_t f) {
    return (f - p.band_lo) <= p.band_span ? p.band_gin : p.band_gout;
}
// PITCHC: 1 = pitch 1, 2 / 3 = that pitch at compile time (pitch_store_pair), 0 = any pitch > 1 from HopParams
// TABW (round 5, pitch 1 only): a caller's window - analysis / synthesis window and envelope values come from the engine's
// tables (L2-resident) in batches of 4 rows instead of the rotation of the computed hanning window
template <int PITCHC, bool BAND = false, bool TABW = false>
__global__ __launch_bounds__(256, 3) void hop4_kernel(const HopParams p) {
    constexpr bool PITCH1 = PITCHC == 1;
    static_assert(!TABW || (PITCH1 && !BAND), "the table-window instantiation exists for pitch 1 without the band mask");
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
            {
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
            }
            *slot = got;
        }
        __syncthreads();
        // the ticket is the same for every lane: as a scalar, the run, the channel, the hop counter and the hop's phase
        // key (two 64-bit multiplies of mix64) live in SGPRs and run on the scalar unit. Read through LDS it was a
        // VGPR value: the key was computed per lane and its channel word was one of the kernel's spills, reloaded
        // every hop behind an s_waitcnt vmcnt(0) (tools/isa_stats.py --spills)
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
    const bool stash_first = seam && run > 0;
    const bool has_next = seam && run + 1 < p.runs_per_channel;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    GFW outc = (GFW)p.out + (size_t)ch * p.out_stride;
    const unsigned lane2 = 2u * (unsigned)tid;
    GV2 wtab = (GV2)p.wtab;
    const uint32_t pitch = PITCHC ? (uint32_t)PITCHC : p.pitch;

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
        if constexpr (!TABW) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float2 a = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid);
                const float2 b = ldg2((GV2)p.hann_rot + 512 * i + 2 * tid + 1);
                lds[T_H + 512 * i + 2 * tid] = make_float2(a.x, b.x);
                lds[T_H + 512 * i + 2 * tid + 1] = make_float2(a.y, b.y);
            }
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
            // (buffer stores like the input loads: descriptor + one lane offset + the row in the scalar offset; aux 2 = nt)
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, 0x40000000, 0x00020000);
            typedef unsigned v2u __attribute__((ext_vector_type(2)));
#define HOP4_STORE(o, row)                                                                                                  \
    do {                                                                                                                    \
        if constexpr (RC_HOP4_BUFLOAD && !TABW)                                                                             \
            __builtin_amdgcn_raw_buffer_store_b64(v2u{__float_as_uint((o).x), __float_as_uint((o).y)}, rd,                 \
                                                  (int)(4u * lane2), 4 * 2 * T * (row), 2);                                 \
        else __builtin_nontemporal_store(o, (GV2W)(dst + 2 * T * (row) + lane2));                                           \
    } while (0)
            if constexpr (TABW) {
                GF et2 = per_hop(p.env) + lane2;
#pragma unroll
                for (int q0 = 0; q0 < PH; q0 += 4) {
                    float e0[4], e1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        e0[q] = (et2 + 2 * T * (q0 + q))[0];
                        e1[q] = (et2 + 2 * T * (q0 + q))[1];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v2f o = (head[q0 + q] + tail[q0 + q]) * v2f{e0[q], e1[q]} * amp2;  // stretcher.rs:97-100
                        HOP4_STORE(o, q0 + q);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                const v2f o = (head[q] + tail[q]) * er;  // (y + tail) * (env * amp), stretcher.rs:97-100
                // non-temporal: the output is written once; the window overlap of consecutive hops stays in the XCD's L2
                HOP4_STORE(o, q);
            }
        } else {
            const int64_t kq = g0 / pitch;
            const uint32_t kr = (uint32_t)(g0 % pitch);
            GFW dst = outc + (kq - p.out_origin);
            int t2 = tid;
            opaque(t2);
            // F[t] = O[t * pitch] (src/resampler.rs:3-18): sample a = kr + i of this hop is kept iff a % pitch == 0,
            // at dst[a / pitch]. One division per hop and lane (q = 0); every further register pair is 2T = 512
            // samples on: quotient and remainder advance by the uniform 512 / pitch and 512 % pitch. The destination is
            // a uniform base (SGPRs) + a 32-bit byte offset per lane, as for pitch 1: no 64-bit address per store.
            // (Gathering a row's kept samples across the wave with ds_bpermute into ONE contiguous store per row, with
            // or without a lane mask, was measured: 4.68 / 4.72 ms against 4.67 for C3 - the two sparse stores are not
            // what pitch > 1 costs.)
            // Branch-free: the stores are raw buffer stores whose offset is out of range for a lane that keeps nothing
            // (the hardware drops it), so the epilogue stays ONE basic block - with an exec-masked branch around each of
            // the 32 stores the allocator spilled 20 more registers (31 dwords in and out of scratch per hop and lane
            // against 11 for pitch 1), and that, not the sparse stores, was what pitch > 1 cost.
            const unsigned long long da = (unsigned long long)dst;
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)da);
            const unsigned dhi = __builtin_amdgcn_readfirstlane((unsigned)(da >> 32));
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(((unsigned long long)dhi << 32) | dlo), 0, 0x40000000, 0x00020000);  // raw buffer, 1 GiB window
            if constexpr (PITCHC > 1) {
                const PitchOffsets<PITCHC> po = pitch_offsets<PITCHC>(kr + 2u * (uint32_t)t2);
#pragma unroll
                for (int q = 0; q < PH; ++q) {
                    const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                                   __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                    const v2f o = (head[q] + tail[q]) * er;
                    const float ox = o.x, oy = o.y;
                    pitch_store_pair<PITCHC, 2 * T>(rsrc, po, q, ox, oy);
                }
                return;
            }
            constexpr uint32_t DROP = 0xFFFFFFFCu;
            const uint32_t a00 = kr + 2u * (uint32_t)t2;
            const uint32_t d0 = a00 / pitch;
            uint32_t r = a00 - d0 * pitch, d4 = 4u * d0;
            const uint32_t qs4 = 4u * ((2u * T) / pitch), rs = (2u * T) - (qs4 / 4u) * pitch;
#pragma unroll
            for (int q = 0; q < PH; ++q) {
                const v2f er = __builtin_elementwise_fma(v2f{HANN_E14.s[q], HANN_E14.s[q]}, sbE,
                               __builtin_elementwise_fma(v2f{HANN_E14.c[q], HANN_E14.c[q]}, cbE, halfa));
                const v2f o = (head[q] + tail[q]) * er;
                const float ox = o.x, oy = o.y;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(ox), rsrc, r == 0 ? d4 : DROP, 0, 0);  // a0 = d * pitch
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(oy), rsrc, r + 1 == pitch ? d4 + 4u : DROP, 0, 0);
                d4 += qs4;
                r += rs;
                if (r >= pitch) {
                    r -= pitch;
                    d4 += 4u;
                }
            }
            // (one store per row - the two samples are never both kept - was measured too: 4.64-4.76 against 4.66-4.69 ms)
        }
    };
    // (Output stores: a CU drains about 11 bytes per clock towards memory, so the 16 back-to-back stores of a hop
    // stall a wave at issue - timing-only builds without them ran 9 % faster. Deferring them into the next hop's
    // first pass was built in round 2 and measured worse at every depth (the registers it needs spill): not here.)
    for (int64_t k = ((k_begin > 0 && !stash_first) ? k_begin - 1 : k_begin); k < k_end; ++k) {
        const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
        v2f v[P];
        st.mark(26);
        {   // register brev5(q) := z[q * T + t] * window ; F1 = stages 0..4
            GF src = hop_src(p, xc, xt, k);
            float xr0[P], xr1[P];
            if constexpr (RC_HOP4_BUFLOAD && !TABW) {
                // buffer loads: the hop's base in a resource descriptor (SGPRs), ONE 32-bit lane offset for all rows, the row
                // in the scalar offset - no 64-bit VALU address arithmetic (the global_load form spent 2 x 16 v_add_co /
                // v_addc per hop and wave on it, with an address pair live per two rows). Not in the table-window
                // instantiation: there the allocator spills 10 more dwords with them and the kernel is 7 % slower
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x40000000, 0x00020000);
                typedef unsigned v2u __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const v2u x = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(4u * lane2), 4 * 2 * T * q, 0);
                    xr0[q] = __uint_as_float(x.x);
                    xr1[q] = __uint_as_float(x.y);
                }
            } else {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    xr0[q] = (src + 2 * T * q)[lane2];
                    xr1[q] = (src + 2 * T * q)[lane2 + 1];
                }
            }
            const v2f cb = to_v(lds[T_H + 2 * tid]), sb = to_v(lds[T_H + 2 * tid + 1]);
            // stage 0 pairs registers brev5(q) and brev5(q + 16) = brev5(q) + 1: a +- b with a = x_q w_q and
            // b = x_{q+16} w_{q+16} is one multiply and two FMAs
            if constexpr (TABW) {
                GF wt2 = per_hop(p.window) + lane2;
#pragma unroll
                for (int q0 = 0; q0 < 16; q0 += 4) {
                    float l0[4], l1[4], h0[4], h1[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        l0[q] = (wt2 + 2 * T * (q0 + q))[0];
                        l1[q] = (wt2 + 2 * T * (q0 + q))[1];
                        h0[q] = (wt2 + 2 * T * (q0 + q + 16))[0];
                        h1[q] = (wt2 + 2 * T * (q0 + q + 16))[1];
                    }
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int q = q0 + qq;
                        const v2f a = v2f{xr0[q], xr1[q]} * v2f{l0[qq], l1[qq]}, xh = v2f{xr0[q + 16], xr1[q + 16]};
                        const v2f wh = v2f{h0[qq], h1[qq]};
                        v[2 * brev_c(q, 4)] = __builtin_elementwise_fma(xh, wh, a);
                        v[2 * brev_c(q, 4) + 1] = __builtin_elementwise_fma(-xh, wh, a);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const v2f wl = __builtin_elementwise_fma(v2f{HANN_W14.s[q], HANN_W14.s[q]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q], HANN_W14.c[q]}, cb, half2));
                const v2f wh = __builtin_elementwise_fma(v2f{HANN_W14.s[q + 16], HANN_W14.s[q + 16]}, sb,
                               __builtin_elementwise_fma(v2f{HANN_W14.c[q + 16], HANN_W14.c[q + 16]}, cb, half2));
                const v2f a = v2f{xr0[q], xr1[q]} * wl, xh = v2f{xr0[q + 16], xr1[q + 16]};
                v[2 * brev_c(q, 4)] = __builtin_elementwise_fma(xh, wh, a);
                v[2 * brev_c(q, 4) + 1] = __builtin_elementwise_fma(-xh, wh, a);
            }
            st.mark(0);
            dit_stages<32, m, 1, 4, 0, false, false>(v);
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
        if constexpr (TABW) {
            GF wt2 = per_hop(p.window) + lane2;
            const v2f kap = half2k + half2k;  // -1/(4N) (-1/(2N) with the product fold)
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
        } else
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
            } else {
                store_head(k, y);
            }
        }
#pragma unroll
        for (int q = 0; q < PH; ++q) tail[q] = y[q + PH];
        st.mark(25);
    }
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

}  // namespace

#ifndef RC_HOP4_TABW
#define RC_HOP4_TABW 1  // a caller's window at pitch 1 through hop4_kernel<1, false, true> (0: hop2_kernel, for A/B)
#endif
#ifndef RC_PITCHC
#define RC_PITCHC 1  // pitch 2 and 3 run instantiations with the pitch at compile time (0: the runtime-pitch kernel, for A/B)
#endif
// N = 16384, fused path: default hanning window -> hop4_kernel (optionally with the band mask in its pair stage),
// caller-supplied window -> hop2_kernel's table variant. The test-hook library (RC_TEST_HOOKS) can also run the
// previous generation (hop3_kernel, rc_hop16k_prev.hip) and hop2_kernel's computed-window variant for A/B runs.
hipError_t launch_hop16k(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(256);
    const bool hann = p.hann_rot != nullptr;
#if RC_TEST_HOOKS
    if (hann && (p.diag_flags & RC_DIAG_PREV_KERNEL)) return launch_hop16k_prev(p, s);
    if (hann && (p.diag_flags & RC_DIAG_HOP2_HANN)) {
        const size_t lds2 = sizeof(float2) * (size_t)HOP2_LDS_FLOAT2;
        if (p.pitch == 1) hipLaunchKernelGGL((hop2_kernel<true, true>), grid, block, lds2, s, p);
        else hipLaunchKernelGGL((hop2_kernel<false, true>), grid, block, lds2, s, p);
        return hipGetLastError();
    }
#endif
    if (hann) {
        const size_t lds4 = sizeof(float2) * (size_t)HOP4_LDS_FLOAT2;
        if (p.band_on) {
            if (p.pitch == 1) hipLaunchKernelGGL((hop4_kernel<1, true>), grid, block, lds4, s, p);
            else hipLaunchKernelGGL((hop4_kernel<0, true>), grid, block, lds4, s, p);
        } else if (p.pitch == 1) hipLaunchKernelGGL((hop4_kernel<1>), grid, block, lds4, s, p);
        else if (p.pitch == 2 && RC_PITCHC) hipLaunchKernelGGL((hop4_kernel<2>), grid, block, lds4, s, p);
        else if (p.pitch == 3 && RC_PITCHC) hipLaunchKernelGGL((hop4_kernel<3>), grid, block, lds4, s, p);
        else hipLaunchKernelGGL((hop4_kernel<0>), grid, block, lds4, s, p);
    } else {
        if (!p.window || !p.env) return hipErrorInvalidValue;  // (the table-window kernels dereference both)
        const size_t lds2 = sizeof(float2) * (size_t)HOP2_LDS_FLOAT2;
        // pitch 1: hop4 with table windows (three workgroups per CU, run tickets, seams: hop_workgroups_per_cu says so
        // to the planner); other pitches: hop2_kernel
        if (p.pitch == 1 && RC_HOP4_TABW)
            hipLaunchKernelGGL((hop4_kernel<1, false, true>), grid, block, sizeof(float2) * (size_t)HOP4_LDS_FLOAT2, s, p);
        else if (p.pitch == 1) hipLaunchKernelGGL((hop2_kernel<true, false>), grid, block, lds2, s, p);
        else hipLaunchKernelGGL((hop2_kernel<false, false>), grid, block, lds2, s, p);
    }
    return hipGetLastError();
}
}  // namespace rc
