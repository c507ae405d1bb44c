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
#include "rc_passes.hpp"
#ifndef RC_HOPW
#define RC_HOPW 31  // default window: bit 0 N = 4096 runs hopw_kernel, bit 1 N = 8192 hopw2_kernel, bit 2 N = 2048 hopw11_kernel, bit 3 N = 1024 hopw10_kernel, bit 4 N = 512 hopw9_kernel (0: generic, for A/B)
#endif
#include "rc_dit.hpp"  // (the constexpr sine / cosine of the computed-window constants)

namespace rc {
namespace {

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
            // the pair in packed (re, im) arithmetic (rc_dit.hpp, as the N = 16384 kernels): 17 v_pk_* + 10
            // transcendental + the two hashes instead of ~50 scalar operations
            const uint32_t x1 = jb * key.mul + key.k0;
            v2f VAv, VBv;
            pair_regs_pk<LOG2N>(to_v(A), to_v(Bp), to_v(ww), x1, key, VAv, VBv);
            const float2 VA = to_f2(VAv), VB = to_f2(VBv);
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

// Default-window fast path (as HANN_W14 / HANN_E14 of the N = 16384 kernels, for every power of two below): thread t
// touches samples i = 2 T q + 2 t + e, and both windows::hanning (src/windows.rs:4-9) and the crossfade envelope
// (src/crossfade.rs:4-10) are 0.5 - a cos(2 pi i / (len - 1)) = 0.5 + c[q] cos(beta) + s[q] sin(beta) with
// beta = 2 pi (2 t + e) / (len - 1) from HopParams::hann_rot (4 + 4 floats per thread and hop) and compile-time c, s:
// two FMAs per sample pair replace a table load - 5 P loads per hop that were waited on in place.
template <int LOG2N>
struct HannG {
    static constexpr int P = Geo<LOG2N>::P;
    float wc[P], ws[P], ec[P], es[P];
};
template <int LOG2N>
constexpr HannG<LOG2N> make_hann_g() {
    using G = Geo<LOG2N>;
    HannG<LOG2N> k{};
    for (int q = 0; q < G::P; ++q) {
        const double aw = 2.0 * CX_PI * (2.0 * G::T * q) / (double)(G::N - 1);
        k.wc[q] = (float)(-0.5 * cx_cos(aw));
        k.ws[q] = (float)(0.5 * cx_sin(aw));
        const double ae = q < G::P / 2 ? 2.0 * CX_PI * (2.0 * G::T * q) / (double)(G::M - 1) : 0.0;
        k.ec[q] = (float)(-HANN_ENV_AMP * cx_cos(ae));
        k.es[q] = (float)(HANN_ENV_AMP * cx_sin(ae));
    }
    return k;
}
template <int LOG2N>
__device__ constexpr HannG<LOG2N> HANN_G = make_hann_g<LOG2N>();

// Hop slots (round 4): below N = 512 a hop takes fewer than 64 threads (T = 32 / 16 / 8 / 4 at N = 256 / 128 / 64 / 32), and a
// workgroup of T threads left most of its one wave idle. The fused path now packs SLOTS = 64 / T runs into one wave:
// thread = (slot, t), every slot its own run of hops, LDS region, phase keys and source pointers (per lane: nothing is
// forced uniform); the workgroup is one wave, so its barriers only order the wave's own LDS traffic.
template <int LOG2N, int MODE>
constexpr int hop_slots_of() { return (MODE == MODE_FUSED && Geo<LOG2N>::T < 64) ? 64 / Geo<LOG2N>::T : 1; }

template <int LOG2N, int MODE, bool PITCH1, bool HANN = false>
__global__ __launch_bounds__((Geo<LOG2N>::T * hop_slots_of<LOG2N, MODE>()), Geo<LOG2N>::WPS) void hop_kernel(const HopParams p) {
    using G = Geo<LOG2N>;
    constexpr int P = G::P, T = G::T, M = G::M, N = G::N, H = M;
    constexpr int LL = last_lor<G>(G::m);
    constexpr int SLOTS = hop_slots_of<LOG2N, MODE>();
    extern __shared__ __attribute__((aligned(16))) float2 lds_all[];
    ThreadCtx<G> ctx;
    ctx.tid = SLOTS > 1 ? (int)(threadIdx.x % T) : (int)threadIdx.x;
    fill_lds_bases<G, G::m>(ctx);
    const int tid = ctx.tid;
    const uint32_t slot = SLOTS > 1 ? threadIdx.x / T : 0u;
    float2 *lds = lds_all + (size_t)slot * G::LDS_FLOAT2;
    uint32_t gidx = blockIdx.x;  // run index over all channels
    bool live = true;
    if constexpr (SLOTS > 1) {
        gidx = blockIdx.x * SLOTS + slot;
        live = gidx < p.runs_per_channel * p.n_channels;
        if (!live) gidx = 0;
    }
    const uint32_t run = gidx % p.runs_per_channel;
    const uint32_t ch = gidx / p.runs_per_channel;
    int64_t k_begin_ = p.hop_first + (int64_t)run * p.run_len;
    int64_t k_end = k_begin_ + p.run_len;
    if (k_end > p.hop_first + p.hop_count) k_end = p.hop_first + p.hop_count;
    if constexpr (SLOTS > 1) {
        if (!live || k_begin_ >= k_end) k_begin_ = k_end = 0;  // an empty slot runs no iteration, not even the recomputed
                                                               // predecessor hop (its lanes stay with the wave)
    } else {
        if (k_begin_ >= k_end) return;
    }
    const int64_t k_begin = k_begin_;
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
            __syncthreads();
            middle_stage<LOG2N, MODE_FORWARD>(lds, tid, PhaseKey{0u, 1u}, rtab, spec);
            __syncthreads();
        }
    } else if constexpr (MODE == MODE_RESYNTH) {
        for (int64_t k = k_begin; k < k_end; ++k) {
            const size_t hop_idx = (size_t)ch * p.hop_count + (size_t)(k - p.hop_first);
            GV2W spec = (GV2W)p.spec + hop_idx * N;
            const PhaseKey key = make_phase_key(p.seed_mixed, p.ch_first + ch, k);
            middle_stage<LOG2N, MODE_RESYNTH>(lds, tid, key, rtab, spec);
            __syncthreads();
            lds_load<G, LL>(v, lds, ctx.lb[LL]);
            __syncthreads();
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
            v2f cbW = {0.f, 0.f}, sbW = cbW, cbE = cbW, sbE = cbW;
            const v2f half2 = {0.5f, 0.5f};
            if constexpr (HANN) {  // this thread's {cos, sin}(beta) for the window and the envelope (per hop: L1 hits)
                int t2 = tid;
                opaque(t2);
                GV2 hr = (GV2)per_hop(p.hann_rot) + 2 * t2;
                const float2 a0 = ldg2(hr), a1 = ldg2(hr + 1), e0 = ldg2(hr + 2 * T), e1 = ldg2(hr + 2 * T + 1);
                cbW = v2f{a0.x, a1.x};
                sbW = v2f{a0.y, a1.y};
                cbE = v2f{e0.x, e1.x};
                sbE = v2f{e0.y, e1.y};
                GF src = SLOTS > 1 ? hop_src_lane(p, xc, xt, k) : hop_src(p, xc, xt, k);
                float xr0[P], xr1[P];
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    xr0[q] = (src + 2 * T * q)[lane2];
                    xr1[q] = (src + 2 * T * q)[lane2 + 1];
                }
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const v2f wq = __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.ws[q], HANN_G<LOG2N>.ws[q]}, sbW,
                                   __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.wc[q], HANN_G<LOG2N>.wc[q]}, cbW, half2));
                    v[q] = to_f2(v2f{xr0[q], xr1[q]} * wq);
                }
            } else if constexpr (SLOTS > 1) {  // load_hop with a per-lane source pointer (the slots' hops differ)
                GF src = hop_src_lane(p, xc, xt, k);
                GF win = per_hop(p.window);
#pragma unroll
                for (int q = 0; q < P; ++q)
                    v[q] = make_float2((src + 2 * T * q)[lane2] * (win + 2 * T * q)[lane2],
                                       (src + 2 * T * q)[lane2 + 1] * (win + 2 * T * q)[lane2 + 1]);
            } else {
                load_hop<LOG2N>(v, p, xc, xt, per_hop(p.window), k, lane2);
            }
            st.mark(0);
            forward_passes<G, G::m, 0, true>(v, lds, ctx, wtab, st);
            lds_store<G, LL>(v, lds, ctx.lb[LL]);
            __syncthreads();
            st.mark(12);
            middle_fused<LOG2N>(lds, tid, key, rtab);
            st.mark(13);
            __syncthreads();
            st.mark(14);
            lds_load<G, LL>(v, lds, ctx.lb[LL]);
            __syncthreads();
            st.mark(15);
            inverse_passes<G, G::m>(v, lds, ctx, wtab, st);
            if constexpr (HANN) {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const v2f wq = __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.ws[q], HANN_G<LOG2N>.ws[q]}, sbW,
                                   __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.wc[q], HANN_G<LOG2N>.wc[q]}, cbW, half2));
                    v[q] = to_f2(to_v(v[q]) * wq);
                }
            } else {
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
            if (k >= k_begin) {
                const int64_t g0 = k * (int64_t)H;  // absolute O index of this hop's first sample
                GF esrc = per_hop(p.env);
                if constexpr (PITCH1) {
                    GFW dst = outc + (g0 - p.out_origin);
                    float er0[PH], er1[PH];
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        if constexpr (HANN) {
                            const v2f ev = __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.es[q], HANN_G<LOG2N>.es[q]}, sbE,
                                           __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.ec[q], HANN_G<LOG2N>.ec[q]}, cbE, half2));
                            er0[q] = ev.x, er1[q] = ev.y;
                        } else {
                            er0[q] = (esrc + 2 * T * q)[lane2];
                            er1[q] = (esrc + 2 * T * q)[lane2 + 1];
                        }
                    }
                    if constexpr (!HANN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < PH; ++q) {
                        float2 o;  // stretcher.rs:97-100 operation order
                        o.x = (v[q].x + tail[q].x) * er0[q] * p.amp;
                        o.y = (v[q].y + tail[q].y) * er1[q] * p.amp;
                        // (non-temporal: the output is written once and never read here; the windows of neighbouring
                        // hops keep the L2)
                        __builtin_nontemporal_store(to_v(o), (GV2W)(dst + 2 * T * q + lane2));
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
                        float ex, ey;
                        if constexpr (HANN) {
                            const v2f ev = __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.es[q], HANN_G<LOG2N>.es[q]}, sbE,
                                           __builtin_elementwise_fma(v2f{HANN_G<LOG2N>.ec[q], HANN_G<LOG2N>.ec[q]}, cbE, half2));
                            ex = ev.x, ey = ev.y;
                        } else {
                            ex = (esrc + 2 * T * q)[lane2], ey = (esrc + 2 * T * q)[lane2 + 1];
                        }
                        const float o0 = (v[q].x + tail[q].x) * ex * p.amp;
                        const float o1 = (v[q].y + tail[q].y) * ey * p.amp;
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

template <int LOG2N>
hipError_t launch_hop_n(HopMode mode, const HopParams &p, hipStream_t s) {
    using G = Geo<LOG2N>;
    constexpr int SF = hop_slots_of<LOG2N, MODE_FUSED>();  // the fused path packs SF runs into one wave (N < 512)
    const uint32_t total_runs = p.runs_per_channel * p.n_channels;
    const dim3 grid(mode == MODE_FUSED ? (total_runs + SF - 1) / SF : total_runs), block(mode == MODE_FUSED ? G::T * SF : G::T);
    const size_t lds = sizeof(float2) * G::LDS_FLOAT2 * (mode == MODE_FUSED ? SF : 1);
    switch (mode) {
        case MODE_FUSED:
            if constexpr (LOG2N == 14) return launch_hop16k(p, s);  // hop4_kernel / hop2_kernel (rc_hop16k.hip)
            else {
                // 512 ... 8192: the wave-local kernels (rc_hopw.hip), default window or a caller's; the generic kernel's
                // fused instantiations then serve the shorter lengths only
                constexpr bool WL = (LOG2N == 12 && (RC_HOPW & 1)) || (LOG2N == 13 && (RC_HOPW & 2)) || (LOG2N == 11 && (RC_HOPW & 4)) ||
                                    (LOG2N == 10 && (RC_HOPW & 8)) || (LOG2N == 9 && (RC_HOPW & 16));
                if constexpr (WL) {
                    // (round 5: a caller's window too - the kernels' TABW instantiations read its tables)
                    if constexpr (LOG2N == 12) return launch_hopw(p, s);     // one wave per hop
                    if constexpr (LOG2N == 13) return launch_hopw2(p, s);    // two waves per hop
                    if constexpr (LOG2N == 11) return launch_hopw11(p, s);   // one wave, 16 points per lane
                    if constexpr (LOG2N == 10) return launch_hopw10(p, s);   // two hops per wave
                    if constexpr (LOG2N == 9) return launch_hopw9(p, s);     // two hops per wave, 8 points per lane
                } else {
                    if (p.hann_rot && p.pitch == 1) {
                        hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, true, true>), grid, block, lds, s, p);
                        break;
                    }
                    if (p.hann_rot) {
                        hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, false, true>), grid, block, lds, s, p);
                        break;
                    }
                }
                if (p.pitch == 1) hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, true>), grid, block, lds, s, p);
                else hipLaunchKernelGGL((hop_kernel<LOG2N, MODE_FUSED, false>), grid, block, lds, s, p);
            }
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
}  // namespace

#ifndef RC_HOP4_TABW
#define RC_HOP4_TABW 1
#endif
int hop_workgroups_per_cu(int log2n, bool default_window, bool pitch1) {
    // hop4_kernel: three workgroups per CU - the default window at any pitch, a caller's window at pitch 1 (round 5)
    return (log2n == 14 && (default_window || (pitch1 && RC_HOP4_TABW))) ? 3 : 0;
}

int hop_slots(int log2n) {  // runs per workgroup of the fused generic kernel (hop_kernel: one wave holds 64 / T runs below N = 512)
    if (log2n < 5 || log2n > 8) return 1;
    const int M = 1 << (log2n - 1), T = M <= 128 ? cmax(2, M / 8) : cmax(M / RC_PMAX, cmin(64, M / 4));
    return T < 64 ? 64 / T : 1;
}

int hop_resident_workgroups(int log2n, bool default_window) {
    (void)default_window;  // (round 5: the wave-local kernels serve a caller's window as well)
    if (log2n == 12 && (RC_HOPW & 1)) return 12;  // hopw_kernel: one wave each, three per SIMD
    if (log2n == 13 && (RC_HOPW & 2)) return 6;   // hopw2_kernel: two waves each
#ifndef RC_HOPW11_RES
#define RC_HOPW11_RES 12  // (16 are resident at 112 VGPRs; 24 runs per CU measured best: 0.635 against 0.65 ms with 16 / 32 / 48)
#endif
    if (log2n == 11 && (RC_HOPW & 4)) return RC_HOPW11_RES;  // hopw11_kernel: one wave each
    if (log2n == 10 && (RC_HOPW & 8)) return 16;  // hopw10_kernel: one wave (two hops at a time) each, four per SIMD
    if (log2n == 9 && (RC_HOPW & 16)) return 16;  // hopw9_kernel
    return 0;
}

bool hop_geometry(int log2n, int *threads, size_t *lds_bytes) {
    if (log2n < 5 || log2n > 14) return false;
    const int m = log2n - 1, M = 1 << m;
    const int T = M <= 128 ? cmax(2, M / 8) : cmax(M / RC_PMAX, cmin(64, M / 4));
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
}  // namespace rc
