// Large windows (N = 32768 / 65536), quarter-FFT pipeline through HBM scratch: big_a / big_b / big_c / big_cr.
// Used by the spectrum paths (host and device frequency kernels) and negative pitch multiples; the plain
// stretch runs the fused big4_kernel (rc_big4.hip). DESIGN.md 5.3.
#include "rc_passes.hpp"

namespace rc {
namespace {

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

}  // namespace rc
