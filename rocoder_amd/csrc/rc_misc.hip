// Small kernels around the hop kernels: gather-form overlap-add (user-kernel path, negative pitch), curated
// device frequency kernels, O(N^2) DFTs for window lengths that are not a power of two, the per-job prep launch.
#include "rc_dev.hpp"

namespace rc {

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

// ---- window lengths that are not a power of two (rc_kernels.h, launch_gen) -----------
#if !RC_BLUESTEIN
// O(N^2) DFTs: the first implementation, kept for A/B builds
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
#endif
// magnitudes x phasors in place on all N bins (between the two transforms)
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
#if !RC_BLUESTEIN
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
#endif

// ---- ... and chirp-z (Bluestein) transforms for the lengths whose packed half fits the LDS (N <= 16384) -----------
// The window is real and N even: z[n] = x[2n] + i x[2n+1] (M = N/2 points), Z = DFT_M(z) by
//   n k = (n^2 + k^2 - (k - n)^2) / 2  =>  Z[k] = c[k] sum_n (z[n] c[n]) conj(c)[k - n],  c[n] = exp(-i pi n^2 / M):
// a circular convolution of length L = 2^l >= 2M - 1 = one DIF FFT_L (natural -> bit-reversed), a pointwise product
// with the precomputed FFT_L of conj(c) (stored bit-reversed, 1/L folded in), one DIT inverse FFT_L (bit-reversed ->
// natural): no bit reversal anywhere. Then the real split X[j] = E[j] + W_N^j O[j] (src/fft.rs:50-61 computes the full
// complex N-point FFT of the real frame: the upper half is its conjugate mirror). The inverse takes the N resynthesised
// bins (independent phases: not Hermitian), keeps the Hermitian part - Re(IFFT(Z)) = IFFT((Z + conj(mirror Z)) / 2),
// src/fft.rs:69-73 - merges it to M packed bins and runs the same transform on the conjugate.
// One workgroup per hop; radix-2 stages in place in LDS (a path for odd window lengths of the CLI's -w, built for
// O(N log N) instead of O(N^2), not for the roofline).
constexpr uint32_t BL_BLOCK = 16384;  // points of an FFT_L that one workgroup holds in LDS

struct BlCtx {
    uint32_t N, M, l2, L;
    GV2 chirp, twl, bbr, twn;
    GF win;
    size_t row;   // first element of this (channel, hop) in spec / ybuf
    size_t rowl;  // ... in bl_wk
};
__device__ __forceinline__ BlCtx bl_ctx(const HopParams &p, int64_t hl, uint32_t ch) {
    BlCtx c;
    c.N = p.n_generic;
    c.M = c.N / 2;
    c.l2 = p.bl_log2l;
    c.L = 1u << c.l2;
    c.chirp = (GV2)p.bl_tab;
    c.twl = c.chirp + c.M;
    c.bbr = c.twl + c.L / 2;
    c.twn = (GV2)p.tw_generic;
    c.win = (GF)p.window;
    c.row = ((size_t)ch * p.hop_count + (size_t)hl) * c.N;
    c.rowl = ((size_t)ch * p.hop_count + (size_t)hl) * c.L;
    return c;
}
// element n of the sequence the forward / inverse transform convolves (zero from M on)
__device__ __forceinline__ float2 bl_in_fwd(const BlCtx &c, GF src, uint32_t n) {
    if (n >= c.M) return make_float2(0.f, 0.f);
    return cmul(make_float2(src[2 * n] * c.win[2 * n], src[2 * n + 1] * c.win[2 * n + 1]), ldg2(c.chirp + n));
}
__device__ __forceinline__ float2 bl_in_inv(const BlCtx &c, GV2 Z, uint32_t k) {
    if (k >= c.M) return make_float2(0.f, 0.f);
    const float2 z0 = ldg2(Z + k), z0m = ldg2(Z + (k ? c.N - k : 0)), z1 = ldg2(Z + k + c.M), z1m = ldg2(Z + c.M - k);
    // Hermitian part at k and k + M, then even / odd halves: E = (Zh[k] + Zh[k+M]) / 2, W^k O = (Zh[k] - Zh[k+M]) / 2
    const float2 h0 = make_float2(0.5f * (z0.x + z0m.x), 0.5f * (z0.y - z0m.y));
    const float2 h1 = make_float2(0.5f * (z1.x + z1m.x), 0.5f * (z1.y - z1m.y));
    const float2 E = make_float2(0.5f * (h0.x + h1.x), 0.5f * (h0.y + h1.y));
    const float2 D = make_float2(0.5f * (h0.x - h1.x), 0.5f * (h0.y - h1.y));
    const float2 w = ldg2(c.twn + k);
    const float2 O = cmul(D, make_float2(w.x, -w.y));
    const float2 G = make_float2(E.x - O.y, E.y + O.x);      // E + i O
    return cmul(make_float2(G.x, -G.y), ldg2(c.chirp + k));  // IDFT_M(G) = conj(DFT_M(conj G)) / M
}
__device__ __forceinline__ GF bl_src(const HopParams &p, int64_t hl, uint32_t ch) {
    const int64_t hop = p.hop_first + hl;
    GF xc = (GF)p.x + (size_t)ch * p.in_stride;
    GF xt = (GF)p.xtail + (size_t)ch * p.tail_stride;
    return (hop >= p.tail_hop_first) ? xt + (hop * (int64_t)p.step - p.tail_origin) : xc + (hop * (int64_t)p.step - p.in_origin);
}
// real split of Z = DFT_M(z) (zj = Z[j mod M], zm = Z[(M - j) mod M]) into bin j of the N-point spectrum and its mirror
__device__ __forceinline__ void bl_split_store(const BlCtx &c, GV2W X, uint32_t j, float2 zj, float2 zm) {
    const float2 E = make_float2(0.5f * (zj.x + zm.x), 0.5f * (zj.y - zm.y));
    const float2 O = make_float2(0.5f * (zj.y + zm.y), -0.5f * (zj.x - zm.x));  // (zj - conj zm) / (2 i)
    const float2 t = cmul(O, ldg2(c.twn + j));
    const float2 x = make_float2(E.x + t.x, E.y + t.y);
    stg2(X + j, x);
    if (j > 0 && j < c.M) stg2(X + (c.N - j), make_float2(x.x, -x.y));
}
// v = (convolution output)[n]: conj(v c[n]) / M = (y[2n], y[2n+1]) of IDFT_N; times the window (fft.rs:70-73)
__device__ __forceinline__ void bl_out_inv(const BlCtx &c, GFW y, uint32_t n, float2 v) {
    const float2 r = cmul(v, ldg2(c.chirp + n));
    const float inv = 1.0f / (float)c.M;
    y[2 * n] = r.x * inv * c.win[2 * n];
    y[2 * n + 1] = -r.y * inv * c.win[2 * n + 1];
}
// radix-2 stages s_hi .. s_lo (DIF, descending) or s_lo .. s_hi (DIT with conjugate twiddles, ascending) of an FFT_L on
// `cnt` points in LDS whose index bits above the block are `base` (twiddles depend on the index below the stage only)
template <bool DIT>
__device__ __forceinline__ void bl_lds_stages(float2 *a, uint32_t cnt, int s_lo, int s_hi, uint32_t l2, GV2 twl) {
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    for (int q = 0; q <= s_hi - s_lo; ++q) {
        const int s = DIT ? s_lo + q : s_hi - q;
        const uint32_t half = 1u << s;
        for (uint32_t b = tid; b < cnt / 2; b += T) {
            const uint32_t lo = b & (half - 1), i = ((b >> s) << (s + 1)) | lo, j = i + half;
            const float2 w = ldg2(twl + ((size_t)lo << (l2 - 1 - s)));
            if (DIT) {
                const float2 u = a[i], v = cmul(a[j], make_float2(w.x, -w.y));
                a[i] = make_float2(u.x + v.x, u.y + v.y);
                a[j] = make_float2(u.x - v.x, u.y - v.y);
            } else {
                const float2 u = a[i], v = a[j];
                a[i] = make_float2(u.x + v.x, u.y + v.y);
                a[j] = cmul(make_float2(u.x - v.x, u.y - v.y), w);
            }
        }
        __syncthreads();
    }
}

// L <= 16384: the whole transform in one workgroup
template <bool INV>
__global__ __launch_bounds__(1024) void bluestein_kernel(const HopParams p) {
    extern __shared__ __attribute__((aligned(16))) float2 a[];
    const int64_t hl = blockIdx.x;
    const uint32_t ch = blockIdx.y, T = blockDim.x, tid = threadIdx.x;
    const BlCtx c = bl_ctx(p, hl, ch);
    if (!INV) {
        GF src = bl_src(p, hl, ch);
        for (uint32_t n = tid; n < c.L; n += T) a[n] = bl_in_fwd(c, src, n);
    } else {
        GV2 Z = (GV2)p.spec + c.row;
        for (uint32_t k = tid; k < c.L; k += T) a[k] = bl_in_inv(c, Z, k);
    }
    __syncthreads();
    bl_lds_stages<false>(a, c.L, 0, (int)c.l2 - 1, c.l2, c.twl);  // DIF, natural -> bit-reversed
    for (uint32_t i = tid; i < c.L; i += T) a[i] = cmul(a[i], ldg2(c.bbr + i));
    __syncthreads();
    bl_lds_stages<true>(a, c.L, 0, (int)c.l2 - 1, c.l2, c.twl);   // DIT, bit-reversed -> natural
    if (!INV) {
        for (uint32_t k = tid; k < c.M; k += T) a[k] = cmul(a[k], ldg2(c.chirp + k));  // Z = DFT_M(z)
        __syncthreads();
        GV2W X = (GV2W)p.spec + c.row;
        for (uint32_t j = tid; j <= c.M; j += T) bl_split_store(c, X, j, a[j == c.M ? 0 : j], a[j == 0 ? 0 : c.M - j]);
    } else {
        GFW y = (GFW)p.ybuf + c.row;
        for (uint32_t n = tid; n < c.M; n += T) bl_out_inv(c, y, n, a[n]);
    }
}

// L = 2^LB x 16384 (N up to 65536): the top LB stages in registers through the work buffer, the rest per block in LDS
//   bl_top_kernel<INV, LB>   builds the sequence and runs DIF stages l2-1 .. 14     (grid: 64 x hops x channels, 256 threads)
//   bl_block_kernel          per 16384-point block: DIF 13 .. 0, x FFT(conj chirp), DIT 0 .. 13   (2^LB x hops x channels)
//   bl_bottom_kernel<INV,LB> DIT stages 14 .. l2-1; forward: Z = y c into the work buffer; inverse: the output samples
//   bl_split_kernel          forward only: real split of Z into the N bins
template <bool INV, int LB>
__global__ __launch_bounds__(256) void bl_top_kernel(const HopParams p) {
    constexpr uint32_t B = 1u << LB;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.z;
    const int64_t hl = blockIdx.y;
    const BlCtx c = bl_ctx(p, hl, ch);
    float2 v[B];
    if (!INV) {
        GF src = bl_src(p, hl, ch);
#pragma unroll
        for (uint32_t q = 0; q < B; ++q) v[q] = bl_in_fwd(c, src, i + q * BL_BLOCK);
    } else {
        GV2 Z = (GV2)p.spec + c.row;
#pragma unroll
        for (uint32_t q = 0; q < B; ++q) v[q] = bl_in_inv(c, Z, i + q * BL_BLOCK);
    }
#pragma unroll
    for (int t = LB - 1; t >= 0; --t) {  // stage s = 14 + t pairs q and q + 2^t
#pragma unroll
        for (uint32_t q = 0; q < B; ++q) {
            if (q & (1u << t)) continue;
            const uint32_t lo = i + (q & ((1u << t) - 1)) * BL_BLOCK;
            const float2 w = ldg2(c.twl + ((size_t)lo << (LB - 1 - t)));  // l2 - 1 - s = LB - 1 - t
            const float2 u = v[q], x = v[q | (1u << t)];
            v[q] = make_float2(u.x + x.x, u.y + x.y);
            v[q | (1u << t)] = cmul(make_float2(u.x - x.x, u.y - x.y), w);
        }
    }
    GV2W wk = (GV2W)p.bl_wk + c.rowl;
#pragma unroll
    for (uint32_t q = 0; q < B; ++q) stg2(wk + i + q * BL_BLOCK, v[q]);
}
__global__ __launch_bounds__(1024) void bl_block_kernel(const HopParams p) {
    extern __shared__ __attribute__((aligned(16))) float2 a[];
    const uint32_t blk = blockIdx.x, ch = blockIdx.z, T = blockDim.x, tid = threadIdx.x;
    const BlCtx c = bl_ctx(p, blockIdx.y, ch);
    GV2W wk = (GV2W)p.bl_wk + c.rowl + (size_t)blk * BL_BLOCK;
    for (uint32_t i = tid; i < BL_BLOCK; i += T) a[i] = ldg2((GV2)wk + i);
    __syncthreads();
    bl_lds_stages<false>(a, BL_BLOCK, 0, 13, c.l2, c.twl);
    for (uint32_t i = tid; i < BL_BLOCK; i += T) a[i] = cmul(a[i], ldg2(c.bbr + (size_t)blk * BL_BLOCK + i));
    __syncthreads();
    bl_lds_stages<true>(a, BL_BLOCK, 0, 13, c.l2, c.twl);
    for (uint32_t i = tid; i < BL_BLOCK; i += T) stg2(wk + i, a[i]);
}
template <bool INV, int LB>
__global__ __launch_bounds__(256) void bl_bottom_kernel(const HopParams p) {
    constexpr uint32_t B = 1u << LB;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.z;
    const BlCtx c = bl_ctx(p, blockIdx.y, ch);
    GV2W wk = (GV2W)p.bl_wk + c.rowl;
    float2 v[B];
#pragma unroll
    for (uint32_t q = 0; q < B; ++q) v[q] = ldg2((GV2)wk + i + q * BL_BLOCK);
#pragma unroll
    for (int t = 0; t < LB; ++t) {
#pragma unroll
        for (uint32_t q = 0; q < B; ++q) {
            if (q & (1u << t)) continue;
            const uint32_t lo = i + (q & ((1u << t) - 1)) * BL_BLOCK;
            const float2 w = ldg2(c.twl + ((size_t)lo << (LB - 1 - t)));
            const float2 u = v[q], x = cmul(v[q | (1u << t)], make_float2(w.x, -w.y));
            v[q] = make_float2(u.x + x.x, u.y + x.y);
            v[q | (1u << t)] = make_float2(u.x - x.x, u.y - x.y);
        }
    }
#pragma unroll
    for (uint32_t q = 0; q < B; ++q) {
        const uint32_t n = i + q * BL_BLOCK;
        if (n >= c.M) continue;
        if (!INV) stg2(wk + n, cmul(v[q], ldg2(c.chirp + n)));  // (every thread rewrites only what it read)
        else bl_out_inv(c, (GFW)p.ybuf + c.row, n, v[q]);
    }
}
__global__ __launch_bounds__(256) void bl_split_kernel(const HopParams p) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x, ch = blockIdx.z;
    const BlCtx c = bl_ctx(p, blockIdx.y, ch);
    if (j > c.M) return;
    GV2 Z = (GV2)p.bl_wk + c.rowl;
    bl_split_store(c, (GV2W)p.spec + c.row, j, ldg2(Z + (j == c.M ? 0 : j)), ldg2(Z + (j == 0 ? 0 : c.M - j)));
}
template <int LB>
void launch_bl_large(int stage, const HopParams &q, unsigned ny, hipStream_t s) {
    const dim3 gtop(BL_BLOCK / 256, ny, q.n_channels), gblk(1u << LB, ny, q.n_channels);
    if (stage == 0) hipLaunchKernelGGL((bl_top_kernel<false, LB>), gtop, dim3(256), 0, s, q);
    else hipLaunchKernelGGL((bl_top_kernel<true, LB>), gtop, dim3(256), 0, s, q);
    hipLaunchKernelGGL(bl_block_kernel, gblk, dim3(1024), sizeof(float2) * (size_t)BL_BLOCK, s, q);
    if (stage == 0) {
        hipLaunchKernelGGL((bl_bottom_kernel<false, LB>), gtop, dim3(256), 0, s, q);
        hipLaunchKernelGGL(bl_split_kernel, dim3((q.n_generic / 2 + 256) / 256, ny, q.n_channels), dim3(256), 0, s, q);
    } else {
        hipLaunchKernelGGL((bl_bottom_kernel<true, LB>), gtop, dim3(256), 0, s, q);
    }
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
        if (p.bl_log2l > 16) return hipErrorInvalidValue;
        if (p.bl_log2l > 14 && stage != 1) {
            q.bl_wk = p.bl_wk ? p.bl_wk + ((size_t)h0 << p.bl_log2l) : nullptr;
            if (!q.bl_wk) return hipErrorInvalidValue;
            if (p.bl_log2l == 15) launch_bl_large<1>(stage, q, ny, s);
            else launch_bl_large<2>(stage, q, ny, s);
        } else if (p.bl_log2l && stage != 1) {
            const uint32_t L = 1u << p.bl_log2l;
            const dim3 bgrid(ny, p.n_channels), bblock(L / 2 < 64 ? 64 : (L / 2 > 1024 ? 1024 : L / 2));
            if (stage == 0) hipLaunchKernelGGL(bluestein_kernel<false>, bgrid, bblock, sizeof(float2) * (size_t)L, s, q);
            else hipLaunchKernelGGL(bluestein_kernel<true>, bgrid, bblock, sizeof(float2) * (size_t)L, s, q);
        } else if (stage == 1) hipLaunchKernelGGL(gen_phase_kernel, grid, block, 0, s, q);
#if !RC_BLUESTEIN
        else if (stage == 0) hipLaunchKernelGGL(gen_fwd_kernel, grid, block, 0, s, q);
        else hipLaunchKernelGGL(gen_inv_kernel, grid, block, 0, s, q);
#else
        else return hipErrorInvalidValue;  // (the engine builds the chirp-z tables for every such length)
#endif
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

__global__ __launch_bounds__(256) void resample_slower_kernel(const ResampleParams p) {
    const int64_t hl = blockIdx.x;
    const uint32_t ch = blockIdx.y, f = p.f;
    GF o = (GF)p.obuf + (size_t)ch * p.o_stride + (size_t)hl * p.half;
    GFW dst = (GFW)p.out + (size_t)ch * p.out_stride + ((p.hop_first + hl) * (int64_t)p.window_out_len - p.out_origin);
    for (uint32_t m = threadIdx.x; m < p.window_out_len; m += blockDim.x) {
        const uint32_t i = m / f, j = m - i * f;
        const float cur = o[i], nxt = o[i + 1];
        dst[m] = cur + (nxt - cur) * ((float)j / (float)f);  // math::lerp, src/math.rs:28-30
    }
}
hipError_t launch_resample_slower(const ResampleParams &p, hipStream_t s) {
    const int64_t per = 1 << 20;
    for (int64_t h0 = 0; h0 < p.hop_count; h0 += per) {
        ResampleParams q = p;
        q.obuf = p.obuf + (size_t)h0 * p.half;
        q.hop_first = p.hop_first + h0;
        q.hop_count = std::min<int64_t>(per, p.hop_count - h0);
        hipLaunchKernelGGL(resample_slower_kernel, dim3((unsigned)q.hop_count, p.n_channels), dim3(256), 0, s, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Box calibration (rc_calib_valu, measurement support): CALIB_ITERS x 16 independent v_pk_fma_f32 per wave, 256
// threads x 8 workgroups per CU = eight waves per SIMD, no memory traffic - the instruction the fused kernels are
// bound by, at the occupancy where its issue rate is flat (tools/valurate.hip is the full table).
__global__ __launch_bounds__(256) void calib_valu_kernel(float *out, float seed) {
    v2f pk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) pk[i] = v2f{seed + 0.001f * (float)(threadIdx.x + i), seed};
    for (int it = 0; it < CALIB_ITERS / 4; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(pk[i]));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += pk[i].x + pk[i].y;
    if (acc == 12345.678f) out[0] = acc;  // (never true: keeps the chains alive)
}
hipError_t launch_calib_valu(float *d_out, int n_cu, hipStream_t s) {
    hipLaunchKernelGGL(calib_valu_kernel, dim3((unsigned)n_cu * 8), dim3(256), 0, s, d_out, 1.0f);
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
