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
