// Previous kernel generation of the N = 16384 default-window path (hop3_kernel, round 1's bench kernel). Built
// only into the test-hook library (make hooks): the A/B partner of hop4_kernel under ROCODER_DIAG=2.
#include "rc_dit.hpp"

namespace rc {
namespace {

// =============== v3: hop2's math with two-round exchanges, three workgroups per CU ===============
// The exchange buffer of hop2_kernel (8192 complex = 66 KiB) limits a CU to two workgroups. Here
// every exchange runs in two rounds over a HALF buffer (4096 complex): round A moves the elements
// whose position has a chosen bit (4 for exchanges 1 and 3, 8 for 2 and 4) clear, round B the others;
// the first store of exchanges 2 and 4 is in place (the thread overwrites what it read last). 46 KiB
// of LDS and <= 168 VGPRs per workgroup: three workgroups = 3 waves per SIMD, at 14 barriers per hop
// instead of 6. Default (hanning) window only; no software pipelining (no registers for it).
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


}  // namespace

hipError_t launch_hop16k_prev(const HopParams &p, hipStream_t s) {
    const dim3 grid(p.runs_per_channel * p.n_channels), block(256);
    const size_t lds3 = sizeof(float2) * (size_t)HOP3_LDS_FLOAT2;
    if (p.pitch == 1) hipLaunchKernelGGL((hop3_kernel<true>), grid, block, lds3, s, p);
    else hipLaunchKernelGGL((hop3_kernel<false>), grid, block, lds3, s, p);
    return hipGetLastError();
}
}  // namespace rc
