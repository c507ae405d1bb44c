"""numpy prototype of the HIP hop kernel's index math (dev tool, not shipped, not a test oracle).

Emulates one workgroup: T threads x P registers, bit-group DIF passes forward (natural in,
bit-reversed out), the position-based r2c/phase/c2r "quad" middle stage on the bit-reversed
spectrum in (padded) LDS, and mirrored DIT passes inverse. Checks against oracle_np.resynth.
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from oracle import oracle_np as onp  # noqa: E402


def geometry(log2n):
    m = log2n - 1
    M = 1 << m
    T = max(M // 32, min(64, M // 4))
    P = M // T
    B = P.bit_length() - 1
    return m, M, T, P, B


def pass_list(m, B):
    out = []
    prev = m
    while prev > 0:
        lo = max(0, prev - B)
        # register bits [lo_r, lo_r+B-1]; for the last (short) pass register bits start at 0
        lo_r = lo if lo + B <= m else max(0, m - B)
        lo_r = min(lo_r, lo)
        lo_r = lo if prev - lo == B else 0
        out.append((lo_r, lo, prev - 1))  # (register LO, transformed bits lo..hi)
        prev = lo
    return out


def pos_of(tid, q, LO, B):
    l = tid & ((1 << LO) - 1)
    u = tid >> LO
    return (u << (LO + B)) | (q << LO) | l


def pad(n):
    return n + (n >> 5)


def brev(x, bits):
    r = 0
    for b in range(bits):
        if x & (1 << b):
            r |= 1 << (bits - 1 - b)
    return r


def run_pass(v, tid, m, B, LO, s_lo, s_hi, inverse):
    """v: [T, P] complex; transform absolute bits s_lo..s_hi (register bit r = s - LO)."""
    P = 1 << B
    l = tid & ((1 << LO) - 1)
    order = range(s_hi, s_lo - 1, -1) if not inverse else range(s_lo, s_hi + 1)
    for s in order:
        r = s - LO
        half = 1 << r
        base = np.exp(-2j * np.pi * l / (1 << (s + 1)))  # per-thread base twiddle
        for q0 in range(P):
            if q0 & half:
                continue
            c = q0 & (half - 1)
            K = np.exp(-2j * np.pi * c / (1 << (r + 1)))
            w = base * K
            a = v[:, q0].copy()
            b = v[:, q0 | half].copy()
            if not inverse:
                v[:, q0] = a + b
                v[:, q0 | half] = (a - b) * w
            else:
                t = np.conj(w) * b
                v[:, q0] = a + t
                v[:, q0 | half] = a - t
    return v


def exchange(v, tid, B, LO_from, LO_to, lds):
    P = 1 << B
    for q in range(P):
        lds[pad(pos_of(tid, q, LO_from, B))] = v[:, q]
    out = np.empty_like(v)
    for q in range(P):
        out[:, q] = lds[pad(pos_of(tid, q, LO_to, B))]
    return out


def pair(A, Bp, w, ja, key, N, M, kappa):
    """one (ja, M-ja) pair: returns V[ja], V[M-ja]."""
    E = (A + np.conj(Bp))          # 2E
    D = (A - np.conj(Bp))          # 2D
    Tt = w * D
    X1 = E - 1j * Tt               # 2 X[ja]
    X2c = E + 1j * Tt              # 2 conj(X[M-ja])
    m1, m2 = np.abs(X1), np.abs(X2c)
    th = lambda b: onp.phase_theta(key, np.asarray(b) % N, N).astype(np.float64)  # noqa: E731
    t1, t2, t3, t4 = th(ja), th(N - ja), th(M - ja), th(M + ja)
    Pz = (m1 * kappa) * ((np.cos(t1) + np.cos(t2)) + 1j * (np.sin(t1) - np.sin(t2)))
    Q = (m2 * kappa) * ((np.cos(t3) + np.cos(t4)) + 1j * (np.sin(t4) - np.sin(t3)))
    S, R = Pz + Q, Pz - Q
    U = np.conj(w) * R
    VA = S + 1j * U
    VB = np.conj(S - 1j * U)
    return VA, VB


def hop(x, window, key, log2n):
    m, M, T, P, B = geometry(log2n)
    N = 2 * M
    tid = np.arange(T)
    passes = pass_list(m, B)
    lds = np.zeros(M + (M >> 5) + 1, np.complex128)
    # load: pass 0 layout (LO = passes[0][0])
    LO0 = passes[0][0]
    v = np.empty((T, P), np.complex128)
    for q in range(P):
        n = pos_of(tid, q, LO0, B)
        v[:, q] = x[2 * n] * window[2 * n] + 1j * x[2 * n + 1] * window[2 * n + 1]
    for i, (LO, s_lo, s_hi) in enumerate(passes):
        if i > 0:
            v = exchange(v, tid, B, passes[i - 1][0], LO, lds)
        v = run_pass(v, tid, m, B, LO, s_lo, s_hi, False)
    LOl = passes[-1][0]
    for q in range(P):
        lds[pad(pos_of(tid, q, LOl, B))] = v[:, q]
    # check forward: position p holds Zf[brev(p)]
    z = x[0::2][:M] * window[0::2][:M] + 1j * x[1::2][:M] * window[1::2][:M]
    Zf = np.fft.fft(z)
    got = np.array([lds[pad(p)] for p in range(M)])
    ref = np.array([Zf[brev(p, m)] for p in range(M)])
    assert np.allclose(got, ref, atol=1e-9 * M), "forward mismatch"
    # middle: quads
    kappa = 1.0 / (4.0 * N)
    rtab = np.exp(-2j * np.pi * np.arange(M // 4 + 1) / N)
    nq = max(1, (M // 4) // T)
    for s in range(nq):
        for t in range(T):
            c = t + T * s
            if c >= M // 4 or c == 0:
                continue
            j = brev(2 * c, m - 1)
            j2 = M // 2 - j
            p1 = 4 * c
            assert p1 == brev(j, m)
            p2 = brev(j2, m)
            A1, A2, B1, B2 = lds[pad(p1)], lds[pad(p1 + 1)], lds[pad(p2)], lds[pad(p2 + 1)]
            w = rtab[j]
            VA, VB = pair(A1, B2, w, j, key, N, M, kappa)
            lds[pad(p1)], lds[pad(p2 + 1)] = VA, VB
            w2 = -1j * np.conj(w)
            VA, VB = pair(B1, A2, w2, j2, key, N, M, kappa)
            lds[pad(p2)], lds[pad(p1 + 1)] = VA, VB
    # special block (thread 0): bins 0/Nyquist (pos 0), M/2 (pos 1), pair (M/4, 3M/4) (pos 2, 3)
    Z0, Zh = lds[pad(0)], lds[pad(1)]
    VA, _ = pair(Z0, Z0, 1.0 + 0j, 0, key, N, M, kappa)
    lds[pad(0)] = VA
    VA, _ = pair(Zh, Zh, -1j, M // 2, key, N, M, kappa)
    lds[pad(1)] = VA
    if M >= 4:
        Zq, Z3q = lds[pad(2)], lds[pad(3)]
        VA, VB = pair(Zq, Z3q, rtab[M // 4], M // 4, key, N, M, kappa)
        lds[pad(2)], lds[pad(3)] = VA, VB
    # inverse passes (mirror)
    for q in range(P):
        v[:, q] = lds[pad(pos_of(tid, q, LOl, B))]
    for i in range(len(passes) - 1, -1, -1):
        LO, s_lo, s_hi = passes[i]
        v = run_pass(v, tid, m, B, LO, s_lo, s_hi, True)
        if i > 0:
            v = exchange(v, tid, B, LO, passes[i - 1][0], lds)
    y = np.empty(N)
    for q in range(P):
        n = pos_of(tid, q, LO0, B)
        y[2 * n] = v[:, q].real * window[2 * n]
        y[2 * n + 1] = v[:, q].imag * window[2 * n + 1]
    return y


if __name__ == "__main__":
    for log2n in range(3, 15):
        N = 1 << log2n
        m, M, T, P, B = geometry(log2n)
        x = onp.synth_input(2, N).astype(np.float64)
        w = onp.hanning(N).astype(np.float64)
        key = onp.phase_key(0x5EED, 1, 3)
        y = hop(x, w, key, log2n)
        ref = onp.resynth(x, w.astype(np.float32), key)
        err = np.sqrt(np.mean((y - ref) ** 2))
        print(f"N={N} T={T} P={P} passes={pass_list(m, B)} rms_err={err:.3e} rms={np.sqrt(np.mean(ref**2)):.3e}")
        assert err < 1e-9
