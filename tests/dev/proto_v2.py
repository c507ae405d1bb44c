"""numpy prototype of the v2 hop kernel (dev tool): both transforms DIT, middle stage in registers.

M = 2^m complex points, T = M/32 threads, 32 registers per thread.
Forward (DIT, bit-reversed input realised by the LOAD ORDER, natural output):
  F1 stages 0..4    registers = position bits 0..4, thread t plays position-thread u = brev(t)
  F2 stages 5..m-5  registers = position bits 4..8 (bit 4 passive)      [LOR = 4 layout]
  F3 stages m-4..m-1 two groups of 16: residues r = t and RB = 512 - t (mod 512) [natural bins]
middle in registers: pair (A[q], B[15-q]); thread 0 (residues 0 and 256) via a wave-0 side path
Inverse (DIT from bit-reversed positions p = brev(j)):
  I1 stages 0..3 in registers (register index brev4(q)), I2 stages 4..8, I3 stages 9..12 (bit 8 passive)
LDS index maps: f(n) = n + (n>>5) and f3(n) = n + (n>>5) + (n>>8); bank conflicts are counted.
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 3)[0])
from oracle import oracle_np as onp  # noqa: E402


def brev(x, bits):
    x = np.asarray(x)
    r = np.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def f1(n):
    return n + (n >> 5)


def f3(n):
    return n + (n >> 5) + (n >> 8)


CONFLICTS = {}


def count_conflicts(name, idx):
    """idx: [T] LDS complex index per lane for one wave-instruction (ds_*_b64): per 32-lane half,
    extra cycles = max multiplicity of (idx mod 32) - 1."""
    idx = np.asarray(idx)
    worst = 0
    for h in range(0, idx.size, 32):
        banks = idx[h:h + 32] % 32
        worst = max(worst, np.bincount(banks, minlength=32).max() - 1)
    CONFLICTS[name] = max(CONFLICTS.get(name, 0), worst)


def dit_stages(v, l, s_lo, s_hi, reg_lo, m, conj):
    """v: [T, R]; register bit (s - reg_lo) <-> position bit s; lower position bits: runtime l (bits
    below reg_lo) plus lower register bits. DIT: a' = a + w b, b' = a - w b,
    w = exp(-2 pi i (p mod 2^s) / 2^(s+1)) (conjugated for the inverse)."""
    R = v.shape[1]
    for s in range(s_lo, s_hi + 1):
        rb = s - reg_lo
        half = 1 << rb
        for q0 in range(R):
            if q0 & half:
                continue
            c = q0 & (half - 1)
            e = l + (c << reg_lo)  # p mod 2^s
            w = np.exp(-2j * np.pi * e / (1 << (s + 1)))
            if conj:
                w = np.conj(w)
            a = v[:, q0].copy()
            b = v[:, q0 | half].copy()
            v[:, q0] = a + w * b
            v[:, q0 | half] = a - w * b
    return v


def pair(A, Bp, w, ja, key, N, M, kappa):
    E = A + np.conj(Bp)
    D = A - np.conj(Bp)
    Tt = w * D
    X1 = E - 1j * Tt
    X2c = E + 1j * Tt
    m1, m2 = np.abs(X1), np.abs(X2c)
    th = lambda b: onp.phase_theta(key, np.asarray(b) % N, N).astype(np.float64)  # noqa: E731
    t1, t2, t3, t4 = th(ja), th(N - ja), th(M - ja), th(M + ja)
    Pz = (m1 * kappa) * ((np.cos(t1) + np.cos(t2)) + 1j * (np.sin(t1) - np.sin(t2)))
    Q = (m2 * kappa) * ((np.cos(t3) + np.cos(t4)) + 1j * (np.sin(t4) - np.sin(t3)))
    S, R = Pz + Q, Pz - Q
    U = np.conj(w) * R
    return S + 1j * U, np.conj(S - 1j * U)


def hop(x, window, key, log2n):
    m = log2n - 1
    M = 1 << m
    N = 2 * M
    T = M // 32
    lt = m - 5          # log2 T
    hb = m - 4          # first bit of the last pass (4 stages hb..m-1); 512 = 2^hb residues
    RES = 1 << hb
    tid = np.arange(T)
    lds = np.zeros(f3(M) + 64, np.complex128)
    kappa = 1.0 / (4.0 * N)

    # ---- load in F1 order: register q of thread t holds z[brev5(q) * T + t]
    v = np.empty((T, 32), np.complex128)
    for q in range(32):
        n = int(brev(np.array(q), 5)) * T + tid
        v[:, q] = x[2 * n] * window[2 * n] + 1j * x[2 * n + 1] * window[2 * n + 1]
    # F1: stages 0..4, constants only
    v = dit_stages(v, 0, 0, 4, 0, m, False)
    # E1 store: position p = (u << 5) | q, u = brev_lt(t) ; index map f3
    u = brev(tid, lt)
    for q in range(32):
        idx = f3((u << 5) | q)
        count_conflicts("E1 store", idx)
        lds[idx] = v[:, q]
    # E1 load: LOR=4 layout: p = (uu << 9) | (q << 4) | l ; tid = l | uu << 4
    l4, uu = tid & 15, tid >> 4
    for q in range(32):
        idx = f3((uu << 9) | (q << 4) | l4)
        count_conflicts("E1 load", idx)
        v[:, q] = lds[idx]
    # F2: stages 5..hb-1 on register bits 1.. (reg_lo = 4)
    v = dit_stages(v, l4, 5, hb - 1, 4, m, False)
    # E2 store (f1), same layout
    for q in range(32):
        idx = f1((uu << 9) | (q << 4) | l4)
        count_conflicts("E2 store", idx)
        lds[idx] = v[:, q]
    # E2 load: two natural groups: A: r + RES*q, B: rb + RES*q
    r = tid.copy()
    rb = (RES - tid) % RES
    rb[0] = RES // 2
    va = np.empty((T, 16), np.complex128)
    vb = np.empty((T, 16), np.complex128)
    for q in range(16):
        ia, ib = f1(r + RES * q), f1(rb + RES * q)
        count_conflicts("E2 load A", ia)
        count_conflicts("E2 load B", ib)
        va[:, q], vb[:, q] = lds[ia], lds[ib]
    # F3: stages hb..m-1 per group
    va = dit_stages(va, r, hb, m - 1, hb, m, False)
    vb = dit_stages(vb, rb, hb, m - 1, hb, m, False)
    # check: natural-order spectrum
    z = x[0::2][:M] * window[0::2][:M] + 1j * x[1::2][:M] * window[1::2][:M]
    Zf = np.fft.fft(z)
    for q in range(16):
        assert np.allclose(va[:, q], Zf[r + RES * q], atol=1e-9 * M)
        assert np.allclose(vb[:, q], Zf[rb + RES * q], atol=1e-9 * M)
    # ---- middle in registers
    save0 = np.concatenate([va[0], vb[0]])  # thread 0 dumps its 32 values to LDS scratch
    wr = np.exp(-2j * np.pi * r / N)
    for q in range(16):
        ja = r + RES * q
        w = wr * np.exp(-2j * np.pi * q * RES / N)
        VA, VB = pair(va[:, q], vb[:, 15 - q], w, ja, key, N, M, kappa)
        va[:, q], vb[:, 15 - q] = VA, VB
    # thread 0 side path (lanes 0..16 of wave 0)
    S = save0.copy()
    out = S.copy()
    for i in range(17):
        if i == 0:
            ja, a, b = 0, 0, 0
        elif i <= 7:
            ja, a, b = RES * i, i, 16 - i
        elif i == 8:
            ja, a, b = RES * 8, 8, 8
        else:
            qq = i - 9
            ja, a, b = RES // 2 + RES * qq, 16 + qq, 16 + 15 - qq
        w = np.exp(-2j * np.pi * ja / N)
        VA, VB = pair(S[a], S[b], w, ja, key, N, M, kappa)
        out[a] = VA
        if b != a:
            out[b] = VB
    va[0], vb[0] = out[:16], out[16:]
    # ---- I1: stages 0..3 in registers, register index brev4(q) (position low bits)
    b4 = [int(brev(np.array(q), 4)) for q in range(16)]
    pa = np.empty_like(va)
    pb = np.empty_like(vb)
    for q in range(16):
        pa[:, b4[q]] = va[:, q]
        pb[:, b4[q]] = vb[:, q]
    pa = dit_stages(pa, 0, 0, 3, 0, m, True)
    pb = dit_stages(pb, 0, 0, 3, 0, m, True)
    # E3 store: p = (brev_hb(r) << 4) | q'  (f3)
    ba, bb = brev(r, hb), brev(rb, hb)
    for q in range(16):
        ia, ib = f3((ba << 4) | q), f3((bb << 4) | q)
        count_conflicts("E3 store A", ia)
        count_conflicts("E3 store B", ib)
        lds[ia] = pa[:, q]
        lds[ib] = pb[:, q]
    # E3 load: LOR=4 layout
    for q in range(32):
        idx = f3((uu << 9) | (q << 4) | l4)
        count_conflicts("E3 load", idx)
        v[:, q] = lds[idx]
    # I2: stages 4..8 (all five register bits)
    v = dit_stages(v, l4, 4, 8, 4, m, True)
    # E4 store (f1) LOR=4 -> load LO=8 layout: p = (q << lt) | tid
    for q in range(32):
        idx = f1((uu << 9) | (q << 4) | l4)
        count_conflicts("E4 store", idx)
        lds[idx] = v[:, q]
    for q in range(32):
        idx = f1((q << lt) | tid)
        count_conflicts("E4 load", idx)
        v[:, q] = lds[idx]
    # I3: stages 9..m-1 (register bit 0 = position bit lt = 8 passive)
    v = dit_stages(v, tid, lt + 1, m - 1, lt, m, True)
    y = np.empty(N)
    for q in range(32):
        n = (q << lt) | tid
        y[2 * n] = v[:, q].real * window[2 * n]
        y[2 * n + 1] = v[:, q].imag * window[2 * n + 1]
    return y


if __name__ == "__main__":
    for log2n in (14,):
        N = 1 << log2n
        x = onp.synth_input(2, N).astype(np.float64)
        w = onp.hanning(N).astype(np.float64)
        key = onp.phase_key(0x5EED, 1, 3)
        y = hop(x, w, key, log2n)
        ref = onp.resynth(x, w.astype(np.float32), key)
        err = np.sqrt(np.mean((y - ref) ** 2))
        print(f"N={N} rms_err={err:.3e} rms={np.sqrt(np.mean(ref**2)):.3e}")
        for k, c in CONFLICTS.items():
            print(f"  {k:12s} worst extra LDS cycles per half-wave: {c}")
        assert err < 1e-9
