"""Index model of big4_kernel (fused large-window kernel, N = 32768 / 65536): dev tool, numpy only.

One workgroup of T = 512 threads transforms M = N/2 = 512 R complex points held R per thread (R = 32
for N = 32768, R = 64 for N = 65536), b = log2 R:
  F1  stages 0..b-1        registers = position bits p0..p(b-1)           thread t = brev9(p_b..p_(b+8))
  F2  stages b..b+4        R/32 groups of 32 registers = p_b..p_(b+4)     thread = (lf = p0..p4, uu = top 4 bits)
                           (R = 64: the group index is p5)
  F3  stages b+5..b+8      R/16 sets of 16 registers = top 4 bits          thread tau holds the residues
                           tau, RES - tau (and tau + 512, RES - 512 - tau for R = 64), RES = 2^(b+5)
  middle stage in registers (pairs (A[q], B[15 - q]) of a residue and its partner), I1 = stages 0..3 of the
  inverse on the same sets, I2 = stages 4..8 on groups of 32, I3 = stages 9..m-1 on R registers.
Exchanges through a 16384-element LDS buffer: one round for R = 32, two rounds for R = 64 (each round moves
32 registers per thread). This model checks who gets what and the bank conflicts of every wave instruction.
"""
import sys

T = 512


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


class Lds:
    def __init__(self):
        self.mem = {}
        self.worst = {}

    def access(self, name, kind, addrs, vals=None):
        grp = 16 if kind == "w" else 32
        worst = 0
        for g in range(0, 64, grp):
            banks = {}
            for x in set(addrs[g:g + grp]):
                banks.setdefault(x % (16 if kind == "w" else 32), []).append(x)
            worst = max(worst, max(len(v) for v in banks.values()) - 1)
        self.worst[name] = max(self.worst.get(name, 0), worst)
        if kind == "w":
            for a, v in zip(addrs, vals):
                assert 0 <= a < 16400, a
                self.mem[a] = v
            return None
        return [self.mem[a] for a in addrs]


def e1_map(n):  # reduced 14-bit index -> slot
    return n + ((n >> 10) & 15)


def run(R):
    b = R.bit_length() - 1
    m = b + 9
    RES = 1 << (b + 5)
    G = R // 32           # groups of 32 registers (F2 / I2) = rounds per exchange
    NS = R // 16          # sets of 16 registers (F3 / I1)
    lds = Lds()
    waves = [list(range(w * 64, w * 64 + 64)) for w in range(T // 64)]

    def residues(tau):
        """the NS residues of thread tau, in set order (A0, B0[, A1, B1])"""
        out = []
        for grp in range(NS // 2):
            r = tau + 512 * grp
            out += [r, (RES // 2 if r == 0 else RES - r)]
        return out

    # ---------------------------------------------------------------- forward
    F1 = {t: [brev(t, 9) << b | q for q in range(R)] for t in range(T)}
    F2 = {t: [None] * R for t in range(T)}
    for g in range(G):      # round g moves registers with p5 = g (R = 64) / all (R = 32)
        for wv in waves:
            for q in range(32):
                ad = [e1_map(q | brev(t, 9) << 5) for t in wv]
                lds.access("E1.st", "w", ad, [F1[t][32 * g + q] for t in wv])
        for wv in waves:
            for j in range(32):
                ad = [e1_map((t & 31) | j << 5 | (t >> 5) << 10) for t in wv]
                got = lds.access("E1.ld", "r", ad)
                for t, x in zip(wv, got):
                    F2[t][32 * g + j] = x
    for t in range(T):
        lf, uu = t & 31, t >> 5
        for g in range(G):
            for j in range(32):
                low = lf | (g << 5 if R == 64 else 0)
                assert F2[t][32 * g + j] == (low | j << b | uu << (b + 5)), ("F2", t, g, j)
    # E2: reduced index = (residue mod 1024) | uu << 10 ; R = 64: round A = residues < 1024 (j < 16), B = the rest
    F3 = {t: [[None] * 16 for _ in range(NS)] for t in range(T)}
    for rnd in range(G):
        for wv in waves:
            for k in range(32):
                ad, vals = [], []
                for t in wv:
                    lf, uu = t & 31, t >> 5
                    if R == 32:
                        res10, reg = lf | k << 5, k
                    else:
                        g, jl = k >> 4, k & 15
                        res10, reg = lf | g << 5 | jl << 6, 32 * g + 16 * rnd + jl
                    ad.append(res10 | uu << 10)
                    vals.append(F2[t][reg])
                lds.access("E2.st", "w", ad, vals)
        for wv in waves:
            sets = range(NS) if R == 32 else ([0, 2] if rnd == 0 else [1, 3])
            for s in sets:
                for q in range(16):
                    ad = [(residues(t)[s] & 1023) | q << 10 for t in wv]
                    got = lds.access("E2.ld", "r", ad)
                    for t, x in zip(wv, got):
                        F3[t][s][q] = x
    for t in range(T):
        for s, r in enumerate(residues(t)):
            for q in range(16):
                assert F3[t][s][q] == (r | q << (b + 5)), ("F3", t, s, q)
            if R == 64:
                assert (r >= 1024) == (s in (1, 3)), (t, s, r)
    # ---------------------------------------------------------------- inverse: ids = inverse positions P
    # I1 set s of thread tau: P = q' | brev_{b+5}(residue) << 4
    I1 = {t: [[qq | brev(r, b + 5) << 4 for qq in range(16)] for r in residues(t)] for t in range(T)}
    # I2: registers = P4..P8 (x groups), thread = (l4 = P0..P3, hi = P9..P13 [5 bits]); R = 64: group = P14
    I2 = {t: [None] * R for t in range(T)}
    for rnd in range(G):
        # the elements with P4 = brev-top residue bit ... split on P4 = residue bit (b+4) as in E2
        for wv in waves:
            sets = range(NS) if R == 32 else ([0, 2] if rnd == 0 else [1, 3])
            for s in sets:
                for qq in range(16):
                    ad = []
                    for t in wv:
                        r = residues(t)[s]
                        P = I1[t][s][qq]
                        ad.append(red_inv(P, R, b))
                    lds.access("E3.st", "w", ad, [I1[t][s][qq] for t in wv])
        for wv in waves:
            for k in range(32):
                ad = []
                for t in wv:
                    l4, hi = t & 15, t >> 4
                    if R == 32:
                        P = l4 | k << 4 | hi << 9
                    else:
                        jl, g = k & 15, k >> 4      # register = P5..P8 (P4 = rnd), group g = P14
                        P = l4 | rnd << 4 | jl << 5 | hi << 9 | g << 14
                    ad.append(red_inv(P, R, b))
                got = lds.access("E3.ld", "r", ad)
                for t, x in zip(wv, got):
                    if R == 32:
                        I2[t][k] = x
                    else:
                        I2[t][32 * (k >> 4) + 2 * (k & 15) + rnd] = x
    for t in range(T):
        l4, hi = t & 15, t >> 4
        for g in range(G):
            for j in range(32):
                want = l4 | j << 4 | hi << 9 | (g << 14 if R == 64 else 0)
                assert I2[t][32 * g + j] == want, ("I2", t, g, j, I2[t][32 * g + j], want)
    # E4: I2 -> I3 (registers = P9..P(m-1), thread = P0..P8), round g moves group g (P14 = g)
    I3 = {t: [None] * R for t in range(T)}
    for g in range(G):
        for wv in waves:
            for j in range(32):
                ad = [e4_map((t & 15) | j << 4 | (t >> 4) << 9) for t in wv]
                lds.access("E4.st", "w", ad, [I2[t][32 * g + j] for t in wv])
        for wv in waves:
            for q in range(32):
                ad = [e4_map(t | q << 9) for t in wv]
                got = lds.access("E4.ld", "r", ad)
                for t, x in zip(wv, got):
                    I3[t][q + 32 * g] = x
    for t in range(T):
        for q in range(R):
            assert I3[t][q] == (t | q << 9), ("I3", t, q)
    return lds.worst


def red_inv(P, R, b):
    """E3 index: drop P4 (R = 64: it is the round) so that the index is (P0..P3, P5..) on 14 bits"""
    if R == 32:
        return e3_map(P)
    return e3_map((P & 15) | (P >> 5) << 4)


def e3_map(n):
    return n + ((n >> 10) & 15)


def e4_map(n):
    return n


if __name__ == "__main__":
    for R in (32, 64):
        worst = run(R)
        print("R =", R)
        for k in sorted(worst):
            print(f"  {k:6s} worst extra LDS cycles per lane group: {worst[k]}")
    print("all layouts check out")
