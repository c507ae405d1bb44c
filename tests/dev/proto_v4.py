"""Index model of hop4_kernel's LDS exchanges (dev tool, numpy only; no audio is computed here).

hop4 keeps hop3's arithmetic and register layouts per pass and changes only WHO holds WHAT between
passes, so that two of the four exchanges never leave a wave:

  F1 (global load order)  thread t, reg q:           forward position p = brev8(t) << 5 | q
  F2 / I2                 thread (w, c, uu|l4'):      wave w = phi(low nibble), c = index inside the class
  F3 / middle / I1        thread (w, c, rho):         residues r = rho << 4 | member(w, c)  and  512 - r
  I3 (global store order) thread tid = P0..P7

phi groups the 16 low nibbles into 4 classes closed under negation mod 16, so a residue and its
partner 512 - r live in the same wave: E2 (F2 -> F3) and E3 (I1 -> I2) are wave-local (no s_barrier,
the LDS executes one wave's instructions in order). E1 and E4 cross waves in two rounds each:
round A = write-own-region / read-all, round B = write-all / read-own-region, so a region is only
ever overwritten by the wave that read it last: 3 barriers per cross exchange, 6 per hop.

The model moves element ids through an LDS image with exactly the address expressions of the kernel
and checks (1) every thread receives the elements its next pass expects, (2) bank conflicts of every
wave instruction under MI355X_MICROARCH.md's model (ds_write_b64: 16-lane groups on 32 banks;
ds_read_b64: 32-lane groups on 64 banks), (3) the hazard discipline (who touches which region when).
"""
import numpy as np

T = 256
REG = 1040           # float2 slots per wave region (16 x 65)
S65 = 65             # register stride of the wave-local exchanges


def brev(x, bits):
    r = 0
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def phi(l):            # class (= wave) of a low nibble
    l0, l1, l2 = l & 1, (l >> 1) & 1, (l >> 2) & 1
    return (1 + 2 * (l1 ^ l2)) if l0 else 2 * l1


def member(w, c):      # c = l2 << 1 | l3 : nibble of class w
    l3, l2 = c & 1, c >> 1
    l0 = w & 1
    l1 = ((w >> 1) ^ l2) if l0 else (w >> 1)
    return l3 << 3 | l2 << 2 | l1 << 1 | l0


def cidx(l):           # index of nibble l inside its class
    return ((l >> 2) & 1) << 1 | (l >> 3)


for w_ in range(4):
    assert sorted(member(w_, c_) for c_ in range(4)) == sorted(l for l in range(16) if phi(l) == w_)
    for c_ in range(4):
        l_ = member(w_, c_)
        assert cidx(l_) == c_ and phi((16 - l_) & 15) == w_


class Lds:
    def __init__(self):
        self.mem = {}
        self.worst = {}

    def access(self, name, kind, wave, addrs, vals=None):
        """one wave instruction: addrs[64] float2 indices"""
        addrs = np.asarray(addrs)
        grp = 16 if kind == "w" else 32
        worst = 0
        for g in range(0, 64, grp):
            a = addrs[g:g + grp]
            # distinct addresses on one bank conflict; equal addresses broadcast (reads)
            banks = {}
            for x in set(a.tolist()):
                banks.setdefault(x % (16 if kind == "w" else 32), []).append(x)
            worst = max(worst, max(len(v) for v in banks.values()) - 1)
        self.worst[name] = max(self.worst.get(name, 0), worst)
        if kind == "w":
            for a, v in zip(addrs.tolist(), vals):
                self.mem[a] = v
            return None
        return [self.mem[a] for a in addrs.tolist()]


def f2_ident(tid):
    w, c, uu = tid >> 6, (tid >> 4) & 3, tid & 15
    return w, c, uu, member(w, c)


def f3_ident(tid):
    w, c, rho = tid >> 6, (tid >> 4) & 3, tid & 15
    r = rho << 4 | member(w, c)
    rb = 256 if r == 0 else 512 - r
    return w, c, rho, r, rb


def holder_of_upper(x):      # thread (as residue r < 256) whose vb is residue 256 + x
    return (256 - x) & 255


def tid3_of_residue(r):
    return phi(r & 15) << 6 | cidx(r & 15) << 4 | (r >> 4)


def run():
    lds = Lds()
    waves = [list(range(w * 64, w * 64 + 64)) for w in range(4)]
    # ------------------------------------------------------------------ forward
    # F1 registers: v[t][q] = element p
    F1 = {t: [brev(t, 8) << 5 | q for q in range(32)] for t in range(T)}
    F2 = {tid: [None] * 32 for tid in range(T)}
    # E1 round A (p4 = 0): write own region, read all
    for wv in waves:
        for q in range(16):
            ad = [(t >> 6) * REG + q * 64 + ((t & 63) ^ (16 * ((q >> 3) & 1))) for t in wv]
            lds.access("E1A.st", "w", wv[0] >> 6, ad, [F1[t][q] for t in wv])
    for wv in waves:
        for j in range(16):
            ad = []
            for tid in wv:
                w, c, uu, l4 = f2_ident(tid)
                ts = brev(j, 4) << 4 | brev(uu, 4)          # source thread
                ad.append((ts >> 6) * REG + l4 * 64 + ((ts & 63) ^ (16 * ((l4 >> 3) & 1))))
            got = lds.access("E1A.ld", "r", wv[0] >> 6, ad)
            for tid, g in zip(wv, got):
                F2[tid][2 * j] = g
    # E1 round B (p4 = 1): write all, read own
    for wv in waves:
        for q in range(16):
            ad = []
            for t in wv:
                wd, cd = phi(q), cidx(q)
                uu, j = brev(t & 15, 4), brev(t >> 4, 4)
                ad.append(wd * REG + j * 64 + (cd << 4 | uu))
            lds.access("E1B.st", "w", wv[0] >> 6, ad, [F1[t][16 + q] for t in wv])
    for wv in waves:
        for j in range(16):
            ad = [(tid >> 6) * REG + j * 64 + (tid & 63) for tid in wv]
            got = lds.access("E1B.ld", "r", wv[0] >> 6, ad)
            for tid, g in zip(wv, got):
                F2[tid][2 * j + 1] = g
    for tid in range(T):
        w, c, uu, l4 = f2_ident(tid)
        for r5 in range(32):
            assert F2[tid][r5] == (l4 | r5 << 4 | uu << 9), ("F2", tid, r5)
    # E2 (wave-local). Round A: p8 = 0 (registers 0..15), round B: p8 = 1
    VA = {tid: [None] * 16 for tid in range(T)}
    VB = {tid: [None] * 16 for tid in range(T)}
    for rnd in range(2):
        for wv in waves:
            for rho in range(16):
                ad = [(tid >> 6) * REG + rho * S65 + (tid & 63) for tid in wv]
                lds.access(f"E2{'AB'[rnd]}.st", "w", wv[0] >> 6, ad, [F2[tid][16 * rnd + rho] for tid in wv])
            for q in range(16):
                ad = []
                for tid in wv:
                    w, c, rho3, r, rb = f3_ident(tid)
                    low8 = r if rnd == 0 else (rb - 256)
                    src_c, src_rho = cidx(low8 & 15), low8 >> 4
                    assert phi(low8 & 15) == w
                    ad.append(w * REG + src_rho * S65 + (src_c << 4 | q))
                got = lds.access(f"E2{'AB'[rnd]}.ld", "r", wv[0] >> 6, ad)
                for tid, g in zip(wv, got):
                    (VA if rnd == 0 else VB)[tid][q] = g
    for tid in range(T):
        w, c, rho3, r, rb = f3_ident(tid)
        for q in range(16):
            assert VA[tid][q] == (r | q << 9) and VB[tid][q] == (rb | q << 9), ("F3", tid, q)
    # ------------------------------------------------------------------ inverse (element ids = inverse positions P)
    PA = {tid: [None] * 16 for tid in range(T)}
    PB = {tid: [None] * 16 for tid in range(T)}
    for tid in range(T):
        w, c, rho3, r, rb = f3_ident(tid)
        for qq in range(16):
            PA[tid][qq] = qq | brev(r, 9) << 4
            PB[tid][qq] = qq | brev(rb, 9) << 4
    I2 = {tid: [None] * 32 for tid in range(T)}
    for rnd in range(2):
        for wv in waves:
            for qq in range(16):
                ad = [(tid >> 6) * REG + qq * S65 + (tid & 63) for tid in wv]
                lds.access(f"E3{'AB'[rnd]}.st", "w", wv[0] >> 6, ad, [(PA if rnd == 0 else PB)[tid][qq] for tid in wv])
            for j in range(16):
                ad = []
                for tid in wv:
                    w, cp, l4p = tid >> 6, (tid >> 4) & 3, tid & 15
                    nib = member(w, cp)
                    x = brev(j, 4) << 4 | nib           # low 8 bits of the residue wanted
                    rs = x if rnd == 0 else holder_of_upper(x)
                    assert phi(rs & 15) == w
                    ad.append(w * REG + l4p * S65 + (cidx(rs & 15) << 4 | rs >> 4))
                got = lds.access(f"E3{'AB'[rnd]}.ld", "r", wv[0] >> 6, ad)
                for tid, g in zip(wv, got):
                    I2[tid][2 * j + rnd] = g
    for tid in range(T):
        w, cp, l4p = tid >> 6, (tid >> 4) & 3, tid & 15
        uup = brev(member(w, cp), 4)
        for r5 in range(32):
            assert I2[tid][r5] == (l4p | r5 << 4 | uup << 9), ("I2", tid, r5)
    # E4 round A (P8 = 0): write own, read all
    I3 = {tid: [None] * 32 for tid in range(T)}
    for wv in waves:
        for rho in range(16):
            ad = [(tid >> 6) * REG + rho * 64 + ((tid & 63) ^ (16 * (rho & 1))) for tid in wv]
            lds.access("E4A.st", "w", wv[0] >> 6, ad, [I2[tid][rho] for tid in wv])
    for wv in waves:
        for j in range(16):
            nib = brev(j, 4)
            ws, cs = phi(nib), cidx(nib)
            ad = []
            for tid in wv:
                l4p, rho = tid & 15, tid >> 4
                ad.append(ws * REG + rho * 64 + ((cs << 4 | l4p) ^ (16 * (rho & 1))))
            got = lds.access("E4A.ld", "r", wv[0] >> 6, ad)
            for tid, g in zip(wv, got):
                I3[tid][2 * j] = g
    # E4 round B (P8 = 1): write all, read own
    for wv in waves:
        for rho in range(16):
            ad = []
            for tid in wv:
                w, cp, l4p = tid >> 6, (tid >> 4) & 3, tid & 15
                uup = brev(member(w, cp), 4)
                td = l4p | rho << 4
                ad.append((td >> 6) * REG + uup * 64 + (td & 63))
            lds.access("E4B.st", "w", wv[0] >> 6, ad, [I2[tid][16 + rho] for tid in wv])
    for wv in waves:
        for j in range(16):
            ad = [(tid >> 6) * REG + j * 64 + (tid & 63) for tid in wv]
            got = lds.access("E4B.ld", "r", wv[0] >> 6, ad)
            for tid, g in zip(wv, got):
                I3[tid][2 * j + 1] = g
    for tid in range(T):
        for r5 in range(32):
            assert I3[tid][r5] == (tid | r5 << 8), ("I3", tid, r5)
    return lds.worst


if __name__ == "__main__":
    worst = run()
    for k in sorted(worst):
        print(f"{k:8s} worst extra LDS cycles per lane group: {worst[k]}")
    print("all layouts check out")
