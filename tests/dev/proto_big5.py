"""Index model of big5_kernel (N = 65536: the wave-local exchange scheme of round 5): dev tool, numpy-free.

T = 512 threads x R = 64 complex points, eight waves, one LDS buffer cut into eight regions of 2111 float2 (2048 + the skew padding of the wave-local exchanges).
  F1  stages 0..5    thread = tid (sample order), registers P0..5                                  (as big4)
  F2  stages 6..10   wave k = residue class, lane = uu | a << 4 (uu = P11..14, lf = P0..4 = member(k, a)), registers
                     j = P6..10, round g = P5
  F3  stages 11..14  same wave, lane = x | a << 4: thread tau = member(k, a) | x << 5 holds residues tau, RES - tau,
                     tau + 512, RES - 512 - tau (16 registers = P11..14 each): E2 never leaves the wave
  I1 stages 0..3 on the same sets; I2 stages 4..9: same wave, lane = l4 | a << 4 (l4 = P'0..3, class bits = P'10..14),
                     registers j = P'4..8, group = P'9: E3 never leaves the wave
  I3  stages 10..14  thread = tid = P'0..8, registers P'10..14 per round P'9 (stage 14 in the epilogue)
Residue classes (low five residue bits, closed under negation so that the (j, M - j) pair stage stays in a thread):
  class k = {k, k + 16, 32 - k, 16 - k} (k = 1..7), class 0 = {0, 16, 24, 8}; all members of a class have k's parity,
  so in E4 the waves of even classes hold the P'14 = 0 half (round 0) and the odd ones the other (round 1).
Cross-wave exchanges: E1 round 0 writer-major (a wave stores into its OWN region, everybody reads everywhere), round 1
reader-major (stores go everywhere, a wave reads its OWN region): three barriers, none at the entry. E4 the same, its
round being the I2 group P'9 (I2 runs stages 4..9, I3 stages 10..13 on P'10..13; a first design with I2 = 4..8 made the
round P'14 = the class parity: half of the waves would have stored 64 registers per round and kept them live too long).
Checks: who gets what, and the bank conflicts of every wave instruction (ds_write_b64: 16-lane groups over 16 slots,
ds_read_b64: 32-lane groups over 32 slots, as measured in round 3)."""
T, R, b, m, RES = 512, 64, 6, 15, 2048


def brev(x, bits):
    r = 0
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def member(k, a):
    low4 = (16 - k if k else 8) if a & 2 else k
    return (16 if a in (1, 2) else 0) | (low4 & 15)


CLASS_OF = {}
for k_ in range(8):
    for a_ in range(4):
        CLASS_OF[member(k_, a_)] = (k_, a_)
assert len(CLASS_OF) == 32
for k_ in range(8):
    ms = {member(k_, a_) for a_ in range(4)}
    assert {(-x) % 32 for x in ms} == ms and len({x & 1 for x in ms}) == 1, k_


class Lds:
    def __init__(self):
        self.mem, self.worst, self.owner = {}, {}, {}

    def access(self, name, kind, addrs, vals=None, active=None):
        grp, mod = (16, 16) if kind == "w" else (32, 32)
        worst = 0
        for g in range(0, 64, grp):
            lanes = [i for i in range(g, g + grp) if active is None or active[i]]
            banks = {}
            for x in {addrs[i] for i in lanes}:
                banks.setdefault(x % mod, []).append(x)
            if banks:
                worst = max(worst, max(len(v) for v in banks.values()) - 1)
        self.worst[name] = max(self.worst.get(name, 0), worst)
        if kind == "w":
            for i, (a, v) in enumerate(zip(addrs, vals)):
                if active is None or active[i]:
                    assert 0 <= a < 8 * REGION, a
                    self.mem[a] = v
            return None
        return [self.mem[a] for a in addrs]


def residues(tau):
    out = []
    for gp in range(2):
        r = tau + 512 * gp
        out += [r, (RES // 2 if r == 0 else RES - r)]
    return out


def split_res(r):
    """residue mod 1024 -> (class k, member index a, y = residue bits 5..9)"""
    k, a = CLASS_OF[r & 31]
    return k, a, (r >> 5) & 31


# ---- index maps (in-region, 0 .. 2047) -------------------------------------------------------------------------
def e1r0(P):   # writer-major: region = writer wave; uu | (P4 | P0..3 << 1 | P9 << 5 | P10 << 6) << 4
    uu = (P >> 11) & 15
    Y = ((P >> 4) & 1) | (P & 15) << 1 | ((P >> 9) & 3) << 5
    return uu | Y << 4


def e1r1(P):   # reader-major: region = class of P0..4; uu | a << 4 | j << 6
    _, a = CLASS_OF[P & 31]
    return ((P >> 11) & 15) | a << 4 | ((P >> 6) & 31) << 6


# The two wave-local exchanges are transposes (lanes <-> registers), so one side needs a skew. Both sides address as
# base(lane[, set]) + constant(register) - the ds instructions' immediate offsets, no VALU per access - which forces
# idx = alpha(a) + beta(uu) + gamma(x) + delta(gp): rows of 32 (x | (a & 1) << 4) at a stride of 33 over uu, four blocks
# of 528 for (a >> 1, gp): 2111 slots per region instead of 2048 (8 x 63 float2 = 4 KB of padding in all)
def e2(r10, uu):
    _, a, y = split_res(r10)
    return (y & 15) + 16 * (a & 1) + 33 * uu + 528 * ((a >> 1) | (y >> 4) << 1)


def e3(r10, qq):   # the same shape with the roles swapped: rows (q' | (a & 1) << 4) at a stride of 33 over x
    _, a, y = split_res(r10)
    return qq + 16 * (a & 1) + 33 * (y & 15) + 528 * ((a >> 1) | (y >> 4) << 1)


REGION = 2111


def run():
    lds = Lds()
    waves = [list(range(w * 64, w * 64 + 64)) for w in range(8)]
    F1 = {t: [brev(t, 9) << b | q for q in range(R)] for t in range(T)}
    # ---- E1
    F2 = {t: [None] * R for t in range(T)}
    for g in range(2):
        for w, wv in enumerate(waves):
            for q in range(32):
                ad = []
                for t in wv:
                    P = F1[t][32 * g + q]
                    if g == 0:
                        assert brev((P >> 6) & 7, 3) == w   # own region
                        ad.append(REGION * w + e1r0(P))
                    else:
                        ad.append(REGION * CLASS_OF[P & 31][0] + e1r1(P))
                lds.access(f"E1.r{g}.st", "w", ad, [F1[t][32 * g + q] for t in wv])
        for k, wv in enumerate(waves):
            for j in range(32):
                ad = []
                for t in wv:
                    lane = t & 63
                    uu, a = lane & 15, lane >> 4
                    P = member(k, a) | g << 5 | j << 6 | uu << 11
                    ad.append(REGION * brev(j & 7, 3) + e1r0(P) if g == 0 else REGION * k + e1r1(P))
                got = lds.access(f"E1.r{g}.ld", "r", ad)
                for t, x in zip(wv, got):
                    F2[t][32 * g + j] = x
    for k, wv in enumerate(waves):
        for t in wv:
            lane = t & 63
            for g in range(2):
                for j in range(32):
                    assert F2[t][32 * g + j] == (member(k, lane >> 4) | g << 5 | j << 6 | (lane & 15) << 11)
    # ---- E2 (wave-local): round = residue bit 10 = j bit 4
    F3 = {t: [[None] * 16 for _ in range(4)] for t in range(T)}
    tau_of = {}
    for k, wv in enumerate(waves):
        for t in wv:
            lane = t & 63
            tau_of[t] = member(k, lane >> 4) | (lane & 15) << 5
    assert sorted(tau_of.values()) == list(range(512))
    for rnd in range(2):
        for k, wv in enumerate(waves):
            for kk in range(32):
                g, jl = kk >> 4, kk & 15
                ad, vals = [], []
                for t in wv:
                    lane = t & 63
                    r10 = member(k, lane >> 4) | g << 5 | jl << 6
                    ad.append(REGION * k + e2(r10, lane & 15))
                    vals.append(F2[t][32 * g + 16 * rnd + jl])
                lds.access("E2.st", "w", ad, vals)
            for s in ([0, 2] if rnd == 0 else [1, 3]):
                for q in range(16):
                    ad = [REGION * k + e2(residues(tau_of[t])[s] & 1023, q) for t in wv]
                    for t in wv:
                        assert split_res(residues(tau_of[t])[s] & 1023)[0] == k   # stays in the wave's region
                    got = lds.access("E2.ld", "r", ad)
                    for t, x in zip(wv, got):
                        F3[t][s][q] = x
    for t in range(T):
        for s, r in enumerate(residues(tau_of[t])):
            for q in range(16):
                assert F3[t][s][q] == (r | q << 11), ("F3", t, s, q)
            assert (r >= 1024) == (s in (1, 3))
    # ---- inverse. I1 set s of thread tau: P' = q' | brev11(residue) << 4
    I1 = {t: [[qq | brev(r, 11) << 4 for qq in range(16)] for r in residues(tau_of[t])] for t in range(T)}
    I2 = {t: [None] * R for t in range(T)}
    for rnd in range(2):
        for k, wv in enumerate(waves):
            for s in ([0, 2] if rnd == 0 else [1, 3]):
                for qq in range(16):
                    ad = [REGION * k + e3(residues(tau_of[t])[s] & 1023, qq) for t in wv]
                    lds.access("E3.st", "w", ad, [I1[t][s][qq] for t in wv])
            for kk in range(32):
                grp, jl = kk >> 4, kk & 15   # grp = P'9 = residue bit 5; jl = P'5..8 = residue bits 9, 8, 7, 6
                ad = []
                for t in wv:
                    lane = t & 63
                    l4, a = lane & 15, lane >> 4
                    y = grp | brev(jl, 4) << 1          # residue bits 5..9
                    r10 = member(k, a) | y << 5
                    ad.append(REGION * k + e3(r10, l4))
                got = lds.access("E3.ld", "r", ad)
                for t, x in zip(wv, got):
                    I2[t][32 * grp + 2 * jl + rnd] = x
    for k, wv in enumerate(waves):
        for t in wv:
            lane = t & 63
            l4, a = lane & 15, lane >> 4
            for grp in range(2):
                for j in range(32):
                    want = l4 | j << 4 | grp << 9 | brev(member(k, a), 5) << 10
                    assert I2[t][32 * grp + j] == want, ("I2", t, grp, j, I2[t][32 * grp + j], want)
    # ---- E4. I2 runs stages 4..9 (4..8 on each group of 32, then stage 9 across the groups), so the round of E4 is
    # the group g = P'9 and I3 = stages 10..13 on the registers P'10..13 with P'9 and P'14 looking on (14 in the epilogue).
    # round 0 writer-major: own region, index l4 | j << 4 | a << 9; round 1 reader-major: region P'6..8, index
    # P'0..5 | (P'10..14) << 6. No barrier at the entry (own region last read by the wave's own E3): three barriers.
    I3 = {t: [None] * R for t in range(T)}
    for g in range(2):
        for k, wv in enumerate(waves):
            for j in range(32):
                ad, vals = [], []
                for t in wv:
                    lane = t & 63
                    l4, a = lane & 15, lane >> 4
                    P = I2[t][32 * g + j]
                    assert ((P >> 9) & 1) == g
                    if g == 0:
                        ad.append(REGION * k + (l4 | j << 4 | a << 9))
                    else:
                        ad.append(REGION * ((P >> 6) & 7) + ((P & 63) | (P >> 10) << 6))
                    vals.append(P)
                lds.access(f"E4.r{g}.st", "w", ad, vals)
        for w, wv in enumerate(waves):
            for r5 in range(32):
                ad = []
                for t in wv:
                    P = t | g << 9 | r5 << 10
                    if g == 0:
                        k, a = CLASS_OF[brev(r5, 5)]
                        ad.append(REGION * k + (t | a << 9))
                    else:
                        ad.append(REGION * w + ((t & 63) | r5 << 6))
                got = lds.access(f"E4.r{g}.ld", "r", ad)
                for t, x in zip(wv, got):
                    I3[t][(g | (r5 & 15) << 1) + 32 * (r5 >> 4)] = x   # y[q + 32 h]: q = P'9..13, h = P'14
    for t in range(T):
        for q in range(R):
            assert I3[t][q] == (t | q << 9), ("I3", t, q)
    return lds.worst


def run32():
    """big5s_kernel (N = 32768, 32 points per thread): single-round exchanges with big5's maps - E1 writer-major (round 0's
    map on a 14-bit position: P0..4 = register, P5..13 = brev9(tid)), E2 / E3 wave-local (y = the five residue bits 5..9 =
    the F2 register), E4 reader-major (round 1's map)."""
    RES32 = 1024
    lds = Lds()
    waves = [list(range(w * 64, w * 64 + 64)) for w in range(8)]

    def e1(P):   # uu = P10..13, Y = P4 | P0..3 << 1 | P8 << 5 | P9 << 6
        return ((P >> 10) & 15) | ((((P >> 4) & 1) | (P & 15) << 1 | ((P >> 8) & 3) << 5) << 4)

    F1 = {t: [brev(t, 9) << 5 | q for q in range(32)] for t in range(T)}
    F2 = {t: [None] * 32 for t in range(T)}
    for w, wv in enumerate(waves):
        for q in range(32):
            ad = []
            for t in wv:
                P = F1[t][q]
                assert brev((P >> 5) & 7, 3) == w
                ad.append(REGION * w + e1(P))
            lds.access("E1.st", "w", ad, [F1[t][q] for t in wv])
    for k, wv in enumerate(waves):
        for j in range(32):
            ad = []
            for t in wv:
                lane = t & 63
                P = member(k, lane >> 4) | j << 5 | (lane & 15) << 10
                ad.append(REGION * brev(j & 7, 3) + e1(P))
            got = lds.access("E1.ld", "r", ad)
            for t, x in zip(wv, got):
                F2[t][j] = x
    for k, wv in enumerate(waves):
        for t in wv:
            lane = t & 63
            for j in range(32):
                assert F2[t][j] == (member(k, lane >> 4) | j << 5 | (lane & 15) << 10)
    tau_of = {t: member(t >> 6, (t & 63) >> 4) | (t & 15) << 5 for t in range(T)}
    assert sorted(tau_of.values()) == list(range(512))

    def res32(tau):
        return [tau, (RES32 // 2 if tau == 0 else RES32 - tau)]

    F3 = {t: [[None] * 16 for _ in range(2)] for t in range(T)}
    for k, wv in enumerate(waves):
        for j in range(32):
            ad = [REGION * k + e2(member(k, (t & 63) >> 4) | j << 5, t & 15) for t in wv]
            lds.access("E2.st", "w", ad, [F2[t][j] for t in wv])
        for s_ in range(2):
            for q in range(16):
                ad = [REGION * k + e2(res32(tau_of[t])[s_] & 1023, q) for t in wv]
                got = lds.access("E2.ld", "r", ad)
                for t, x in zip(wv, got):
                    F3[t][s_][q] = x
    for t in range(T):
        for s_, r in enumerate(res32(tau_of[t])):
            for q in range(16):
                assert F3[t][s_][q] == ((r & 1023) | q << 10), ("F3", t, s_, q)   # (residue 512 <-> RES/2 for tau = 0 handled by & 1023)
    I1 = {t: [[qq | brev(r & 1023, 10) << 4 for qq in range(16)] for r in res32(tau_of[t])] for t in range(T)}
    I2 = {t: [None] * 32 for t in range(T)}
    for k, wv in enumerate(waves):
        for s_ in range(2):
            for qq in range(16):
                ad = [REGION * k + e3(res32(tau_of[t])[s_] & 1023, qq) for t in wv]
                lds.access("E3.st", "w", ad, [I1[t][s_][qq] for t in wv])
        for j in range(32):
            ad = []
            for t in wv:
                lane = t & 63
                ad.append(REGION * k + e3(member(k, lane >> 4) | brev(j, 5) << 5, lane & 15))
            got = lds.access("E3.ld", "r", ad)
            for t, x in zip(wv, got):
                I2[t][j] = x
    for k, wv in enumerate(waves):
        for t in wv:
            lane = t & 63
            for j in range(32):
                assert I2[t][j] == ((lane & 15) | j << 4 | brev(member(k, lane >> 4), 5) << 9), ("I2", t, j)
    I3 = {t: [None] * 32 for t in range(T)}
    for k, wv in enumerate(waves):
        for j in range(32):
            ad = [REGION * ((I2[t][j] >> 6) & 7) + ((I2[t][j] & 63) | (I2[t][j] >> 9) << 6) for t in wv]
            lds.access("E4.st", "w", ad, [I2[t][j] for t in wv])
    for w, wv in enumerate(waves):
        for q in range(32):
            got = lds.access("E4.ld", "r", [REGION * w + ((t & 63) | q << 6) for t in wv])
            for t, x in zip(wv, got):
                I3[t][q] = x
    for t in range(T):
        for q in range(32):
            assert I3[t][q] == (t | q << 9), ("I3", t, q)
    return lds.worst


if __name__ == "__main__":
    worst = run()
    for k in sorted(worst):
        print(f"  {k:10s} worst extra LDS cycles per lane group: {worst[k]}")
    worst32 = run32()
    for k in sorted(worst32):
        print(f"  R32 {k:6s} worst extra LDS cycles per lane group: {worst32[k]}")
    print("all layouts check out")
