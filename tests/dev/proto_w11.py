"""dev: index model of hopw_kernel at N = 2048 (M = 1024 = 64 lanes x 16 points, passes (4, 3, 3) / (3, 3, 4)): searches
conflict-free weights for the four exchanges, replays them and checks every hand-over (see proto_w.py for N = 4096)."""
import itertools
import numpy as np

L = np.arange(64)


def bit(x, i):
    return (x >> i) & 1


def wr_conf(a):
    return sum(np.bincount(np.unique(a[16 * g:16 * g + 16]) % 16, minlength=16).max() - 1 for g in range(4))


def rd_conf(a):
    return sum(np.bincount(np.unique(a[32 * g:32 * g + 32]) % 32, minlength=32).max() - 1 for g in range(2))


def addr(W, P):
    return sum(w * bit(P, b) for b, w in W.items())


def brev(x, n):
    r = 0
    for i in range(n):
        r |= bit(x, i) << (n - 1 - i)
    return r


# ---- layouts: functions (lane, reg) -> position (P for the forward side, Q for the inverse side)
def P_L1(t, r):  # reg r = P0..P3, lane t: P4 = t5 ... P9 = t0
    return r | sum(bit(t, 5 - i) << (4 + i) for i in range(6))


def P_L2(l, j):  # lane (P0,P1,P2,P7,P8,P9), reg j = (P3,P4,P5,P6)
    return bit(l, 0) | bit(l, 1) << 1 | bit(l, 2) << 2 | (j << 3) | bit(l, 3) << 7 | bit(l, 4) << 8 | bit(l, 5) << 9


def res_of(tau, s):
    return tau if s == 0 else (64 if tau == 0 else 128 - tau)


def P_L3(tau, s, q):  # residue (7 bits) | q << 7
    return res_of(tau, s) | (q << 7)


def Q_L4(tau, s, rho):  # reg rho = Q0..Q2 = brev3(q)
    return brev(res_of(tau, s) + 128 * brev(rho, 3), 10)


def Q_L5(l, k):  # lane (Q0,Q1,Q2,Q7,Q8,Q9), reg k = (Q3,Q4,Q5,Q6)
    return bit(l, 0) | bit(l, 1) << 1 | bit(l, 2) << 2 | (k << 3) | bit(l, 3) << 7 | bit(l, 4) << 8 | bit(l, 5) << 9


def Q_L6(t, r):  # lane t = Q0..Q5, reg r = Q6..Q9
    return t | (r << 6)


def run(name, W, writer, reader, nreg_w, nreg_r, round_of):
    """writer / reader: (lane, reg) -> position; registers with round_of(position) == h move in round h."""
    conf = 0
    size = 0
    for h in range(2):
        buf = {}
        for r in range(nreg_w):
            Ps = np.array([writer(l, r) for l in L])
            if round_of(Ps[0]) != h:
                continue
            assert all(round_of(p) == h for p in Ps)
            a = np.array([addr(W, p) for p in Ps])
            conf += wr_conf(a)
            for l in L:
                assert a[l] not in buf, (name, "collision")
                buf[a[l]] = Ps[l]
        size = max(size, max(buf) + 1)
        for r in range(nreg_r):
            Ps = np.array([reader(l, r) for l in L])
            if round_of(Ps[0]) != h:
                continue
            a = np.array([addr(W, p) for p in Ps])
            conf += rd_conf(a)
            for l in L:
                assert buf[a[l]] == Ps[l], (name, h, r, l)
    return conf, size


def search(name, bits, fixed, writer, reader, nreg_w, nreg_r, round_of, limit=560):
    """bits: position bits still without a weight; tries small weight sets greedily (ascending, injective)."""
    best = None

    def rec(W, rest, top):
        nonlocal best
        if best is not None:
            return
        if not rest:
            try:
                conf, size = run(name, W, writer, reader, nreg_w, nreg_r, round_of)
            except AssertionError:
                return
            if conf == 0 and size <= limit:
                best = (dict(W), size)
            return
        b = rest[0]
        for w in range(top + 1, top + 12):
            W[b] = w
            rec(W, rest[1:], top + w)
            del W[b]
            if best is not None:
                return

    rec(dict(fixed), bits, sum(fixed.values()))
    return best


E = {}
# E1: round = P3
E["E1"] = search("E1", [0, 1, 2, 4, 5], {9: 1, 8: 2, 7: 4, 6: 8}, P_L1, P_L2, 16, 16, lambda p: bit(p, 3))
# E2: round = P6 = the set; reader register r = 8 s + q
E["E2"] = search("E2", [7, 8, 9], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 32}, P_L2, lambda l, r: P_L3(l, r >> 3, r & 7), 16, 16,
                 lambda p: bit(p, 6))
# E3: round = Q3 = the set; writer register r = 8 s + rho; residue bits j0..j5 = Q9..Q4 keep weights 1..32
E["E3"] = search("E3", [0, 1, 2], {9: 1, 8: 2, 7: 4, 6: 8, 5: 16, 4: 32}, lambda l, r: Q_L4(l, r >> 3, r & 7), Q_L5, 16, 16,
                 lambda q: bit(q, 3))
# E4: round = Q6
E["E4"] = search("E4", [7, 8, 9], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 32}, Q_L5, Q_L6, 16, 16, lambda q: bit(q, 6))
for k, v in E.items():
    print(k, v)
# the kernel's address expressions for the set-B side: base (64 - lane) & 63 on the residue's low six bits
W2, W3 = E["E2"][0], E["E3"][0]
for l in L:
    for q in range(8):
        assert addr(W2, P_L3(l, 1, q)) == ((64 - l) & 63) + sum(W2[7 + i] * bit(q, i) for i in range(3))
        assert addr(W2, P_L3(l, 0, q)) == l + sum(W2[7 + i] * bit(q, i) for i in range(3))
    for rho in range(8):
        assert addr(W3, Q_L4(l, 1, rho)) == ((64 - l) & 63) + sum(W3[i] * bit(rho, i) for i in range(3))
        assert addr(W3, Q_L4(l, 0, rho)) == l + sum(W3[i] * bit(rho, i) for i in range(3))
print("address expressions ok")
