"""dev: index model of hopw10_kernel (N = 1024: M = 512 = 32 lanes x 16 points, TWO hops per wave - one per half-wave,
passes (4, 2, 3) / (3, 2, 4)): searches conflict-free weights for the four exchanges of one half-wave (the other half uses
the same map in its own region), replays them and checks every hand-over. Bank groups: ds_write_b64 16 lanes, ds_read_b64
32 lanes = exactly one half-wave."""
import numpy as np

L = np.arange(32)


def bit(x, i):
    return (x >> i) & 1


def wr_conf(a):
    return sum(np.bincount(np.unique(a[16 * g:16 * g + 16]) % 16, minlength=16).max() - 1 for g in range(2))


def rd_conf(a):
    return np.bincount(np.unique(a) % 32, minlength=32).max() - 1


def addr(W, P):
    return sum(w * bit(P, b) for b, w in W.items())


def brev(x, n):
    r = 0
    for i in range(n):
        r |= bit(x, i) << (n - 1 - i)
    return r


def P_L1(t, r):  # reg r = P0..P3, lane t (5 bits): P4 = t4 ... P8 = t0
    return r | sum(bit(t, 4 - i) << (4 + i) for i in range(5))


def P_L2(l, j):  # lane (P0,P1,P2,P7,P8), reg j = (P3,P4,P5,P6)
    return bit(l, 0) | bit(l, 1) << 1 | bit(l, 2) << 2 | (j << 3) | bit(l, 3) << 7 | bit(l, 4) << 8


def res_of(tau, s):
    return tau if s == 0 else (32 if tau == 0 else 64 - tau)


def P_L3(tau, s, q):  # residue (6 bits) | q << 6
    return res_of(tau, s) | (q << 6)


def Q_L4(tau, s, rho):  # reg rho = Q0..Q2 = brev3(q)
    return brev(res_of(tau, s) + 64 * brev(rho, 3), 9)


def Q_L5(l, k):  # lane (Q0,Q1,Q2,Q7,Q8), reg k = (Q3,Q4,Q5,Q6)
    return bit(l, 0) | bit(l, 1) << 1 | bit(l, 2) << 2 | (k << 3) | bit(l, 3) << 7 | bit(l, 4) << 8


def Q_L6(t, r):  # lane t = Q0..Q4, reg r = Q5..Q8
    return t | (r << 5)


def run(name, W, writer, reader, round_of):
    conf, size = 0, 0
    for h in range(2):
        buf = {}
        for r in range(16):
            Ps = np.array([writer(l, r) for l in L])
            if round_of(Ps[0]) != h:
                continue
            assert all(round_of(p) == h for p in Ps)
            a = np.array([addr(W, p) for p in Ps])
            conf += wr_conf(a)
            for l in L:
                assert a[l] not in buf
                buf[a[l]] = Ps[l]
        size = max(size, max(buf) + 1)
        for r in range(16):
            Ps = np.array([reader(l, r) for l in L])
            if round_of(Ps[0]) != h:
                continue
            a = np.array([addr(W, p) for p in Ps])
            conf += rd_conf(a)
            for l in L:
                assert buf[a[l]] == Ps[l]
    return conf, size


def search(name, bits, fixed, writer, reader, round_of, limit=300):
    best = None

    def rec(W, rest, top):
        nonlocal best
        if best is not None:
            return
        if not rest:
            try:
                conf, size = run(name, W, writer, reader, round_of)
            except AssertionError:
                return
            if conf == 0 and size <= limit:
                best = (dict(W), size)
            return
        b = rest[0]
        for w in range(top + 1, top + 14):
            W[b] = w
            rec(W, rest[1:], top + w)
            del W[b]
            if best is not None:
                return

    rec(dict(fixed), bits, sum(fixed.values()))
    return best


E = {}
E["E1"] = search("E1", [0, 1, 2, 4], {8: 1, 7: 2, 6: 4, 5: 8}, P_L1, P_L2, lambda p: bit(p, 3))                       # round P3
E["E2"] = search("E2", [6, 7, 8], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16}, P_L2, lambda l, r: P_L3(l, r >> 3, r & 7),
                 lambda p: bit(p, 5))                                                                                  # round P5 = set
E["E3"] = search("E3", [0, 1, 2], {8: 1, 7: 2, 6: 4, 5: 8, 4: 16}, lambda l, r: Q_L4(l, r >> 3, r & 7), Q_L5,
                 lambda q: bit(q, 3))                                                                                  # round Q3 = set
E["E4"] = search("E4", [7, 6, 8], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16}, Q_L5, Q_L6, lambda q: bit(q, 5))                    # round Q5
for k, v in E.items():
    print(k, v)
W2, W3 = E["E2"][0], E["E3"][0]
for l in L:
    for q in range(8):
        assert addr(W2, P_L3(l, 1, q)) == ((32 - l) & 31) + sum(W2[6 + i] * bit(q, i) for i in range(3))
        assert addr(W2, P_L3(l, 0, q)) == l + sum(W2[6 + i] * bit(q, i) for i in range(3))
    for rho in range(8):
        assert addr(W3, Q_L4(l, 1, rho)) == ((32 - l) & 31) + sum(W3[i] * bit(rho, i) for i in range(3))
        assert addr(W3, Q_L4(l, 0, rho)) == l + sum(W3[i] * bit(rho, i) for i in range(3))
print("address expressions ok")
