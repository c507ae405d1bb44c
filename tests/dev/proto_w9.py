"""dev: index model of hopw9_kernel (N = 512: M = 256 = 32 lanes x 8 points, two hops per wave, passes (3, 3, 2) /
(2, 3, 3), every exchange one full round of 8 stores + 8 loads per lane): weight search + replay (cf. proto_w10.py)."""
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


def P_L1(t, r):  # reg r = P0..P2, lane t: P3 = t4 ... P7 = t0
    return r | sum(bit(t, 4 - i) << (3 + i) for i in range(5))


def P_L2(l, j):  # lane (P0,P1,P2,P6,P7), reg j = (P3,P4,P5)
    return bit(l, 0) | bit(l, 1) << 1 | bit(l, 2) << 2 | (j << 3) | bit(l, 3) << 6 | bit(l, 4) << 7


def res_of(tau, s):
    return tau if s == 0 else (32 if tau == 0 else 64 - tau)


def P_L3(tau, r):  # reg r = s | q << 1 (s = set, q = (P6, P7))
    return res_of(tau, r & 1) | ((r >> 1) << 6)


def Q_L4(tau, r):  # reg r = s | rho << 1, rho = (Q0, Q1) = brev2(q)
    return brev(res_of(tau, r & 1) + 64 * brev(r >> 1, 2), 8)


def Q_L5(l, k):  # lane (Q0,Q1,Q5,Q6,Q7), reg k = (Q2,Q3,Q4)
    return bit(l, 0) | bit(l, 1) << 1 | (k << 2) | bit(l, 2) << 5 | bit(l, 3) << 6 | bit(l, 4) << 7


def Q_L6(t, r):  # lane t = Q0..Q4, reg r = Q5..Q7
    return t | (r << 5)


def run(W, writer, reader):
    conf, buf = 0, {}
    for r in range(8):
        Ps = np.array([writer(l, r) for l in L])
        a = np.array([addr(W, p) for p in Ps])
        conf += wr_conf(a)
        for l in L:
            assert a[l] not in buf
            buf[a[l]] = Ps[l]
    for r in range(8):
        Ps = np.array([reader(l, r) for l in L])
        a = np.array([addr(W, p) for p in Ps])
        conf += rd_conf(a)
        for l in L:
            assert buf[a[l]] == Ps[l]
    return conf, max(buf) + 1


def search(bits, fixed, writer, reader, limit=300):
    best = None

    def rec(W, rest, top):
        nonlocal best
        if best is not None:
            return
        if not rest:
            try:
                conf, size = run(W, writer, reader)
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
E["E1"] = search([0, 1, 2], {7: 1, 6: 2, 5: 4, 4: 8, 3: 16}, P_L1, P_L2)
E["E2"] = search([5, 6, 7], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16}, P_L2, P_L3)
E["E3"] = search([2, 0, 1], {7: 1, 6: 2, 5: 4, 4: 8, 3: 16}, Q_L4, Q_L5)
E["E4"] = search([5, 6, 7], {0: 1, 1: 2, 2: 4, 3: 8, 4: 16}, Q_L5, Q_L6)
for k, v in E.items():
    print(k, v)
W2, W3 = E["E2"][0], E["E3"][0]
for l in L:
    for q in range(4):
        assert addr(W2, P_L3(l, 1 | q << 1)) == ((32 - l) & 31) + W2[5] + sum(W2[6 + i] * bit(q, i) for i in range(2))
        assert addr(W2, P_L3(l, q << 1)) == l + sum(W2[6 + i] * bit(q, i) for i in range(2))
    for rho in range(4):
        assert addr(W3, Q_L4(l, 1 | rho << 1)) == ((32 - l) & 31) + W3[2] + sum(W3[i] * bit(rho, i) for i in range(2))
        assert addr(W3, Q_L4(l, rho << 1)) == l + sum(W3[i] * bit(rho, i) for i in range(2))
print("address expressions ok")
