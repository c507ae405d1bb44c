"""dev: index model of hopw2_kernel (N = 8192: TWO waves per hop, 128 threads x 32 complex points). E2 / E3 are
wave-local (the wave is the lowest position / bin bit, which residue r and its partner 256 - r share), E1 / E4 cross the
two waves with a workgroup barrier around each round. Replays the exchanges with the kernel's address expressions,
checks every hand-over and counts bank conflicts (banking model of MI355X_MICROARCH.md)."""
import numpy as np

M = 4096


def bits(x, n):
    return [(x >> i) & 1 for i in range(n)]


def wr_conf(a):
    return sum(np.bincount(np.unique(a[16 * g:16 * g + 16]) % 16, minlength=16).max() - 1 for g in range(4))


def rd_conf(a):
    return sum(np.bincount(np.unique(a[32 * g:32 * g + 32]) % 32, minlength=32).max() - 1 for g in range(2))


W1 = {8: 1, 9: 2, 10: 4, 11: 8, 1: 16, 2: 33, 3: 72, 0: 144, 5: 288, 6: 576, 7: 1152}   # E1 (cross): 11 reduced bits
W2 = {1: 1, 2: 2, 3: 4, 4: 8, 5: 16, 6: 32, 8: 64, 9: 136, 10: 264, 11: 528}           # E2 (wave-local)
W3 = {10: 1, 9: 2, 8: 4, 7: 8, 6: 16, 5: 32, 0: 65, 1: 132, 2: 264, 3: 528}            # E3 (wave-local, bits Q)
W4 = {0: 1, 1: 2, 2: 4, 3: 8, 4: 16, 5: 32, 6: 64, 7: 128, 9: 256, 10: 512, 11: 1024}  # E4 (cross)


def addr(Wm, P):
    return sum(w * ((P >> b) & 1) for b, w in Wm.items())


L = np.arange(64)
conf, size = {}, {}


# L1: thread t (7 bits: lane = t & 63, wave = t >> 6), reg r = P0..P4; P5 = t6 ... P11 = t0
def P_L1(t, r):
    tb = bits(t, 7)
    P = r
    for i in range(7):
        P |= tb[6 - i] << (5 + i)
    return P


# L2: wave = P0, lane l2 = P1 + 2 P2 + 4 P3 + 8 P9 + 16 P10 + 32 P11, reg j = P4..P8
def P_L2(w, l2, j):
    lb, jb = bits(l2, 6), bits(j, 5)
    return w | (lb[0] << 1) | (lb[1] << 2) | (lb[2] << 3) | (jb[0] << 4) | (jb[1] << 5) | (jb[2] << 6) | (jb[3] << 7) | (jb[4] << 8) | (lb[3] << 9) | (lb[4] << 10) | (lb[5] << 11)


for h in range(2):  # E1, round = P4
    buf = {}
    for w in range(2):
        for rho in range(16):
            a = np.array([addr(W1, P_L1(64 * w + l, rho | (h << 4))) for l in L])
            conf[("E1 st", h, w, rho)] = wr_conf(a)
            for l in L:
                assert a[l] not in buf
                buf[a[l]] = P_L1(64 * w + l, rho | (h << 4))
    size["E1"] = max(size.get("E1", 0), max(buf) + 1)
    for w in range(2):
        for sg in range(16):
            j = h | (sg << 1)
            a = np.array([addr(W1, P_L2(w, l, j)) for l in L])
            conf[("E1 ld", h, w, sg)] = rd_conf(a)
            for l in L:
                assert buf[a[l]] == P_L2(w, l, j)
print("E1 ok")


# L3: thread tau = 2 a + w: set A residue tau, set B residue 256 - tau (tau = 0: 128); reg q = P8..P11
def res_of(tau, s):
    return tau if s == 0 else (128 if tau == 0 else 256 - tau)


for w in range(2):  # E2 wave-local, round = P7 = set; own region of 1056 slots
    for h in range(2):
        buf = {}
        for kk in range(16):  # regs with j3 (= P7) = h: (P4, P5, P6, P8) = kk bits
            kb = bits(kk, 4)
            j = kb[0] | (kb[1] << 1) | (kb[2] << 2) | (h << 3) | (kb[3] << 4)
            a = np.array([addr(W2, P_L2(w, l, j)) for l in L])
            conf[("E2 st", h, w, kk)] = wr_conf(a)
            for l in L:
                assert a[l] not in buf
                buf[a[l]] = P_L2(w, l, j)
        size["E2"] = max(size.get("E2", 0), max(buf) + 1)
        for q in range(16):
            P = np.array([res_of(2 * a_ + w, h) | (q << 8) for a_ in L])
            a = np.array([addr(W2, p) for p in P])
            base = L if h == 0 else (64 - L - w) & 63
            assert np.array_equal(a, base + 64 * (q & 1) + 136 * ((q >> 1) & 1) + 264 * ((q >> 2) & 1) + 528 * ((q >> 3) & 1)), ("E2 expr", w, h, q)
            conf[("E2 ld", h, w, q)] = rd_conf(a)
            for a_ in L:
                assert buf[a[a_]] == P[a_], ("E2", w, h, q, a_)
print("E2 ok")


def brev(x, n):
    r = 0
    for i in range(n):
        r |= ((x >> i) & 1) << (n - 1 - i)
    return r


def Q_L4(tau, s, rho):
    return brev(res_of(tau, s) + 256 * brev(rho, 4), 12)


# L5: wave = Q11, lane l5 = Q0 + 2 Q1 + 4 Q2 + 8 Q3 + 16 Q9 + 32 Q10, reg k = Q4..Q8
def Q_L5(w, l5, k):
    lb, kb = bits(l5, 6), bits(k, 5)
    return lb[0] | (lb[1] << 1) | (lb[2] << 2) | (lb[3] << 3) | (kb[0] << 4) | (kb[1] << 5) | (kb[2] << 6) | (kb[3] << 7) | (kb[4] << 8) | (lb[4] << 9) | (lb[5] << 10) | (w << 11)


for w in range(2):  # E3 wave-local, round = Q4 = set
    for h in range(2):
        buf = {}
        for rho in range(16):
            Q = np.array([Q_L4(2 * a_ + w, h, rho) for a_ in L])
            a = np.array([addr(W3, q) for q in Q])
            base = L if h == 0 else (64 - L - w) & 63
            assert np.array_equal(a, base + 65 * (rho & 1) + 132 * ((rho >> 1) & 1) + 264 * ((rho >> 2) & 1) + 528 * ((rho >> 3) & 1)), ("E3 expr", w, h, rho)
            conf[("E3 st", h, w, rho)] = wr_conf(a)
            for a_ in L:
                assert ((Q[a_] >> 4) & 1) == h and ((Q[a_] >> 11) & 1) == w
                assert a[a_] not in buf
                buf[a[a_]] = Q[a_]
        size["E3"] = max(size.get("E3", 0), max(buf) + 1)
        for sg in range(16):
            k = h | (sg << 1)
            a = np.array([addr(W3, Q_L5(w, l, k)) for l in L])
            conf[("E3 ld", h, w, sg)] = rd_conf(a)
            for l in L:
                assert buf[a[l]] == Q_L5(w, l, k)
print("E3 ok")


# L6: thread t = Q0..Q6 (lane = Q0..Q5, wave = Q6), reg = Q7..Q11
def Q_L6(t, r):
    return t | (r << 7)


for h in range(2):  # E4 cross, round = Q8
    buf = {}
    for w in range(2):
        for kk in range(16):
            a = np.array([addr(W4, Q_L5(w, l, kk | (h << 4))) for l in L])
            conf[("E4 st", h, w, kk)] = wr_conf(a)
            for l in L:
                assert a[l] not in buf
                buf[a[l]] = Q_L5(w, l, kk | (h << 4))
    size["E4"] = max(size.get("E4", 0), max(buf) + 1)
    for w in range(2):
        for rr in range(16):  # regs (Q7, Q9, Q10, Q11) = rr bits, Q8 = h
            rb = bits(rr, 4)
            r = rb[0] | (h << 1) | (rb[1] << 2) | (rb[2] << 3) | (rb[3] << 4)
            a = np.array([addr(W4, Q_L6(64 * w + l, r)) for l in L])
            assert np.array_equal(a, L + 64 * w + 128 * rb[0] + 256 * rb[1] + 512 * rb[2] + 1024 * rb[3])
            conf[("E4 ld", h, w, rr)] = rd_conf(a)
            for l in L:
                assert buf[a[l]] == Q_L6(64 * w + l, r)
print("E4 ok")
tot = {}
for key, c in conf.items():
    tot[key[0]] = tot.get(key[0], 0) + c
print("extra LDS cycles from bank conflicts:", tot)
print("buffer slots needed:", size)
