"""Search of the LDS index maps of hop2_kernel (13 position bits, F3_W) and hop3_kernel (12 reduced
position bits, G12_W): one weight per position bit, so that an index splits into a per-thread base and
a compile-time offset per register. Banking model (MI355X_MICROARCH.md, LDS): ds_write_b64 is served
in 16-lane groups on 32 four-byte banks, ds_read_b64 in 32-lane groups on 64 banks; a store costs
max(6, array cycles) because moving its VGPRs takes 6 cycles anyway.

    python tools/lds_map_search.py hop2 | hop3 [tries]
prints improving (score, store cycles, load cycles, weights, buffer size) lines; the committed weights
are the best lines of such runs."""
import itertools
import random
import sys

import numpy as np


def brev(x, bits):
    x = np.asarray(x)
    r = np.zeros_like(x)
    for b in range(bits):
        r |= ((x >> b) & 1) << (bits - 1 - b)
    return r


def rd_cost(idx):  # cycles per wave instruction, ideal 2
    tot = 0
    for w in range(0, len(idx), 64):
        for g in range(2):
            u = np.unique(idx[w + 32 * g:w + 32 * g + 32])
            tot += np.bincount(u % 32, minlength=32).max()
    return tot / (len(idx) // 64)


def wr_cost(idx):  # LDS-array cycles per wave instruction, ideal 4
    tot = 0
    for w in range(0, len(idx), 64):
        for g in range(4):
            u = np.unique(idx[w + 16 * g:w + 16 * g + 16])
            tot += np.bincount(u % 16, minlength=16).max()
    return tot / (len(idx) // 64)


tid = np.arange(256)
u8 = brev(tid, 8)
l4, uu = tid & 15, tid >> 4


def patterns(kind):
    """name -> (list of per-register position arrays over the 256 threads, is_store, multiplicity)"""
    if kind == "hop2":
        r = tid.copy()
        rb = (512 - tid) % 512
        rb[0] = 256
        ba, bb = brev(r, 9), brev(rb, 9)
        lay4 = [(uu << 9) | (q << 4) | l4 for q in range(32)]
        return 13, {
            "E1 store": ([(u8 << 5) | q for q in range(32)], True, 1),
            "E1/E3 load": (lay4, False, 2),
            "E2/E4 store": (lay4, True, 2),
            "E2 load A": ([r + 512 * q for q in range(16)], False, 1),
            "E2 load B": ([rb + 512 * q for q in range(16)], False, 1),
            "E3 store A": ([(ba << 4) | q for q in range(16)], True, 1),
            "E3 store B": ([(bb << 4) | q for q in range(16)], True, 1),
            "E4 load": ([(q << 8) | tid for q in range(32)], False, 1),
        }
    r, rbp = tid, (256 - tid) & 255
    lay4 = [(uu << 8) | (j << 4) | l4 for j in range(16)]
    return 12, {
        "E1 store": ([(u8 << 4) | q for q in range(16)], True, 2),
        "E2/E4 store": (lay4, True, 4),
        "E3 store A": ([(brev(r, 8) << 4) | q for q in range(16)], True, 1),
        "E3 store B": ([(brev(rbp, 8) << 4) | q for q in range(16)], True, 1),
        "E1/E3 load": (lay4, False, 4),
        "E2 load A": ([(q << 8) | r for q in range(16)], False, 1),
        "E2 load B": ([(q << 8) | rbp for q in range(16)], False, 1),
        "E4 load": ([(j << 8) | tid for j in range(16)], False, 2),
    }


def cost(nbits, pats, w, verbose=False):
    w = np.asarray(w)

    def f(n):
        n = np.asarray(n)
        out = np.zeros_like(n)
        for i in range(nbits):
            out += ((n >> i) & 1) * w[i]
        return out

    st = ld = 0.0
    for name, (regs, is_store, mult) in pats.items():
        c = [max(6.0, wr_cost(f(p))) if is_store else rd_cost(f(p)) for p in regs]
        if verbose:
            print(f"  {name:12s} {np.mean(c):.2f} cycles per wave instruction")
        if is_store:
            st += sum(c) * mult
        else:
            ld += sum(c) * mult
    return st, ld


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "hop3"
    tries = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    nbits, pats = patterns(kind)
    committed = ([1, 2, 4, 8, 16, 32, 64, 131, 259, 520, 1038, 2079, 4156] if kind == "hop2"
                 else [1, 2, 4, 8, 16, 32, 66, 130, 263, 526, 1052, 2104])
    print("committed weights:", committed, cost(nbits, pats, committed, True))
    random.seed(1)
    combos = list(itertools.product(range(4), repeat=nbits - 5))
    random.shuffle(combos)
    best = None
    for x in combos[:tries]:  # super-increasing weights (injective), 0..3 extra per bit above bit 4
        w = [1, 2, 4, 8, 16]
        for i in range(5, nbits):
            w.append(1 + sum(w) + x[i - 5])
        if sum(w) > (1 << nbits) + (1 << nbits) // 16:
            continue
        st, ld = cost(nbits, pats, w)
        if best is None or st + ld < best[0]:
            best = (st + ld, st, ld, w, sum(w) + 1)
            print(best, flush=True)
