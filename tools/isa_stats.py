#!/usr/bin/env python3
"""Instruction histogram of one kernel in hipcc's device assembly (-S --cuda-device-only).

usage: isa_stats.py rc_kernels.s hop3_kernelILb1E [--loop]
Prices the VALU stream with the issue costs measured in profiles/r01_ubench_instruction_rates.txt
(v_pk_* / v_mul_lo / v_alignbit / v_mad_u32 4 cycles, transcendental 8, other VALU 2) and counts LDS,
vector-memory, barrier and wait instructions. --loop restricts the count to the largest basic-block
span that ends in a backward branch (the hop loop).
"""
import collections
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        if not on:
            if re.match(r"^_Z\S*%s\S*:" % re.escape(name), ln):
                on = True
            continue
        if ln.startswith(".Lfunc_end"):
            break
        out.append(ln.rstrip("\n"))
    return out


def classify(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_pk_"):
        return "valu_pk", 4
    if op in ("v_sin_f32", "v_cos_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32"):
        return "valu_trans", 8
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_alignbit", "v_mad_u32", "v_mad_i32", "v_xad")):
        return "valu_half", 4
    if op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
        return "valu_quarter", 8
    if op.startswith("v_"):
        return "valu", 2
    if op.startswith("ds_"):
        return ("lds_write" if "write" in op else "lds_read"), 0
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return ("vmem_store" if "store" in op else "vmem_load"), 0
    if op == "s_barrier":
        return "barrier", 0
    if op == "s_waitcnt":
        return "waitcnt", 0
    if op.startswith("s_"):
        return "salu", 0
    return "other", 0


def main():
    path, name = sys.argv[1], sys.argv[2]
    loop = "--loop" in sys.argv
    lines = kernel_lines(path, name)
    if not lines:
        raise SystemExit("kernel not found")
    # instruction lines: tab + mnemonic
    ins = []
    labels = {}
    for ln in lines:
        m = re.match(r"^(\.LBB\S+):", ln)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^\t([a-z][a-z0-9_]+)\b(.*)", ln)
        if m and not m.group(1).startswith("."):
            ins.append((m.group(1), m.group(2)))
    lo, hi = 0, len(ins)
    if loop:
        best = (0, 0, 0)
        for i, (op, rest) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                t = rest.strip().split()[-1]
                if t in labels and labels[t] < i and i - labels[t] > best[0]:
                    best = (i - labels[t], labels[t], i + 1)
        _, lo, hi = best
    hist = collections.Counter()
    ops = collections.Counter()
    cyc = 0
    for op, _ in ins[lo:hi]:
        c, w = classify(op)
        hist[c] += 1
        ops[op] += 1
        cyc += w
    print(f"{name}: instructions {hi - lo} ({'loop' if loop else 'whole kernel'})")
    for k, v in sorted(hist.items(), key=lambda kv: -kv[1]):
        print(f"  {k:14s} {v}")
    nv = sum(v for k, v in hist.items() if k.startswith("valu"))
    print(f"  VALU instructions {nv}, priced VALU issue cycles {cyc}")
    if "--ops" in sys.argv:
        for k, v in sorted(ops.items(), key=lambda kv: -kv[1])[:60]:
            print(f"    {k:28s} {v}")


if __name__ == "__main__":
    main()
