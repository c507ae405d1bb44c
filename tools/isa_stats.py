#!/usr/bin/env python3
"""Instruction histogram of one kernel in hipcc's device assembly (-S --cuda-device-only).

usage: isa_stats.py rc_kernels.s hop3_kernelILb1E [--loop]
Prices the VALU stream with the issue costs measured in profiles/r01_ubench_instruction_rates.txt
(v_pk_* / v_mul_lo / v_alignbit / v_mad_u32 4 cycles, transcendental 8, other VALU 2) and counts LDS,
vector-memory, barrier and wait instructions. --loop restricts the count to the largest basic-block
span that ends in a backward branch (the hop loop).

       isa_stats.py rc_kernels.s hop4_kernelILi1ELb0E --spills
Where the register allocator's spills sit: every scratch_* instruction of the kernel with its position relative to the
hop loop (before it: once per workgroup; inside: once per hop, with the number of s_barriers passed since the loop
head, i.e. which exchange group it falls into; after it), the slot it touches, and per slot how often it is stored
and reloaded inside the loop.
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


def spills_report(name, ins):
    """ins: [(op, rest)]. The hop loop = the largest span closed by a backward branch (as --loop)."""
    labels_at = {}
    best = (0, 0, 0)
    for i, (op, rest) in enumerate(ins):
        if op == "label":
            labels_at[rest] = i
    for i, (op, rest) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            t = rest.strip().split()[-1]
            if t in labels_at and labels_at[t] < i and i - labels_at[t] > best[0]:
                best = (i - labels_at[t], labels_at[t], i + 1)
    _, lo, hi = best
    rows, barriers = [], 0
    per_slot = collections.defaultdict(lambda: [0, 0, 0, 0])  # loop stores, loop loads, outside stores, outside loads
    for i, (op, rest) in enumerate(ins):
        if i == lo:
            barriers = 0
        if op == "s_barrier":
            barriers += 1
        if not op.startswith("scratch_"):
            continue
        m = re.search(r"offset:(\d+)", rest)
        slot = int(m.group(1)) if m else 0
        width = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}.get(op.split("_")[-1], 4)
        where = "before the loop" if i < lo else ("after the loop" if i >= hi else f"IN THE HOP LOOP, after barrier {barriers}")
        store = "store" in op
        k = (0 if store else 1) if lo <= i < hi else (2 if store else 3)
        per_slot[(slot, width)][k] += 1
        rows.append((i, op, slot, width, where))
    print(f"{name}: hop loop = instructions {lo}..{hi} of {len(ins)}; {sum(1 for o, _ in ins[lo:hi] if o == 's_barrier')} s_barrier per iteration")
    in_loop = [r for r in rows if r[4].startswith("IN")]
    print(f"  scratch instructions: {len(rows)} ({len(in_loop)} inside the hop loop)")
    for i, op, slot, width, where in rows:
        print(f"    #{i:5d} {op:24s} slot {slot:3d} ({width} B)  {where}")
    print("  per slot (offset, bytes): stores / reloads inside the loop, stores / reloads outside")
    for (slot, width), v in sorted(per_slot.items()):
        kind = "loop-invariant value parked before the loop, reloaded per hop" if v[0] == 0 and v[1] > 0 else (
            "spilled and reloaded per hop" if v[0] > 0 else "outside the loop only")
        print(f"    {slot:3d} {width:2d} B   loop {v[0]} / {v[1]}   outside {v[2]} / {v[3]}   {kind}")
    dwords = sum(w // 4 for (_, w) in per_slot)
    print(f"  {dwords} spilled dwords in {len(per_slot)} slots; per hop and lane: "
          f"{sum(v[0] for v in per_slot.values())} scratch stores, {sum(v[1] for v in per_slot.values())} scratch loads")


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
    if "--spills" in sys.argv:
        seq = []
        for ln in lines:
            m = re.match(r"^(\.LBB\S+):", ln)
            if m:
                seq.append(("label", m.group(1)))
                continue
            m = re.match(r"^\t([a-z][a-z0-9_]+)\b(.*)", ln)
            if m and not m.group(1).startswith("."):
                seq.append((m.group(1), m.group(2)))
        spills_report(name, seq)
        return
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
