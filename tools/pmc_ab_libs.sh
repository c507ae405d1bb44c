#!/bin/bash
# PMC comparison of two engine libraries on the bench command (same box): tools/pmc_ab_libs.sh libA.so libB.so "CNT ..." ["CNT ..."]
set -u
A=$1; B=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab_libs; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in $A $B; do
 i=0
 for CNT in "$@"; do
  i=$((i+1))
  ROCODER_HIP_LIB=$GRAFT_REPO_ROOT/rocoder_amd/$L rocprofv3 --pmc $CNT --output-format csv -d $OUT/${L}_pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline --no-extras > $OUT/${L}_pass$i.log 2>&1
 done
done
python3 - "$OUT" "$A" "$B" <<'PY'
import csv, glob, sys, collections
out, A, B = sys.argv[1:4]
res = {}
for d in (A, B):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{d}_pass*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hop" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[d] = {k: sum(v) / len(v) for k, v in acc.items()}
with open(out + "/summary.txt", "w") as g:
    for k in sorted(res[A]):
        a, b = res[A].get(k, 0), res[B].get(k, 0)
        line = f"{k:26s} {A}={a:.5g} {B}={b:.5g} ratio={a / b if b else float('nan'):.3f}"
        print(line); g.write(line + "\n")
PY
