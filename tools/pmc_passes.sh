#!/bin/bash
# Collect PMC counters for bench.py in separate passes (no trace domains mixed in).
# usage: tools/pmc_passes.sh <outdir-under-gpurun_out> [bench args...]
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CNT in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
  "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
# summarise: per counter, mean over the hop_kernel dispatches
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hop" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as g:
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); g.write(line + "\n")
PY
