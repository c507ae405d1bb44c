#!/bin/bash
# second PMC set: instruction fetch, queue levels, FIFO stalls
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_[A-Z_0-9]*" | sort -u | tr "\n" " " > $OUT/sqc_counters.txt
i=0
for CNT in \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_THREAD_CYCLES_VALU SQ_CYCLES SQ_BUSY_CU_CYCLES" \
  "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_ACTIVE_INST_VALU2 SQ_INST_CYCLES_SALU SQ_INSTS" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
  "SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ" ; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $OUT/pass$i.log | cut -c1-200
done
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
        line = f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); g.write(line + "\n")
PY
