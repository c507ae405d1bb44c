#!/bin/bash
# PMC comparison of hop4 (ROCODER_DIAG=0) and the previous generation (ROCODER_DIAG=2), same box, same counters.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ROCODER_HIP_LIB=$GRAFT_REPO_ROOT/rocoder_amd/librocoder_hip_hooks.so  # (the build that reads ROCODER_DIAG)
for d in 0 2; do
 i=0
 for CNT in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
  "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  ROCODER_DIAG=$d rocprofv3 --pmc $CNT --output-format csv -d $OUT/d${d}_pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --preheat-s 0.3 --no-cpu-baseline --no-extras > $OUT/d${d}_pass$i.log 2>&1
 done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
res = {}
for d in ("0", "2"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/d{d}_pass*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "hop" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[d] = {k: sum(v) / len(v) for k, v in acc.items()}
with open(out + "/summary.txt", "w") as g:
    for k in sorted(res["0"]):
        a, b = res["0"].get(k, 0), res["2"].get(k, 0)
        line = f"{k:26s} hop4={a:.5g} prev={b:.5g} ratio={a / b if b else float('nan'):.3f}"
        print(line); g.write(line + "\n")
PY
