#!/bin/bash
# One call on the GPU box -> every file profiles/ wants for a kernel state, all from the SAME binary and
# the SAME pre-heated command (bench.py, >= 2 s of back-to-back launches before anything is measured):
#   <tag>_bench.json          un-profiled bench line
#   <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats (per-kernel totals, warm-up launches included)
#   <tag>_kernel_trace.txt    per-launch durations of the hop kernel: mean / median of the LAST `steps`
#                             launches only (the timed ones), which is what the bench line must agree with
#   <tag>_pmc_summary.txt     separate --pmc passes, mean per hop-kernel launch, headed by `kernel_id:`
# usage (from the repo root on the box): tools/profile_round.sh r02a [pmc]   -> gpurun_out/<tag>/
set -u
TAG=$1
WITH_PMC=${2:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
STEPS=20
BENCH="$GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 5 --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
echo "bench rc=$?"; cut -c1-400 $OUT/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
python3 - "$OUT" "$TAG" "$STEPS" <<'PY'
import csv, glob, statistics, sys
out, tag, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    open(f"{out}/{tag}_kernel_stats.csv", "w").write(open(f).read())
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hop" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r["Kernel_Name"]))
rows.sort()
with open(f"{out}/{tag}_kernel_trace.txt", "w") as g:
    if rows:
        last = [d for _, d, _ in rows[-steps:]]
        alld = [d for _, d, _ in rows]
        g.write(f"kernel: {rows[-1][2]}\n")
        g.write(f"launches traced: {len(rows)} (warm-up + pre-heat + timed)\n")
        g.write(f"all launches:        mean {sum(alld)/len(alld):.4f} ms  median {statistics.median(alld):.4f}  min {min(alld):.4f}  max {max(alld):.4f}\n")
        g.write(f"last {steps} (timed) launches: mean {sum(last)/len(last):.4f} ms  median {statistics.median(last):.4f}  min {min(last):.4f}  max {max(last):.4f}\n")
    print(open(f"{out}/{tag}_kernel_trace.txt").read())
PY
[ "$WITH_PMC" = "pmc" ] || exit 0
i=0
for CNT in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
  "SQ_INSTS_VALU_TRANS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  # the DEFAULT bench command (same steps, warm-up and pre-heat as the un-profiled line above), minus its CPU legs
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $BENCH > $OUT/pass$i.log 2>&1
  echo "pmc pass $i rc=$?"
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections, json
out, tag = sys.argv[1], sys.argv[2]
kid, med = "unknown", "?"
try:
    roof = json.loads(open(f"{out}/{tag}_bench.json").read().strip().splitlines()[-1])["roofline"]
    kid, med = roof["kernel_id"], roof["kernel_ms"]
except Exception as e:  # noqa: BLE001
    print("no kernel id:", e)
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hop" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_pmc_summary.txt", "w") as g:
    g.write(f"# kernel_id: {kid}\n# this box, same call, un-profiled: kernel median {med} ms ({tag}_bench.json)\n"
            "# mean per hop-kernel launch over all launches of the DEFAULT bench command (`bench.py --steps 20 --warmup 5 "
            "--no-cpu-baseline --no-extras`, 2 s pre-heat), separate --pmc passes\n")
    for k in sorted(acc):
        v = acc[k]
        line = f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); g.write(line + "\n")
PY
