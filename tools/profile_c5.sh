#!/bin/bash
# rocprofv3 kernel stats of BASELINE C5 on one GPU (tools/bench_c5.py: fused big5_kernel; the window-32768 twin: big5s_kernel) -> gpurun_out/<tag>/
set -u
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/bench_c5.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err; cat $OUT/${TAG}_bench.json
ROCODER_HIP_LIB=$GRAFT_REPO_ROOT/rocoder_amd/librocoder_hip_hooks.so ROCODER_DIAG=2 python3 $GRAFT_REPO_ROOT/tools/bench_c5.py > $OUT/${TAG}_bench_prev_pipeline.json 2>> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/bench_c5.py > $OUT/trace.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
head -6 $OUT/${TAG}_kernel_stats.csv | cut -c1-200
i=0
for CNT in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/tools/bench_c5.py > $OUT/pass$i.log 2>&1
done
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "big4" in r["Kernel_Name"] or "big5" in r["Kernel_Name"]:  # (round 5: N = 65536 runs big5_kernel, N = 32768 big5s_kernel)
            key = ("R32 (N=32768)" if "big5s" in r["Kernel_Name"] or "<32" in r["Kernel_Name"] or "Li32E" in r["Kernel_Name"]
                   else "R64 (N=65536)")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_pmc_summary.txt", "w") as g:
    try:  # the family id of the kernels these counters belong to (rc_kernel_id(): a hash of the big4 sources)
        import os, re
        sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
        from rocoder_amd import _lib
        g.write("# kernel_id: big4=" + re.search(r"big4=([0-9a-f]+)", _lib.lib().rc_kernel_id().decode()).group(1) + "\n")
    except Exception as e:  # noqa: BLE001
        g.write(f"# (no kernel id: {e})\n")
    try:
        import json
        b = json.loads(open(f"{out}/{tag}_bench.json").read())
        g.write(f"# this box, same call, un-profiled: C5 {b['N65536']['ms_median']} ms, window-32768 twin {b['N32768']['ms_median']} ms ({tag}_bench.json)\n")
    except Exception as e:  # noqa: BLE001
        g.write(f"# (no bench line: {e})\n")
    g.write("# big5_kernel (N = 65536) and big5s_kernel (N = 32768), tools/bench_c5.py (8 ch x 5 292 000, factor 32), mean per launch, separate --pmc passes\n")
    for key in sorted(acc):
        g.write(f"## {key}\n")
        for k in sorted(acc[key]):
            v = acc[key][k]
            line = f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
            print(key, line); g.write(line + "\n")
PY
