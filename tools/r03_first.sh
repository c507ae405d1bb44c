#!/bin/bash
# first GPU call of round 3: the full GPU test suite on the split build, the default bench, the other configs,
# C5, and the FETCH_SIZE / WRITE_SIZE calibration for 4 / 8 / 16 bytes per lane
set -u
TAG=${1:-r03a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 10 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json
timeout -k 10 300 python tools/bench_c5.py > $OUT/c5.json 2>> $OUT/bench.err; cat $OUT/c5.json
timeout -k 10 300 python tools/bench_configs.py > $OUT/other_configs.json 2>> $OUT/bench.err; tail -c 600 $OUT/other_configs.json
timeout -k 10 300 python tools/bench_windows.py > $OUT/windows.json 2>> $OUT/bench.err; cat $OUT/windows.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetchcal.hip -o /tmp/fetchcal
cd /tmp && export TMPDIR=/tmp
for CNT in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $CNT --output-format csv -d $OUT/cal_$CNT -- /tmp/fetchcal > $OUT/cal_$CNT.log 2>&1
  python3 - "$OUT/cal_$CNT" <<'PY' | tee -a $OUT/fetchcal_summary.txt
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Counter_Name"], r["Kernel_Name"][:60], r["Counter_Value"])
PY
done
