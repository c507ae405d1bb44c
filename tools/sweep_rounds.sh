#!/bin/bash
# sweep the run planner of the N = 16384 kernel on one box: tools/sweep_rounds.sh "4 6 8 12" "4 8"
export ROCODER_HIP_LIB=$PWD/rocoder_amd/librocoder_hip_hooks.so  # (the build that reads the tuning variables)
for rep in 1 2; do
for mr in $2; do
for r in $1; do
  ROCODER_ROUNDS=$r ROCODER_MIN_RUN=$mr timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('rounds $r min_run $mr', 'kernel_ms', r['roofline']['kernel_ms'], 'min', r['roofline']['kernel_ms_min'], 'step', r['ms_per_step'])"
done; done; done
