#!/bin/bash
# One gpurun call that re-checks a tree: the whole GPU suite, smoke(), the default bench line.
#   gpurun --timeout 1200 -- 'bash tools/final_check.sh'   -> gpurun_out/final_*.log, final_bench.json
# (tools/gpu_steps.sh: a step that times out or faults on the GPU ends the sequence; any failed step, a red test run or a
# bench without its JSON line fails the script)
source tools/gpu_steps.sh
step 1100 final_tests_gpu python -m pytest tests -q -x -m gpu
step 300 final_smoke python __graft_entry__.py smoke
step 400 final_bench python bench.py
grep -h '^{' gpurun_out/final_bench.log > gpurun_out/final_bench.json
if ! grep -q '^{' gpurun_out/final_bench.json; then echo "[final_bench printed no JSON line]"; FAIL=1; fi
finish
