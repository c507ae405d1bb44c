#!/bin/bash
source tools/gpu_steps.sh
step 400 r05k_bench python bench.py
grep -h '^{' gpurun_out/r05k_bench.log > gpurun_out/r05k_bench_$(date +%s).json
finish
