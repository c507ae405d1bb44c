#!/bin/bash
# A/B timing of engine library variants: tools/ab_bench.sh lib1.so lib2.so ...  (ms_per_step each)
for lib in "$@"; do
  export ROCODER_HIP_LIB=$PWD/rocoder_amd/$lib
  r=$(timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*')
  echo "$lib $r"
done
