#!/bin/bash
# A/B timing of engine library variants inside one gpurun call (same box, pre-heated bench each):
#   tools/ab_bench.sh lib_a.so lib_b.so ...   (files under rocoder_amd/, built by tools/build_variant.sh)
for lib in "$@"; do
  export ROCODER_HIP_LIB=$PWD/rocoder_amd/$lib
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', 'kernel_ms median', r['roofline']['kernel_ms'], 'min', r['roofline']['kernel_ms_min'], 'step', r['ms_per_step'])"
done
