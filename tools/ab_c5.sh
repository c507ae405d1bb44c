#!/bin/bash
# A/B of engine library variants on BASELINE C5 (tools/bench_c5.py) inside one gpurun call, alternating:
#   tools/ab_c5.sh [rounds] lib_a.so lib_b.so ...   (files under rocoder_amd/)
R=$1; shift
for i in $(seq 1 $R); do
for lib in "$@"; do
  ROCODER_HIP_LIB=$PWD/rocoder_amd/$lib timeout -k 10 200 python tools/bench_c5.py 2>/dev/null \
    | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$lib', 'C5', r['N65536']['ms_median'], 'min', r['N65536']['ms_min'], ' twin', r['N32768']['ms_median'])"
done; done
