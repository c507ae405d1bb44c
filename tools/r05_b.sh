#!/bin/bash
source tools/gpu_steps.sh
step 1100 r05b_tests_gpu python -m pytest tests -q -x -m gpu
step 900 r05b_ab_hop4 bash -c 'for i in 1 2 3; do tools/ab_bench.sh lib_hop4r04.so librocoder_hip.so; done'
step 600 r05b_e2e_pop bash -c 'for m in lib_pop0.so lib_pop1.so lib_pop2.so librocoder_hip.so; do echo "== $m"; ROCODER_HIP_LIB=$PWD/rocoder_amd/$m python tests/dev/e2e_host.py | grep -v "^multi"; done; cat /sys/kernel/mm/transparent_hugepage/enabled; nproc'
step 600 r05b_bench python bench.py
grep -h '^{' gpurun_out/r05b_bench.log > gpurun_out/r05b_bench.json
finish
