#!/bin/bash
source tools/gpu_steps.sh
step 600 r05c_tests_big python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "65536 or c5 or C5 or big"
step 900 r05c_ab_c5 bash -c 'for i in 1 2 3; do echo big4; ROCODER_HIP_LIB=$PWD/rocoder_amd/librocoder_hip_hooks.so ROCODER_DIAG=8 python tools/bench_c5.py 2>/dev/null; echo big5; python tools/bench_c5.py 2>/dev/null; done'
step 600 r05c_shards bash -c 'for i in 1 2; do ROCODER_HIP_LIB=$PWD/rocoder_amd/lib_planold.so python tools/bench_shards.py 2>/dev/null; python tools/bench_shards.py 2>/dev/null; done'
finish
