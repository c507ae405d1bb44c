#!/bin/bash
source tools/gpu_steps.sh
step 700 r05j_tests python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "table or caller_window or 16384 or random_configurations or multi_device_equals"
step 600 r05j_windows python tools/bench_windows.py
ROCODER_HIP_LIB=$PWD/rocoder_amd/lib_hop2tab.so step 600 r05j_windows_hop2 python tools/bench_windows.py
finish
