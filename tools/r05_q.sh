#!/bin/bash
source tools/gpu_steps.sh
step 600 r05q_tests_big python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "65536 or c5 or C5 or big or i8"
step 900 r05q_ab_c5 tools/ab_c5.sh 3 lib_b5nobuf.so librocoder_hip.so
finish
