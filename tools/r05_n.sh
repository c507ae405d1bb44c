#!/bin/bash
source tools/gpu_steps.sh
step 600 r05n_tests_big python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "65536 or c5 or C5 or big or i8"
step 900 r05n_profile_c5 tools/profile_c5.sh r05d_c5
finish
