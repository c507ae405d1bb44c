#!/bin/bash
# round 6, item 2: big5 with run seams (short runs per XCD, stash hand-over): parity first, then the same-box A/B
source tools/gpu_steps.sh
step 900 pytest_big python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "65536 or C5 or big5 or large or closed_job or multi_device_equals"
step 600 ab_c5 tools/ab_c5.sh 3 lib_b5aux2.so librocoder_hip.so
finish
