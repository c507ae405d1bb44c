#!/bin/bash
source tools/gpu_steps.sh
step 1000 r05d_ab_c5 tools/ab_c5.sh 2 librocoder_hip.so lib_b5tl4.so lib_b5tl3.so lib_b5k3.so lib_b5k6.so
step 300 r05d_stream python tests/dev/stream_rate.py
step 600 r05d_tests python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -x -k "stream or view or cli or seam"
finish
