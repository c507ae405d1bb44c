#!/bin/bash
source tools/gpu_steps.sh
step 600 r05p_tests python -m pytest tests/test_gpu_parity.py -q -x -k "16384 or C2 or C3 or edge_lengths or seam or table"
step 900 r05p_ab bash -c 'for i in 1 2 3 4; do tools/ab_bench.sh lib_nobuf.so librocoder_hip.so; done'
finish
