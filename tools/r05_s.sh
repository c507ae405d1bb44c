#!/bin/bash
source tools/gpu_steps.sh
step 700 r05s_tests python -m pytest tests/test_gpu_parity.py -q -x -k "stretch_matches_oracle or caller_window or random_configurations or edge_lengths or streaming"
step 600 r05s_windows python tools/bench_windows.py
finish
