#!/bin/bash
source tools/gpu_steps.sh
step 600 r05i_tests python -m pytest tests/test_gpu_parity.py -q -x -k "caller_window or table or hop_slots or random_configurations or default_window_detection"
step 300 r05i_smoke python __graft_entry__.py smoke
step 600 r05i_windows python tools/bench_windows.py
finish
