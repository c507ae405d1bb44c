#!/bin/bash
source tools/gpu_steps.sh
step 700 r05t_tests python -m pytest tests/test_gpu_parity.py -q -x -k "16384 or table or caller_window"
step 600 r05t_windows python tools/bench_windows.py
step 1100 r05t_profile_c2 tools/profile_round.sh r05f pmc
finish
