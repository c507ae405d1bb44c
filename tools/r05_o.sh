#!/bin/bash
source tools/gpu_steps.sh
step 600 r05o_tests python -m pytest tests/test_gpu_multi.py -q -x -k "rehearsal or rccl"
finish
