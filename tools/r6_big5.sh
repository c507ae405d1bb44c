#!/bin/bash
source tools/gpu_steps.sh
step 900 pytest_big python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -q -x -k "65536 or run_seams or seam_wait"
finish
