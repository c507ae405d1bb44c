#!/bin/bash
source tools/gpu_steps.sh
step 900 pytest_grp python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "closed_job_group"
finish
