#!/bin/bash
source tools/gpu_steps.sh
step 300 r05u_soak_big python tests/dev/soak_big.py 60 5
step 850 r05u_soak python tests/dev/soak.py 60 31
finish
