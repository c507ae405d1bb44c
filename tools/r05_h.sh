#!/bin/bash
source tools/gpu_steps.sh
step 900 r05h_soak_big python tests/dev/soak_big.py 80 11
step 900 r05h_soak python tests/dev/soak.py 150 23
step 300 r05h_smoke python __graft_entry__.py smoke
finish
