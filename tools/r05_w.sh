#!/bin/bash
source tools/gpu_steps.sh
step 600 r05w_e2e python tests/dev/e2e_host.py
finish
