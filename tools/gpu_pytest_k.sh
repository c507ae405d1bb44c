#!/bin/bash
# usage: tools/gpu_pytest_k.sh "<-k expression>" [test files...]: one pytest process on the gpurun box, log under gpurun_out/
source tools/gpu_steps.sh
K=$1; shift
step 900 pytest_k python -m pytest ${@:-tests} -m gpu -q -x -k "$K"
finish
