#!/bin/bash
source tools/gpu_steps.sh
export ROCODER_STAMPS=$PWD/gpurun_out/r05f_stamps.txt
ROCODER_HIP_LIB=$PWD/rocoder_amd/lib_stamp.so step 300 r05f_stamp python tests/dev/stamp_big.py
finish
