#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> rocoder_amd/lib_NAME.so (A/B timing / diagnostic builds;
# load it with ROCODER_HIP_LIB=rocoder_amd/lib_NAME.so). Add -DRC_TEST_HOOKS=1 for a variant that reads ROCODER_DIAG
# and the run-planner tuning variables (then also pass KERNELS through: see `make hooks`).
set -e
NAME=$1; shift
make -s -j6 -C "$(dirname "$0")/../rocoder_amd/csrc" variant NAME="$NAME" EXTRA="$*"
echo built rocoder_amd/lib_$NAME.so
