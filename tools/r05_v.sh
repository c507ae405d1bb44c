#!/bin/bash
source tools/gpu_steps.sh
step 600 r05v_e2e bash -c 'for m in lib_w8.so librocoder_hip.so lib_w8.so librocoder_hip.so; do echo "== $m"; ROCODER_HIP_LIB=$PWD/rocoder_amd/$m python tests/dev/e2e_host.py | grep -v "^multi"; done; nproc'
finish
