#!/bin/bash
source tools/gpu_steps.sh
step 800 r05x_e2e_pop bash -c 'for m in lib_pop0.so librocoder_hip.so lib_pop3.so lib_pop0.so librocoder_hip.so lib_pop3.so; do echo "== $m"; ROCODER_HIP_LIB=$PWD/rocoder_amd/$m python tests/dev/e2e_host.py | grep "^engine pageable"; done'
finish
