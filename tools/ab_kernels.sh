#!/bin/bash
# A/B of the two N = 16384 kernel generations inside ONE gpurun call (same box): hop4 (default) vs the
# previous one (ROCODER_DIAG=2), alternating, pre-heated bench each. usage: tools/ab_kernels.sh [rounds]
R=${1:-2}
# only the test-hook build (make hooks) reads ROCODER_DIAG; it runs hop4 under ROCODER_DIAG=0
export ROCODER_HIP_LIB=$PWD/rocoder_amd/librocoder_hip_hooks.so
for i in $(seq 1 $R); do
  for d in 0 2; do
    ROCODER_DIAG=$d python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null \
      | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('diag=$d', r['roofline']['kernel_id'] if $d==0 else 'previous', 'kernel_ms median', r['roofline']['kernel_ms'], 'min', r['roofline']['kernel_ms_min'], 'step', r['ms_per_step'])"
  done
done
