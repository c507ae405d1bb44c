#!/bin/bash
source tools/gpu_steps.sh
step 900 r05a_tests_new python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -q -x -k "host_pipeline or c4_at_its_own or i8_corners or rccl"
step 600 r05a_bench python bench.py
step 120 r05a_bench_gpus2 python bench.py --gpus 2
ROCODER_BENCH_REHEARSAL=1 step 600 r05a_rehearsal4 python bench.py --gpus 4 --steps 5 --warmup 2
step 600 r05a_e2e_host python tests/dev/e2e_host.py
grep -h '^{' gpurun_out/r05a_bench.log > gpurun_out/r05a_bench.json
finish
