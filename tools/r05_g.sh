#!/bin/bash
source tools/gpu_steps.sh
step 1100 r05g_tests_gpu python -m pytest tests -q -x -m gpu
step 400 r05g_bench python bench.py
grep -h '^{' gpurun_out/r05g_bench.log > gpurun_out/r05g_bench.json
step 300 r05g_torchrun1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-extras
ROCODER_BENCH_REHEARSAL=1 step 600 r05g_rehearsal4 python bench.py --gpus 4 --steps 5 --warmup 2
step 120 r05g_gpus2 python bench.py --gpus 2
finish
