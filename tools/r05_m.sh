#!/bin/bash
source tools/gpu_steps.sh
step 1100 r05m_tests_gpu python -m pytest tests -q -x -m gpu
step 300 r05m_smoke python __graft_entry__.py smoke
step 400 r05m_bench python bench.py
grep -h '^{' gpurun_out/r05m_bench.log > gpurun_out/r05m_bench.json
finish
