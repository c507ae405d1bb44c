#!/bin/bash
# round 6, item 4 / 6: the streaming seam from C (closed stereo C2 job: copy / view; live latency), then the GPU suite
source tools/gpu_steps.sh
step 300 seam_bench rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 2
step 300 seam_bench_mono rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 1
step 1000 pytest_gpu python -m pytest tests -m gpu -q
finish
