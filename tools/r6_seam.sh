#!/bin/bash
source tools/gpu_steps.sh
step 300 seam_bench rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 2
step 300 seam_bench_nopin rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 2 nopin
step 300 seam_bench2 rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 2
step 300 seam_bench_mono rocoder_amd/bin/seam_bench rocoder_amd/librocoder_hip.so 1
finish
