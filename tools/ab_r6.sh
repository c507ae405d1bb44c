#!/bin/bash
# round 6, item 1: same-box A/B of hop4 with (a) paired butterflies (RC_BF2) and (b) the product fold (RC_FOLDPROD)
source tools/gpu_steps.sh
for i in 1 2 3; do
  step 300 ab_r6_$i tools/ab_bench.sh lib_r6base.so lib_r6a.so lib_r6b.so librocoder_hip.so
done
step 900 pytest_gpu python -m pytest tests -m gpu -x -q
finish
