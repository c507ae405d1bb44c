#!/bin/bash
# round 6, item 1: same-box A/B of hop4 with (a) paired butterflies (RC_BF2) and (b) the product fold (RC_FOLDPROD):
#   tools/build_variant.sh r6base -DRC_BF2=0 -DRC_FOLDPROD=0 (round 5's kernel) against the product, alternating
source tools/gpu_steps.sh
for i in 1 2 3 4 5 6; do
  step 300 ab_r6_$i tools/ab_bench.sh lib_r6base.so librocoder_hip.so
done
finish
