#!/bin/bash
source tools/gpu_steps.sh
step 1100 r05r_profile_c2 tools/profile_round.sh r05e pmc
step 900 r05r_profile_c5 tools/profile_c5.sh r05e_c5
finish
