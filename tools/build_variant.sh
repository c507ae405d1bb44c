#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags...] -> rocoder_amd/lib_NAME.so (A/B timing builds)
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rocoder_amd/csrc"
B=/tmp/rcvar_$NAME; mkdir -p $B
FLAGS="-DRC_PMAX=32 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize $*"
/opt/rocm/bin/hipcc $FLAGS -c rc_kernels.hip -o $B/k.o &
/opt/rocm/bin/hipcc $FLAGS -x hip -c rc_engine.cpp -o $B/e.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_$NAME.so $B/k.o $B/e.o
echo built ../lib_$NAME.so
