#!/bin/bash
# Runs GPU steps one after the other on the gpurun box; a step that times out or is killed ends the sequence (no
# further GPU step is started after a hang), an ordinary failure (a red test) does not.
#   usage: source tools/gpu_steps.sh; step <seconds> <name> <command...>
mkdir -p gpurun_out
STOP=0
FAIL=0
step() {
    local limit=$1 name=$2; shift 2
    if [ "$STOP" != 0 ]; then echo "[skip $name: an earlier step was killed]"; return; fi
    echo "=== $name"
    timeout -k 10 "$limit" "$@" > "gpurun_out/$name.log" 2>&1
    local rc=$?
    echo "[$name rc=$rc]"
    tail -n 6 "gpurun_out/$name.log"
    if [ $rc -ne 0 ]; then FAIL=1; fi  # (a red step must fail the script, not only a GPU fault)
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then STOP=1; fi
    # a GPU fault (an out-of-bounds access in a kernel) ends the sequence too: nothing else runs on that box
    if grep -q "Memory access fault\|HSA_STATUS_ERROR\|GPU core dump" "gpurun_out/$name.log"; then STOP=1; FAULT=1; fi
}
FAULT=0
finish() {
    if [ "$FAULT" != 0 ]; then echo "[a step faulted on the GPU]"; exit 1; fi
    if [ "$STOP" != 0 ]; then echo "[a step was killed at its limit]"; exit 1; fi
    if [ "$FAIL" != 0 ]; then echo "[a step failed]"; exit 1; fi
    exit 0
}
