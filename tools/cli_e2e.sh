#!/bin/bash
# dev: the C++ CLI end to end on the C2 job as files (600 s stereo f32 WAV in, 1.69 GB WAV out), phases on stderr (ROCODER_CLI_TIMING)
set -e
D=${1:-/dev/shm/rocoder_cli_e2e}
mkdir -p $D
python3 - "$D/in.wav" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
import wavutil
sr, n = 44100, 44100 * 600
t = np.arange(n, dtype=np.float64) / sr
x = np.stack([0.5 * np.sin(2 * np.pi * 220.0 * (c + 1) * t) for c in range(2)]).astype(np.float32)
wavutil.write_wav(sys.argv[1], x, sr, "f32")
PY
ls -la $D/in.wav
for i in 1 2; do
  t0=$(date +%s.%N)
  ROCODER_CLI_TIMING=1 rocoder_amd/bin/rocoder -i $D/in.wav -o $D/out.wav -w 16384 -f 8 --seed 1
  t1=$(date +%s.%N)
  python3 -c "print(\"wall %.2f s\" % ($t1 - $t0))"
  ls -la $D/out.wav
done
rm -rf $D
