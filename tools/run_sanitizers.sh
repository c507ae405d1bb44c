#!/bin/bash
# Host sanitizer runs of the C++ CLI twin (CPU only; GPU ASan / XNACK are not available on this pool):
#   1. AddressSanitizer + UndefinedBehaviorSanitizer build of host/rocoder_cli.cpp over the real engine library:
#      tests/test_cli_host.py (duration grammar, WAV decoding incl. the 230-file hostile corpus, flags)
#   2. ThreadSanitizer build over tests/c/stub_engine.c (computes nothing): file-to-file runs that exercise the
#      StretcherProcessor thread, the bounded WindowQueues, the AudioBus drain and the kernel hot-swap watcher
#   3. the ENGINE's own host code (rc_engine.cpp) built host-only over tests/c/hip_stub.cpp (device memory = host
#      memory, kernels compute nothing) and driven by tests/c/engine_host_driver.cpp: once with ASan + UBSan, once with
#      TSan - worker pools, the three-set pinned pipeline of the frequency-kernel path, rc_multi's persistent
#      workers on {0,0,0} / {0,1,2} / eight entries, the streaming seam pushed and pulled from different threads
# usage: tools/run_sanitizers.sh [log]      (default log: profiles/r04_sanitizers.txt)
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r04_sanitizers.txt}
cd "$ROOT"
{
echo "# host sanitizer runs, $(date -u +%Y-%m-%dT%H:%MZ), $(g++ --version | head -1)"
make -s -C rocoder_amd/csrc all >/dev/null && make -s -C rocoder_amd/csrc/host -f sanitize.mk >/dev/null || { echo "BUILD FAILED"; exit 1; }
echo "## 1. ASan + UBSan: tests/test_cli_host.py on rocoder_amd/bin/rocoder_asan (leak detection on)"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 ROCODER_CLI=$ROOT/rocoder_amd/bin/rocoder_asan \
    python -m pytest tests/test_cli_host.py -q 2>&1 | tail -3
echo "## 2. TSan: rocoder_amd/bin/rocoder_tsan over the stub engine"
T=$(mktemp -d)
python - "$T" <<'PY'
import sys, numpy as np
sys.path.insert(0, "tests")
from wavutil import write_wav
rng = np.random.default_rng(3)
write_wav(sys.argv[1] + "/in2.wav", rng.uniform(-1, 1, (2, 200_000)), 44100, "i16")
write_wav(sys.argv[1] + "/in5.wav", rng.uniform(-1, 1, (5, 30_001)), 8000, "f32")
open(sys.argv[1] + "/k.c", "w").write('#include <stddef.h>\n#include <stdint.h>\nint apply(uint64_t t, const float *in, float *out, size_t n, void *u) { for (size_t i = 0; i < 2 * n; ++i) out[i] = 2.0f * in[i]; return 0; }\n')
PY
fail=0
run() { echo "+ rocoder_tsan $*"; TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 "$ROOT/rocoder_amd/bin/rocoder_tsan" "$@" > "$T/out.txt" 2> "$T/err.txt"; rc=$?; grep -c "WARNING: ThreadSanitizer" "$T/err.txt" | sed 's/^/  ThreadSanitizer warnings: /'; echo "  exit $rc"; [ $rc -ne 0 ] && { fail=1; tail -5 "$T/err.txt"; }; grep -q "WARNING: ThreadSanitizer" "$T/err.txt" && { fail=1; grep -A12 "WARNING: ThreadSanitizer" "$T/err.txt" | head -40; }; }
run -i "$T/in2.wav" -o "$T/o1.wav" -w 1024 -f 2
run -i "$T/in2.wav" -o "$T/o2.wav" -w 16384 -f 8 -b 0.05
run -i "$T/in5.wav" -o "$T/o3.wav" -w 512 -f 1 --rotate-channels -s 0.5 -d 2
run -i "$T/in2.wav" -o "$T/o4.wav" -w 2048 -f 2 --freq-kernel "$T/k.c"
rm -rf "$T"
echo "## 3. the engine's host code over the HIP stub (tests/c/hip_stub.cpp + engine_host_driver.cpp)"
echo "+ engine_asan (ASan + UBSan, leak detection on)"
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 "$ROOT/rocoder_amd/bin/engine_asan" > "$ROOT/.san_out.txt" 2>&1; rc=$?
tail -1 "$ROOT/.san_out.txt" | sed 's/^/  /'; echo "  exit $rc"
[ $rc -ne 0 ] || grep -q "Sanitizer\|runtime error" "$ROOT/.san_out.txt" && { [ $rc -ne 0 ] && fail=1; grep -q "Sanitizer\|runtime error" "$ROOT/.san_out.txt" && { fail=1; head -40 "$ROOT/.san_out.txt"; }; }
echo "+ engine_tsan"
TSAN_OPTIONS=halt_on_error=0:second_deadlock_stack=1 "$ROOT/rocoder_amd/bin/engine_tsan" > "$ROOT/.san_out.txt" 2>&1; rc=$?
grep -c "WARNING: ThreadSanitizer" "$ROOT/.san_out.txt" | sed 's/^/  ThreadSanitizer warnings: /'
tail -1 "$ROOT/.san_out.txt" | sed 's/^/  /'; echo "  exit $rc"
[ $rc -ne 0 ] && fail=1
grep -q "WARNING: ThreadSanitizer" "$ROOT/.san_out.txt" && { fail=1; grep -A12 "WARNING: ThreadSanitizer" "$ROOT/.san_out.txt" | head -60; }
rm -f "$ROOT/.san_out.txt"
[ $fail -eq 0 ] && echo "RESULT: clean" || echo "RESULT: FINDINGS (above)"
} 2>&1 | tee "$LOG"
