import sys, time, numpy as np
sys.path.insert(0, '.')
import rocoder_amd
from oracle import cbind as oc, oracle_np as onp
for (N, L, f, p, ch) in ((32768, 200000, 8.0, 1, 2), (65536, 400000, 32.0, 1, 2), (65536, 300000, 4.0, 2, 1), (32768, 32768, 1.0, 1, 1), (65536, 1000, 2.0, 1, 1), (32768, 150000, 4.0, 3, 1)):
    x = np.stack([onp.synth_input(c, L) for c in range(ch)])
    got = rocoder_amd.stretch(x, window_len=N, factor=f, pitch_multiple=p, seed=0xC5)
    ref = oc.stretch_offline(x, N, f, 1.0, p, seed=0xC5)
    err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))); rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print("N", N, "L", L, "f", f, "p", p, "err", err, "rms", rms, "OK" if err <= 1e-4 * max(rms, 1e-3) else "BAD", flush=True)
