import sys, numpy as np
sys.path.insert(0, '.')
import rocoder_amd
from oracle import cbind as oc, oracle_np as onp
for (L, f, ch) in ((100000, 1.0, 1), (150000, 8.0, 2)):
    x = np.stack([onp.synth_input(c, L) for c in range(ch)])
    got = rocoder_amd.stretch(x, window_len=16384, factor=f, seed=0x5EED)
    ref = oc.stretch_offline(x, 16384, f, 1.0, 1, seed=0x5EED)
    err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))); rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print("ok", L, f, "err", err, "rms", rms, flush=True)
