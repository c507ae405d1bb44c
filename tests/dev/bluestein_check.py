"""dev: chirp-z path (even window lengths that are not a power of two, N <= 16384) against the oracle: relative RMS
error and time per job (tests/test_gpu_parity.py holds the gate)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocoder_amd as ra
from oracle import cbind as oc
from oracle import oracle_np as onp
for N, L, f in ((36, 700, 0.3), (1000, 20000, 4.0), (3000, 12000, 8.0), (6000, 30000, 2.0), (12000, 60000, 4.0), (16382, 30000, 2.0), (16390, 30000, 2.0), (24000, 30000, 2.0), (32770, 50000, 2.0), (50000, 60000, 2.0), (65534, 66000, 1.0)):
    x = np.stack([onp.synth_input(0, L)])
    t0 = time.perf_counter(); got = ra.stretch(x, window_len=N, factor=f, seed=7); t1 = time.perf_counter()
    got = ra.stretch(x, window_len=N, factor=f, seed=7); t2 = time.perf_counter()
    ref = oc.stretch_offline(x, N, f, 1.0, 1, seed=7)
    err = np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    print(f"N={N:6d} hops~{int(L*f/(N/2)):6d} rel rms {err:.2e}  second call {1e3*(t2-t1):8.2f} ms")
