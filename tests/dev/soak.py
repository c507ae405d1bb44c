"""dev: a longer randomized differential run than tests/test_gpu_parity.py::test_random_configurations_match_oracle
(power-of-two and other even window lengths, pitch -4 ... 7, 1-3 channels, ragged lengths), emphasising the window
lengths of the wave-local kernels. python tests/dev/soak.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd as ra
from oracle import cbind as oc
from oracle import oracle_np as onp

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
worst, ran, t0 = 0.0, 0, time.time()
while ran < n_cases:
    kind = rng.integers(0, 10)
    if kind < 6:
        N = int(rng.choice([512, 1024, 2048, 4096, 8192, 16384]))
    elif kind < 8:
        N = 1 << int(rng.integers(5, 17))
    else:
        N = 2 * int(rng.integers(2, 9000))
    f = float(np.round(np.exp(rng.uniform(np.log(0.1), np.log(40.0))), 3))
    p = int(rng.choice([-4, -3, -2, 1, 1, 1, 2, 2, 3, 3, 4, 5, 7]))
    ch = int(rng.integers(1, 4))
    d = onp.derive(N, f, 1.0, p)
    if d["step"] < 1:
        continue
    hops = int(rng.integers(0, 400 if N <= 16384 else 40))
    L = int(max(0, hops * d["step"] + rng.integers(-N // 2, N)))
    if L * max(f, 1.0) * ch > 8e6 or ((N & (N - 1)) and hops * N * N > 3e10):
        continue
    x = np.stack([onp.synth_input(c, L) for c in range(ch)]) if L else np.zeros((ch, 0), np.float32)
    got = ra.stretch(x, window_len=N, factor=f, pitch_multiple=p, seed=ran)
    ref = oc.stretch_offline(x, N, f, 1.0, p, seed=ran)
    assert got.shape == ref.shape, (N, f, p, ch, L, got.shape, ref.shape)
    if L and ref.size:
        r = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        e = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
        rel = e / r if r > 0 else e
        worst = max(worst, rel)
        if rel > 7e-7:
            print(f"  N={N} f={f} p={p} ch={ch} L={L}: {rel:.2e}", flush=True)
        assert rel <= 2e-6 or e <= 1e-9, (N, f, p, ch, L, rel)
    ran += 1
    if ran % 25 == 0:
        print(f"{ran} cases, worst relative RMS error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ran} cases, worst {worst:.2e}")
