"""dev: randomized differential run on the window lengths of the generic kernel's hop slots (N = 32 ... 256; a caller's
table window at 512 / 1024 too): 1-9 channels, pitch -4 ... 7, ragged lengths, factors 0.1 ... 40, both window kinds.
python -u tests/dev/soak_small.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocoder_amd as ra
from oracle import cbind as oc
from oracle import oracle_np as onp

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
worst, ran, t0 = 0.0, 0, time.time()
while ran < n_cases:
    N = int(rng.choice([32, 64, 128, 256, 256, 512, 1024]))
    table = bool(rng.integers(0, 2)) or N >= 512
    f = float(np.round(np.exp(rng.uniform(np.log(0.1), np.log(40.0))), 3))
    p = int(rng.choice([-4, -3, -2, 1, 1, 1, 2, 2, 3, 3, 4, 5, 7]))
    ch = int(rng.integers(1, 10))
    d = onp.derive(N, f, 1.0, p)
    if d["step"] < 1:
        continue
    hops = int(rng.integers(0, 3000))
    L = int(max(0, hops * d["step"] + rng.integers(-N // 2, N)))
    if L * max(f, 1.0) * ch > 4e6:
        continue
    x = np.stack([onp.synth_input(c, L) for c in range(ch)]) if L else np.zeros((ch, 0), np.float32)
    w = (oc.hanning(N).astype(np.float64) ** 1.5).astype(np.float32) if table else None
    with ra.Engine(window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=ran, **({"window": w} if table else {})) as e:
        got = e.stretch_host(x)
    if not table:
        ref = oc.stretch_offline(x, N, f, 1.0, p, seed=ran)
    else:
        chans = []
        for c in range(ch):
            st = oc.Stretcher(channels=ch, factor=f, pitch_multiple=p, window=w, seed=ran, channel_index=c)
            st.send(x[c]); st.close_input()
            wins = []
            while not st.is_done():
                wins.append(st.next_window())
            chans.append(np.concatenate(wins))
        ref = np.stack(chans)
    assert got.shape == ref.shape, (N, f, p, ch, L, table, got.shape, ref.shape)
    if L and ref.size:
        r = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        e_ = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
        rel = e_ / r if r > 0 else e_
        worst = max(worst, rel)
        assert rel <= 1e-5, (N, f, p, ch, L, table, rel)
    ran += 1
    if ran % 25 == 0:
        print(f"{ran} cases, worst relative RMS error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ran} cases, worst {worst:.2e}")
