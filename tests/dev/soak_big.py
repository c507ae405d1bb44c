"""dev: randomized differential run on the large-window kernels (N = 65536: big5_kernel, N = 32768: big4_kernel<32>):
factors, pitches 1 ... 5 and -2 / -3, 1 ... 9 channels (more channels than the run planner's rounds, odd counts),
ragged lengths incl. jobs shorter than a window, the computed default window and a caller's table window.
python tests/dev/soak_big.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocoder_amd as ra
from oracle import cbind as oc
from oracle import oracle_np as onp

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
worst, ran, t0 = 0.0, 0, time.time()
while ran < n_cases:
    N = int(rng.choice([65536, 65536, 65536, 32768]))
    f = float(np.round(np.exp(rng.uniform(np.log(0.3), np.log(64.0))), 3))
    p = int(rng.choice([1, 1, 1, 1, 2, 3, 5, -2, -3]))
    ch = int(rng.integers(1, 10))
    d = onp.derive(N, f, 1.0, p)
    if d["step"] < 1:
        continue
    hops = int(rng.integers(0, 60))
    L = int(max(0, hops * d["step"] + rng.integers(-N // 2, N)))
    if hops * ch > 240:
        continue
    table = bool(rng.integers(0, 4) == 0) and p >= 1
    x = np.stack([onp.synth_input(c, L) for c in range(ch)]) if L else np.zeros((ch, 0), np.float32)
    if not table:
        got = ra.stretch(x, window_len=N, factor=f, pitch_multiple=p, seed=ran)
        ref = oc.stretch_offline(x, N, f, 1.0, p, seed=ran)
    else:
        w = (oc.hanning(N).astype(np.float64) ** 1.5).astype(np.float32)
        with ra.Engine(window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=ran, window=w) as e:
            got = e.stretch_host(x)
        rows = []
        for c in range(ch):
            st = oc.Stretcher(channels=ch, factor=f, pitch_multiple=p, window=w, seed=ran, channel_index=c)
            st.send(x[c])
            st.close_input()
            wins = []
            while not st.is_done():
                wins.append(st.next_window())
            rows.append(np.concatenate(wins))
        ref = np.stack(rows)
    assert got.shape == ref.shape, (N, f, p, ch, L, got.shape, ref.shape)
    if L and ref.size:
        r = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
        e = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)))
        rel = e / r if r > 0 else e
        worst = max(worst, rel)
        if rel > 1.2e-6:
            print(f"  N={N} f={f} p={p} ch={ch} L={L} table={table}: {rel:.2e}", flush=True)
        assert rel <= 4e-6 or e <= 1e-9, (N, f, p, ch, L, table, rel)
    ran += 1
    if ran % 10 == 0:
        print(f"{ran} cases, worst relative RMS error {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {ran} cases, worst {worst:.2e}")
