import json, os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import rocoder_amd
from oracle import cbind as oc
from oracle import oracle_np as onp
dev = torch.device("cuda", 0)
res = {}
# parity at tiny windows, several channels, pitches, ragged lengths
for N, f, p, ch, L in ((32, 2.0, 1, 3, 5000), (64, 4.0, 1, 2, 20000), (64, 1.5, 2, 1, 777), (128, 8.0, 3, 5, 30001), (256, 4.0, 1, 2, 50000), (256, 2.0, 2, 7, 9999), (128, 0.3, 1, 2, 20000), (64, 2.0, 1, 9, 100)):
    x = np.stack([onp.synth_input(c, L) for c in range(ch)])
    got = rocoder_amd.stretch(x, window_len=N, factor=f, pitch_multiple=p, seed=3)
    ref = oc.stretch_offline(x, N, f, 1.0, p, seed=3)
    err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))); r = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    print(N, f, p, ch, L, got.shape == ref.shape, err / max(r, 1e-30))
    assert got.shape == ref.shape and err <= 2e-6 * r + 1e-9
    w = (oc.hanning(N).astype(np.float64) ** 1.5).astype(np.float32)
    with rocoder_amd.Engine(window_len=N, factor=f, pitch_multiple=p, channels=ch, seed=3, window=w) as e:
        got = e.stretch_host(x)
    chans = []
    for c in range(ch):
        st = oc.Stretcher(channels=ch, factor=f, pitch_multiple=p, window=w, seed=3, channel_index=c)
        st.send(x[c]); st.close_input()
        wins = []
        while not st.is_done(): wins.append(st.next_window())
        chans.append(np.concatenate(wins))
    ref = np.stack(chans)
    err = float(np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2))); r = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    assert got.shape == ref.shape and err <= 2e-6 * r + 1e-9, (N, err / r)
x = (torch.rand((2, 2_000_000), device=dev) - 0.5)
for N in (32, 64, 128, 256):
    e = rocoder_amd.Engine(window_len=N, factor=8.0, channels=2, seed=1)
    out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
    for _ in range(20): e.stretch_tensor(x, out=out)
    torch.cuda.synchronize()
    ms = statistics.median(e.kernel_times(10)); _, hops, _ = e.last_kernel_stats()
    res[N] = (round(ms, 3), round(hops * 4.0 * N / ms / 1e6 / 8000.0, 4))
    e.close()
print(json.dumps(res))
