"""dev: one channel whose OUTPUT has more than 2^31 samples (L = 280 000 000, window 16384, factor 8: 2.24e9 samples, 8.96 GB):
hops near the start, across the 2^31-sample boundary and at the very end against the oracle's single-hop resynthesis."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import rocoder_amd as ra
from oracle import cbind as oc
from oracle import oracle_np as onp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
f = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
L = int(sys.argv[3]) if len(sys.argv) > 3 else 280_000_000
seed = 77
H = N // 2
g = torch.Generator(device="cuda"); g.manual_seed(3)
xt = (torch.rand((1, L), device="cuda", generator=g) - 0.5)
with ra.Engine(window_len=N, factor=f, channels=1, seed=seed) as e:
    n_out = e.output_len(L)
    print("n_out", n_out, "> 2^31:", n_out > 2 ** 31)
    out = e.stretch_tensor(xt)
    torch.cuda.synchronize()
    K = n_out // H
    d = onp.derive(N, f, 1.0, 1)
    step, amp = d["step"], np.float32(d["amp"])
    assert n_out > 2 ** 31
    r = oc.ReFFT(oc.hanning(N))
    env = oc.hanning_crossfade_compensation(H)
    kb = (2 ** 31) // H
    worst = 0.0
    for k in [0, 1, 5000, kb - 2, kb - 1, kb, kb + 1, K // 2 + 12345, K - 2, K - 1]:
        def y(kk):
            seg = np.zeros(N, np.float32)
            a, b = kk * step, min(kk * step + N, L)
            if b > a:
                seg[:b - a] = xt[0, a:b].cpu().numpy()
            return r.resynth(seg, oc.phase_key(seed, 0, kk))
        prev = y(k - 1)[H:] if k > 0 else np.zeros(H, np.float32)
        O = (y(k)[:H] + prev) * env * amp
        got = out[0, k * H:(k + 1) * H].cpu().numpy()
        err = float(np.sqrt(np.mean((got.astype(np.float64) - O) ** 2))) / float(np.sqrt(np.mean(O.astype(np.float64) ** 2)))
        worst = max(worst, err)
        print("hop", k, "rel err", f"{err:.2e}")
        assert err < 2e-6
    print("OK worst", f"{worst:.2e}", "kernel ms", e.last_kernel_stats())
