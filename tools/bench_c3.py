"""BASELINE C3 on one GPU (stereo, window 16384, factor 8, pitch 3, L = 26 460 000 per channel): pre-heated, median of
the per-launch kernel times."""
import json, os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd
dev = torch.device("cuda", 0)
x = (torch.rand((2, 26_460_000), device=dev) - 0.5)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    e = rocoder_amd.Engine(window_len=16384, factor=float(sys.argv[1]) if len(sys.argv) > 1 else 8.0, pitch_multiple=int(sys.argv[2]) if len(sys.argv) > 2 else 3, channels=2, seed=1)
    out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        for _ in range(4):
            e.stretch_tensor(x, out=out)
        stream.synchronize()
    for _ in range(10):
        e.stretch_tensor(x, out=out)
    stream.synchronize()
    ms = e.kernel_times(10)
    _, hops, _ = e.last_kernel_stats()
med = statistics.median(ms)
print(json.dumps(dict(ms_median=round(med, 3), ms_min=round(min(ms), 3), hops=hops, ns_per_hop=round(med * 1e6 / hops, 2),
                      frac_hbm=round(hops * 65536.0 / med / 1e6 / 8000.0, 4))))
