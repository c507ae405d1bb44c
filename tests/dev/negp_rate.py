import os, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rocoder_amd
dev = torch.device("cuda", 0)
x = (torch.rand((2, 13_230_000), device=dev) - 0.5)
for N, p in ((16384, -2), (16384, 1), (4096, -2), (16384, -3)):
    e = rocoder_amd.Engine(window_len=N, factor=8.0, pitch_multiple=p, channels=2, seed=1)
    out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
    for _ in range(3): e.stretch_tensor(x, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): e.stretch_tensor(x, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(N, p, round(dt * 1e3, 2), "ms", out.shape[1])
    e.close(); del out
