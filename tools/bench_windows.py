"""Throughput by window length (stereo, factor 8, default hanning window, inputs resident in HBM): which kernel
serves which -w and how far each is from the N = 16384 bench kernel. One GPU, pre-heated, median per-launch time."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd  # noqa: E402
from boxclock import ClockSampler  # noqa: E402

dev = torch.device("cuda", 0)
x = (torch.rand((2, 13_230_000), device=dev) - 0.5)
res = {}
stream = torch.cuda.Stream(dev)
clk = ClockSampler(0)
clk.__enter__()
with torch.cuda.stream(stream):
    # "t" rows: a caller-supplied window (hanning ** 1.5: not the default, so the table-window kernels run -
    # hop2_kernel at 16384, the TABW instantiations of the wave-local kernels at 512 ... 8192 since round 5)
    for N in (64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 12000, 24000, "16384t", "8192t", "4096t", "2048t", "1024t", "512t"):
        table = isinstance(N, str)
        tag = N
        N = int(N[:-1]) if table else N
        short = not (N >= 64 and (N & (N - 1)) == 0)
        xs = x[:, :200_000].contiguous() if short else (x[:, :2_000_000].contiguous() if N < 512 else x)
        kw = {}
        if table:
            import numpy as np
            w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / (N - 1))) ** 1.5
            kw["window"] = w.astype(np.float32)
        e = rocoder_amd.Engine(window_len=N, factor=8.0, channels=2, seed=1, **kw)
        out = torch.empty((2, e.output_len(xs.shape[1])), device=dev)
        e.stretch_tensor(xs, out=out)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.7:
            for _ in range(2):
                e.stretch_tensor(xs, out=out)
            stream.synchronize()
        for _ in range(6):
            e.stretch_tensor(xs, out=out)
        stream.synchronize()
        ms = statistics.median(e.kernel_times(6))
        _, hops, launches = e.last_kernel_stats()
        res[f"N{tag}"] = dict(kernel_ms=round(ms, 3), hops=hops, launches=launches, in_samples=xs.shape[1],
                            out_gsamples_s=round(out.numel() / ms / 1e6, 1),
                            frac_hbm_read=round(hops * 4.0 * N / ms / 1e6 / 8000.0, 4))
        e.close()
        del out
clk.__exit__()
res["sclk_mhz_under_load"] = clk.median_mhz()
print(json.dumps(res, indent=1))
