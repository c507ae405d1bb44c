"""BASELINE C5 on one GPU (8 channels, window 65536, factor 32, L = 5 292 000 per channel) and the same job at
window 32768: pre-heated, median of the per-launch event times the engine records. ROCODER_DIAG=2 runs the
previous three-kernel pipeline instead of the fused big4_kernel."""
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
res = {"diag": os.environ.get("ROCODER_DIAG", "0")}
x8 = (torch.rand((8, 5_292_000), device=dev) - 0.5)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    for N, f in ((65536, 32.0), (32768, 32.0)):
        e = rocoder_amd.Engine(window_len=N, factor=f, channels=8, seed=1)
        out = torch.empty((8, e.output_len(x8.shape[1])), device=dev)
        e.stretch_tensor(x8, out=out)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(4):
                e.stretch_tensor(x8, out=out)
            stream.synchronize()
        with ClockSampler(0) as clk:
            for _ in range(40):
                e.stretch_tensor(x8, out=out)
            stream.synchronize()
        ms = e.kernel_times(10)
        _, hops, launches = e.last_kernel_stats()
        med = statistics.median(ms)
        res[f"N{N}"] = dict(ms_median=round(med, 3), ms_min=round(min(ms), 3), hops=hops, launches=launches,
                            hops_per_s=round(hops / med * 1e3), out_gsamples_s=round(out.numel() / med / 1e6, 1),
                            algo_read_GBs=round(hops * 4.0 * N / med / 1e6, 1),
                            frac_hbm=round(hops * 4.0 * N / med / 1e6 / 8000.0, 4), sclk_mhz_under_load=clk.median_mhz())
        e.close()
        del out
print(json.dumps(res))
