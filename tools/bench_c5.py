"""BASELINE C5 on one GPU (8 channels, window 65536, factor 32): wall time per job; run under
rocprofv3 --kernel-trace --stats for the per-kernel split of the large-window pipeline."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd  # noqa: E402

dev = torch.device("cuda", 0)
x8 = (torch.rand((8, 5_292_000), device=dev) - 0.5)
e = rocoder_amd.Engine(window_len=65536, factor=32.0, channels=8, seed=1)
out = torch.empty((8, e.output_len(x8.shape[1])), device=dev)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    e.stretch_tensor(x8, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.stretch_tensor(x8, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
hops = out.shape[1] // 32768 * 8
print(f"C5 one GPU: {dt * 1e3:.2f} ms per job, {hops} hops, {hops / dt / 1e6:.2f} M hops/s, "
      f"{out.numel() / dt / 1e9:.1f} Gsamples/s")
