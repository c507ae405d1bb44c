"""dev: device-to-host rate of ONE 16 MiB copy into pinned memory vs the same bytes split over 2 / 4 streams
(the streaming seam's batch copy)."""
import time
import torch
dev = torch.device("cuda", 0)
n = 4 << 20  # floats = 16 MiB
src = torch.rand(n, device=dev)
dst = torch.empty(n, pin_memory=True)
for parts in (1, 2, 4, 8):
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(20):
        t0 = time.perf_counter()
        for i, s in enumerate(streams):
            a, b = i * n // parts, (i + 1) * n // parts
            with torch.cuda.stream(s):
                dst[a:b].copy_(src[a:b], non_blocking=True)
        for s in streams:
            s.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(parts, "streams:", round(4 * n / best / 1e9, 1), "GB/s", round(best * 1e3, 3), "ms")
for mb in (1, 4, 16, 64):
    n2 = mb << 18
    s2 = torch.rand(n2, device=dev); d2 = torch.empty(n2, pin_memory=True)
    torch.cuda.synchronize(); best = 1e9
    for rep in range(20):
        t0 = time.perf_counter(); d2.copy_(s2, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(mb, "MiB single copy:", round(4 * n2 / best / 1e9, 1), "GB/s")
