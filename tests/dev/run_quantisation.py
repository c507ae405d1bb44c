"""dev: does the last, partly filled round of runs cost hop4_kernel time? C2 has 67.26 hops per resident workgroup slot in runs
of 9 (47 % of the workgroups walk 8 runs, the rest 7); jobs with exactly 64 / 68 / 72 hops per slot run at the same 27.0-27.2 ns
per hop - a workgroup that finishes early leaves its CU to the two that remain (round 4; no run-length mixing needed)."""
import statistics, sys, time
import torch
sys.path.insert(0, '.')
import rocoder_amd
dev = torch.device("cuda", 0)
res = {}
for hops_ch in (25826, 24576, 25826, 26112, 25826, 27648, 25826):
    L = (hops_ch - 1) * 1024 + 16384
    x = (torch.rand((2, L), device=dev) - 0.5)
    e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1)
    out = torch.empty((2, e.output_len(L)), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        for _ in range(8): e.stretch_tensor(x, out=out)
        torch.cuda.synchronize()
    for _ in range(20): e.stretch_tensor(x, out=out)
    torch.cuda.synchronize()
    ms = statistics.median(e.kernel_times(20)); _, hops, _ = e.last_kernel_stats()
    print(hops_ch, hops, round(ms, 4), "ns/hop", round(ms * 1e6 / hops, 2), "hops per WG slot", round(hops / 768, 2))
    e.close()
