"""A/B of run-planner settings of the N = 16384 kernel (or ROCODER_AB_N=4096 / 8192: the wave-local kernels, 13.2 M
samples per channel) inside ONE process, interleaved (cancels clock / thermal drift):
python tools/ab_rounds.py "4,8" "8,8" "12,4" ...   (rounds,min_run per configuration; one engine of the test-hook
build per configuration: it reads ROCODER_ROUNDS / ROCODER_MIN_RUN once, at create)"""
import os, statistics, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd
from rocoder_amd import _lib
cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4, 8), (8, 8)]
dev = torch.device("cuda", 0)
NW = int(os.environ.get("ROCODER_AB_N", "16384"))
x = (torch.rand((2, 26_460_000 if NW == 16384 else 13_230_000), device=dev) - 0.5)
stream = torch.cuda.Stream(dev)
res = {c: [] for c in cfgs}
with torch.cuda.stream(stream):
    eng = {}
    with _lib.hooks_library():
        for c in cfgs:
            os.environ["ROCODER_ROUNDS"], os.environ["ROCODER_MIN_RUN"] = str(c[0]), str(c[1])
            eng[c] = rocoder_amd.Engine(window_len=NW, factor=8.0, channels=2, seed=1)
    e = eng[cfgs[0]]
    out = torch.empty((2, e.output_len(x.shape[1])), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        for _ in range(16):
            e.stretch_tensor(x, out=out)
        stream.synchronize()
    for rep in range(12):
        for c in cfgs:
            e = eng[c]
            for _ in range(12):
                e.stretch_tensor(x, out=out)
            stream.synchronize()
            res[c] += e.kernel_times(10)
for c in cfgs:
    v = res[c]
    print(f"rounds {c[0]:3d} min_run {c[1]:2d}: median {statistics.median(v):.4f} ms  mean {sum(v)/len(v):.4f}  min {min(v):.4f}  n={len(v)}")
