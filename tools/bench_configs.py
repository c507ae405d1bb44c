"""Timings of the other BASELINE.json configs (parity-test cases, not the bench line): C3 (pitch 3),
C4 (x2.0 apply() kernel through the C-ABI, host-callback bound), C5 (8 ch, window 65536, factor 32 —
all channels on this one GPU), and the PCIe-inclusive host-buffer rate of C2."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd  # noqa: E402


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


res = {}
dev = torch.device("cuda", 0)
x2 = (torch.rand((2, 26_460_000), device=dev) - 0.5)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    for name, kw in [("C2", dict(factor=8.0)), ("C3", dict(factor=8.0, pitch_multiple=3))]:
        e = rocoder_amd.Engine(window_len=16384, channels=2, seed=1, **kw)
        out = torch.empty((2, e.output_len(x2.shape[1])), device=dev)
        dt = timed(lambda: e.stretch_tensor(x2, out=out), 5)
        res[name] = dict(ms=round(dt * 1e3, 3), out_msamples_s=round(out.numel() / dt / 1e6, 1))
        e.close()
    x8 = (torch.rand((8, 5_292_000), device=dev) - 0.5)
    e = rocoder_amd.Engine(window_len=65536, factor=32.0, channels=8, seed=1)
    out = torch.empty((8, e.output_len(x8.shape[1])), device=dev)
    dt = timed(lambda: e.stretch_tensor(x8, out=out), 2)
    res["C5_one_gpu"] = dict(ms=round(dt * 1e3, 2), out_msamples_s=round(out.numel() / dt / 1e6, 1))
    e.close()
    del out, x8
# C4: compiled C kernel (x2.0), shorter input (host-callback bound)
src = "/tmp/k2.c"
open(src, "w").write("#include <stddef.h>\n#include <stdint.h>\nint apply(uint64_t t,const float*in,float*out,size_t n,void*u){for(size_t i=0;i<2*n;i++)out[i]=in[i]*2.0f;return 0;}\n")
os.system(f"cc -O3 -shared -fPIC -o /tmp/k2.so {src}")
k = rocoder_amd.load_kernel_library("/tmp/k2.so")
xh = np.random.default_rng(0).uniform(-0.5, 0.5, (2, 2_646_000)).astype(np.float32)
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1, kernel=k)
t0 = time.perf_counter()
y = e.stretch_host(xh)
dt = time.perf_counter() - t0
res["C4_x2_kernel_host_callback"] = dict(ms=round(dt * 1e3, 1), out_msamples_s=round(y.size / dt / 1e6, 1),
                                         note="L=2646000/ch; includes PCIe + per-hop apply() on one host thread")
e.close()
# PCIe-inclusive C2 (host buffers in and out)
xh = x2.cpu().numpy()
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1)
e.stretch_host(xh[:, :1_000_000])
t0 = time.perf_counter()
y = e.stretch_host(xh)
dt = time.perf_counter() - t0
res["C2_pcie_inclusive_host_buffers"] = dict(ms=round(dt * 1e3, 1), out_msamples_s=round(y.size / dt / 1e6, 1))
print(json.dumps(res, indent=1))
