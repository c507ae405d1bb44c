"""BASELINE C4 through the HOST frequency-kernel callback (compiled C kernel, x2.0), L = 2 646 000 per channel, stereo,
window 16384, factor 8: host buffers in and out (what tools/bench_configs.py reports) and device-resident in / out
(the spectra still cross PCIe both ways), 1 and 2 kernel threads. Median of 5 after one warm-up."""
import json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd
src = "/tmp/k2.c"
open(src, "w").write("#include <stddef.h>\n#include <stdint.h>\nint apply(uint64_t t,const float*in,float*out,size_t n,void*u){for(size_t i=0;i<2*n;i++)out[i]=in[i]*2.0f;return 0;}\n")
os.system(f"cc -O3 -shared -fPIC -o /tmp/k2.so {src}")
k = rocoder_amd.load_kernel_library("/tmp/k2.so")
xh = np.random.default_rng(0).uniform(-0.5, 0.5, (2, 2_646_000)).astype(np.float32)
xd = torch.from_numpy(xh).cuda()
res = {}
for threads in (1, 2):
    e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1, kernel=k, kernel_threads=threads)
    e.stretch_host(xh[:, :300_000])
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        y = e.stretch_host(xh)
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    res[f"host_io_{threads}t"] = dict(ms=round(dt * 1e3, 1), out_gsamples_s=round(y.size / dt / 1e9, 3))
    out = torch.empty((2, e.output_len(xd.shape[1])), device="cuda")
    e.stretch_tensor(xd, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        e.stretch_tensor(xd, out=out)
        torch.cuda.synchronize()
        e.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = statistics.median(ts)
    res[f"device_io_{threads}t"] = dict(ms=round(dt * 1e3, 1), out_gsamples_s=round(out.numel() / dt / 1e9, 3))
    e.close()
print(json.dumps(res))
