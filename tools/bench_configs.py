"""Timings of the other BASELINE.json configs (parity-test cases, not the bench line), one GPU, pre-heated,
median of the per-launch event times the engine records around its kernels:
  C3  pitch 3                                  (fused hop kernel, decimating store)
  C4  x2.0 apply() kernel at N = 16384: host callback on 1 and on 2 host threads (PCIe + host bound),
      and the same kernel as the curated on-GPU gain (rc_config::device_kernel: no host round trip)
  C5  8 ch, window 65536, factor 32, all channels on this GPU (fused big4_kernel) and the window-32768 twin
  streaming: Stretcher::next_window loop through the C-ABI (one channel per call, H2D + kernel + D2H per batch)
"""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rocoder_amd  # noqa: E402
from rocoder_amd import _lib  # noqa: E402
from boxclock import ClockSampler  # noqa: E402


def device_job(x, n=10, heat_s=1.0, **kw):
    e = rocoder_amd.Engine(channels=x.shape[0], seed=1, **kw)
    out = torch.empty((x.shape[0], e.output_len(x.shape[1])), device=x.device)
    e.stretch_tensor(x, out=out)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < heat_s:
        for _ in range(4):
            e.stretch_tensor(x, out=out)
        torch.cuda.current_stream().synchronize()
    with ClockSampler(x.device.index or 0) as clk:
        for _ in range(max(n, int(0.3 / max(1e-4, 1e-3 * statistics.median(e.kernel_times(4)))))):
            e.stretch_tensor(x, out=out)
        torch.cuda.current_stream().synchronize()
    ms = statistics.median(e.kernel_times(n))
    _, hops, launches = e.last_kernel_stats()
    N = kw["window_len"]
    r = dict(kernel_ms=round(ms, 3), hops=hops, launches=launches, out_msamples_s=round(out.numel() / ms / 1e3, 1),
             frac_hbm_read_roofline=round(hops * 4.0 * N / ms / 1e6 / 8000.0, 4), sclk_mhz_under_load=clk.median_mhz())
    e.close()
    return r


res = {"kernel_id": _lib.lib().rc_kernel_id().decode()}
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
with torch.cuda.stream(stream):
    x2 = (torch.rand((2, 26_460_000), device=dev) - 0.5)
    res["C2"] = device_job(x2, window_len=16384, factor=8.0)
    res["C3_pitch3"] = device_job(x2, window_len=16384, factor=8.0, pitch_multiple=3)
    res["C4_device_gain_x2"] = device_job(x2, window_len=16384, factor=8.0, device_kernel=("gain", 2.0))
    xs = x2[:, :2_646_000].contiguous()
    res["band_mask_on_gpu_L2646000"] = device_job(xs, n=5, heat_s=0.3, window_len=16384, factor=8.0,
                                                  device_kernel=("band", 64, 2048, 1.0, 0.05))
    x8 = (torch.rand((8, 5_292_000), device=dev) - 0.5)
    res["C5_one_gpu"] = device_job(x8, window_len=65536, factor=32.0)
    res["C5_window32768"] = device_job(x8, window_len=32768, factor=32.0)
    del x8
# C4: compiled C kernel (x2.0) through the host callback, L = 2 646 000 per channel
src = "/tmp/k2.c"
open(src, "w").write("#include <stddef.h>\n#include <stdint.h>\nint apply(uint64_t t,const float*in,float*out,size_t n,void*u){for(size_t i=0;i<2*n;i++)out[i]=in[i]*2.0f;return 0;}\n")
os.system(f"cc -O3 -shared -fPIC -o /tmp/k2.so {src}")
k = rocoder_amd.load_kernel_library("/tmp/k2.so")
xh = np.random.default_rng(0).uniform(-0.5, 0.5, (2, 2_646_000)).astype(np.float32)
for threads in (1, 2):
    e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1, kernel=k, kernel_threads=threads)
    e.stretch_host(xh[:, :300_000])
    t0 = time.perf_counter()
    y = e.stretch_host(xh)
    dt = time.perf_counter() - t0
    res[f"C4_host_callback_{threads}_thread"] = dict(ms=round(dt * 1e3, 1), out_msamples_s=round(y.size / dt / 1e6, 1),
                                                     note="L=2646000/ch; host buffers in AND out (a fresh 169 MB output array "
                                                          "per call) + the spectra over PCIe both ways + apply() per hop")
    # the same job with input and output resident in HBM (as for `value`): only the spectra cross PCIe
    xd = torch.from_numpy(xh).to(dev)
    outd = torch.empty((2, e.output_len(xd.shape[1])), device=dev)
    e.stretch_tensor(xd, out=outd)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        e.stretch_tensor(xd, out=outd)
        torch.cuda.synchronize()
        e.synchronize()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    res[f"C4_host_callback_{threads}_thread_device_io"] = dict(ms=round(dt * 1e3, 1), out_msamples_s=round(outd.numel() / dt / 1e6, 1),
                                                               note="device-resident input and output; median of 5")
    del xd, outd
    e.close()
# streaming seam: one channel, chunks of 1 s, windows pulled as they become computable
import queue  # noqa: E402

xs1 = np.random.default_rng(1).uniform(-0.5, 0.5, 44100 * 120).astype(np.float32)
q = queue.Queue()
i32 = np.arange(16384, dtype=np.float32)
w = (np.float32(0.5) - np.cos((i32 * np.float32(2 * np.pi)) / np.float32(16383), dtype=np.float32) * np.float32(0.5))
st = rocoder_amd.Stretcher(rocoder_amd.AudioSpec(1, 44100), q, 8.0, 1.0, 1, w, seed=1)
for i in range(0, xs1.size, 44100):
    q.put(xs1[i:i + 44100])
q.put(None)
t0 = time.perf_counter()
n = 0
while not st.is_done():
    n += st.next_window().size
dt = time.perf_counter() - t0
res["streaming_next_window_mono"] = dict(ms=round(dt * 1e3, 1), out_msamples_s=round(n / dt / 1e6, 1),
                                         note="120 s mono, window 16384, factor 8: rc_engine_next_window per window")
print(json.dumps(res, indent=1))
