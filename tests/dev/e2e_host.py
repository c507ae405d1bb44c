"""dev: PCIe-inclusive rate of rc_engine_stretch_host (host buffers in and out), C2 shape."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import rocoder_amd
L = int(sys.argv[1]) if len(sys.argv) > 1 else 26_460_000
x = np.random.default_rng(0).uniform(-0.5, 0.5, (2, L)).astype(np.float32)
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1)
y = e.stretch_host(x[:, :1_000_000])
for i in range(3):
    y = None  # (freeing 1.7 GB inside the timed region would cost ~50 ms)
    t0 = time.perf_counter()
    y = e.stretch_host(x)
    dt = time.perf_counter() - t0
    print(f"run {i}: {dt*1e3:.1f} ms  {y.size/dt/1e9:.2f} Gsamples/s  ({y.nbytes/dt/1e9:.1f} GB/s out)", flush=True)
yd = e.stretch_tensor(__import__("torch").from_numpy(x).cuda()).cpu().numpy()
print("equal to device path:", np.array_equal(y, yd))
