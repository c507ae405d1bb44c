"""dev: PCIe-inclusive rate of rc_engine_stretch_host / rc_multi_stretch_host (host buffers in and out), C2 shape:
pinned rows (rc_host_alloc), pageable rows with a reused output array, pageable with a fresh output array per call."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import rocoder_amd
L = int(sys.argv[1]) if len(sys.argv) > 1 else 26_460_000
x = np.random.default_rng(0).uniform(-0.5, 0.5, (2, L)).astype(np.float32)


def report(tag, fn, n, reps=3):
    fn()
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        print(f"{tag} run {i}: {dt*1e3:.1f} ms  {n/dt/1e9:.2f} Gsamples/s  ({4*n/dt/1e9:.1f} GB/s out)", flush=True)


def run(e, tag):
    n_out = e.output_len(L)
    n = 2 * n_out
    xp = rocoder_amd.pinned_empty(x.shape)
    xp[:] = x
    yp = rocoder_amd.pinned_empty((2, n_out))
    report(tag + " pinned         ", lambda: e.stretch_host(xp, out=yp), n)
    yh = np.empty((2, n_out), np.float32)
    report(tag + " pageable reused", lambda: e.stretch_host(x, out=yh), n)
    report(tag + " pinned in, pageable out", lambda: e.stretch_host(xp, out=yh), n)
    keep = [e.stretch_host(x)]
    for i in range(2):  # a new output array per call, the previous one released before the clock starts (unmapping 1.7 GB costs tens of ms)
        keep[0] = None
        t0 = time.perf_counter()
        keep[0] = e.stretch_host(x)
        dt = time.perf_counter() - t0
        print(f"{tag} pageable fresh  run {i}: {dt*1e3:.1f} ms  {n/dt/1e9:.2f} Gsamples/s  ({4*n/dt/1e9:.1f} GB/s out)", flush=True)
    keep[0] = None
    return yp, yh


e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=2, seed=1)
yp, yh = run(e, "engine")
yd = e.stretch_tensor(__import__("torch").from_numpy(x).cuda()).cpu().numpy()
print("equal to device path:", np.array_equal(yp, yd), np.array_equal(yh, yd))
m = rocoder_amd.MultiEngine([0, 0], window_len=16384, factor=8.0, channels=2, seed=1)
yp2, yh2 = run(m, "multi{0,0}")
print("multi equal:", np.array_equal(yp2, yd), np.array_equal(yh2, yd))
