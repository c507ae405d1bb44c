"""dev: rate of the streaming seam (rc_engine_next_window per window) for a closed mono channel, through ctypes."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '.')
import rocoder_amd
from rocoder_amd import _lib
L = 44100 * 600
x = np.random.default_rng(1).uniform(-0.5, 0.5, L).astype(np.float32)
e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=1, seed=1)
lib = _lib.lib()
h = e._h if hasattr(e, "_h") else e.handle
t0 = time.perf_counter()
lib.rc_engine_push_input(h, 0, x.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), x.size)
lib.rc_engine_close_input(h, 0)
out = np.empty(16384, dtype=np.float32)
n = ctypes.c_size_t(0)
total = 0
po = out.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
while lib.rc_engine_is_done(h, 0) != 1:
    rc = lib.rc_engine_next_window(h, 0, po, out.size, ctypes.byref(n))
    assert rc == 0, rc
    total += n.value
dt = time.perf_counter() - t0
print(f"{total} samples in {dt*1e3:.1f} ms = {total/dt/1e9:.3f} Gsamples/s")
