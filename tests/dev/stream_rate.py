"""dev: rate of the streaming seam for a closed mono channel, through ctypes: rc_engine_next_window (a copy per
window into the caller's buffer) and rc_engine_next_window_view (a pointer into the engine's pinned block), the input
push timed separately."""
import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, '.')
import rocoder_amd
from rocoder_amd import _lib
L = 44100 * 600
x = np.random.default_rng(1).uniform(-0.5, 0.5, L).astype(np.float32)
lib = _lib.lib()
fp = ctypes.POINTER(ctypes.c_float)
res = {}
for mode in ("copy", "view", "copy", "view"):
    e = rocoder_amd.Engine(window_len=16384, factor=8.0, channels=1, seed=1)
    h = e._h
    t0 = time.perf_counter()
    lib.rc_engine_push_input(h, 0, x.ctypes.data_as(fp), x.size)
    lib.rc_engine_close_input(h, 0)
    t1 = time.perf_counter()
    out = np.empty(16384, dtype=np.float32)
    po = out.ctypes.data_as(fp)
    pv = fp()
    n = ctypes.c_size_t(0)
    total = 0
    acc = 0.0
    while lib.rc_engine_is_done(h, 0) != 1:
        if mode == "copy":
            rc = lib.rc_engine_next_window(h, 0, po, out.size, ctypes.byref(n))
        else:
            rc = lib.rc_engine_next_window_view(h, 0, ctypes.byref(pv), ctypes.byref(n))
            acc += pv[0]  # touch the window
        assert rc == 0, rc
        total += n.value
    t2 = time.perf_counter()
    res[mode] = dict(samples=total, push_ms=round((t1 - t0) * 1e3, 1), windows_ms=round((t2 - t1) * 1e3, 1),
                     gsamples_s_windows=round(total / (t2 - t1) / 1e9, 3), gsamples_s_with_push=round(total / (t2 - t0) / 1e9, 3))
    e.close()
print(json.dumps(res))
