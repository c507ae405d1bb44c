"""ctypes binding of oracle/librocoder_oracle.so (TEST INFRASTRUCTURE ONLY).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librocoder_oracle.so")

FREQ_KERNEL = C.CFUNCTYPE(C.c_int, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                          C.c_size_t, C.c_void_p)

RCO_OK, RCO_WOULD_BLOCK, RCO_EINVAL = 0, 1, -1


_SO_BASE = os.path.join(_HERE, "librocoder_cpubase.so")


def build(force: bool = False) -> str:
    """Compile the C oracle (and the measured CPU baseline) with gcc: building the checker is not
    using it."""
    src = os.path.join(_HERE, "rocoder_oracle.c")
    hdr = os.path.join(_HERE, "rocoder_oracle.h")
    stale = (not os.path.exists(_SO)
             or os.path.getmtime(_SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librocoder_oracle.so"],
                              stdout=subprocess.DEVNULL)
    bsrc = os.path.join(_HERE, "rocoder_cpu_baseline.c")
    if force or not os.path.exists(_SO_BASE) or os.path.getmtime(_SO_BASE) < os.path.getmtime(bsrc):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librocoder_cpubase.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_base = None


def cpu_baseline_stretch(channels_in, window_len=16384, factor=1.0, amplitude=1.0, pitch_multiple=1,
                         seed=0, threads=1, ch_first=0) -> np.ndarray:
    """oracle/rocoder_cpu_baseline.c: the reference's algorithm written for speed (optimised FFT,
    optional OpenMP over hop ranges) - what bench.py times as `cpu_baseline`. [C, L] -> [C, n_out].
    `ch_first`: the rows are channels ch_first.. of a larger job (the phase key takes the job's channel index)."""
    global _base
    if _base is None:
        build()
        _base = C.CDLL(_SO_BASE)
        _base.rcb_output_len.restype = C.c_size_t
        _base.rcb_output_len.argtypes = [C.c_size_t, C.c_uint32, C.c_float, C.c_int]
        _base.rcb_stretch_from.restype = C.c_int
        _base.rcb_stretch_from.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                           C.c_float, C.c_float, C.c_int, C.c_uint64, C.POINTER(C.c_float), C.c_int]
    x = np.ascontiguousarray(np.atleast_2d(channels_in), dtype=np.float32)
    nch, length = x.shape
    n_out = _base.rcb_output_len(length, window_len, factor, pitch_multiple)
    if n_out == 0:
        raise ValueError("unsupported parameters for the CPU baseline")
    out = np.zeros((nch, n_out), np.float32)
    rc = _base.rcb_stretch_from(_fp(x), length, nch, ch_first, window_len, factor, amplitude, pitch_multiple,
                                seed, _fp(out), threads)
    if rc != 0:
        raise ValueError(f"rcb_stretch rc={rc}")
    return out


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    fp = C.POINTER(C.c_float)
    sz = C.c_size_t
    L.rco_hanning.argtypes = [sz, fp]
    L.rco_rectangular.argtypes = [sz, fp]
    L.rco_inverse.argtypes = [fp, sz, fp]
    L.rco_hanning_crossfade_compensation.argtypes = [sz, fp]
    L.rco_lerp.argtypes = [C.c_float] * 3
    L.rco_lerp.restype = C.c_float
    L.rco_resample_len.argtypes = [sz, C.c_int]
    L.rco_resample_len.restype = sz
    L.rco_resample.argtypes = [fp, sz, C.c_int, fp]
    L.rco_resample.restype = sz
    L.rco_phase_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    L.rco_phase_key.restype = C.c_uint64
    L.rco_phase_hash.argtypes = [C.c_uint64, C.c_uint32]
    L.rco_phase_hash.restype = C.c_uint32
    L.rco_phase_theta.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    L.rco_phase_theta.restype = C.c_float
    L.rco_refft_new.argtypes = [fp, sz]
    L.rco_refft_new.restype = C.c_void_p
    L.rco_refft_free.argtypes = [C.c_void_p]
    L.rco_refft_forward.argtypes = [C.c_void_p, fp, sz, fp]
    L.rco_refft_resynth_from_spectrum.argtypes = [C.c_void_p, fp, C.c_uint64, fp]
    L.rco_refft_resynth.argtypes = [C.c_void_p, fp, sz, C.c_uint64, FREQ_KERNEL, C.c_void_p,
                                    C.c_uint64, fp]
    L.rco_stretcher_new.argtypes = [C.c_uint32, C.c_uint16, C.c_float, C.c_float, C.c_int, fp, sz,
                                    C.c_float, C.c_uint64, C.c_uint32, FREQ_KERNEL, C.c_void_p]
    L.rco_stretcher_new.restype = C.c_void_p
    L.rco_stretcher_free.argtypes = [C.c_void_p]
    L.rco_stretcher_send.argtypes = [C.c_void_p, fp, sz]
    L.rco_stretcher_close_input.argtypes = [C.c_void_p]
    L.rco_stretcher_is_done.argtypes = [C.c_void_p]
    L.rco_stretcher_is_done.restype = C.c_int
    L.rco_stretcher_channel_bound.argtypes = [C.c_void_p]
    L.rco_stretcher_channel_bound.restype = sz
    L.rco_stretcher_max_window_out.argtypes = [C.c_void_p]
    L.rco_stretcher_max_window_out.restype = sz
    L.rco_stretcher_ensure_input.argtypes = [C.c_void_p, sz]
    L.rco_stretcher_ensure_input.restype = C.c_int
    L.rco_stretcher_input_len.argtypes = [C.c_void_p]
    L.rco_stretcher_input_len.restype = sz
    L.rco_stretcher_input_ptr.argtypes = [C.c_void_p]
    L.rco_stretcher_input_ptr.restype = fp
    L.rco_stretcher_next_window.argtypes = [C.c_void_p, fp, C.POINTER(sz)]
    L.rco_stretcher_next_window.restype = C.c_int
    L.rco_stretcher_step.argtypes = [C.c_void_p]
    L.rco_stretcher_step.restype = sz
    L.rco_stretcher_samples_needed.argtypes = [C.c_void_p]
    L.rco_stretcher_samples_needed.restype = sz
    L.rco_stretcher_amp.argtypes = [C.c_void_p]
    L.rco_stretcher_amp.restype = C.c_float
    L.rco_stretcher_hops_done.argtypes = [C.c_void_p]
    L.rco_stretcher_hops_done.restype = C.c_uint64
    L.rco_stretcher_set_time_ms.argtypes = [C.c_void_p, C.c_uint64]
    L.rco_stretch_offline.argtypes = [C.c_uint16, C.POINTER(fp), sz, C.c_uint32, sz, C.c_float,
                                      C.c_float, C.c_int, C.c_uint64, FREQ_KERNEL, C.c_void_p,
                                      C.POINTER(fp), sz, C.POINTER(sz)]
    L.rco_stretch_offline.restype = C.c_int
    L.rco_offline_output_len.argtypes = [sz, sz, C.c_float, C.c_int]
    L.rco_offline_output_len.restype = sz
    _lib = L
    return L


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


NULL_KERNEL = C.cast(None, FREQ_KERNEL)


def hanning(n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().rco_hanning(n, _fp(out))
    return out


def rectangular(n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().rco_rectangular(n, _fp(out))
    return out


def inverse(a) -> np.ndarray:
    a = _f32(a)
    out = np.empty_like(a)
    lib().rco_inverse(_fp(a), a.size, _fp(out))
    return out


def hanning_crossfade_compensation(n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().rco_hanning_crossfade_compensation(n, _fp(out))
    return out


def lerp(a: float, b: float, r: float) -> float:
    return float(lib().rco_lerp(a, b, r))


def resample(samples, factor: int) -> np.ndarray:
    s = _f32(samples)
    n = lib().rco_resample_len(s.size, factor)
    if n == C.c_size_t(-1).value:
        raise ValueError("invalid resample factor")  # the reference panics (resampler.rs:11)
    out = np.empty(max(n, 1), np.float32)
    m = lib().rco_resample(_fp(s), s.size, factor, _fp(out))
    return out[:m].copy()


def phase_key(seed: int, channel: int, hop: int) -> int:
    return int(lib().rco_phase_key(seed, channel, hop))


def phase_hash(key: int, bins) -> np.ndarray:
    L = lib()
    return np.array([L.rco_phase_hash(key, int(b)) for b in np.atleast_1d(bins)], np.uint32)


def phase_theta(key: int, bins, n_bins: int) -> np.ndarray:
    L = lib()
    return np.array([L.rco_phase_theta(key, int(b), n_bins) for b in np.atleast_1d(bins)], np.float32)


def wrap_kernel(pyfunc):
    """pyfunc(time_ms, spec: complex64[N]) -> complex64[N] (or raise => 'panic')."""
    if pyfunc is None:
        return NULL_KERNEL

    def tramp(time_ms, pin, pout, n, _user):
        try:
            a = np.ctypeslib.as_array(pin, shape=(2 * n,)).view(np.complex64)
            r = np.asarray(pyfunc(int(time_ms), a.copy()), dtype=np.complex64)
            if r.size != n:
                return 2
            np.ctypeslib.as_array(pout, shape=(2 * n,))[:] = r.view(np.float32)
            return 0
        except Exception:
            return 1

    return FREQ_KERNEL(tramp)


class ReFFT:
    """src/fft.rs ReFFT, with the phase key passed explicitly."""

    def __init__(self, window):
        self.window = _f32(window)
        self.n = self.window.size
        self._h = lib().rco_refft_new(_fp(self.window), self.n)

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().rco_refft_free(self._h)
            self._h = None

    def forward_fft(self, samples) -> np.ndarray:
        s = _f32(samples)
        out = np.empty(2 * self.n, np.float32)
        lib().rco_refft_forward(self._h, _fp(s), s.size, _fp(out))
        return out.view(np.complex64)

    def resynth_from_fft_result(self, spec, key: int) -> np.ndarray:
        sp = np.ascontiguousarray(spec, dtype=np.complex64).view(np.float32)
        out = np.empty(self.n, np.float32)
        lib().rco_refft_resynth_from_spectrum(self._h, _fp(sp), key, _fp(out))
        return out

    def resynth(self, samples, key: int, kernel=None, time_ms: int = 0) -> np.ndarray:
        s = _f32(samples)
        out = np.empty(self.n, np.float32)
        k = wrap_kernel(kernel)
        lib().rco_refft_resynth(self._h, _fp(s), s.size, key, k, None, time_ms, _fp(out))
        return out


class Stretcher:
    """src/stretcher.rs Stretcher (one channel)."""

    def __init__(self, sample_rate=44100, channels=2, factor=1.0, amplitude=1.0, pitch_multiple=1,
                 window=None, buffer_secs=1.0, seed=0, channel_index=0, kernel=None):
        self.window = _f32(window)
        self._k = wrap_kernel(kernel)
        self._h = lib().rco_stretcher_new(sample_rate, channels, factor, amplitude, pitch_multiple,
                                          _fp(self.window), self.window.size, buffer_secs, seed,
                                          channel_index, self._k, None)
        if not self._h:
            raise ValueError("invalid stretcher parameters (reference would assert or hang)")

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:
            lib().rco_stretcher_free(self._h)
            self._h = None

    def send(self, chunk):
        c = _f32(chunk)
        lib().rco_stretcher_send(self._h, _fp(c), c.size)

    def close_input(self):
        lib().rco_stretcher_close_input(self._h)

    def is_done(self) -> bool:
        return bool(lib().rco_stretcher_is_done(self._h))

    def channel_bound(self) -> int:
        return int(lib().rco_stretcher_channel_bound(self._h))

    def ensure_input_samples_available(self, n: int) -> int:
        return int(lib().rco_stretcher_ensure_input(self._h, n))

    def input_buf(self) -> np.ndarray:
        n = lib().rco_stretcher_input_len(self._h)
        p = lib().rco_stretcher_input_ptr(self._h)
        return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.empty(0, np.float32)

    @property
    def step(self) -> int:
        return int(lib().rco_stretcher_step(self._h))

    @property
    def samples_needed_per_window(self) -> int:
        return int(lib().rco_stretcher_samples_needed(self._h))

    @property
    def amp(self) -> float:
        return float(lib().rco_stretcher_amp(self._h))

    @property
    def hops_done(self) -> int:
        return int(lib().rco_stretcher_hops_done(self._h))

    def set_time_ms(self, t: int):
        lib().rco_stretcher_set_time_ms(self._h, t)

    def next_window(self) -> np.ndarray:
        cap = lib().rco_stretcher_max_window_out(self._h)
        out = np.empty(max(cap, 1), np.float32)
        n = C.c_size_t(0)
        rc = lib().rco_stretcher_next_window(self._h, _fp(out), C.byref(n))
        if rc == RCO_WOULD_BLOCK:
            raise BlockingIOError("input channel empty and not closed")
        if rc != RCO_OK:
            raise ValueError(f"next_window failed rc={rc}")
        return out[: n.value].copy()


def offline_output_len(length: int, window_len: int, factor: float, pitch_multiple: int) -> int:
    return int(lib().rco_offline_output_len(length, window_len, factor, pitch_multiple))


def stretch_offline(channels_in, window_len=16384, factor=1.0, amplitude=1.0, pitch_multiple=1,
                    seed=0, sample_rate=44100, kernel=None) -> np.ndarray:
    """main.rs:131-155 + stretcher_processor.rs:56-71 for `-o` mode. channels_in: [C, L]."""
    x = np.ascontiguousarray(np.atleast_2d(channels_in), dtype=np.float32)
    nch, length = x.shape
    n_out = offline_output_len(length, window_len, factor, pitch_multiple)
    if n_out == 0:
        raise ValueError("invalid parameters")
    out = np.zeros((nch, n_out), np.float32)
    fp = C.POINTER(C.c_float)
    ins = (fp * nch)(*[_fp(x[c]) for c in range(nch)])
    outs = (fp * nch)(*[_fp(out[c]) for c in range(nch)])
    got = C.c_size_t(0)
    k = wrap_kernel(kernel)
    rc = lib().rco_stretch_offline(nch, ins, length, sample_rate, window_len, factor, amplitude,
                                   pitch_multiple, seed, k, None, outs, n_out, C.byref(got))
    if rc != RCO_OK:
        raise ValueError(f"rco_stretch_offline rc={rc}")
    assert got.value == n_out, (got.value, n_out)
    return out
