"""Host-side mirror of the reference's interface for the hot path, above the C-ABI.

Names, argument meaning and error behaviour follow the reference (file:line = rocoder v0.4.0):
  * `Stretcher`          — src/stretcher.rs:12-136 (`new`, `next_window`, `is_done`, `channel_bound`)
  * `ReFFT`              — src/fft.rs:15-109 (`forward_fft`, `resynth`)
  * `StretcherProcessor` — src/stretcher_processor.rs:21-89 (one thread, round-robin channels,
                           bounded queues, Shutdown control message)
  * `AudioSpec`/`AudioBus` — src/audio.rs:31-37,141-224
All numerics run in the HIP engine (librocoder_hip.so); nothing here computes audio on the CPU.
"""
from __future__ import annotations

import ctypes as C
import functools
import queue
import threading
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import RC_EINVAL, RC_WOULD_BLOCK, RocoderError, check, rc_config, rc_params


@dataclass(frozen=True)
class AudioSpec:  # src/audio.rs:31-37
    channels: int = 2
    sample_rate: int = 44100


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _wrap_kernel(pyfunc):
    """Python callable (time_ms, complex64[N]) -> complex64[N] as the C-ABI rc_freq_kernel.
    Raising (or returning the wrong length) is the equivalent of a panic (src/fft.rs:100-106)."""
    if pyfunc is None:
        return C.cast(None, _lib.FREQ_KERNEL)
    if isinstance(pyfunc, _lib.FREQ_KERNEL):
        return pyfunc

    def tramp(time_ms, pin, pout, n, _user):
        try:
            a = np.ctypeslib.as_array(pin, shape=(2 * n,)).view(np.complex64)
            r = np.asarray(pyfunc(int(time_ms), a.copy()), dtype=np.complex64)
            if r.size != n:
                return 2
            np.ctypeslib.as_array(pout, shape=(2 * n,))[:] = r.view(np.float32)
            return 0
        except Exception:  # noqa: BLE001 - a panicking kernel must not poison the batch
            return 1

    return _lib.FREQ_KERNEL(tramp)


def load_kernel_library(path: str):
    """dlopen a shared object exporting the C symbol `apply` with the rc_freq_kernel signature
    (the C-ABI form of README.md:106-112; src/fft.rs:93-94 resolves the same name)."""
    so = C.CDLL(path)
    fn = so.apply
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t, C.c_void_p]
    k = C.cast(fn, _lib.FREQ_KERNEL)
    k._keepalive = so  # keep the library loaded, like the kernel stack at src/fft.rs:21
    return k


def make_config(window_len=16384, factor=1.0, amplitude=1.0, pitch_multiple=1, sample_rate=44100,
                channels=1, buffer_secs=1.0, seed=0, device=0, window=None, kernel=None,
                kernel_time_ms=0, max_batch_hops=0, kernel_threads=0, device_kernel=None):
    """device_kernel: None, ("gain", g), ("band", lo_bin, hi_bin, gain_inside, gain_outside) or
    ("shift", bins) - the curated on-GPU frequency kernels of include/rocoder_hip.h (RC_DK_*)."""
    cfg = rc_config()
    cfg.struct_size = C.sizeof(rc_config)
    cfg.window_len = int(window_len)
    cfg.factor = float(factor)
    cfg.amplitude = float(amplitude)
    cfg.pitch_multiple = int(pitch_multiple)
    cfg.sample_rate = int(sample_rate)
    cfg.channels = int(channels)
    cfg.buffer_secs = float(buffer_secs)
    cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    cfg.device = int(device)
    cfg.max_batch_hops = int(max_batch_hops)
    keep = []
    if window is not None:
        w = np.ascontiguousarray(window, dtype=np.float32)
        if w.size != window_len:
            raise ValueError("window length mismatch")
        cfg.window = _fp(w)
        keep.append(w)
    k = _wrap_kernel(kernel)
    cfg.kernel = k
    keep.append(k)
    cfg.kernel_time_ms = int(kernel_time_ms)
    cfg.kernel_threads = int(kernel_threads)
    if device_kernel is not None:
        kind = device_kernel[0]
        if kind == "gain":
            cfg.device_kernel, cfg.dk_gain = _lib.RC_DK_GAIN, float(device_kernel[1])
        elif kind == "band":
            cfg.device_kernel = _lib.RC_DK_BAND
            cfg.dk_lo_bin, cfg.dk_hi_bin = int(device_kernel[1]), int(device_kernel[2])
            cfg.dk_gain, cfg.dk_gain_outside = float(device_kernel[3]), float(device_kernel[4])
        elif kind == "shift":
            cfg.device_kernel, cfg.dk_shift_bins = _lib.RC_DK_SHIFT, int(device_kernel[1])
        else:
            raise ValueError(f"unknown device kernel {kind!r}")
    return cfg, keep


def derive_params(**kw) -> rc_params:
    """Stretcher::new parameter derivation (src/stretcher.rs:40-56) — host only, no device."""
    cfg, _keep = make_config(**kw)
    out = rc_params()
    check(_lib.lib().rc_derive_params(C.byref(cfg), C.byref(out)))
    return out


def offline_output_len(in_len: int, **kw) -> int:
    cfg, _keep = make_config(**kw)
    return int(_lib.lib().rc_offline_output_len(C.byref(cfg), in_len))


class _ViewOwner:
    """`base` of the arrays next_window_view hands out: keeps the engine alive and knows whether it is still current."""

    def __init__(self, engine, channel: int, ptr: int, nbytes: int):
        self.engine, self.channel, self.current = engine, channel, True
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, True), "version": 3}


class _PinnedBlock:
    """Owner of one rc_host_alloc block; numpy arrays made over it keep it alive through their `base` chain."""

    def __init__(self, nbytes: int):
        self._L = _lib.lib()
        p = C.c_void_p()
        check(self._L.rc_host_alloc(nbytes, C.byref(p)), self._L)
        self.ptr, self.nbytes = p.value, nbytes
        self.__array_interface__ = {"shape": (max(nbytes, 1),), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            self._L.rc_host_free(C.c_void_p(p))


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
    """numpy array in page-locked host memory (rc_host_alloc): rows handed to `Engine.stretch_host` /
    `MultiEngine.stretch_host` from here cross PCIe by DMA without a staging copy. The block is freed when the last
    array (or view) over it is garbage-collected."""
    shape = tuple(int(v) for v in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    n = int(np.prod(shape))
    blk = _PinnedBlock(n * dt.itemsize)
    return np.asarray(blk)[:n * dt.itemsize].view(dt).reshape(shape)


class Engine:
    """Thin RAII wrapper of rc_engine (all channels of one job)."""

    def __init__(self, **kw):
        self._L = _lib.lib()  # (the test-hook build inside `_lib.hooks_library()`)
        self._check = functools.partial(check, L=self._L)
        self._cfg, self._keep = make_config(**kw)
        h = C.c_void_p()
        self._check(self._L.rc_engine_create(C.byref(self._cfg), C.byref(h)))
        self._h = h
        self.params = rc_params()
        self._check(self._L.rc_engine_get_params(self._h, C.byref(self.params)))
        self.channels = int(self._cfg.channels)
        self.window_len = int(self._cfg.window_len)

    def close(self):
        for o in list((getattr(self, "_views", None) or {}).values()):
            o.current = False
        if getattr(self, "_h", None):
            self._L.rc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- streaming seam
    def push_input(self, channel: int, samples):
        s = np.ascontiguousarray(samples, dtype=np.float32)
        self._check(self._L.rc_engine_push_input(self._h, channel, _fp(s), s.size))

    def close_input(self, channel: int):
        self._check(self._L.rc_engine_close_input(self._h, channel))

    def next_window(self, channel: int) -> Optional[np.ndarray]:
        """One Stretcher::next_window; None when the reference would block on recv()."""
        out = np.empty(self.params.window_out_len, np.float32)
        n = C.c_size_t(0)
        rc = self._check(self._L.rc_engine_next_window(self._h, channel, _fp(out), out.size, C.byref(n)))
        if rc == RC_WOULD_BLOCK:
            return None
        return out[: n.value]

    def next_window_view(self, channel: int) -> Optional[np.ndarray]:
        """The same hand-out without the copy (rc_engine_next_window_view): a read-only array over the engine's
        pinned block. None when the reference would block.

        LIFETIME: the array aliases memory the engine owns. It is valid only until the NEXT hand-out of this channel
        (next_window / next_window_view) or close(): after that the block may be refilled with other windows or freed,
        and reading a retained array is undefined (stale data or a fault) - numpy cannot be told. Consume it (write it
        to the sink, src/main.rs:197-203) or copy it before asking for the next window; use next_window() for an array
        you may keep. The array keeps the Engine object itself alive (its `base` chain refers to it), so garbage
        collection never closes an engine under a live view; `view_is_current(a)` tells whether `a` is still the
        latest hand-out of its channel."""
        p = C.POINTER(C.c_float)()
        n = C.c_size_t(0)
        rc = self._check(self._L.rc_engine_next_window_view(self._h, channel, C.byref(p), C.byref(n)))
        if rc == RC_WOULD_BLOCK:
            return None
        owner = _ViewOwner(self, channel, C.addressof(p.contents), n.value * 4)
        # weak: the owner refers to the engine (an array keeps its engine alive), the engine must not refer back -
        # a cycle through an object with __del__ would leave the engine, its HBM and its pinned blocks to a cyclic GC
        # pass instead of the refcount
        if getattr(self, "_views", None) is None:
            import weakref

            self._views = weakref.WeakValueDictionary()
        prev = self._views.get(channel)
        if prev is not None:
            prev.current = False
        self._views[channel] = owner
        a = np.asarray(owner).view(np.float32)
        a.flags.writeable = False
        return a

    def view_is_current(self, a: np.ndarray) -> bool:
        """True while `a` (from next_window_view) is the latest hand-out of its channel on an open engine."""
        o = a
        while o is not None and not isinstance(o, _ViewOwner):
            o = getattr(o, "base", None)
        return bool(o is not None and o.current and o.engine is self and self._h)

    def is_done(self, channel: int) -> bool:
        return bool(self._check(self._L.rc_engine_is_done(self._h, channel)))

    def channel_bound(self) -> int:
        return int(self._L.rc_engine_channel_bound(self._h))

    # ---- offline
    def output_len(self, in_len: int) -> int:
        return int(self._L.rc_offline_output_len(C.byref(self._cfg), in_len))

    def stretch_host(self, channels_in, out: Optional[np.ndarray] = None) -> np.ndarray:
        """Host arrays in, host arrays out (rc_engine_stretch_host). `out` (float32 [channels, >= output_len], rows
        contiguous) is filled and returned when given - reuse it across calls, and allocate both sides with
        `pinned_empty` to let the DMA engines read / write them directly (no staging copy)."""
        x = np.atleast_2d(channels_in)
        if x.dtype != np.float32 or x.strides[-1] != 4:
            x = np.ascontiguousarray(x, dtype=np.float32)
        if x.shape[0] != self.channels:
            raise ValueError("channel count mismatch")
        n_out = self.output_len(x.shape[1])
        if out is None:
            out = np.empty((self.channels, n_out), np.float32)
        elif (out.dtype != np.float32 or out.ndim != 2 or out.shape[0] != self.channels or out.shape[1] < n_out
              or out.strides[1] != 4):
            raise ValueError(f"out must be float32 [{self.channels}, >= {n_out}] with contiguous rows")
        fp = C.POINTER(C.c_float)
        ins = (fp * self.channels)(*[_fp(x[c]) for c in range(self.channels)])
        outs = (fp * self.channels)(*[_fp(out[c]) for c in range(self.channels)])
        got = C.c_size_t(0)
        self._check(self._L.rc_engine_stretch_host(self._h, ins, x.shape[1], outs, out.shape[1], C.byref(got)))
        assert got.value == n_out
        return out[:, :n_out]

    def stretch_device_ptr(self, d_in: int, in_stride: int, in_len: int, d_out: int, out_stride: int,
                           out_cap: int, stream: int = 0) -> int:
        got = C.c_size_t(0)
        self._check(self._L.rc_engine_stretch_device(self._h, C.c_void_p(d_in), in_stride, in_len,
                                               C.c_void_p(d_out), out_stride, out_cap, C.byref(got),
                                               C.c_void_p(stream)))
        return got.value

    def stretch_device_range_ptr(self, d_in: int, in_stride: int, in_len: int, ch_first: int,
                                 ch_count: int, win_first: int, win_count: int, d_out: int,
                                 out_stride: int, out_cap: int, stream: int = 0):
        self._check(self._L.rc_engine_stretch_device_range(self._h, C.c_void_p(d_in), in_stride, in_len,
                                                     ch_first, ch_count, win_first, win_count,
                                                     C.c_void_p(d_out), out_stride, out_cap,
                                                     C.c_void_p(stream)))

    def stretch_tensor(self, x, out=None, stream=None):
        """x: torch float32 CUDA tensor [channels, L] (device-resident). Returns [channels, n_out].
        Asynchronous on the caller's stream: a device-side failure (a run-seam wait that expired leaves samples
        unwritten) is raised by the NEXT call on this engine - call `synchronize()` once the stream is synced
        before trusting `out` when no further call follows."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        n_out = self.output_len(x.shape[1])
        if out is None:
            out = torch.empty((self.channels, n_out), dtype=torch.float32, device=x.device)
        s = torch.cuda.current_stream(x.device).cuda_stream if stream is None else stream
        if s == 0:
            # legacy default stream: handle 0 means "the engine's own stream" in the C-ABI, so
            # order the call by hand (inputs ready before, outputs ready after)
            torch.cuda.current_stream(x.device).synchronize()
        self.stretch_device_ptr(x.data_ptr(), x.stride(0), x.shape[1], out.data_ptr(), out.stride(0),
                                out.shape[1], s)
        if s == 0:
            self.synchronize()
        return out

    def synchronize(self):
        self._check(self._L.rc_engine_synchronize(self._h))

    def last_kernel_stats(self):
        ms, hops, launches = C.c_float(0), C.c_uint64(0), C.c_uint32(0)
        self._check(self._L.rc_engine_last_kernel_stats(self._h, C.byref(ms), C.byref(hops),
                                                  C.byref(launches)))
        return float(ms.value), int(hops.value), int(launches.value)

    def kernel_times(self, n: int = 64) -> List[float]:
        """Event times (ms) of the kernel launches of the last min(n, 64) offline calls, oldest first."""
        buf = (C.c_float * n)()
        got = C.c_size_t(0)
        self._check(self._L.rc_engine_kernel_times(self._h, buf, n, C.byref(got)))
        return [float(buf[i]) for i in range(got.value)]

    # ---- single hop (ReFFT seam)
    def forward_fft(self, samples) -> np.ndarray:
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.size == self.window_len
        out = np.empty(2 * self.window_len, np.float32)
        self._check(self._L.rc_engine_forward_fft(self._h, _fp(s), _fp(out)))
        return out.view(np.complex64)

    def resynth(self, channel: int, hop: int, samples) -> np.ndarray:
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.size == self.window_len
        out = np.empty(self.window_len, np.float32)
        self._check(self._L.rc_engine_resynth(self._h, channel, hop, _fp(s), _fp(out)))
        return out



class MultiEngine:
    """One process, several GPUs of one node (include/rocoder_hip.h, rc_multi): one engine + one host thread per
    listed device, the job cut by `rc_shard_plan`, every shard copied once into its place. The multi-PROCESS layer
    (one rank per GPU over torch.distributed / RCCL) is `rocoder_amd.distributed`; both cut a job identically."""

    def __init__(self, device_ids: Sequence[int], **kw):
        self._L = _lib.lib()
        self._check = functools.partial(check, L=self._L)
        self._cfg, self._keep = make_config(**kw)
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        self._check(self._L.rc_multi_create(C.byref(self._cfg), ids, len(device_ids), C.byref(h)))
        self._h = h
        self.device_ids = [int(d) for d in device_ids]
        self.channels = int(self._cfg.channels)
        self.params = derive_params(**kw)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rc_multi_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def output_len(self, in_len: int) -> int:
        return int(self._L.rc_offline_output_len(C.byref(self._cfg), in_len))

    def stretch_host(self, x: np.ndarray, out: Optional[np.ndarray] = None) -> np.ndarray:
        x = np.atleast_2d(x)
        if x.dtype != np.float32 or x.strides[-1] != 4:
            x = np.ascontiguousarray(x, dtype=np.float32)
        assert x.shape[0] == self.channels
        n_out = self.output_len(x.shape[1])
        if out is None:
            out = np.empty((self.channels, n_out), np.float32)
        elif (out.dtype != np.float32 or out.ndim != 2 or out.shape[0] != self.channels or out.shape[1] < n_out
              or out.strides[1] != 4):
            raise ValueError(f"out must be float32 [{self.channels}, >= {n_out}] with contiguous rows")
        fp = C.POINTER(C.c_float)
        ins = (fp * self.channels)(*[r.ctypes.data_as(fp) for r in x])
        outs = (fp * self.channels)(*[r.ctypes.data_as(fp) for r in out])
        got = C.c_size_t(0)
        self._check(self._L.rc_multi_stretch_host(self._h, ins, x.shape[1], outs, out.shape[1], C.byref(got)))
        assert got.value == n_out
        return out[:, :n_out]

    def set_staging(self, force: bool):
        """Diagnostic (rc_multi_set_staging): shares on the root's own device take the copy path of a remote one."""
        self._check(self._L.rc_multi_set_staging(self._h, 1 if force else 0))

    def stretch_tensor(self, x, out=None, root: int = 0):
        """x, out: torch float32 CUDA tensors on the ROOT device of the list (device_ids[root]). Blocking."""
        import torch

        if not 0 <= root < len(self.device_ids):
            raise RocoderError(RC_EINVAL, f"root {root} is not an index of the device list {self.device_ids}")
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
        assert x.shape[0] == self.channels, (x.shape, self.channels)
        assert x.device.index == self.device_ids[root], \
            f"x lives on cuda:{x.device.index}, the root of the list is cuda:{self.device_ids[root]}"
        n_out = self.output_len(x.shape[1])
        if out is None:
            out = torch.empty((self.channels, n_out), dtype=torch.float32, device=x.device)
        assert out.device == x.device and out.dtype == torch.float32 and out.dim() == 2
        assert out.shape[0] == self.channels and out.shape[1] >= n_out and out.stride(1) == 1, (out.shape, out.stride())
        got = C.c_size_t(0)
        s = torch.cuda.current_stream(x.device)
        s.synchronize()
        self._check(self._L.rc_multi_stretch_device(self._h, root, C.c_void_p(x.data_ptr()), x.stride(0), x.shape[1],
                                                   C.c_void_p(out.data_ptr()), out.stride(0), out.shape[1],
                                                   C.byref(got), None))
        return out


class ReFFT:
    """src/fft.rs ReFFT: `new(window, kernel_src)`, `forward_fft`, `resynth`. The unseedable
    thread_rng of src/fft.rs:64 is replaced by the (seed, channel, hop) phase key."""

    def __init__(self, window, kernel=None, seed=0, channel_index=0, device=0, kernel_time_ms=0):
        w = np.ascontiguousarray(window, dtype=np.float32)
        self.window_len = w.size
        self.channel_index = channel_index
        self.hop = 0
        self._e = Engine(window_len=w.size, window=w, channels=channel_index + 1, seed=seed,
                         device=device, kernel=kernel, kernel_time_ms=kernel_time_ms)

    def forward_fft(self, samples) -> np.ndarray:  # src/fft.rs:50-61
        return self._e.forward_fft(samples)

    def resynth(self, samples, hop: Optional[int] = None) -> np.ndarray:  # src/fft.rs:42-48
        k = self.hop if hop is None else hop
        out = self._e.resynth(self.channel_index, k, samples)
        if hop is None:
            self.hop += 1
        return out


class Stretcher:
    """src/stretcher.rs Stretcher — one channel. `input` is the Receiver<Vec<f32>> side: a
    `queue.Queue` carrying float arrays, `None` meaning the Sender was dropped."""

    def __init__(self, spec: AudioSpec, input: "queue.Queue", factor: float, amplitude: float,  # noqa: A002
                 pitch_multiple: int, window, buffer_dur: float = 1.0, frequency_kernel=None,
                 seed: int = 0, channel_index: int = 0, device: int = 0, _engine: Engine = None,
                 kernel_time_ms: int = 0):
        self.spec = spec
        self.input = input
        self.channel_index = channel_index
        w = np.ascontiguousarray(window, dtype=np.float32)
        self._e = _engine or Engine(
            window_len=w.size, window=w, factor=factor, amplitude=amplitude,
            pitch_multiple=pitch_multiple, sample_rate=spec.sample_rate,
            channels=max(spec.channels, channel_index + 1), buffer_secs=buffer_dur, seed=seed,
            device=device, kernel=frequency_kernel, kernel_time_ms=kernel_time_ms)
        self.window_len = w.size
        self._closed = False

    def is_done(self) -> bool:  # src/stretcher.rs:78-80
        return self._e.is_done(self.channel_index)

    def channel_bound(self) -> int:  # src/stretcher.rs:82-85
        return self._e.channel_bound()

    def _pump(self, block: bool) -> bool:
        """Move chunks from the input queue into the engine; True if anything arrived."""
        got = False
        while not self._closed:
            try:
                chunk = self.input.get(block=block and not got)
            except queue.Empty:
                break
            got = True
            if chunk is None:
                self._e.close_input(self.channel_index)
                self._closed = True
            else:
                self._e.push_input(self.channel_index, chunk)
        return got

    def next_window(self) -> np.ndarray:  # src/stretcher.rs:87-121
        self._pump(block=False)
        while True:
            w = self._e.next_window(self.channel_index)
            if w is not None:
                return w
            # the reference blocks in self.input.recv() (src/stretcher.rs:125)
            self._pump(block=True)


class AudioBus:  # src/audio.rs:141-224
    def __init__(self, spec: AudioSpec, channels: List["queue.Queue"],
                 expected_total_samples: Optional[int]):
        self.spec = spec
        self.channels = channels
        self.expected_total_samples = expected_total_samples

    def into_audio(self, timeout: float = 0.005) -> List[np.ndarray]:  # src/audio.rs:152-172
        out: List[List[np.ndarray]] = [[] for _ in self.channels]
        closed = [False] * len(self.channels)
        while not all(closed):
            for i, ch in enumerate(self.channels):
                if closed[i]:
                    continue
                try:
                    chunk = ch.get(timeout=timeout)
                except queue.Empty:
                    continue
                if chunk is None:
                    closed[i] = True
                else:
                    out[i].append(chunk)
        return [np.concatenate(c) if c else np.zeros(0, np.float32) for c in out]


class StretcherProcessor:
    """src/stretcher_processor.rs: ONE thread, `loop { ctrl; for ch { if done break 'outer;
    send(next_window()) } }`, bounded(channel_bound()) output queues (back-pressure)."""

    SHUTDOWN = "Shutdown"  # StretcherProcessorControlMessage::Shutdown (:11-19)

    def __init__(self, channel_stretchers: Sequence[Stretcher],
                 expected_total_samples: Optional[int] = None):
        self.spec = channel_stretchers[0].spec
        self._channels = []
        receivers = []
        for s in channel_stretchers:  # :33-37
            q: "queue.Queue" = queue.Queue(maxsize=max(1, s.channel_bound()))
            self._channels.append((q, s))
            receivers.append(q)
        self.bus = AudioBus(self.spec, receivers, expected_total_samples)
        self._ctrl: "queue.Queue" = queue.Queue()
        self._finished = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.error: Optional[BaseException] = None

    @classmethod
    def new(cls, channel_stretchers, expected_total_samples=None):  # :26-46
        p = cls(channel_stretchers, expected_total_samples)
        return p, p.bus

    def _handle_control_messages(self) -> bool:  # :77-88 (non-blocking try_recv)
        try:
            msg = self._ctrl.get_nowait()
        except queue.Empty:
            return False
        return msg == self.SHUTDOWN

    def _run(self):
        try:
            running = True
            while running:  # :56-71
                if self._handle_control_messages():
                    break
                for out, stretcher in self._channels:
                    if stretcher.is_done():  # :64-68 "assuming each stretcher finishes at the same time"
                        running = False
                        break
                    out.put(stretcher.next_window())  # :69 blocks when the bounded queue is full
        except BaseException as e:  # noqa: BLE001
            self.error = e
        finally:
            for out, _ in self._channels:  # Senders dropped when the thread ends
                out.put(None)
            self._finished.set()  # :72

    def start(self):  # :50-75
        self._thread = threading.Thread(target=self._run, name="stretcher-processor", daemon=True)
        self._thread.start()
        return self

    def shutdown(self):  # Node::shutdown, src/signal_flow/node.rs:45-48
        self._ctrl.put(self.SHUTDOWN)

    def join(self, timeout=None):  # Node::join, src/signal_flow/node.rs:50-52
        if self._thread:
            self._thread.join(timeout)
        if self.error:
            raise self.error

    def is_finished(self) -> bool:  # src/signal_flow/node.rs:54-56
        return self._finished.is_set()


def stretch(channels_in, window_len=16384, factor=1.0, amplitude=1.0, pitch_multiple=1, seed=0,
            sample_rate=44100, kernel=None, device=0, kernel_time_ms=0, kernel_threads=0,
            device_kernel=None) -> np.ndarray:
    """Offline `-o` run (src/main.rs:124-160 minus file I/O): host [C, L] -> host [C, n_out]."""
    x = np.ascontiguousarray(np.atleast_2d(channels_in), dtype=np.float32)
    with Engine(window_len=window_len, factor=factor, amplitude=amplitude,
                pitch_multiple=pitch_multiple, sample_rate=sample_rate, channels=x.shape[0],
                seed=seed, device=device, kernel=kernel, kernel_time_ms=kernel_time_ms,
                kernel_threads=kernel_threads, device_kernel=device_kernel) as e:
        return e.stretch_host(x)


__all__ = ["AudioSpec", "AudioBus", "Engine", "ReFFT", "Stretcher", "StretcherProcessor", "stretch",
           "derive_params", "offline_output_len", "load_kernel_library", "RocoderError"]
