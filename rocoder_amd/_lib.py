"""ctypes loader for librocoder_hip.so — the C-ABI declared in include/rocoder_hip.h.

The library is built in-tree by `rocoder_amd.build.build()` (hipcc, gfx950). There is no
Python/CPU fallback: if the library is missing, loading fails loudly.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ROCODER_HIP_LIB") or os.path.join(_HERE, "librocoder_hip.so")
# the same sources built with -DRC_TEST_HOOKS=1 (`make hooks`): reads ROCODER_DIAG and the run-planner tuning
# variables at rc_engine_create and contains the previous kernel generation. Tests and A/B tools only.
HOOKS_PATH = os.path.join(_HERE, "librocoder_hip_hooks.so")

RC_OK, RC_WOULD_BLOCK = 0, 1
RC_EINVAL, RC_ENODEVICE, RC_EUNSUPPORTED, RC_ENOMEM, RC_EHIP, RC_ECAPACITY = -1, -2, -3, -4, -5, -6

FREQ_KERNEL = C.CFUNCTYPE(C.c_int, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_float),
                          C.c_size_t, C.c_void_p)


class rc_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("window_len", C.c_uint32),
        ("factor", C.c_float),
        ("amplitude", C.c_float),
        ("pitch_multiple", C.c_int32),
        ("sample_rate", C.c_uint32),
        ("channels", C.c_uint32),
        ("buffer_secs", C.c_float),
        ("seed", C.c_uint64),
        ("device", C.c_int32),
        ("max_batch_hops", C.c_uint32),
        ("window", C.POINTER(C.c_float)),
        ("kernel", FREQ_KERNEL),
        ("kernel_user", C.c_void_p),
        ("kernel_time_ms", C.c_uint64),
        ("kernel_threads", C.c_uint32),
        ("device_kernel", C.c_uint32),
        ("dk_gain", C.c_float),
        ("dk_gain_outside", C.c_float),
        ("dk_lo_bin", C.c_uint32),
        ("dk_hi_bin", C.c_uint32),
        ("dk_shift_bins", C.c_int32),
    ]


RC_DK_NONE, RC_DK_GAIN, RC_DK_BAND, RC_DK_SHIFT = 0, 1, 2, 3


class rc_params(C.Structure):
    _fields_ = [
        ("window_len", C.c_uint32),
        ("half_window_len", C.c_uint32),
        ("samples_needed_per_window", C.c_uint64),
        ("sample_step_len", C.c_uint32),
        ("hops_per_window", C.c_uint32),
        ("window_out_len", C.c_uint32),
        ("corrected_amp_factor", C.c_float),
        ("pitch_shifted_factor", C.c_float),
    ]


class rc_shard(C.Structure):
    _fields_ = [
        ("device_index", C.c_uint32),
        ("ch_first", C.c_uint32),
        ("ch_count", C.c_uint32),
        ("win_first", C.c_uint64),
        ("win_count", C.c_uint64),
    ]


# every symbol include/rocoder_hip.h declares: name -> (restype, argtypes)
_fp = C.POINTER(C.c_float)
_sz = C.c_size_t
_eng = C.c_void_p
SYMBOLS = {
    "rc_last_error": (C.c_char_p, []),
    "rc_abi_version": (C.c_int, []),
    "rc_kernel_id": (C.c_char_p, []),
    "rc_device_count": (C.c_int, []),
    "rc_derive_params": (C.c_int, [C.POINTER(rc_config), C.POINTER(rc_params)]),
    "rc_offline_output_len": (_sz, [C.POINTER(rc_config), _sz]),
    "rc_phase_key": (C.c_uint64, [C.c_uint64, C.c_uint32, C.c_uint64]),
    "rc_phase_hash": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "rc_phase_theta": (C.c_float, [C.c_uint64, C.c_uint32, C.c_uint32]),
    "rc_engine_create": (C.c_int, [C.POINTER(rc_config), C.POINTER(_eng)]),
    "rc_engine_destroy": (None, [_eng]),
    "rc_engine_get_params": (C.c_int, [_eng, C.POINTER(rc_params)]),
    "rc_engine_push_input": (C.c_int, [_eng, C.c_uint32, _fp, _sz]),
    "rc_engine_close_input": (C.c_int, [_eng, C.c_uint32]),
    "rc_engine_next_window": (C.c_int, [_eng, C.c_uint32, _fp, _sz, C.POINTER(_sz)]),
    "rc_engine_next_window_view": (C.c_int, [_eng, C.c_uint32, C.POINTER(_fp), C.POINTER(_sz)]),
    "rc_engine_is_done": (C.c_int, [_eng, C.c_uint32]),
    "rc_engine_channel_bound": (_sz, [_eng]),
    "rc_engine_stretch_host": (C.c_int, [_eng, C.POINTER(_fp), _sz, C.POINTER(_fp), _sz,
                                         C.POINTER(_sz)]),
    "rc_host_alloc": (C.c_int, [_sz, C.POINTER(C.c_void_p)]),
    "rc_host_free": (C.c_int, [C.c_void_p]),
    "rc_engine_stretch_device": (C.c_int, [_eng, C.c_void_p, _sz, _sz, C.c_void_p, _sz, _sz,
                                           C.POINTER(_sz), C.c_void_p]),
    "rc_engine_stretch_device_range": (C.c_int, [_eng, C.c_void_p, _sz, _sz, C.c_uint32, C.c_uint32,
                                                 C.c_uint64, C.c_uint64, C.c_void_p, _sz, _sz,
                                                 C.c_void_p]),
    "rc_engine_synchronize": (C.c_int, [_eng]),
    "rc_engine_last_kernel_stats": (C.c_int, [_eng, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint32)]),
    "rc_engine_kernel_times": (C.c_int, [_eng, _fp, _sz, C.POINTER(_sz)]),
    "rc_engine_forward_fft": (C.c_int, [_eng, _fp, _fp]),
    "rc_engine_resynth": (C.c_int, [_eng, C.c_uint32, C.c_uint64, _fp, _fp]),
    "rc_shard_plan": (_sz, [C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(rc_shard), _sz]),
    "rc_multi_create": (C.c_int, [C.POINTER(rc_config), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(_eng)]),
    "rc_multi_destroy": (None, [_eng]),
    "rc_multi_device_count": (C.c_uint32, [_eng]),
    "rc_multi_set_staging": (C.c_int, [_eng, C.c_int]),
    "rc_multi_stretch_host": (C.c_int, [_eng, C.POINTER(_fp), _sz, C.POINTER(_fp), _sz, C.POINTER(_sz)]),
    "rc_multi_stretch_device": (C.c_int, [_eng, C.c_uint32, C.c_void_p, _sz, _sz, C.c_void_p, _sz, _sz,
                                          C.POINTER(_sz), C.c_void_p]),
    "rc_calib_valu": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, _fp, _fp]),
}

_lib = None


class RocoderError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rocoder_hip error {code}: {msg}")
        self.code = code


def _load(path: str) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7). Two
    # HIP runtimes in one process do not work ("No HIP GPUs are available" for the second), so if
    # torch is installed let it load first: our DT_NEEDED libamdhip64.so.7 then resolves to the
    # runtime that is already mapped. Without torch the system runtime under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return L


def lib() -> C.CDLL:
    """Load the engine library (fails loudly when it has not been built)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH)
    return _lib


_hooks = None


@contextlib.contextmanager
def hooks_library():
    """Tests / A-B tools: inside the block `lib()` is the test-hook build (it alone honours ROCODER_DIAG,
    ROCODER_ROUNDS, ROCODER_MIN_RUN, ROCODER_B4_ROUNDS). Engines created inside keep it for their lifetime."""
    global _lib, _hooks
    if _hooks is None:
        _hooks = _load(HOOKS_PATH)
    prev, _lib = _lib, _hooks
    try:
        yield _hooks
    finally:
        _lib = prev


def check(rc: int, L: C.CDLL | None = None) -> int:
    if rc < 0:
        raise RocoderError(rc, (L or lib()).rc_last_error().decode(errors="replace"))
    return rc
