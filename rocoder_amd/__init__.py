"""rocoder_amd — MI355X (gfx950) engine for rocoder's analysis -> kernel -> resynthesis ->
overlap-add stretch path. The compute lives in librocoder_hip.so (hand-written HIP, C-ABI in
include/rocoder_hip.h); this package is the host-side mirror of the reference interface."""
from .stretcher import (AudioBus, AudioSpec, Engine, MultiEngine, ReFFT, RocoderError, Stretcher,  # noqa: F401
                        StretcherProcessor, derive_params, load_kernel_library,
                        offline_output_len, pinned_empty, stretch)

__version__ = "0.1.0"
