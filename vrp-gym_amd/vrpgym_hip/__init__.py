"""ctypes binding of libvrpgym_hip.so (see include/vrpgym_hip.h).

PyTorch is used for device memory and streams only; all per-step work is in the
HIP library.  There is no CPU fallback: every entry point raises if the library
or the GPU is missing.
"""
from ._lib import (  # noqa: F401
    ABI_VERSION, MAX_LAYERS, KIND_IRP, KIND_TSP, KIND_VRP, DecoderGrads, DecoderWeights, EncoderGrads, EncoderWeights, Env, RolloutIO,
    check, current_stream, lib, library_path, ptr, require_gpu,
)
