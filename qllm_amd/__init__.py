"""qllm_amd: MI355X-native (gfx950 / CDNA4) fused int4 dequant + matmul for QLLM-format quantized linears.

Only the inference hot path of wejoncy/QLLM is here: the q_layer modules with the
reference's constructor / buffers / pack / unpack contract, the dispatch helpers, and libqllm_mi355x.so
(hand-written HIP kernels behind a C ABI, include/qllm_mi355x.h).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"


def is_available() -> bool:
    """True iff the HIP library is built and device 0 is gfx950."""
    import torch

    if not _lib.is_built() or not torch.cuda.is_available():
        return False
    try:
        return _lib.device_info(0)["arch"].startswith("gfx950")
    except Exception:
        return False
