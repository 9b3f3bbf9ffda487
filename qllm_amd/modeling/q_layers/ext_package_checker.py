"""Presence probes the dispatch keys on (reference: qllm/modeling/q_layers/ext_package_checker.py:8-28).

The reference asks "is the CUDA extension importable and is sm >= 75".  Here the single native library serves
both roles; the probe is "libqllm_mi355x.so is built AND device 0 reports gfx950".  There is no CPU fallback
behind a False answer: forward() raises.
"""
import functools

from ... import _lib


@functools.lru_cache()
def has_package(package_name: str) -> bool:
    """True iff `package_name` can actually be imported (kept for callers that probe optional packages by name)."""
    import importlib
    import importlib.util

    try:
        spec = importlib.util.find_spec(package_name)
    except (ImportError, ValueError):
        return False
    if spec is None:
        return False
    try:
        importlib.import_module(package_name)
    except Exception:  # noqa: BLE001
        return False
    return True


@functools.lru_cache()
def has_mi355x_engine() -> bool:
    import torch

    if not _lib.is_built() or not torch.cuda.is_available():
        return False
    try:
        return _lib.device_info(0)["arch"].startswith("gfx950")
    except Exception:  # noqa: BLE001
        return False


def has_awq_inference_engine() -> bool:
    return has_mi355x_engine()


def is_the_machine_support_awq_engine(nbits: int) -> bool:
    return has_awq_inference_engine() and nbits == 4


def has_ort_ops() -> bool:
    return has_mi355x_engine()
