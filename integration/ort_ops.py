"""Drop-in `qllm/ort_ops.py` for wejoncy/QLLM on MI355X -- the reference-side binding of INTEGRATION.md, Level 1.

Replaces the CUDAExtension built from csrc/ort_cuda (reference setup.py:185-194).  Self-contained: ctypes over the C ABI of
libqllm_mi355x.so (include/qllm_mi355x.h) and torch for device memory / the current stream; it imports nothing from qllm_amd.
The three functions keep the pybind names, argument order and return convention of csrc/ort_cuda/ort_ops.cc:
    gemv(x, qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias) -> y[..., N]       (:94-98)
    dequant(qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias) -> W[K, N] f16      (:58-63)
    Dequantize4Bits(qweight u8, scales, qzeros, g_idx, block_size, in_features, out_features) -> W[N, K]     (:161-166)
tests/test_integration_level1_gpu.py runs exactly this file against the oracle.
"""
import ctypes
import os

import torch

_lib = ctypes.CDLL(os.environ.get("QLLM_MI355X_LIB", "libqllm_mi355x.so"))
_vp, _i32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
_lib.qllm_ort_gemv.argtypes = [_vp] * 5 + [_i32] * 4 + [_vp, _i32, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_ort_gemv.restype = ctypes.c_int
_lib.qllm_ort_dequant.argtypes = [_vp] * 4 + [_i32] * 4 + [_vp, _i32, _vp]
_lib.qllm_ort_dequant.restype = ctypes.c_int
_lib.qllm_ort_dequantize4bits.argtypes = [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp]
_lib.qllm_ort_dequantize4bits.restype = ctypes.c_int
_lib.qllm_last_error.restype = ctypes.c_char_p
_ws = {}


def _check(rc):
    if rc:
        raise RuntimeError(_lib.qllm_last_error().decode())  # TORCH_CHECK -> RuntimeError, as before (ort_ops.cc:67-73)


def _workspace(device):
    """Scratch for the split-K paths: 64 MB covers every shape; zero-filled once, the kernels leave it clean; one per
    (device, stream)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    if key not in _ws:
        _ws[key] = torch.zeros(64 << 20, dtype=torch.uint8, device=device)
    return _ws[key]


def _f16(t):
    return t if t.dtype == torch.float16 else t.to(torch.float16)  # the reference casts bf16 scales too (ort_ops.cc:79-90)


def gemv(x, qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias):
    if not (x.is_cuda and qweight.is_cuda):
        raise RuntimeError("ort_ops.gemv needs device tensors")  # CHECK_INPUT (ort_ops.cc:6-10)
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    n = qweight.shape[1]
    y = torch.empty((x2.shape[0], n), dtype=x.dtype, device=x.device)
    ws = _workspace(x.device)
    scales = _f16(scales).contiguous()
    with torch.cuda.device(x.device):
        _check(_lib.qllm_ort_gemv(x2.data_ptr(), qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
                                  g_idx.data_ptr() if g_idx is not None else None, groupsize, bits, in_features, add_zero_bias,
                                  y.data_ptr(), x2.shape[0], n, 0 if x.dtype == torch.float16 else 1,
                                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream))
    return y.reshape(x.shape[:-1] + (n,))


def dequant(qweight, scales, qzeros, g_idx, groupsize, bits, in_features, add_zero_bias):
    out = torch.empty((in_features, qweight.shape[1]), dtype=torch.float16, device=qweight.device)
    scales16 = _f16(scales).contiguous()
    with torch.cuda.device(qweight.device):
        _check(_lib.qllm_ort_dequant(qweight.data_ptr(), scales16.data_ptr(), qzeros.data_ptr(),
                                     g_idx.data_ptr() if g_idx is not None else None, groupsize, bits, in_features,
                                     add_zero_bias, out.data_ptr(), qweight.shape[1], torch.cuda.current_stream().cuda_stream))
    return out if scales.dtype == torch.float16 else out.to(scales.dtype)


def Dequantize4Bits(qweight, scales, qzeros, g_idx, block_size, in_features, out_features):
    out = torch.empty((out_features, in_features), dtype=torch.float16, device=qweight.device)
    scales16 = _f16(scales).contiguous()
    zeros_f16 = int(qzeros.dtype != torch.uint8)
    z = qzeros if not zeros_f16 else _f16(qzeros).contiguous()
    with torch.cuda.device(qweight.device):
        _check(_lib.qllm_ort_dequantize4bits(qweight.data_ptr(), scales16.data_ptr(), z.data_ptr(), zeros_f16,
                                             g_idx.data_ptr() if g_idx is not None else None, block_size, in_features,
                                             out_features, out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out if scales.dtype == torch.float16 else out.to(scales.dtype)
