"""Drop-in `qllm/awq_inference_engine.py` for wejoncy/QLLM on MI355X -- INTEGRATION.md, Level 1.

Replaces the extension built from csrc/awq_cuda (pybind_awq.cpp:13-20).  Self-contained ctypes binding of
`qllm_awq_gemm_forward` (include/qllm_mi355x.h); keeps the pybind signature of csrc/awq_cuda/quantization/gemm_cuda.h:3-4:
    gemm_forward_cuda(x[M, K] f16, qweight[K, N/8] i32, scales[K/g, N] f16, qzeros[K/g, N/8] i32, split_k_iters) -> y[M, N]
`split_k_iters` is accepted and ignored (the reduction over K is carried in fp32 inside the kernels).
tests/test_integration_level1_gpu.py runs exactly this file against the oracle.
"""
import ctypes
import os

import torch

_lib = ctypes.CDLL(os.environ.get("QLLM_MI355X_LIB", "libqllm_mi355x.so"))
_vp, _i32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t
_lib.qllm_awq_gemm_forward.argtypes = [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_awq_gemm_forward.restype = ctypes.c_int
_lib.qllm_last_error.restype = ctypes.c_char_p
_ws = {}


def gemm_forward_cuda(x, qweight, scales, qzeros, split_k_iters):
    if x.dim() != 2:
        raise ValueError("x must be [M, K]")
    M, K = x.shape
    N = qweight.shape[1] * 8
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    if key not in _ws:
        _ws[key] = torch.zeros(64 << 20, dtype=torch.uint8, device=x.device)  # zero-filled once; kernels leave it clean
    ws = _ws[key]
    s16 = scales if scales.dtype == torch.float16 else scales.to(torch.float16)
    with torch.cuda.device(x.device):
        rc = _lib.qllm_awq_gemm_forward(x.contiguous().data_ptr(), qweight.data_ptr(), s16.contiguous().data_ptr(), qzeros.data_ptr(),
                                        split_k_iters, y.data_ptr(), M, K, N, K // scales.shape[0],
                                        0 if x.dtype == torch.float16 else 1, ws.data_ptr(), ws.numel(),
                                        torch.cuda.current_stream().cuda_stream)
    if rc:
        raise RuntimeError(_lib.qllm_last_error().decode())
    return y
