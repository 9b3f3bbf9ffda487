"""Drop-in `qllm/awq_inference_engine.py` for wejoncy/QLLM on MI355X -- INTEGRATION.md, Level 1.

Replaces the extension built from csrc/awq_cuda (pybind_awq.cpp:13-20).  Self-contained ctypes binding of
`qllm_awq_gemm_forward` / `qllm_linear_forward` (include/qllm_mi355x.h); keeps the pybind signature of
csrc/awq_cuda/quantization/gemm_cuda.h:3-4:
    gemm_forward_cuda(x[M, K] f16, qweight[K, N/8] i32, scales[K/g, N] f16, qzeros[K/g, N/8] i32, split_k_iters) -> y[M, N]
`split_k_iters` is accepted and ignored (the reduction over K is carried in fp32 inside the kernels).

Decode (M <= 64): an AWQ row is only N/2 bytes, so no column strip of that layout covers all of K with whole cache lines and
the in-place kernel has to split K (11-16 us per Llama-7B linear at M = 1).  The first decode call on a weight therefore builds
a row-stream copy of the SAME integers (bit-exact, on device, with the library's unpack / pack kernels; +0.5 byte per weight),
cached on the identity and version of the caller's tensors, and later calls stream that through the full-K strip kernel
(5-8 us).  QLLM_AWQ_DECODE_SHADOW=0 keeps every call on the in-place layout.
tests/test_integration_level1_gpu.py runs exactly this file against the oracle.
"""
import ctypes
import os

import torch

_lib = ctypes.CDLL(os.environ.get("QLLM_MI355X_LIB", "libqllm_mi355x.so"))
_vp, _i32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t


class _Weight(ctypes.Structure):  # qllm_weight_t
    _fields_ = [("qweight", _vp), ("scales", _vp), ("qzeros", _vp), ("g_idx", _vp), ("bias", _vp), ("K", _i32), ("N", _i32),
                ("group_size", _i32), ("bits", _i32), ("layout", _i32), ("add_zero_bias", _i32)]


_GPTQ, _AWQ = 0, 1
_lib.qllm_awq_gemm_forward.argtypes = [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_awq_gemm_forward.restype = ctypes.c_int
_lib.qllm_linear_forward.argtypes = [ctypes.POINTER(_Weight), _vp, _vp, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_linear_forward.restype = ctypes.c_int
_lib.qllm_unpack_qweight.argtypes = [_vp, _i32, _i32, _i32, _i32, _vp, _vp]
_lib.qllm_unpack_qweight.restype = ctypes.c_int
_lib.qllm_pack_qweight.argtypes = [_vp, _i32, _i32, _i32, _i32, _vp, _vp]
_lib.qllm_pack_qweight.restype = ctypes.c_int
_lib.qllm_last_error.restype = ctypes.c_char_p
_ws = {}
_rows = {}  # identity + version of (qweight, scales, qzeros) -> (descriptor, tensors it points into)
_DECODE_MAX_M = 64


def _check(rc):
    if rc:
        raise RuntimeError(_lib.qllm_last_error().decode())


def _row_stream(qweight, s16, qzeros, K, N, g, stream):
    """The same 4-bit integers as [K/8, N] words (8 consecutive k of one column per word) + zero points in natural column order."""
    key = tuple((t.data_ptr(), t._version) for t in (qweight, s16, qzeros)) + (K, N)
    hit = _rows.get(key)
    if hit is None:
        dev = qweight.device
        q = torch.empty((K, N), dtype=torch.int32, device=dev)
        _check(_lib.qllm_unpack_qweight(qweight.data_ptr(), _AWQ, 4, K, N, q.data_ptr(), stream))
        qw = torch.empty((K // 8, N), dtype=torch.int32, device=dev)
        _check(_lib.qllm_pack_qweight(q.data_ptr(), _GPTQ, 4, K, N, qw.data_ptr(), stream))
        groups = qzeros.shape[0]
        z = torch.empty((groups, N), dtype=torch.int32, device=dev)  # zero points share the weights' column interleave
        _check(_lib.qllm_unpack_qweight(qzeros.contiguous().data_ptr(), _AWQ, 4, groups, N, z.data_ptr(), stream))
        shifts = torch.arange(0, 32, 4, device=dev, dtype=torch.int64)
        qz = (z.to(torch.int64).view(groups, N // 8, 8) << shifts).sum(-1).to(torch.int32).contiguous()  # (wraps into the sign bit)
        desc = _Weight(qw.data_ptr(), s16.data_ptr(), qz.data_ptr(), None, None, K, N, g, 4, _GPTQ, 0)
        if len(_rows) > 4096:  # weights that were re-created many times: do not grow without bound
            _rows.clear()
        hit = _rows[key] = (desc, (qw, qz, s16))
    return hit[0]


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
    s16 = (scales if scales.dtype == torch.float16 else scales.to(torch.float16)).contiguous()
    xc = x.contiguous()
    act = 0 if x.dtype == torch.float16 else 1
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        g = K // scales.shape[0] if scales.shape[0] and K % scales.shape[0] == 0 else 0
        if (0 < M <= _DECODE_MAX_M and g > 0 and K % 32 == 0 and N % 16 == 0 and qweight.shape[0] == K and scales.dtype == torch.float16
                and os.environ.get("QLLM_AWQ_DECODE_SHADOW", "1") != "0"):
            w = _row_stream(qweight, s16, qzeros, K, N, g, stream)
            rc = _lib.qllm_linear_forward(ctypes.byref(w), xc.data_ptr(), y.data_ptr(), M, act, ws.data_ptr(), ws.numel(), stream)
        else:
            rc = _lib.qllm_awq_gemm_forward(xc.data_ptr(), qweight.data_ptr(), s16.data_ptr(), qzeros.data_ptr(), split_k_iters, y.data_ptr(),
                                            M, K, N, K // scales.shape[0], act, ws.data_ptr(), ws.numel(), stream)
    _check(rc)
    return y
