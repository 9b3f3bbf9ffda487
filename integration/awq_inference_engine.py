"""Drop-in `qllm/awq_inference_engine.py` for wejoncy/QLLM on MI355X -- INTEGRATION.md, Level 1.

Replaces the extension built from csrc/awq_cuda (pybind_awq.cpp:13-20).  Self-contained ctypes binding of
`qllm_awq_gemm_forward` / `qllm_linear_forward` (include/qllm_mi355x.h); keeps the pybind signature of
csrc/awq_cuda/quantization/gemm_cuda.h:3-4:
    gemm_forward_cuda(x[M, K] f16, qweight[K, N/8] i32, scales[K/g, N] f16, qzeros[K/g, N/8] i32, split_k_iters) -> y[M, N]
`split_k_iters` is accepted and ignored (the reduction over K is carried in fp32 inside the kernels).

Decode (M <= 64): an AWQ row is only N/2 bytes, so no column strip of that layout covers all of K with whole cache lines and
the in-place kernel has to split K (11-16 us per Llama-7B linear at M = 1).  The first decode call on a weight therefore builds
the library's native strip-major copy of the SAME integers (`qllm_repack_native`: bit-exact, on device, +0.5 byte per weight),
cached per qweight tensor OBJECT (weak references to the caller's three tensors plus their versions: a freed model's entry dies
with its tensors and can never be served to another model that happens to get the same addresses), and later calls stream that
through the full-K strip kernel (3-6 us).  QLLM_AWQ_DECODE_SHADOW=0 keeps every call on the in-place layout.
tests/test_integration_level1_gpu.py runs exactly this file against the oracle.
"""
import ctypes
import os
import weakref

import torch

_lib = ctypes.CDLL(os.environ.get("QLLM_MI355X_LIB", "libqllm_mi355x.so"))
_vp, _i32, _sz = ctypes.c_void_p, ctypes.c_int32, ctypes.c_size_t


class _Weight(ctypes.Structure):  # qllm_weight_t
    _fields_ = [("qweight", _vp), ("scales", _vp), ("qzeros", _vp), ("g_idx", _vp), ("bias", _vp), ("K", _i32), ("N", _i32),
                ("group_size", _i32), ("bits", _i32), ("layout", _i32), ("add_zero_bias", _i32)]


_GPTQ, _AWQ, _NATIVE = 0, 1, 3
_UNSUPPORTED = 2  # QLLM_ERR_UNSUPPORTED
_lib.qllm_awq_gemm_forward.argtypes = [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_awq_gemm_forward.restype = ctypes.c_int
_lib.qllm_linear_forward.argtypes = [ctypes.POINTER(_Weight), _vp, _vp, _i32, _i32, _vp, _sz, _vp]
_lib.qllm_linear_forward.restype = ctypes.c_int
_lib.qllm_native_sizes.argtypes = [ctypes.POINTER(_Weight), ctypes.POINTER(_sz), ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
_lib.qllm_native_sizes.restype = ctypes.c_int
_lib.qllm_repack_native.argtypes = [ctypes.POINTER(_Weight), _vp, _vp, _vp, _vp]
_lib.qllm_repack_native.restype = ctypes.c_int
_lib.qllm_last_error.restype = ctypes.c_char_p
_ws = {}
_native = {}  # id(qweight) -> (weakrefs of (qweight, scales, qzeros), their versions, descriptor, tensors it points into)
_DECODE_MAX_M = 128  # strips to 32 rows, the panel kernel to 128 (csrc/panel.hip): both stream the native copy


def _check(rc):
    if rc:
        raise RuntimeError(_lib.qllm_last_error().decode())


def _ver(t):
    return 0 if t.is_inference() else t._version


def _native_copy(qweight, scales, s16, qzeros, K, N, g, stream):
    """The same 4-bit integers, scales and zero points in the library's native layout ([N/16][K/8][16] words: one contiguous
    region per 16-column strip), or None when the shape cannot be held in it."""
    srcs = (qweight, scales, qzeros)
    hit = _native.get(id(qweight))
    if hit is not None and all(r() is t for r, t in zip(hit[0], srcs)) and hit[1] == tuple(_ver(t) for t in srcs):
        return hit[2]
    src = _Weight(qweight.data_ptr(), s16.data_ptr(), qzeros.data_ptr(), None, None, K, N, g, 4, _AWQ, 0)
    bw, bs, bz = _sz(0), _sz(0), _sz(0)
    if _lib.qllm_native_sizes(ctypes.byref(src), ctypes.byref(bw), ctypes.byref(bs), ctypes.byref(bz)):
        return None
    dev = qweight.device
    nq = torch.empty(bw.value // 4, dtype=torch.int32, device=dev)
    ns = torch.empty(bs.value // 2, dtype=torch.float16, device=dev)
    nz = torch.empty(bz.value // 4, dtype=torch.int32, device=dev)
    _check(_lib.qllm_repack_native(ctypes.byref(src), nq.data_ptr(), ns.data_ptr(), nz.data_ptr(), stream))
    desc = _Weight(nq.data_ptr(), ns.data_ptr(), nz.data_ptr(), None, None, K, N, g, 4, _NATIVE, 0)
    key = id(qweight)
    _native[key] = (tuple(weakref.ref(t) for t in srcs), tuple(_ver(t) for t in srcs), desc, (nq, ns, nz))
    weakref.finalize(qweight, _native.pop, key, None)  # the entry dies with the caller's tensor
    return desc


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
        # the native copy only where the strip kernels take it (group sizes 32 / 64 / 128); the caller's own qzeros tensor is the cache
        # key, so a non-contiguous one (its .contiguous() would be a fresh temporary per call) stays on the in-place path
        if (0 < M <= _DECODE_MAX_M and g in (32, 64, 128) and K % 32 == 0 and N % 16 == 0 and qweight.shape[0] == K
                and scales.dtype == torch.float16 and qzeros.is_contiguous() and os.environ.get("QLLM_AWQ_DECODE_SHADOW", "1") != "0"):
            w = _native_copy(qweight, scales, s16, qzeros, K, N, g, stream)
        else:
            w = None
        rc = _UNSUPPORTED
        if w is not None:
            rc = _lib.qllm_linear_forward(ctypes.byref(w), xc.data_ptr(), y.data_ptr(), M, act, ws.data_ptr(), ws.numel(), stream)
        if rc == _UNSUPPORTED:  # no native copy, or a (shape, M) the native kernels do not serve: the caller's buffers in place
            rc = _lib.qllm_awq_gemm_forward(xc.data_ptr(), qweight.data_ptr(), s16.data_ptr(), qzeros.data_ptr(), split_k_iters, y.data_ptr(),
                                            M, K, N, K // scales.shape[0], act, ws.data_ptr(), ws.numel(), stream)
    _check(rc)
    return y
