"""torch-facing wrappers over the C ABI: pass data_ptr()s + the current HIP stream, allocate outputs and the
split-K workspace with torch's caching allocator (the library itself never allocates).

PyTorch is plumbing here (device memory, streams); all compute is in libqllm_mi355x.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import (DT_BF16, DT_F16, LAYOUT_AWQ_GEMM, LAYOUT_GPTQ, LAYOUT_HQQ, LAYOUT_NATIVE, LAYOUT_NATIVE_F16Z, QllmError,
                   QllmUnsupported, QllmWeight)

LAYOUTS = {"GPTQ": LAYOUT_GPTQ, "GEMM": LAYOUT_AWQ_GEMM, "AWQ": LAYOUT_AWQ_GEMM, "HQQ": LAYOUT_HQQ, "NATIVE": LAYOUT_NATIVE,
           "NATIVE_F16Z": LAYOUT_NATIVE_F16Z}

_workspaces: dict = {}


def _check_input(t: torch.Tensor, name: str):
    # the reference's CHECK_INPUT (csrc/ort_cuda/ort_ops.cc:6-10): device tensor, contiguous
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a HIP/CUDA tensor (qllm_amd has no CPU forward path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _act_dtype(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return DT_F16
    if t.dtype == torch.bfloat16:
        return DT_BF16
    raise RuntimeError(f"activations must be float16 or bfloat16, got {t.dtype}")


WORKSPACE_BYTES = 64 << 20


def workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per-(device, stream) scratch for split-K slabs + arrival counters; zero-filled once (kernels leave it clean).
    The key uses the CURRENT stream of `device` (not of whatever device happens to be current).

    The persistent buffer is 64 MB and NEVER moves or grows: hipGraphs captured earlier hold its address (arrival counters,
    split-K slabs).  Counters + slabs are bounded by 16 KB + 32 MB for every shape; only the fp16 staging copy of a bf16
    activation (M * K * 2 bytes, dtype-aware sizing: qllm_workspace_bytes_act) can ask for more -- such a call gets a buffer of
    its own from torch's caching allocator (inside a capture: from the graph's private pool, i.e. static for that graph), with
    its counter page zeroed on the call's stream."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = torch.zeros(WORKSPACE_BYTES, dtype=torch.uint8, device=device)
    if nbytes <= ws.numel():
        return ws
    big = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    big[:16384].zero_()
    return big


def make_weight(layout: str, qweight, scales, qzeros, g_idx, bias, in_features: int, out_features: int,
                group_size: int, bits: int, add_zero_bias: int = 0):
    """Build the C descriptor.  Returns (QllmWeight, keepalive tuple of tensors whose pointers it holds)."""
    lay = LAYOUTS[layout.upper()]
    for name, t in (("qweight", qweight), ("scales", scales)):
        _check_input(t, name)
    if scales.dtype != torch.float16:
        raise RuntimeError("scales must be float16 at the C boundary (cast bf16 scales once, like ort_ops.cc:79-90)")
    if qweight.dtype != torch.int32:
        raise RuntimeError("qweight must be int32")
    if qzeros is not None:
        _check_input(qzeros, "qzeros")
        want = torch.float16 if lay in (LAYOUT_HQQ, LAYOUT_NATIVE_F16Z) else torch.int32
        if qzeros.dtype != want:
            raise RuntimeError(f"qzeros must be {want} for layout {layout}")
    if g_idx is not None:
        _check_input(g_idx, "g_idx")
        if g_idx.dtype != torch.int32:
            raise RuntimeError("g_idx must be int32")
    if bias is not None:
        _check_input(bias, "bias")
        if bias.dtype != torch.float16:
            raise RuntimeError("bias must be float16 at the C boundary")
    w = QllmWeight(
        qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr() if qzeros is not None else None,
        g_idx.data_ptr() if g_idx is not None else None, bias.data_ptr() if bias is not None else None,
        int(in_features), int(out_features), int(group_size), int(bits), lay, int(add_zero_bias))
    return w, (qweight, scales, qzeros, g_idx, bias)


def _check_x(x2d: torch.Tensor, ws_desc: Sequence[QllmWeight]):
    _check_input(x2d, "x")
    if x2d.dim() != 2:
        raise RuntimeError(f"x must be 2-D [M, K], got {tuple(x2d.shape)}")
    for w in ws_desc:
        if x2d.shape[1] != w.K:
            raise RuntimeError(f"x must be [M, {w.K}], got {tuple(x2d.shape)}")


def _check_out(o: torch.Tensor, m: int, n_cols: int, x2d: torch.Tensor):
    if o.device != x2d.device or o.dtype != x2d.dtype or tuple(o.shape) != (m, n_cols) or not o.is_contiguous():
        raise RuntimeError(f"out must be a contiguous [{m}, {n_cols}] {x2d.dtype} tensor on {x2d.device}, "
                           f"got {tuple(o.shape)} {o.dtype} on {o.device}")


def linear_forward(w: QllmWeight, x2d: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y[M,N] = x2d[M,K] . dequant(w) (+bias) through the fused HIP kernels.  Raises QllmUnsupported when the
    library has no fused kernel for the configuration (caller may then use dequant() + matmul)."""
    _check_x(x2d, (w,))
    lib = _lib.load()
    m = x2d.shape[0]
    if out is not None:
        _check_out(out, m, w.N, x2d)
    if m == 0:
        return out if out is not None else torch.empty((0, w.N), dtype=x2d.dtype, device=x2d.device)
    if out is None:
        out = torch.empty((m, w.N), dtype=x2d.dtype, device=x2d.device)
    with torch.cuda.device(x2d.device):
        nbytes = lib.qllm_workspace_bytes_act(C.byref(w), m, _act_dtype(x2d))
        ws = workspace(x2d.device, nbytes)
        rc = lib.qllm_linear_forward(C.byref(w), x2d.data_ptr(), out.data_ptr(), m, _act_dtype(x2d), ws.data_ptr(),
                                     ws.numel(), _stream_ptr())
    _lib.check(rc)
    return out


_LAST_CONVERT: dict = {}


def _tensor_version(t) -> int:
    return 0 if t.is_inference() else t._version


def bf16_as_f16(x2d: torch.Tensor, key: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp16 copy of a contiguous bf16 activation matrix (round to nearest even, what `.to(float16)` does), made by the library's
    kernel and REMEMBERED per device for the tensor OBJECT it was made from: q/k/v -- and gate/up -- are called with the same
    tensor one after the other, so three prefill calls convert once.  `key` is that object (module code passes the `x` its forward
    received: `x.reshape(-1, K)` is a fresh object per call and would never hit); it is held weakly and its death drops the copy,
    so the cache never keeps a prefill-sized buffer beyond its input's life.  Inference tensors (no version counter: an in-place
    update could not be seen) are converted every time: the sharing works under `torch.no_grad()`, not under
    `torch.inference_mode()`.  A hit also requires the caller's current stream to be the one the copy was made on."""
    import weakref
    k = x2d if key is None else key
    cacheable = not k.is_inference() and k.device == x2d.device
    dev = x2d.device
    if cacheable:
        hit = _LAST_CONVERT.get(dev)
        # (the copy was produced on ONE stream: a caller on another stream has no ordering with it -- convert again there)
        if hit is not None and hit[0]() is k and hit[1] == k._version and hit[2].shape == x2d.shape and hit[3] == _stream_ptr():
            return hit[2]
    _check_input(x2d, "x")
    out = torch.empty(x2d.shape, dtype=torch.float16, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().qllm_convert_bf16_to_f16(x2d.data_ptr(), out.data_ptr(), x2d.numel(), _stream_ptr()))
    if cacheable:
        def _drop(ref, dev=dev):
            cur = _LAST_CONVERT.get(dev)
            if cur is not None and cur[0] is ref:
                _LAST_CONVERT.pop(dev, None)
        _LAST_CONVERT[dev] = (weakref.ref(k, _drop), k._version, out, _stream_ptr())
    else:
        _LAST_CONVERT.pop(dev, None)
    return out


def linear_forward_bf16_via_f16(w: QllmWeight, x2d: torch.Tensor, key: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Prefill-sized bf16 call with the conversion of x hoisted and shared (QLLM_F16_IN_BF16_OUT): bit-identical to
    `linear_forward(w, x2d)` on the 256x128 prefill kernel, which converts x into the workspace on every call.  Raises
    QllmUnsupported where that kernel does not serve the call (callers then use linear_forward) -- decided from the plan BEFORE x is
    converted, so that a call the panel kernel / gemm2 / the fallbacks take does not pay for a copy it cannot use."""
    _check_x(x2d, (w,))
    if x2d.dtype != torch.bfloat16 or x2d.shape[0] <= 64 or x2d.numel() % 8 != 0:
        raise QllmUnsupported(_lib.QLLM_ERR_UNSUPPORTED, "bf16 prefill calls only")
    if w.bits != 4 or not plan_describe([w], x2d.shape[0]).startswith("gemm3"):
        raise QllmUnsupported(_lib.QLLM_ERR_UNSUPPORTED, "not a call of the 256x128 prefill kernel")
    lib = _lib.load()
    m = x2d.shape[0]
    xh = bf16_as_f16(x2d, key)
    out = torch.empty((m, w.N), dtype=torch.bfloat16, device=x2d.device)
    with torch.cuda.device(x2d.device):
        nbytes = lib.qllm_workspace_bytes_act(C.byref(w), m, DT_F16)
        ws = workspace(x2d.device, nbytes)
        rc = lib.qllm_linear_forward(C.byref(w), xh.data_ptr(), out.data_ptr(), m, _lib.DT_F16_IN_BF16_OUT, ws.data_ptr(), ws.numel(),
                                     _stream_ptr())
    _lib.check(rc)
    return out


def _bf16_native(w: QllmWeight) -> bool:
    """Round 6: the 256x128 prefill kernel takes bf16 activations as they are on the 4-bit row-stream / strip-major layouts (bf16 W,
    bf16 MFMA: csrc/gemm3.hip, BF) -- nothing to convert, nothing to share.  `set_knob("QLLM_GEMM3_BF16", 0)` brings the fp16
    conversion pre-pass (the reference's shim, quant_linear_awq.py:29-36) back."""
    return (w.bits == 4 and w.layout in (LAYOUTS["GPTQ"], LAYOUTS["HQQ"], LAYOUTS["NATIVE"], LAYOUTS["NATIVE_F16Z"])
            and get_knob("QLLM_GEMM3_BF16") != 0)


def linear_forward_shared(w: QllmWeight, x2d: torch.Tensor, key: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear_forward for module code: bf16 prefill calls go through the shared fp16 copy of x where the kernel allows it.  `key`:
    the tensor object whose identity marks "the same input" across sibling calls (default: x2d itself)."""
    if x2d.dtype == torch.bfloat16 and x2d.shape[0] > 64 and not _bf16_native(w):
        try:
            return linear_forward_bf16_via_f16(w, x2d, key)
        except QllmUnsupported:
            pass
    return linear_forward(w, x2d)


def linear_forward_grouped(ws_desc: Sequence[QllmWeight], x2d: torch.Tensor,
                           outs: Optional[Sequence[torch.Tensor]] = None):
    """Several layers sharing x (q/k/v, gate/up) in ONE launch (decode sizes only)."""
    _check_x(x2d, ws_desc)
    lib = _lib.load()
    n = len(ws_desc)
    m = x2d.shape[0]
    if outs is not None:
        if len(outs) != n:
            raise RuntimeError(f"outs must hold {n} tensors, got {len(outs)}")
        for o, w in zip(outs, ws_desc):
            _check_out(o, m, w.N, x2d)
    if m == 0:
        return list(outs) if outs is not None else [torch.empty((0, w.N), dtype=x2d.dtype, device=x2d.device) for w in ws_desc]
    if outs is None:
        outs = [torch.empty((m, w.N), dtype=x2d.dtype, device=x2d.device) for w in ws_desc]
    arr = (QllmWeight * n)(*ws_desc)
    ys = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    with torch.cuda.device(x2d.device):
        nbytes = sum(lib.qllm_workspace_bytes_act(C.byref(w), m, _act_dtype(x2d)) for w in ws_desc)
        wsp = workspace(x2d.device, nbytes)
        rc = lib.qllm_linear_forward_grouped(arr, ys, n, x2d.data_ptr(), m, _act_dtype(x2d), wsp.data_ptr(),
                                             wsp.numel(), _stream_ptr())
    _lib.check(rc)
    return list(outs)


def dequant(w: QllmWeight, device: torch.device, dtype=torch.float16, transposed: bool = False) -> torch.Tensor:
    """W[K,N] (or [N,K]) bit-identical to the reference's DequantizeLinearBlockWise / unpack()."""
    lib = _lib.load()
    shape = (w.N, w.K) if transposed else (w.K, w.N)
    out = torch.empty(shape, dtype=dtype, device=device)
    dt = DT_F16 if dtype == torch.float16 else DT_BF16
    if dtype not in (torch.float16, torch.bfloat16):
        raise RuntimeError("dequant output must be float16 or bfloat16")
    with torch.cuda.device(device):
        rc = lib.qllm_dequant(C.byref(w), out.data_ptr(), dt, 1 if transposed else 0, _stream_ptr())
    _lib.check(rc)
    return out


def plan_describe(ws_desc: Sequence[QllmWeight], m: int, have_workspace: bool = True) -> str:
    """Which kernel `linear_forward` (one descriptor) / `linear_forward_grouped` (several) would run for m rows."""
    arr = (QllmWeight * len(ws_desc))(*ws_desc)
    buf = C.create_string_buffer(256)
    _lib.check(_lib.load().qllm_plan_describe(arr, len(ws_desc), int(m), 1 if have_workspace else 0, buf, 256))
    return buf.value.decode()


def set_knob(name: str, value: int) -> None:
    """Move one of the planner's thresholds (include/qllm_mi355x.h, qllm_set_knob: e.g. QLLM_GEMM3_MIN_M, QLLM_PANEL_GROUP_MIN_M) for
    this process; `plan_describe` reflects it.  Raises for names / values the release library does not accept."""
    _lib.check(_lib.load().qllm_set_knob(name.encode(), int(value)))


def get_knob(name: str) -> Optional[int]:
    """The override set for `name`, or None when the planner uses its measured default."""
    v, is_set = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.load().qllm_get_knob(name.encode(), C.byref(v), C.byref(is_set)))
    return int(v.value) if is_set.value else None


def reset_knobs() -> None:
    _lib.load().qllm_reset_knobs()


def ort_dequantize4bits(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: Optional[torch.Tensor],
                        block_size: int, in_features: int, out_features: int) -> torch.Tensor:
    """ORT / MatMulNBits blob -> W[N,K] in the scales' dtype (ort_ops.Dequantize4Bits, ort_ops.cc:161-197)."""
    for name, t in (("qweight", qweight), ("scales", scales), ("qzeros", qzeros)):
        _check_input(t, name)
    if qweight.dtype != torch.uint8:
        raise RuntimeError("qweight must be uint8 (ORT blob layout)")
    zf = qzeros.dtype != torch.uint8
    s16 = scales if scales.dtype == torch.float16 else scales.to(torch.float16)
    z = qzeros if (not zf or qzeros.dtype == torch.float16) else qzeros.to(torch.float16)
    gi = None
    if g_idx is not None:
        gi = g_idx.to(device=qweight.device, dtype=torch.int32).contiguous()
    out = torch.empty((out_features, in_features), dtype=torch.float16, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = _lib.load().qllm_ort_dequantize4bits(qweight.data_ptr(), s16.data_ptr(), z.data_ptr(), 1 if zf else 0,
                                                  gi.data_ptr() if gi is not None else None, block_size, in_features,
                                                  out_features, out.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return out if scales.dtype == torch.float16 else out.to(scales.dtype)


def gather_columns(x2d: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    """x2d[:, perm] for a contiguous [M, K] fp16 / bf16 matrix and an int32 device permutation (act-order layers: the
    activation side of the row-sorted weight copy).  The library's LDS-staged gather; shapes it does not take
    (K % 8, K > 28672) fall back to index_select."""
    _check_x(x2d, ())
    if perm.dtype != torch.int32 or perm.device != x2d.device or perm.numel() != x2d.shape[1] or not perm.is_contiguous():
        raise RuntimeError("perm must be a contiguous int32 tensor of K entries on x's device")
    out = torch.empty_like(x2d)
    if x2d.shape[0] == 0:
        return out
    with torch.cuda.device(x2d.device):
        rc = _lib.load().qllm_gather_columns(x2d.data_ptr(), perm.data_ptr(), out.data_ptr(), x2d.shape[0], x2d.shape[1],
                                             _act_dtype(x2d), _stream_ptr())
    if rc == _lib.QLLM_ERR_UNSUPPORTED:
        return x2d.index_select(1, perm.long())
    _lib.check(rc)
    return out


def unpack_qweight(qweight: torch.Tensor, layout: str, bits: int, in_features: int, out_features: int) -> torch.Tensor:
    _check_input(qweight, "qweight")
    q = torch.empty((in_features, out_features), dtype=torch.int32, device=qweight.device)
    with torch.cuda.device(qweight.device):
        rc = _lib.load().qllm_unpack_qweight(qweight.data_ptr(), LAYOUTS[layout.upper()], bits, in_features,
                                             out_features, q.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return q


def pack_qweight(q_kn: torch.Tensor, layout: str, bits: int) -> torch.Tensor:
    _check_input(q_kn, "q_kn")
    if q_kn.dtype != torch.int32:
        raise RuntimeError("q_kn must be int32")
    k, n = q_kn.shape
    lay = LAYOUTS[layout.upper()]
    shape = (k, n // 8) if lay == LAYOUT_AWQ_GEMM else (k * bits // 32, n)
    out = torch.empty(shape, dtype=torch.int32, device=q_kn.device)
    with torch.cuda.device(q_kn.device):
        rc = _lib.load().qllm_pack_qweight(q_kn.data_ptr(), lay, bits, k, n, out.data_ptr(), _stream_ptr())
    _lib.check(rc)
    return out


def repack_native(w: QllmWeight, keep):
    """The layer behind descriptor `w` (GPTQ / AWQ GEMM / HQQ buffers, no g_idx) re-laid-out into the library's strip-major native
    layout (include/qllm_mi355x.h, "native layout") on its device: returns (QllmWeight, keepalive) like make_weight.  A pure integer
    permutation (qllm_repack_native); the bias tensor is shared.  Raises QllmUnsupported for shapes the layout cannot hold."""
    lib = _lib.load()
    qweight, scales, qzeros, _g, bias = keep
    dev = qweight.device
    sz = [C.c_size_t(0) for _ in range(3)]
    _lib.check(lib.qllm_native_sizes(C.byref(w), C.byref(sz[0]), C.byref(sz[1]), C.byref(sz[2])))
    nq = torch.empty(sz[0].value // 4, dtype=torch.int32, device=dev)
    ns = torch.empty(sz[1].value // 2, dtype=torch.float16, device=dev)
    f16z = w.layout == LAYOUT_HQQ
    nz = None
    if sz[2].value:
        nz = torch.empty(sz[2].value // 2, dtype=torch.float16, device=dev) if f16z else torch.empty(sz[2].value // 4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.qllm_repack_native(C.byref(w), nq.data_ptr(), ns.data_ptr(), nz.data_ptr() if nz is not None else None, _stream_ptr())
    _lib.check(rc)
    out = QllmWeight(nq.data_ptr(), ns.data_ptr(), nz.data_ptr() if nz is not None else None, None,
                     bias.data_ptr() if bias is not None else None, w.K, w.N, w.group_size, w.bits,
                     LAYOUT_NATIVE_F16Z if f16z else LAYOUT_NATIVE, w.add_zero_bias)
    return out, (nq, ns, nz, None, bias)


def unpack_native(w: QllmWeight, keep, layout: str):
    """Inverse of repack_native: the reference buffers (qweight, scales, qzeros) of `layout` ("GPTQ", "GEMM" or "HQQ"), bit-exact."""
    lib = _lib.load()
    nq = keep[0]
    dev = nq.device
    lay = LAYOUTS[layout.upper()]
    groups = (w.K + w.group_size - 1) // w.group_size
    qshape = (w.K, w.N // 8) if lay == LAYOUT_AWQ_GEMM else (w.K * w.bits // 32, w.N)
    qweight = torch.empty(qshape, dtype=torch.int32, device=dev)
    scales = torch.empty((groups, w.N), dtype=torch.float16, device=dev)
    qzeros = None
    if keep[2] is not None:
        qzeros = (torch.empty((groups, w.N), dtype=torch.float16, device=dev) if lay == LAYOUT_HQQ
                  else torch.empty((groups, w.N * w.bits // 32), dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        rc = lib.qllm_unpack_native(C.byref(w), lay, qweight.data_ptr(), scales.data_ptr(),
                                    qzeros.data_ptr() if qzeros is not None else None, _stream_ptr())
    _lib.check(rc)
    return qweight, scales, qzeros


__all__ = ["make_weight", "linear_forward", "linear_forward_grouped", "dequant", "gather_columns", "unpack_qweight", "pack_qweight",
           "workspace", "QllmUnsupported", "LAYOUTS", "plan_describe", "repack_native", "unpack_native"]
