"""Drop-in for the reference's `qllm.ort_ops` pybind module (csrc/ort_cuda/ort_ops.cc:199-205): same function
names, argument order and meaning, backed by libqllm_mi355x.so.  `sys.modules["qllm.ort_ops"] = qllm_amd.ort_ops`
makes the reference's own QuantLinearGPTQ (quant_linear_gptq.py:76-82) run on MI355X unchanged."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops


def _f16(t: torch.Tensor) -> torch.Tensor:
    # the reference casts bf16 scales/activations to fp16 per call (ort_ops.cc:79-90,119-138)
    return t if t.dtype == torch.float16 else t.to(torch.float16)


def gemv(input_a: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor,
         g_idx: Optional[torch.Tensor], groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0):
    """ort_ops.gemv (ort_ops.cc:94-140): fused GPTQ-layout dequant+matmul; output shaped like the input's leading dims."""
    for n, t in (("input_a", input_a), ("qweight", qweight), ("scales", scales), ("qzeros", qzeros)):
        ops._check_input(t, n)
    if qweight.dim() != 2:
        raise RuntimeError("qweight must be 2-dimensional")
    ori = scales.dtype
    x2d = input_a.reshape(-1, input_a.shape[-1])
    if ori == torch.bfloat16:
        x2d = _f16(x2d)
    lay = "HQQ" if qzeros.dtype.is_floating_point else "GPTQ"
    w, keep = ops.make_weight(lay, qweight, _f16(scales), _f16(qzeros) if lay == "HQQ" else qzeros, g_idx, None,
                              in_features, qweight.shape[1], groupsize, bits, add_zero_bias)
    try:
        y = ops.linear_forward(w, x2d.contiguous())
    except ops.QllmUnsupported:
        y = torch.matmul(x2d, ops.dequant(w, qweight.device).to(x2d.dtype))
    y = y.reshape(input_a.shape[:-1] + (qweight.shape[1],))
    return y.to(torch.bfloat16) if ori == torch.bfloat16 else y


def dequant(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: Optional[torch.Tensor],
            groupsize: int, bits: int, in_features: int, add_zero_bias: int = 0):
    """ort_ops.dequant (ort_ops.cc:58-92): W[K,N] in the scales' dtype."""
    for n, t in (("qweight", qweight), ("scales", scales), ("qzeros", qzeros)):
        ops._check_input(t, n)
    if qweight.dim() != 2:
        raise RuntimeError("qweight must be 2-dimensional")
    lay = "HQQ" if qzeros.dtype.is_floating_point else "GPTQ"
    w, keep = ops.make_weight(lay, qweight, _f16(scales), _f16(qzeros) if lay == "HQQ" else qzeros, g_idx, None,
                              in_features, qweight.shape[1], groupsize, bits, add_zero_bias)
    out = ops.dequant(w, qweight.device, torch.float16)
    return out.to(torch.bfloat16) if scales.dtype == torch.bfloat16 else out


def Dequantize4Bits(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, g_idx: Optional[torch.Tensor],
                    block_size: int, in_features: int, out_features: int):
    """ort_ops.Dequantize4Bits (ort_ops.cc:161-197): ORT / MatMulNBits blob -> W[out_features, in_features]."""
    return ops.ort_dequantize4bits(qweight, scales, qzeros, g_idx, block_size, in_features, out_features)
