"""Layout conversion between pack modes -- SURVEY.md section 8(f) row 2.

The reference converts by `unpack()` (dequantise to fp16 W) -> `pack()` (re-quantise with an un-clamped round)
(qllm/auto_model_quantization.py:115-147).  Here the conversion stays in the INTEGER domain: unpack the 4-bit grid and
the zero points, re-pack them in the target layout; scales are copied.  That is exact by construction (no fp16 round
trip) and runs in the library's pack/unpack kernels when the layer is on a HIP device."""
from __future__ import annotations

import torch

from .modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT, WQLinear_GEMM
from .modeling.q_layers.compress_weight import pack_bitstream
from .utils import modelutils

_CLS = {"GPTQ": QuantLinearGPTQ, "GEMM": WQLinear_GEMM, "AWQ": WQLinear_GEMM, "HQQ": QuantLinearHQQ, "ORT": QuantLinearORT}


def repack_layer(layer, new_pack_mode: str):
    """Return a new q_layer of `new_pack_mode` holding the same integers, scales, zeros, g_idx and bias."""
    new_pack_mode = new_pack_mode.upper()
    target = _CLS[new_pack_mode]
    if isinstance(layer, target):
        return layer
    dev = layer.qweight.device
    if target is WQLinear_GEMM and layer.bits != 4:
        raise NotImplementedError("AWQ GEMM layout is 4-bit only")
    if (target is QuantLinearORT or isinstance(layer, QuantLinearORT)) and layer.bits != 4:
        raise NotImplementedError("the ORT blob layout is 4-bit only")
    q = layer.unpack_qweight(dev)          # [K, N] int32, natural order
    z = layer.unpack_qzeros(dev)           # [G, N] int (GPTQ/AWQ/ORT) or fp16 (HQQ, ORT with real-valued zeros)
    scales_gn = layer.scales_gn() if isinstance(layer, QuantLinearORT) else layer.scales  # [G, N]
    new = target(layer.bits, layer.groupsize, layer.infeatures, layer.outfeatures, layer.bias is not None,
                 dtype=layer.scales.dtype)
    new.g_idx = layer.g_idx.clone()
    if target is QuantLinearORT:
        if layer.infeatures % layer.groupsize != 0:
            raise ValueError("the ORT blob layout needs in_features % groupsize == 0")
        new.scales = scales_gn.clone()     # pack_on_device flattens it to [N*G]
        zt = z if z.dtype.is_floating_point else z.to(torch.int32)
        new.pack_on_device(q, zt.to(new.scales.dtype) if z.dtype.is_floating_point else zt)
        if layer.bias is not None:
            new.bias = layer.bias.clone()
        return new.to(dev)
    if target is WQLinear_GEMM:
        k = torch.arange(layer.infeatures, device=layer.g_idx.device) // layer.groupsize
        if not torch.equal(layer.g_idx.to(torch.int64), k):
            # (the reference's own check lets a true act-order g_idx slip through, quant_linear_awq.py:96-103)
            raise ValueError("the AWQ GEMM layout has no act-order: cannot repack a layer with a non-trivial g_idx")
        if z.dtype.is_floating_point:
            raise ValueError("HQQ fp16 zeros cannot be stored in the AWQ layout")
        idx = _awq_index(layer.outfeatures, dev)
        new.qweight = pack_bitstream(q[:, idx], 4, axis=1)
        new.qzeros = pack_bitstream(z.to(torch.int32)[:, idx], 4, axis=1)
    elif target is QuantLinearGPTQ:
        if z.dtype.is_floating_point:
            raise ValueError("HQQ fp16 zeros cannot be stored as packed GPTQ zeros")
        new.qweight = _pack_rows(q, layer.bits)
        new.qzeros = pack_bitstream(z.to(torch.int32), layer.bits, axis=1)
    else:  # HQQ: un-packed zeros in the layer dtype
        new.qweight = _pack_rows(q, layer.bits)
        new.qzeros = z.to(layer.scales.dtype)
    new.scales = scales_gn.clone()
    if layer.bias is not None:
        new.bias = layer.bias.clone()
    return new.to(dev)


def _pack_rows(q, bits):
    if q.is_cuda:
        from . import ops
        return ops.pack_qweight(q.to(torch.int32).contiguous(), "GPTQ", bits)
    return pack_bitstream(q, bits, axis=0)


def _awq_index(n, dev):
    base = torch.arange(0, n, 8, device=dev).unsqueeze(1)
    return (base + torch.tensor((0, 2, 4, 6, 1, 3, 5, 7), device=dev).unsqueeze(0)).reshape(-1)


def repack_to_new_mode(model: torch.nn.Module, new_pack_mode: str):
    """Model-level conversion (auto_model_quantization.py:115-147)."""
    layers = modelutils.find_layers(model, [QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT, WQLinear_GEMM])
    for name, layer in layers.items():
        modelutils.set_op_by_name(model, name, repack_layer(layer, new_pack_mode))
    if hasattr(model, "quant_config"):
        model.quant_config.version = "GEMM" if new_pack_mode.upper() == "AWQ" else new_pack_mode.upper()
    return model
