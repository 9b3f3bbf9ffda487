"""QuantLinearHQQ with the reference's contract (qllm/modeling/q_layers/quant_linear_hqq.py:47-80): GPTQ-style
qweight, un-packed non-integer fp16 zeros.  The reference has no native kernel for it (pure torch on every
device); here it shares the fused MI355X kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin, export_module_hooks
from .compress_weight import CompressWeight


@export_module_hooks
class QuantLinearHQQ(nn.Module, CompressWeight, HipForwardMixin):
    """Buffers: qweight i32 [K//32*bits, N] (column bit streams, as GPTQ); qzeros and scales in the module dtype,
    [ceil(K/g), N] each (the zeros are real numbers: HQQ does not round them); bias [N] or None.  g_idx is a plain
    attribute (never act-order), exactly as in the reference."""


    SUPPORTED_BITS = (2, 3, 4, 5, 6, 7, 8)

    def __init__(self, bits, groupsize, infeatures, outfeatures, bias, dtype=None):
        super().__init__()
        if bits not in self.SUPPORTED_BITS:
            raise NotImplementedError("Only 2,4,5,6,7,8 bits are supported.")
        self.dtype = dtype if dtype is not None else torch.get_default_dtype()
        self.bits, self.infeatures, self.outfeatures = bits, infeatures, outfeatures
        self.groupsize = infeatures if groupsize == -1 else groupsize
        self.pack_mode, self.orig_fp_weight = "HQQ", None
        n_groups = math.ceil(infeatures / self.groupsize)
        self.g_idx = torch.div(torch.arange(infeatures), self.groupsize, rounding_mode="floor").to(torch.int32)
        per_group = lambda: torch.zeros((n_groups, outfeatures), dtype=self.dtype)  # noqa: E731
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * bits, outfeatures), dtype=torch.int32))
        self.register_buffer("qzeros", per_group())
        self.register_buffer("scales", per_group())
        if bias:
            self.register_buffer("bias", torch.zeros(outfeatures, dtype=self.dtype))
        else:
            self.bias = None

    def _layout_name(self):
        return "HQQ"

    def unpack_qzeros(self, device):
        self._real_buffers()
        return self.qzeros.to(device)

    def pack_qzeros(self, intzeros, device):
        self.qzeros = intzeros.contiguous().to("cpu")

    def forward(self, x):
        return self._hip_linear(x, None, 0)
