"""QuantLinearHQQ with the reference's contract (qllm/modeling/q_layers/quant_linear_hqq.py:47-80): GPTQ-style
qweight, un-packed non-integer fp16 zeros.  The reference has no native kernel for it (pure torch on every
device); here it shares the fused MI355X kernels."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin
from .compress_weight import CompressWeight


class QuantLinearHQQ(nn.Module, CompressWeight, HipForwardMixin):
    def __init__(self, bits, groupsize, infeatures, outfeatures, bias, dtype=None):
        super().__init__()
        self.dtype = torch.get_default_dtype() if dtype is None else dtype
        if bits not in [2, 3, 4, 5, 6, 7, 8]:
            raise NotImplementedError("Only 2,4,5,6,7,8 bits are supported.")
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.bits = bits
        self.groupsize = groupsize if groupsize != -1 else infeatures
        self.pack_mode = "HQQ"
        self.orig_fp_weight = None
        self.g_idx = (torch.arange(infeatures) // self.groupsize).to(torch.int32)  # plain attribute, as the reference
        groups = math.ceil(infeatures / self.groupsize)
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * self.bits, outfeatures), dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros((groups, outfeatures), dtype=self.dtype))
        self.register_buffer("scales", torch.zeros((groups, outfeatures), dtype=self.dtype))
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures), dtype=self.dtype))
        else:
            self.bias = None

    def _layout_name(self):
        return "HQQ"

    def unpack_qzeros(self, device):
        return self.qzeros.to(device)

    def pack_qzeros(self, intzeros, device):
        self.qzeros = intzeros.contiguous().to("cpu")

    def forward(self, x):
        return self._hip_linear(x, None, 0)
