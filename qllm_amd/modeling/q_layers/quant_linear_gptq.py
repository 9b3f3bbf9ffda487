"""QuantLinearGPTQ with the reference's contract (qllm/modeling/q_layers/quant_linear_gptq.py:92-143); forward is
the fused MI355X kernel path (no W materialisation, no CPU branch)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin, autogptq_compat
from .compress_weight import CompressWeight, general_pack_on_row, general_unpack_on_row


class QuantLinearGPTQ(nn.Module, CompressWeight, HipForwardMixin):
    """Buffers (state-dict compatible with the reference / AutoGPTQ-style checkpoints):
        qweight i32 [K//32*bits, N]   column n = bit stream along K
        qzeros  i32 [ceil(K/g), N//32*bits]
        scales  dtype [ceil(K/g), N]
        g_idx   i32 [K] (registered buffer; default k // g)
        bias    dtype [N] or None
    """

    def __init__(self, bits, groupsize, infeatures, outfeatures, bias, dtype=None):
        super().__init__()
        if bits not in [2, 3, 4, 5, 6, 7, 8]:
            raise NotImplementedError("Only 2,4,5,6,7,8 bits are supported.")
        self.dtype = torch.get_default_dtype() if dtype is None else dtype
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.bits = bits
        self.act_order = None
        self.orig_fp_weight = None
        self.maxq = 2 ** self.bits - 1
        self.groupsize = groupsize if groupsize != -1 else infeatures
        self.pack_mode = "GPTQ"
        groups = math.ceil(infeatures / self.groupsize)
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * self.bits, outfeatures), dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros((groups, outfeatures // 32 * self.bits), dtype=torch.int32))
        self.register_buffer("scales", torch.zeros((groups, outfeatures), dtype=self.dtype))
        self.register_buffer("g_idx", (torch.arange(infeatures) // self.groupsize).to(torch.int32))
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures), dtype=self.dtype))
        else:
            self.bias = None

    def _layout_name(self):
        return "GPTQ"

    def handle_qzeros_for_autogptq(self):
        """AutoGPTQ checkpoints store zero-1: re-pack as (z+1) & mask (reference quant_linear_gptq.py:119-134)."""
        if self.qzeros.numel() == 0:
            return
        qzeros = self.qzeros
        groups = math.ceil(self.infeatures / self.groupsize)
        zeros = torch.zeros((groups, self.outfeatures), dtype=torch.int32, device=qzeros.device)
        general_unpack_on_row(qzeros, zeros, self.bits)
        zeros = (zeros + 1) & (2 ** self.bits - 1)
        new_q = torch.zeros_like(qzeros)
        general_pack_on_row(new_q, zeros, self.bits)
        self.qzeros = new_q
        self._desc = None

    def forward(self, x):
        if self.act_order is None:
            # lazy detect, as the reference: trivial g_idx => first `groupsize` entries are all zero (:137-138)
            self.act_order = bool(self.g_idx[: self.groupsize].sum() != 0)
        g_idx = self.g_idx if self.act_order else None
        # COMPATIBLE_WITH_AUTOGPTQ is read per forward by the reference (:75); it becomes add_zero_bias here
        return self._hip_linear(x, g_idx, autogptq_compat())
