"""WQLinear_GEMM with the reference's contract (qllm/modeling/q_layers/quant_linear_awq.py:38-153): AWQ "GEMM"
layout -- qweight i32 [K, N/8], nibble i of word (k, j) holds column 8j + [0,2,4,6,1,3,5,7][i]; qzeros i32
[K/g, N/8] interleaved the same way; scales [K/g, N].  Forward = the fused MI355X kernels reading that layout in
place (the reference calls awq_inference_engine.gemm_forward_cuda, :142-148)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin, export_module_hooks
from .compress_weight import CompressWeight, pack_bitstream, unpack_bitstream

AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)


def _awq_col_index(n: int, device) -> torch.Tensor:
    """idx[8j+i] = 8j + AWQ_ORDER[i]: gather with it = natural -> AWQ order."""
    base = torch.arange(0, n, 8, device=device).unsqueeze(1)
    return (base + torch.tensor(AWQ_ORDER, device=device).unsqueeze(0)).reshape(-1)


@export_module_hooks
class WQLinear_GEMM(nn.Module, CompressWeight, HipForwardMixin):
    def __init__(self, w_bit, group_size, in_features, out_features, bias, dtype=None):
        super().__init__()
        self.dtype = torch.get_default_dtype() if dtype is None else dtype
        if w_bit not in [4]:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.infeatures = in_features
        self.outfeatures = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.groupsize = self.group_size
        self.bits = w_bit
        self.orig_fp_weight = None
        self.pack_mode = "GEMM"
        # quick sanity check (alignment), as the reference (:55-56)
        assert self.infeatures % self.group_size == 0
        assert out_features % (32 // self.w_bit) == 0
        self.g_idx = (torch.arange(in_features) // self.group_size).to(torch.int32)  # plain attribute
        pack = 32 // self.w_bit
        self.register_buffer("qweight", torch.zeros((in_features, out_features // pack), dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros((in_features // self.group_size, out_features // pack), dtype=torch.int32))
        self.register_buffer("scales", torch.zeros((in_features // self.group_size, out_features), dtype=self.dtype))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features), dtype=self.dtype))
        else:
            self.bias = None

    def _layout_name(self):
        return "GEMM"

    # ---- AWQ interleave ---------------------------------------------------------------------------------------
    def reorder_int_tensor(self, int_tensor):
        """[R, N] natural columns -> [N, R] rows in AWQ nibble order (reference :95-119: reorder then transpose).
        Refuses act-order like the reference (:96-103)."""
        if self.g_idx is not None:
            self.act_order = self.g_idx[: self.group_size // self.bits].sum().item() != 0
            trivial = (torch.arange(self.infeatures, device=self.g_idx.device) // self.groupsize).to(torch.int32)
            assert self.act_order is True or torch.equal(self.g_idx.to(torch.int32), trivial)
        assert int_tensor.shape[-1] % (32 // self.bits) == 0
        idx = _awq_col_index(int_tensor.shape[1], int_tensor.device)
        return int_tensor[:, idx].T.contiguous()

    def reverse_reorder_int_tensor(self, int_tensor):
        """[N, R] AWQ order -> [R, N] natural columns (reference :121-140)."""
        t = int_tensor.T.contiguous()
        idx = _awq_col_index(t.shape[1], t.device)
        out = torch.empty_like(t)
        out[:, idx] = t
        return out

    def pack_qzeros(self, qzeros, device):
        qzeros = self.reorder_int_tensor(qzeros).T.contiguous()  # [G, N] in AWQ order
        assert max(1, qzeros.shape[1] // 32 * self.bits) == int(round(qzeros.shape[1] * self.bits / 32 + 0.5))
        super().pack_qzeros(qzeros, device)

    def unpack_qzeros(self, device):
        zeros = super().unpack_qzeros(device)  # [G, N] AWQ order
        return self.reverse_reorder_int_tensor(zeros.T.contiguous())

    def unpack_qweight(self, device):
        self._real_buffers()
        qweight = self.qweight.to(device)
        if qweight.is_cuda:
            from ... import ops
            return ops.unpack_qweight(qweight.contiguous(), "GEMM", 4, self.infeatures, self.outfeatures)
        w = unpack_bitstream(qweight, self.bits, self.outfeatures, axis=1)  # [K, N] AWQ order
        return self.reverse_reorder_int_tensor(w.T.contiguous())

    # ---- decode: the native copy ----------------------------------------------------------------------------------------------
    # An AWQ row is only N/2 bytes, so no column strip of this layout covers all of K with whole cache lines; at decode sizes that
    # forces a split-K reduction whose extra DRAM round trips dominate the launch.  The mixin's native_descriptor()
    # therefore matters most here: qllm_repack_native reads the AWQ words and zero points directly (nibble interleave undone on
    # the fly), no intermediate copy.  QLLM_NATIVE_LAYOUT=0: every call on the in-place layout (split-K decode kernel).

    def forward(self, x):
        return self._hip_linear(x, None, 0)

    def extra_repr(self) -> str:
        return "infeatures={}, outfeatures={}, bias={}, w_bit={}, group_size={}".format(
            self.infeatures, self.outfeatures, self.bias is not None, self.w_bit, self.group_size)
