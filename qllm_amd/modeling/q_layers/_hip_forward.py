"""Shared forward plumbing of the three q_layers: build (and cache) the C descriptor of the module's buffers,
call the fused HIP path, fall back to the library's dequant kernel + a plain GEMM only for configurations the fused
kernels do not cover (the reference's own branch (B), quant_linear_gptq.py:81-85).  Never computes on the CPU."""
from __future__ import annotations

import os

import torch

from ... import ops


class HipForwardMixin:
    _desc = None
    _desc_key = None
    _desc_keep = None

    def _layout_name(self) -> str:
        raise NotImplementedError

    def _f16(self, t):
        # bf16 modules: the kernels take fp16 scales/zeros/bias (the reference casts per call, ort_ops.cc:79-90);
        # cast once and cache alongside the descriptor.
        if t is None or t.dtype == torch.float16:
            return t
        return t.to(torch.float16)

    def _descriptor(self, act_order_g_idx, add_zero_bias: int):
        qzeros = self.qzeros
        bias = self.bias
        key = (self.qweight.data_ptr(), self.scales.data_ptr(), qzeros.data_ptr() if qzeros is not None else 0,
               bias.data_ptr() if bias is not None else 0,
               act_order_g_idx.data_ptr() if act_order_g_idx is not None else 0, add_zero_bias)
        if self._desc is None or key != self._desc_key:
            lay = self._layout_name()
            scales = self._f16(self.scales).contiguous()
            if lay == "HQQ":
                qzeros = self._f16(qzeros).contiguous()
            elif qzeros is not None:
                qzeros = qzeros.contiguous()
            g = act_order_g_idx
            if g is not None:
                g = g.to(device=self.qweight.device, dtype=torch.int32).contiguous()
            b = self._f16(bias).contiguous() if bias is not None else None
            self._desc, self._desc_keep = ops.make_weight(
                lay, self.qweight.contiguous(), scales, qzeros, g, b, self.infeatures, self.outfeatures,
                self.groupsize, self.bits, add_zero_bias)
            self._desc_key = key
        return self._desc

    def decode_descriptor(self, act_order_g_idx=None, add_zero_bias: int = 0):
        """Descriptor the decode-sized (M <= 64) kernels should stream.  Default: the module's own buffers."""
        return self._descriptor(act_order_g_idx, add_zero_bias)

    def _hip_linear(self, x: torch.Tensor, act_order_g_idx=None, add_zero_bias: int = 0) -> torch.Tensor:
        if not x.is_cuda or not self.qweight.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward needs HIP tensors on an MI355X: qllm_amd ships no CPU / eager fallback "
                f"(x on {x.device}, qweight on {self.qweight.device})")
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        if x2d.shape[0] <= 64:  # the full-K strip kernels (M <= 64) stream the row-stream view
            w = self.decode_descriptor(act_order_g_idx, add_zero_bias)
        else:
            w = self._descriptor(act_order_g_idx, add_zero_bias)
        try:
            y = ops.linear_forward(w, x2d)
        except ops.QllmUnsupported:
            # e.g. 3/5/6/7/8-bit at prefill sizes: dequantise with the library kernel, then a plain library GEMM
            wt = ops.dequant(w, x.device, torch.float16)
            y = torch.matmul(x2d, wt.to(x2d.dtype))
            if self.bias is not None:
                y = y + self.bias.to(y.dtype)
        return y.reshape(x.shape[:-1] + (self.outfeatures,))


def autogptq_compat() -> int:
    return int(os.environ.get("COMPATIBLE_WITH_AUTOGPTQ", "0"))
