"""Shared forward plumbing of the three q_layers: build (and cache) the C descriptor of the module's buffers,
call the fused HIP path, fall back to the library's dequant kernel + a plain GEMM only for configurations the fused
kernels do not cover (the reference's own branch (B), quant_linear_gptq.py:81-85).  Never computes on the CPU."""
from __future__ import annotations

import os

import torch

from ... import ops


def _tkey(t):
    """Cache key of one buffer: storage address AND in-place version (load_state_dict / .copy_ bump `_version`)."""
    return (0, 0) if t is None else (t.data_ptr(), t._version)


# derived, pointer-holding state (ctypes descriptors, shadows, sibling groups): rebuilt on demand, never copied or pickled
_DERIVED = ("_desc", "_desc_key", "_desc_keep", "_shadow", "_shadow_key", "_ao", "_ao_key", "_rs", "_rs_key", "_siblings")


class HipForwardMixin:
    _desc = None
    _desc_key = None
    _desc_keep = None
    _siblings = None  # SiblingGroup (fused.py) when the layer shares its input with q/k/v or gate/up siblings

    def __getstate__(self):
        # nn.Module pickles / deep-copies its __dict__: drop the caches that hold raw device pointers into THIS module's
        # buffers (a copy must build its own), like the reference's modules, which carry no such state.  nn.Module defines
        # __getstate__ itself and comes first in the q_layers' MRO, so each q_layer class re-exports this one explicitly.
        state = torch.nn.Module.__getstate__(self)
        for k in _DERIVED:
            state.pop(k, None)
        return state

    def _invalidate(self):
        for k in _DERIVED[:-1]:
            self.__dict__.pop(k, None)

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
        key = (_tkey(self.qweight), _tkey(self.scales), _tkey(qzeros), _tkey(bias), _tkey(act_order_g_idx), add_zero_bias)
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

    def _prefill_through_row_stream(self) -> bool:
        """Layouts whose decode view is a separate buffer (AWQ) may also serve prefill from it: the row-stream form dequantises
        with 19 VALU per 8 weights against ~34 for the 8-column AWQ words, which the prefill kernels feel (measured M = 2048:
        900 / 920 / 1012 vs 808 / 793 / 858 TFLOP/s).  Default off for modules that have no such view."""
        return False

    def _hip_linear(self, x: torch.Tensor, act_order_g_idx=None, add_zero_bias: int = 0) -> torch.Tensor:
        if not x.is_cuda or not self.qweight.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward needs HIP tensors on an MI355X: qllm_amd ships no CPU / eager fallback "
                f"(x on {x.device}, qweight on {self.qweight.device})")
        if self._siblings is not None and act_order_g_idx is None:
            y = self._siblings.forward_for(self, x)  # q/k/v, gate/up: one grouped launch for the whole group (fused.py)
            if y is not None:
                return y
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        if x2d.shape[0] <= 64 or self._prefill_through_row_stream():
            w = self.decode_descriptor(act_order_g_idx, add_zero_bias)  # row-stream view (the module's own buffers if GPTQ/HQQ)
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
