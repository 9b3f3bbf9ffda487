"""QuantLinearGPTQ with the reference's contract (qllm/modeling/q_layers/quant_linear_gptq.py:92-143); forward is
the fused MI355X kernel path (no W materialisation, no CPU branch)."""
from __future__ import annotations

import math
import os
import weakref

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin, export_module_hooks, _tkey, autogptq_compat, tensor_version
from .compress_weight import CompressWeight, general_pack_on_row, general_unpack_on_row


# ---- act-order: one gather of x per distinct permutation ----------------------------------------------------------------
# GPTQ derives the act-order permutation from the Hessian of the layer's INPUT (reference: qllm/quantization/gptq/gptq.py:168,
# perm = argsort(diag(H)), H accumulated from the inputs, :97-102), so layers fed by the same tensor -- q/k/v, gate/up -- carry the
# same g_idx.  Equal permutations are interned to ONE device tensor, and the most recent gather per device is kept with a
# reference to the very tensor it was made from (identity + version, like the sibling groups): the second and third sibling
# reuse it instead of gathering again.
_PERMS: dict = {}
_LAST_GATHER: dict = {}


def _intern_perm(perm: torch.Tensor) -> torch.Tensor:
    import hashlib
    key = (perm.device, perm.numel(), hashlib.sha1(perm.cpu().numpy().tobytes()).hexdigest())
    return _PERMS.setdefault(key, perm)


def _gathered(x: torch.Tensor, perm: torch.Tensor) -> torch.Tensor:
    from ... import ops
    hit = _LAST_GATHER.get(x.device)
    if hit is not None and hit[0]() is x and hit[1] == tensor_version(x) and hit[2] is perm:
        return hit[3]
    out = ops.gather_columns(x.reshape(-1, x.shape[-1]).contiguous(), perm)
    # (x is held weakly: the cache must not keep a prefill-sized activation alive; the gathered copy lives until the next gather
    #  on the device replaces it)
    _LAST_GATHER[x.device] = (weakref.ref(x), tensor_version(x), perm, out)
    return out


@export_module_hooks
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
        self.materialize_reference()
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

    # ---- act-order: the native copy holds the rows sorted by group ----------------------------------------------------------
    # With act-order every k has its own group (g_idx gather per nibble).  GPTQ assigns whole groups of `groupsize` rows, so
    # sorting the rows by group (perm = argsort(g_idx)) gives a plain contiguous-group layer: the native copy of an act-order
    # layer is built from that row-permuted arrangement of its own integers (library unpack / pack kernels; bit-exact; the
    # row-stream intermediate is dropped) and the forward feeds it x[..., perm] (qllm_gather_columns).  The fused kernels then
    # run at their no-act-order speed plus one gather of x (3- and 4-bit layers).  Groups that are not uniform, or QLLM_ACTORDER_SHADOW=0: no native
    # copy -> the in-place gather kernel on the reference buffers.
    _perm = None

    def _resolve_act_order(self) -> bool:
        """Lazy act-order detection, as the reference: a trivial g_idx has its first `groupsize` entries all zero
        (quant_linear_gptq.py:137-138).  Called by every path that branches on `act_order` (forward, forward_into, the native copy)."""
        if self.act_order is None:
            self.act_order = bool(self.g_idx[: self.groupsize].sum() != 0)
        return self.act_order

    def _native_source(self):
        if not self._resolve_act_order():
            return HipForwardMixin._native_source(self)
        self._perm = None
        bits = self.bits
        if os.environ.get("QLLM_ACTORDER_SHADOW", "1") == "0" or bits not in (3, 4) or not self.qweight.is_cuda:
            return None
        from ... import ops
        dev = self.qweight.device
        g = self.g_idx.to(dev).long()
        groups = math.ceil(self.infeatures / self.groupsize)
        counts = torch.bincount(g, minlength=groups)
        if self.infeatures % self.groupsize != 0 or counts.numel() != groups or not bool((counts == self.groupsize).all()):
            return None
        perm = torch.argsort(g, stable=True)
        q = ops.unpack_qweight(self.qweight.contiguous(), "GPTQ", bits, self.infeatures, self.outfeatures)
        qw = ops.pack_qweight(q.index_select(0, perm).contiguous(), "GPTQ", bits)
        del q
        b = self._f16(self.bias).contiguous() if self.bias is not None else None
        self._perm = _intern_perm(perm.to(torch.int32).contiguous())
        return ops.make_weight("GPTQ", qw, self._f16(self.scales).contiguous(), self.qzeros.contiguous(), None, b,
                               self.infeatures, self.outfeatures, self.groupsize, bits, 0)

    def _regenerate_reference(self):
        qweight, scales, qzeros = HipForwardMixin._regenerate_reference(self)
        if self.act_order and self._perm is not None:   # the native rows are sorted by group: undo the permutation
            from ... import ops
            q = ops.unpack_qweight(qweight, "GPTQ", self.bits, self.infeatures, self.outfeatures)
            inv = torch.empty_like(self._perm, dtype=torch.long)
            inv[self._perm.long()] = torch.arange(self._perm.numel(), device=inv.device)
            qweight = ops.pack_qweight(q.index_select(0, inv).contiguous(), "GPTQ", self.bits)
        return qweight, scales, qzeros

    def forward(self, x):
        self._resolve_act_order()
        # COMPATIBLE_WITH_AUTOGPTQ is read per forward by the reference (:75); it becomes add_zero_bias here
        azb = autogptq_compat()
        if self.act_order and x.is_cuda:
            w = self.native_descriptor(azb)
            if w is not None and self._perm is not None:
                from ... import ops
                x2d = _gathered(x, self._perm)
                try:
                    if self._siblings is not None:
                        # round 6: act-order siblings share their permutation, hence the gathered x: ONE grouped launch for them too
                        # (the group is keyed on the gathered tensor, the same object for every sibling: _gathered above)
                        y = self._siblings.forward_for(self, x2d, azb)
                        if y is not None:
                            return y.reshape(x.shape[:-1] + (self.outfeatures,))
                    return ops.linear_forward_shared(w, x2d).reshape(x.shape[:-1] + (self.outfeatures,))
                except ops.QllmUnsupported:
                    self._needs_reference = True   # a shape the native kernels do not serve: the in-place gather kernel below
        g_idx = self.g_idx if self.act_order else None
        return self._hip_linear(x, g_idx, azb)
