"""Shared forward plumbing of the three q_layers: build (and cache) the C descriptor of the module's buffers,
call the fused HIP path, fall back to the library's dequant kernel + a plain GEMM only for configurations the fused
kernels do not cover (the reference's own branch (B), quant_linear_gptq.py:81-85).  Never computes on the CPU."""
from __future__ import annotations

import os

import torch

from ... import ops


def tensor_version(t) -> int:
    """`t._version`, or 0 for inference tensors (created under torch.inference_mode(): they do not track a version counter and
    reading it raises; they are also immutable outside inference mode, so identity alone is a sound key for them)."""
    return 0 if t.is_inference() else t._version


def _tkey(t):
    """Cache key of one buffer: storage address AND in-place version (load_state_dict / .copy_ bump `_version`)."""
    return (0, 0) if t is None else (t.data_ptr(), tensor_version(t))


# derived, pointer-holding state (ctypes descriptors, shadows, sibling groups): rebuilt on demand, never copied or pickled
_DERIVED = ("_desc", "_desc_key", "_desc_keep", "_native", "_native_key", "_ao", "_ao_key", "_perm", "_siblings")
# (a released module is materialised before it is pickled / deep-copied: see __getstate__)


class HipForwardMixin:
    _desc = None
    _desc_key = None
    _desc_keep = None
    _siblings = None  # SiblingGroup (fused.py) when the layer shares its input with q/k/v or gate/up siblings

    def __getstate__(self):
        # nn.Module pickles / deep-copies its __dict__: drop the caches that hold raw device pointers into THIS module's
        # buffers (a copy must build its own), like the reference's modules, which carry no such state.  nn.Module defines
        # __getstate__ itself and comes first in the q_layers' MRO, so each q_layer class re-exports this one explicitly.
        self.materialize_reference()
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
        self.materialize_reference()
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

    # ---- the library's native (strip-major) layout of this layer --------------------------------------------------------
    _native = None
    _native_key = None
    # Memory policy.  The native copy holds exactly the information of qweight / qzeros / scales (pure permutations), so once it
    # exists the reference buffers are redundant on the device: with `release_reference` set they are replaced by empty
    # placeholders (the layer then costs 1.0x its packed bytes of HBM, not 2.0x) and regenerated bit-exactly on demand --
    # state_dict() / save_pretrained, load_state_dict, .to(), unpack(), or a call no native kernel serves.  The loader
    # (modeling/base.load_quantized) and bench.py switch it on; QLLM_RELEASE_REFERENCE=1 makes it the default for every module.
    release_reference = os.environ.get("QLLM_RELEASE_REFERENCE", "0") == "1"
    _released = None         # dict(name -> (shape, dtype)) of the buffers currently released, or None
    _needs_reference = False  # a call had to fall back to the reference buffers: never release this layer again
    _RELEASABLE = ("qweight", "qzeros", "scales")

    def native_descriptor(self, add_zero_bias: int = 0):
        """This layer in the library's strip-major native layout (include/qllm_mi355x.h): built once on the device from the
        module's own buffers by qllm_repack_native -- a pure integer permutation, the state dict is untouched -- and cached on the
        buffers' identity and version.  The kernels stream it 15-25 % faster at decode sizes than the reference layouts (every
        workgroup reads ONE contiguous region).  None when the layer cannot be held in that layout (bits not 3 / 4, odd shapes,
        act-order: see QuantLinearGPTQ) or QLLM_NATIVE_LAYOUT=0: callers then stream the reference buffers in place."""
        if os.environ.get("QLLM_NATIVE_LAYOUT", "1") == "0":
            self.materialize_reference()
            return None
        if self._released is None:
            key = (_tkey(self.qweight), _tkey(self.scales), _tkey(getattr(self, "qzeros", None)), _tkey(self.bias))
            if self._native_key != key:
                self._native, self._native_key = None, key
                src = self._native_source()
                if src is not None:
                    try:
                        self._native = ops.repack_native(*src)
                    except ops.QllmUnsupported:
                        self._native = None
            if self._native and self.release_reference and not self._needs_reference:
                self._release_reference()
        if not self._native:
            return None
        w = self._native[0]
        if w.add_zero_bias != add_zero_bias:  # (the AutoGPTQ offset is a field of the descriptor, not of the stored zero points)
            w = ops.QllmWeight(w.qweight, w.scales, w.qzeros, w.g_idx, w.bias, w.K, w.N, w.group_size, w.bits, w.layout, int(add_zero_bias))
        return w

    def _native_source(self):
        """(descriptor, keepalive) of the reference-layout buffers the native copy is made from; None = no native copy."""
        self._descriptor(None, 0)
        return self._desc, self._desc_keep

    def _release_reference(self):
        """Replace the buffers the native copy duplicates by empty placeholders (same dtype / device).  fp16 modules only: a bf16
        module's scales were rounded to fp16 for the kernels and could not be regenerated."""
        if self.scales.dtype != torch.float16 or (self._layout_name() == "HQQ" and self.qzeros.dtype != torch.float16):
            return
        rel = {}
        for name in self._RELEASABLE:
            t = getattr(self, name, None)
            if t is None or not t.is_cuda:
                return
            rel[name] = (tuple(t.shape), t.dtype)
        if not rel:   # nothing to release (QuantLinearORT keeps its blob): stay keyed on the buffers
            return
        self._desc = self._desc_key = self._desc_keep = None   # they point into the buffers that go away
        for name in rel:
            setattr(self, name, torch.empty(0, dtype=rel[name][1], device=self.qweight.device))
        self._released = rel

    def _regenerate_reference(self):
        """(qweight, scales, qzeros) of the reference layout, bit-exact, from the native copy."""
        return ops.unpack_native(self._native[0], self._native[1], self._layout_name())

    def materialize_reference(self):
        """Make qweight / qzeros / scales real again (no-op unless they are released).  Afterwards the native copy is still
        valid and keyed on the regenerated tensors; the next decode releases them again unless `_needs_reference` was set."""
        if self._released is None:
            return
        rel, self._released = self._released, None
        qweight, scales, qzeros = self._regenerate_reference()
        for name, t in (("qweight", qweight), ("scales", scales), ("qzeros", qzeros)):
            if name in rel and t is not None:
                setattr(self, name, t.reshape(rel[name][0]).to(rel[name][1]))
        self._native_key = (_tkey(self.qweight), _tkey(self.scales), _tkey(getattr(self, "qzeros", None)), _tkey(self.bias))

    # nn.Module comes first in the q_layers' MRO, so each q_layer class re-exports these three explicitly (like __getstate__)
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.materialize_reference()
        return torch.nn.Module._save_to_state_dict(self, destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self.materialize_reference()   # the incoming tensors are copied INTO the buffers: they must have their shapes
        return torch.nn.Module._load_from_state_dict(self, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.materialize_reference()   # .to() / .cuda() / .half(): move real data, the derived copies are rebuilt where it lands
        return torch.nn.Module._apply(self, fn, *args, **kwargs)

    def decode_descriptor(self, act_order_g_idx=None, add_zero_bias: int = 0):
        """Descriptor the fused kernels should stream: the native copy when there is one, else the module's own buffers in place."""
        if act_order_g_idx is None:
            w = self.native_descriptor(add_zero_bias)
            if w is not None:
                return w
        return self._descriptor(act_order_g_idx, add_zero_bias)

    def forward_into(self, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """y = forward(x) written into the caller's contiguous [M, out_features] tensor `out` (no allocation; used by the
        tensor-parallel wrappers, whose shard kernels write straight into their slice of the gathered output).  Layers without a
        plain fused path for the call (act-order, unsupported shapes) compute normally and copy."""
        x2d = x.reshape(-1, x.shape[-1])
        resolve = getattr(self, "_resolve_act_order", None)   # (GPTQ detects act-order lazily: do it before branching on it)
        if resolve is not None:
            resolve()
        if getattr(self, "act_order", None) or not x2d.is_contiguous() or not out.is_contiguous():
            out.copy_(self(x).reshape(out.shape))
            return out
        azb = autogptq_compat() if self._layout_name() == "GPTQ" else 0
        try:
            ops.linear_forward(self.decode_descriptor(None, azb), x2d, out=out.view(x2d.shape[0], self.outfeatures))
        except ops.QllmUnsupported:
            out.copy_(self(x).reshape(out.shape))
        return out

    def forward_allreduce_into(self, x: torch.Tensor, out: torch.Tensor, reducer) -> bool:
        """Row-parallel shard at batch 1: y = sum over ranks of forward(x_rank), in ONE launch per rank (the batch-1 kernel pushes its
        partial outputs straight into the peers' staging buffers of `reducer`, a qllm_amd.comm.OneShotAllReduce; csrc/strip1_kernel.hpp,
        AR).  False when the call is not served (rows, layout, shape, act-order): the caller then runs forward_into + the collective."""
        x2d = x.reshape(-1, x.shape[-1])
        resolve = getattr(self, "_resolve_act_order", None)
        if resolve is not None:
            resolve()
        if (x2d.shape[0] != 1 or getattr(self, "act_order", None) or not x2d.is_contiguous() or not out.is_contiguous()
                or not hasattr(reducer, "linear_all_reduce")):
            return False
        azb = autogptq_compat() if self._layout_name() == "GPTQ" else 0
        w = self.native_descriptor(azb)
        return w is not None and reducer.linear_all_reduce(w, x2d, out.view(1, self.outfeatures))

    def _hip_linear(self, x: torch.Tensor, act_order_g_idx=None, add_zero_bias: int = 0) -> torch.Tensor:
        if not x.is_cuda or not self.qweight.is_cuda:
            raise RuntimeError(
                f"{type(self).__name__}.forward needs HIP tensors on an MI355X: qllm_amd ships no CPU / eager fallback "
                f"(x on {x.device}, qweight on {self.qweight.device})")
        if self._siblings is not None and act_order_g_idx is None:
            y = self._siblings.forward_for(self, x, add_zero_bias)  # q/k/v, gate/up: one grouped launch for the group (fused.py)
            if y is not None:
                return y
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        w = self.decode_descriptor(act_order_g_idx, add_zero_bias)
        try:
            try:
                # (bf16 prefill: x is converted to fp16 ONCE per distinct tensor -- siblings share it -- instead of once per call)
                y = ops.linear_forward_shared(w, x2d, key=x) if act_order_g_idx is None else ops.linear_forward(w, x2d)
            except ops.QllmUnsupported:
                if w.layout not in (ops.LAYOUTS["NATIVE"], ops.LAYOUTS["NATIVE_F16Z"]):
                    raise
                # a shape the native layout is not served at (e.g. M > 64 with N % 128 != 0): the reference buffers in place,
                # and they stay on the device from now on
                self._needs_reference = True
                w = self._descriptor(act_order_g_idx, add_zero_bias)
                y = ops.linear_forward(w, x2d)
        except ops.QllmUnsupported:
            # e.g. 3/5/6/7/8-bit at prefill sizes: dequantise with the library kernel, then a plain library GEMM
            wt = ops.dequant(w, x.device, torch.float16)
            y = torch.matmul(x2d, wt.to(x2d.dtype))
            if self.bias is not None:
                y = y + self.bias.to(y.dtype)
        return y.reshape(x.shape[:-1] + (self.outfeatures,))


def export_module_hooks(cls):
    """Class decorator for the q_layers: nn.Module precedes HipForwardMixin in their MRO, so the mixin's overrides of nn.Module
    methods have to be re-exported on the class itself."""
    for name in ("__getstate__", "_save_to_state_dict", "_load_from_state_dict", "_apply"):
        setattr(cls, name, getattr(HipForwardMixin, name))
    return cls


def autogptq_compat() -> int:
    return int(os.environ.get("COMPATIBLE_WITH_AUTOGPTQ", "0"))
