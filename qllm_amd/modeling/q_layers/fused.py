"""Sibling groups: quantized linears that read the SAME activation tensor (q/k/v, gate/up) served by ONE grouped launch at
decode sizes, without touching the model code or the state dict.

The reference swaps one module per nn.Linear (qllm/utils/modelutils.py:161-181) and the HF model code calls
`q_proj(x)`, `k_proj(x)`, `v_proj(x)` one after the other.  Here the modules stay in place (same names, same buffers); they
additionally share a `SiblingGroup`.  The first sibling called with a tensor `x` launches the grouped kernel for ALL
siblings (`qllm_linear_forward_grouped`: one launch, x read once, 2-3x more bytes in flight per launch) and parks the others'
outputs; when the model then calls the next sibling with the very same tensor object (and the tensor was not modified in
between: `x._version`), it gets the parked output.  Anything else -- another tensor, a larger M, act-order, a shape the
grouped kernel does not take -- falls back to the module's own single launch, so results never depend on call patterns.
"""
from __future__ import annotations

import os
import weakref
from typing import Optional, Sequence

import torch

from ... import ops
from ._hip_forward import tensor_version

GROUP_MAX_M = 128  # the grouped entry point serves decode and mid-batch sizes (strips to 32 rows, the panel kernel to 128)

# attribute names of siblings inside one parent module (Llama / Mistral / Qwen2, OPT, Falcon-style MLPs, ...)
SIBLING_PATTERNS = (("q_proj", "k_proj", "v_proj"), ("gate_proj", "up_proj"), ("w1", "w3"))


class SiblingGroup:
    def __init__(self, layers: Sequence[torch.nn.Module]):
        self.layers = list(layers)
        self.enabled = True
        self._x = None       # weak reference to the tensor the parked outputs were computed from (never keeps activations alive)
        self._ver = -1
        self._azb = 0
        self._parked: dict = {}
        self._misses = 0     # consecutive grouped launches whose parked outputs nobody collected (see forward_for)
        self._refused_from = GROUP_MAX_M + 1  # smallest row count the grouped entry point refused for this group (see forward_for)
        self._prefill_refused_upto = GROUP_MAX_M  # largest prefill-sized row count (> GROUP_MAX_M) the grouped entry point refused
        self.grouped_launches = 0  # diagnostics / tests

    def describe(self, m: int = 1) -> str:
        return ops.plan_describe([l.decode_descriptor() for l in self.layers], m)

    def compatible(self) -> bool:
        a = self.layers[0]
        for l in self.layers:
            if type(l) is not type(a) or l.infeatures != a.infeatures or l.bits != a.bits or l.groupsize != a.groupsize:
                return False
        # act-order (GPTQ): the siblings' native copies hold their rows sorted by group and are fed x[:, perm] -- q/k/v (gate/up) of a real
        # checkpoint carry the SAME permutation (it comes from the shared input's Hessian, qllm/quantization/gptq/gptq.py:168; equal
        # permutations are interned to one tensor), so the group is served with the one gathered x its callers share.  Different
        # permutations, or a layer without a row-sorted native copy: every layer on its own.
        ao = [bool(getattr(l, "act_order", None)) for l in self.layers]
        if any(ao):
            perm = getattr(a, "_perm", None)
            if not all(ao) or perm is None or any(getattr(l, "_perm", None) is not perm for l in self.layers):
                return False
        return True

    def forward_for(self, layer, x: torch.Tensor, add_zero_bias: int = 0) -> Optional[torch.Tensor]:
        """Output of `layer` for x, or None when the caller must run its own launch.  `add_zero_bias`: the caller's
        COMPATIBLE_WITH_AUTOGPTQ value for this forward (the reference reads it per call, quant_linear_gptq.py:75): it is part of
        the descriptors the group launches with and of the key the parked outputs are matched on."""
        if not self.enabled:
            return None
        key = id(layer)
        held = self._x() if self._x is not None else None
        if held is x and tensor_version(x) == self._ver and add_zero_bias == self._azb and key in self._parked:
            out = self._parked.pop(key)
            self._misses = 0
            if not self._parked:
                self._x = None
            return out
        if self._parked:
            # the siblings' outputs of the previous launch were never asked for (the caller hands every sibling a NEW tensor
            # object, e.g. through accelerate hooks): grouping then multiplies the work instead of saving launches
            self._misses += 1
            if self._misses >= 3:
                self.enabled = False
        self._x, self._parked = None, {}
        if not self.enabled:
            return None
        x2d = x.reshape(-1, x.shape[-1])
        m = x2d.shape[0]
        for l in self.layers:   # (act-order siblings: their row-sorted native copies, and with them the permutations, are built lazily)
            if getattr(l, "act_order", None) and getattr(l, "_perm", None) is None:
                l.native_descriptor(add_zero_bias)
        # Round 6: prefill-sized calls too -- ONE launch of the 256x128 kernel carries the tiles of all siblings (csrc/gemm3.hip, grouped
        # form: from 384 rows, at least one tile per CU).  A refusal is remembered as "nothing up to this many rows" (the condition is
        # monotone in the row count), separately from the decode / mid-batch range.
        prefill = m > GROUP_MAX_M
        if prefill and (m <= self._prefill_refused_upto or os.environ.get("QLLM_FUSE_PREFILL", "1") == "0"):
            return None
        if (not prefill and m >= self._refused_from) or m == 0 or not x2d.is_contiguous() or not self.compatible():
            return None
        try:
            descs = []
            for l in self.layers:
                if getattr(l, "act_order", None):   # only the row-sorted native copy matches the gathered x this call was given
                    d = l.native_descriptor(add_zero_bias)
                    if d is None:
                        return None
                else:
                    d = l.decode_descriptor(None, add_zero_bias)
                descs.append(d)
            outs = ops.linear_forward_grouped(descs, x2d)
        except ops.QllmUnsupported:
            # no grouped kernel for this many rows (wide groups above 32 rows: the layers run one by one, panel.hip): do not ask
            # again from here up -- but keep grouping the smaller batches.  (Round 4: this used to switch the group off for good,
            # so one 40-token prefill cost every later decode step its grouped launches.)  Refused at one row: nothing to keep.
            if prefill:
                self._prefill_refused_upto = max(self._prefill_refused_upto, m)
                return None
            self._refused_from = m
            if m == 1:
                self.enabled = False
            return None
        self.grouped_launches += 1
        shape = x.shape[:-1]
        mine = None
        for l, o in zip(self.layers, outs):
            o = o.reshape(shape + (l.outfeatures,))
            if l is layer:
                mine = o
            else:
                self._parked[id(l)] = o
        self._x, self._ver, self._azb = weakref.ref(x), tensor_version(x), add_zero_bias
        return mine


def fuse_siblings(layers: Sequence[torch.nn.Module]) -> SiblingGroup:
    """Make `layers` (q_layers with equal in_features / bits / groupsize) one sibling group."""
    g = SiblingGroup(layers)
    for l in layers:
        l._siblings = g
    return g


def _group_key(l):
    """What the layers of one grouped launch must agree on (qllm_linear_forward_grouped: K, bits, group size, layout family)."""
    return (type(l), l.infeatures, l.bits, l.groupsize)


def install_sibling_groups(model: torch.nn.Module, q_layer_types) -> int:
    """Group the q_layers of every parent module that match one of SIBLING_PATTERNS.  Returns the number of groups.
    Mixed-precision checkpoints (the reference's quant_config_by_layer.json, qllm/utils/modelutils.py:167-179, gives every layer
    its own bits / groupsize): siblings that disagree are partitioned -- q/k at 3 bits and v at 4 become ONE grouped launch for
    q/k and v's own launch, instead of three single launches (round 6; until then such a parent got no group at all).
    QLLM_FUSE_SIBLINGS=0 disables it."""
    if os.environ.get("QLLM_FUSE_SIBLINGS", "1") == "0":
        return 0
    n = 0
    types = tuple(q_layer_types)
    for parent in model.modules():
        for names in SIBLING_PATTERNS:
            subs = [getattr(parent, nm, None) for nm in names]
            if not all(isinstance(s, types) for s in subs):
                continue
            parts: dict = {}
            for s in subs:
                parts.setdefault(_group_key(s), []).append(s)
            for members in parts.values():
                if len(members) < 2:
                    continue
                g = SiblingGroup(members)
                if g.compatible():
                    for s in members:
                        s._siblings = g
                    n += 1
    return n
