"""Host logic of the sibling groups (q_layers/fused.py) with the library call replaced by a fake (no GPU): what a refused grouped
call does to later calls, and the parked-output bookkeeping."""
import types

import torch

from qllm_amd import ops
from qllm_amd.modeling.q_layers import fused


class _Layer:
    infeatures, bits, groupsize, act_order = 64, 4, 128, None

    def __init__(self, n):
        self.outfeatures = n

    def decode_descriptor(self, g_idx=None, azb=0):
        return ("desc", self.outfeatures, azb)


def _group(monkeypatch, refuse_from):
    calls = []

    def fake_grouped(descs, x2d, outs=None):
        calls.append(x2d.shape[0])
        if x2d.shape[0] >= refuse_from:
            raise ops.QllmUnsupported(2, "no grouped kernel for this many rows")
        return [torch.full((x2d.shape[0], d[1]), float(i)) for i, d in enumerate(descs)]

    monkeypatch.setattr(fused.ops, "linear_forward_grouped", fake_grouped)
    layers = [_Layer(8), _Layer(4), _Layer(4)]
    return fused.SiblingGroup(layers), layers, calls


def test_a_refusal_is_remembered_per_row_count_and_keeps_the_group(monkeypatch):
    g, layers, calls = _group(monkeypatch, refuse_from=65)
    x1 = torch.zeros(1, 64)
    assert g.forward_for(layers[0], x1).shape == (1, 8) and calls == [1]
    assert float(g.forward_for(layers[1], x1)[0, 0]) == 1.0 and calls == [1]      # parked output of the same launch
    assert float(g.forward_for(layers[2], x1)[0, 0]) == 2.0
    x100 = torch.zeros(100, 64)
    assert g.forward_for(layers[0], x100) is None and calls == [1, 100] and g.enabled   # refused: the caller runs its own launch
    assert g.forward_for(layers[1], x100) is None and calls == [1, 100]                 # ... and is not asked again at that size
    assert g.forward_for(layers[0], torch.zeros(80, 64)) is None and calls == [1, 100, 80]   # smaller: asked once, refused too
    assert g.forward_for(layers[0], torch.zeros(90, 64)) is None and calls == [1, 100, 80]   # between the two: not asked
    x64 = torch.zeros(64, 64)
    assert g.forward_for(layers[0], x64).shape == (64, 8) and calls[-1] == 64           # below the refusals: grouped as before
    x1b = torch.zeros(1, 64)
    assert g.forward_for(layers[0], x1b) is not None and g.grouped_launches == 3
    # prefill-sized calls (round 6: one grouped grid of the 256x128 kernel): asked; a refusal is remembered as "nothing up to this many
    # rows" without touching the decode-sized grouping; QLLM_FUSE_PREFILL=0 never asks
    big = fused.GROUP_MAX_M + 1
    assert g.forward_for(layers[0], torch.zeros(big + 100, 64)) is None and calls[-1] == big + 100
    n = len(calls)
    assert g.forward_for(layers[0], torch.zeros(big, 64)) is None and len(calls) == n            # fewer rows than a refused count: not asked
    assert g.forward_for(layers[0], torch.zeros(big + 200, 64)) is None and calls[-1] == big + 200   # more rows: asked again
    assert g.forward_for(layers[0], torch.zeros(1, 64)) is not None                               # decode-sized grouping untouched
    monkeypatch.setenv("QLLM_FUSE_PREFILL", "0")
    n = len(calls)
    assert g.forward_for(layers[0], torch.zeros(4096, 64)) is None and len(calls) == n


def test_prefill_sized_calls_are_grouped_when_the_library_serves_them(monkeypatch):
    g, layers, calls = _group(monkeypatch, refuse_from=10 ** 9)
    x = torch.zeros(2048, 64)
    assert g.forward_for(layers[0], x).shape == (2048, 8) and calls == [2048]
    assert g.forward_for(layers[1], x).shape == (2048, 4) and g.forward_for(layers[2], x).shape == (2048, 4) and calls == [2048]


def test_a_refusal_at_one_row_switches_the_group_off(monkeypatch):
    g, layers, calls = _group(monkeypatch, refuse_from=1)
    assert g.forward_for(layers[0], torch.zeros(1, 64)) is None and not g.enabled
    assert g.forward_for(layers[0], torch.zeros(1, 64)) is None and calls == [1]


def test_parked_outputs_are_keyed_on_tensor_identity_version_and_offset(monkeypatch):
    g, layers, calls = _group(monkeypatch, refuse_from=1000)
    x = torch.zeros(2, 64)
    g.forward_for(layers[0], x)
    x.add_(1.0)                                              # modified in place: the parked outputs are stale
    assert g.forward_for(layers[1], x) is not None and calls == [2, 2]
    assert g.forward_for(layers[2], x, add_zero_bias=1) is not None and calls == [2, 2, 2]   # another AutoGPTQ offset: relaunched
