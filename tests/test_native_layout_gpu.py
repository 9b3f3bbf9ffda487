"""-m gpu: the library's native (strip-major) layout.  qllm_repack_native against the numpy restatement of the layout
(oracle/ref_cpu.native_layout, from the integer grids the oracle recovers from the reference-minted goldens), unpack(repack(x)) == x
bit for bit for every source layout, and the decode kernels on native descriptors against the oracle (every zero-point kind, 3 and
4 bits, g64 / g128 and -- 4 bits -- g32, bias, AutoGPTQ offset, M = 1 .. 64, grouped launches)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NATIVE_GOLDENS = ["gptq_w4_g128_asym", "gptq_w4_g128_sym", "gptq_w4_g128_autogptq", "gptq_w4_g128_opt_bias", "gptq_w3_g128_asym",
                  "awq_w4_g128_asym", "awq_w4_g64_bias", "hqq_w4_g64", "hqq_w3_g64"]


def _ints(g):
    """(q[K, N], stored integer / fp16 zero points [G, N], zeros_f16) of a golden, through the oracle's unpackers."""
    lay, bits, K, N = g["layout"], g["bits"], g["K"], g["N"]
    if lay == "GEMM":
        return O.awq_int_weight(g["qweight"], K, N), O.awq_int_zeros(g["qzeros"], N), False
    q = O.gptq_int_weight(g["qweight"], bits, K)
    if lay == "HQQ":
        return q, g["qzeros"], True
    return q, O.gptq_int_zeros(g["qzeros"], bits, N, 0), False


@pytest.mark.parametrize("name", NATIVE_GOLDENS)
def test_repack_matches_the_layout_definition_and_round_trips(name):
    from qllm_amd import ops
    g = load_golden(name)
    layer = to_layer(dict(g), DEV)
    layer._descriptor(None, 0)
    src, keep = layer._desc, layer._desc_keep
    nat, nkeep = ops.repack_native(src, keep)
    q, z, zf = _ints(g)
    want_q, want_s, want_z = O.native_layout(q, g["scales"], z, g["bits"], zf)
    assert np.array_equal(nkeep[0].cpu().numpy().reshape(want_q.shape), want_q.view(np.int32))
    assert np.array_equal(nkeep[1].cpu().numpy().reshape(want_s.shape).view(np.uint16), want_s.view(np.uint16))
    got_z = nkeep[2].cpu().numpy().reshape(want_z.shape)
    assert np.array_equal(got_z.view(np.uint16) if zf else got_z, want_z.view(np.uint16) if zf else want_z)
    # back to the reference's buffers: bit-identical to what was loaded
    back = ops.unpack_native(nat, nkeep, g["layout"])
    assert torch.equal(back[0], layer.qweight.reshape(back[0].shape))
    assert torch.equal(back[1].view(torch.int16), layer.scales.to(torch.float16).view(torch.int16))
    assert torch.equal(back[2].view(torch.int16) if zf else back[2], layer.qzeros.view(torch.int16) if zf else layer.qzeros)
    # ... and GPTQ <-> AWQ through the native layout is the integer-exact repack
    if g["layout"] in ("GPTQ", "GEMM") and g["bits"] == 4:
        other = "GEMM" if g["layout"] == "GPTQ" else "GPTQ"
        oq, os_, oz = ops.unpack_native(nat, nkeep, other)
        if other == "GEMM":
            assert np.array_equal(O.awq_int_weight(oq.cpu().numpy(), g["K"], g["N"]), q)
            assert np.array_equal(O.awq_int_zeros(oz.cpu().numpy(), g["N"]), z)
        else:
            assert np.array_equal(O.gptq_int_weight(oq.cpu().numpy(), 4, g["K"]), q)
            assert np.array_equal(O.gptq_int_zeros(oz.cpu().numpy(), 4, g["N"], 0), z)


@pytest.mark.parametrize("layout,g,K,N", [("GPTQ", 32, 1024, 512), ("GEMM", 32, 2048, 1152), ("HQQ", 32, 1024, 1024), ("GPTQ", 256, 1024, 512)])
def test_repack_of_other_group_sizes_matches_the_layout_definition(layout, g, K, N):
    """The goldens stop at g64 / g128; the layout is generic in the group size (any multiple of 32): 32-wide groups (served by the
    strip kernels since round 4) and 256-wide ones (held natively, decoded by the prefill kernels only) on synthetic layers."""
    from qllm_amd import ops
    d = synth(layout, 4, g, K, N, seed=g + K)
    layer = to_layer(d, DEV)
    layer._descriptor(None, 0)
    nat, nkeep = ops.repack_native(layer._desc, layer._desc_keep)
    q, z, zf = _ints(dict(d, layout=layout, bits=4))
    want_q, want_s, want_z = O.native_layout(q, d["scales"], z, 4, zf)
    assert np.array_equal(nkeep[0].cpu().numpy().reshape(want_q.shape), want_q.view(np.int32))
    assert np.array_equal(nkeep[1].cpu().numpy().reshape(want_s.shape).view(np.uint16), want_s.view(np.uint16))
    got_z = nkeep[2].cpu().numpy().reshape(want_z.shape)
    assert np.array_equal(got_z.view(np.uint16) if zf else got_z, want_z.view(np.uint16) if zf else want_z)
    back = ops.unpack_native(nat, nkeep, layout)
    assert torch.equal(back[0], layer.qweight.reshape(back[0].shape))
    assert torch.equal(back[1].view(torch.int16), layer.scales.view(torch.int16))
    assert torch.equal(back[2].view(torch.int16) if zf else back[2], layer.qzeros.view(torch.int16) if zf else layer.qzeros)


CASES = [  # layout, bits, g, K, N, zero kind, bias
    ("GPTQ", 4, 128, 4096, 4096, "asym", False), ("GPTQ", 4, 128, 11008, 4096, "asym", True), ("GEMM", 4, 128, 4096, 11008, "asym", False),
    ("GPTQ", 4, 128, 4096, 1024, "sym", True), ("HQQ", 4, 64, 4096, 4096, "asym", False), ("HQQ", 3, 64, 4096, 4096, "asym", True),
    ("GPTQ", 3, 128, 4096, 4096, "asym", False), ("GPTQ", 4, 64, 2048, 1152, "asym", False), ("GEMM", 4, 64, 1024, 512, "asym", True),
    ("GPTQ", 4, 128, 8192, 1024, "asym", False), ("HQQ", 4, 64, 11008, 4096, "asym", False), ("GPTQ", 4, 128, 1024, 8192, "asym", False),
    # 32-wide groups (round 4): one-round lds-slab blocks at batch 1 (K = 4096: 16 waves, K = 1024: 4; other K: register-A), the DMA form with
    # shorter rings at M = 2..32 (K = 2112: a 5-k-step chunk rounded to whole pairs), register-A above
    ("GPTQ", 4, 32, 4096, 4096, "asym", False), ("GPTQ", 4, 32, 11008, 4096, "asym", True), ("GEMM", 4, 32, 1024, 512, "asym", True),
    ("HQQ", 4, 32, 2048, 1152, "asym", False), ("GPTQ", 4, 32, 4096, 1024, "sym", True), ("GPTQ", 4, 32, 2112, 4096, "asym", False),
]


@pytest.mark.parametrize("layout,bits,g,K,N,zk,bias", CASES)
def test_native_decode_kernels_vs_oracle(layout, bits, g, K, N, zk, bias):
    from qllm_amd import ops
    d = synth(layout, bits, g, K, N, zk, False, bias, seed=K + N + bits)
    d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5 * 0.5).astype(np.float16)
    layer = to_layer(d, DEV)
    if zk == "sym":
        layer._descriptor(None, 0)
        src = ops.make_weight("GPTQ", layer.qweight, layer.scales, None, None, layer.bias, K, N, g, bits, 0)  # no qzeros buffer: 2^(bits-1)
        nat = ops.repack_native(*src)[0:2]
        w, keep = nat
    else:
        w = layer.native_descriptor(0)
        assert w is not None
    ref = Ref(d)
    for m in (1, 2, 4, 5, 16, 17, 32, 33, 64):
        if bits == 3 and m > 32:
            continue
        x = randx(m, K, seed=m)
        assert "layout=strip-major" in ops.plan_describe([w], m), (m, ops.plan_describe([w], m))
        y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert np.isfinite(y.astype(np.float32)).all()
        assert O.rel_err(y, ref.y16(x)) <= 1e-2, (m, ops.plan_describe([w], m))
        assert O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3, (m, ops.plan_describe([w], m))
    # bf16 activations on the same descriptor
    for mb in (3, 16, 24):
        xb = torch.from_numpy(randx(mb, K, seed=9)).to(DEV).to(torch.bfloat16)
        if not (bits == 3 and g == 64 and "register-A" in ops.plan_describe([w], mb)):
            yb = ops.linear_forward(w, xb).float().cpu().numpy()
            assert O.rel_err(yb, ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2, (mb, ops.plan_describe([w], mb))


@pytest.mark.parametrize("g", [128, 32])
def test_native_grouped_launch_and_autogptq_offset(g):
    """q/k/v-like group (unequal widths) in one launch on native descriptors; add_zero_bias = 1 (COMPATIBLE_WITH_AUTOGPTQ)."""
    from qllm_amd import ops
    ds = [synth("GPTQ", 4, g, 4096, n, seed=80 + i, bias=(i == 2)) for i, n in enumerate((4096, 1024, 512))]
    layers = [to_layer(d, DEV) for d in ds]
    for compat in (0, 1):
        ws = [l.native_descriptor(compat) for l in layers]
        assert "layout=strip-major" in ops.plan_describe(ws, 1)
        for m in (1, 4, 16):
            x = randx(m, 4096, seed=3 + m)
            outs = ops.linear_forward_grouped(ws, torch.from_numpy(x).to(DEV))
            for o, d in zip(outs, ds):
                assert O.rel_err(o.cpu().numpy(), Ref(dict(d, compat=compat)).y16(x)) <= 1e-2, (compat, m)


@pytest.mark.parametrize("layout,bits,g", [("HQQ", 4, 64), ("HQQ", 3, 64), ("GPTQ", 3, 128), ("GEMM", 4, 128), ("GPTQ", 4, 32)])
@pytest.mark.parametrize("widths", [(4096, 4096, 4096), (11008, 11008), (5152, 5152)])
def test_native_multi_strip_blocks_at_batch_16(layout, bits, g, widths):
    """M = 5..16 on wide grouped launches (BASELINE configs[3]: HQQ g64, mixed 3 / 4 bits, batch 16): blocks of several adjacent
    strips sharing one activation stream (strip_dma.hpp).  q/k/v-like: blocks of four; gate/up-like: blocks of six (4 bits; HQQ 3 bits) with a
    ragged last block per layer (688 strips = 114 x 6 + 4); (5152, 5152): 322 strips each, blocks of four, the last one of two."""
    from qllm_amd import ops
    ds = [synth(layout, bits, g, 4096, n, seed=90 + i, bias=(i == 0)) for i, n in enumerate(widths)]
    layers = [to_layer(d, DEV) for d in ds]
    ws = [l.native_descriptor(0) for l in layers]
    for m in (5, 16):
        plan = ops.plan_describe(ws, m)
        assert "form=dma-A" in plan and "layout=strip-major" in plan, plan
        if g == 32:             # (32-wide groups: blocks of two strips at most)
            assert "cpl=2" in plan, plan
        elif widths[0] == 4096:
            assert "cpl=4" in plan, plan   # (768 strips: 192 blocks of four; three per block would be 258 blocks, blocks do not span layers)
        elif widths[0] == 5152:
            assert "cpl=3" in plan or "cpl=4" in plan, plan   # (644 strips: 216 blocks of three, round 5)
        elif widths[0] == 11008:  # (3 bits: six strips only with fp16 zero points at 64-wide groups -- HQQ, round 5 -- else four at most)
            assert ("cpl=6" in plan) if (bits == 4 or layout == "HQQ") else ("cpl=4" in plan or "cpl=3" in plan or "cpl=2" in plan), plan
        x = randx(m, 4096, seed=m)
        outs = ops.linear_forward_grouped(ws, torch.from_numpy(x).to(DEV))
        for o, d in zip(outs, ds):
            ref = Ref(d)
            assert O.rel_err(o.cpu().numpy(), ref.y16(x)) <= 1e-2, (layout, bits, m, plan)
            assert O.rel_err(o.cpu().numpy().astype(np.float64), ref.y64(x)) <= 2e-3, (layout, bits, m, plan)
    xb = torch.from_numpy(randx(16, 4096, seed=7)).to(DEV).to(torch.bfloat16)
    outs = ops.linear_forward_grouped(ws, xb)
    for o, d in zip(outs, ds):
        assert O.rel_err(o.float().cpu().numpy(), Ref(d).y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2


PANEL_CASES = [  # layout, g, K, N, zero kind, bias
    ("GPTQ", 128, 4096, 4096, "asym", False), ("GPTQ", 128, 11008, 4096, "asym", True), ("GEMM", 128, 4096, 11008, "asym", False),
    ("HQQ", 64, 4096, 4096, "asym", True), ("GPTQ", 32, 2048, 1152, "asym", False), ("GPTQ", 128, 4096, 1024, "sym", True),
    ("GPTQ", 64, 2112, 4096, "asym", False), ("GEMM", 64, 1024, 512, "asym", True), ("HQQ", 64, 11008, 4096, "asym", False),
    # (round 5: up to 4096 x 4096 the strips keep 17..32 rows -- wider layers for the two-row-tile panels of every group size / zero kind)
    ("GPTQ", 32, 1024, 4224, "asym", True), ("GPTQ", 128, 1024, 4224, "sym", False),
]
PANEL_CASES_3BIT = [  # the 3-bit stream (up to 64 rows): fp16 (HQQ), packed and symmetric zero points, both group sizes, a ragged K
    ("HQQ", 64, 4096, 4096, "asym", True), ("GPTQ", 128, 4096, 1024, "asym", False), ("HQQ", 64, 11008, 4096, "asym", False),
    ("GPTQ", 64, 2112, 4096, "sym", True), ("GPTQ", 64, 1024, 512, "asym", False),
    ("GPTQ", 128, 1024, 4224, "asym", True), ("GPTQ", 64, 1024, 4224, "sym", False),
]


@pytest.mark.parametrize("bits,layout,g,K,N,zk,bias", [(4,) + c for c in PANEL_CASES] + [(3,) + c for c in PANEL_CASES_3BIT])
def test_panel_kernel_vs_oracle(bits, layout, g, K, N, zk, bias):
    """17 <= M <= 128 on native layers (from 33 rows where the layer is no larger than 4096 x 4096: the strips keep 17..32 rows there, and
    everything below 17 -- round 5, profiles/r05_batch16.md): the panel kernel (csrc/panel.hip: 64-column panels, A tiles shared through LDS, B
    fragments q - z from registers, split-K partial panels through the workspace).  Every zero-point kind, g32 / g64 / g128, a K whose
    k-steps do not fill the last K-tile or split (2112 = 66 k-steps), bias, fp16 and bf16 activations, against the oracle."""
    from qllm_amd import ops
    d = synth(layout, bits, g, K, N, zk, False, bias, seed=K + N + g + bits)
    d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5 * 0.5).astype(np.float16)
    layer = to_layer(d, DEV)
    if zk == "sym":
        layer._descriptor(None, 0)
        src = ops.make_weight("GPTQ", layer.qweight, layer.scales, None, None, layer.bias, K, N, g, bits, 0)
        w, keep = ops.repack_native(*src)[0:2]
    else:
        w = layer.native_descriptor(0)
    ref = Ref(d)
    for m in (9, 16, 17, 24, 32, 33, 48, 64, 65, 100, 128):
        strips = m < 17 or (m <= 32 and K <= 4096 and N <= 4096)   # (checked against the oracle like the panels)
        if m > 64 and (g == 32 or bits == 3 or K * N > 2 ** 25):   # (eight row tiles: 4 bits, 64- / 128-wide groups, layers of up to 2^25 weights; else
            assert ops.plan_describe([w], m).startswith("gemm"), (m, ops.plan_describe([w], m))   # the 256-row tiles take over at 65 rows)
            continue
        assert ops.plan_describe([w], m).startswith("strip " if strips else "panel "), (m, ops.plan_describe([w], m))
        x = randx(m, K, seed=m)
        y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert np.isfinite(y.astype(np.float32)).all()
        assert O.rel_err(y, ref.y16(x)) <= 1e-2, (m, ops.plan_describe([w], m))
        assert O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3, (m, ops.plan_describe([w], m))
    # repeated calls re-use the workspace's arrival counters (the last arriver re-arms them)
    x = randx(64, K, seed=5)
    xt = torch.from_numpy(x).to(DEV)
    y0 = ops.linear_forward(w, xt)
    for _ in range(3):
        assert torch.equal(ops.linear_forward(w, xt), y0)
    for mb in (40, 128 if (g != 32 and bits == 4 and K * N <= 2 ** 25) else 64):
        xb = torch.from_numpy(randx(mb, K, seed=9)).to(DEV).to(torch.bfloat16)
        yb = ops.linear_forward(w, xb).float().cpu().numpy()
        assert O.rel_err(yb, ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2, mb


def test_native_layout_is_what_the_modules_decode_from():
    """The module path: decode-sized forwards stream the native copy (plan text), prefill-sized ones the reference buffers; the
    state dict is untouched and QLLM_NATIVE_LAYOUT=0 keeps everything in place."""
    import os
    from qllm_amd import ops
    d = synth("GEMM", 4, 128, 4096, 4096, seed=5)
    layer = to_layer(d, DEV)
    before = {k: v.clone() for k, v in layer.state_dict().items()}
    x = torch.from_numpy(randx(1, 4096)).to(DEV)
    y = layer(x)
    assert "layout=strip-major" in ops.plan_describe([layer.decode_descriptor()], 1)
    assert O.rel_err(y.cpu().numpy(), Ref(d).y16(x.cpu().numpy())) <= 1e-2
    for k, v in layer.state_dict().items():
        assert torch.equal(v, before[k]), k
    os.environ["QLLM_NATIVE_LAYOUT"] = "0"
    try:
        l2 = to_layer(d, DEV)
        assert ops.plan_describe([l2.decode_descriptor()], 1).startswith("skinny")   # AWQ in place: the split-K kernel
        assert O.rel_err(l2(x).cpu().numpy(), y.cpu().numpy()) <= 2e-3
    finally:
        del os.environ["QLLM_NATIVE_LAYOUT"]
