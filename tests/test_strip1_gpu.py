"""-m gpu: the batch-1 kernel of the native layout (csrc/strip1_kernel.hpp, round 5) against the oracle, through the C ABI: every
(waves, round) form of its shape table -- exact and with a shifted last window -- packed / fp16 / symmetric zero points, bias, the
AutoGPTQ offset, bf16 activations, and grouped launches whose layers differ in width (the layer is blockIdx.y; surplus blocks leave)."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# K -> the plan line's prefix (strip1.hip: strip1_shape)
FORMS = [(256, "nw=4 round=8 grid"), (1024, "nw=4 round=8 exact"), (1152, "nw=4 round=16 grid"), (2048, "nw=4 round=16 exact"),
         (3584, "nw=7 round=16 exact"), (3840, "nw=8 round=16 grid"), (4096, "nw=8 round=16 exact"), (5120, "nw=8 round=24 grid"), (6144, "nw=8 round=24 exact"),
         (8192, "nw=8 round=32 exact"), (6656, "nw=8 round=32 grid"), (11008, "nw=15 round=24 grid"), (11776, "nw=16 round=24 grid"),
         (12288, "nw=16 round=24 exact"),
         (14336, "nw=16 round=32 grid"), (16384, "nw=16 round=32 exact"),
         # round 6: rounds of 40 .. 64 k-steps, three / four accumulator sets (Qwen2-7B's down_proj is K = 18944, Llama-2-70B's 28672)
         (18944, "nw=16 round=40 grid"), (20480, "nw=16 round=40 exact"), (22016, "nw=16 round=48 grid"), (24576, "nw=16 round=48 exact"),
         (26624, "nw=16 round=56 grid"), (28672, "nw=16 round=56 exact"), (32768, "nw=16 round=64 exact")]


def _native(layer, d, zk, compat=0):
    from qllm_amd import ops
    if zk == "sym":
        layer._descriptor(None, 0)
        src = ops.make_weight("GPTQ", layer.qweight, layer.scales, None, None, layer.bias, d["K"], d["N"], d["groupsize"], d["bits"], compat)
        return ops.repack_native(*src)[0:2]
    return layer.native_descriptor(compat), None


@pytest.mark.parametrize("K,form", FORMS)
def test_every_form_vs_oracle(K, form):
    from qllm_amd import ops
    N = 1024 if K > 8192 else 8192   # (more strips than CUs: the wide-launch forms; one block per CU at most is the test below)
    for layout, zk, bias in (("GPTQ", "asym", False), ("HQQ", "asym", True), ("GPTQ", "sym", K % 1024 == 0)):
        d = synth(layout, 4, 128, K, N, zk, False, bias, seed=K + len(layout))
        d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5 * 0.5).astype(np.float16)
        layer = to_layer(d, DEV)
        w, keep = _native(layer, d, zk)
        assert ops.plan_describe([w], 1).startswith("strip1 " + form), ops.plan_describe([w], 1)
        ref = Ref(d)
        for seed in (1, 2):
            x = randx(1, K, seed=seed)
            y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
            assert np.isfinite(y.astype(np.float32)).all()
            assert O.rel_err(y, ref.y16(x)) <= 1e-2, (layout, zk)
            assert O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3, (layout, zk)
        xb = torch.from_numpy(randx(1, K, seed=9)).to(DEV).to(torch.bfloat16)
        yb = ops.linear_forward(w, xb)
        assert yb.dtype == torch.bfloat16
        assert O.rel_err(yb.float().cpu().numpy(), ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2, (layout, zk)


def test_one_block_per_cu_form():
    """K <= 4096 with at most one 16-column strip per CU: four waves x 32 k-steps (o_proj)."""
    from qllm_amd import ops
    for K, N in ((4096, 4096), (3072, 1024), (2176, 2048)):
        d = synth("GPTQ", 4, 128, K, N, "asym", False, True, seed=K + N)
        layer = to_layer(d, DEV)
        w = layer.native_descriptor(0)
        assert ops.plan_describe([w], 1).startswith("strip1 nw=4 round=32"), ops.plan_describe([w], 1)
        ref = Ref(d)
        x = randx(1, K, seed=4)
        y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert O.rel_err(y, ref.y16(x)) <= 1e-2 and O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3


@pytest.mark.parametrize("K,widths", [(4096, (4096, 1024, 512)), (8192, (1024, 128, 128)), (11008, (512, 2048)), (4096, (11008, 11008)),
                                      (1024, (16, 4096, 16, 48, 2048, 16, 16, 32))])
def test_grouped_launches_of_unequal_widths(K, widths):
    """q/k/v-like groups: one launch, the layer is blockIdx.y; add_zero_bias = 1 (COMPATIBLE_WITH_AUTOGPTQ); equal to layer-by-layer calls."""
    from qllm_amd import ops
    ds = [synth("GPTQ", 4, 128, K, n, seed=80 + i, bias=(i % 3 == 2)) for i, n in enumerate(widths)]
    layers = [to_layer(d, DEV) for d in ds]
    for compat in (0, 1):
        ws = [l.native_descriptor(compat) for l in layers]
        assert ops.plan_describe(ws, 1).startswith("strip1 ") and "x %d" % len(ws) in ops.plan_describe(ws, 1)
        x = randx(1, K, seed=3 + compat)
        xt = torch.from_numpy(x).to(DEV)
        outs = ops.linear_forward_grouped(ws, xt)
        for o, d, w in zip(outs, ds, ws):
            assert O.rel_err(o.cpu().numpy(), Ref(dict(d, compat=compat)).y16(x)) <= 1e-2, compat
            assert torch.equal(o, ops.linear_forward(w, xt))


def test_properties_at_full_size():
    """Size-independent properties on the Llama-2-7B shapes: x -> 2 x doubles y exactly (power-of-two scaling), zero activations
    give the bias, a one-hot x returns one dequantised row of W (the unrounded s (q - z), within fp16 rounding of the output), and
    replays are bit-identical."""
    from qllm_amd import ops
    for K, N in ((4096, 4096), (11008, 4096), (4096, 11008)):
        d = synth("GPTQ", 4, 128, K, N, "asym", False, True, seed=K + N)
        layer = to_layer(d, DEV)
        w = layer.native_descriptor(0)
        x = torch.from_numpy(randx(1, K, seed=5) * np.float16(0.25)).to(DEV)
        y1, y2 = ops.linear_forward(w, x), ops.linear_forward(w, x * 2)
        b = layer.bias.float()
        assert torch.equal(y1, ops.linear_forward(w, x))
        assert torch.allclose((y2.float() - b) / 2, y1.float() - b, rtol=0, atol=2e-3 * float(y1.abs().max()))
        assert torch.equal(ops.linear_forward(w, torch.zeros_like(x)), layer.bias.view(1, -1))
        k = K - 77
        e = torch.zeros_like(x)
        e[0, k] = 1.0
        row = ops.linear_forward(w, e).float().cpu().numpy()[0]
        want = Ref(d).w[k].astype(np.float32) + d["bias"].astype(np.float32)
        assert np.abs(row - want).max() <= 2e-3 * np.abs(want).max()


G64_FORMS = [(256, "nw=4 round=8 g64"), (1024, "nw=4 round=8 exact g64"), (2048, "nw=4 round=16 exact g64"), (3584, "nw=7 round=16 exact g64"),
             (4096, "nw=8 round=16 exact g64"), (5120, "nw=8 round=24 g64"), (8192, "nw=8 round=32 exact g64"), (11008, "nw=15 round=24 g64"),
             (14336, "nw=16 round=32 g64"), (18944, "nw=16 round=40 g64"), (24576, "nw=16 round=48 exact g64")]


@pytest.mark.parametrize("K,form", G64_FORMS)
def test_every_form_with_64_wide_groups(K, form):
    """Round 6: 64-wide groups on the batch-1 kernel (G64: two A rows per group, two groups per lane and pass) -- HQQ fp16 zero points,
    GPTQ packed zero points with bias, symmetric; fp16 and bf16; against the oracle and float64 of the reference's W."""
    from qllm_amd import ops
    N = 1024 if K > 8192 else 8192
    for layout, zk, bias in (("HQQ", "asym", False), ("GPTQ", "asym", True), ("GPTQ", "sym", False)):
        d = synth(layout, 4, 64, K, N, zk, False, bias, seed=K + len(layout) + 64)
        d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5 * 0.5).astype(np.float16)
        layer = to_layer(d, DEV)
        w, keep = _native(layer, d, zk)
        assert ops.plan_describe([w], 1).startswith("strip1 " + form), ops.plan_describe([w], 1)
        ref = Ref(d)
        for seed in (1, 2):
            x = randx(1, K, seed=seed)
            y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
            assert O.rel_err(y, ref.y16(x)) <= 1e-2, (layout, zk)
            assert O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3, (layout, zk)
        xb = torch.from_numpy(randx(1, K, seed=9)).to(DEV).to(torch.bfloat16)
        yb = ops.linear_forward(w, xb)
        assert yb.dtype == torch.bfloat16
        assert O.rel_err(yb.float().cpu().numpy(), ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2, (layout, zk)
        # the general strip kernel on the same descriptor (QLLM_STRIP1 = 2: 128-wide groups only): same contract, close results
        try:
            ops.set_knob("QLLM_STRIP1", 2)
            assert ops.plan_describe([w], 1).startswith("strip nw=")
            x = torch.from_numpy(randx(1, K, seed=1)).to(DEV)
            y_gen = ops.linear_forward(w, x)
        finally:
            ops.reset_knobs()
        assert O.rel_err(ops.linear_forward(w, x).cpu().numpy(), y_gen.cpu().numpy()) <= 1e-3


def test_grouped_launch_with_64_wide_groups():
    """q/k/v of an HQQ g64 layer stack at batch 1: one grouped launch of the G64 form, unequal widths."""
    from qllm_amd import ops
    K = 4096
    ds = [synth("HQQ", 4, 64, K, n, "asym", False, i == 2, seed=70 + i) for i, n in enumerate((4096, 1024, 1024))]
    layers = [to_layer(d, DEV) for d in ds]
    descs = [l.native_descriptor(0) for l in layers]
    assert ops.plan_describe(descs, 1) == "strip1 nw=8 round=16 exact g64 grid=strips x 3 layout=strip-major"
    x = randx(1, K, seed=3)
    outs = ops.linear_forward_grouped(descs, torch.from_numpy(x).to(DEV))
    for o, d in zip(outs, ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d).y16(x)) <= 1e-2
        assert O.rel_err(o.cpu().numpy().astype(np.float64), Ref(d).y64(x)) <= 2e-3


@pytest.mark.parametrize("K,N", [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 1024), (3584, 8192), (16384, 1024), (1152, 2048)])
def test_batches_2_to_4_on_the_four_row_forms(K, N):
    """Round 6: with 128-wide groups the four A rows of a group carry four BATCH rows (rows past M: zeros): batches 2..4 through the
    batch-1 kernel's weight stream.  Against the oracle, float64 of the reference's W, strip_dma on the same descriptor, and row by row
    against the batch-1 launch (same arithmetic per row: bit-equal)."""
    from qllm_amd import ops
    for layout, zk, bias in (("GPTQ", "asym", True), ("HQQ", "asym", False)):
        d = synth(layout, 4, 128, K, N, zk, False, bias, seed=K + N + len(layout))
        d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5 * 0.5).astype(np.float16)
        layer = to_layer(d, DEV)
        w = layer.native_descriptor(0)
        ref = Ref(d)
        for m in (2, 3, 4):
            assert " rows=4 " in ops.plan_describe([w], m), ops.plan_describe([w], m)
            x = randx(m, K, seed=m)
            xt = torch.from_numpy(x).to(DEV)
            y = ops.linear_forward(w, xt)
            assert y.shape == (m, N)
            assert O.rel_err(y.cpu().numpy(), ref.y16(x)) <= 1e-2, (layout, m)
            assert O.rel_err(y.cpu().numpy().astype(np.float64), ref.y64(x)) <= 2e-3, (layout, m)
            for r in range(m):
                assert torch.equal(y[r:r + 1], ops.linear_forward(w, xt[r:r + 1].contiguous())), (layout, m, r)
            try:
                ops.set_knob("QLLM_STRIP1_MAX_M", 1)
                assert ops.plan_describe([w], m).startswith("strip nw=")
                y_dma = ops.linear_forward(w, xt)
            finally:
                ops.reset_knobs()
            assert O.rel_err(y.cpu().numpy(), y_dma.cpu().numpy()) <= 1e-3
            yb = ops.linear_forward(w, xt.to(torch.bfloat16))
            assert yb.dtype == torch.bfloat16 and O.rel_err(yb.float().cpu().numpy(), ref.y64(x)) <= 2e-2
    # grouped launch of unequal widths at batch 3
    ds = [synth("GPTQ", 4, 128, 4096, n, "asym", False, i == 0, seed=90 + i) for i, n in enumerate((4096, 1024, 1024))]
    glayers = [to_layer(d_, DEV) for d_ in ds]          # (kept alive: the descriptors point into their native copies)
    descs = [l.native_descriptor(0) for l in glayers]
    assert ops.plan_describe(descs, 3) == "strip1 nw=8 round=16 exact rows=4 grid=strips x 3 layout=strip-major"
    x = randx(3, 4096, seed=8)
    for o, d_ in zip(ops.linear_forward_grouped(descs, torch.from_numpy(x).to(DEV)), ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d_).y16(x)) <= 1e-2


B3_FORMS = [(1024, "nw=4 round=8 exact"), (2048, "nw=4 round=16 exact"), (3584, "nw=7 round=16 exact"), (4096, "nw=8 round=16 exact"),
            (5120, "nw=8 round=24"), (8192, "nw=8 round=32 exact"), (11008, "nw=15 round=24"), (14336, "nw=16 round=32"), (16384, "nw=16 round=32 exact")]


@pytest.mark.parametrize("K,form", B3_FORMS)
def test_every_form_with_3_bit_weights(K, form):
    """Round 6: 3-bit layers on the batch-1 kernel (B3 forms: two word loads + a funnel shift per k-step, slot-weighted patterns) -- HQQ g64
    with fp16 zero points, GPTQ g128 packed and symmetric zero points, bias, bf16; against the oracle, float64 of the reference's W, and the
    general strip kernel on the same descriptor."""
    from qllm_amd import ops
    N = 1024 if K > 8192 else 8192
    for layout, g, zk, bias in (("HQQ", 64, "asym", False), ("GPTQ", 128, "asym", True), ("GPTQ", 128, "sym", False), ("HQQ", 128, "asym", True)):
        d = synth(layout, 3, g, K, N, zk, False, bias, seed=K + g + len(layout))
        d["scales"] = (d["scales"].astype(np.float32) * (4096 / K) ** 0.5).astype(np.float16)
        layer = to_layer(d, DEV)
        w, keep = _native(layer, d, zk)
        plan = ops.plan_describe([w], 1)
        assert plan.startswith("strip1 " + form) and " bits=3 " in plan and ((" g64 " in plan) == (g == 64)), plan
        ref = Ref(d)
        for seed in (1, 2):
            x = randx(1, K, seed=seed)
            y = ops.linear_forward(w, torch.from_numpy(x).to(DEV)).cpu().numpy()
            assert O.rel_err(y, ref.y16(x)) <= 1e-2, (layout, g, zk)
            assert O.rel_err(y.astype(np.float64), ref.y64(x)) <= 2e-3, (layout, g, zk)
        xb = torch.from_numpy(randx(1, K, seed=9)).to(DEV).to(torch.bfloat16)
        yb = ops.linear_forward(w, xb)
        assert yb.dtype == torch.bfloat16 and O.rel_err(yb.float().cpu().numpy(), ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2
        try:
            ops.set_knob("QLLM_STRIP1_3BIT", 0)
            assert ops.plan_describe([w], 1).startswith("strip nw=")
            x = torch.from_numpy(randx(1, K, seed=1)).to(DEV)
            y_gen = ops.linear_forward(w, x)
        finally:
            ops.reset_knobs()
        assert O.rel_err(ops.linear_forward(w, x).cpu().numpy(), y_gen.cpu().numpy()) <= 1e-3
    # grouped q/k/v of unequal widths, HQQ g64 3 bits
    ds = [synth("HQQ", 3, 64, 4096, n, "asym", False, i == 1, seed=170 + i) for i, n in enumerate((4096, 1024, 1024))]
    glayers = [to_layer(d_, DEV) for d_ in ds]
    descs = [l.native_descriptor(0) for l in glayers]
    assert ops.plan_describe(descs, 1) == "strip1 nw=8 round=16 exact g64 bits=3 grid=strips x 3 layout=strip-major"
    x = randx(1, 4096, seed=4)
    for o, d_ in zip(ops.linear_forward_grouped(descs, torch.from_numpy(x).to(DEV)), ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d_).y16(x)) <= 1e-2
