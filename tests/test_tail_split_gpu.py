"""-m gpu: gemm3's K-split of the ragged last round of tiles (round 6) -- shapes with more 256x128 tiles than CUs (Llama-2-13B's
5120-wide layers: 320 tiles) against the oracle, against the unsplit launch, fp16 and bf16, every layout the kernel reads."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-2


@pytest.mark.parametrize("layout,K,N,M,ts", [("GEMM", 5120, 5120, 2048, 4), ("GPTQ", 4096, 4096, 2304, 4), ("GPTQ", 2048, 13824, 2048, 2)])
def test_tail_split_matches_oracle_and_unsplit_launch(layout, K, N, M, ts):
    from qllm_amd import ops
    d = synth(layout, 4, 128, K, N, "asym", False, True, seed=K + N)
    layer = to_layer(d, DEV)
    ref = Ref(d)
    x = randx(M, K, seed=5)
    xt = torch.from_numpy(x).to(DEV)
    y = layer(xt)
    assert ops.plan_describe([layer.decode_descriptor()], M).endswith(f"tail_split={ts} layout=strip-major")
    want = ref.y16(x)
    assert O.rel_err(y.cpu().numpy(), want) <= TOL
    rows = np.r_[0:8, M - 264:M - 248, M - 8:M]          # (fp64 truth on a few rows of the first / a middle / the last row tile)
    assert O.rel_err(y[rows].cpu().numpy(), ref.y64(x[rows])) <= 2e-3
    y2 = layer(xt)
    assert torch.equal(y, y2)                              # fixed-order sum of the partial tiles: deterministic
    try:
        ops.set_knob("QLLM_GEMM3_TAIL", 0)
        assert "tail_split" not in ops.plan_describe([layer.decode_descriptor()], M)
        y_unsplit = layer(xt)
    finally:
        ops.reset_knobs()
    assert O.rel_err(y.cpu().numpy(), y_unsplit.cpu().numpy()) <= 1e-3   # same products, another fp32 summation order in the tail tiles
    full = (M // 256) * (N // 128) // 256 * 256                          # tiles of the whole rounds: bit-identical to the unsplit launch
    if full:
        first_rows = full // (N // 128) * 256                            # (n-fastest tile order: whole row tiles in the unsplit segment)
        assert torch.equal(y[:first_rows], y_unsplit[:first_rows])
    # bf16 activations: x converted into the workspace behind the tail's slabs, result rounded to bf16
    yb = layer(xt.to(torch.bfloat16))
    assert yb.dtype == torch.bfloat16 and O.rel_err(yb.float().cpu().numpy(), want) <= TOL
    # the workspace is left clean (counters re-armed): a split-K decode call right after still works
    y1 = layer(xt[:1])
    assert O.rel_err(y1.cpu().numpy(), ref.y16(x[:1])) <= TOL


@pytest.mark.parametrize("layout,g,zk", [("GPTQ", 128, "asym"), ("GEMM", 128, "asym"), ("HQQ", 64, "f16"), ("GPTQ", 32, "sym")])
def test_prefill_kernel_dequantises_bit_exactly(layout, g, zk):
    """x = the identity (4096 one-hot rows): y IS the kernel's W, every product exact and alone in its fp32 sum -- it must equal the
    reference's fp16(fp16(s q) - fp16(z s)) (DequantizeLinearBlockWise, quant_linear_gptq.py:46-48) bit for bit.  Pins the staging waves'
    arithmetic of gemm3 (round 6: odd nibbles under the 64 + q pattern, even ones under 1024 + q) on the native copy and in place."""
    import os
    from qllm_amd import ops
    from gpu_util import oracle_w
    K, N = 4096, 512
    d = synth(layout, 4, g, K, N, zk, False, False, seed=g + N)
    want = oracle_w(d)                                   # [K, N] fp16, the oracle's (= the reference's) W
    eye = torch.eye(K, dtype=torch.float16, device=DEV)
    for native in ("1", "0"):
        os.environ["QLLM_NATIVE_LAYOUT"] = native
        try:
            layer = to_layer(d, DEV)
            assert ops.plan_describe([layer.decode_descriptor()], K).startswith("gemm3"), ops.plan_describe([layer.decode_descriptor()], K)
            got = layer(eye).cpu().numpy()
        finally:
            os.environ.pop("QLLM_NATIVE_LAYOUT", None)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (layout, native)


@pytest.mark.parametrize("layout,shapes,g,zk,dtype", [
    ("GEMM", [(4096, 4096), (4096, 1024), (4096, 1024)], 128, "asym", torch.float16),      # GQA q/k/v: 384 tiles, tail split
    ("HQQ", [(4096, 5632), (4096, 5632)], 64, "f16", torch.float16),                        # gate/up, fp16 zero points
    ("GPTQ", [(2048, 2048), (2048, 2048), (2048, 2048)], 128, "asym", torch.bfloat16),      # bf16: the native form, grouped
])
def test_prefill_sized_sibling_group_is_one_launch_and_matches_single_launches(layout, shapes, g, zk, dtype):
    """Round 6: a sibling group at M = 2048 -- ONE launch of the 256x128 kernel (gemm3.hip, grouped form) for q/k/v / gate/up.  Tiles
    of the whole rounds are computed exactly as in the single launches (bit-equal); tiles of a K-split last round differ by the fp32
    summation order only; every output within 1e-2 of the oracle; bias per layer."""
    from qllm_amd import ops
    from qllm_amd.modeling.q_layers import fuse_siblings
    M = 2048
    ds = [synth(layout, 4, g, K, N, zk, False, i == 1, seed=31 * i + N) for i, (K, N) in enumerate(shapes)]
    singles = [to_layer(d, DEV) for d in ds]
    grouped = [to_layer(d, DEV) for d in ds]
    grp = fuse_siblings(grouped)
    x = torch.from_numpy(randx(M, shapes[0][0], seed=11)).to(DEV).to(dtype)
    assert f"layers={len(shapes)}" in grp.describe(M) and grp.describe(M).startswith("gemm3")
    before = grp.grouped_launches
    outs = [l(x) for l in grouped]
    assert grp.grouped_launches == before + 1
    tol = 1e-2 if dtype == torch.bfloat16 else 1e-3
    for o, s_, d in zip(outs, singles, ds):
        y1 = s_(x)
        assert o.shape == (M, d["N"]) and o.dtype == dtype and o.is_contiguous()
        assert O.rel_err(o.float().cpu().numpy(), y1.float().cpu().numpy()) <= tol
        want = Ref(d).y16(x.to(torch.float16).cpu().numpy())
        assert O.rel_err(o.float().cpu().numpy(), want) <= TOL
    if "tail_split" not in grp.describe(M) and not any("tail_split" in ops.plan_describe([s_.decode_descriptor()], M) for s_ in singles):
        assert all(torch.equal(o, s_(x)) for o, s_ in zip(outs, singles))      # whole rounds only, here and there: the very same tiles
    # a second call with a new tensor, and determinism
    x2 = x.clone()
    assert all(torch.equal(a, b) for a, b in zip([l(x2) for l in grouped], outs))
    # below 384 rows the group is not served in one launch: every layer runs its own, results unchanged, and the refusal does not
    # switch the decode-sized grouping off
    xs = x[:200].contiguous()
    y_small = grouped[0](xs)
    assert O.rel_err(y_small.float().cpu().numpy(), singles[0](xs).float().cpu().numpy()) <= tol
    launches = grp.grouped_launches
    grouped[0](x[:1].contiguous())
    assert grp.grouped_launches == launches + 1


def test_act_order_siblings_run_as_one_group():
    """Round 6: q/k/v of a GPTQ act-order checkpoint share their permutation (gptq.py:168: it comes from the shared input's Hessian), so
    their row-sorted native copies take the SAME gathered x: one gather + ONE grouped launch at decode, mid-batch and prefill sizes --
    results equal to the layers on their own, and the oracle's with the reference's in-place g_idx gather."""
    from qllm_amd import ops
    from qllm_amd.modeling.q_layers import fuse_siblings
    from gpu_util import oracle_y
    K, g = 4096, 128
    base = synth("GPTQ", 4, g, K, 4096, "asym", True, False, seed=41)
    ds = [base] + [dict(synth("GPTQ", 4, g, K, n, "asym", False, i == 0, seed=42 + i), g_idx=base["g_idx"].copy()) for i, n in enumerate((1024, 1024))]
    singles = [to_layer(d, DEV) for d in ds]
    grouped = [to_layer(d, DEV) for d in ds]
    refs = [Ref(d) for d in ds]                             # (the oracle's W, g_idx gather included, converted once per layer)
    grp = fuse_siblings(grouped)
    calls, real = [], ops.gather_columns
    ops.gather_columns = lambda x, perm: (calls.append(1), real(x, perm))[1]
    try:
        for m in (1, 16, 64, 2048):
            x = torch.from_numpy(randx(m, K, seed=m)).to(DEV)
            [l(x) for l in grouped]                          # (first pass builds the row-sorted copies and interns the permutation)
            x = x.clone()
            calls.clear()
            before = grp.grouped_launches
            ys = [l(x) for l in grouped]
            assert len(calls) == 1 and grp.grouped_launches == before + 1, (m, calls, grp.grouped_launches - before)
            for r, y, s_ in zip(refs, ys, singles):
                assert O.rel_err(y.cpu().numpy(), r.y16(x.cpu().numpy())) <= TOL, m
                assert O.rel_err(y.cpu().numpy(), s_(x).cpu().numpy()) <= 1e-3, m
    finally:
        ops.gather_columns = real
    # a sibling with ANOTHER permutation: the group stands down, every layer on its own (still right)
    other = to_layer(synth("GPTQ", 4, g, K, 1024, "asym", True, False, seed=77), DEV)
    mixed = [to_layer(ds[0], DEV), other]
    g2 = fuse_siblings(mixed)
    x = torch.from_numpy(randx(4, K, seed=5)).to(DEV)
    for _ in range(2):
        ys = [l(x) for l in mixed]
    assert g2.grouped_launches == 0
    assert O.rel_err(ys[0].cpu().numpy(), refs[0].y16(x.cpu().numpy())) <= TOL
