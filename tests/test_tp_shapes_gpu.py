"""-m gpu: BASELINE configs[4] -- Llama-2-70B AWQ/GPTQ w4 g128 shapes, full and as the per-rank shards of an 8-way
tensor-parallel split: 8192 -> 8192 / 1024 (q, o / k, v), 8192 -> 28672 (gate, up), 28672 -> 8192 (down);
shards 8192 -> 1024, 8192 -> 128, 8192 -> 3584 (column-parallel), 1024 -> 8192, 3584 -> 8192 (row-parallel).

The full-size layers are too big for a full-matrix oracle in a test (235 M weights), so the packed buffers are random words
(any int32 is a valid qweight) and the oracle dequantises COLUMN SLICES of them: every output column depends only on its own
column of qweight / scales / qzeros, so a slice of the oracle's y is the oracle of the slice."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import LAYER, randx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = 128


def _rand_layer(layout, K, N, seed, bias=False):
    rng = np.random.default_rng(seed)
    if layout == "GEMM":
        qweight = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32)
    else:
        qweight = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // 8, N), dtype=np.int64).astype(np.int32)
    qzeros = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // G, N // 8), dtype=np.int64).astype(np.int32)
    scales = ((rng.random((K // G, N)) * 0.4 + 0.8) / (K ** 0.5 * 6.5)).astype(np.float16)
    b = (rng.standard_normal(N) * 0.1).astype(np.float16) if bias else None
    return dict(layout=layout, K=K, N=N, qweight=qweight, qzeros=qzeros, scales=scales, bias=b)


def _module(d):
    layer = LAYER[d["layout"]](4, G, d["K"], d["N"], d["bias"] is not None, dtype=torch.float16)
    layer.qweight, layer.qzeros = torch.from_numpy(d["qweight"]), torch.from_numpy(d["qzeros"])
    layer.scales = torch.from_numpy(d["scales"])
    if d["bias"] is not None:
        layer.bias = torch.from_numpy(d["bias"])
    return layer.to(DEV)


def _cols(d, c0, c1):
    """the layer restricted to output columns [c0, c1) (multiples of 8)"""
    s = dict(d)
    s["N"] = c1 - c0
    s["qweight"] = d["qweight"][:, c0 // 8:c1 // 8] if d["layout"] == "GEMM" else d["qweight"][:, c0:c1]
    s["qzeros"] = d["qzeros"][:, c0 // 8:c1 // 8]
    s["scales"] = d["scales"][:, c0:c1]
    s["bias"] = d["bias"][c0:c1] if d["bias"] is not None else None
    return {k: (np.ascontiguousarray(v) if isinstance(v, np.ndarray) else v) for k, v in s.items()}


def _oracle_y(d, x):
    w = O.dequant(d["layout"], d["qweight"], d["scales"], d["qzeros"], None, 4, G, d["K"], 0)
    y = torch.from_numpy(x).double() @ torch.from_numpy(w).double()
    return (y + torch.from_numpy(d["bias"]).double() if d["bias"] is not None else y).numpy()


SHAPES = [  # (K, N, what)
    (8192, 8192, "q / o full"), (8192, 1024, "k / v full = q shard"), (8192, 128, "k / v shard"),
    (8192, 28672, "gate / up full"), (8192, 3584, "gate / up shard"), (28672, 8192, "down full"),
    (3584, 8192, "down shard (row-parallel)"), (1024, 8192, "o shard (row-parallel)"),
]


@pytest.mark.parametrize("layout", ["GPTQ", "GEMM"])
@pytest.mark.parametrize("K,N,what", SHAPES)
def test_llama70b_shapes_decode_and_prefill_vs_oracle(layout, K, N, what):
    d = _rand_layer(layout, K, N, seed=K + N, bias=(N == 1024))
    layer = _module(d)
    # three 128-column windows: first, one in the middle, last
    wins = sorted({0, (N // 2) // 128 * 128, N - 128})
    for m in (1, 4, 16, 300) + ((2048,) if K * N <= 8192 * 8192 else ()):
        x = randx(m, K, seed=m)
        y = layer(torch.from_numpy(x).to(DEV)).float().cpu().numpy()
        assert y.shape == (m, N) and np.isfinite(y).all()
        for c0 in wins:
            ref = _oracle_y(_cols(d, c0, c0 + 128), x)
            err = np.abs(y[:, c0:c0 + 128] - ref).max() / np.abs(ref).max()
            assert err <= 2e-3, (what, layout, m, c0, err)


def test_column_shard_equals_slice_of_full_and_row_shards_sum():
    """Column-parallel: rank r's output IS columns [r N/P, (r+1) N/P) of the unsharded output.  The arithmetic per column is
    the same fp32 sum of the same products; only its ORDER can differ, because the narrower shard may be served by another plan
    (more split-K blocks at prefill sizes, another K split over waves in the decode strips), so the two agree to one fp16
    rounding of an fp32 sum rather than bit for bit: almost every element identical, none off by more than an ulp or two.
    Row-parallel: the P partial products sum to the unsharded output."""
    import qllm_amd.parallel as TP
    P = 8
    for layout in ("GPTQ", "GEMM"):
        full_d = _rand_layer(layout, 8192, 8192, seed=5)
        full_cpu = LAYER[layout](4, G, 8192, 8192, False, dtype=torch.float16)
        full_cpu.qweight, full_cpu.qzeros = torch.from_numpy(full_d["qweight"]), torch.from_numpy(full_d["qzeros"])
        full_cpu.scales = torch.from_numpy(full_d["scales"])
        full = _module(full_d)
        xs = {m: torch.from_numpy(randx(m, 8192, seed=m)).to(DEV) for m in (1, 256)}
        y_full = {m: full(x) for m, x in xs.items()}
        nl = 8192 // P
        for r in (0, 3, 7):
            shard = TP.shard_columns(full_cpu, r, P).to(DEV)
            for m in (256, 1):
                a, b = shard(xs[m]).float(), y_full[m][:, r * nl:(r + 1) * nl].float()
                assert (a - b).abs().max() <= 1e-3 * b.abs().max(), (layout, r, m)
                assert (a == b).float().mean() >= 0.9, (layout, r, m)
        # row-parallel (o_proj / down_proj): sum of the partials of the K shards
        parts = []
        kl = 8192 // P
        for r in range(P):
            sh = TP.shard_rows(full_cpu, r, P).to(DEV)
            parts.append(sh(xs[256][:, r * kl:(r + 1) * kl].contiguous()).float())
        y_sum = torch.stack(parts).sum(0)
        assert (y_sum - y_full[256].float()).abs().max() <= 2e-3 * y_full[256].float().abs().max()


def test_tp_block_through_modules_uses_grouped_shard_launches():
    """One rank's decoder block of the 70B TP=8 stack (tools/tp_bench.py): q/k/v and gate/up shards are sibling groups ->
    grouped strip launches; world size 1 => no collective."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    from tools import tp_bench
    blocks = tp_bench.build_stack(8, 2, torch.device(DEV), seed=3)
    b0 = blocks[0]
    assert b0.q_proj._siblings.describe(1).startswith("strip") and b0.gate_proj._siblings.describe(1) == "strip1 nw=8 round=32 exact grid=strips x 2 layout=strip-major"
    h = torch.randn(1, tp_bench.H70, device=DEV, dtype=torch.float16)
    y = h
    for b in blocks:
        y = b(y)
    assert y.shape == (1, tp_bench.H70) and torch.isfinite(y.float()).all()
    assert b0.q_proj._siblings.grouped_launches == 1 and b0.gate_proj._siblings.grouped_launches == 1
