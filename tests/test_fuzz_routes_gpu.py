"""-m gpu: seeded random walk over the planner's routes (round 6 added several: bitgemv, the batch-1 kernel's long / g64 / four-row forms,
tail split, grouped prefill grids, native bf16).  Every case goes through the MODULES (native copies, sibling groups where the case has
siblings) and must match the oracle within the north-star tolerance and float64 of the reference's own W within 3e-3 -- whatever kernel
the planner picked; the plan string is only printed on failure."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MS = (1, 2, 3, 4, 5, 8, 16, 17, 31, 33, 64, 65, 128, 129, 200, 384, 500, 777, 1030)


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    bits = int(rng.choice([2, 3, 4, 4, 4, 5, 6, 8]))
    layout = str(rng.choice(["GPTQ", "HQQ", "GEMM"] if bits == 4 else ["GPTQ", "HQQ"]))
    g = int(rng.choice([32, 64, 128, 128]))
    K = int(rng.choice([256, 512, 1024, 1536, 2048, 3072, 4096, 5120]))
    N = int(rng.choice([128, 256, 512, 1024, 1536, 2048, 4096, 1000 if layout != "GEMM" and bits in (2, 4, 8) else 768]))
    zk = "f16" if layout == "HQQ" else str(rng.choice(["asym", "asym", "sym"]))
    if layout == "GEMM":
        zk = "asym"
    if layout == "GPTQ" and (N * bits) % 32:   # (packed zero points -- the symmetric ones of the synthetic layer too -- need whole words per row)
        N = 1024
    act = bool(layout == "GPTQ" and bits in (3, 4) and zk == "asym" and rng.random() < 0.2 and K % g == 0)
    bias = bool(rng.random() < 0.4)
    ms = [int(m) for m in rng.choice(MS, size=4, replace=False)]
    bf16 = bool(rng.random() < 0.3)
    return dict(bits=bits, layout=layout, g=g, K=K, N=N, zk=zk, act=act, bias=bias, ms=ms, bf16=bf16)


N_SINGLE, N_GROUP = int(os.environ.get("QLLM_FUZZ_SINGLE", "48")), int(os.environ.get("QLLM_FUZZ_GROUP", "16"))   # (more seeds: a soak run)


@pytest.mark.parametrize("seed", range(N_SINGLE))
def test_random_single_layer(seed):
    from qllm_amd import ops
    c = _case(seed)
    d = synth(c["layout"], c["bits"], c["g"], c["K"], c["N"], c["zk"], c["act"], c["bias"], seed=seed)
    d["scales"] = (d["scales"].astype(np.float32) * (16.0 / 2 ** c["bits"]) * (1024 / c["K"]) ** 0.5).astype(np.float16)
    layer = to_layer(d, DEV)
    ref = Ref(d)
    for m in c["ms"]:
        x = randx(m, c["K"], seed=seed * 31 + m)
        xt = torch.from_numpy(x).to(DEV)
        xt = xt.to(torch.bfloat16) if c["bf16"] else xt
        y = layer(xt)
        try:
            plan = ops.plan_describe([layer.decode_descriptor(None, 0)], m) if not c["act"] else "act-order"
        except Exception as e:  # noqa: BLE001
            plan = f"({e})"
        xin = xt.float().cpu().numpy().astype(np.float16)
        tol16, tol64 = (2e-2, 1.2e-2) if c["bf16"] else (1e-2, 3e-3)
        assert y.shape == (m, c["N"]) and y.dtype == xt.dtype, (c, m, plan)
        assert O.rel_err(y.float().cpu().numpy(), ref.y16(xin)) <= tol16, (c, m, plan)
        assert O.rel_err(y.float().cpu().numpy().astype(np.float64), ref.y64(xin)) <= tol64, (c, m, plan)


@pytest.mark.parametrize("seed", range(N_GROUP))
def test_random_sibling_group(seed):
    """2-3 siblings of random widths sharing x, 4 bits (and a 3-bit / 2-bit sibling now and then: the group is partitioned or stands down)."""
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM, install_sibling_groups
    rng = np.random.default_rng(5000 + seed)
    layout = str(rng.choice(["GPTQ", "HQQ", "GEMM"]))
    g = int(rng.choice([64, 128])) if layout != "GEMM" else 128
    K = int(rng.choice([1024, 2048, 4096]))
    widths = [int(w) for w in rng.choice([128, 256, 512, 1024, 2048, 4096], size=int(rng.choice([2, 3])))]
    bits = [4] * len(widths)
    if layout != "GEMM" and rng.random() < 0.3:
        bits[-1] = int(rng.choice([2, 3]))
    zk = "f16" if layout == "HQQ" else "asym"
    ds = [synth(layout, b, g, K, n, zk, False, bool(i == 0), seed=seed * 7 + i) for i, (n, b) in enumerate(zip(widths, bits))]
    for d in ds:
        d["scales"] = (d["scales"].astype(np.float32) * (16.0 / 2 ** d["bits"]) * (1024 / K) ** 0.5).astype(np.float16)

    class Parent(torch.nn.Module):
        pass
    par = Parent()
    names = ("q_proj", "k_proj", "v_proj") if len(ds) == 3 else ("gate_proj", "up_proj")
    for nm, d in zip(names, ds):
        setattr(par, nm, to_layer(d, DEV))
    install_sibling_groups(par, [QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM])
    for m in (1, 3, 16, 40, 130, 400, 2048):
        x = randx(m, K, seed=seed + m)
        xt = torch.from_numpy(x).to(DEV)
        outs = [getattr(par, nm)(xt) for nm in names]
        for o, d in zip(outs, ds):
            assert O.rel_err(o.cpu().numpy(), Ref(d).y16(x)) <= 1e-2, (layout, g, K, widths, bits, m)
            assert O.rel_err(o.cpu().numpy().astype(np.float64), Ref(d).y64(x)) <= 3e-3, (layout, g, K, widths, bits, m)
