"""-m gpu: the bit-stream matvec (csrc/bitgemv.hip, round 6) -- 2 / 5 / 6 / 7 / 8-bit layers at decode sizes, fused, against the oracle
(the reference's CPU path: DequantizeLinearBlockWise + matmul, quant_linear_gptq.py:13-52,85) and float64 of the reference's own W."""
import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-2


@pytest.mark.parametrize("bits", [2, 5, 6, 7, 8])
@pytest.mark.parametrize("layout,g,K,N,zk,bias", [("GPTQ", 128, 4096, 4096, "asym", True), ("HQQ", 64, 4096, 11008, "f16", False),
                                                  ("GPTQ", 128, 11008, 4096, "sym", False), ("GPTQ", 32, 1024, 1000, "asym", True)])
def test_bitgemv_matches_oracle(bits, layout, g, K, N, zk, bias):
    from qllm_amd import ops
    if layout == "GPTQ" and zk == "asym" and (N * bits) % 32:
        pytest.skip("packed zero points need N * bits % 32 == 0")
    d = synth(layout, bits, g, K, N, zk, False, bias, seed=K + N + bits)
    ref = Ref(d)
    if zk == "sym":
        d = dict(d, qzeros=None)
        qw, sc = torch.from_numpy(d["qweight"]).to(DEV), torch.from_numpy(d["scales"]).to(DEV)
        w, keep = ops.make_weight("GPTQ", qw, sc, None, None, None, K, N, g, bits, 0)
        fwd = lambda xt: ops.linear_forward(w, xt)  # noqa: E731
    else:
        layer = to_layer(d, DEV)
        w = layer.decode_descriptor()
        fwd = layer
    assert ops.plan_describe([w], 1).startswith(f"bitgemv bits={bits} ")
    for m in (1, 3, 8, 16):
        x = randx(m, K, seed=m)
        y = fwd(torch.from_numpy(x).to(DEV)).cpu().numpy()
        assert y.shape == (m, N)
        assert O.rel_err(y, ref.y16(x)) <= TOL, (bits, layout, m)
        assert O.rel_err(y, ref.y64(x)) <= 2e-3, (bits, layout, m)
    # bf16 activations: converted while they are staged; bf16 result
    xb = torch.from_numpy(randx(4, K, seed=9)).to(DEV).to(torch.bfloat16)
    yb = fwd(xb)
    assert yb.dtype == torch.bfloat16
    assert O.rel_err(yb.float().cpu().numpy(), ref.y64(xb.float().cpu().numpy().astype(np.float16))) <= TOL
    # deterministic (fixed-order sums), and the workspace is left clean
    x1 = torch.from_numpy(randx(1, K, seed=1)).to(DEV)
    assert torch.equal(fwd(x1), fwd(x1))


def test_bitgemv_replaces_dequant_plus_gemm_and_can_be_switched_off():
    """The module path: an 8-bit layer at batch 1 runs ONE fused launch; with QLLM_BITGEMV = 0 the library refuses the call and the module
    falls back to the library's dequant kernel + a dense GEMM (the reference's branch (B)) -- same result within the contract."""
    from qllm_amd import ops
    d = synth("GPTQ", 8, 128, 4096, 4096, "asym", False, True, seed=88)
    layer = to_layer(d, DEV)
    x = torch.from_numpy(randx(2, 4096, seed=2)).to(DEV)
    y = layer(x)
    try:
        ops.set_knob("QLLM_BITGEMV", 0)
        assert ops.plan_describe([layer.decode_descriptor()], 2).startswith("unsupported")
        y_b = layer(x)
    finally:
        ops.reset_knobs()
    assert O.rel_err(y.cpu().numpy(), y_b.cpu().numpy()) <= 2e-3
    assert O.rel_err(y.cpu().numpy(), Ref(d).y16(x.cpu().numpy())) <= TOL
