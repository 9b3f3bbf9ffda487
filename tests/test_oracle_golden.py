"""Pin the CPU oracle (oracle/ref_cpu.py) against golden vectors minted from the reference itself
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import ref_cpu as O
from conftest import golden_names, load_golden


def _int_zeros(g):
    if g["layout"] == "HQQ":
        return None
    return g["zeros"]


def test_packed_buffers_bit_exact(golden):
    g = golden
    if g["layout"] == "GPTQ":
        qw, qz = O.pack_gptq(g["q"], g["zeros"], g["bits"], g["compat"])
    elif g["layout"] == "HQQ":
        qw = O.pack_along_rows(g["q"], g["bits"])
        qz = g["zeros"]
    else:
        qw, qz = O.pack_awq(g["q"], g["zeros"])
    assert qw.dtype == g["qweight"].dtype and qw.shape == g["qweight"].shape
    assert np.array_equal(qw, g["qweight"])
    assert qz.shape == g["qzeros"].shape
    assert np.array_equal(qz, g["qzeros"])


def test_int_unpack_bit_exact(golden):
    g = golden
    if g["layout"] in ("GPTQ", "HQQ"):
        q = O.gptq_int_weight(g["qweight"], g["bits"], g["K"])
    else:
        q = O.awq_int_weight(g["qweight"], g["K"], g["N"])
    assert np.array_equal(q, g["q"])
    if g["layout"] == "GPTQ":
        assert np.array_equal(O.gptq_int_zeros(g["qzeros"], g["bits"], g["N"], g["compat"]), g["zeros"])
    elif g["layout"] == "GEMM":
        assert np.array_equal(O.awq_int_zeros(g["qzeros"], g["N"]), g["zeros"])


def test_dequant_bit_exact_vs_reference_forward_path(golden):
    g = golden
    if "W_fwd" not in g:
        pytest.skip("AWQ GEMM has no CPU forward dequant in the reference")
    gi = g["g_idx"] if O.is_act_order(g["g_idx"], g["groupsize"]) else None
    w = O.dequant(g["layout"], g["qweight"], g["scales"], g["qzeros"], gi, g["bits"], g["groupsize"], g["K"],
                  g["compat"])
    assert w.dtype == np.float16
    assert np.array_equal(w.view(np.uint16), g["W_fwd"].view(np.uint16))


def test_dequant_bit_exact_vs_reference_unpack(golden):
    g = golden
    if "W_unpack" not in g:
        pytest.skip("large fixture keeps W_fwd only")
    gi = g["g_idx"]  # unpack() always gathers by g_idx (compress_weight.py:143)
    if g["layout"] == "GPTQ":
        # unpack() never applies the AutoGPTQ offset: it sees the stored zeros as-is
        w = O.dequant_gptq(g["qweight"], g["scales"], g["qzeros"], gi, g["bits"], g["groupsize"], g["K"], 0)
    else:
        w = O.dequant(g["layout"], g["qweight"], g["scales"], g["qzeros"], None, g["bits"], g["groupsize"], g["K"])
    assert np.array_equal(w.T.view(np.uint16), g["W_unpack"].view(np.uint16))


def test_forward_matches_reference(golden):
    g = golden
    y = O.forward(g["layout"], g["x"], g["qweight"], g["scales"], g["qzeros"], g["g_idx"], g["bias"], g["bits"],
                  g["groupsize"], g["K"], g["compat"]).numpy()
    assert y.shape == g["y"].shape and y.dtype == np.float16
    assert O.rel_err(y, g["y"]) <= 1e-3
    y1 = O.forward(g["layout"], g["x"][:1], g["qweight"], g["scales"], g["qzeros"], g["g_idx"], g["bias"],
                   g["bits"], g["groupsize"], g["K"], g["compat"]).numpy()
    assert O.rel_err(y1, g["y1"]) <= 1e-3
    # the fp32-carried form used for large M agrees with the half GEMM to rounding
    gi0 = g["g_idx"] if (g["layout"] == "GPTQ" and O.is_act_order(g["g_idx"], g["groupsize"])) else None
    w0 = O.dequant(g["layout"], g["qweight"], g["scales"], g["qzeros"], gi0, g["bits"], g["groupsize"], g["K"], g["compat"])
    assert O.rel_err(O.matmul_f16_via_f32(g["x"], w0, g["bias"]).numpy(), g["y"]) <= 1e-3
    # and the fp16 path is itself close to exact arithmetic on the same operands
    gi = g["g_idx"] if (g["layout"] == "GPTQ" and O.is_act_order(g["g_idx"], g["groupsize"])) else None
    w = O.dequant(g["layout"], g["qweight"], g["scales"], g["qzeros"], gi, g["bits"], g["groupsize"], g["K"],
                  g["compat"])
    assert O.rel_err(g["y"], O.matmul_f64(g["x"], w, g["bias"])) <= 2e-3


@pytest.mark.parametrize("name", ["gptq_w4_g128_actorder", "gptq_w3_g64_actorder", "awq_w4_g128_asym",
                                  "hqq_w3_g64", "gptq_w5_g128_asym", "gptq_w4_g128_autogptq"])
def test_loop_restatement_agrees(name):
    """Second, element-by-element derivation of the layouts (small slices only)."""
    g = load_golden(name)
    k_small = 64 if g["bits"] in (2, 4, 8) else 64
    gi = g["g_idx"] if (g["layout"] == "GPTQ" and O.is_act_order(g["g_idx"], g["groupsize"])) else None
    w_vec = O.dequant(g["layout"], g["qweight"], g["scales"], g["qzeros"], gi, g["bits"], g["groupsize"], g["K"],
                      g["compat"])
    w_loop = O.dequant_loops(g["layout"], g["qweight"], g["scales"][:, :16], g["qzeros"], gi, g["bits"],
                             g["groupsize"], k_small, g["compat"])
    assert np.array_equal(w_loop.view(np.uint16), w_vec[:k_small, :16].view(np.uint16))


def test_autogptq_fixup():
    g = load_golden("gptq_w4_g128_autogptq")
    fixed = O.autogptq_fixup_qzeros(g["qzeros"], g["bits"], g["N"])
    assert np.array_equal(fixed, g["qzeros_fixed"])
    # after the fix-up the stored zeros are the plain zeros, and forward with add_zero_bias=0 is unchanged
    assert np.array_equal(O.gptq_int_zeros(fixed, g["bits"], g["N"], 0), g["zeros"])


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 6, 7, 8])
def test_pack_unpack_roundtrip_random(bits):
    rng = np.random.default_rng(bits)
    q = rng.integers(0, 2 ** bits, size=(96, 40), dtype=np.int32)
    assert np.array_equal(O.unpack_along_rows(O.pack_along_rows(q, bits), bits, 96), q)
    z = rng.integers(0, 2 ** bits, size=(5, 64), dtype=np.int32)
    assert np.array_equal(O.unpack_along_cols(O.pack_along_cols(z, bits), bits, 64), z)


# ---- the torch formulation that bench.py times as cpu_baseline (oracle/ref_torch.py) ------------------------------------
@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("gptq_w") and any(f"_w{b}_" in n for b in (2, 4, 8))])
def test_torch_formulation_is_bit_identical_to_the_oracle_and_the_reference(name):
    import torch
    from oracle import ref_torch as T
    g = load_golden(name)
    act = O.is_act_order(g["g_idx"], g["groupsize"])
    t = {k: torch.from_numpy(np.ascontiguousarray(g[k])) for k in ("qweight", "scales", "qzeros", "g_idx", "x")}
    gi = t["g_idx"] if act else None
    w = T.dequant_gptq_torch(t["qweight"], t["scales"], t["qzeros"], g["groupsize"], g["bits"], gi, g["compat"])
    assert w.dtype == torch.float16
    w_oracle = O.dequant("GPTQ", g["qweight"], g["scales"], g["qzeros"], g["g_idx"] if act else None, g["bits"], g["groupsize"],
                         g["K"], g["compat"])
    assert np.array_equal(w.numpy().view(np.uint16), w_oracle.view(np.uint16))
    if "W_fwd" in g:
        assert np.array_equal(w.numpy().view(np.uint16), g["W_fwd"].view(np.uint16))      # the reference's own W, bit for bit
    bias = torch.from_numpy(g["bias"]) if g["bias"] is not None else None
    y = T.forward_gptq_torch(t["x"], t["qweight"], t["scales"], t["qzeros"], g["groupsize"], g["bits"], gi, bias, g["compat"])
    assert O.rel_err(y.numpy(), g["y"]) <= 1e-3
