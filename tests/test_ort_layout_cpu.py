"""ORT / MatMulNBits blob layout, CPU side: the oracle against the goldens minted from the
reference (tests/golden/make_goldens_ort.py), and the host logic of QuantLinearORT (pack / unpack / state dict /
dispatch).  No compute goes through the HIP library here."""
import numpy as np
import pytest
import torch

from conftest import load_golden, ort_golden_names
from oracle import ref_cpu as O


def _bits(a):
    a = np.asarray(a)
    return a.view(np.uint16) if a.dtype == np.float16 else a


@pytest.mark.parametrize("name", ort_golden_names())
def test_oracle_reproduces_reference_ort_goldens(name):
    g = load_golden(name)
    K, N, gs = g["K"], g["N"], g["groupsize"]
    assert np.array_equal(O.ort_int_weight(g["qweight"]), g["q"])
    if g["qzeros"].dtype == np.uint8:
        assert np.array_equal(O.ort_int_zeros(g["qzeros"], N, K // gs), g["zeros"])
    qw, qz, sf = O.pack_ort(g["q"], g["zeros"], g["scales"])
    assert np.array_equal(qw, g["qweight"]) and np.array_equal(_bits(qz), _bits(g["qzeros"]))
    assert np.array_equal(_bits(sf), _bits(g["scales_flat"]))
    gi = g["g_idx"] if O.ort_is_act_order(g["g_idx"]) else None
    w_nk = O.dequant_ort(g["qweight"], g["scales_flat"], g["qzeros"], gi, gs, K, N)
    assert np.array_equal(_bits(w_nk), _bits(g["W_unpack"]))  # bit-exact W
    y = O.matmul_f16(g["x"], np.ascontiguousarray(w_nk.T), g["bias"])
    assert O.rel_err(y, g["y"]) <= 1e-3
    assert O.rel_err(O.matmul_f16(g["x"][:1], np.ascontiguousarray(w_nk.T), g["bias"]), g["y1"]) <= 1e-3


def test_oracle_odd_block_count_padding_nibble():
    """The reference's own CPU dequant cannot reshape an odd number of blocks; the blob format still defines it (a padding
    high nibble per row).  Pack -> unpack round trip of the restated layout."""
    rng = np.random.default_rng(3)
    K, N, gs = 384, 64, 128
    q = rng.integers(0, 16, (K, N)).astype(np.int32)
    z = rng.integers(0, 16, (K // gs, N)).astype(np.int32)
    s = (rng.random((K // gs, N)) * 0.01 + 0.002).astype(np.float16)
    qw, qz, sf = O.pack_ort(q, z, s)
    assert qz.shape == (N * 2,) and np.all((qz.reshape(N, 2)[:, 1] >> 4) == 0)
    assert np.array_equal(O.ort_int_weight(qw), q) and np.array_equal(O.ort_int_zeros(qz, N, 3), z)
    w = O.dequant_ort(qw, sf, qz, None, gs, K, N)
    gi = np.arange(K) // gs
    expect = ((q - z[gi]).astype(np.float16).astype(np.float32) * s[gi].astype(np.float32)).astype(np.float16).T
    assert np.array_equal(_bits(w), _bits(expect))


def _layer_from_golden(g):
    from qllm_amd.modeling.q_layers import QuantLinearORT
    layer = QuantLinearORT(4, g["groupsize"], g["K"], g["N"], g["bias"] is not None, dtype=torch.float16)
    gi = torch.from_numpy(g["g_idx"]).long()
    w_kn = torch.from_numpy(g["scales"]).double()[gi] * (torch.from_numpy(g["q"]).double() - torch.from_numpy(g["zeros"]).double()[gi])
    lin = torch.nn.Linear(g["K"], g["N"], bias=False, dtype=torch.float64)
    lin.weight.data = w_kn.T.contiguous()
    zeros = torch.from_numpy(g["zeros"])
    z_arg = zeros if zeros.dtype == torch.float16 else zeros.to(torch.float32)
    layer.pack(lin, torch.from_numpy(g["scales"]).float().T.contiguous(), z_arg.T.contiguous(), torch.from_numpy(g["g_idx"]).clone())
    if g["bias"] is not None:
        layer.bias = torch.from_numpy(g["bias"]).clone()
    return layer


@pytest.mark.parametrize("name", ort_golden_names())
def test_module_pack_and_unpack_match_reference(name):
    g = load_golden(name)
    layer = _layer_from_golden(g)
    assert layer.qweight.dtype == torch.uint8 and np.array_equal(layer.qweight.numpy(), g["qweight"])
    assert np.array_equal(_bits(layer.qzeros.numpy()), _bits(g["qzeros"]))
    assert np.array_equal(_bits(layer.scales.numpy()), _bits(g["scales_flat"]))
    assert layer.act_order == O.ort_is_act_order(g["g_idx"]) or "actorder" not in name
    w, scales, zeros = layer.unpack()
    assert np.array_equal(_bits(w.numpy()), _bits(g["W_unpack"]))
    assert np.array_equal(_bits(scales.numpy()), _bits(g["scales"]))
    assert np.array_equal(_bits(zeros.numpy().astype(g["zeros"].dtype)), _bits(g["zeros"]))


def test_module_contract_and_dispatch():
    from qllm_amd.modeling.q_layers import QuantLinearORT
    from qllm_amd.utils.modelutils import select_quant_linear
    layer = QuantLinearORT(4, 128, 512, 256, True, dtype=torch.float16)
    sd = layer.state_dict()
    assert sd["qweight"].shape == (256, 4, 64) and sd["qweight"].dtype == torch.uint8
    assert sd["qzeros"].shape == (4 * 128,) and sd["qzeros"].dtype == torch.uint8
    assert sd["scales"].shape == (4 * 256,) and sd["g_idx"].shape == (512,) and sd["bias"].shape == (256,)
    assert layer.pack_mode == "ORT" and layer.groupsize == 128
    assert QuantLinearORT(4, 128, 384, 64, False).qzeros.shape == (4 * 32,)  # odd block count: padded to even
    assert QuantLinearORT(4, -1, 256, 64, False).groupsize == 256
    assert select_quant_linear("ORT", 4, "gptq") is QuantLinearORT
    assert select_quant_linear("ORT", 4, "hqq") is QuantLinearORT  # ORT wins over hqq, as in the reference's table
    with pytest.raises(NotImplementedError):
        QuantLinearORT(9, 128, 256, 64, False)
    with pytest.raises(RuntimeError):  # no CPU forward
        layer(torch.zeros(1, 512, dtype=torch.float16))


@pytest.mark.parametrize("name", ort_golden_names())
def test_row_stream_view_is_the_same_layer(name):
    """The cached view the fused kernels stream (blob rows as 32-bit words, transposed) holds exactly the layer's integers,
    scales and zero points in the GPTQ/HQQ row-stream arrangement."""
    g = load_golden(name)
    layer = _layer_from_golden(g)
    qw, scales, zeros = layer.row_stream_view()
    K, N, gs = g["K"], g["N"], g["groupsize"]
    assert qw.dtype == torch.int32 and tuple(qw.shape) == (K // 8, N)
    assert np.array_equal(O.gptq_int_weight(qw.numpy(), 4, K), g["q"])
    assert np.array_equal(_bits(scales.numpy()), _bits(g["scales"]))
    assert np.array_equal(zeros.numpy().astype(np.float32), g["zeros"].astype(np.float32))
    # HQQ-form dequant of the view vs the reference's ORT W: same values up to the different rounding sequence
    gi = g["g_idx"].astype(np.int64)
    w_view = O.dequant("HQQ", qw.numpy(), scales.numpy(), zeros.numpy(), None, 4, gs, K)  # trivial groups
    if not O.ort_is_act_order(g["g_idx"]):
        ref = g["W_unpack"].T.astype(np.float32)
        assert np.max(np.abs(w_view.astype(np.float32) - ref)) <= 3 * 2.0 ** -10 * np.max(np.abs(ref)) / 8 + 1e-4
        y = O.matmul_f16(g["x"], w_view, g["bias"])
        assert O.rel_err(y, g["y"]) <= 2e-3
    else:
        assert gi.max() == K // gs - 1


def test_repack_between_ort_and_the_other_layouts_is_integer_exact():
    from qllm_amd.repack import repack_layer
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, QuantLinearORT
    g = load_golden("ort_w4_g128_bias")
    ort = _layer_from_golden(g)
    gptq = repack_layer(ort, "GPTQ")
    assert isinstance(gptq, QuantLinearGPTQ)
    eq, ez = O.pack_gptq(g["q"], g["zeros"], 4)
    assert np.array_equal(gptq.qweight.numpy(), eq) and np.array_equal(gptq.qzeros.numpy(), ez)
    assert np.array_equal(_bits(gptq.scales.numpy()), _bits(g["scales"])) and torch.equal(gptq.bias, ort.bias)
    back = repack_layer(gptq, "ORT")
    assert isinstance(back, QuantLinearORT)
    assert torch.equal(back.qweight, ort.qweight) and torch.equal(back.qzeros, ort.qzeros) and torch.equal(back.scales, ort.scales)
    awq = repack_layer(ort, "GEMM")
    aq, az = O.pack_awq(g["q"], g["zeros"])
    assert np.array_equal(awq.qweight.numpy(), aq) and np.array_equal(awq.qzeros.numpy(), az)
    # real-valued zero points travel between HQQ and ORT
    gf = load_golden("ort_w4_g64_f16zeros")
    ortf = _layer_from_golden(gf)
    hqq = repack_layer(ortf, "HQQ")
    assert isinstance(hqq, QuantLinearHQQ) and np.array_equal(_bits(hqq.qzeros.numpy()), _bits(gf["zeros"]))
    assert np.array_equal(O.gptq_int_weight(hqq.qweight.numpy(), 4, gf["K"]), gf["q"])
    backf = repack_layer(hqq, "ORT")
    assert torch.equal(backf.qweight, ortf.qweight) and np.array_equal(_bits(backf.qzeros.numpy()), _bits(ortf.qzeros.numpy()))
    with pytest.raises(ValueError):
        repack_layer(ortf, "GPTQ")  # fp16 zeros cannot be stored as packed GPTQ zeros


def test_ort_checkpoint_round_trip_and_model_repack(tmp_path):
    """version=ORT checkpoints load into QuantLinearORT (uint8 blobs survive safetensors) and convert to the other modes."""
    import json
    import os
    import transformers
    from qllm_amd.modeling import base
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearORT
    from qllm_amd.repack import repack_to_new_mode
    from qllm_amd.utils import modelutils
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=1, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=128, max_position_embeddings=64, torch_dtype="float16",
                                   tie_word_embeddings=False)
    torch.manual_seed(0)
    model = transformers.LlamaForCausalLM(cfg).half()
    names = [n for n in modelutils.find_layers(model, [torch.nn.Linear]) if n != "lm_head"]
    qcfg = base.QuantConfig(bits=4, group_size=128, version="ORT", quant_method="gptq")
    assert base.swap_quantized_linears(model, names, qcfg) is QuantLinearORT
    rng = np.random.default_rng(2)
    truth = {}
    for n, layer in modelutils.find_layers(model, [QuantLinearORT]).items():
        K, N = layer.infeatures, layer.outfeatures
        q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
        z = rng.integers(0, 16, size=(K // 128, N), dtype=np.int32)
        s = (rng.random((K // 128, N)) * 0.004 + 0.001).astype(np.float16)
        qw, qz, sf = O.pack_ort(q, z, s)
        layer.qweight, layer.qzeros, layer.scales = torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(sf)
        truth[n] = (q, z, s)
    model.quant_config = qcfg
    d = str(tmp_path / "ort")
    base.save_quantized(model, d)
    assert json.load(open(os.path.join(d, "quantize_config.json")))["version"] == "ORT"
    loaded = base.load_quantized(d, device=None)
    assert loaded.load_report["quantized_layers"] == len(names) == 7 and not loaded.load_report["unexpected_keys"]
    a, b = model.state_dict(), loaded.state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    lay = loaded.model.layers[0].mlp.down_proj
    assert isinstance(lay, QuantLinearORT) and lay.qweight.dtype == torch.uint8
    w_before = lay.unpack()[0]
    repack_to_new_mode(loaded, "GPTQ")
    lay2 = loaded.model.layers[0].mlp.down_proj
    assert isinstance(lay2, QuantLinearGPTQ) and loaded.quant_config.version == "GPTQ"
    q, z, s = truth["model.layers.0.mlp.down_proj"]
    eq, ez = O.pack_gptq(q, z, 4)
    assert np.array_equal(lay2.qweight.numpy(), eq) and np.array_equal(lay2.qzeros.numpy(), ez)
    # same integers, scales and zeros; the two modules' W differ only by their rounding sequences (s*q - s*z vs (q-z)*s)
    w_after = lay2.unpack()[0]
    assert float((w_after.float() - w_before.float()).abs().max()) <= 2.0 ** -9 * float(w_before.float().abs().max())
