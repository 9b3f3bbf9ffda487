"""SURVEY 8(f) rows 1-2 on CPU: checkpoint round trip through a tiny HF Llama, AutoGPTQ normalisation, config parsing,
integer-domain repacking checked against the reference-minted goldens."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu as O
from qllm_amd.modeling import base
from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM
from qllm_amd.repack import repack_layer, repack_to_new_mode
from qllm_amd.utils import modelutils


def _layer_from_golden(name):
    g = load_golden(name)
    cls = {"GPTQ": QuantLinearGPTQ, "GEMM": WQLinear_GEMM, "HQQ": QuantLinearHQQ}[g["layout"]]
    layer = cls(g["bits"], g["groupsize"], g["K"], g["N"], g["bias"] is not None, dtype=torch.float16)
    layer.qweight, layer.qzeros = torch.from_numpy(g["qweight"]), torch.from_numpy(g["qzeros"])
    layer.scales, layer.g_idx = torch.from_numpy(g["scales"]), torch.from_numpy(g["g_idx"])
    if g["bias"] is not None:
        layer.bias = torch.from_numpy(g["bias"])
    return g, layer


def test_repack_gptq_awq_hqq_exact():
    g, gptq = _layer_from_golden("gptq_w4_g128_asym")
    awq = repack_layer(gptq, "GEMM")
    qw, qz = O.pack_awq(g["q"], g["zeros"])
    assert np.array_equal(awq.qweight.numpy(), qw) and np.array_equal(awq.qzeros.numpy(), qz)
    back = repack_layer(awq, "GPTQ")
    assert torch.equal(back.qweight, gptq.qweight) and torch.equal(back.qzeros, gptq.qzeros)
    hqq = repack_layer(gptq, "HQQ")
    assert hqq.qzeros.dtype == torch.float16 and np.array_equal(hqq.qzeros.numpy(), g["zeros"].astype(np.float16))
    # same dequantised weights in every layout
    w0 = gptq.unpack()[0]
    assert torch.equal(awq.unpack()[0], w0) and torch.equal(hqq.unpack()[0], w0)
    ga, awq_gold = _layer_from_golden("awq_w4_g64_bias")
    g2 = repack_layer(awq_gold, "GPTQ")
    eq, ez = O.pack_gptq(ga["q"], ga["zeros"], 4)
    assert np.array_equal(g2.qweight.numpy(), eq) and np.array_equal(g2.qzeros.numpy(), ez)
    assert torch.equal(g2.bias, awq_gold.bias)
    _, act = _layer_from_golden("gptq_w4_g128_actorder")
    with pytest.raises(ValueError):
        repack_layer(act, "GEMM")  # the AWQ layout has no act-order
    _, w3 = _layer_from_golden("gptq_w3_g128_asym")
    with pytest.raises(NotImplementedError):
        repack_layer(w3, "GEMM")


def _tiny_llama():
    import transformers
    cfg = transformers.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=128, max_position_embeddings=64,
                                   torch_dtype="float16", tie_word_embeddings=False)
    torch.manual_seed(0)
    return transformers.LlamaForCausalLM(cfg).half()


def _quantize_in_place(model, pack_mode, compat=0):
    """Swap every decoder linear for a q_layer filled with exact-integer synthetic weights."""
    rng = np.random.default_rng(1)
    names = [n for n in modelutils.find_layers(model, [torch.nn.Linear]) if n != "lm_head"]
    qcfg = base.QuantConfig(bits=4, group_size=128, version=pack_mode, quant_method="awq" if pack_mode == "GEMM" else "gptq",
                            compatible_with_autogptq=bool(compat))
    base.swap_quantized_linears(model, names, qcfg)
    for n, layer in modelutils.find_layers(model, [QuantLinearGPTQ, WQLinear_GEMM]).items():
        K, N = layer.infeatures, layer.outfeatures
        q = rng.integers(0, 16, size=(K, N), dtype=np.int32)
        z = rng.integers(0, 16, size=(K // 128, N), dtype=np.int32)
        qw, qz = (O.pack_awq(q, z) if pack_mode == "GEMM" else O.pack_gptq(q, z, 4, compat))
        layer.qweight, layer.qzeros = torch.from_numpy(qw), torch.from_numpy(qz)
        layer.scales = torch.from_numpy((rng.random((K // 128, N)) * 0.004 + 0.001).astype(np.float16))
    model.quant_config = qcfg
    return model, names


@pytest.mark.parametrize("pack_mode", ["GPTQ", "GEMM"])
def test_checkpoint_round_trip(tmp_path, pack_mode):
    model, names = _quantize_in_place(_tiny_llama(), pack_mode)
    d = str(tmp_path / pack_mode)
    base.save_quantized(model, d)
    assert json.load(open(os.path.join(d, "quantize_config.json")))["version"] == pack_mode
    loaded = base.load_quantized(d, device=None)
    assert loaded.load_report["quantized_layers"] == len(names) == 14 and not loaded.load_report["unexpected_keys"]
    assert isinstance(loaded.lm_head, torch.nn.Linear)  # un-quantized layer detected by the missing .qweight key
    a, b = model.state_dict(), loaded.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    cls = WQLinear_GEMM if pack_mode == "GEMM" else QuantLinearGPTQ
    assert isinstance(loaded.model.layers[1].mlp.down_proj, cls)
    # model-level repack keeps the dequantised weights
    other = "GPTQ" if pack_mode == "GEMM" else "GEMM"
    w_before = loaded.model.layers[0].self_attn.q_proj.unpack()[0]
    repack_to_new_mode(loaded, other)
    assert torch.equal(loaded.model.layers[0].self_attn.q_proj.unpack()[0], w_before)
    assert loaded.quant_config.version == other


def test_autogptq_checkpoint_is_normalised(tmp_path):
    """No `version` key => GPTQ layout with zeros stored minus one; the loader re-packs them (+1) once."""
    model, _ = _quantize_in_place(_tiny_llama(), "GPTQ", compat=1)
    plain, _ = _quantize_in_place(_tiny_llama(), "GPTQ", compat=0)
    d = str(tmp_path / "autogptq")
    base.save_quantized(model, d)
    for f in ("quantize_config.json",):
        json.dump({"bits": 4, "group_size": 128, "desc_act": False}, open(os.path.join(d, f), "w"))
    cfgj = json.load(open(os.path.join(d, "config.json")))
    cfgj.pop("quantization_config", None)
    json.dump(cfgj, open(os.path.join(d, "config.json"), "w"))
    cfg = base.QuantConfig.from_dir(d)
    assert cfg.version == "GPTQ" and cfg.compatible_with_autogptq
    loaded = base.load_quantized(d, device=None)
    for (n1, l1), (n2, l2) in zip(modelutils.find_layers(loaded, [QuantLinearGPTQ]).items(),
                                  modelutils.find_layers(plain, [QuantLinearGPTQ]).items()):
        assert n1 == n2 and torch.equal(l1.qzeros, l2.qzeros) and torch.equal(l1.qweight, l2.qweight)


def test_quant_config_variants(tmp_path):
    d = tmp_path / "awq"
    d.mkdir()
    json.dump({"w_bit": 4, "q_group_size": 64, "version": "gemm", "zero_point": True}, open(d / "quant_config.json", "w"))
    c = base.QuantConfig.from_dir(str(d))
    assert (c.bits, c.group_size, c.version, c.quant_method) == (4, 64, "GEMM", "awq")
    d2 = tmp_path / "hf"
    d2.mkdir()
    json.dump({"quantization_config": {"bits": 3, "group_size": 128, "version": "GPTQ", "quant_method": "gptq"}},
              open(d2 / "config.json", "w"))
    assert base.QuantConfig.from_dir(str(d2)).bits == 3
    with pytest.raises(FileNotFoundError):
        base.QuantConfig.from_dir(str(tmp_path))
