"""-m gpu: BASELINE configs[3] as ONE model -- HQQ g64 with 3- and 4-bit layers mixed per layer, the way the reference builds it
(`make_mixbits_quant_linear`, /root/reference/qllm/utils/modelutils.py:161-181, fed by `quant_config_by_layer.json`,
/root/reference/qllm/modeling/config.py:72-76), loaded through the loader and run at batch 16 (round-5 verdict, Missing #2)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-2


def _bits_of(name: str, pattern: str) -> int:
    """Two ways a mixed-precision recipe assigns widths: by decoder layer (sensitivity per depth) or by module kind."""
    layer = int(name.split("layers.")[1].split(".")[0])
    if pattern == "by_layer":
        return 4 if layer % 2 == 0 else 3
    # by module kind: v_proj, o_proj and down_proj keep 4 bits (the sensitive ones), q / k / gate / up go to 3
    return 4 if name.rsplit(".", 1)[1] in ("v_proj", "o_proj", "down_proj") else 3


def _mixed_hqq_model(pattern):
    from test_loader_repack_cpu import _tiny_llama
    from qllm_amd.modeling import base
    from qllm_amd.modeling.q_layers import QuantLinearHQQ
    from qllm_amd.utils import modelutils
    model = _tiny_llama()
    names = [n for n in modelutils.find_layers(model, [torch.nn.Linear]) if n != "lm_head"]
    cfg = base.QuantConfig(bits=4, group_size=64, version="HQQ", quant_method="hqq")
    cfg.by_layer = {n: {"wbits": _bits_of(n, pattern), "groupsize": 64} for n in names}
    base.swap_quantized_linears(model, names, cfg)
    for i, (n, layer) in enumerate(modelutils.find_layers(model, [QuantLinearHQQ]).items()):
        d = synth("HQQ", layer.bits, 64, layer.infeatures, layer.outfeatures, "f16", False, False, seed=100 + i)
        # (scales sized so that two decoder layers keep activations O(1))
        d["scales"] = (d["scales"].astype(np.float32) * 0.4).astype(np.float16)
        layer.qweight, layer.qzeros, layer.scales = torch.from_numpy(d["qweight"]), torch.from_numpy(d["qzeros"]), torch.from_numpy(d["scales"])
    model.quant_config = cfg
    return model, names


@pytest.mark.parametrize("pattern", ["by_layer", "by_module"])
def test_mixed_bits_checkpoint_loads_and_runs_at_batch_16(tmp_path, pattern):
    from qllm_amd.modeling import base
    from qllm_amd.modeling.q_layers import QuantLinearHQQ
    from qllm_amd.utils import modelutils
    model, names = _mixed_hqq_model(pattern)
    d = str(tmp_path / pattern)
    base.save_quantized(model, d)
    by_layer = json.load(open(os.path.join(d, "quant_config_by_layer.json")))
    assert {by_layer[n]["wbits"] for n in names} == {3, 4}
    # CPU truth: every q_layer -> nn.Linear holding unpack()[0] (the reference's own dequantised weights), float32 math
    ref = copy.deepcopy(model)
    for n, layer in modelutils.find_layers(ref, [QuantLinearHQQ]).items():
        lin = torch.nn.Linear(layer.infeatures, layer.outfeatures, bias=False)
        lin.weight.data = layer.unpack()[0].float()
        modelutils.set_op_by_name(ref, n, lin)
    ref = ref.float().eval()
    ids = torch.randint(0, 128, (16, 6), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        logits_ref = ref(ids).logits

    loaded = base.load_quantized(d, device=DEV)
    got = {n: l.bits for n, l in modelutils.find_layers(loaded, [QuantLinearHQQ]).items()}
    assert got == {n: _bits_of(n, pattern) for n in names}                      # every layer at its own width
    # by_layer: q/k/v and gate/up of a decoder layer agree -> 2 groups per layer; by_module: {q, k} (v alone) and {gate, up}
    assert loaded.sibling_groups == 4
    attn0 = loaded.model.layers[0].self_attn
    if pattern == "by_module":
        assert attn0.q_proj._siblings is attn0.k_proj._siblings and attn0.v_proj._siblings is None
        assert len(attn0.q_proj._siblings.layers) == 2
    else:
        assert len(attn0.q_proj._siblings.layers) == 3
    with torch.no_grad():
        step = loaded(ids[:, :1].to(DEV)).logits.float().cpu()                  # batch 16, one token each: M = 16
        logits = loaded(ids.to(DEV)).logits.float().cpu()                       # M = 96
    assert attn0.q_proj._siblings.grouped_launches >= 1
    assert O.rel_err(step.numpy(), logits_ref[:, :1].numpy()) <= 2e-2           # whole model, fp16 vs fp32, two layers deep
    assert O.rel_err(logits.numpy(), logits_ref.numpy()) <= 2e-2
    assert (logits.argmax(-1) == logits_ref.argmax(-1)).float().mean().item() >= 0.9
    # the mixed checkpoint survives a save -> load round trip bit for bit (released reference buffers are regenerated)
    d2 = str(tmp_path / (pattern + "_again"))
    base.save_quantized(loaded, d2)
    a, b = model.state_dict(), base.load_quantized(d2, device=None).state_dict()
    assert set(a) == set(b) and all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a)


def test_mixed_bits_decoder_layer_at_7b_shapes_batch_16():
    """One Llama-2-7B decoder layer's linears, HQQ g64, q/k/gate/up at 3 bits and v/o/down at 4, batch 16: partial sibling groups
    ({q, k} in one launch, v alone; {gate, up}) give the results of seven single launches (to 1e-3: summation order) and the oracle's within 1e-2."""
    from qllm_amd.modeling.q_layers import install_sibling_groups, QuantLinearHQQ
    H, I = 4096, 11008
    spec = {"q_proj": (H, H, 3), "k_proj": (H, H, 3), "v_proj": (H, H, 4), "o_proj": (H, H, 4),
            "gate_proj": (H, I, 3), "up_proj": (H, I, 3), "down_proj": (I, H, 4)}

    class Blk(torch.nn.Module):
        pass
    blk, data = Blk(), {}
    for i, (n, (K, N, bits)) in enumerate(spec.items()):
        data[n] = synth("HQQ", bits, 64, K, N, "f16", False, False, seed=7 * i + bits)
        setattr(blk, n, to_layer(data[n], DEV))
    assert install_sibling_groups(blk, [QuantLinearHQQ]) == 2
    assert blk.v_proj._siblings is None and blk.q_proj._siblings is blk.k_proj._siblings
    x = {H: torch.from_numpy(randx(16, H, seed=3)).to(DEV), I: torch.from_numpy(randx(16, I, seed=4)).to(DEV)}
    grouped = {n: getattr(blk, n)(x[spec[n][0]]) for n in spec}               # model order: q, k, v, o, gate, up, down
    assert blk.q_proj._siblings.grouped_launches == 1 and blk.gate_proj._siblings.grouped_launches == 1
    assert "strip" in blk.q_proj._siblings.describe(16) and "strip" in blk.gate_proj._siblings.describe(16)
    for n in spec:
        getattr(blk, n)._siblings = None
    for n, (K, N, bits) in spec.items():
        single = getattr(blk, n)(x[K])
        # (a grouped launch may split K over its waves differently from the single one: fp32 summation order, not bit-equality)
        assert O.rel_err(single.cpu().numpy(), grouped[n].cpu().numpy()) <= 1e-3, n
    for n in ("q_proj", "v_proj", "down_proj"):
        ref = Ref(data[n])
        xin = x[spec[n][0]].cpu().numpy()
        assert O.rel_err(grouped[n].cpu().numpy(), ref.y16(xin)) <= TOL, n
        assert O.rel_err(grouped[n].cpu().numpy(), ref.y64(xin)) <= 2e-3, n
