"""-m gpu: the reference's `--eval` steps on a quantized model whose linears run on the MI355X kernels:
llama.cpp-style perplexity (qllm/plugin/perplexity_utils.py:97-201) and the 50-token generate smoke
(qllm/auto_model_quantization.py:59-76), against the same model evaluated on the CPU from its dequantised weights."""
import copy
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class _Tok:
    """What the two helpers need from a tokenizer (no tokenizer files offline): fixed ids for any text."""
    bos_token_id = 1
    eos_token_id = 2
    model_max_length = 0

    def __init__(self, ids):
        self.ids = ids

    def __call__(self, text, truncation=False, return_tensors="pt"):
        out = {"input_ids": self.ids.clone(), "attention_mask": torch.ones_like(self.ids)}
        return _Batch(out)

    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


class _Batch(dict):
    def to(self, device):
        return _Batch({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})

    @property
    def input_ids(self):
        return self["input_ids"]


@pytest.mark.parametrize("pack_mode", ["GPTQ", "GEMM"])
def test_perplexity_and_generate_on_quantized_model(tmp_path, pack_mode):
    from test_loader_repack_cpu import _quantize_in_place, _tiny_llama
    from qllm_amd.modeling import base
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, WQLinear_GEMM
    from qllm_amd.plugin.perplexity_utils import Perplexity, generate_smoke
    from qllm_amd.utils import modelutils

    model, _ = _quantize_in_place(_tiny_llama(), pack_mode)
    d = str(tmp_path / pack_mode)
    base.save_quantized(model, d)
    ref = copy.deepcopy(model)                       # CPU truth: every q_layer -> nn.Linear(unpack()[0]), float32
    for n, layer in modelutils.find_layers(ref, [QuantLinearGPTQ, WQLinear_GEMM]).items():
        lin = torch.nn.Linear(layer.infeatures, layer.outfeatures, bias=False)
        lin.weight.data = layer.unpack()[0].float()
        modelutils.set_op_by_name(ref, n, lin)
    ref = ref.float().eval()
    loaded = base.load_quantized(d, device=DEV)
    assert loaded.sibling_groups == 4                 # 2 layers x (q/k/v, gate/up)

    tokens = torch.randint(3, 128, (1, 3 * 32 + 5), generator=torch.Generator().manual_seed(3))
    ppl_gpu = Perplexity(loaded, tokens=tokens, bos_token_id=1).calculate_perplexity(n_ctx=32, n_batch=32)
    ppl_cpu = Perplexity(ref, tokens=tokens, bos_token_id=1).calculate_perplexity(n_ctx=32, n_batch=32)
    assert len(ppl_gpu) == 3 and np.allclose(ppl_gpu, ppl_cpu, rtol=2e-2), (ppl_gpu, ppl_cpu)

    # generate smoke: greedy decode runs the decode-sized kernels (M = 1) step by step through the sibling groups
    prompt = torch.tensor([[1, 17, 33, 5, 90, 41, 7]])
    text_gpu = generate_smoke(loaded, _Tok(prompt), max_length=24)
    text_cpu = generate_smoke(ref, _Tok(prompt), max_length=24)
    a, b = text_gpu.split(), text_cpu.split()
    assert len(a) == len(b) == 24 and a[:7] == b[:7]
    agree = sum(x == y for x, y in zip(a, b)) / len(a)
    assert agree >= 0.7, (text_gpu, text_cpu)         # fp16 kernels vs fp32 CPU on a random model: near-ties may flip late tokens
    groups = {id(m._siblings): m._siblings for m in loaded.modules() if getattr(m, "_siblings", None) is not None}
    assert len(groups) == 4 and all(g.grouped_launches > 0 for g in groups.values())
