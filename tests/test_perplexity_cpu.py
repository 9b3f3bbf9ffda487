"""Perplexity helper against a literal restatement of the reference's per-token loop
(qllm/plugin/perplexity_utils.py:97-201) on a tiny random Llama, CPU only (no quantized layers involved)."""
import numpy as np
import pytest
import torch


def _tiny():
    import transformers
    cfg = transformers.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                   num_key_value_heads=4, vocab_size=97, max_position_embeddings=256)
    torch.manual_seed(0)
    return transformers.LlamaForCausalLM(cfg).eval()


def _reference_loop(model, tokens, n_ctx, bos):
    """numpy softmax per position, as the reference does it."""
    nll, count, out = 0.0, 0, []
    for i in range(tokens.shape[1] // n_ctx):
        start = i * n_ctx
        win = tokens[:, start:start + n_ctx].clone()
        win[0, 0] = bos
        with torch.no_grad():
            logits = model(win).logits[0]
        for j in range(min(512, n_ctx // 2), n_ctx - 1):
            lg = logits[j].numpy().astype(np.float64)
            e = np.exp(lg - lg.max())
            p = (e / e.sum())[int(tokens[0, start + j + 1])]
            nll += -np.log(p)
            count += 1
        out.append(float(np.exp(nll / count)))
    return out


def test_perplexity_matches_reference_algorithm():
    from qllm_amd.plugin.perplexity_utils import Perplexity
    model = _tiny()
    tokens = torch.randint(3, 97, (1, 3 * 32 + 7), generator=torch.Generator().manual_seed(1))
    got = Perplexity(model, tokens=tokens, bos_token_id=1).calculate_perplexity(n_ctx=32, n_batch=32)
    want = _reference_loop(model, tokens, 32, 1)
    assert len(got) == 3 and np.allclose(got, want, rtol=1e-5)
    assert got[-1] > 1.0
    with pytest.raises(ValueError):
        Perplexity(model, tokens=tokens).calculate_perplexity(32, 32)  # no BOS id available


def test_perplexity_reproduces_the_reference_class():
    """tests/golden/perplexity_ref.json was minted by running the REFERENCE's Perplexity.calculate_perplexity
    (qllm/plugin/perplexity_utils.py:97-201) on this tiny random Llama (tests/golden/make_goldens_ppl.py): same tokens, same
    model recipe -> same running perplexities."""
    import json
    import os
    import transformers
    from conftest import GOLDEN_DIR
    from qllm_amd.plugin.perplexity_utils import Perplexity
    ref = json.load(open(os.path.join(GOLDEN_DIR, "perplexity_ref.json")))
    torch.manual_seed(ref["seed"])
    model = transformers.LlamaForCausalLM(transformers.LlamaConfig(**ref["config"])).eval()
    tokens = torch.tensor([ref["tokens"]])
    for case in ref["cases"]:
        got = Perplexity(model, tokens=tokens, bos_token_id=ref["bos"]).calculate_perplexity(ref["n_ctx"], case["n_batch"])
        assert np.allclose(got, case["perplexity"], rtol=2e-4), (got, case["perplexity"])
