#!/usr/bin/env python3
"""Mint tests/golden/perplexity_ref.json by running the REFERENCE's own Perplexity class
(/root/reference/qllm/plugin/perplexity_utils.py:10-223) in the build container.

The class is imported from the reference tree as it is; only its inputs are synthetic: a tiny random Llama built from a literal
config with a fixed seed (no checkpoints offline) and a stand-in tokenizer that returns fixed token ids (no datasets offline:
`_prepare_data` is bypassed by constructing the object without __init__ and setting the three attributes it reads).
The fixture stores the token ids, the model recipe and the perplexities the reference computed; tests/test_perplexity_cpu.py and
the -m gpu eval test rebuild the same model and must reproduce them.   Usage: python tests/golden/make_goldens_ppl.py"""
import importlib.util
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/qllm/plugin/perplexity_utils.py"

CFG = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
           vocab_size=97, max_position_embeddings=256)
SEED, BOS, N_TOK, N_CTX = 0, 1, 3 * 48 + 11, 48


def tiny_model():
    import transformers
    torch.manual_seed(SEED)
    return transformers.LlamaForCausalLM(transformers.LlamaConfig(**CFG)).eval()


def main():
    if "datasets" not in sys.modules:  # the reference imports it at module level; the offline image may lack the package data
        try:
            import datasets  # noqa: F401
        except Exception:  # noqa: BLE001
            sys.modules["datasets"] = types.SimpleNamespace(load_dataset=None)
    spec = importlib.util.spec_from_file_location("ref_perplexity_utils", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    model = tiny_model()
    tokens = torch.randint(3, CFG["vocab_size"], (1, N_TOK), generator=torch.Generator().manual_seed(1))

    class Tok:  # what calculate_perplexity needs from a tokenizer: __call__ -> .input_ids, .bos_token_id, .model_max_length
        bos_token_id = BOS
        model_max_length = 0

        def __call__(self, text, truncation=False, return_tensors="pt"):
            return types.SimpleNamespace(input_ids=tokens.clone())

    ppl = object.__new__(mod.Perplexity)
    ppl._model, ppl._tokenizer, ppl._text = model, Tok(), "unused"
    model.device  # noqa: B018  (the reference reads model.device)
    out = {"n_ctx": N_CTX, "cases": []}
    for n_batch in (N_CTX, 512):
        vals = [float(v) for v in ppl.calculate_perplexity(N_CTX, n_batch)]
        out["cases"].append({"n_batch": n_batch, "perplexity": vals})
    out.update(config=CFG, seed=SEED, bos=BOS, tokens=tokens[0].tolist(), torch=torch.__version__,
               source="reference qllm/plugin/perplexity_utils.py Perplexity.calculate_perplexity, CPU fp32")
    json.dump(out, open(os.path.join(HERE, "perplexity_ref.json"), "w"), indent=1)
    print(out["cases"])


if __name__ == "__main__":
    main()
