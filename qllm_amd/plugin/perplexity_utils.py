"""llama.cpp-style perplexity and the 50-token generate smoke of the reference's `--eval`
(qllm/plugin/perplexity_utils.py:10-223, qllm/auto_model_quantization.py:59-76), for models whose linears run on
this library.

Same class name, constructor arguments and `calculate_perplexity(n_ctx, n_batch)` contract as the reference: the text is
cut in windows of `n_ctx` tokens, the first token of every window is replaced by BOS for the forward pass, and the
tokens in the SECOND half of each window (positions min(512, n_ctx/2) .. n_ctx-2) are scored; the running
exp(mean NLL) after every window is returned.

Differences, all outside the arithmetic being measured:
  * there is no network here, so besides `dataset_path` (tried through `datasets` exactly like the reference) the text or
    the token ids can be handed in directly (`text=` / `tokens=`);
  * the per-token numpy softmax loop is one log_softmax + gather on the model's device (float32), identical to fp32
    round-off.
"""
from __future__ import annotations

import sys
from typing import List, Optional

import numpy as np
import torch


class Perplexity:
    def __init__(self, model, tokenizer=None, dataset_path="wikitext", dataset_name=None, split="test", text_column="text",
                 text: Optional[str] = None, tokens: Optional[torch.Tensor] = None, bos_token_id: Optional[int] = None):
        self._model = model
        self._tokenizer = tokenizer
        self._dataset_path = dataset_path
        self._dataset_name = dataset_name
        self._split = split
        self._text_column = text_column
        self._tokens = tokens
        self._bos = bos_token_id if bos_token_id is not None else getattr(tokenizer, "bos_token_id", None)
        self._text = text if (text is not None or tokens is not None) else self._prepare_data()

    def _prepare_data(self) -> str:
        if self._dataset_path == "wikitext":
            self._dataset_name = "wikitext-2-raw-v1"
        from datasets import load_dataset  # needs the dataset on disk / a network; pass text= or tokens= otherwise

        data = load_dataset(self._dataset_path, self._dataset_name, split=self._split)
        return "".join(" \n" if s == "" else s for s in data[self._text_column])

    def _token_ids(self) -> torch.Tensor:
        if self._tokens is not None:
            t = self._tokens
            return (t if t.dim() == 2 else t.unsqueeze(0)).clone()
        self._tokenizer.model_max_length = sys.maxsize
        return self._tokenizer(self._text, truncation=False, return_tensors="pt").input_ids

    @torch.no_grad()
    def calculate_perplexity(self, n_ctx: int = 512, n_batch: int = 512) -> List[float]:
        device = next(self._model.parameters()).device if hasattr(self._model, "parameters") else self._model.device
        tokens = self._token_ids().to(device)
        if self._bos is None:
            raise ValueError("a BOS token id is needed (tokenizer.bos_token_id or bos_token_id=)")
        nll, count, out = 0.0, 0, []
        first = min(512, n_ctx // 2)
        for i in range(tokens.shape[1] // n_ctx):
            start = i * n_ctx
            window = tokens[:, start:start + n_ctx].clone()
            window[0, 0] = self._bos
            # the reference scores only logits[0] (the first n_batch chunk); with n_batch >= n_ctx that is the whole window
            chunk = window[:, :min(n_ctx, n_batch)]
            logp = torch.log_softmax(self._model(chunk).logits[0].float(), dim=-1)
            last = min(n_ctx - 1, chunk.shape[1])
            if last > first:
                pos = torch.arange(first, last, device=device)
                target = tokens[0, start + pos + 1]
                nll -= float(logp[pos, target].double().sum())
                count += int(pos.numel())
            out.append(float(np.exp(nll / max(count, 1))))
        return out


@torch.no_grad()
def generate_smoke(model, tokenizer, prompt: str = "compared with awq, gptq is", max_length: int = 50) -> str:
    """The reference's post-load sanity generation (auto_model_quantization.py:59-76)."""
    inputs = tokenizer(prompt, return_tensors="pt").to(next(model.parameters()).device)
    inputs["pad_token_id"] = tokenizer.eos_token_id
    out = model.generate(**inputs, max_length=max_length)
    return tokenizer.decode(out[0])
