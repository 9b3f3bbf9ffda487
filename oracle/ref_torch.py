"""Torch-CPU restatement of the reference's CPU forward for 2/4/8-bit GPTQ-layout layers -- the TIMED CPU baseline.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE (same rule as ref_cpu.py: tests/, smoke(), bench.py's cpu_baseline leg only). ***

ref_cpu.py (numpy) is the checker; it is not how the reference spends its time.  What the reference actually executes on a
CPU is a handful of torch tensor ops (/root/reference/qllm/modeling/q_layers/quant_linear_gptq.py:13-52, then :85):
broadcast right-shifts of the packed words against a shift table, a narrowing cast to int8, an AND with the value mask, the
per-group affine map in the scales' dtype with the zero term formed separately, a reshape to [K, N], and one dense
`torch.matmul` in fp16.  SURVEY.md section 8(d) names exactly this formulation as the CPU baseline, so this file restates it
with the same operator sequence and the same intermediate dtypes/shapes (int8 [K/8, 8, N] nibble tensor, fp16 [G, g, N]
products) -- which is what determines its speed -- and tests/test_oracle_golden.py pins its output bit-for-bit to ref_cpu's.
"""
from __future__ import annotations

import torch


def dequant_gptq_torch(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, groupsize: int, bits: int,
                       g_idx: torch.Tensor | None = None, add_zero_bias: int = 0) -> torch.Tensor:
    """W[K, N] in scales.dtype.  qweight i32 [K*bits/32, N], qzeros i32 [G, N*bits/32], scales [G, N]; bits in {2, 4, 8}."""
    assert bits in (2, 4, 8), "the shift/mask branch of the reference (quant_linear_gptq.py:17-28)"
    per_word = 32 // bits
    n = qweight.shape[1]
    narrow = torch.int16 if bits == 8 else torch.int8
    mask = (1 << bits) - 1
    shifts = torch.arange(0, 32, bits, dtype=torch.int32)
    # zero points: [G, N/per_word, 1] >> [1, 1, per_word] -> [G, N/per_word, per_word] -> [G, 1, N]
    z = (qzeros.unsqueeze(2) >> shifts.view(1, 1, per_word)).to(narrow)
    z = ((z + add_zero_bias) & mask).reshape(-1, 1, n)
    # values: [K/per_word, 1, N] >> [1, per_word, 1] -> [K/per_word, per_word, N]
    q = (qweight.unsqueeze(1) >> shifts.view(1, per_word, 1)).to(narrow)
    q &= mask
    s = scales.reshape(-1, 1, n)
    if g_idx is not None:  # act-order: per-input-channel group lookup (quant_linear_gptq.py:38-44)
        s2, z2 = s.squeeze(1), z.squeeze(1)
        zs = z2 * s2
        gi = g_idx.long()
        return s2[gi] * q.reshape(-1, n) - zs[gi]
    zs = (z * s).to(s.dtype)
    w = s * q.reshape(-1, groupsize, n) - zs
    return w.reshape(-1, n)


def forward_gptq_torch(x: torch.Tensor, qweight, scales, qzeros, groupsize: int, bits: int, g_idx=None, bias=None,
                       add_zero_bias: int = 0) -> torch.Tensor:
    """y = x @ W (+ bias): branch (C) of QuantLinearTorchFunction.forward + QuantLinearGPTQ.forward (:83-85, :136-143)."""
    y = torch.matmul(x, dequant_gptq_torch(qweight, scales, qzeros, groupsize, bits, g_idx, add_zero_bias))
    return y + bias if bias is not None else y
