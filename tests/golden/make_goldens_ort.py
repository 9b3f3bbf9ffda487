#!/usr/bin/env python3
"""Mint golden vectors for the ORT / MatMulNBits blob layout from the REFERENCE's own
Python, in the build container only:   python tests/golden/make_goldens_ort.py

Same rules as make_goldens.py: /root/reference is imported read-only with the two in-process shims, layers are built
from exact integers, packed with the reference's ``pack()``, and only DATA (inputs + expected outputs) is written.

Fixture fields:  layout="ORT", bits=4, groupsize, K, N, compat=0, q [K,N] int, zeros [G,N] (int or f16), scales [G,N] f16,
  g_idx [K] i32, bias [N] f16 or empty, x [33,K] f16,
  qweight u8 [N, K/g, g/2], qzeros (u8 flat nibble pairs per row, or f16 [N,G]), scales_flat f16 [N*G]  (reference pack()),
  W_unpack [N,K] f16 = layer.unpack()[0] (= dequantize_blockwise_4bits), y [33,N] = layer.forward(x), y1 = forward(x[:1]).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import import_reference  # noqa: E402

CASES = [
    # name, g, K, N, zero_kind, act_order, bias
    ("ort_w4_g128_asym", 128, 256, 128, "asym", False, False),
    ("ort_w4_g128_bias", 128, 512, 128, "asym", False, True),  # (odd block counts: the reference's own CPU dequant cannot reshape them)
    ("ort_w4_g32_actorder", 32, 256, 192, "asym", True, False),
    ("ort_w4_g64_f16zeros", 64, 256, 128, "f16", False, False),
]


def make_case(cls, name, g, K, N, zero_kind, act_order, has_bias, seed):
    gen = torch.Generator().manual_seed(seed)
    G = K // g
    q = torch.randint(0, 16, (K, N), generator=gen, dtype=torch.int32)
    scales = (torch.rand((G, N), generator=gen) * 0.010 + 0.002).to(torch.float16)
    if zero_kind == "asym":
        zeros = torch.randint(0, 16, (G, N), generator=gen, dtype=torch.int32)
    else:
        zeros = (torch.rand((G, N), generator=gen) * 15).to(torch.float16)
    if act_order:
        g_idx = (torch.arange(K) // g)[torch.randperm(K, generator=gen)].to(torch.int32)
        if int(g_idx[:32].sum()) == 0 or int(g_idx[: g // 4].sum()) == 0:
            g_idx[0] = G - 1
    else:
        g_idx = (torch.arange(K) // g).to(torch.int32)
    bias = (torch.randn(N, generator=gen) * 0.5).to(torch.float16) if has_bias else None
    x = torch.randn((33, K), generator=gen).to(torch.float16)
    gi = g_idx.long()
    w_kn = scales.double()[gi] * (q.double() - zeros.double()[gi])
    lin = torch.nn.Linear(K, N, bias=has_bias, dtype=torch.float64)
    lin.weight.data = w_kn.T.contiguous()
    layer = cls(4, g, K, N, has_bias, dtype=torch.float16)
    z_arg = zeros if zero_kind == "f16" else zeros.to(torch.float32)
    layer.pack(lin, scales.float().T.contiguous(), z_arg.T.contiguous(), g_idx.clone())
    if has_bias:
        layer.bias = bias.clone()
    out = dict(layout="ORT", bits=4, groupsize=g, K=K, N=N, compat=0,
               q=q.numpy(), zeros=zeros.numpy(), scales=scales.numpy(), g_idx=layer.g_idx.numpy().astype(np.int32),
               bias=(bias.numpy() if has_bias else np.zeros((0,), np.float16)), x=x.numpy(),
               qweight=layer.qweight.numpy(), qzeros=layer.qzeros.numpy(), scales_flat=layer.scales.numpy())
    with torch.no_grad():
        out["W_unpack"] = layer.unpack()[0].to(torch.float16).numpy()
        out["y"] = layer(x).numpy()
        out["y1"] = layer(x[:1]).numpy()
    return out


def main():
    import_reference()
    from qllm.modeling.q_layers.quant_linear_onnxruntime import QuantLinearORT
    for i, case in enumerate(CASES):
        data = make_case(QuantLinearORT, *case, seed=4321 + i)
        path = os.path.join(HERE, case[0] + ".npz")
        np.savez_compressed(path, **data)
        print(f"{case[0]:32s} {os.path.getsize(path) / 1024:8.1f} KiB  y.absmax={np.abs(data['y']).max():.3f} "
              f"qweight{tuple(data['qweight'].shape)} qzeros{tuple(data['qzeros'].shape)}:{data['qzeros'].dtype}")


if __name__ == "__main__":
    main()
