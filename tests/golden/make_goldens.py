#!/usr/bin/env python3
"""Mint golden vectors by running the REFERENCE's own Python (CPU path) in the build container.

Run here only:   python tests/golden/make_goldens.py
It imports /root/reference (read-only; absent on the GPU box) with two in-process shims,
builds exact-integer synthetic layers, packs them with the reference's ``pack()``, runs the reference's
``forward`` / ``unpack()``, and writes small ``.npz`` fixtures next to this file.  Only DATA is written:
inputs and expected outputs.  No reference source travels.

Fixture fields:  layout, bits, groupsize, K, N, compat (COMPATIBLE_WITH_AUTOGPTQ at pack+forward time),
  q [K,N] int, zeros [G,N] (int or f16), scales [G,N] f16, g_idx [K] i32, bias [N] f16 or empty,
  qweight, qzeros (as produced by the reference pack()), x [33,K] f16,
  W_fwd [K,N] f16  = DequantizeLinearBlockWise / DequantAndUnpack output (GPTQ / HQQ only),
  W_unpack [N,K] f16 = layer.unpack()[0],
  y [33,N] f16 = layer.forward(x) (GPTQ/HQQ) or F.linear(x, W_unpack) + bias (AWQ: no CPU forward exists),
  y1 [1,N] f16 = same for x[:1],
  qzeros_fixed (compat case only) = qzeros after handle_qzeros_for_autogptq().
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    # shim 1: texttable is imported at module scope by the GPTQ quantizer (not on our path)
    sys.modules.setdefault("texttable", types.SimpleNamespace(Texttable=object))

    # shim 2: has_awq_inference_engine() queries device 0 unguarded at import time
    class _Props:
        major = 0
        minor = 0

    torch.cuda.get_device_properties = lambda *a, **k: _Props()
    sys.path.insert(0, REF)
    from qllm.modeling.q_layers.quant_linear_gptq import QuantLinearGPTQ, DequantizeLinearBlockWise
    from qllm.modeling.q_layers.quant_linear_awq import WQLinear_GEMM
    from qllm.modeling.q_layers.quant_linear_hqq import QuantLinearHQQ, DequantAndUnpack

    return dict(GPTQ=QuantLinearGPTQ, GEMM=WQLinear_GEMM, HQQ=QuantLinearHQQ,
                deq_gptq=DequantizeLinearBlockWise, deq_hqq=DequantAndUnpack)


CASES = [
    # name, layout, bits, g, K, N, zero_kind, act_order, bias, compat
    ("gptq_w4_g128_sym", "GPTQ", 4, 128, 256, 128, "sym", False, False, 0),
    ("gptq_w4_g128_asym", "GPTQ", 4, 128, 256, 128, "asym", False, False, 0),
    ("gptq_w4_g128_actorder", "GPTQ", 4, 128, 512, 128, "asym", True, False, 0),
    ("gptq_w4_g128_opt_bias", "GPTQ", 4, 128, 768, 768, "sym", False, True, 0),
    ("gptq_w4_g128_autogptq", "GPTQ", 4, 128, 256, 128, "asym", False, False, 1),
    ("gptq_w4_g32_actorder_bias", "GPTQ", 4, 32, 256, 192, "asym", True, True, 0),
    ("gptq_w3_g128_asym", "GPTQ", 3, 128, 256, 128, "asym", False, False, 0),
    ("gptq_w3_g64_actorder", "GPTQ", 3, 64, 256, 128, "asym", True, False, 0),
    ("gptq_w2_g64_asym", "GPTQ", 2, 64, 256, 128, "asym", False, False, 0),
    ("gptq_w8_g128_asym", "GPTQ", 8, 128, 256, 128, "asym", False, False, 0),
    ("gptq_w5_g128_asym", "GPTQ", 5, 128, 256, 128, "asym", False, False, 0),
    ("gptq_w6_g128_asym", "GPTQ", 6, 128, 256, 128, "asym", False, False, 0),
    ("gptq_w7_g128_asym", "GPTQ", 7, 128, 256, 128, "asym", False, False, 0),
    ("awq_w4_g128_asym", "GEMM", 4, 128, 256, 128, "asym", False, False, 0),
    ("awq_w4_g64_bias", "GEMM", 4, 64, 256, 256, "asym", False, True, 0),
    ("hqq_w4_g64", "HQQ", 4, 64, 256, 128, "f16", False, False, 0),
    ("hqq_w3_g64", "HQQ", 3, 64, 256, 128, "f16", False, False, 0),
    ("hqq_w2_g64_bias", "HQQ", 2, 64, 256, 128, "f16", False, True, 0),
    ("hqq_w8_g128", "HQQ", 8, 128, 256, 128, "f16", False, False, 0),
]


def make_case(ref, name, layout, bits, g, K, N, zero_kind, act_order, has_bias, compat, seed):
    gen = torch.Generator().manual_seed(seed)
    G = K // g
    maxq = 2 ** bits - 1
    q = torch.randint(0, maxq + 1, (K, N), generator=gen, dtype=torch.int32)
    scales = (torch.rand((G, N), generator=gen) * 0.010 + 0.002).to(torch.float16)
    if zero_kind == "sym":
        zeros = torch.full((G, N), 2 ** (bits - 1), dtype=torch.int32)
    elif zero_kind == "asym":
        zeros = torch.randint(0, maxq + 1, (G, N), generator=gen, dtype=torch.int32)
    else:  # HQQ: non-integer fp16 zeros
        zeros = (torch.rand((G, N), generator=gen) * maxq).to(torch.float16)
    if act_order:
        g_idx = (torch.arange(K) // g)[torch.randperm(K, generator=gen)].to(torch.int32)
        if int(g_idx[:g].sum()) == 0:  # keep the lazy detect meaningful
            g_idx[0] = G - 1
    else:
        g_idx = (torch.arange(K) // g).to(torch.int32)
    bias = (torch.randn(N, generator=gen) * 0.5).to(torch.float16) if has_bias else None
    x = torch.randn((33, K), generator=gen).to(torch.float16)

    # exact-integer construction in float64 so the reference's un-clamped round() is exact
    gi = g_idx.long()
    w_kn = scales.double()[gi] * (q.double() - zeros.double()[gi])
    lin = torch.nn.Linear(K, N, bias=has_bias, dtype=torch.float64)
    lin.weight.data = w_kn.T.contiguous()
    if has_bias:
        lin.bias.data = bias.double()

    os.environ["COMPATIBLE_WITH_AUTOGPTQ"] = str(compat)
    layer = ref[layout](bits, g, K, N, has_bias, dtype=torch.float16)
    z_arg = zeros if zero_kind == "f16" else zeros.to(torch.float32)
    layer.pack(lin, scales.float().T.contiguous(), z_arg.T.contiguous(), g_idx.clone())
    if has_bias:
        layer.bias = bias.clone()
    qweight = layer.qweight.clone()
    qzeros = layer.qzeros.clone()
    assert layer.scales.dtype == torch.float16 and torch.equal(layer.scales, scales)

    out = dict(layout=layout, bits=bits, groupsize=g, K=K, N=N, compat=compat,
               q=q.numpy(), zeros=zeros.numpy(), scales=scales.numpy(), g_idx=layer.g_idx.numpy().astype(np.int32),
               bias=(bias.numpy() if has_bias else np.zeros((0,), np.float16)),
               qweight=qweight.numpy(), qzeros=qzeros.numpy(), x=x.numpy())

    w_unpack = layer.unpack()[0]
    if K * N <= 256 * 1024:  # keep every fixture < 1 MB; the big OPT-shaped case keeps W_fwd only
        out["W_unpack"] = w_unpack.to(torch.float16).numpy()
    with torch.no_grad():
        if layout == "GPTQ":
            layer.act_order = None
            y = layer(x)
            layer.act_order = None
            y1 = layer(x[:1])
            gi_fwd = layer.g_idx if bool(layer.act_order) else None
            out["W_fwd"] = ref["deq_gptq"](layer.qweight, layer.scales, layer.qzeros, g, bits, K, gi_fwd).numpy()
        elif layout == "HQQ":
            y = layer(x)
            y1 = layer(x[:1])
            out["W_fwd"] = ref["deq_hqq"].apply(layer.qweight, layer.scales, layer.qzeros, g, bits, K).numpy()
        else:  # AWQ GEMM has no CPU forward: truth = unpack() + F.linear
            y = torch.nn.functional.linear(x, w_unpack.to(torch.float16), bias)
            y1 = torch.nn.functional.linear(x[:1], w_unpack.to(torch.float16), bias)
    out["y"] = y.numpy()
    out["y1"] = y1.numpy()
    if compat:
        layer.handle_qzeros_for_autogptq()
        out["qzeros_fixed"] = layer.qzeros.numpy()
    os.environ["COMPATIBLE_WITH_AUTOGPTQ"] = "0"
    return out


def main():
    ref = import_reference()
    for i, case in enumerate(CASES):
        data = make_case(ref, *case, seed=1234 + i)
        path = os.path.join(HERE, case[0] + ".npz")
        np.savez_compressed(path, **data)
        print(f"{case[0]:32s} {os.path.getsize(path) / 1024:8.1f} KiB  y.absmax={np.abs(data['y']).max():.3f}")


if __name__ == "__main__":
    main()
