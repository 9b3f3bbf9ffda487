#!/usr/bin/env python3
"""Decode-size timing of 32-wide-group layers (GPTQ w4 g32, Llama-2-7B shapes): the strip kernels on the native layout against the
split-K kernel on the reference layout in place (what served g32 before round 4).  hipGraph replay, HIP events.  GPU box only."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer  # noqa: E402  (random packed layers straight on the device: no oracle involved)
from qllm_amd.modeling.q_layers import QuantLinearGPTQ  # noqa: E402
from qllm_amd import ops  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters // 10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 10 * 10)


for g in (32, 128):
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
        gen = torch.Generator(device=DEV).manual_seed(K + N)
        layer = make_layer(QuantLinearGPTQ, K, N, DEV, gen, group=g)
        nat = layer.native_descriptor(0)
        layer._needs_reference = True
        layer.materialize_reference()
        ref = layer._descriptor(None, 0)
        for m in (1, 4, 16):
            x = torch.from_numpy(np.random.default_rng(m).standard_normal((m, K)).astype(np.float16)).to(DEV)
            y = torch.empty((m, N), dtype=torch.float16, device=DEV)
            t_nat = timed(lambda: ops.linear_forward(nat, x, out=y))
            t_ref = timed(lambda: ops.linear_forward(ref, x, out=y))
            print(f"g{g:<3d} {K:5d}x{N:<5d} M={m:2d}  native {t_nat:6.2f} us [{ops.plan_describe([nat], m)[:60]}]   reference layout in place {t_ref:6.2f} us [{ops.plan_describe([ref], m)[:40]}]", flush=True)
