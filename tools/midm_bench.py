#!/usr/bin/env python3
"""Mid-batch timing (17 <= M <= 256) of one Llama-2-7B linear: the native layout (strips / panel kernel / 256-row tiles, whatever the
dispatcher picks) against the reference layout in place (strips / gemm2).  hipGraph replay over 8 rotating weight sets, HIP events.
GPU box only.   python tools/midm_bench.py [g [only this M [bf16]]]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer  # noqa: E402  (random packed layers straight on the device: no oracle involved)
from qllm_amd.modeling.q_layers import QuantLinearGPTQ  # noqa: E402
from qllm_amd import ops  # noqa: E402

DEV = "cuda:0"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ONLY_M = int(sys.argv[2]) if len(sys.argv) > 2 else 0
DT = torch.bfloat16 if (len(sys.argv) > 3 and sys.argv[3] == "bf16") else torch.float16
SETS = 8


def timed(fns, iters=160):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = max(1, iters // len(fns))
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(fns))


for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
    gen = torch.Generator(device=DEV).manual_seed(K + N)
    layers = [make_layer(QuantLinearGPTQ, K, N, DEV, gen, group=G) for _ in range(SETS)]
    nats = [l.native_descriptor(0) for l in layers]
    for l in layers:
        l._needs_reference = True
        l.materialize_reference()
    refs = [l._descriptor(None, 0) for l in layers]
    for m in ((ONLY_M,) if ONLY_M else (17, 32, 33, 48, 64, 96, 128, 129, 256)):
        x = torch.from_numpy(np.random.default_rng(m).standard_normal((m, K)).astype(np.float16)).to(DEV).to(DT)
        y = torch.empty((m, N), dtype=DT, device=DEV)
        t_nat = timed([(lambda w=w: ops.linear_forward(w, x, out=y)) for w in nats])
        t_ref = timed([(lambda w=w: ops.linear_forward(w, x, out=y)) for w in refs])
        print(f"g{G:<3d} {K:5d}x{N:<5d} M={m:3d}  native {t_nat:6.2f} us [{ops.plan_describe([nats[0]], m)[:44]}]   in place {t_ref:6.2f} us [{ops.plan_describe([refs[0]], m)[:30]}]", flush=True)
