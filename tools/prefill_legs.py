#!/usr/bin/env python3
"""The three prefill legs of bench.py on their own (4 decoder layers at M = 2048, graph replay: AWQ fp16, GPTQ act-order fp16, AWQ
bf16): the target of `rocprofv3 --kernel-trace --stats` when the question is what the act-order / bf16 paths add to the plain one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearGPTQ, WQLinear_GEMM  # noqa: E402

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # e.g. "awq": one leg (the rocprofv3 per-launch table of the module step)
for tag, cls, act, xdt in (("awq", WQLinear_GEMM, False, torch.float16), ("gptq_actorder", QuantLinearGPTQ, True, torch.float16),
                           ("awq_bf16", WQLinear_GEMM, False, torch.bfloat16), ("awq_bf16_shim", WQLinear_GEMM, False, torch.bfloat16)):
    if only and tag not in only:
        continue
    from qllm_amd import ops
    ps = bench.Stack(cls, 4, dev, seed=99, act_order=act)
    xp = torch.randn(2048, bench.HIDDEN, device=dev, dtype=xdt)
    if tag.endswith("_shim"):   # (round 6: the fp16 conversion pre-pass instead of the native bf16 MFMA form)
        ops.set_knob("QLLM_GEMM3_BF16", 0)
    try:
        gp, _ = bench.capture(lambda: ps(xp))
        ms = bench.time_events(gp.replay, iters)
    finally:
        ops.reset_knobs()
    print(f"{tag}: {ms:.3f} ms per 4 layers, {bench.flops_per_pass(4, 2048) / ms / 1e9:.1f} TFLOP/s", flush=True)
    del gp, ps
