#!/usr/bin/env python3
"""BASELINE configs[3] on its own (the target of the rocprofv3 passes behind profiles/r03_hqq_summary.md): HQQ g64 fp16 zero
points, batch 16, a stack of decoder layers' linears of each width through the modules (sibling groups, native layout), a few graph
replays.  Usage: python tools/hqq_leg.py [replays=5] [layers=32 (the model's depth, like bench.hqq_leg)] [4,3,by_layer,by_module]
(by_layer / by_module: the stack as ONE mixed 3/4-bit model, bench.MIX_BY_LAYER / MIX_BY_MODULE)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearHQQ  # noqa: E402

replays = int(sys.argv[1]) if len(sys.argv) > 1 else 5
layers = int(sys.argv[2]) if len(sys.argv) > 2 else bench.LAYERS
dev = torch.device("cuda:0")
x16 = torch.randn(16, bench.HIDDEN, device=dev, dtype=torch.float16)
sel = {"4": 4, "3": 3, "by_layer": bench.MIX_BY_LAYER, "by_module": bench.MIX_BY_MODULE}
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["4", "3"]   # (the rocprofv3 summaries of rounds 3-5 are over the two uniform stacks)
for name in which:
    bits = sel[name]
    hs = bench.Stack(QuantLinearHQQ, layers, dev, seed=7 + (bits if isinstance(bits, int) else 5), bits=bits, group=64)
    g, _ = bench.capture(lambda: hs(x16))
    ms = bench.time_events(g.replay, replays) / layers
    b0 = hs.blocks[0]
    print(f"hqq w{name} g64 M=16: {ms * 1e3:.1f} us per decoder layer ({layers}-layer graph); q/k/v: {b0.q_proj._siblings.describe(16)}; gate/up: "
          f"{b0.gate_proj._siblings.describe(16)}", flush=True)
    del g, hs
