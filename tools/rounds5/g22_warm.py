#!/usr/bin/env python3
"""round 5, call 22: do the extra legs of bench.py see a clock ramp?  The same captured graph timed three times back to back
(20, 20, 100 replays) for the configs[3] 4-bit stack and the TP = 8 shard stack.  Informational (bench.py is not changed by it)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearHQQ  # noqa: E402
from tools import tp_bench  # noqa: E402

dev = torch.device("cuda:0")
x16 = torch.randn(16, bench.HIDDEN, device=dev, dtype=torch.float16)
hs = bench.Stack(QuantLinearHQQ, 32, dev, seed=11, bits=4, group=64)
g, _ = bench.capture(lambda: hs(x16))
print("hqq w4 32 layers, us per layer:", [round(bench.time_events(g.replay, n) * 1e3 / 32, 2) for n in (20, 20, 100, 20)], flush=True)
del g, hs
torch.cuda.empty_cache()
blocks = tp_bench.build_stack(8, 80, dev, seed=77)
h1 = torch.randn(1, tp_bench.H70, device=dev, dtype=torch.float16)
def fwd():
    h = h1
    for b in blocks:
        h = b(h)
    return h
g, _ = tp_bench._capture(fwd)
print("tp shard 80 layers, us per layer:", [round(tp_bench._time(g.replay, n) * 1e3 / 80, 2) for n in (20, 20, 100, 20)], flush=True)
