#!/usr/bin/env python3
"""Batches 1..4 (and 8) through the Llama-2-7B decoder stack (AWQ w4 g128, modules, sibling groups, hipGraph): the batch-1 kernel's four-row
forms (round 6) against strip_dma (`QLLM_STRIP1_MAX_M` = 1), us per decoder layer and tokens/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qllm_amd import ops  # noqa: E402
from qllm_amd.modeling.q_layers import WQLinear_GEMM  # noqa: E402

dev = torch.device("cuda:0")
st = bench.Stack(WQLinear_GEMM, 32, dev, seed=3)
for rnd in range(2):
    for M in (1, 2, 3, 4, 8):
        x = torch.randn(M, bench.HIDDEN, device=dev, dtype=torch.float16) * 0.1
        line = f"M={M}:"
        for tag, knob in (("four-row strip1", 4), ("strip_dma", 1)):
            ops.set_knob("QLLM_STRIP1_MAX_M", knob)
            try:
                g, _ = bench.capture(lambda: st(x))
                ms = bench.time_events(g.replay, 20)
                plan = st.blocks[0].q_proj._siblings.describe(M)
                del g
            finally:
                ops.reset_knobs()
            line += f"  {tag}: {ms * 1e3 / 32:6.1f} us per layer, {M * 1e3 / ms:7.0f} tok/s ({plan.split(' grid')[0].split(' form')[0]})"
        print(line, flush=True)
