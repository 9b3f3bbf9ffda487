#!/usr/bin/env python3
"""HQQ g64 (fp16 zero points) decoder stack at batch 1 through the modules, 4 / 3 bits and AWQ g128 beside it (us per decoder layer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearHQQ, WQLinear_GEMM  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(1, bench.HIDDEN, device=dev, dtype=torch.float16)
for name, cls, bits, g in (("awq w4 g128", WQLinear_GEMM, 4, 128), ("hqq w4 g64", QuantLinearHQQ, 4, 64), ("hqq w3 g64", QuantLinearHQQ, 3, 64), ("hqq w4 g128", QuantLinearHQQ, 4, 128)):
    st = bench.Stack(cls, 32, dev, seed=3, bits=bits, group=g)
    gph, _ = bench.capture(lambda: st(x))
    ms = bench.time_events(gph.replay, 20) / 32
    b0 = st.blocks[0]
    print(f"{name} M=1: {ms * 1e3:.1f} us per decoder layer; q/k/v: {b0.q_proj._siblings.describe(1)}; o: {b0.o_proj.decode_descriptor() and __import__('qllm_amd').ops.plan_describe([b0.o_proj.decode_descriptor()], 1)}", flush=True)
    del gph, st
    torch.cuda.empty_cache()
