#!/usr/bin/env python3
"""Launch one (layout, K, N, M) problem a few times: the target for `rocprofv3 --pmc` passes and ISA experiments.
`--ref` also times torch.matmul (hipBLASLt) on a dense fp16 weight of the same shape, for context."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.kbench import rand_layer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layout", default="GPTQ")
ap.add_argument("--k", type=int, default=4096)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--m", type=int, default=2048)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--ref", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
layers = [rand_layer(a.layout, a.k, a.n, 128, dev) for _ in range(4)]
x = torch.randn(a.m, a.k, device=dev, dtype=torch.float16)
for l in layers:
    l(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(a.iters):
    layers[i % 4](x)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
print(f"fused {a.layout} M={a.m} K={a.k} N={a.n}: {us:.1f} us  {2.0 * a.m * a.k * a.n / us / 1e6:.0f} TFLOP/s")
if a.ref:
    ws = [torch.randn(a.k, a.n, device=dev, dtype=torch.float16) * 0.02 for _ in range(4)]
    for w in ws:
        x @ w
    torch.cuda.synchronize()
    e0.record()
    for i in range(a.iters):
        x @ ws[i % 4]
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"torch.matmul fp16 dense M={a.m} K={a.k} N={a.n}: {us:.1f} us  {2.0 * a.m * a.k * a.n / us / 1e6:.0f} TFLOP/s")
