#!/usr/bin/env python3
"""In-kernel timeline of a chained decode step (no profiler in the way): every chained launch records 100 MHz timestamps of its
first and last block (qllm_debug_timeline).  Prints, per launch and relative to the step's first timestamp, in microseconds:
entry / weight loads issued / input complete / exit of block 0 and of the last block.
Usage: python tools/chain_timeline.py [layers=4] [graph=1] [fused=1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qllm_amd import _lib, ops  # noqa: E402
from qllm_amd.modeling.q_layers import WQLinear_GEMM  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
use_graph = int(sys.argv[2]) if len(sys.argv) > 2 else 1
fused = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
stack = bench.Stack(WQLinear_GEMM, layers, dev, seed=1, fused=bool(fused))
h0 = torch.randn(1, bench.HIDDEN, device=dev, dtype=torch.float16)
chain = ops.DecodeChain(dev)
step = bench.decode_step_fn(stack, h0, chain)
for _ in range(2):
    step()
torch.cuda.synchronize()
n = layers * (4 if fused else 7)
buf = torch.zeros(n * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.qllm_debug_timeline(buf.data_ptr(), n)
if use_graph:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    lib.qllm_debug_timeline(None, 0)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"graph replay: {e0.elapsed_time(e1) / 20 * 1e3 / layers:.2f} us per layer")
else:
    step()
    lib.qllm_debug_timeline(None, 0)
    torch.cuda.synchronize()
chain.check()
t = buf.cpu().view(n, 2, 4).double() / 100.0  # us
t0 = t[t > 0].min()
names = (["qkv", "o", "gate/up", "down"] if fused else ["q", "k", "v", "o", "gate", "up", "down"])
print("launch      |  block 0: entry  issued  x_ok    exit  | last block: entry  issued  x_ok    exit")
for i in range(n):
    a, b = (t[i, 0] - t0).tolist(), (t[i, 1] - t0).tolist()
    print(f"{i:3d} {names[i % len(names)]:8s}|  " + " ".join(f"{v:7.2f}" for v in a) + "  |  " + " ".join(f"{v:7.2f}" for v in b))
