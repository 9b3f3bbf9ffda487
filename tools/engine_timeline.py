#!/usr/bin/env python3
"""In-kernel timeline of one decode-engine launch (csrc/engine.hip, diagnostics instantiation): per link, across the blocks
that own strips of it, when consumer wave 0 reached the link / had its input / finished its last strip, and when the loader
(courier) delivered the link's input vector; per block, how long wave 0 waited for inputs and for slabs.  Microseconds from the earliest stamp.
Usage: python tools/engine_timeline.py [layers=4] [fused=1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qllm_amd import _lib, ops  # noqa: E402
from qllm_amd.modeling.q_layers import WQLinear_GEMM, QuantLinearGPTQ  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fused = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
stack = bench.Stack(QuantLinearGPTQ, layers, dev, seed=1, fused=bool(fused))
h0 = torch.randn(1, bench.HIDDEN, device=dev, dtype=torch.float16)
chain = ops.DecodeChain(dev, mode="engine")
step = bench.decode_step_fn(stack, h0, chain)
for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    step()
e1.record()
torch.cuda.synchronize()
print(f"eager engine step: {e0.elapsed_time(e1) / 20 * 1e3 / layers:.2f} us per layer; links per step {chain.links}")
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    gr.replay()
e1.record()
torch.cuda.synchronize()
print(f"graph replay: {e0.elapsed_time(e1) / 50 * 1e3 / layers:.2f} us per layer")
NB = int(os.environ.get("QLLM_ENGINE_GRID", "256"))
n = layers * 7
buf = torch.zeros((NB * n + 2 * NB) * 4, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.qllm_debug_timeline(buf.data_ptr(), buf.numel() // 8 + 1)
step()
torch.cuda.synchronize()
lib.qllm_debug_timeline(None, 0)
chain.check()
t = buf.cpu().double() / 100.0
per = t[:NB * n * 4].view(NB, n, 4)
tail = t[NB * n * 4:(NB * n + NB) * 4].view(NB, 4)
tail2 = t[(NB * n + NB) * 4:].view(NB, 4)
t0 = per[per > 0].min()
names = ["q", "k", "v", "o", "gate", "up", "down"]
print("link      blocks | enter min  max | x_ok  min   med   max | done  min   max | courier x: min  max | busy(med) = done - x_ok")
for l in range(n):
    own = per[:, l, 2] > 0
    if not own.any():
        print(f"{l:3d} {names[l % 7]:5s} (no stamps)")
        continue
    a = per[own, l, :] - t0
    ld = per[:, l, 3]
    ld = ld[ld > 0] - t0
    lds = f"{ld.min():7.2f} {ld.max():7.2f}" if ld.numel() else "      -       -"
    print(f"{l:3d} {names[l % 7]:5s} {int(own.sum()):4d}   | {a[:, 0].min():7.2f} {a[:, 0].max():7.2f} | {a[:, 1].min():7.2f} {a[:, 1].median():7.2f} "
          f"{a[:, 1].max():7.2f} | {a[:, 2].min():7.2f} {a[:, 2].max():7.2f} | {lds} | {(a[:, 2] - a[:, 1]).median():6.2f}")
print(f"per block (mean / max, us): waited for inputs {tail[:, 0].mean():.1f} / {tail[:, 0].max():.1f}; waited for slabs "
      f"{tail[:, 1].mean():.1f} / {tail[:, 1].max():.1f}; total {tail[:, 2].mean():.1f} / {tail[:, 2].max():.1f}; loader waited for free slots "
      f"{tail[:, 3].mean():.1f} / {tail[:, 3].max():.1f}")
print(f"wave 0 per block (mean, us): LDS read batches {tail2[:, 0].mean():.1f}; arithmetic {tail2[:, 1].mean():.1f}; strip reductions "
      f"{tail2[:, 2].mean():.1f}; once-per-input rewrites {tail2[:, 3].mean():.1f}")
