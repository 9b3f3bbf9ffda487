#!/usr/bin/env python3
"""Decode (M small) latency of narrow-N / tensor-parallel shard shapes under the current dispatch knobs:
   QLLM_MI355X_LIB=tools/lab/libqllm_lab.so QLLM_STRIP_MIN=192 python tools/narrow_ab.py   vs   ... QLLM_STRIP_MIN=48 ...
Prints plan + us per launch (graph replay over distinct weight sets, HIP events).  The release library compiles its knobs in:
with a QLLM_* knob in the environment this tool insists on the LAB build (tools/lab/build_lab.sh; qllm_is_lab_build)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qllm_amd import ops  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearGPTQ  # noqa: E402

if "QLLM_STRIP_MIN" in os.environ and not ops._lib.load().qllm_is_lab_build():
    raise SystemExit("QLLM_STRIP_MIN is set but the loaded library is the release build (knobs are compile-time constants there): "
                     "run with QLLM_MI355X_LIB=tools/lab/libqllm_lab.so")
dev = torch.device("cuda:0")
SHAPES = [(8192, [1024, 128, 128]), (8192, [1024]), (8192, [3584, 3584]), (1024, [8192]), (3584, [8192]), (4096, [1024]),
          (4096, [2048]), (4096, [512, 512, 512]), (8192, [8192]), (8192, [28672])]
M = int(os.environ.get("M", "1"))
COPIES = 12


def mk(K, N):
    l = QuantLinearGPTQ(4, 128, K, N, False, dtype=torch.float16)
    l.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, l.qweight.shape, dtype=torch.int32, device=dev)
    l.qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, l.qzeros.shape, dtype=torch.int32, device=dev)
    l.scales = ((torch.rand(l.scales.shape, device=dev) * 0.4 + 0.8) / (K ** 0.5 * 6.5)).half()
    return l.to(dev)


for K, Ns in SHAPES:
    sets = [[mk(K, n) for n in Ns] for _ in range(COPIES)]
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    descs = [[l.decode_descriptor() for l in s] for s in sets]

    def run():
        for d in descs:
            if len(d) == 1:
                ops.linear_forward(d[0], x)
            else:
                ops.linear_forward_grouped(d, x)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (20 * COPIES)
    nbytes = sum(K * n // 2 + (K // 128) * n * 5 // 2 for n in Ns)
    print(f"STRIP_MIN={os.environ.get('QLLM_STRIP_MIN', '192'):>4} M={M} K={K:5d} N={'+'.join(map(str, Ns)):>14}  {us:7.2f} us  "
          f"{nbytes / us / 1e6:6.2f} TB/s  {ops.plan_describe(descs[0], M)}", flush=True)
    del sets, descs, g
    torch.cuda.empty_cache()
