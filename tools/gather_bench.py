#!/usr/bin/env python3
"""Column gather (act-order activation permutation): library kernel vs index_select, us and GB/s (read + write)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qllm_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for (M, K) in ((2048, 4096), (2048, 11008), (8192, 4096), (16, 4096), (1, 4096)):
    x = torch.randn(M, K, device=dev, dtype=torch.float16)
    perm = torch.randperm(K, device=dev).to(torch.int32)
    pl = perm.long()
    res = []
    for fn in (lambda: ops.gather_columns(x, perm), lambda: x.index_select(1, pl)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 50 * 1e3)
    gb = M * K * 4 / 1e3
    print(f"M={M} K={K}: gather kernel {res[0]:.2f} us ({gb / res[0]:.0f} GB/s), index_select {res[1]:.2f} us ({gb / res[1]:.0f} GB/s)")
