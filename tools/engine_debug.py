#!/usr/bin/env python3
"""Debug: single layers through the decode engine vs the ordinary launch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import synth, to_layer, randx
from qllm_amd import ops
DEV = "cuda:0"
for (K, N) in ((4096, 4096), (768, 4096), (4096, 11008), (11008, 4096), (4096, 128), (256, 64)):
    d = synth("GPTQ", 4, 128, K, N, seed=K + N)
    layer = to_layer(d, DEV)
    x = torch.from_numpy(randx(1, K, seed=3)).to(DEV)
    ref = layer(x)
    torch.cuda.synchronize()
    ch = ops.DecodeChain(DEV, mode="engine")
    with ch:
        y = layer(x)
    torch.cuda.synchronize()
    err = int(ch.err.item())
    yf, rf = y.float(), ref.float()
    bad = ~torch.isfinite(yf)
    print(f"K={K} N={N} links={ch.links} fallbacks={ch.fallbacks} err={err} nan={int(bad.sum())} "
          f"maxdiff={float((yf - rf)[~bad].abs().max()) if (~bad).any() else -1:.4g} refmax={float(rf.abs().max()):.4g} "
          f"first_bad_cols={bad.nonzero()[:6, 1].tolist()}")
