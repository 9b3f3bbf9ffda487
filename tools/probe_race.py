"""Stress the split-K hand-off: alternate launches with different inputs and look for stale/raced tiles."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import synth, to_layer, randx
from qllm_amd import ops

for layout, K, N, M in (("GEMM", 4096, 11008, 4), ("GEMM", 4096, 4096, 1), ("GPTQ", 4096, 11008, 4), ("GPTQ", 11008, 4096, 1)):
    d = synth(layout, 4, 128, K, N, seed=9)
    layer = to_layer(d)
    xn = randx(M, K); xn[np.abs(xn) < 1e-2] = 1e-2
    x = torch.from_numpy(xn).cuda()
    W = ops.dequant(layer._descriptor(None, 0), x.device).float()
    ref1 = x.float() @ W
    bad_scale = 0; worst = 0.0; nd = 0
    y_first = layer(x).clone()
    for it in range(50):
        y = layer(x)
        y2 = layer(x * 2)
        yb = layer(x)
        e1 = (y.float() - ref1).abs().max().item() / ref1.abs().max().item()
        e2 = (y2.float() - 2 * ref1).abs().max().item() / (2 * ref1.abs().max().item())
        worst = max(worst, e1, e2)
        neq = (y2 != y * 2)
        if neq.any():
            bad_scale += 1
            if bad_scale == 1:
                idx = neq.nonzero()
                print("  first mismatch: count", int(neq.sum()), "cols", idx[:8, 1].tolist(), "rows", idx[:8, 0].tolist(),
                      "maxdiff", float((y2.float() - 2 * y.float()).abs().max()))
        if not torch.equal(y, yb) or not torch.equal(y, y_first):
            nd += 1
    print(f"{layout} K={K} N={N} M={M}: scale-mismatch iters {bad_scale}/50, nondeterministic iters {nd}/50, worst rel err vs fp32 ref {worst:.2e}")
