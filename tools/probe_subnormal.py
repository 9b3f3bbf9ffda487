"""One-off probe: does the f16 MFMA flush subnormal fp16 activations?  (explains why y(2x) != 2y(x) bitwise when x has subnormals)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from gpu_util import synth, to_layer
d = synth("GPTQ", 4, 128, 256, 128, seed=1)
layer = to_layer(d)
for val in (3e-5, 6.2e-5, 1e-3):
    x = torch.full((1, 256), val, dtype=torch.float16, device="cuda:0")
    y = layer(x)
    print(f"x={float(x[0,0]):.3e} (subnormal={float(x[0,0]) < 6.1e-5})  |y|max={float(y.abs().max()):.4e}  y/x={float(y[0,0])/float(x[0,0]):.4f}")
