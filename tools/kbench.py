#!/usr/bin/env python3
"""Developer micro-benchmark: per-shape kernel time of the fused path with HBM-resident (rotating) weights.
Usage: python tools/kbench.py [--m 1 16 2048] [--layouts GEMM GPTQ HQQ] [--iters 200]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM  # noqa: E402

SHAPES = [(4096, 4096), (4096, 11008), (11008, 4096)]
# total bytes of the rotating weight copies: 640 MB = beyond the 256 MB MALL (HBM-cold); ~100 MB = MALL-warm but L2-cold
FOOTPRINT_MB = int(os.environ.get("KBENCH_FOOTPRINT_MB", "640"))
CLS = {"GPTQ": QuantLinearGPTQ, "GEMM": WQLinear_GEMM, "HQQ": QuantLinearHQQ}


def rand_layer(layout, K, N, g, dev, act_order=False):
    layer = CLS[layout](4, g, K, N, False, dtype=torch.float16)
    gen = torch.Generator().manual_seed(K + N)
    layer.qweight = torch.randint(-2**31, 2**31 - 1, layer.qweight.shape, generator=gen, dtype=torch.int32)
    if layout == "HQQ":
        layer.qzeros = (torch.rand(layer.qzeros.shape, generator=gen) * 15).half()
    else:
        layer.qzeros = torch.randint(-2**31, 2**31 - 1, layer.qzeros.shape, generator=gen, dtype=torch.int32)
    layer.scales = (torch.rand(layer.scales.shape, generator=gen) * 0.01 + 0.002).half()
    if act_order:
        layer.g_idx = layer.g_idx[torch.randperm(K, generator=gen)].contiguous()
    return layer.to(dev)


def alg_bytes(K, N, g, M, layout, act):
    G = (K + g - 1) // g
    z = G * N * 2 if layout == "HQQ" else G * N // 2
    return K * N // 2 + G * N * 2 + z + (4 * K if act else 0) + 2 * M * K + 2 * M * N


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[1, 16, 2048])
    ap.add_argument("--layouts", nargs="+", default=["GEMM", "GPTQ"])
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--g", type=int, default=128)
    ap.add_argument("--act", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for layout in args.layouts:
        for (K, N) in SHAPES:
            wbytes = K * N // 2
            ncopy = max(2, min(48, (FOOTPRINT_MB << 20) // wbytes))
            layers = [rand_layer(layout, K, N, args.g, dev, args.act and layout == "GPTQ") for _ in range(ncopy)]
            for M in args.m:
                x = torch.randn(M, K, device=dev, dtype=torch.float16)
                iters = args.iters if M <= 64 else max(20, args.iters // 10)
                for i in range(min(ncopy, 4)):
                    layers[i](x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(iters):
                    layers[i % ncopy](x)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
                # graph replay removes the python launch overhead from the picture
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    for i in range(ncopy):
                        layers[i](x)
                    torch.cuda.synchronize()
                    with torch.cuda.graph(g, stream=s):
                        for i in range(ncopy):
                            layers[i](x)
                torch.cuda.current_stream().wait_stream(s)
                g.replay()
                torch.cuda.synchronize()
                reps = max(2, iters // ncopy)
                e0.record()
                for _ in range(reps):
                    g.replay()
                e1.record()
                torch.cuda.synchronize()
                usg = e0.elapsed_time(e1) * 1e3 / (reps * ncopy)
                b = alg_bytes(K, N, args.g, M, layout, args.act)
                fl = 2.0 * M * K * N
                print(f"{layout:5s} K={K:5d} N={N:5d} M={M:5d}  eager {us:8.2f} us  graph {usg:8.2f} us/launch  "
                      f"{b / usg / 1e3:7.1f} GB/s ({b / usg / 1e3 / 8000 * 100:4.1f}% of 8TB/s)  {fl / usg / 1e6:8.1f} TFLOP/s", flush=True)
            del layers
            torch.cuda.empty_cache()


def grouped():
    """q/k/v (3 x 4096x4096) and gate/up (2 x 4096x11008) as single grouped launches, M=1."""
    from qllm_amd import ops
    dev = torch.device("cuda:0")
    for name, shapes in (("qkv", [(4096, 4096)] * 3), ("gate_up", [(4096, 11008)] * 2), ("o", [(4096, 4096)]), ("down", [(11008, 4096)])):
        nbytes = sum(alg_bytes(K, N, 128, 1, "GPTQ", False) for K, N in shapes)
        ncopy = max(2, min(32, (FOOTPRINT_MB << 20) // nbytes))
        sets = [[rand_layer("GPTQ", K, N, 128, dev) for (K, N) in shapes] for _ in range(ncopy)]
        x = torch.randn(1, shapes[0][0], device=dev, dtype=torch.float16)
        descs = [[l.decode_descriptor() for l in s] for s in sets]
        for d in descs[:3]:
            ops.linear_forward_grouped(d, x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for d in descs:
                ops.linear_forward_grouped(d, x)
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for d in descs:
                    ops.linear_forward_grouped(d, x)
        torch.cuda.current_stream().wait_stream(s)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (10 * ncopy)
        print(f"grouped {name:8s} {nbytes / 1e6:6.1f} MB  {us:7.2f} us/launch  {nbytes / us / 1e3:7.1f} GB/s", flush=True)
        del sets, descs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    if "--grouped" in sys.argv:
        grouped()
    else:
        main()
