#!/usr/bin/env python3
"""The bit-stream matvec (csrc/bitgemv.hip, round 6) against what served these widths until round 5: the library's dequant kernel + a
dense fp16 GEMM (the reference's branch (B), quant_linear_gptq.py:81-85).  Llama-2-7B shapes, HQQ g64 (HQQ's default widths include 2 and
8) and GPTQ g128, M = 1 / 4 / 16, hipGraph replay over rotating layer copies (HBM-cold).  Prints a markdown table.
    python tools/bitgemv_bench.py > profiles/r06_bitgemv.md"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qllm_amd import ops  # noqa: E402
from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(3)
print("| layout | bits | K | N | M | plan | fused us | GB/s (packed bytes) | of 8 TB/s | dequant + GEMM us | speed-up |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
quick = "--quick" in sys.argv   # (A/B runs: HQQ only, widths 2 / 5 / 8, no dequant + GEMM leg)
for cls, g, zeros in ((QuantLinearHQQ, 64, "f16"),) if quick else ((QuantLinearHQQ, 64, "f16"), (QuantLinearGPTQ, 128, "packed")):
    for bits in (2, 5, 8) if quick else (2, 5, 6, 7, 8):
        for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
            wbytes = K * N * bits // 8
            ncopy = max(2, min(24, (512 << 20) // wbytes + 1))
            layers = [bench.make_layer(cls, K, N, dev, gen, bits=bits, group=g) for _ in range(ncopy)]
            for M in (1, 4, 16):
                x = torch.randn(M, K, device=dev, dtype=torch.float16)
                res = {}
                for tag, on in (("fused", 1),) if quick else (("fused", 1), ("dequant", 0)):
                    ops.set_knob("QLLM_BITGEMV", on)
                    try:
                        plan = ops.plan_describe([layers[0].decode_descriptor()], M) if on else None
                        gph, _ = bench.capture(lambda: [l(x) for l in layers])
                        res[tag] = bench.time_events(gph.replay, 10, warm=3) / ncopy
                        del gph
                        if on:
                            res["plan"] = plan
                    finally:
                        ops.reset_knobs()
                G = K // g
                nbytes = wbytes + G * N * 2 + (G * N * 2 if zeros == "f16" else G * N * bits // 8) + 2 * M * K + 2 * M * N
                res.setdefault("dequant", float("nan"))
                print(f"| {cls.__name__[11:]} g{g} | {bits} | {K} | {N} | {M} | {res['plan']} | {res['fused'] * 1e3:.2f} | {nbytes / res['fused'] / 1e6:.0f} | "
                      f"{nbytes / res['fused'] / 1e6 / 8000:.3f} | {res['dequant'] * 1e3:.2f} | {res['dequant'] / res['fused']:.1f}x |", flush=True)
            del layers
            torch.cuda.empty_cache()
