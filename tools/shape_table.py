#!/usr/bin/env python3
"""Plan + rate of every launch of a decoder layer for model families OTHER than the two the planner was tuned on (round-5 verdict,
Missing #5: "evidence outside two models' shapes").  AWQ w4 g128 through the q_layer modules with the loader's sibling groups
(q/k/v and gate/up as one grouped launch at decode sizes), native layout, hipGraph replay over rotating weight copies (> 512 MB in
flight: HBM-cold).  Prints a markdown table: plan string (qllm_plan_describe), us per launch, GB/s of the algorithmic bytes at
M = 1 / 16, TFLOP/s at M = 2048, and the fraction of the 8 TB/s / 2.5 PFLOP/s peaks.

    python tools/shape_table.py [--families llama3-8b qwen2-7b ...] [--m 1 16 2048] > profiles/r06_shape_table.md
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from qllm_amd import ops  # noqa: E402
from qllm_amd.modeling.q_layers import WQLinear_GEMM, QuantLinearGPTQ, install_sibling_groups  # noqa: E402

# name: (hidden, intermediate, kv width = n_kv_heads x head_dim, group size)
FAMILIES = {
    "llama2-7b": (4096, 11008, 4096, 128),       # (the tuned one: reference line)
    "llama2-13b": (5120, 13824, 5120, 128),
    "llama3-8b/mistral-7b": (4096, 14336, 1024, 128),
    "qwen2-7b": (3584, 18944, 512, 128),
    "qwen2-1.5b": (1536, 8960, 256, 128),
    "tinyllama-1.1b": (2048, 5632, 256, 128),
    "phi3-mini": (3072, 8192, 3072, 128),
    "llama2-70b": (8192, 28672, 1024, 128),
    "falcon-7b-like(g64)": (4544, 18176, 4544, 64),   # K = 4544 is not a multiple of 128: 64-wide groups (GPTQ row-stream layout)
}
FOOTPRINT = 640 << 20


def launches(block):
    """(name, modules fed by the same input, K) for the four launches of a decoder layer."""
    return (("q/k/v", [block.q_proj, block.k_proj, block.v_proj]), ("o_proj", [block.o_proj]),
            ("gate/up", [block.gate_proj, block.up_proj]), ("down_proj", [block.down_proj]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", nargs="+", default=list(FAMILIES))
    ap.add_argument("--m", type=int, nargs="+", default=[1, 16, 2048])
    ap.add_argument("--replays", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    print("| family | launch | K | N | M | plan | us | GB/s | TFLOP/s | of peak |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for fam in a.families:
        hidden, inter, kv, g = FAMILIES[fam]
        cls = WQLinear_GEMM if g == 128 else QuantLinearGPTQ
        wbytes = (2 * hidden * hidden + 2 * hidden * kv + 3 * hidden * inter) // 2
        ncopy = max(2, min(24, FOOTPRINT // wbytes + 1))
        gen = torch.Generator(device=dev).manual_seed(1)
        blocks = [bench.Block(cls, dev, gen, hidden=hidden, inter=inter, kv=kv, group=g) for _ in range(ncopy)]
        holder = torch.nn.ModuleList(blocks)
        install_sibling_groups(holder, [cls])
        for li in range(4):
            name, mods0 = launches(blocks[0])[li]
            K = mods0[0].infeatures
            Ns = [m.outfeatures for m in mods0]
            for M in a.m:
                x = torch.randn(M, K, device=dev, dtype=torch.float16)

                def step():
                    for b in blocks:
                        for m in launches(b)[li][1]:
                            m(x)
                try:
                    step()   # (builds the native copies / decides the plan)
                    descs = [m.decode_descriptor() for m in mods0]
                    plan = ops.plan_describe(descs, M) if len(descs) > 1 and mods0[0]._siblings is not None else "unsupported"
                    if plan.startswith("unsupported"):   # no grouped launch for this many rows: the layers run one by one
                        plan = (f"{len(descs)} launches: " if len(descs) > 1 else "") + ops.plan_describe(descs[:1], M)
                    gph, _ = bench.capture(step)
                    ms = bench.time_events(gph.replay, a.replays, warm=5) / ncopy
                    del gph
                except Exception as e:  # noqa: BLE001
                    print(f"| {fam} | {name} | {K} | {'+'.join(map(str, Ns))} | {M} | ERROR {type(e).__name__}: {str(e)[:80]} | | | | |", flush=True)
                    continue
                nbytes = sum(bench.alg_bytes(K, N, M, g) for N in Ns)
                flops = sum(2.0 * M * K * N for N in Ns)
                gbps, tf = nbytes / ms / 1e6, flops / ms / 1e9
                frac = tf / bench.MFMA_PEAK_TFLOPS if M >= 256 else gbps / bench.HBM_PEAK_GBPS
                print(f"| {fam} | {name} | {K} | {'+'.join(map(str, Ns))} | {M} | {plan} | {ms * 1e3:.2f} | {gbps:.0f} | {tf:.1f} | {frac:.3f} |", flush=True)
        del blocks, holder
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
