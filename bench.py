#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): W4A16 g128 dequant-GEMM on Llama-2-7B shapes.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched under torch.distributed.run)

Workload (config.workload = "llama2-7b-awq-w4-g128-decode-b1"): BASELINE.json configs[1] -- the 32 x 7 quantized
linears of Llama-2-7B in the AWQ "GEMM" pack mode, w4 g128 asymmetric zeros, batch 1.  ONE STEP = one decode token
through the whole linear stack (224 fused dequant+matvec launches, chained q/k/v -> o -> gate/up -> down so every
launch depends on the previous layer like in the model), replayed from a hipGraph.  Weights are synthetic (no
network: random packed int4 words, random fp16 scales sized to keep activations O(1)) and RESIDENT IN HBM before the
timed region; 3.5 GB of weights per pass means nothing is served from the 256 MB Infinity Cache.

value = decode tokens/s over all ranks (rank r runs an independent replica: batch elements are independent units,
no data-path collective -> "scaling": "weak").

Extra objects on the JSON line:
  roofline     dominant kernel = the decode matvec (one kernel function serves all 224 launches).
               achieved = algorithmic bytes per launch (packed weights + scales + zeros + x + y,
               3,369,484,288 B per token / 224) / average launch duration, the latter measured here with HIP events on
               the launch stream over the K timed steps (ms_per_step / 224; it therefore includes the inter-kernel
               gaps, which rocprofv3's per-kernel average in profiles/ does not).
  cpu_baseline the CPU oracle (oracle/ref_cpu.py: the reference's torch dequant + fp16 matmul restated) timed on this
               host's cores over a bounded sample (one decoder layer's 7 linears), extrapolated x32.
  extra        per-shape decode GB/s and the M=2048 prefill TFLOP/s of the same layers (act-order GPTQ, configs[2]).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HIDDEN, INTER, LAYERS, GROUP = 4096, 11008, 32, 128
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0   # dense f16/bf16 MFMA peak (no sparsity)


def alg_bytes(K, N, M, g=GROUP, zeros="packed", act_order=False):
    G = (K + g - 1) // g
    z = G * N * 2 if zeros == "f16" else G * N // 2
    return K * N // 2 + G * N * 2 + z + (4 * K if act_order else 0) + 2 * M * K + 2 * M * N


def make_layer(cls, K, N, dev, gen, act_order=False):
    layer = cls(4, GROUP, K, N, False, dtype=torch.float16)
    shape_w, shape_z, shape_s = layer.qweight.shape, layer.qzeros.shape, layer.scales.shape
    layer.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, shape_w, dtype=torch.int32, device=dev, generator=gen)
    layer.qzeros = torch.randint(-2 ** 31, 2 ** 31 - 1, shape_z, dtype=torch.int32, device=dev, generator=gen)
    # std(q - z) ~ 6.5 for independent uniform nibbles: scale so each linear roughly preserves magnitude
    base = 1.0 / (K ** 0.5 * 6.5)
    layer.scales = ((torch.rand(shape_s, device=dev, generator=gen) * 0.4 + 0.8) * base).to(torch.float16)
    if act_order:
        layer.g_idx = layer.g_idx[torch.randperm(K)].contiguous()
    return layer.to(dev)


class Stack:
    """The quantized linears of the decoder stack, driven in model order."""

    def __init__(self, cls, n_layers, dev, seed, act_order=False):
        gen = torch.Generator(device=dev).manual_seed(seed)
        self.blocks = []
        for _ in range(n_layers):
            blk = dict(
                q=make_layer(cls, HIDDEN, HIDDEN, dev, gen, act_order), k=make_layer(cls, HIDDEN, HIDDEN, dev, gen, act_order),
                v=make_layer(cls, HIDDEN, HIDDEN, dev, gen, act_order), o=make_layer(cls, HIDDEN, HIDDEN, dev, gen, act_order),
                gate=make_layer(cls, HIDDEN, INTER, dev, gen, act_order), up=make_layer(cls, HIDDEN, INTER, dev, gen, act_order),
                down=make_layer(cls, INTER, HIDDEN, dev, gen, act_order))
            self.blocks.append(blk)

    def forward(self, h, grouped=False):
        for b in self.blocks:
            if grouped:
                from qllm_amd import ops
                q, k, v = ops.linear_forward_grouped([b[n].decode_descriptor() for n in ("q", "k", "v")], h)
                o = b["o"](q)
                gate, up = ops.linear_forward_grouped([b[n].decode_descriptor() for n in ("gate", "up")], o)
            else:
                q = b["q"](h)
                k = b["k"](h)  # noqa: F841  (results feed attention in the real model)
                v = b["v"](h)  # noqa: F841
                o = b["o"](q)
                gate = b["gate"](o)
                up = b["up"](o)  # noqa: F841
            h = b["down"](gate)
        return h

    def launches_per_pass(self, grouped=False):
        return len(self.blocks) * (4 if grouped else 7)


def bytes_per_token(n_layers=LAYERS, M=1):
    return n_layers * (4 * alg_bytes(HIDDEN, HIDDEN, M) + 2 * alg_bytes(HIDDEN, INTER, M) + alg_bytes(INTER, HIDDEN, M))


def flops_per_pass(n_layers, M):
    return n_layers * 2.0 * M * (4 * HIDDEN * HIDDEN + 3 * HIDDEN * INTER)


def capture(fn, warm=2):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def time_events(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters  # ms


def cpu_baseline_leg():
    """Oracle (port of the reference's CPU torch path) on this host's cores, bounded sample: one decoder layer."""
    import numpy as np
    from oracle import ref_cpu as O

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rng = np.random.default_rng(0)
    shapes = [(HIDDEN, HIDDEN)] * 4 + [(HIDDEN, INTER)] * 2 + [(INTER, HIDDEN)]
    layers = []
    for (K, N) in shapes:
        qw = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32)
        qz = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(K // GROUP, N // 8), dtype=np.int64).astype(np.int32)
        sc = ((rng.random((K // GROUP, N)) * 0.4 + 0.8) / (K ** 0.5 * 6.5)).astype(np.float16)
        layers.append((K, N, qw, qz, sc))
    xs = {K: rng.standard_normal((1, K)).astype(np.float16) for K in (HIDDEN, INTER)}

    def one_layer():
        for (K, N, qw, qz, sc) in layers:
            O.forward("GEMM", xs[K], qw, sc, qz, None, None, 4, GROUP, K)

    one_layer()  # warm
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        one_layer()
    per_layer = (time.perf_counter() - t0) / reps
    return dict(value=1.0 / (per_layer * LAYERS), unit="tokens/s", cores=cores, kind="port",
                sample=f"1 of {LAYERS} decoder layers (7 AWQ w4 g128 linears, M=1), {reps} timed passes after 1 warm-up, "
                       f"extrapolated x{LAYERS}; torch threads={torch.get_num_threads()}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grouped", type=int, default=int(os.environ.get("QLLM_BENCH_GROUPED", "1")),
                    help="1: q/k/v and gate/up as single grouped launches (4 launches per layer instead of 7)")
    ap.add_argument("--no-extra", action="store_true", help="skip the per-shape / prefill / CPU legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from qllm_amd import _lib
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, WQLinear_GEMM

    info = _lib.device_info(local_rank)  # raises unless gfx950 + library present: no fallback is ever benchmarked

    grouped = bool(args.grouped)
    stack = Stack(WQLinear_GEMM, LAYERS, dev, seed=1234 + rank)
    h0 = torch.randn(1, HIDDEN, device=dev, dtype=torch.float16)
    graph, out = capture(lambda: stack.forward(h0, grouped))
    assert torch.isfinite(out.float()).all(), "synthetic stack diverged"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        graph.replay()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        graph.replay()
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if world > 1:
        t = torch.tensor([wall, ev_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(t[0]), float(t[1])

    ms_per_step = wall * 1e3 / args.steps
    tokens_per_s = world * args.steps / wall
    launches = stack.launches_per_pass(grouped)
    bpt = bytes_per_token()
    avg_launch_us = ev_ms * 1e3 / (args.steps * launches)
    achieved = (bpt / launches) / (avg_launch_us * 1e-6) / 1e9

    traffic = None
    try:  # HBM traffic per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
        traffic = pmc["bytes_per_launch_avg_over_step"] if grouped else None
    except Exception:  # noqa: BLE001
        pass

    result = {
        "metric": "decode_tokens_per_s_llama2_7b_w4a16_g128_linear_stack", "value": round(tokens_per_s, 2),
        "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": "llama2-7b-awq-w4-g128-decode-b1", "pack_mode": "GEMM", "bits": 4, "group_size": GROUP,
                   "layers": LAYERS, "linears_per_layer": 7, "batch": 1, "launches_per_step": launches,
                   "grouped_qkv_gateup": grouped, "graph": True, "parallelism": f"replicas x{world}",
                   "device": info["arch"], "compute_units": info["compute_units"]},
        "roofline": {"bound": "hbm", "kernel": "qllm::strip_kernel (decode matvec; serves every launch of the step)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": "profiles/r01_pmc.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)",
                     "bytes_per_launch": bpt // launches, "avg_launch_us": round(avg_launch_us, 3)},
    }

    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        # per-shape decode bandwidth (32 distinct weight sets per shape => HBM resident, not cache resident)
        for name, key, (K, N) in (("attn_4096x4096", "q", (HIDDEN, HIDDEN)), ("mlp_4096x11008", "gate", (HIDDEN, INTER)),
                                  ("mlp_11008x4096", "down", (INTER, HIDDEN))):
            x = torch.randn(1, K, device=dev, dtype=torch.float16)
            ls = [b[key] for b in stack.blocks]
            g, _ = capture(lambda: [l(x) for l in ls])
            ms = time_events(g.replay, 20) / len(ls)
            extra[f"decode_{name}"] = {"us": round(ms * 1e3, 2), "GBps": round(alg_bytes(K, N, 1) / ms / 1e6, 1)}
        # the other launch granularity (grouped: q/k/v and gate/up as single launches; ungrouped: one per linear)
        gg, _ = capture(lambda: stack.forward(h0, not grouped))
        ms = time_events(gg.replay, 20)
        extra["decode_stack_" + ("ungrouped" if grouped else "grouped")] = {
            "ms_per_token": round(ms, 4), "tokens_per_s": round(1e3 / ms, 1), "GBps": round(bpt / ms / 1e6, 1)}
        # prefill M=2048 (BASELINE configs[2]: GPTQ act-order) and AWQ, one layer's 7 linears x 4 layers
        for tag, cls, act in (("awq", WQLinear_GEMM, False), ("gptq_actorder", QuantLinearGPTQ, True)):
            ps = Stack(cls, 4, dev, seed=99, act_order=act)
            xp = torch.randn(2048, HIDDEN, device=dev, dtype=torch.float16)
            gp, _ = capture(lambda: ps.forward(xp))  # graph replay, like the headline leg: kernel time, not Python / allocator time
            ms = time_events(gp.replay, 10)
            del gp
            tf = flops_per_pass(4, 2048) / ms / 1e9
            extra[f"prefill_m2048_{tag}"] = {"ms_per_4_layers": round(ms, 3), "TFLOPs": round(tf, 1),
                                             "frac_of_mfma_peak": round(tf / MFMA_PEAK_TFLOPS, 4)}
            del ps
        # BASELINE configs[3]: HQQ g64 fp16 zeros, batch 16, alternating 4-bit / 3-bit layers.  Two decoder layers' 7 linears each.
        from qllm_amd.modeling.q_layers import QuantLinearHQQ
        hq = []
        for li, bits in enumerate((4, 3)):
            for (K, N) in [(HIDDEN, HIDDEN)] * 4 + [(HIDDEN, INTER)] * 2 + [(INTER, HIDDEN)]:
                l = QuantLinearHQQ(bits, 64, K, N, False, dtype=torch.float16)
                l.qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, l.qweight.shape, dtype=torch.int32, device=dev)
                l.qzeros = (torch.rand(l.qzeros.shape, device=dev) * (2 ** bits - 1)).half()
                l.scales = ((torch.rand(l.scales.shape, device=dev) * 0.4 + 0.8) / (K ** 0.5 * 6.5)).half()
                hq.append((l.to(dev), K))
        xs16 = {K: torch.randn(16, K, device=dev, dtype=torch.float16) for K in (HIDDEN, INTER)}
        for bits_sel, tag in ((4, "hqq_w4_g64_m16"), (3, "hqq_w3_g64_m16")):
            ls = [(l, K) for (l, K) in hq if l.bits == bits_sel]
            gh, _ = capture(lambda: [l(xs16[K]) for l, K in ls])
            ms = time_events(gh.replay, 20)
            del gh
            nbytes = sum(alg_bytes(l.infeatures, l.outfeatures, 16, 64, "f16") * bits_sel // 4 for l, _ in ls)
            extra[tag] = {"ms_per_layer": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 1)}
        result["extra"] = extra
        result["cpu_baseline"] = cpu_baseline_leg()
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
