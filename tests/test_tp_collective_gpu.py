"""-m gpu: a REAL collective behind REAL HIP shards (verdict r02, Missing #1): two ranks on the ONE leased GPU (both cuda:0), gloo
over device tensors -- RCCL refuses two ranks on one device, and no multi-GPU box is in reach of the tests.  What the 8-GPU run
will execute -- column shards + all_gather / all_reduce, row shards + all_reduce, the Megatron pair with its single collective,
and tools/tp_bench.run itself -- against the unsharded HIP result.  SURVEY section 8e; the reference has no counterpart
(/root/reference/qllm/modeling/base.py:294-295 asserts the sharded branch away)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


def _ulps(a, b):
    """max distance in fp16 units of least precision (monotone integer view of the bit patterns)"""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7fff), i)
    return int((key(a) - key(b)).abs().max())


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, HERE)
        sys.path.insert(0, os.path.dirname(HERE))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        msgs = _body(rank, world)
        q.put((rank, msgs))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, [f"rank {rank}: {type(e).__name__}: {e}", traceback.format_exc()]))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _body(rank, world):
    import qllm_amd.parallel as P
    from gpu_util import randx, synth, to_layer
    dev = "cuda:0"
    msgs = []
    K, N = 4096, 4096
    full = {lay: to_layer(synth(lay, 4, 128, K, N, seed=11), dev) for lay in ("GEMM", "GPTQ")}
    calls = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor
    dist.all_reduce = lambda *a, **k: (calls.__setitem__("all_reduce", calls["all_reduce"] + 1), real_ar(*a, **k))[1]
    dist.all_gather_into_tensor = lambda *a, **k: (calls.__setitem__("all_gather", calls["all_gather"] + 1), real_ag(*a, **k))[1]
    for lay, layer in full.items():
        for m in (1, 5, 200):
            x = torch.from_numpy(randx(m, K, seed=m)).to(dev)
            y_full = layer(x)
            # ---- column-parallel: real HIP shard + ONE collective; every rank ends with the whole output
            for coll in ("all_gather", "all_reduce"):
                for static in (False, True):
                    cp = P.ColumnParallelQuantLinear.from_full(layer, collective=coll, static_output=static)
                    before = dict(calls)
                    y = cp(x)
                    used = {k: calls[k] - before[k] for k in calls}
                    if sum(used.values()) != 1 or used[coll] != 1:
                        msgs.append(f"{lay} column/{coll} m={m}: collectives {used}")
                    if y.shape != y_full.shape or _ulps(y, y_full) > 1:
                        msgs.append(f"{lay} column/{coll} m={m} static={static}: {_ulps(y, y_full)} ulp from the unsharded result")
                    if static:   # the output lives in the module's own buffer, re-used by every call ...
                        y2 = cp(x)
                        if y2.data_ptr() != y.data_ptr() or _ulps(y2, y_full) > 1:
                            msgs.append(f"{lay} column/{coll} m={m}: static output buffer not re-used")
                        # ... and the local half (the shard kernel writing its slice) allocates nothing (the gloo backend
                        # stages device tensors through buffers of its own, so the collective is left out of this count)
                        nl = cp.shard.outfeatures
                        tgt = torch.empty((m, nl), dtype=x.dtype, device=dev)
                        cp.shard.forward_into(x, tgt)
                        torch.cuda.synchronize()
                        a0 = torch.cuda.memory_allocated()
                        for _ in range(3):
                            cp.shard.forward_into(x, tgt)
                        torch.cuda.synchronize()
                        if torch.cuda.memory_allocated() != a0:
                            msgs.append(f"{lay} column/{coll} m={m}: the shard's in-place forward allocated memory")
            # ---- row-parallel: partial products over K + ONE all_reduce
            rp = P.RowParallelQuantLinear.from_full(layer, input_is_parallel=False, static_output=True)
            before = dict(calls)
            y = rp(x)
            if calls["all_reduce"] - before["all_reduce"] != 1 or calls["all_gather"] != before["all_gather"]:
                msgs.append(f"{lay} row m={m}: wrong collective count")
            if _rel(y, y_full) > 1e-3:
                msgs.append(f"{lay} row m={m}: rel err {_rel(y, y_full):.2e}")
    # ---- the Megatron pair: column (output stays sharded) -> row: ONE collective for two layers
    up = to_layer(synth("GEMM", 4, 128, 4096, 11008, seed=21), dev)
    dsyn = synth("GEMM", 4, 128, 11008, 4096, seed=22)
    dsyn["scales"] = (dsyn["scales"].astype(np.float32) * 0.2).astype(np.float16)
    down = to_layer(dsyn, dev)
    cp = P.ColumnParallelQuantLinear.from_full(up, gather_output=False)
    rp = P.RowParallelQuantLinear.from_full(down, input_is_parallel=True)
    for m in (1, 64):
        x = torch.from_numpy(randx(m, 4096, seed=40 + m)).to(dev)
        before = dict(calls)
        y = rp(cp(x))
        if calls["all_reduce"] - before["all_reduce"] != 1 or calls["all_gather"] != before["all_gather"]:
            msgs.append(f"megatron pair m={m}: expected exactly one all_reduce")
        y_full = down(up(x))
        if _rel(y, y_full) > 2e-3:
            msgs.append(f"megatron pair m={m}: rel err {_rel(y, y_full):.2e}")
    # ---- act-order GPTQ (4 and 3 bits) under tensor parallelism (a freshly built layer has act_order = None until
    #      something resolves it -- the shards' forward_into must): column shards keep the g_idx, row shards take whole groups of the
    #      group-sorted arrangement (shard.input_index), the pair's producer is sharded by the consumer's input channels
    for bits, g in ((4, 128), (3, 64)):
        ao = to_layer(synth("GPTQ", bits, g, 4096, 4096, act_order=True, seed=30 + bits), dev)
        ao2 = synth("GPTQ", bits, g, 4096, 4096, act_order=True, seed=40 + bits)
        ao2["scales"] = (ao2["scales"].astype(np.float32) * 0.2).astype(np.float16)
        ao2 = to_layer(ao2, dev)
        for m in (1, 5, 200):
            x = torch.from_numpy(randx(m, 4096, seed=60 + m)).to(dev)
            fresh = to_layer(synth("GPTQ", bits, g, 4096, 4096, act_order=True, seed=30 + bits), dev)   # act_order still None
            cp = P.ColumnParallelQuantLinear.from_full(fresh, static_output=True)
            y_full = ao(x)
            y = cp(x)
            if _ulps(y, y_full) > 1:
                msgs.append(f"act-order w{bits} column m={m}: {_ulps(y, y_full)} ulp from the unsharded result")
            rp = P.RowParallelQuantLinear.from_full(ao, input_is_parallel=False)
            if _rel(rp(x), y_full) > 1e-3:
                msgs.append(f"act-order w{bits} row m={m}: rel err {_rel(rp(x), y_full):.2e}")
            row = P.RowParallelQuantLinear.from_full(ao2, input_is_parallel=True)
            col = P.ColumnParallelQuantLinear(P.shard_columns(ao, rank, world, columns=row.shard.input_index), ao.outfeatures, gather_output=False)
            before = dict(calls)
            y = row(col(x))
            if calls["all_reduce"] - before["all_reduce"] != 1 or calls["all_gather"] != before["all_gather"]:
                msgs.append(f"act-order w{bits} pair m={m}: expected exactly one all_reduce")
            if _rel(y, ao2(ao(x))) > 2e-3:
                msgs.append(f"act-order w{bits} pair m={m}: rel err {_rel(y, ao2(ao(x))):.2e}")
    dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
    # ---- the local half of a decode-sized column-parallel forward captures into a hipGraph (the collective itself is a
    #      stream operation only under RCCL; gloo's is host-side)
    cp = P.ColumnParallelQuantLinear.from_full(full["GEMM"], static_output=True)
    x = torch.from_numpy(randx(1, K, seed=3)).to(dev)
    y_ref = cp(x).clone()
    buf = cp._bufs[next(iter(cp._bufs))]
    nl = cp.shard.outfeatures
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        cp.shard.forward_into(x, buf[:, rank * nl:(rank + 1) * nl])
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cp.shard.forward_into(x, buf[:, rank * nl:(rank + 1) * nl])
    buf.zero_()
    g.replay()
    torch.cuda.synchronize()
    if not torch.equal(buf[:, rank * nl:(rank + 1) * nl], y_ref[:, rank * nl:(rank + 1) * nl]):
        msgs.append("graph replay of the shard's in-place forward differs")
    # ---- the one-shot all-reduce (csrc/comm.hip, qllm_amd/comm.py): HIP IPC staging buffers mapped across the two PROCESSES (one
    #      device here, xGMI peers on a node), every call one kernel; against dist.all_reduce on copies: the same fp32 sum in rank
    #      order -> within one rounding of the collective's result; bit-identical across ranks; epochs / parities over many calls;
    #      in place behind a row-parallel layer; captured into a hipGraph and replayed
    from qllm_amd.comm import OneShotAllReduce
    ar = OneShotAllReduce(max_bytes=64 * 1024)
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    for dtype in (torch.float16, torch.bfloat16):
        for n in (8, 8192, 32768):
            for it in range(7):
                t = torch.randn(n, device=dev, dtype=torch.float32, generator=gen).to(dtype)
                ref = t.clone()
                dist.all_reduce(ref)
                out = ar.all_reduce(t.clone())
                torch.cuda.synchronize()
                if _rel(out, ref) > (1e-3 if dtype == torch.float16 else 8e-3):
                    msgs.append(f"one-shot all-reduce {dtype} n={n} call {it}: rel err {_rel(out, ref):.2e}")
                both = [torch.empty_like(out) for _ in range(world)]
                dist.all_gather(both, out)
                if not torch.equal(both[0], both[1]):
                    msgs.append(f"one-shot all-reduce {dtype} n={n}: ranks disagree")
    ar.check()
    big = torch.randn(1 << 20, device=dev, dtype=torch.float16, generator=gen)   # does not fit a slot: falls through to the group's collective
    ref = big.clone()
    dist.all_reduce(ref)
    if not torch.equal(ar.all_reduce(big.clone()), ref):
        msgs.append("one-shot wrapper: large tensor did not take dist.all_reduce")
    rp1 = P.RowParallelQuantLinear.from_full(full["GEMM"], input_is_parallel=False, static_output=True, reducer=ar)
    x1 = torch.from_numpy(randx(1, K, seed=77)).to(dev)
    if _rel(rp1(x1), full["GEMM"](x1)) > 1e-3:
        msgs.append("row-parallel layer behind the one-shot reducer differs from the unsharded layer")
    # ---- fused (round 5, csrc/strip1_kernel.hpp AR): at batch 1 the row-parallel shard's launch pushes its partial outputs to the
    #      peers itself and its last block sums: no separate all-reduce launch, BIT-IDENTICAL to GEMV + one-shot all-reduce, equal on
    #      every rank, over many calls (epochs, parities, the ticket re-arming), fp16 and bf16 activations, with a bias, the
    #      shifted-window form (K / 2 = 5504), mixed with plain one-shot calls on the same buffers, and as a hipGraph
    dbias = synth("GPTQ", 4, 128, 11008, 4096, bias=True, seed=23)
    dbias["scales"] = (dbias["scales"].astype(np.float32) * 0.2).astype(np.float16)
    cases = [("GEMM 4096x4096", full["GEMM"], K), ("GPTQ 4096x4096", full["GPTQ"], K), ("GPTQ 11008x4096 bias", to_layer(dbias, dev), 11008)]
    real_one = ar.all_reduce
    n_one = [0]
    ar.all_reduce = lambda t_: (n_one.__setitem__(0, n_one[0] + 1), real_one(t_))[1]
    for name, layer, kk in cases:
        rf = P.RowParallelQuantLinear.from_full(layer, input_is_parallel=False, static_output=True, reducer=ar, fuse_reduce=True)
        ru = P.RowParallelQuantLinear.from_full(layer, input_is_parallel=False, static_output=True, reducer=ar, fuse_reduce=False)
        for dtype in (torch.float16, torch.bfloat16):
            for it in range(5):
                xx = torch.from_numpy(randx(1, kk, seed=500 + it)).to(dev).to(dtype)
                n0 = n_one[0]
                yf = rf(xx).clone()
                if n_one[0] != n0:
                    msgs.append(f"fused row-parallel {name} {dtype}: the separate all-reduce ran (fused launch not taken)")
                yu = ru(xx).clone()
                torch.cuda.synchronize()
                if not torch.equal(yf, yu):
                    msgs.append(f"fused row-parallel {name} {dtype} call {it}: differs from GEMV + one-shot all-reduce ({_rel(yf, yu):.2e})")
                both = [torch.empty_like(yf) for _ in range(world)]
                dist.all_gather(both, yf)
                if not torch.equal(both[0], both[1]):
                    msgs.append(f"fused row-parallel {name} {dtype}: ranks disagree")
        if _rel(rf(torch.from_numpy(randx(1, kk, seed=9)).to(dev)), layer(torch.from_numpy(randx(1, kk, seed=9)).to(dev))) > 2e-3:
            msgs.append(f"fused row-parallel {name}: differs from the unsharded layer")
    ar.all_reduce = real_one
    ar.check()
    rf = P.RowParallelQuantLinear.from_full(full["GEMM"], input_is_parallel=False, static_output=True, reducer=ar)
    xg = torch.from_numpy(randx(1, K, seed=91)).to(dev)
    exp_f = rf(xg).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        rf(xg)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gf = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf):
        yg = rf(xg)
    dist.barrier()
    for _ in range(3):
        yg.zero_()
        gf.replay()
        torch.cuda.synchronize()
        if not torch.equal(yg, exp_f):
            msgs.append("graph replay of the fused row-parallel launch differs")
        dist.barrier()
    ar.check()
    t = torch.randn(8192, device=dev, dtype=torch.float16, generator=gen)
    tin = t.clone()
    exp = tin.clone()
    dist.all_reduce(exp)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ar.all_reduce(t)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        ar.all_reduce(t)
    dist.barrier()
    for _ in range(3):   # replays advance the epoch on the device, like the captured call did
        t.copy_(tin)
        g.replay()
        torch.cuda.synchronize()
        if _rel(t, exp) > 1e-3:
            msgs.append("graph replay of the one-shot all-reduce differs")
        dist.barrier()
    ar.check()
    ar.close()
    # ---- tools/tp_bench.run, as `bench.py --tp 2` would drive it (two layers, eager because the backend is gloo)
    from qllm_amd import _lib
    from tools import tp_bench
    args = types.SimpleNamespace(tp=world, steps=10, warmup=2, tp_layers=8, keep_process_group=True)
    tp_bench.run(args, world, rank, torch.device(dev), _lib.device_info(0))
    return msgs


def test_hip_shards_behind_a_real_collective_two_ranks_one_gpu(capfd):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    for rank, msgs in sorted(results):
        assert not msgs, (rank, msgs)
    out = capfd.readouterr().out
    try:   # (evidence: the fused / unfused timings of this run, copied to profiles/ by the round's evidence script)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "tp_bench_two_ranks_one_gpu.log"), "w") as f:
            f.write("\n".join(l for l in out.splitlines() if "tp_bench" in l or l.startswith("{")) + "\n")
    except OSError:
        pass
    assert "[tp_bench] world_size=2 backend=gloo tp_degree=2 layers=8" in out
    assert "[tp_bench] sharded == unsharded on 2 rank(s)" in out
    assert "[tp_bench] row-parallel sums: one-shot peer-write kernel" in out and '"oneshot_all_reduce_us_16KB"' in out
    assert "[tp_bench] step runs as: hipGraph replay" in out   # (every sum of the step is one of the library's kernels: it captures under gloo too)
    # both forms are timed and the faster one is reported (on a shared GPU that may be either)
    assert "[tp_bench] row-parallel GEMV + all-reduce: fused (one launch)" in out and "-> reporting the" in out
    assert '"ms_per_step_fused_all_reduce"' in out and '"ms_per_step_unfused_all_reduce"' in out
    assert '"ranks": 2' in out and '"all_reduces_per_layer": 2' in out


def test_bench_gpus_2_carries_the_tensor_parallel_leg():
    """`bench.py --gpus 2` -- the ONLY command the driver's scaling run issues -- end to end on the leased GPU: bench.py starts its two
    ranks itself, both on cuda:0 (QLLM_BENCH_SHARE_GPU0=1: gloo, RCCL refuses two ranks on one device), measures the replicas and then
    the Llama-2-70B TP = 2 leg, and prints ONE line whose extra.tp70b is a measurement (round-5 verdict item 2)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QLLM_BENCH_SHARE_GPU0"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--tp-layers", "4",
                        "--min-timed-s", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["ranks_seen"] == 2 and rec["config"]["backend"] == "gloo"
    tp = rec["extra"]["tp70b"]
    assert "error" not in tp, tp
    assert tp["tokens_per_s"] > 0 and tp["ranks_seen"] == 2 and tp["tp_degree"] == 2 and tp["layers"] == 4
    assert tp["sharded_vs_unsharded"]["column_parallel"] == "bit-exact"
    assert all(v <= 2e-3 for v in tp["sharded_vs_unsharded"]["megatron_pairs_rel_err"].values())
    assert tp["all_reduce_us_16KB"] > 0
