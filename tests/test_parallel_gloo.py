"""N > 1 path on CPU: world_size-2 gloo processes.  The shard's local forward needs the GPU, so here it is replaced
by the CPU oracle (tests may use the oracle as the checker); what is under test is the sharding arithmetic of
qllm_amd.parallel and the collectives: sharded result == unsharded result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_forward(layer, x):
    from oracle import ref_cpu as O
    lay = layer.pack_mode
    gi = layer.g_idx.numpy() if lay == "GPTQ" else None
    y = O.forward(lay, x.numpy(), layer.qweight.numpy(), layer.scales.numpy(), layer.qzeros.numpy(), gi,
                  layer.bias.numpy() if layer.bias is not None else None, layer.bits, layer.groupsize, layer.infeatures)
    return y


def _build(name):
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM
    g = load_golden(name)
    cls = {"GPTQ": QuantLinearGPTQ, "GEMM": WQLinear_GEMM, "HQQ": QuantLinearHQQ}[g["layout"]]
    layer = cls(g["bits"], g["groupsize"], g["K"], g["N"], g["bias"] is not None, dtype=torch.float16)
    layer.qweight = torch.from_numpy(g["qweight"])
    layer.qzeros = torch.from_numpy(g["qzeros"])
    layer.scales = torch.from_numpy(g["scales"])
    layer.g_idx = torch.from_numpy(g["g_idx"])
    if g["bias"] is not None:
        layer.bias = torch.from_numpy(g["bias"])
    return g, layer


def _worker(rank, world, port, names, q):
    try:
        _worker_body(rank, world, port, names, q)
    except Exception as e:  # noqa: BLE001 -- surface the failure instead of letting the parent time out
        import traceback
        q.put((False, [f"rank {rank}: {e}", traceback.format_exc()]))
        raise


def _worker_body(rank, world, port, names, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import qllm_amd.parallel as P
    from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM
    for cls in (QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM):
        cls.forward = _oracle_forward  # CPU stand-in for the HIP forward (test only)
        cls.forward_into = lambda self, x, out: out.copy_(_oracle_forward(self, x).reshape(out.shape))
    ok = True
    msgs = []
    for name in names:
        g, layer = _build(name)
        x = torch.from_numpy(g["x"][:5])
        y_full = _oracle_forward(layer, x)
        for coll in ("all_gather", "all_reduce"):
            cp = P.ColumnParallelQuantLinear.from_full(layer, collective=coll)
            assert cp.shard.outfeatures == g["N"] // world
            y = cp(x)
            if not torch.equal(y, y_full):  # columns are independent: bit-exact
                ok = False
                msgs.append(f"{name} column/{coll} mismatch")
        # row-parallel, replicated input: plain layers take their slice of x, act-order layers (uniform groups) gather the
        # input channels of their groups (shard.input_index) and run as contiguous-group layers
        rp = P.RowParallelQuantLinear.from_full(layer, input_is_parallel=False)
        if "actorder" in name:
            assert rp.shard.input_index.numel() == g["K"] // world and not rp.shard.act_order
            assert torch.equal(rp.shard.g_idx, (torch.arange(g["K"] // world) // g["groupsize"]).to(torch.int32))
        y = rp(x)
        err = float((y.float() - y_full.float()).abs().max() / y_full.float().abs().max())
        if err > 2e-3:  # different summation order + fp16 partials in this CPU stand-in
            ok = False
            msgs.append(f"{name} row-parallel err {err}")
    # Megatron pair (o_proj after q/k/v, down after gate/up): column-parallel WITHOUT the gather feeding row-parallel with
    # input_is_parallel: rank r's output columns are exactly rank r's input rows of the second layer, so the pair costs ONE
    # all-reduce and no all-gather.  Count the collectives, compare with the unsharded pair.
    from oracle import ref_cpu as O
    rng = np.random.default_rng(5)

    def mk(K, N, bias):
        qw, qz = O.pack_gptq(rng.integers(0, 16, (K, N), dtype=np.int32), rng.integers(0, 16, (K // 128, N), dtype=np.int32), 4)
        l = QuantLinearGPTQ(4, 128, K, N, bias, dtype=torch.float16)
        l.qweight, l.qzeros = torch.from_numpy(qw), torch.from_numpy(qz)
        l.scales = torch.from_numpy((rng.random((K // 128, N)) * 0.01 + 0.002).astype(np.float16))
        if bias:
            l.bias = torch.from_numpy((rng.standard_normal(N) * 0.1).astype(np.float16))
        return l

    A, B = mk(256, 512, False), mk(512, 256, True)
    x = torch.from_numpy(rng.standard_normal((3, 256)).astype(np.float16))
    y_full = _oracle_forward(B, _oracle_forward(A, x))
    calls = {"all_reduce": 0, "all_gather": 0}
    real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor

    def count_ar(*a, **k):
        calls["all_reduce"] += 1
        return real_ar(*a, **k)

    def count_ag(*a, **k):
        calls["all_gather"] += 1
        return real_ag(*a, **k)

    dist.all_reduce, dist.all_gather_into_tensor = count_ar, count_ag
    try:
        col = P.ColumnParallelQuantLinear.from_full(A, gather_output=False)
        row = P.RowParallelQuantLinear.from_full(B, input_is_parallel=True)
        h = col(x)
        assert h.shape == (3, 512 // world)
        y = row(h)
    finally:
        dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
    if calls != {"all_reduce": 1, "all_gather": 0}:
        ok = False
        msgs.append(f"Megatron pair used {calls}")
    # (the bias of a row-parallel layer must be added once, not once per rank)
    err = float((y.float() - y_full.float()).abs().max() / y_full.float().abs().max())
    if err > 2e-3:
        ok = False
        msgs.append(f"Megatron pair err {err}")
    # the same pair with an ACT-ORDER consumer: the producer is sharded by the consumer's input channels
    # (shard_columns(columns=consumer_shard.input_index)), so the local activation is already the consumer's local input
    gB = (torch.arange(512) // 128)[torch.from_numpy(rng.permutation(512))].to(torch.int32)
    Bo = mk(512, 256, True)
    Bo.g_idx = gB
    Bo.act_order = True
    y_full = _oracle_forward(Bo, _oracle_forward(A, x))
    calls = {"all_reduce": 0, "all_gather": 0}
    dist.all_reduce, dist.all_gather_into_tensor = count_ar, count_ag
    try:
        row = P.RowParallelQuantLinear.from_full(Bo, input_is_parallel=True)
        col = P.ColumnParallelQuantLinear(P.shard_columns(A, rank, world, columns=row.shard.input_index), A.outfeatures, gather_output=False)
        y = row(col(x))
    finally:
        dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
    if calls != {"all_reduce": 1, "all_gather": 0}:
        ok = False
        msgs.append(f"act-order Megatron pair used {calls}")
    err = float((y.float() - y_full.float()).abs().max() / y_full.float().abs().max())
    if err > 2e-3:
        ok = False
        msgs.append(f"act-order Megatron pair err {err}")
    if rank == 0:
        q.put((ok, msgs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_equals_unsharded_world2():
    names = ["gptq_w4_g128_asym", "gptq_w4_g128_actorder", "awq_w4_g64_bias", "hqq_w4_g64", "gptq_w4_g128_opt_bias",
             "gptq_w3_g128_asym", "gptq_w3_g64_actorder", "gptq_w4_g32_actorder_bias"]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, names, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, msgs = q.get(timeout=240)
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.kill()
    assert ok, msgs
    assert all(p.exitcode == 0 for p in procs)


def test_shard_shapes_and_errors():
    import qllm_amd.parallel as P
    _, layer = _build("awq_w4_g64_bias")
    s1 = P.shard_columns(layer, 1, 2)
    assert s1.qweight.shape == (256, 16) and s1.scales.shape == (4, 128) and s1.bias.shape == (128,)
    assert torch.equal(s1.qweight, layer.qweight[:, 16:])
    with pytest.raises(ValueError):
        P.shard_columns(layer, 0, 3)
    _, gl = _build("gptq_w4_g128_actorder")
    r1 = P.shard_rows(gl, 1, 2)   # act-order, uniform groups: the rows of the upper half of the groups, group-sorted
    gi = gl.g_idx.long()
    assert torch.equal(r1.input_index.long(), torch.argsort(gi, stable=True)[gl.infeatures // 2:])
    assert bool((gi[r1.input_index.long()] >= gi.max().item() // 2 + 1).all()) and r1.act_order is False
    gl.g_idx = gl.g_idx.clone()
    gl.g_idx[0] = gl.g_idx[1] = gl.g_idx[2]   # groups no longer uniform
    with pytest.raises(ValueError):
        P.shard_rows(gl, 0, 2)
    _, g0 = _build("gptq_w4_g128_asym")
    cs = P.shard_columns(g0, 0, 2, columns=torch.arange(g0.outfeatures - 1, -1, -2))   # every other column, reversed
    cols = torch.arange(g0.outfeatures - 1, -1, -2)
    assert cs.outfeatures == g0.outfeatures // 2 and torch.equal(cs.scales, g0.scales[:, cols])
    x = torch.from_numpy(load_golden("gptq_w4_g128_asym")["x"][:3])
    assert torch.equal(_oracle_forward(cs, x), _oracle_forward(g0, x)[:, cols])   # columns are independent: bit-exact
    _, hl = _build("hqq_w4_g64")
    r0 = P.shard_rows(hl, 0, 2)
    assert r0.qweight.shape == (16, 128) and r0.qzeros.shape == (2, 128) and r0.infeatures == 128
