"""-m gpu: the decode step as a loaded model drives it -- sibling groups (q/k/v, gate/up served by one grouped launch through the
unchanged module API) over the native-layout copies of the layers -- eagerly and under hipGraph replay, against the same
modules run one launch at a time on the reference buffers in place, and against the oracle."""
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


def test_decode_step_matches_in_place_launches_and_oracle(monkeypatch):
    """The bench's step (4 launches per block: grouped q/k/v, o, grouped gate/up, down; native strip-major copies) against the
    same layers run ungrouped on their reference buffers in place (QLLM_NATIVE_LAYOUT=0), and every launch against the oracle on
    its own input."""
    blocks = _blocks(2)
    h0 = torch.from_numpy(randx(1, H, seed=3)).to(DEV)
    kept = []
    y = _step(blocks, h0, kept)
    torch.cuda.synchronize()
    assert all(blk["q"][0]._siblings.grouped_launches == 1 for blk in blocks)
    assert "layout=strip-major" in blocks[0]["q"][0]._siblings.describe(1)
    monkeypatch.setenv("QLLM_NATIVE_LAYOUT", "0")
    monkeypatch.setenv("QLLM_FUSE_SIBLINGS", "0")
    plain_blocks = _blocks(2)            # same seeds -> same integers; these modules never build a native copy
    for blk in plain_blocks:
        for name in blk:
            blk[name][0]._siblings = None
    plain = []
    y_plain = _step(plain_blocks, h0, plain)
    torch.cuda.synchronize()
    for tp, tc in zip(plain, kept):
        for a, b in zip(tp, tc):
            assert torch.isfinite(b.float()).all()
            assert O.rel_err(b.cpu().numpy(), a.cpu().numpy()) <= 2e-3   # same math; the K split over waves differs
    assert O.rel_err(y.cpu().numpy(), y_plain.cpu().numpy()) <= 2e-3
    blk = blocks[1]
    q, k, v, o, gate, up, h = kept[1]
    h_in = kept[0][-1].cpu().numpy()
    for name, x_np, yy in (("q", h_in, q), ("v", h_in, v), ("o", q.cpu().numpy(), o), ("up", o.cpu().numpy(), up),
                           ("down", gate.cpu().numpy(), h)):
        ref = Ref(blk[name][1])
        assert O.rel_err(yy.cpu().numpy(), ref.y16(x_np)) <= 1e-2, name
        assert O.rel_err(yy.float().cpu().numpy(), ref.y64(x_np)) <= 2e-3, name


def test_decode_step_under_graph_replay_is_stable():
    """Capture one step, replay it many times while another stream keeps part of the chip busy (uneven load, warm caches): every
    replay must reproduce the eager result bit for bit."""
    blocks = _blocks(3, seed=100)
    h0 = torch.from_numpy(randx(1, H, seed=5)).to(DEV)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y_eager = _step(blocks, h0).clone()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = _step(blocks, h0)
    g.replay()
    torch.cuda.synchronize()
    first = y.clone()
    assert torch.equal(first, y_eager)
    noise = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for it in range(60):
        if it % 3 == 0:
            with torch.cuda.stream(noise):
                junk = junk @ junk * 1e-4   # a compute-bound kernel on another stream: uneven load on the CUs
        g.replay()
        if it % 10 == 9:
            torch.cuda.synchronize()
            assert torch.equal(y, first), it
    torch.cuda.synchronize()
    assert torch.equal(y, first)


def test_sibling_group_honours_autogptq_compat_per_forward(monkeypatch):
    """COMPATIBLE_WITH_AUTOGPTQ is read per forward by the reference (quant_linear_gptq.py:75); a sibling
    group must launch with the same zero-point offset the ungrouped module would use, and must not hand out outputs parked
    under the other setting."""
    from qllm_amd.modeling.q_layers import fuse_siblings
    ds = [synth("GPTQ", 4, 128, H, n, seed=40 + i) for i, n in enumerate((2048, 1024))]
    grouped = [to_layer(d, DEV) for d in ds]
    g = fuse_siblings(grouped)
    x = torch.from_numpy(randx(1, H, seed=11)).to(DEV)
    for compat in (1, 0, 1):
        monkeypatch.setenv("COMPATIBLE_WITH_AUTOGPTQ", str(compat))
        before = g.grouped_launches
        outs = [l(x) for l in grouped]
        assert g.grouped_launches == before + 1
        for o, d in zip(outs, ds):
            ref = Ref(dict(d, compat=compat))
            assert O.rel_err(o.cpu().numpy(), ref.y16(x.cpu().numpy())) <= 1e-2, compat
            other = Ref(dict(d, compat=1 - compat))
            assert O.rel_err(o.cpu().numpy(), other.y16(x.cpu().numpy())) > 5e-2    # the offset is not a rounding-level effect
    # a parked output computed under compat=1 is not served to a compat=0 call on the same tensor
    monkeypatch.setenv("COMPATIBLE_WITH_AUTOGPTQ", "1")
    grouped[0](x)
    monkeypatch.setenv("COMPATIBLE_WITH_AUTOGPTQ", "0")
    y1 = grouped[1](x)
    assert O.rel_err(y1.cpu().numpy(), Ref(dict(ds[1], compat=0)).y16(x.cpu().numpy())) <= 1e-2


def test_forward_under_inference_mode():
    """inference tensors have no version counter; the descriptor / sibling / gather caches must not read it."""
    from qllm_amd.modeling.q_layers import fuse_siblings
    ds = [synth("GPTQ", 4, 128, 1024, 512, seed=50 + i) for i in range(2)]
    da = synth("GPTQ", 4, 128, 1024, 512, act_order=True, seed=60)
    with torch.inference_mode():
        grouped = [to_layer(d, DEV) for d in ds]       # buffers created inside inference mode
        fuse_siblings(grouped)
        act = to_layer(da, DEV)
        x = torch.from_numpy(randx(1, 1024, seed=2)).to(DEV)
        outs = [l(x) for l in grouped]
        ya = act(x)
        torch.cuda.synchronize()
    for o, d in zip(outs, ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d).y16(x.cpu().numpy())) <= 1e-2
    assert O.rel_err(ya.cpu().numpy(), Ref(da).y16(x.cpu().numpy())) <= 1e-2


def test_sibling_group_switches_itself_off_when_nobody_collects():
    """callers that hand every sibling a new tensor object (accelerate hooks) would make the group launch all
    siblings per call; after three such launches it disables itself, and it never keeps the activation alive."""
    import gc
    import weakref
    from qllm_amd.modeling.q_layers import fuse_siblings
    ds = [synth("GPTQ", 4, 128, 1024, 512, seed=70 + i) for i in range(2)]
    grouped = [to_layer(d, DEV) for d in ds]
    g = fuse_siblings(grouped)
    x = torch.from_numpy(randx(1, 1024, seed=4)).to(DEV)
    grouped[0](x)
    r = weakref.ref(x)
    del x
    gc.collect()
    assert r() is None                                   # the group held x weakly
    for i in range(5):
        xi = torch.from_numpy(randx(1, 1024, seed=10 + i)).to(DEV)
        y = grouped[0](xi)                               # the sibling is never called: parked outputs pile up unclaimed
        assert O.rel_err(y.cpu().numpy(), Ref(ds[0]).y16(xi.cpu().numpy())) <= 1e-2
    assert not g.enabled


def test_a_refused_mid_batch_call_does_not_cost_the_group_its_decode_launches():
    """Round 4: a grouped call the library has no kernel for (here: 32-wide groups above 64 rows; before the grouped panel launch
    also every q/k/v call above 32 rows) must not switch the group off: a prefill chunk is followed by thousands of one-row decode steps."""
    from qllm_amd.modeling.q_layers import fuse_siblings
    ds = [synth("GPTQ", 4, 32, H, 1024, seed=30 + i) for i in range(3)]
    layers = [to_layer(d, DEV) for d in ds]
    g = fuse_siblings(layers)
    x1 = torch.from_numpy(randx(1, H, seed=1)).to(DEV)
    [l(x1) for l in layers]
    assert g.grouped_launches == 1
    x100 = torch.from_numpy(randx(100, H, seed=2)).to(DEV)
    outs = [l(x100) for l in layers]                       # refused by the grouped entry point: three launches of their own
    assert g.grouped_launches == 1 and g.enabled
    for o, d in zip(outs, ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d).y16(x100.cpu().numpy())) <= 1e-2
    [l(x100.clone()) for l in layers]                      # ... and not asked again at that size
    assert g.grouped_launches == 1
    x1b = torch.from_numpy(randx(1, H, seed=3)).to(DEV)
    outs = [l(x1b) for l in layers]
    assert g.grouped_launches == 2 and g.enabled           # decode is grouped as before
    for o, d in zip(outs, ds):
        assert O.rel_err(o.cpu().numpy(), Ref(d).y16(x1b.cpu().numpy())) <= 1e-2


def test_sibling_groups_take_one_panel_launch_from_17_rows():
    """q/k/v and gate/up at 17..128 rows: ONE grouped launch of the panel kernel (csrc/panel.hip) for the whole group."""
    from qllm_amd import ops
    from qllm_amd.modeling.q_layers import fuse_siblings
    for widths, layout, g_, bits in (((H, 1024, 1024), "GPTQ", 128, 4), ((I, I), "GEMM", 128, 4), ((H, H, H), "HQQ", 64, 4), ((H, H, H), "HQQ", 64, 3)):
        ds = [synth(layout, bits, g_, H, n, seed=50 + i, bias=(i == 1)) for i, n in enumerate(widths)]
        layers = [to_layer(d, DEV) for d in ds]
        grp = fuse_siblings(layers)
        for m in (17, 40, 64, 100, 128):
            if m > 64 and (sum(widths) > 16384 or bits == 3):   # (65..128 rows: 4 bits, groups of up to 16384 columns)
                assert grp.describe(m).startswith("unsupported")
                continue
            assert grp.describe(m).startswith("panel ") and f"layers={len(widths)}" in grp.describe(m), grp.describe(m)
            x = torch.from_numpy(randx(m, H, seed=m)).to(DEV)
            before = grp.grouped_launches
            outs = [l(x) for l in layers]
            assert grp.grouped_launches == before + 1
            for o, d in zip(outs, ds):
                ref = Ref(d)
                assert O.rel_err(o.cpu().numpy(), ref.y16(x.cpu().numpy())) <= 1e-2, (widths, m)
                assert O.rel_err(o.cpu().numpy().astype(np.float64), ref.y64(x.cpu().numpy())) <= 2e-3, (widths, m)
        xb = torch.from_numpy(randx(48, H, seed=9)).to(DEV).to(torch.bfloat16)
        outs = [l(xb) for l in layers]
        for o, d in zip(outs, ds):
            assert O.rel_err(o.float().cpu().numpy(), Ref(d).y64(xb.float().cpu().numpy().astype(np.float16))) <= 2e-2


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
