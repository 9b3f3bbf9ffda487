"""-m gpu: the reference-side stubs of INTEGRATION.md Level 1 (integration/ort_ops.py, integration/awq_inference_engine.py) --
the files a QLLM maintainer would drop in as qllm/ort_ops.py and qllm/awq_inference_engine.py -- executed as they are: raw
ctypes over qllm_ort_gemv / qllm_ort_dequant / qllm_ort_dequantize4bits / qllm_awq_gemm_forward, nothing imported from qllm_amd."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, ort_golden_names
from oracle import ref_cpu as O
from gpu_util import Ref, oracle_w, randx, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load_stub(name):
    os.environ["QLLM_MI355X_LIB"] = os.path.join(ROOT, "qllm_amd", "libqllm_mi355x.so")
    spec = importlib.util.spec_from_file_location("qllm_stub_" + name, os.path.join(ROOT, "integration", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert "qllm_amd" not in getattr(mod, "__dict__", {})  # self-contained: only ctypes + torch
    return mod


def _t(d, *keys):
    return [torch.from_numpy(np.ascontiguousarray(d[k])).to(DEV) for k in keys]


def test_ort_ops_stub_gemv_and_dequant():
    ort_ops = _load_stub("ort_ops")
    # decode (M <= 8: the reference's own call site, quant_linear_gptq.py:76-80), prefill-sized M, act-order, AutoGPTQ offset
    for seed, (K, N, act, m) in enumerate(((4096, 4096, False, 1), (4096, 11008, False, 8), (1024, 512, True, 3), (4096, 4096, False, 300))):
        d = synth("GPTQ", 4, 128, K, N, "asym", act, False, seed=80 + seed)
        qweight, scales, qzeros, g_idx = _t(d, "qweight", "scales", "qzeros", "g_idx")
        ref = Ref(d)
        w = ort_ops.dequant(qweight, scales, qzeros, g_idx if act else None, 128, 4, K, 0)
        assert np.array_equal(w.cpu().numpy().view(np.uint16), ref.w.view(np.uint16))            # bit-exact W[K, N]
        x = randx(m, K, seed=seed)
        y = ort_ops.gemv(torch.from_numpy(x).to(DEV), qweight, scales, qzeros, g_idx if act else None, 128, 4, K, 0)
        assert y.shape == (m, N)
        assert O.rel_err(y.cpu().numpy(), ref.y16(x)) <= 1e-2
    # leading batch dims and the error convention
    x3 = torch.from_numpy(randx(6, K, seed=9)).to(DEV).reshape(2, 3, K)
    assert ort_ops.gemv(x3, qweight, scales, qzeros, None, 128, 4, K, 0).shape == (2, 3, N)
    with pytest.raises(RuntimeError):
        ort_ops.gemv(x3.cpu(), qweight, scales, qzeros, None, 128, 4, K, 0)
    with pytest.raises(RuntimeError):
        ort_ops.dequant(qweight, scales, qzeros, None, 128, 9, K, 0)                               # bits out of range -> message
    # COMPATIBLE_WITH_AUTOGPTQ: stored zero + 1
    g = load_golden("gptq_w4_g128_autogptq")
    qweight, scales, qzeros = _t(g, "qweight", "scales", "qzeros")
    w = ort_ops.dequant(qweight, scales, qzeros, None, g["groupsize"], 4, g["K"], 1)
    assert np.array_equal(w.cpu().numpy().view(np.uint16), g["W_fwd"].view(np.uint16))


@pytest.mark.parametrize("name", ort_golden_names())
def test_ort_ops_stub_dequantize4bits(name):
    ort_ops = _load_stub("ort_ops")
    g = load_golden(name)
    K, N, block = g["K"], g["N"], g["groupsize"]
    qweight, scales, qzeros, g_idx = _t(g, "qweight", "scales_flat", "qzeros", "g_idx")
    act = O.ort_is_act_order(g["g_idx"])
    w = ort_ops.Dequantize4Bits(qweight, scales, qzeros, g_idx if act else None, block, K, N)
    assert w.shape == (N, K)
    assert np.array_equal(w.cpu().numpy().view(np.uint16), g["W_unpack"].view(np.uint16))      # the reference's own W[N, K]


def test_awq_inference_engine_stub():
    eng = _load_stub("awq_inference_engine")
    for seed, (K, N, m) in enumerate(((4096, 4096, 1), (4096, 11008, 16), (11008, 4096, 2048), (1024, 512, 130))):
        d = synth("GEMM", 4, 128, K, N, seed=90 + seed)
        qweight, scales, qzeros = _t(d, "qweight", "scales", "qzeros")
        x = randx(m, K, seed=seed)
        y = eng.gemm_forward_cuda(torch.from_numpy(x).to(DEV), qweight, scales, qzeros, 8)
        assert y.shape == (m, N) and y.dtype == torch.float16
        assert O.rel_err(y.cpu().numpy(), Ref(d).y16(x)) <= 1e-2
    with pytest.raises(RuntimeError):
        eng.gemm_forward_cuda(torch.from_numpy(x).to(DEV)[:, :100].contiguous(), qweight, scales, qzeros, 8)   # K not a multiple of g


def test_awq_inference_engine_stub_decode_uses_a_native_copy():
    """M <= 128 through the stub: served from a cached native-layout copy of the caller's integers (strip / panel kernels); the copy follows
    in-place updates of the caller's tensors (keyed on identity AND version), dies with them, and can be switched off."""
    eng = _load_stub("awq_inference_engine")
    d = synth("GEMM", 4, 128, 4096, 4096, seed=7)
    qweight, scales, qzeros = _t(d, "qweight", "scales", "qzeros")
    for m in (1, 5, 64, 100):
        x = randx(m, 4096, seed=m)
        y = eng.gemm_forward_cuda(torch.from_numpy(x).to(DEV), qweight, scales, qzeros, 8)
        assert O.rel_err(y.cpu().numpy(), Ref(d).y16(x)) <= 1e-2 and O.rel_err(y.float().cpu().numpy(), Ref(d).y64(x)) <= 2e-3
    assert len(eng._native) == 1                                         # one weight, one copy, reused across M
    x = randx(1, 4096, seed=1)
    xt = torch.from_numpy(x).to(DEV)
    y_shadow = eng.gemm_forward_cuda(xt, qweight, scales, qzeros, 8)
    os.environ["QLLM_AWQ_DECODE_SHADOW"] = "0"
    try:
        y_inplace = eng.gemm_forward_cuda(xt, qweight, scales, qzeros, 8)
    finally:
        del os.environ["QLLM_AWQ_DECODE_SHADOW"]
    assert O.rel_err(y_shadow.cpu().numpy(), y_inplace.cpu().numpy()) <= 2e-3
    # the caller overwrites its weights in place (e.g. loads another checkpoint into the same buffers): no stale copy
    d2 = synth("GEMM", 4, 128, 4096, 4096, seed=8)
    qweight.copy_(torch.from_numpy(d2["qweight"]))
    qzeros.copy_(torch.from_numpy(d2["qzeros"]))
    scales.copy_(torch.from_numpy(d2["scales"]))
    y2 = eng.gemm_forward_cuda(xt, qweight, scales, qzeros, 8)
    assert O.rel_err(y2.cpu().numpy(), Ref(d2).y16(x)) <= 1e-2
    # the cache entry dies with the caller's tensors -- a later model whose weights land on the same addresses (the
    # caching allocator hands them out again) can never be served the old copy
    import gc
    ptr = qweight.data_ptr()
    del qweight, scales, qzeros
    gc.collect()
    assert len(eng._native) == 0
    d3 = synth("GEMM", 4, 128, 4096, 4096, seed=9)
    qweight, scales, qzeros = _t(d3, "qweight", "scales", "qzeros")
    y3 = eng.gemm_forward_cuda(xt, qweight, scales, qzeros, 8)
    assert O.rel_err(y3.cpu().numpy(), Ref(d3).y16(x)) <= 1e-2, ("same address reused" if qweight.data_ptr() == ptr else "new address")
