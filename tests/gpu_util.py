"""Helpers for the -m gpu parity tests: synthetic layers in packed form + oracle evaluation (checker only)."""
import numpy as np
import torch

from oracle import ref_cpu as O
from qllm_amd.modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM

LAYER = {"GPTQ": QuantLinearGPTQ, "GEMM": WQLinear_GEMM, "HQQ": QuantLinearHQQ}


def synth(layout, bits, g, K, N, zero_kind="asym", act_order=False, bias=False, seed=0):
    """Random quantized layer directly in packed form (any int32 is a valid qweight word)."""
    rng = np.random.default_rng(seed)
    G = (K + g - 1) // g
    q = rng.integers(0, 2 ** bits, size=(K, N), dtype=np.int32)
    scales = (rng.random((G, N)) * 0.010 + 0.002).astype(np.float16)
    if layout == "HQQ":
        zeros = (rng.random((G, N)) * (2 ** bits - 1)).astype(np.float16)
        qweight, qzeros = O.pack_along_rows(q, bits), zeros
    else:
        zeros = (np.full((G, N), 2 ** (bits - 1), np.int32) if zero_kind == "sym"
                 else rng.integers(0, 2 ** bits, size=(G, N), dtype=np.int32))
        qweight, qzeros = (O.pack_awq(q, zeros) if layout == "GEMM" else O.pack_gptq(q, zeros, bits))
    g_idx = O.trivial_g_idx(K, g)
    if act_order:
        g_idx = g_idx[rng.permutation(K)].astype(np.int32)
        if g_idx[:g].sum() == 0:
            g_idx[0] = G - 1
    b = (rng.standard_normal(N) * 0.5).astype(np.float16) if bias else None
    return dict(layout=layout, bits=bits, groupsize=g, K=K, N=N, qweight=qweight, qzeros=qzeros, scales=scales,
                g_idx=g_idx, bias=b, compat=0)


def to_layer(d, device="cuda:0", dtype=torch.float16):
    layer = LAYER[d["layout"]](d["bits"], d["groupsize"], d["K"], d["N"], d["bias"] is not None, dtype=dtype)
    layer.qweight = torch.from_numpy(np.ascontiguousarray(d["qweight"]))
    layer.qzeros = torch.from_numpy(np.ascontiguousarray(d["qzeros"])).to(dtype if d["layout"] == "HQQ" else torch.int32)
    layer.scales = torch.from_numpy(d["scales"]).to(dtype)
    layer.g_idx = torch.from_numpy(np.ascontiguousarray(d["g_idx"]))
    if d["bias"] is not None:
        layer.bias = torch.from_numpy(d["bias"]).to(dtype)
    return layer.to(device)


def oracle_w(d):
    gi = d["g_idx"] if (d["layout"] == "GPTQ" and O.is_act_order(d["g_idx"], d["groupsize"])) else None
    return O.dequant(d["layout"], d["qweight"], d["scales"], d["qzeros"], gi, d["bits"], d["groupsize"], d["K"],
                     d.get("compat", 0))


def oracle_y(d, x, w=None):
    w = oracle_w(d) if w is None else w
    return O.matmul_f16(x, w, d["bias"]).numpy()


def randx(m, k, seed=1):
    return np.random.default_rng(seed).standard_normal((m, k)).astype(np.float16)


class Ref:
    """Oracle results for one synthetic layer, with the dequantised W converted once (the big-K cases are
    otherwise dominated by re-converting W to float64 for every M)."""

    def __init__(self, d):
        self.d = d
        self.w = oracle_w(d)
        self.w16 = torch.from_numpy(self.w)
        self.w64 = self.w16.double()
        self.w32 = self.w16.float()
        self.b16 = torch.from_numpy(d["bias"]) if d["bias"] is not None else None

    def y16(self, x):
        """what the reference's CPU path returns: fp16 matmul (+ bias)"""
        xt = torch.from_numpy(x)
        if x.shape[0] <= 8:
            y = torch.matmul(xt, self.w16)  # the reference's own op; fast enough only for a few rows
        else:
            y = torch.matmul(xt.float(), self.w32).half()  # oracle.matmul_f16_via_f32
        if self.b16 is not None:
            y = y + self.b16
        return y.numpy()

    def y64(self, x):
        y = torch.from_numpy(x).double() @ self.w64
        if self.b16 is not None:
            y = y + self.b16.double()
        return y.numpy()
