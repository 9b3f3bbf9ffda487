"""CPU oracle for the QLLM W4A16 dequant+matmul hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  Nothing under
``qllm_amd/`` imports it; the product path has no CPU compute fallback.

What it is: an independent restatement (numpy for the integer/bit work, numpy float16 for the
per-element dequant arithmetic, torch CPU for the fp16 matmul exactly like the reference) of the
reference's CPU path for the hot path named in BASELINE.json:

  * ``DequantizeLinearBlockWise``      /root/reference/qllm/modeling/q_layers/quant_linear_gptq.py:13-52
  * ``QuantLinearTorchFunction.forward`` branch (C) + ``torch.matmul``          ...quant_linear_gptq.py:71-85
  * ``QuantLinearGPTQ.forward`` (lazy act-order detect, bias)                    ...quant_linear_gptq.py:136-143
  * ``DequantAndUnpack.forward`` / ``QuantLinearHQQ.forward``                    ...quant_linear_hqq.py:8-38,76-80
  * ``WQLinear_GEMM`` CPU truth = ``unpack()`` + ``F.linear``                    ...quant_linear_awq.py:76-140,
                                                                                 ...compress_weight.py:105-151
  * bit layouts (``general_pack_on_row`` / ``general_unpack_on_row``)            ...compress_weight.py:10-92
  * ``pack_qzeros`` (COMPATIBLE_WITH_AUTOGPTQ offset)                            ...compress_weight.py:156-172
  * ``handle_qzeros_for_autogptq``                                               ...quant_linear_gptq.py:119-134
  * ORT / MatMulNBits blob layout: ``dequantize_blockwise_4bits`` / ``QuantLinearORT.pack_on_device``
    / ``forward``                                       ...quant_linear_onnxruntime.py:52-82, 116-153, 169-174
    (goldens: ``tests/golden/make_goldens_ort.py`` -> ``ort_*.npz``)

Parity pinning: the reference ships no tests, so this oracle is pinned against
golden vectors minted by importing the reference itself in the build container
(``tests/golden/make_goldens.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``): packed buffers bit-exact, dequantised W bit-exact, y within 1e-3.

Numerics contract restated from the reference: every elementwise op is carried out in fp16 with one
round-to-nearest-even per op:  ``W = fp16(fp16(s*q) - fp16(z*s))``.  numpy's float16 arithmetic computes in
float32 and rounds once, which equals the correctly rounded fp16 result for * and - (24 >= 2*11+2), i.e. the same
values torch's CPU half kernels produce.
"""
from __future__ import annotations

import numpy as np

AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)  # quant_linear_awq.py:107  nibble i of a word holds column 8j+AWQ_ORDER[i]


# ----------------------------------------------------------------------------------------------
# bit layouts
# ----------------------------------------------------------------------------------------------
def _as_u32(a) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(a))
    if a.dtype == np.int32:
        return a.view(np.uint32)
    if a.dtype == np.uint32:
        return a
    return a.astype(np.int64).astype(np.uint32)


def unpack_along_rows(packed, bits: int, n_rows: int) -> np.ndarray:
    """packed i32 [n_rows*bits/32, C] -> int32 [n_rows, C].

    Column c is a little-endian bit stream along the row axis: value k occupies stream bits
    [k*bits, (k+1)*bits); stream word w is packed[w, c].  For bits in {2,4,8} this is the reference's
    shift+mask fast path (compress_weight.py:54-66: value j of word r -> row r*(32/bits)+j); for
    bits in {3,5,6,7} it is its bit-stream path (compress_weight.py:69-84).
    """
    p = _as_u32(packed)
    n_words, n_cols = p.shape
    assert n_words * 32 >= n_rows * bits, (p.shape, bits, n_rows)
    mask = (1 << bits) - 1
    if 32 % bits == 0:
        ratio = 32 // bits
        shifts = (np.arange(ratio, dtype=np.uint32) * bits).reshape(1, ratio, 1)
        out = (p[:, None, :] >> shifts) & np.uint32(mask)
        return out.reshape(n_words * ratio, n_cols)[:n_rows].astype(np.int32)
    # general bit stream: a value may straddle two words
    k = np.arange(n_rows, dtype=np.int64)
    bit0 = k * bits
    w0 = bit0 // 32
    off = (bit0 % 32).astype(np.uint64).reshape(-1, 1)
    lo = p[w0].astype(np.uint64)
    w1 = np.minimum(w0 + 1, n_words - 1)
    hi = p[w1].astype(np.uint64)
    both = lo | (hi << np.uint64(32))
    return ((both >> off) & np.uint64(mask)).astype(np.int32)


def pack_along_rows(values, bits: int) -> np.ndarray:
    """int [n_rows, C] -> i32 [n_rows*bits/32, C] (inverse of unpack_along_rows).

    Restates general_pack_on_row (compress_weight.py:10-51).  n_rows*bits must be a multiple of 32.
    Values are masked to `bits` bits (the reference ORs un-masked values; on-grid inputs are identical).
    """
    v = np.asarray(values).astype(np.int64) & ((1 << bits) - 1)
    n_rows, n_cols = v.shape
    assert (n_rows * bits) % 32 == 0
    n_words = n_rows * bits // 32
    out = np.zeros((n_words, n_cols), dtype=np.uint64)
    k = np.arange(n_rows, dtype=np.int64)
    bit0 = k * bits
    w0 = bit0 // 32
    off = (bit0 % 32).astype(np.uint64).reshape(-1, 1)
    shifted = v.astype(np.uint64) << off  # up to 39 bits
    np.bitwise_or.at(out, w0, shifted & np.uint64(0xFFFFFFFF))
    spill = shifted >> np.uint64(32)
    has_spill = ((bit0 % 32) + bits) > 32
    if has_spill.any():
        np.bitwise_or.at(out, w0[has_spill] + 1, spill[has_spill])
    return out.astype(np.uint32).view(np.int32)


def unpack_along_cols(packed, bits: int, n_cols: int) -> np.ndarray:
    """packed i32 [R, n_cols*bits/32] -> int32 [R, n_cols]: the qzeros direction
    (compress_weight.py:121-127 calls general_unpack_on_row on the transposed view)."""
    return np.ascontiguousarray(unpack_along_rows(np.asarray(packed).T, bits, n_cols).T)


def pack_along_cols(values, bits: int) -> np.ndarray:
    return np.ascontiguousarray(pack_along_rows(np.asarray(values).T, bits).T)


def awq_interleave_cols(int_kn: np.ndarray) -> np.ndarray:
    """[R, N] natural column order -> [R, N] AWQ order so that packing 8 consecutive entries along N
    puts column 8j+AWQ_ORDER[i] in nibble i (quant_linear_awq.py:95-119)."""
    r, n = int_kn.shape
    assert n % 8 == 0
    idx = (np.arange(0, n, 8).reshape(-1, 1) + np.asarray(AWQ_ORDER).reshape(1, -1)).reshape(-1)
    return int_kn[:, idx]


def awq_deinterleave_cols(int_kn: np.ndarray) -> np.ndarray:
    r, n = int_kn.shape
    idx = (np.arange(0, n, 8).reshape(-1, 1) + np.asarray(AWQ_ORDER).reshape(1, -1)).reshape(-1)
    out = np.empty_like(int_kn)
    out[:, idx] = int_kn
    return out


# ----------------------------------------------------------------------------------------------
# integer grids from packed buffers, per layout
# ----------------------------------------------------------------------------------------------
def gptq_int_weight(qweight, bits: int, in_features: int) -> np.ndarray:
    """GPTQ/HQQ qweight i32 [K*bits/32, N] -> q int32 [K, N]  (quant_linear_gptq.py:26-36)."""
    return unpack_along_rows(qweight, bits, in_features)


def gptq_int_zeros(qzeros, bits: int, out_features: int, add_zero_bias: int = 0) -> np.ndarray:
    """GPTQ qzeros i32 [G, N*bits/32] -> z int32 [G, N]; `(z + COMPATIBLE_WITH_AUTOGPTQ) & mask`
    (quant_linear_gptq.py:18-24, 33-36)."""
    z = unpack_along_cols(qzeros, bits, out_features)
    return (z + int(add_zero_bias)) & ((1 << bits) - 1)


def awq_int_weight(qweight, in_features: int, out_features: int) -> np.ndarray:
    """AWQ-GEMM qweight i32 [K, N/8] -> q int32 [K, N] natural column order
    (quant_linear_awq.py:82-93 + reverse_reorder :121-140)."""
    q = unpack_along_cols(qweight, 4, out_features)  # [K, N] in AWQ nibble order
    assert q.shape[0] == in_features
    return awq_deinterleave_cols(q)


def awq_int_zeros(qzeros, out_features: int) -> np.ndarray:
    """AWQ qzeros i32 [G, N/8] -> z int32 [G, N] natural order (quant_linear_awq.py:76-80).
    AWQ never applies the AutoGPTQ offset at forward time."""
    return awq_deinterleave_cols(unpack_along_cols(qzeros, 4, out_features))


# ----------------------------------------------------------------------------------------------
# dequantisation  W[K, N] fp16
# ----------------------------------------------------------------------------------------------
def trivial_g_idx(in_features: int, groupsize: int) -> np.ndarray:
    return (np.arange(in_features, dtype=np.int64) // groupsize).astype(np.int32)


def _dequant_from_ints(q: np.ndarray, scales, zeros, g_idx, groupsize: int) -> np.ndarray:
    """W[k,n] = fp16( fp16(s[G,n]*q[k,n]) - fp16(z[G,n]*s[G,n]) ),  G = g_idx[k] or k//groupsize
    (quant_linear_gptq.py:38-48; quant_linear_hqq.py:22-24; compress_weight.py:105-111)."""
    s = np.asarray(scales).astype(np.float16)
    if np.asarray(zeros).dtype.kind == "f":
        z16 = np.asarray(zeros).astype(np.float16)
    else:
        z16 = np.asarray(zeros).astype(np.float16)  # ints 0..255 are exact in fp16
    k = q.shape[0]
    g = trivial_g_idx(k, groupsize) if g_idx is None else np.asarray(g_idx).astype(np.int64)
    sz = (z16 * s).astype(np.float16)  # one rounding
    sq = (s[g] * q.astype(np.float16)).astype(np.float16)  # one rounding
    return (sq - sz[g]).astype(np.float16)  # one rounding


def dequant_gptq(qweight, scales, qzeros, g_idx, bits: int, groupsize: int, in_features: int,
                 add_zero_bias: int = 0) -> np.ndarray:
    """DequantizeLinearBlockWise (quant_linear_gptq.py:13-52) -> W [K, N] fp16.
    `g_idx=None` is the no-act-order branch (:45-48)."""
    n = np.asarray(scales).shape[-1]
    q = gptq_int_weight(qweight, bits, in_features)
    z = gptq_int_zeros(qzeros, bits, n, add_zero_bias)
    return _dequant_from_ints(q, scales, z, g_idx, groupsize)


def dequant_hqq(qweight, scales, qzeros_f16, bits: int, groupsize: int, in_features: int) -> np.ndarray:
    """DequantAndUnpack.forward (quant_linear_hqq.py:8-28): fp16 un-packed zeros [G, N]."""
    q = gptq_int_weight(qweight, bits, in_features)
    return _dequant_from_ints(q, scales, np.asarray(qzeros_f16).astype(np.float16), None, groupsize)


def dequant_awq(qweight, scales, qzeros, groupsize: int, in_features: int) -> np.ndarray:
    """WQLinear_GEMM.unpack()[0].T  (compress_weight.py:136-151 with quant_linear_awq.py:76-93):
    W [K, N] fp16 = q*s[g] - (z*s)[g] with trivial g_idx."""
    n = np.asarray(scales).shape[-1]
    q = awq_int_weight(qweight, in_features, n)
    z = awq_int_zeros(qzeros, n)
    return _dequant_from_ints(q, scales, z, None, groupsize)


def dequant(layout: str, qweight, scales, qzeros, g_idx, bits, groupsize, in_features, add_zero_bias=0):
    layout = layout.upper()
    if layout == "GPTQ":
        return dequant_gptq(qweight, scales, qzeros, g_idx, bits, groupsize, in_features, add_zero_bias)
    if layout == "HQQ":
        return dequant_hqq(qweight, scales, qzeros, bits, groupsize, in_features)
    if layout in ("GEMM", "AWQ"):
        assert bits == 4
        return dequant_awq(qweight, scales, qzeros, groupsize, in_features)
    raise ValueError(layout)


def dequant_loops(layout: str, qweight, scales, qzeros, g_idx, bits, groupsize, in_features, add_zero_bias=0):
    """Element-by-element pure-Python restatement (small cases only): a second, independent derivation of
    the same layouts from SURVEY.md Appendix A, used to cross-check the vectorised functions above."""
    layout = layout.upper()
    qw = _as_u32(qweight)
    s = np.asarray(scales).astype(np.float16)
    n = s.shape[-1]
    k_total = in_features
    mask = (1 << bits) - 1
    out = np.zeros((k_total, n), dtype=np.float16)
    qz = np.asarray(qzeros)
    for k in range(k_total):
        grp = int(g_idx[k]) if g_idx is not None else k // groupsize
        for col in range(n):
            if layout in ("GPTQ", "HQQ"):
                bit0 = k * bits
                w, off = divmod(bit0, 32)
                val = int(qw[w, col]) >> off
                if off + bits > 32:
                    val |= int(qw[w + 1, col]) << (32 - off)
                q = val & mask
            else:  # AWQ GEMM: word (k, col//8); nibble i holds column 8j+ORDER[i]
                i = AWQ_ORDER.index(col % 8)
                q = (int(qw[k, col // 8]) >> (4 * i)) & 0xF
            if layout == "HQQ":
                z16 = np.float16(qz[grp, col])
            elif layout == "GPTQ":
                zb0 = col * bits
                w, off = divmod(zb0, 32)
                zw = _as_u32(qz)
                val = int(zw[grp, w]) >> off
                if off + bits > 32:
                    val |= int(zw[grp, w + 1]) << (32 - off)
                z16 = np.float16(((val & mask) + add_zero_bias) & mask)
            else:
                i = AWQ_ORDER.index(col % 8)
                z16 = np.float16((int(_as_u32(qz)[grp, col // 8]) >> (4 * i)) & 0xF)
            sc = s[grp, col]
            sq = np.float16(np.float32(sc) * np.float32(np.float16(q)))
            sz = np.float16(np.float32(z16) * np.float32(sc))
            out[k, col] = np.float16(np.float32(sq) - np.float32(sz))
    return out


# ----------------------------------------------------------------------------------------------
# forward  y = x @ W (+ bias)
# ----------------------------------------------------------------------------------------------
def is_act_order(g_idx, groupsize: int) -> bool:
    """QuantLinearGPTQ.forward lazy detect: `g_idx[:groupsize].sum() != 0` (quant_linear_gptq.py:137-138)."""
    return bool(np.asarray(g_idx)[:groupsize].astype(np.int64).sum() != 0)


def matmul_f16(x, w_kn, bias=None, num_threads: int | None = None):
    """torch.matmul(x, W) in the activations' dtype on CPU, then `+ bias` -- what the reference executes
    (quant_linear_gptq.py:85,142).  Returns a torch tensor shaped x.shape[:-1] + (N,)."""
    import torch

    if num_threads:
        torch.set_num_threads(num_threads)
    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    wt = w_kn if isinstance(w_kn, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w_kn))
    y = torch.matmul(xt, wt.to(xt.dtype))
    if bias is not None:
        bt = bias if isinstance(bias, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bias))
        y = y + bt.to(y.dtype)
    return y


def matmul_f16_via_f32(x, w_kn, bias=None):
    """Same contraction as matmul_f16 carried in fp32 and rounded once to fp16 (+ bias in fp16 like the reference).
    torch's CPU half GEMM accumulates in fp32 too, so the two agree to about an fp16 ulp (checked in
    tests/test_oracle_golden.py); this form is used where the half GEMM is impractically slow (M > 16 on hosts
    without native fp16 GEMM kernels)."""
    import torch

    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    wt = w_kn if isinstance(w_kn, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w_kn))
    y = torch.matmul(xt.float(), wt.float()).to(xt.dtype)
    if bias is not None:
        bt = bias if isinstance(bias, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bias))
        y = y + bt.to(y.dtype)
    return y


def matmul_f64(x, w_kn, bias=None) -> np.ndarray:
    """High-precision reference (error budgets): float64 accumulate of the same fp16 operands."""
    import torch

    xt = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    wt = w_kn if isinstance(w_kn, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(w_kn))
    y = xt.double().reshape(-1, xt.shape[-1]) @ wt.double()
    if bias is not None:
        bt = bias if isinstance(bias, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(bias))
        y = y + bt.double()
    return y.reshape(tuple(xt.shape[:-1]) + (wt.shape[-1],)).numpy()


def forward(layout: str, x, qweight, scales, qzeros, g_idx, bias, bits: int, groupsize: int, in_features: int,
            add_zero_bias: int = 0):
    """Layer forward as the reference's CPU path computes it.
    GPTQ: act-order only if the lazy detect fires (else g_idx ignored).  HQQ / AWQ: trivial groups."""
    layout = layout.upper()
    if layout == "GPTQ":
        gi = g_idx if (g_idx is not None and is_act_order(g_idx, groupsize)) else None
        w = dequant_gptq(qweight, scales, qzeros, gi, bits, groupsize, in_features, add_zero_bias)
    else:
        w = dequant(layout, qweight, scales, qzeros, None, bits, groupsize, in_features)
    return matmul_f16(x, w, bias)


# ----------------------------------------------------------------------------------------------
# packing (restates CompressWeight.pack_on_device / pack_qzeros so layer.pack() can be checked)
# ----------------------------------------------------------------------------------------------
def pack_gptq(q_kn, z_gn, bits: int, autogptq_compat: int = 0):
    """q int [K,N], z int [G,N] -> (qweight [K*bits/32, N], qzeros [G, N*bits/32])
    (compress_weight.py:174-186, 156-172: stored zero = (z - COMPAT) & mask)."""
    mask = (1 << bits) - 1
    zs = (np.asarray(z_gn).astype(np.int64) - int(autogptq_compat)) & mask
    return pack_along_rows(q_kn, bits), pack_along_cols(zs, bits)


def pack_awq(q_kn, z_gn):
    """q int [K,N], z int [G,N] -> (qweight [K, N/8], qzeros [G, N/8]) in AWQ-GEMM interleave
    (quant_linear_awq.py:70-74, 95-119; compress_weight.py:182-183)."""
    return pack_along_cols(awq_interleave_cols(np.asarray(q_kn)), 4), \
        pack_along_cols(awq_interleave_cols(np.asarray(z_gn)), 4)


def autogptq_fixup_qzeros(qzeros, bits: int, out_features: int) -> np.ndarray:
    """handle_qzeros_for_autogptq (quant_linear_gptq.py:119-134): z -> (z+1)&mask, repacked."""
    z = unpack_along_cols(qzeros, bits, out_features)
    z = (z + 1) & ((1 << bits) - 1)
    return pack_along_cols(z, bits)


# ----------------------------------------------------------------------------------------------
# ORT / MatMulNBits blob layout (4 bits)       quant_linear_onnxruntime.py:52-82 (dequant), :116-153 (pack)
#   qweight u8 [N, K/g, g/2]: byte b of block j of row n = q[g*j + 2b, n] | q[g*j + 2b + 1, n] << 4
#   scales  f16 [N * K/g]   : row-major (n, j)
#   qzeros  u8  [N * ceil(G/2)]: byte i of row n = z[2i, n] | z[2i+1, n] << 4 (odd G: last high nibble is padding)
#           or f16 [N, G]   : un-packed, non-integer zero points
#   W[n, k] = fp16((q - z) * s) for integer zeros (ONE rounding: int32 difference, then one fp16 multiply);
#             fp16(fp16(q - z) * s) for fp16 zeros (the difference is rounded to fp16 first)
#   act-order: z, s taken at block g_idx[k]
# ----------------------------------------------------------------------------------------------
def ort_int_weight(qweight_u8) -> np.ndarray:
    """[N, G, g/2] u8 -> q[K, N] int32."""
    qw = np.asarray(qweight_u8, dtype=np.uint8)
    n = qw.shape[0]
    lo, hi = (qw & 0x0F).astype(np.int32), (qw >> 4).astype(np.int32)
    q_nk = np.stack([lo, hi], axis=-1).reshape(n, -1)
    return np.ascontiguousarray(q_nk.T)


def ort_int_zeros(qzeros_u8, out_features: int, n_blocks: int) -> np.ndarray:
    """flat u8 nibble pairs -> z[G, N] int32."""
    zb = np.asarray(qzeros_u8, dtype=np.uint8).reshape(out_features, -1)
    z_ng = np.stack([(zb & 0x0F).astype(np.int32), (zb >> 4).astype(np.int32)], axis=-1).reshape(out_features, -1)
    return np.ascontiguousarray(z_ng[:, :n_blocks].T)


def pack_ort(q_kn, z_gn, scales_gn):
    """(q[K,N], z[G,N] int or f16, scales[G,N]) -> (qweight u8 [N,G,g/2], qzeros, scales_flat) as pack_on_device does."""
    q_nk = np.asarray(q_kn).T.astype(np.uint8)
    n, k = q_nk.shape
    g_blocks = np.asarray(scales_gn).shape[0]
    qweight = (q_nk[:, 0::2] | (q_nk[:, 1::2] << 4)).reshape(n, g_blocks, k // g_blocks // 2)
    z = np.asarray(z_gn)
    if z.dtype == np.float16:
        qzeros = np.ascontiguousarray(z.T)
    else:
        z_ng = z.T.astype(np.uint8)
        if z_ng.shape[1] & 1:
            z_ng = np.concatenate([z_ng, np.zeros((n, 1), np.uint8)], axis=1)
        qzeros = (z_ng[:, 0::2] | (z_ng[:, 1::2] << 4)).reshape(-1)
    return np.ascontiguousarray(qweight), qzeros, np.ascontiguousarray(np.asarray(scales_gn, np.float16).T).reshape(-1)


def dequant_ort(qweight_u8, scales_flat, qzeros, g_idx, groupsize: int, in_features: int, out_features: int) -> np.ndarray:
    """W[N, K] float16, exactly as dequantize_blockwise_4bits (quant_linear_onnxruntime.py:52-82)."""
    q_kn = ort_int_weight(qweight_u8)[:in_features]
    n_blocks = np.asarray(qweight_u8).shape[1]
    s_gn = np.asarray(scales_flat, dtype=np.float16).reshape(out_features, n_blocks).T
    gi = trivial_g_idx(in_features, groupsize) if g_idx is None else np.asarray(g_idx, dtype=np.int64)
    qz = np.asarray(qzeros)
    if qz.dtype == np.float16:
        z_gn = qz.reshape(out_features, n_blocks).T
        diff = (q_kn.astype(np.float32) - z_gn[gi].astype(np.float32)).astype(np.float16)  # int - half -> half
    else:
        z_gn = ort_int_zeros(qz, out_features, n_blocks)
        diff = (q_kn - z_gn[gi]).astype(np.float16)                                        # exact small integers
    w_kn = (diff.astype(np.float32) * s_gn[gi].astype(np.float32)).astype(np.float16)      # one fp16 multiply
    return np.ascontiguousarray(w_kn.T)


def ort_is_act_order(g_idx) -> bool:
    """dequantize_blockwise_4bits gathers per k only if g_idx[:32] is not all zero (:70)."""
    return g_idx is not None and int(np.asarray(g_idx)[:32].sum()) != 0


# ----------------------------------------------------------------------------------------------
# parity metric
# ----------------------------------------------------------------------------------------------
# ---- the library's native (strip-major) layout, restated (no reference counterpart: it is THIS repo's load-time layout) ------
# include/qllm_mi355x.h "native layout": pins what qllm_repack_native must produce from the integer grids the functions above
# recover from the reference's buffers.
def native_layout(q_kn, scales_gn, zeros_gn, bits: int, zeros_f16: bool = False):
    """(qweight i32 [N/16, K*bits/32, 16], scales f16 [N/16, G, 16], qzeros) from q[K, N] ints, scales [G, N], zeros [G, N]
    (stored integer zero points, fp16 zero points when `zeros_f16`, or None).  Packed zero points: u32 [N/16, G, 2] -- the
    64-bit little-endian pair holds column 16 s + i at bit `bits` * i."""
    q = np.asarray(q_kn)
    k, n = q.shape
    assert n % 16 == 0 and (k * bits) % 32 == 0
    rows = pack_along_rows(q, bits)                                   # [K*bits/32, N]: the GPTQ row stream
    qweight = np.ascontiguousarray(rows.reshape(rows.shape[0], n // 16, 16).transpose(1, 0, 2))
    sc = np.asarray(scales_gn, dtype=np.float16)
    scales = np.ascontiguousarray(sc.reshape(sc.shape[0], n // 16, 16).transpose(1, 0, 2))
    if zeros_gn is None:
        return qweight, scales, None
    if zeros_f16:
        z = np.asarray(zeros_gn, dtype=np.float16)
        return qweight, scales, np.ascontiguousarray(z.reshape(z.shape[0], n // 16, 16).transpose(1, 0, 2))
    z = np.asarray(zeros_gn).astype(np.uint64).reshape(-1, n // 16, 16)
    pair = np.zeros(z.shape[:2], dtype=np.uint64)
    for i in range(16):
        pair |= (z[:, :, i] & np.uint64((1 << bits) - 1)) << np.uint64(bits * i)
    qz = np.stack([(pair & np.uint64(0xFFFFFFFF)).astype(np.uint32), (pair >> np.uint64(32)).astype(np.uint32)], axis=-1)
    return qweight, scales, np.ascontiguousarray(qz.transpose(1, 0, 2)).view(np.int32)


def rel_err(y, y_ref) -> float:
    """max_abs(y - y_ref) / max_abs(y_ref)"""
    a = np.asarray(y, dtype=np.float64)
    b = np.asarray(y_ref, dtype=np.float64)
    denom = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (denom if denom > 0 else 1.0)
