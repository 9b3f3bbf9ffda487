"""QuantLinearORT with the reference's contract (qllm/modeling/q_layers/quant_linear_onnxruntime.py:85-174): the ORT /
MatMulNBits blob layout, 4 bits.

Buffers (state-dict compatible):
    qweight u8  [N, K/g, g/2]        byte b of block j of row n = q[g*j+2b, n] | q[g*j+2b+1, n] << 4
    qzeros  u8  [(G + G%2) * N/2]    per row ceil(G/2) bytes, two 4-bit zero points per byte, low nibble first
            (or the module dtype, [N, G], when packed from non-integer zero points)
    scales  dtype [N * G]            row-major (n, block)
    g_idx   i32 [K] (registered buffer; default k // g), bias dtype [N] or None

forward: the reference dequantises the whole W[N,K] every call and runs a dense matmul (:33-44).  Here the blob is viewed
once as what it already is -- row n of the blob, read as 32-bit words, is column n of a GPTQ row-stream qweight (8
little-endian nibbles = 8 consecutive k) -- so `blob.view(int32)[N, K/8].T` plus the transposed scales / zero points is a
row-stream layer the fused MI355X kernels stream directly.  That transposed copy is built on first use and cached; the
state dict is untouched.  Act-order layers additionally get their rows sorted by block (x is gathered to match).
Rounding: the fused kernels evaluate s*(q-z) (decode: fp32, unrounded; prefill: fp16(fp16(s*q) - fp16(s*z))), the
reference fp16((q-z)*s) -- within 2 ulp of W, far inside the 1e-2 parity bound; `unpack()` / `ort_ops.Dequantize4Bits`
reproduce the reference's W bit for bit.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ._hip_forward import HipForwardMixin, export_module_hooks, _tkey
from .compress_weight import CompressWeight


def dequantize_blockwise_4bits(quant_values, scale, zero_point, g_idx, rows, cols):
    """Same name / arguments / returns as the reference's torch helper (:52-82): (W[cols, rows], zeros[cols, G],
    scales[cols, G]).  On a HIP device W comes from the library kernel (bit-identical); on CPU from this torch code."""
    n_blocks = quant_values.shape[1]
    scale2 = scale.reshape(cols, n_blocks)
    if zero_point.dtype == scale.dtype:
        zeros = zero_point.reshape(cols, -1)[:, :n_blocks]
    else:
        zb = zero_point.reshape(cols, -1)
        zeros = torch.stack([zb & 0x0F, zb >> 4], dim=-1).reshape(cols, -1)[:, :n_blocks].to(torch.int32)
    act = g_idx is not None and bool(g_idx[:32].sum().item() != 0)
    if quant_values.is_cuda:
        from ... import ops
        w = ops.ort_dequantize4bits(quant_values, scale, zero_point, g_idx if act else None,
                                    quant_values.shape[2] * 2, rows, cols)
    else:
        q = torch.stack([quant_values & 0x0F, quant_values >> 4], dim=-1).reshape(cols, -1).to(torch.int32)
        if act:
            gi = g_idx.long()
            w = ((q - zeros[:, gi]) * scale2[:, gi]).to(scale.dtype)
        else:
            blk = quant_values.shape[2] * 2
            w = ((q.reshape(cols, n_blocks, blk) - zeros.unsqueeze(-1)) * scale2.unsqueeze(-1)).to(scale.dtype)
            w = w.reshape(cols, -1)
        w = w[:, :rows]
    return w, zeros, scale2


@export_module_hooks
class QuantLinearORT(nn.Module, CompressWeight, HipForwardMixin):
    def __init__(self, bits, groupsize, infeatures, outfeatures, bias, dtype=None):
        super().__init__()
        self.dtype = torch.get_default_dtype() if dtype is None else dtype
        if bits not in [2, 3, 4, 5, 6, 7, 8]:
            raise NotImplementedError("Only 2,4,5,6,7,8 bits are supported.")
        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.bits = bits
        self.orig_fp_weight = None
        self.maxq = 2 ** self.bits - 1
        self.groupsize = groupsize if groupsize != -1 else infeatures
        self.act_order = None
        self.pack_mode = "ORT"
        q_rows = infeatures // self.groupsize
        self.register_buffer("qweight", torch.zeros((outfeatures, q_rows, self.groupsize // (8 // bits)), dtype=torch.uint8))
        self.register_buffer("qzeros", torch.zeros((q_rows + (q_rows & 1)) * (outfeatures // 8 * self.bits), dtype=torch.uint8))
        self.register_buffer("scales", torch.zeros((math.ceil(infeatures / self.groupsize) * outfeatures), dtype=self.dtype))
        self.register_buffer("g_idx", torch.tensor([i // self.groupsize for i in range(infeatures)], dtype=torch.int32))
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures), dtype=self.dtype))
        else:
            self.bias = None

    # ---- pack / unpack (load-time format conversion; torch, any device) -------------------------------------------------
    def pack_on_device(self, intweight_gpu, intzeros_T):
        """intweight [K, N] int, intzeros_T [G, N] (int, or the scales' dtype for real-valued zero points); self.scales
        is [G, N] at this point (set by accelerate_pack_on_device) and becomes the flat [N*G] blob (:116-153)."""
        self.act_order = bool(self.g_idx[: self.groupsize // self.bits].sum().item() != 0)
        assert self.bits == 4, "only 4bit is supported by ONNXRUNTIME for now."
        rows, cols = intweight_gpu.shape
        assert rows % self.groupsize == 0, "in_features must be a multiple of the block size"
        k_blocks = rows // self.groupsize
        q_nk = intweight_gpu.T.to(torch.uint8)
        self.qweight = (q_nk[:, 0::2] | (q_nk[:, 1::2] << 4)).reshape(cols, k_blocks, self.groupsize // 2).contiguous()
        scales_ng = self.scales.T.to(intweight_gpu.device)
        float_zeros = intzeros_T.dtype == self.scales.dtype
        z_ng = intzeros_T.T if float_zeros else intzeros_T.T.to(torch.uint8)
        if z_ng.shape[1] & 1:
            z_ng = torch.nn.functional.pad(z_ng, (0, 1, 0, 0), "constant", 0)
        if float_zeros:
            self.qzeros = z_ng.contiguous()
        else:
            self.qzeros = (z_ng[:, 0::2] | (z_ng[:, 1::2] << 4)).reshape(-1).contiguous()
        self.scales = scales_ng.reshape(-1).contiguous()
        self._desc = None

    # integer-domain accessors (used by qllm_amd.repack; same names as CompressWeight's)
    def unpack_qweight(self, device):
        """-> q[K, N] int32 in natural order."""
        qw = self.qweight.to(device)
        n = qw.shape[0]
        return torch.stack([qw & 0x0F, qw >> 4], dim=-1).reshape(n, -1).T.contiguous().to(torch.int32)

    def unpack_qzeros(self, device):
        """-> zeros[G, N]: int32 for packed zero points, the stored floating dtype otherwise."""
        groups = self.infeatures // self.groupsize
        qz = self.qzeros.to(device)
        if qz.dtype != torch.uint8:
            return qz.reshape(self.outfeatures, -1)[:, :groups].T.contiguous()
        zb = qz.reshape(self.outfeatures, -1)
        return torch.stack([zb & 0x0F, zb >> 4], dim=-1).reshape(self.outfeatures, -1)[:, :groups].T.contiguous().to(torch.int32)

    def scales_gn(self):
        """scales as [G, N] (the other layers' arrangement)."""
        return self.scales.reshape(self.outfeatures, -1).T.contiguous()

    def unpack(self):
        """-> (W[N,K], scales[G,N], zeros[G,N]) on CPU, like the reference (:156-167)."""
        w, zeros, scales = dequantize_blockwise_4bits(self.qweight, self.scales, self.qzeros, self.g_idx,
                                                      self.infeatures, self.outfeatures)
        return (w.contiguous().to("cpu"), scales.T.contiguous().to("cpu"), zeros.T.contiguous().to("cpu"))

    # ---- forward: row-stream view of the blob, cached ---------------------------------------------------------------------
    _perm = None

    def row_stream_view(self):
        """(qweight i32 [K/8, N], scales f16 [G, N], zeros f16 [G, N]): the blob re-read as a row-stream (GPTQ/HQQ-style)
        layer.  Pure tensor views / transposes, any device; the integers are untouched."""
        n, k, g = self.outfeatures, self.infeatures, self.groupsize
        if self.bits != 4:
            raise NotImplementedError("the ORT blob layout is 4-bit only")
        if k % g != 0 or k % 8 != 0:
            raise RuntimeError(f"QuantLinearORT needs in_features % groupsize == 0 (K={k}, g={g})")
        groups = k // g
        # row n of the blob as K/8 little-endian words == column n of a GPTQ row-stream qweight
        qw = self.qweight.reshape(n, k // 2).view(torch.int32).T.contiguous()
        scales = self._f16(self.scales).reshape(n, groups).T.contiguous()
        if self.qzeros.dtype == torch.uint8:
            zb = self.qzeros.reshape(n, -1)
            zeros = torch.stack([zb & 0x0F, zb >> 4], dim=-1).reshape(n, -1)[:, :groups].T.to(torch.float16).contiguous()
        else:
            zeros = self._f16(self.qzeros).reshape(n, -1)[:, :groups].T.contiguous()
        return qw, scales, zeros

    def _layout_name(self):
        return "HQQ"  # the cached view carries un-packed fp16 zero points [G, N]

    def _descriptor(self, act_order_g_idx=None, add_zero_bias: int = 0):
        from ... import ops
        key = (_tkey(self.qweight), _tkey(self.scales), _tkey(self.qzeros), _tkey(self.g_idx), _tkey(self.bias))
        if self._desc is None or key != self._desc_key:   # (False = cached verdict "irregular act-order": not rebuilt)
            if self.bits != 4:
                raise NotImplementedError("the ORT blob layout is 4-bit only")
            n, k, g = self.outfeatures, self.infeatures, self.groupsize
            groups = k // g if g else 0
            dev = self.qweight.device
            qw, scales, zeros = self.row_stream_view()
            self._perm = None
            if self.act_order is None:
                self.act_order = bool(self.g_idx[:32].sum().item() != 0)
            if self.act_order:
                gi = self.g_idx.to(dev).long()
                counts = torch.bincount(gi, minlength=groups)
                if counts.numel() != groups or not bool((counts == g).all()):
                    # blocks of unequal size cannot be laid out as contiguous groups: such a layer is served the way the
                    # reference serves every layer -- Dequantize4Bits (per-channel gather) + a dense GEMM, on device
                    self._desc, self._desc_keep, self._desc_key = False, None, key
                    return None
                perm = torch.argsort(gi, stable=True)
                q = ops.unpack_qweight(qw, "GPTQ", 4, k, n)
                qw = ops.pack_qweight(q.index_select(0, perm).contiguous(), "GPTQ", 4)
                self._perm = perm
            b = self._f16(self.bias).contiguous() if self.bias is not None else None
            self._desc, self._desc_keep = ops.make_weight("HQQ", qw, scales, zeros, None, b, k, n, g, 4, 0)
            self._desc_key = key
        return self._desc if self._desc else None

    _RELEASABLE = ()   # the blob itself stays (its regeneration would need the transposes back); the row-stream VIEW goes

    def _native_source(self):
        if self._descriptor() is None:
            return None
        return self._desc, self._desc_keep

    def native_descriptor(self, add_zero_bias: int = 0):
        """The native copy is cached on the buffers' identity / version (HipForwardMixin); the transposed row-stream view was
        only its source and is dropped once the copy exists (one derived copy, not two).  `_perm` / `act_order` stay."""
        w = HipForwardMixin.native_descriptor(self, 0)
        if w is not None:
            self._desc = self._desc_keep = None
        return w

    def materialize_reference(self):
        pass

    def decode_descriptor(self, act_order_g_idx=None, add_zero_bias: int = 0):
        w = self.native_descriptor(0)
        return w if w is not None else self._descriptor()

    def forward(self, x):
        # the cached native copy first (a key comparison); the row-stream view is rebuilt only when there is no native copy
        if self.native_descriptor(0) is None and self._descriptor() is None:
            # irregular act-order: the reference's own two-step path, on device
            from ... import ops
            w = ops.ort_dequantize4bits(self.qweight, self.scales, self.qzeros, self.g_idx, self.groupsize,
                                        self.infeatures, self.outfeatures)
            y = torch.matmul(x, w.to(x.dtype).T)
            return y + self.bias.to(y.dtype) if self.bias is not None else y
        if self._perm is not None:
            x = x.index_select(-1, self._perm)
        return self._hip_linear(x, None, 0)
