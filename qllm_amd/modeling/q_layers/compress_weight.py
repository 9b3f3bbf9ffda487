"""Bit layouts and the pack / unpack contract of the quantized layers.

Mirrors the interface of the reference's qllm/modeling/q_layers/compress_weight.py (function and method names,
argument meaning, tensor shapes) with an independent implementation:

  * every packed tensor is a little-endian bit stream (value k at stream bits [k*bits, (k+1)*bits)); one
    vectorised int64 routine handles all widths 2..8 instead of the reference's 2/4/8 fast path + per-bit path
    (compress_weight.py:10-43, 54-84);
  * on a HIP device qweight (un)packing runs in the library's kernels (qllm_pack_qweight / qllm_unpack_qweight),
    and unpack()'s dequantisation runs in qllm_dequant -- bit-identical to the torch formula below.

pack()/unpack() are load-time format conversion, not the forward path; they work on CPU tensors too.
"""
from __future__ import annotations

import math
import os

import torch


def _stream_axis(pack_tensor: torch.Tensor, int_tensor: torch.Tensor) -> int:
    """The reference infers the packing direction from shapes (compress_weight.py:11,29,56,71): equal dim 0 ->
    the stream runs along dim 1 (qzeros), otherwise along dim 0 (qweight)."""
    assert pack_tensor.shape[0] == int_tensor.shape[0] or pack_tensor.shape[1] == int_tensor.shape[1], ''
    return 1 if pack_tensor.shape[0] == int_tensor.shape[0] else 0


def pack_bitstream(values: torch.Tensor, bits: int, axis: int = 0) -> torch.Tensor:
    """int [R, C] -> int32 bit stream along `axis` (length*bits must be a multiple of 32)."""
    v = values if axis == 0 else values.T
    n, c = v.shape
    assert (n * bits) % 32 == 0, "stream length must fill whole 32-bit words"
    dev = v.device
    v64 = v.to(torch.int64) & ((1 << bits) - 1)
    bit0 = torch.arange(n, device=dev, dtype=torch.int64) * bits
    word, off = bit0 // 32, bit0 % 32
    shifted = v64 << off.unsqueeze(1)  # < 2^39
    acc = torch.zeros((n * bits // 32 + 1, c), dtype=torch.int64, device=dev)
    acc.index_add_(0, word, shifted & 0xFFFFFFFF)  # bit fields are disjoint: add == or
    acc.index_add_(0, word + 1, shifted >> 32)
    out = acc[:-1]
    out = torch.where(out >= (1 << 31), out - (1 << 32), out).to(torch.int32)
    return out.contiguous() if axis == 0 else out.T.contiguous()


def unpack_bitstream(packed: torch.Tensor, bits: int, length: int, axis: int = 0) -> torch.Tensor:
    """int32 bit stream along `axis` -> int32 values, `length` of them along that axis."""
    p = packed if axis == 0 else packed.T
    dev = p.device
    p64 = p.to(torch.int64) & 0xFFFFFFFF
    p64 = torch.cat([p64, torch.zeros((1, p64.shape[1]), dtype=torch.int64, device=dev)], dim=0)
    bit0 = torch.arange(length, device=dev, dtype=torch.int64) * bits
    word, off = bit0 // 32, (bit0 % 32).unsqueeze(1)
    both = p64[word] | (p64[word + 1] << 32)
    out = ((both >> off) & ((1 << bits) - 1)).to(torch.int32)
    return out.contiguous() if axis == 0 else out.T.contiguous()


def general_pack_on_row(pack_tensor: torch.Tensor, ori_int32_tensor: torch.Tensor, bits: int):
    """In-place: fill `pack_tensor` with the bit stream of `ori_int32_tensor` (reference compress_weight.py:46-51)."""
    axis = _stream_axis(pack_tensor, ori_int32_tensor)
    pack_tensor.copy_(pack_bitstream(ori_int32_tensor, bits, axis))


def general_unpack_on_row(pack_tensor: torch.Tensor, ori_int32_tensor: torch.Tensor, bits: int):
    """In-place: fill `ori_int32_tensor` from the bit stream in `pack_tensor` (reference compress_weight.py:87-92)."""
    axis = _stream_axis(pack_tensor, ori_int32_tensor)
    ori_int32_tensor.copy_(unpack_bitstream(pack_tensor, bits, ori_int32_tensor.shape[axis], axis))


def _on_hip(t: torch.Tensor) -> bool:
    return t.is_cuda


class CompressWeight(object):
    """Mixin shared by QuantLinearGPTQ / WQLinear_GEMM / QuantLinearHQQ (reference compress_weight.py:95-210).
    Expects the host class to define: bits, groupsize, infeatures, outfeatures, qweight, qzeros, scales, g_idx,
    orig_fp_weight, pack_mode."""

    def __init__(self, dtype=torch.float16):
        self.dtype = dtype

    # ---- quantise / dequantise on an integer grid -----------------------------------------------------------
    def _quant_weight(self, weight, scales, zeros, g_idx, need_transpose=True):
        # weight [K,N]; scales/zeros [G,N]; un-clamped round, like the reference (compress_weight.py:98-103):
        # callers must hand in on-grid weights.
        sz = zeros * scales
        return torch.round((weight + sz[g_idx]) / scales[g_idx]).to(torch.int)

    def _dequant_weight(self, intweight, scales, zeros, g_idx):
        # W = q*s[g] - (z*s)[g], each op rounded in the tensors' dtype (compress_weight.py:105-111)
        sz = zeros * scales
        return intweight * scales[g_idx] - sz[g_idx].to(self.dtype)

    def weight_qdq(self, linear, scales, zeros, g_idx=None):
        if linear.bias is not None:
            self.bias = linear.bias.clone().to(self.dtype)
        g_idx = self.g_idx.to(scales.device) if g_idx is None else g_idx
        q = self._quant_weight(linear.weight.data.T, scales.T, zeros.T, g_idx)
        return self._dequant_weight(q, scales.T, zeros.T, g_idx).T

    # ---- unpack -------------------------------------------------------------------------------------------------
    def _work_device(self):
        return self.qweight.device

    def _real_buffers(self):
        """q_layers may hold their packed buffers as placeholders while the native copy serves the forward
        (HipForwardMixin.release_reference): every accessor below works on the real tensors."""
        m = getattr(self, "materialize_reference", None)
        if m is not None:
            m()

    def unpack_qzeros(self, device):
        self._real_buffers()
        qzeros = self.qzeros.to(device)
        groups = math.ceil(self.infeatures / self.groupsize)
        zeros = torch.zeros((groups, self.outfeatures), dtype=torch.int32, device=device)
        general_unpack_on_row(qzeros, zeros, self.bits)
        return zeros

    def unpack_qweight(self, device):
        self._real_buffers()
        qweight = self.qweight.to(device)
        if _on_hip(qweight):
            from ... import ops
            return ops.unpack_qweight(qweight.contiguous(), "GPTQ", self.bits, self.infeatures, self.outfeatures)
        weight = torch.zeros((self.infeatures, qweight.shape[1]), dtype=torch.int32, device=device)
        general_unpack_on_row(qweight, weight, self.bits)
        return weight

    def unpack(self):
        self._real_buffers()
        """-> (W[N,K] in self.dtype, scales[G,N], zeros[G,N]) on CPU (reference compress_weight.py:136-151).
        The stored zeros are used as-is (no AutoGPTQ offset), exactly like the reference."""
        device = self._work_device()
        scales = self.scales.to(device)
        zeros = self.unpack_qzeros(device)
        weight = self.unpack_qweight(device)
        w = self._dequant_weight(weight, scales, zeros, self.g_idx.to(device).long()).T
        return (w.to("cpu"), scales.to("cpu"), zeros.to("cpu"))

    # ---- pack -----------------------------------------------------------------------------------------------------
    def reorder_int_tensor(self, int_tensor):
        return int_tensor

    def pack_qzeros(self, intzeros, device):
        """intzeros [G,N] -> self.qzeros i32 [G, N*bits/32]; stored value = (z - COMPATIBLE_WITH_AUTOGPTQ) & mask
        (reference compress_weight.py:156-172)."""
        assert max(1, intzeros.shape[1] // 32 * self.bits) == int(round(intzeros.shape[1] * self.bits / 32 + 0.5))
        compat = int(os.environ.get("COMPATIBLE_WITH_AUTOGPTQ", "0"))
        z = (intzeros.to(torch.int64) - compat) & (2 ** self.bits - 1)
        self.qzeros = pack_bitstream(z.to(torch.int32), self.bits, axis=1).to("cpu")

    def pack_on_device(self, intweight_gpu, qzeros):
        """intweight [K,N] int, qzeros [G,N] -> self.qweight / self.qzeros (reference compress_weight.py:174-190)."""
        device = intweight_gpu.device
        intweight_gpu = self.reorder_int_tensor(intweight_gpu)
        rows = intweight_gpu.shape[0]
        assert rows // 32 * self.bits == int(round(rows * self.bits / 32 + 0.5))
        if _on_hip(intweight_gpu) and "GEMM" not in self._get_name():
            from ... import ops
            qweight = ops.pack_qweight(intweight_gpu.to(torch.int32).contiguous(), "GPTQ", self.bits)
        else:
            qweight = pack_bitstream(intweight_gpu, self.bits, axis=0)
        if "GEMM" in self._get_name():
            qweight = qweight.T.contiguous()
        self.qweight = qweight.to("cpu")
        self.pack_qzeros(qzeros, device)
        if self.orig_fp_weight is not None:
            fw, _, iz = self.unpack()
            assert (fw == self.orig_fp_weight.to(fw.device)).all()

    def accelerate_pack_on_device(self, layer_weight, scales, zeros, g_idx=None, device="cuda"):
        self.scales = scales.T.contiguous().to(self.dtype).to("cpu")
        if g_idx is None:
            g_idx = self.g_idx.to(device)
        else:
            self.g_idx = g_idx.clone().to("cpu")
        intweight = self._quant_weight(layer_weight.T, scales.T, zeros.T, g_idx.long())
        return self.pack_on_device(intweight, zeros.T.contiguous())

    def pack(self, linear, scales, zeros, g_idx=None):
        """linear.weight [N,K] on-grid, scales/zeros [N,G] (reference compress_weight.py:205-210)."""
        device = "cuda" if torch.cuda.is_available() else "cpu"
        g = g_idx.to(device) if g_idx is not None else None
        return self.accelerate_pack_on_device(linear.weight.data.to(device), scales.to(device), zeros.to(device), g,
                                              device)
