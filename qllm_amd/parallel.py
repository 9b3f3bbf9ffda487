"""Tensor-parallel wrappers for the quantized linears: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm).  Net-new relative to the reference, which has no distributed code; the
oracle is "sharded result == unsharded result" (section 8e).

  ColumnParallelQuantLinear   rank r owns output columns [r*N/P, (r+1)*N/P): packed words are column-separable in
                              every layout (GPTQ/HQQ: qweight[:, cols]; AWQ: 8-column words, qweight[:, cols/8]), so a
                              shard is a plain slice and the local forward is the same fused kernel on [M, N/P].
                              gather_output=False (Megatron style) leaves y sharded for a following RowParallel layer;
                              gather_output=True does ONE collective: all_gather (default) or, as BASELINE.json words it,
                              all_reduce of the zero-padded [M, N] (numerically identical: the slices are disjoint).
  RowParallelQuantLinear      rank r owns input rows [r*K/P, (r+1)*K/P) (whole quantisation groups); partial products
                              are summed with ONE all_reduce.  Trivial g_idx only.

xGMI note (MI355X: 8 GPUs fully connected, 7 links x ~153 GB/s each): a decode-sized all-reduce (16 KB at hidden 8192)
is latency-bound, a prefill-sized one (32 MB) bandwidth-bound with a direct reduce-scatter + all-gather using all 7
links; RCCL picks the algorithm, the wrappers issue exactly one collective per layer pair on the compute stream.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .modeling.q_layers import QuantLinearGPTQ, QuantLinearHQQ, WQLinear_GEMM


def _world(group):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard_columns(layer: nn.Module, rank: int, world: int) -> nn.Module:
    """New q_layer of the same class holding output columns [rank*N/world, (rank+1)*N/world) of `layer`."""
    n, k, bits, g = layer.outfeatures, layer.infeatures, layer.bits, layer.groupsize
    if n % world != 0:
        raise ValueError(f"out_features {n} not divisible by world size {world}")
    nl = n // world
    c0, c1 = rank * nl, (rank + 1) * nl
    cls = type(layer)
    word_cols = 32 // math.gcd(32, bits)  # columns per whole packed-zero word boundary
    if isinstance(layer, WQLinear_GEMM):
        if nl % 8 != 0:
            raise ValueError("AWQ shards must be whole 8-column words")
        new = cls(bits, g, k, nl, layer.bias is not None, dtype=layer.scales.dtype)
        new.qweight = layer.qweight[:, c0 // 8:c1 // 8].contiguous()
        new.qzeros = layer.qzeros[:, c0 // 8:c1 // 8].contiguous()
    elif isinstance(layer, QuantLinearHQQ):
        new = cls(bits, g, k, nl, layer.bias is not None, dtype=layer.scales.dtype)
        new.qweight = layer.qweight[:, c0:c1].contiguous()
        new.qzeros = layer.qzeros[:, c0:c1].contiguous()
    elif isinstance(layer, QuantLinearGPTQ):
        if nl % word_cols != 0:
            raise ValueError(f"GPTQ shards must align to packed-zero words ({word_cols} columns at {bits} bits)")
        new = cls(bits, g, k, nl, layer.bias is not None, dtype=layer.scales.dtype)
        new.qweight = layer.qweight[:, c0:c1].contiguous()
        new.qzeros = layer.qzeros[:, c0 * bits // 32:c1 * bits // 32].contiguous()
        new.g_idx = layer.g_idx.clone()
        new.act_order = layer.act_order
    else:
        raise TypeError(f"cannot shard {cls.__name__}")
    new.scales = layer.scales[:, c0:c1].contiguous()
    if layer.bias is not None:
        new.bias = layer.bias[c0:c1].contiguous()
    return new


def shard_rows(layer: nn.Module, rank: int, world: int) -> nn.Module:
    """New q_layer holding input rows [rank*K/world, (rank+1)*K/world) (whole groups).  bias stays on rank 0."""
    n, k, bits, g = layer.outfeatures, layer.infeatures, layer.bits, layer.groupsize
    if k % world != 0 or (k // world) % g != 0 or ((k // world) * bits) % 32 != 0:
        raise ValueError(f"in_features {k} / {world} must be whole groups of {g} and whole packed words")
    kl = k // world
    k0, k1 = rank * kl, (rank + 1) * kl
    trivial = torch.equal(layer.g_idx.cpu().to(torch.int64), torch.arange(k) // g)
    if not trivial:
        raise ValueError("row-parallel sharding needs a trivial g_idx (no act-order)")
    cls = type(layer)
    has_bias = layer.bias is not None and rank == 0
    new = cls(bits, g, kl, n, has_bias, dtype=layer.scales.dtype)
    if isinstance(layer, WQLinear_GEMM):
        new.qweight = layer.qweight[k0:k1].contiguous()
    else:
        new.qweight = layer.qweight[k0 * bits // 32:k1 * bits // 32].contiguous()
    new.qzeros = layer.qzeros[k0 // g:k1 // g].contiguous()
    new.scales = layer.scales[k0 // g:k1 // g].contiguous()
    if has_bias:
        new.bias = layer.bias.clone()
    return new


class ColumnParallelQuantLinear(nn.Module):
    def __init__(self, shard: nn.Module, out_features: int, group=None, gather_output: bool = True,
                 collective: str = "all_gather"):
        super().__init__()
        assert collective in ("all_gather", "all_reduce")
        self.shard = shard
        self.out_features = out_features
        self.group = group
        self.gather_output = gather_output
        self.collective = collective

    @classmethod
    def from_full(cls, layer: nn.Module, group=None, gather_output: bool = True, collective: str = "all_gather"):
        rank, world = _world(group)
        return cls(shard_columns(layer, rank, world), layer.outfeatures, group, gather_output, collective)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.shard(x)  # [..., N/P] through the fused kernel
        rank, world = _world(self.group)
        if not self.gather_output or world == 1:
            return y
        nl = y.shape[-1]
        if self.collective == "all_gather":
            y2 = y.reshape(-1, nl).contiguous()
            parts = torch.empty((world * y2.shape[0], nl), dtype=y.dtype, device=y.device)
            dist.all_gather_into_tensor(parts, y2, group=self.group)  # rank-major concatenation along dim 0
            full = parts.view(world, y2.shape[0], nl).permute(1, 0, 2).reshape(y2.shape[0], world * nl)
            return full.reshape(tuple(y.shape[:-1]) + (world * nl,))
        full = torch.zeros(tuple(y.shape[:-1]) + (world * nl,), dtype=y.dtype, device=y.device)
        full[..., rank * nl:(rank + 1) * nl] = y
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)  # disjoint slices: exact
        return full


class RowParallelQuantLinear(nn.Module):
    def __init__(self, shard: nn.Module, group=None, input_is_parallel: bool = True):
        super().__init__()
        self.shard = shard
        self.group = group
        self.input_is_parallel = input_is_parallel

    @classmethod
    def from_full(cls, layer: nn.Module, group=None, input_is_parallel: bool = True):
        rank, world = _world(group)
        return cls(shard_rows(layer, rank, world), group, input_is_parallel)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rank, world = _world(self.group)
        if not self.input_is_parallel and world > 1:
            kl = x.shape[-1] // world
            x = x[..., rank * kl:(rank + 1) * kl]
        y = self.shard(x.contiguous())
        if world > 1:
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y


__all__ = ["shard_columns", "shard_rows", "ColumnParallelQuantLinear", "RowParallelQuantLinear"]
