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
                              are summed with ONE all_reduce.  Act-order GPTQ layers (uniform groups): rank r owns the
                              rows of GROUPS [r*G/P, (r+1)*G/P) -- the group-sorted arrangement the single-GPU path
                              already uses (quant_linear_gptq.py: native copy of the row-sorted integers) -- and the
                              shard records which input channels those are (`input_index`): a replicated input is
                              gathered by it; a Megatron pair shards the PRODUCER by the same index
                              (shard_columns(..., columns=consumer.input_index)), so its local output already is the
                              consumer's local input and the pair still costs one all_reduce.

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


def _packed_columns(qzeros: torch.Tensor, bits: int, n: int, cols: torch.Tensor) -> torch.Tensor:
    """Columns `cols` of a GPTQ qzeros [G, n*bits/32] (bit stream along N), re-packed."""
    from .modeling.q_layers.compress_weight import pack_bitstream, unpack_bitstream
    z = unpack_bitstream(qzeros, bits, n, axis=1)
    return pack_bitstream(z.index_select(1, cols.to(z.device)), bits, axis=1)


def shard_columns(layer: nn.Module, rank: int, world: int, columns: Optional[torch.Tensor] = None) -> nn.Module:
    """New q_layer of the same class holding output columns [rank*N/world, (rank+1)*N/world) of `layer` -- or, with `columns`
    (an index tensor; GPTQ / HQQ layers), exactly those output columns in that order: how the producer of a Megatron pair is
    sharded when its consumer is an act-order row-parallel layer (columns = consumer_shard.input_index)."""
    n, k, bits, g = layer.outfeatures, layer.infeatures, layer.bits, layer.groupsize
    cls = type(layer)
    if columns is not None:
        cols = columns.to(torch.long).reshape(-1)
        nl = int(cols.numel())
        if isinstance(layer, WQLinear_GEMM) or not isinstance(layer, (QuantLinearGPTQ, QuantLinearHQQ)):
            raise TypeError("explicit column sets: GPTQ / HQQ layers (AWQ words interleave 8 columns)")
        if (nl * bits) % 32 != 0:
            raise ValueError(f"a shard of {nl} columns at {bits} bits does not fill whole packed-zero words")
        new = cls(bits, g, k, nl, layer.bias is not None, dtype=layer.scales.dtype)
        dev = layer.qweight.device
        new.qweight = layer.qweight.index_select(1, cols.to(dev)).contiguous()
        if isinstance(layer, QuantLinearHQQ):
            new.qzeros = layer.qzeros.index_select(1, cols.to(dev)).contiguous()
        else:
            new.qzeros = _packed_columns(layer.qzeros, bits, n, cols)
            new.g_idx = layer.g_idx.clone()
            new.act_order = layer.act_order
        new.scales = layer.scales.index_select(1, cols.to(dev)).contiguous()
        if layer.bias is not None:
            new.bias = layer.bias.index_select(0, cols.to(dev)).contiguous()
        return new
    if n % world != 0:
        raise ValueError(f"out_features {n} not divisible by world size {world}")
    nl = n // world
    c0, c1 = rank * nl, (rank + 1) * nl
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
    """New q_layer holding 1/world of the input rows (whole groups).  bias stays on rank 0.

    Trivial g_idx: rows [rank*K/world, (rank+1)*K/world).  Act-order (GPTQ, every group exactly `groupsize` rows -- what GPTQ's
    act-order produces, reference gptq.py:229-237): the rows of groups [rank*G/world, (rank+1)*G/world) in group-sorted order
    (perm = stable argsort(g_idx)); the shard is a plain contiguous-group layer and carries `input_index` = perm[k0:k1], the input
    channels it consumes, in its row order."""
    n, k, bits, g = layer.outfeatures, layer.infeatures, layer.bits, layer.groupsize
    if k % world != 0 or (k // world) % g != 0 or ((k // world) * bits) % 32 != 0:
        raise ValueError(f"in_features {k} / {world} must be whole groups of {g} and whole packed words")
    kl = k // world
    k0, k1 = rank * kl, (rank + 1) * kl
    cls = type(layer)
    has_bias = layer.bias is not None and rank == 0
    gi = layer.g_idx.to(torch.int64).cpu() if getattr(layer, "g_idx", None) is not None else torch.arange(k) // g
    trivial = torch.equal(gi, torch.arange(k) // g)
    new = cls(bits, g, kl, n, has_bias, dtype=layer.scales.dtype)
    if trivial:
        if isinstance(layer, WQLinear_GEMM):
            new.qweight = layer.qweight[k0:k1].contiguous()
        else:
            new.qweight = layer.qweight[k0 * bits // 32:k1 * bits // 32].contiguous()
    else:
        if not isinstance(layer, QuantLinearGPTQ):
            raise ValueError("a non-trivial g_idx on a layer that has no act-order")
        groups = k // g
        counts = torch.bincount(gi, minlength=groups)
        if counts.numel() != groups or not bool((counts == g).all()):
            raise ValueError("row-parallel sharding of an act-order layer needs uniform groups (every group exactly groupsize rows)")
        from .modeling.q_layers.compress_weight import pack_bitstream, unpack_bitstream
        sel = torch.argsort(gi, stable=True)[k0:k1]
        dev = layer.qweight.device
        q = unpack_bitstream(layer.qweight, bits, k, axis=0)
        new.qweight = pack_bitstream(q.index_select(0, sel.to(dev)), bits, axis=0)
        new.act_order = False   # the shard's own g_idx is the trivial default of its constructor
        new.register_buffer("input_index", sel.to(torch.int32).to(dev), persistent=False)
    new.qzeros = layer.qzeros[k0 // g:k1 // g].contiguous()
    new.scales = layer.scales[k0 // g:k1 // g].contiguous()
    if has_bias:
        new.bias = layer.bias.clone()
    return new


class ColumnParallelQuantLinear(nn.Module):
    """`static_output=True`: the gathered output lives in a buffer the module owns and re-uses (the contract of a hipGraph's
    static outputs: valid until the next call) -- the forward then allocates nothing at all."""

    def __init__(self, shard: nn.Module, out_features: int, group=None, gather_output: bool = True,
                 collective: str = "all_gather", static_output: bool = False):
        super().__init__()
        assert collective in ("all_gather", "all_reduce")
        self.shard = shard
        self.out_features = out_features
        self.group = group
        self.gather_output = gather_output
        self.collective = collective
        self.static_output = static_output
        self._bufs: dict = {}

    @classmethod
    def from_full(cls, layer: nn.Module, group=None, gather_output: bool = True, collective: str = "all_gather",
                  static_output: bool = False):
        rank, world = _world(group)
        return cls(shard_columns(layer, rank, world), layer.outfeatures, group, gather_output, collective, static_output)

    def _buffer(self, tag, shape, dtype, device, zero=False):
        if not self.static_output:
            return (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        key = (tag, tuple(shape), dtype, device)
        buf = self._bufs.get(key)
        if buf is None:
            buf = self._bufs[key] = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
        return buf

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rank, world = _world(self.group)
        if not self.gather_output or world == 1:
            return self.shard(x)  # [..., N/P] through the fused kernel
        nl = self.shard.outfeatures
        lead = tuple(x.shape[:-1])
        m = 1
        for d in lead:
            m *= int(d)
        full = self._buffer("full", (m, world * nl), x.dtype, x.device, zero=(self.collective == "all_reduce" and m > 1))
        if m == 1:
            # decode: the rank's slice of the [1, N] output is contiguous -- the shard kernel writes it in place and the
            # collective runs in place on `full` (all_gather: input = the slice at offset rank * nl of the output; all_reduce:
            # the other slices are zero).  One output tensor (none with static_output), no temporaries, no permute.
            if self.collective == "all_reduce":
                full.zero_()
            self.shard.forward_into(x, full[:, rank * nl:(rank + 1) * nl])
            if self.collective == "all_gather":
                dist.all_gather_into_tensor(full.view(world * nl), full.view(world * nl)[rank * nl:(rank + 1) * nl], group=self.group)
            else:
                dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)  # disjoint slices: exact
            return full.view(lead + (world * nl,))
        if self.collective == "all_gather":
            # rank-major [P, M, N/P] gather buffer; the shard writes its own [M, N/P] block in place, one strided copy
            # re-arranges to [M, P * N/P]
            parts = self._buffer("parts", (world, m, nl), x.dtype, x.device)
            self.shard.forward_into(x, parts[rank])
            dist.all_gather_into_tensor(parts.view(world * m, nl), parts[rank], group=self.group)
            full.view(m, world, nl).copy_(parts.permute(1, 0, 2))
            return full.view(lead + (world * nl,))
        y = self._buffer("y", (m, nl), x.dtype, x.device)
        self.shard.forward_into(x, y)
        if self.static_output:
            full.zero_()
        full[:, rank * nl:(rank + 1) * nl] = y
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)  # disjoint slices: exact
        return full.view(lead + (world * nl,))


def _gather_last(x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """x[..., idx]: the library's column-gather kernel on a HIP device (idx int32), index_select elsewhere."""
    if x.is_cuda and x.shape[-1] % 8 == 0 and idx.numel() % 8 == 0 and idx.numel() == x.shape[-1]:
        from . import ops
        return ops.gather_columns(x.reshape(-1, x.shape[-1]).contiguous(), idx).reshape(x.shape[:-1] + (idx.numel(),))
    return x.index_select(-1, idx.to(device=x.device, dtype=torch.long))


class RowParallelQuantLinear(nn.Module):
    """`reducer`: an object with `all_reduce(tensor)` (qllm_amd.comm.OneShotAllReduce: decode-sized sums through the one-shot
    peer-write kernel, larger ones through RCCL); None = `dist.all_reduce`."""

    def __init__(self, shard: nn.Module, group=None, input_is_parallel: bool = True, static_output: bool = False, reducer=None,
                 fuse_reduce: bool = True):
        super().__init__()
        self.shard = shard
        self.group = group
        self.input_is_parallel = input_is_parallel
        self.static_output = static_output
        self.reducer = reducer
        self.fuse_reduce = fuse_reduce   # batch 1 + one-shot reducer: GEMV and all-reduce in one launch (False: two launches)
        self._fuse_refused: set = set()  # activation dtypes for which the library refused the fused launch (shape / layout: final)
        self._bufs: dict = {}

    @classmethod
    def from_full(cls, layer: nn.Module, group=None, input_is_parallel: bool = True, static_output: bool = False, reducer=None,
                  fuse_reduce: bool = True):
        rank, world = _world(group)
        return cls(shard_rows(layer, rank, world), group, input_is_parallel, static_output, reducer, fuse_reduce)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rank, world = _world(self.group)
        idx = getattr(self.shard, "input_index", None)
        if not self.input_is_parallel and (world > 1 or idx is not None):
            if idx is not None:   # act-order shard: the input channels of its groups (one gather, as on a single GPU)
                x = _gather_last(x, idx)
            else:
                kl = x.shape[-1] // world
                x = x[..., rank * kl:(rank + 1) * kl]
        # (input_is_parallel with an act-order shard: the producer was sharded with columns=shard.input_index)
        x = x.contiguous()
        fuse = (world > 1 and self.fuse_reduce and self.reducer is not None and x.is_cuda and x.numel() == x.shape[-1]
                and x.dtype not in self._fuse_refused and hasattr(self.shard, "forward_allreduce_into"))
        if self.static_output or fuse:
            lead = tuple(x.shape[:-1])
            if self.static_output:
                key = (lead, x.dtype, x.device)
                y = self._bufs.get(key)
                if y is None:
                    y = self._bufs[key] = torch.empty(lead + (self.shard.outfeatures,), dtype=x.dtype, device=x.device)
            else:
                y = torch.empty(lead + (self.shard.outfeatures,), dtype=x.dtype, device=x.device)
            # batch 1 with a one-shot reducer: the shard's launch pushes its partial outputs to the peers itself and its last block
            # sums -- ONE launch instead of GEMV + all-reduce (csrc/strip1_kernel.hpp, AR); bit-identical to the two-step path
            if fuse:
                if self.shard.forward_allreduce_into(x, y.view(-1, self.shard.outfeatures), self.reducer):
                    return y
                # refused (3 bits, g != 128, K beyond the batch-1 forms ...): a property of the layer, not of the call -- do not pay
                # for a descriptor, a library call and an error string on every later token
                self._fuse_refused.add(x.dtype)
            self.shard.forward_into(x, y.view(-1, self.shard.outfeatures))
        else:
            y = self.shard(x)
        if world > 1:   # partial products over K: ONE collective, in place
            if self.reducer is not None:
                self.reducer.all_reduce(y)
            else:
                dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        return y


__all__ = ["shard_columns", "shard_rows", "ColumnParallelQuantLinear", "RowParallelQuantLinear"]
