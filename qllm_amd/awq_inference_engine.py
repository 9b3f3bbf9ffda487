"""Drop-in for the reference's `qllm.awq_inference_engine` pybind module (csrc/awq_cuda/pybind_awq.cpp:13-20):
`gemm_forward_cuda` with the reference's signature (gemm_cuda.h:3-4), backed by libqllm_mi355x.so."""
from __future__ import annotations

import torch

from . import ops


def gemm_forward_cuda(in_feats: torch.Tensor, kernel: torch.Tensor, scaling_factors: torch.Tensor,
                      zeros: torch.Tensor, split_k_iters: int = 8) -> torch.Tensor:
    """x[M,K] f16, qweight[K,N/8] i32, scales[K/g,N] f16, qzeros[K/g,N/8] i32 -> y[M,N].
    Argument checks follow gemm_cuda_gen.cu:1128-1135 (ValueError, as std::invalid_argument maps to)."""
    m, k = in_feats.shape
    n = kernel.shape[1] * 8
    group_size = k // scaling_factors.shape[0]
    if n % 8 != 0:
        raise ValueError("OC is not multiple of pack_num = 8")
    if group_size % 32 != 0:
        raise ValueError("Group size should be a multiple of 32")
    w, keep = ops.make_weight("GEMM", kernel, scaling_factors, zeros, None, None, k, n, group_size, 4, 0)
    return ops.linear_forward(w, in_feats.contiguous())
