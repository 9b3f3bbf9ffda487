// Dequant-only and layout-conversion kernels (HBM-bound byte/integer work: coalescing is the whole game).
//
//   qllm_dequant          replaces ort_ops.dequant -> DequantizeAndUnpackWeight{248,248_g,357_g,3567_v2}
//                         (/root/reference/csrc/ort_cuda/dq_gemv.cu:190-454, launcher :696-725) and the torch
//                         paths DequantizeLinearBlockWise (quant_linear_gptq.py:13-52), DequantAndUnpack
//                         (quant_linear_hqq.py:8-28), CompressWeight.unpack (compress_weight.py:136-151).
//   qllm_(un)pack_qweight replace general_(un)pack_on_row (+ AWQ reorder) (compress_weight.py:46-92,
//                         quant_linear_awq.py:95-140).
//
// Numerics: W = fp16(fp16(s*q) - fp16(z*s)), one rounding per op (this TU is compiled with -ffp-contract=off),
// bit-identical to the reference's CPU tensors.
#include "kernels.hpp"

namespace qllm {

struct DequantParams {
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const int32_t *g_idx;
  void *out;
  int K, N, group_size, bits, add_zero_bias, zero_kind, out_bf16, out_transposed;
};

__device__ __forceinline__ void store_w(const DequantParams &p, int k, int n, half_t w) {
  const size_t idx = p.out_transposed ? (size_t)n * p.K + k : (size_t)k * p.N + n;
  if (p.out_bf16)
    ((uint16_t *)p.out)[idx] = f32_to_bf16((float)w);
  else
    ((half_t *)p.out)[idx] = w;
}

__device__ __forceinline__ half_t zero_of(const DequantParams &p, int grp, int n) {
  if (p.zero_kind == ZK_F16) return ((const half_t *)p.qzeros)[(size_t)grp * p.N + n];
  if (p.zero_kind == ZK_SYM) return (half_t)(float)(1 << (p.bits - 1));
  const int zwords = (p.N * p.bits + 31) / 32;
  return (half_t)(float)packed_zero((const uint32_t *)p.qzeros + (size_t)grp * zwords, n, p.bits, p.add_zero_bias);
}

// GPTQ / HQQ row-stream layout, any bits: one thread = one column x 32 consecutive k (= `bits` words).
// Lanes run along N: every load and every store of a wave is one contiguous segment.
__global__ __launch_bounds__(256) void dequant_rows_kernel(DequantParams p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int kblk = blockIdx.y;  // 32 k per block row
  if (n >= p.N) return;
  const int bits = p.bits;
  const uint32_t mask = (1u << bits) - 1u;
  const int k0 = kblk * 32;
  uint32_t words[9];
  const int total_words = (p.K * bits) / 32;  // rows actually present (reference: K//32*bits)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int wr = kblk * bits + i;
    words[i] = (i < bits && wr < total_words) ? p.qweight[(size_t)wr * p.N + n] : 0u;
  }
  words[8] = 0u;
  int cur_g = -1;
  half_t s = (half_t)0.f, zs = (half_t)0.f;
  for (int i = 0; i < 32; ++i) {
    const int k = k0 + i;
    if (k >= p.K) break;
    const int bit0 = i * bits;
    const int w = bit0 >> 5, off = bit0 & 31;
    // static-index-free extraction: select the two candidate words with a small unrolled scan
    uint32_t lo = 0u, hi = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lo = (j == w) ? words[j] : lo;
      hi = (j == w) ? words[j + 1] : hi;
    }
    const uint64_t both = (uint64_t)lo | ((uint64_t)hi << 32);
    const uint32_t q = (uint32_t)(both >> off) & mask;
    const int grp = p.g_idx ? p.g_idx[k] : k / p.group_size;
    if (grp != cur_g) {
      cur_g = grp;
      s = p.scales[(size_t)grp * p.N + n];
      zs = zero_of(p, grp, n) * s;  // one rounding
    }
    const half_t sq = s * (half_t)(float)q;  // one rounding (q <= 255 exact in fp16)
    store_w(p, k, n, sq - zs);               // one rounding
  }
}

// AWQ GEMM layout (4-bit): one thread = one word = row k, 8 interleaved columns -> one 16-byte store.
__global__ __launch_bounds__(256) void dequant_awq_kernel(DequantParams p) {
  const int nw = p.N / 8;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= nw) return;
  const uint32_t w = p.qweight[(size_t)k * nw + j];
  const int grp = k / p.group_size;
  const uint32_t zw = ((const uint32_t *)p.qzeros)[(size_t)grp * nw + j];
  half_t out[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int pos = awq_nibble_of_col(c);
    const uint32_t q = (w >> (4 * pos)) & 0xFu;
    const uint32_t z = (zw >> (4 * pos)) & 0xFu;
    const half_t s = p.scales[(size_t)grp * p.N + 8 * j + c];
    const half_t zs = (half_t)(float)z * s;
    const half_t sq = s * (half_t)(float)q;
    out[c] = sq - zs;
  }
  if (!p.out_transposed && !p.out_bf16) {
    half8_t v = {out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7]};
    *(half8_t *)((half_t *)p.out + (size_t)k * p.N + 8 * j) = v;
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) store_w(p, k, 8 * j + c, out[c]);
  }
}

// ---- integer grid <-> packed -----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_rows_kernel(const uint32_t *qweight, int32_t *q_kn, int K, int N, int bits) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int kblk = blockIdx.y;
  if (n >= N) return;
  const uint32_t mask = (1u << bits) - 1u;
  const int total_words = (K * bits) / 32;
  for (int i = 0; i < 32; ++i) {
    const int k = kblk * 32 + i;
    if (k >= K) break;
    const int bit0 = i * bits;
    const int w = kblk * bits + (bit0 >> 5), off = bit0 & 31;
    uint64_t v = (w < total_words) ? qweight[(size_t)w * N + n] : 0u;
    if (off + bits > 32 && w + 1 < total_words) v |= (uint64_t)qweight[(size_t)(w + 1) * N + n] << 32;
    q_kn[(size_t)k * N + n] = (int32_t)((uint32_t)(v >> off) & mask);
  }
}

__global__ __launch_bounds__(256) void pack_rows_kernel(const int32_t *q_kn, uint32_t *qweight, int K, int N, int bits) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int kblk = blockIdx.y;  // 32 k -> `bits` words
  if (n >= N) return;
  const uint32_t mask = (1u << bits) - 1u;
  uint64_t acc = 0;  // bit accumulator
  int have = 0, wout = kblk * bits;
  for (int i = 0; i < 32; ++i) {
    const int k = kblk * 32 + i;
    const uint32_t q = (k < K) ? ((uint32_t)q_kn[(size_t)k * N + n] & mask) : 0u;
    acc |= (uint64_t)q << have;
    have += bits;
    if (have >= 32) {
      qweight[(size_t)wout * N + n] = (uint32_t)acc;
      ++wout;
      acc >>= 32;
      have -= 32;
    }
  }
}

__global__ __launch_bounds__(256) void unpack_awq_kernel(const uint32_t *qweight, int32_t *q_kn, int K, int N) {
  const int nw = N / 8;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= nw) return;
  const uint32_t w = qweight[(size_t)k * nw + j];
#pragma unroll
  for (int c = 0; c < 8; ++c) q_kn[(size_t)k * N + 8 * j + c] = (int32_t)((w >> (4 * awq_nibble_of_col(c))) & 0xFu);
}

__global__ __launch_bounds__(256) void pack_awq_kernel(const int32_t *q_kn, uint32_t *qweight, int K, int N) {
  const int nw = N / 8;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (j >= nw) return;
  uint32_t w = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) w |= ((uint32_t)q_kn[(size_t)k * N + 8 * j + c] & 0xFu) << (4 * awq_nibble_of_col(c));
  qweight[(size_t)k * nw + j] = w;
}

// ---- host launchers (called from capi.hip after validation) ------------------------------------------------
int launch_dequant(const qllm_weight_t &w, int zero_kind, void *out, int out_dtype, int out_transposed,
                   hipStream_t stream) {
  DequantParams p;
  p.qweight = (const uint32_t *)w.qweight;
  p.scales = (const half_t *)w.scales;
  p.qzeros = w.qzeros;
  p.g_idx = w.g_idx;
  p.out = out;
  p.K = w.K;
  p.N = w.N;
  p.group_size = w.group_size;
  p.bits = w.bits;
  p.add_zero_bias = w.add_zero_bias;
  p.zero_kind = zero_kind;
  p.out_bf16 = (out_dtype == QLLM_BF16);
  p.out_transposed = out_transposed;
  if (w.layout == QLLM_LAYOUT_AWQ_GEMM) {
    dim3 grid((w.N / 8 + 255) / 256, w.K);
    hipLaunchKernelGGL(dequant_awq_kernel, grid, dim3(256), 0, stream, p);
  } else {
    dim3 grid((w.N + 255) / 256, (w.K + 31) / 32);
    hipLaunchKernelGGL(dequant_rows_kernel, grid, dim3(256), 0, stream, p);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_unpack_qweight(const void *qweight, int layout, int bits, int K, int N, int32_t *q_kn, hipStream_t stream) {
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {
    dim3 grid((N / 8 + 255) / 256, K);
    hipLaunchKernelGGL(unpack_awq_kernel, grid, dim3(256), 0, stream, (const uint32_t *)qweight, q_kn, K, N);
  } else {
    dim3 grid((N + 255) / 256, (K + 31) / 32);
    hipLaunchKernelGGL(unpack_rows_kernel, grid, dim3(256), 0, stream, (const uint32_t *)qweight, q_kn, K, N, bits);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_pack_qweight(const int32_t *q_kn, int layout, int bits, int K, int N, void *qweight, hipStream_t stream) {
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {
    dim3 grid((N / 8 + 255) / 256, K);
    hipLaunchKernelGGL(pack_awq_kernel, grid, dim3(256), 0, stream, q_kn, (uint32_t *)qweight, K, N);
  } else {
    dim3 grid((N + 255) / 256, (K + 31) / 32);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, stream, q_kn, (uint32_t *)qweight, K, N, bits);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
