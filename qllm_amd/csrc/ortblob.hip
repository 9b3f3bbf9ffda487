// ORT / MatMulNBits blob layout: dequantise to W[N, K] fp16.
//
//   qweight u8 [N, K/g, g/2]  = N rows of K/2 bytes: byte b of row n holds q[2b, n] (low nibble) and q[2b+1, n]
//   scales  f16 [N, K/g], qzeros u8 [N, ceil(G/2)] (two 4-bit zero points per byte, low nibble first) or f16 [N, K/g]
//   reorder_idx (g_idx) i32 [K]: block of input channel k, when the checkpoint was quantised with act-order
//
// Numerics are the reference's Python path (quant_linear_onnxruntime.py:52-82), which is also what its goldens pin:
//   integer zeros: W = fp16((q - z) * s)         -- integer difference, ONE fp16 rounding
//   fp16 zeros   : W = fp16(fp16(q - z) * s)     -- the difference is rounded to fp16 first
// Both products are exact in fp32 (<= 5 + 11 significant bits), so one v_cvt_f16_f32 gives the correctly rounded value.
//
// Replaces Dequantize4Bits<half, ZeroT> (/root/reference/csrc/ort_cuda/dq.cu:79-245, ort_ops.cc:161-197).
// One thread = 8 bytes of a row (16 consecutive k): 8-byte load, two 16-byte stores; a wave covers 512 B / 2 KB
// contiguous.  HBM-bound: 0.5 B in + 2 B out per weight.
#include "kernels.hpp"

namespace qllm {

struct OrtDequantParams {
  const uint8_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const int32_t *g_idx;
  half_t *out;
  int K, N, block, n_blocks, zrow_bytes, zeros_f16;
};

template <bool REORDER>
__global__ __launch_bounds__(256) void ort_dequant_kernel(const OrtDequantParams p) {
  const int chunks = p.K / 16;  // 16-k chunks per row
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)p.N * chunks) return;
  const int n = (int)(t / chunks), c = (int)(t - (size_t)n * chunks), k0 = 16 * c;
  const uint2_t raw = *(const uint2_t *)(p.qweight + (size_t)n * (p.K / 2) + 8 * c);
  const half_t *srow = p.scales + (size_t)n * p.n_blocks;
  auto zero_of = [&](int j) -> float {
    if (p.zeros_f16) return (float)((const half_t *)p.qzeros)[(size_t)n * p.n_blocks + j];
    const uint8_t zb = ((const uint8_t *)p.qzeros)[(size_t)n * p.zrow_bytes + (j >> 1)];
    return (float)((zb >> (4 * (j & 1))) & 0xF);
  };
  auto deq = [&](uint32_t q, float z, float s) -> half_t {
    float d = (float)q - z;
    if (p.zeros_f16) d = (float)(half_t)d;  // int - half -> half in the reference
    return (half_t)(d * s);
  };
  half_t w[16];
  if constexpr (!REORDER) {
    const int j = k0 / p.block;  // block sizes are multiples of 16: the chunk lies in one block
    const float z = zero_of(j), s = (float)srow[j];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t word = (i < 8) ? raw.x : raw.y;
      w[i] = deq((word >> (4 * (i & 7))) & 0xFu, z, s);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const uint32_t word = (i < 8) ? raw.x : raw.y;
      const int j = p.g_idx[k0 + i];
      w[i] = deq((word >> (4 * (i & 7))) & 0xFu, zero_of(j), (float)srow[j]);
    }
  }
  half8_t lo = {w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
  half8_t hi = {w[8], w[9], w[10], w[11], w[12], w[13], w[14], w[15]};
  half_t *dst = p.out + (size_t)n * p.K + k0;
  *(half8_t *)dst = lo;
  *(half8_t *)(dst + 8) = hi;
}

int launch_ort_dequant(const void *qweight, const void *scales, const void *qzeros, int zeros_f16, const int32_t *g_idx,
                       int block, int K, int N, void *out, hipStream_t stream) {
  OrtDequantParams p;
  p.qweight = (const uint8_t *)qweight;
  p.scales = (const half_t *)scales;
  p.qzeros = qzeros;
  p.g_idx = g_idx;
  p.out = (half_t *)out;
  p.K = K;
  p.N = N;
  p.block = block;
  p.n_blocks = K / block;
  p.zrow_bytes = (p.n_blocks + 1) / 2;
  p.zeros_f16 = zeros_f16;
  const size_t threads = (size_t)N * (K / 16);
  const int grid = (int)((threads + 255) / 256);
  if (g_idx)
    hipLaunchKernelGGL(ort_dequant_kernel<true>, dim3(grid), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL(ort_dequant_kernel<false>, dim3(grid), dim3(256), 0, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
