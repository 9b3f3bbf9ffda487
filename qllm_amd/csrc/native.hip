// Reference state-dict layouts <-> the library's strip-major NATIVE layout (include/qllm_mi355x.h, QLLM_LAYOUT_NATIVE*).
//
// The reference stores a quantized linear for ITS kernels: GPTQ / HQQ as [K*bits/32][N] words (a 16-column strip = K/8
// separate 64-byte segments), AWQ as [K][N/8] words (a row is only N/2 bytes).  A batch-1 matvec on MI355X wants every
// workgroup to stream ONE contiguous region: the native layout stores each 16-column strip of a layer -- its packed words, its
// scales and its zero points -- contiguously:
//     qweight i32 [N/16][K*bits/32][16]    word (s, r, i) = row-stream word (r, 16 s + i): 8 (4-bit) consecutive-k values of column 16s+i
//     scales  f16 [N/16][G][16]            G = ceil(K / group_size)
//     qzeros  i32 [N/16][G][2]             4 bits: nibble e of word j = zero point of column 16 s + 8 j + e (stored value, no offset)
//                                          3 bits: the 64-bit little-endian pair holds column 16 s + i at bit 3 i
//             f16 [N/16][G][16]            (QLLM_LAYOUT_NATIVE_F16Z: HQQ's fp16 zero points)      or NULL (symmetric)
// Pure integer permutations: bit-exact both ways (tests/test_native_layout_*.py pin repack -> unpack == identity on the
// reference-minted goldens).  Counterpart in the reference: the load-time repacks of its own kernels' formats,
// /root/reference/qllm/modeling/q_layers/quant_linear_awq.py:95-140 (AWQ interleave), compress_weight.py:46-92.
#include "kernels.hpp"

namespace qllm {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t rowstream_nibble(const uint32_t *qw, int layout, int N, int k, int n) {
  // 4-bit value q[k, n] of a reference layout
  if (layout == QLLM_LAYOUT_AWQ_GEMM) return (qw[(size_t)k * (N >> 3) + (n >> 3)] >> (4 * awq_nibble_of_col(n & 7))) & 15u;
  return (qw[(size_t)(k >> 3) * N + n] >> (4 * (k & 7))) & 15u;
}

// ---- reference -> native -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void repack_words_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int layout,
                                                                int R, int N, size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  const int i = (int)(o & 15);
  const size_t sr = o >> 4;
  const int r = (int)(sr % (size_t)R), s = (int)(sr / (size_t)R);
  const int n = 16 * s + i;
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {  // 4 bits: word-row r = k 8r .. 8r+7
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) w |= rowstream_nibble(in, layout, N, 8 * r + j, n) << (4 * j);
    out[o] = w;
  } else {
    out[o] = in[(size_t)r * N + n];
  }
}

__global__ __launch_bounds__(kThreads) void repack_halves_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int G, int N,
                                                                 size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  const int i = (int)(o & 15);
  const size_t sg = o >> 4;
  const int gi = (int)(sg % (size_t)G), s = (int)(sg / (size_t)G);
  out[o] = in[(size_t)gi * N + 16 * s + i];
}

// one thread per (strip, group): the strip's 16 zero points -> two words
__global__ __launch_bounds__(kThreads) void repack_zeros_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int layout,
                                                                int bits, int G, int N, size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  const int gi = (int)(o % (size_t)G), s = (int)(o / (size_t)G);
  uint64_t v = 0;
  if (bits == 4) {
    const uint32_t *row = in + (size_t)gi * (N >> 3);
    for (int i = 0; i < 16; ++i) {
      const int n = 16 * s + i;
      const int nib = (layout == QLLM_LAYOUT_AWQ_GEMM) ? awq_nibble_of_col(n & 7) : (n & 7);
      v |= (uint64_t)((row[n >> 3] >> (4 * nib)) & 15u) << (4 * i);
    }
  } else {  // 3 bits, GPTQ bit stream along N
    const uint32_t *row = in + (size_t)gi * ((N * 3) >> 5);
    for (int i = 0; i < 16; ++i) v |= (uint64_t)packed_zero(row, 16 * s + i, 3, 0) << (3 * i);
  }
  out[2 * o] = (uint32_t)v;
  out[2 * o + 1] = (uint32_t)(v >> 32);
}

// ---- native -> reference -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void unpack_words_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int layout,
                                                                int R, int K, int N, size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {  // out word (k, j): nibble p = q[k, 8 j + col_of_nibble(p)]
    const int j = (int)(o % (size_t)(N >> 3)), k = (int)(o / (size_t)(N >> 3));
    uint32_t w = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int n = 8 * j + awq_col_of_nibble(p);
      const uint32_t src = in[((size_t)(n >> 4) * R + (k >> 3)) * 16 + (n & 15)];
      w |= ((src >> (4 * (k & 7))) & 15u) << (4 * p);
    }
    out[o] = w;
  } else {
    const int n = (int)(o % (size_t)N), r = (int)(o / (size_t)N);
    out[o] = in[((size_t)(n >> 4) * R + r) * 16 + (n & 15)];
  }
}

__global__ __launch_bounds__(kThreads) void unpack_halves_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, int G, int N,
                                                                 size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  const int n = (int)(o % (size_t)N), gi = (int)(o / (size_t)N);
  out[o] = in[((size_t)(n >> 4) * G + gi) * 16 + (n & 15)];
}

// one thread per output word of the reference's packed zero rows
__global__ __launch_bounds__(kThreads) void unpack_zeros_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, int layout,
                                                                int bits, int G, int N, size_t total) {
  const size_t o = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (o >= total) return;
  auto field = [&](int gi, int n) -> uint32_t {  // stored zero point of column n, group gi
    const uint32_t *pair = in + ((size_t)(n >> 4) * G + gi) * 2;
    const uint64_t v = ((uint64_t)pair[1] << 32) | pair[0];
    return (uint32_t)(v >> (bits * (n & 15))) & ((1u << bits) - 1u);
  };
  if (bits == 4) {
    const int wpr = N >> 3;
    const int j = (int)(o % (size_t)wpr), gi = (int)(o / (size_t)wpr);
    uint32_t w = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int n = 8 * j + ((layout == QLLM_LAYOUT_AWQ_GEMM) ? awq_col_of_nibble(p) : p);
      w |= field(gi, n) << (4 * p);
    }
    out[o] = w;
  } else {  // 3-bit stream along N: word j of the row holds bits [32 j, 32 j + 32)
    const int wpr = (N * 3) >> 5;
    const int j = (int)(o % (size_t)wpr), gi = (int)(o / (size_t)wpr);
    const int n0 = (32 * j) / 3;  // first column with a bit in this word (its field may start in the previous word)
    uint64_t acc = 0;             // bits [3 n0, 3 n0 + 36) of the stream
    for (int c = 0; c < 12; ++c)
      if (n0 + c < N) acc |= (uint64_t)field(gi, n0 + c) << (3 * c);
    out[o] = (uint32_t)(acc >> (32 * j - 3 * n0));
  }
}

int grid_for(size_t total) { return (int)((total + kThreads - 1) / kThreads); }

}  // namespace

int launch_repack_native(const qllm_weight_t &src, int zero_kind, void *qweight_out, void *scales_out, void *qzeros_out, hipStream_t stream) {
  const int K = src.K, N = src.N, bits = src.bits;
  const int R = K * bits / 32, G = (K + src.group_size - 1) / src.group_size;
  const size_t nw = (size_t)R * N;
  hipLaunchKernelGGL(repack_words_kernel, dim3(grid_for(nw)), dim3(kThreads), 0, stream, (const uint32_t *)src.qweight, (uint32_t *)qweight_out,
                     src.layout, R, N, nw);
  const size_t ns = (size_t)G * N;
  hipLaunchKernelGGL(repack_halves_kernel, dim3(grid_for(ns)), dim3(kThreads), 0, stream, (const uint16_t *)src.scales, (uint16_t *)scales_out, G, N, ns);
  if (zero_kind == ZK_F16) {
    hipLaunchKernelGGL(repack_halves_kernel, dim3(grid_for(ns)), dim3(kThreads), 0, stream, (const uint16_t *)src.qzeros, (uint16_t *)qzeros_out, G, N, ns);
  } else if (zero_kind == ZK_PACKED) {
    const size_t nz = (size_t)G * (N / 16);
    hipLaunchKernelGGL(repack_zeros_kernel, dim3(grid_for(nz)), dim3(kThreads), 0, stream, (const uint32_t *)src.qzeros, (uint32_t *)qzeros_out,
                       src.layout, bits, G, N, nz);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_unpack_native(const qllm_weight_t &src, int dst_layout, void *qweight_out, void *scales_out, void *qzeros_out, hipStream_t stream) {
  const int K = src.K, N = src.N, bits = src.bits;
  const int R = K * bits / 32, G = (K + src.group_size - 1) / src.group_size;
  const size_t nw = (dst_layout == QLLM_LAYOUT_AWQ_GEMM) ? (size_t)K * (N / 8) : (size_t)R * N;
  hipLaunchKernelGGL(unpack_words_kernel, dim3(grid_for(nw)), dim3(kThreads), 0, stream, (const uint32_t *)src.qweight, (uint32_t *)qweight_out,
                     dst_layout, R, K, N, nw);
  const size_t ns = (size_t)G * N;
  hipLaunchKernelGGL(unpack_halves_kernel, dim3(grid_for(ns)), dim3(kThreads), 0, stream, (const uint16_t *)src.scales, (uint16_t *)scales_out, G, N, ns);
  if (src.layout == QLLM_LAYOUT_NATIVE_F16Z) {
    hipLaunchKernelGGL(unpack_halves_kernel, dim3(grid_for(ns)), dim3(kThreads), 0, stream, (const uint16_t *)src.qzeros, (uint16_t *)qzeros_out, G, N, ns);
  } else if (src.qzeros) {
    const size_t nz = (size_t)G * ((size_t)N * bits / 32);
    hipLaunchKernelGGL(unpack_zeros_kernel, dim3(grid_for(nz)), dim3(kThreads), 0, stream, (const uint32_t *)src.qzeros, (uint32_t *)qzeros_out,
                       dst_layout, bits, G, N, nz);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
