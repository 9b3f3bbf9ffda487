// Column gather of an activation matrix: out[m, k] = x[m, perm[k]] (2-byte elements).
//
// The act-order path (reference: quant_linear_gptq.py:38-43, scales[g_idx] per ROW of the weight) is served here from a
// row-sorted copy of the layer (perm = argsort(g_idx), built once at load) plus this gather of x per forward, so that the
// fused kernels see a plain contiguous-group layer.  HBM-bound: M K 2 bytes in, M K 2 bytes out.  As an indexed copy straight
// from global memory it runs at a quarter of that (2-byte reads, 29 us for 2048 x 4096 with the framework's index_select);
// here a block stages whole rows in LDS with 16-byte coalesced loads, gathers from LDS, and stores 16 bytes per lane.  A
// thread keeps the perm entries of its output chunks in registers across the block's rows.
#include "kernels.hpp"

namespace qllm {

constexpr int kGatherThreads = 256;
constexpr int kMaxChunks = 14;  // 8-element output chunks per thread: K <= 256 * 8 * 14 = 28672

template <int CHUNKS>
__global__ __launch_bounds__(kGatherThreads) void gather_columns_kernel(const uint16_t *__restrict__ x, const int32_t *__restrict__ perm,
                                                                        uint16_t *__restrict__ out, int M, int K, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) uint16_t row[];  // [2][K] (double buffer)
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * rows_per_block, m1 = min(M, m0 + rows_per_block);
  const int n_chunks = K >> 3;
  // this thread's output chunks: (part + c * parts) * 256 + tid (gridDim.y = parts > 1 only when M is too small to fill the
  // chip by rows: every part stages the whole row, gathers its share of the columns); their source columns
  const int part = blockIdx.y, parts = gridDim.y;
  int32_t src[CHUNKS][8];
#pragma unroll
  for (int c = 0; c < CHUNKS; ++c) {
    const int ch = (part + c * parts) * kGatherThreads + tid;
    if (ch < n_chunks) {
      const int4 a = *(const int4 *)(perm + ch * 8), b = *(const int4 *)(perm + ch * 8 + 4);
      src[c][0] = a.x; src[c][1] = a.y; src[c][2] = a.z; src[c][3] = a.w;
      src[c][4] = b.x; src[c][5] = b.y; src[c][6] = b.z; src[c][7] = b.w;
    }
  }
  auto stage = [&](int m, int buf) {  // row m -> LDS, 16 bytes per lane
    const uint4_t *g = (const uint4_t *)(x + (size_t)m * K);
    uint4_t *l = (uint4_t *)(row + (size_t)buf * K);
    for (int ch = tid; ch < n_chunks; ch += kGatherThreads) l[ch] = g[ch];
  };
  if (m0 < m1) stage(m0, 0);
  __syncthreads();
  for (int m = m0; m < m1; ++m) {
    const int buf = (m - m0) & 1;
    if (m + 1 < m1) stage(m + 1, buf ^ 1);  // next row's loads fly while this one is gathered
    const uint16_t *r = row + (size_t)buf * K;
    uint4_t *o = (uint4_t *)(out + (size_t)m * K);
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) {
      const int ch = (part + c * parts) * kGatherThreads + tid;
      if (ch < n_chunks) {
        uint4_t v;
        v.x = (uint32_t)r[src[c][0]] | ((uint32_t)r[src[c][1]] << 16);
        v.y = (uint32_t)r[src[c][2]] | ((uint32_t)r[src[c][3]] << 16);
        v.z = (uint32_t)r[src[c][4]] | ((uint32_t)r[src[c][5]] << 16);
        v.w = (uint32_t)r[src[c][6]] | ((uint32_t)r[src[c][7]] << 16);
        o[ch] = v;
      }
    }
    __syncthreads();
  }
}

// Prefill-sized M: ROWS rows per step and block.  Every thread first requests its 16-byte pieces of ALL the step's rows (ROWS * K / 8
// pieces over 512 threads: 8 per thread at K = 4096 -- the memory system sees the whole step at once, where the row-at-a-time kernel
// above exposes one round trip per row), parks them in LDS, and after one barrier gathers its output chunk columns of every row
// from LDS (perm entries of those columns held in registers for the whole launch).  Two blocks per CU overlap each other's request
// and gather phases.  Round 4: 2048 x 4096 6.x us against 10 (profiles/r04_actorder.md).
constexpr int kG2Threads = 512;
template <int ROWS, int CPT>  // CPT: output chunk columns per thread = ceil(K / 8 / 512)
__global__ __launch_bounds__(kG2Threads) void gather_rows_kernel(const uint16_t *__restrict__ x, const int32_t *__restrict__ perm,
                                                                 uint16_t *__restrict__ out, int M, int K) {
  extern __shared__ __attribute__((aligned(16))) uint16_t rows_lds[];  // [ROWS][K]
  const int tid = threadIdx.x;
  const int n_chunks = K >> 3;
  int32_t src[CPT][8];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int ch = c * kG2Threads + tid;
    if (ch < n_chunks) {
      const int4 a = *(const int4 *)(perm + ch * 8), b = *(const int4 *)(perm + ch * 8 + 4);
      src[c][0] = a.x; src[c][1] = a.y; src[c][2] = a.z; src[c][3] = a.w;
      src[c][4] = b.x; src[c][5] = b.y; src[c][6] = b.z; src[c][7] = b.w;
    }
  }
  const int steps = (M + ROWS - 1) / ROWS;
  for (int st = blockIdx.x; st < steps; st += gridDim.x) {
    const int m0 = st * ROWS;
    // request: piece p of the step = row p / n_chunks, chunk p % n_chunks (rows past M re-read the last row; never stored)
    constexpr int kMaxPieces = ROWS * CPT;
    uint4_t v[kMaxPieces];
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int r = i / CPT, ch = (i % CPT) * kG2Threads + tid;
      if (ch < n_chunks) v[i] = *(const uint4_t *)(x + (size_t)min(m0 + r, M - 1) * K + ch * 8);
    }
#pragma unroll
    for (int i = 0; i < kMaxPieces; ++i) {
      const int r = i / CPT, ch = (i % CPT) * kG2Threads + tid;
      if (ch < n_chunks) *(uint4_t *)(rows_lds + (size_t)r * K + ch * 8) = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      const uint16_t *rw = rows_lds + (size_t)r * K;
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        const int ch = c * kG2Threads + tid;
        if (ch < n_chunks && m0 + r < M) {
          uint4_t o;
          o.x = (uint32_t)rw[src[c][0]] | ((uint32_t)rw[src[c][1]] << 16);
          o.y = (uint32_t)rw[src[c][2]] | ((uint32_t)rw[src[c][3]] << 16);
          o.z = (uint32_t)rw[src[c][4]] | ((uint32_t)rw[src[c][5]] << 16);
          o.w = (uint32_t)rw[src[c][6]] | ((uint32_t)rw[src[c][7]] << 16);
          *(uint4_t *)(out + (size_t)(m0 + r) * K + ch * 8) = o;
        }
      }
    }
    __syncthreads();  // the rows are re-used by the next step
  }
}

namespace {
template <int ROWS, int CPT>
int launch_rows(const void *x, const int32_t *perm, void *out, int M, int K, hipStream_t stream) {
  const size_t lds = (size_t)ROWS * K * 2;
  if (lds > 64 * 1024) {
    static DeviceLatch attr_done;
    if (int rc = lds_optin(attr_done, (const void *)gather_rows_kernel<ROWS, CPT>)) return rc;
  }
  const int steps = (M + ROWS - 1) / ROWS;
  const int grid = min(steps, 2 * compute_units());
  hipLaunchKernelGGL((gather_rows_kernel<ROWS, CPT>), dim3(grid), dim3(kG2Threads), lds, stream, (const uint16_t *)x, perm, (uint16_t *)out, M, K);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

template <int CHUNKS>
int launch_b(const void *x, const int32_t *perm, void *out, int M, int K, int parts, hipStream_t stream) {
  // rows per block: enough blocks to cover the chip several times, at least 2 rows each so the double buffer has something to hide
  int rows = 1;
  while (rows < 16 && (M + 2 * rows - 1) / (2 * rows) >= 4 * compute_units()) rows *= 2;
  if (rows == 1 && M >= 2) rows = 2;
  const int grid = (M + rows - 1) / rows;
  if ((size_t)K * 4 > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per device
    static DeviceLatch attr_done;
    if (int rc = lds_optin(attr_done, (const void *)gather_columns_kernel<CHUNKS>)) return rc;
  }
  hipLaunchKernelGGL(gather_columns_kernel<CHUNKS>, dim3(grid, parts), dim3(kGatherThreads), (size_t)K * 4, stream, (const uint16_t *)x, perm,
                     (uint16_t *)out, M, K, rows);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}
}  // namespace

bool gather_columns_ok(int K) { return K % 8 == 0 && K >= 8 && K <= kGatherThreads * 8 * kMaxChunks && (size_t)K * 4 <= 160 * 1024; }

int launch_gather_columns(const void *x, const int32_t *perm, void *out, int M, int K, hipStream_t stream) {
  if (M >= 64 && knob("QLLM_GATHER_ROWS", 1)) {  // prefill sizes: several rows per step (<= 80 KB of LDS: two blocks per CU)
    const int cpt = ((K >> 3) + kG2Threads - 1) / kG2Threads;
    const int want = M / (2 * compute_units());  // rows per step that still leave two steps per CU
    if (cpt == 1) {
      if (want >= 8) return launch_rows<8, 1>(x, perm, out, M, K, stream);
      if (want >= 4) return launch_rows<4, 1>(x, perm, out, M, K, stream);
      return launch_rows<2, 1>(x, perm, out, M, K, stream);
    }
    if (cpt == 2) return want >= 4 ? launch_rows<4, 2>(x, perm, out, M, K, stream) : launch_rows<2, 2>(x, perm, out, M, K, stream);
    if (cpt == 3) return launch_rows<2, 3>(x, perm, out, M, K, stream);   // K <= 12288 (Llama-2-7B down_proj: 11008)
    if (cpt == 4) return launch_rows<2, 4>(x, perm, out, M, K, stream);
  }
  const int groups = ((K >> 3) + kGatherThreads - 1) / kGatherThreads;  // 256-thread passes over a row's chunks
  // decode-sized M: split the columns over up to `groups` blocks per row pair, so that a single row is not one block's job
  int parts = 1;
  while (parts < groups && ((M + 1) / 2) * parts < compute_units() / 4) parts *= 2;
  parts = min(parts, groups);
  const int chunks = (groups + parts - 1) / parts;
  if (chunks <= 1) return launch_b<1>(x, perm, out, M, K, parts, stream);
  if (chunks <= 2) return launch_b<2>(x, perm, out, M, K, parts, stream);
  if (chunks <= 4) return launch_b<4>(x, perm, out, M, K, parts, stream);
  if (chunks <= 6) return launch_b<6>(x, perm, out, M, K, parts, stream);
  if (chunks <= 8) return launch_b<8>(x, perm, out, M, K, parts, stream);
  return launch_b<kMaxChunks>(x, perm, out, M, K, parts, stream);
}

}  // namespace qllm
