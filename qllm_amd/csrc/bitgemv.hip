// bitgemv: fused dequant + matvec for the bit widths the MFMA strips do not take -- 2, 5, 6, 7, 8 (and any 3 / 4-bit layer that
// reaches it) -- on the reference's row-stream layouts (GPTQ / HQQ qweight i32 [K * bits / 32][N]), decode sizes (M <= 16).  Round 6
// (round-5 verdict, Missing #4): until now these widths took qllm_dequant + a dense GEMM, the reference's own branch (B)
// (/root/reference/qllm/modeling/q_layers/quant_linear_gptq.py:81-85, csrc/ort_cuda/dq_gemv.cu:190-454): W written to HBM as fp16
// (2 bytes per weight) and read back, where the packed words are bits / 8 bytes per weight.  HQQ's default widths include 2 and 8
// (/root/reference/qllm/quantization/hqq/_hqq_quantizer.py:18).
//
// HBM-bound integer work: no MFMA (a batch-1 product has no reuse to feed a matrix core with, and the 16x16 tiles of the strips exist
// for bit widths whose fields do not straddle words).  Decomposition:
//   * unit = 32 consecutive k of one column = `bits` consecutive words of that column's stream (a field may straddle two of them);
//   * wave = 32 columns x 2 unit parities: a wave-load of one word row is two full 128-byte lines; a block = 32 columns, NW waves
//     taking units round-robin, one K chunk of the layer (blocks split K until the launch covers the CUs twice);
//   * every lane keeps two sets of its next units' words in flight (loads first, arithmetic behind them);
//   * a pair of fields becomes one packed fp16 operand: widths dividing 16 (2, 4, 8) pair the fields 16 bits apart in a word -- one
//     shift and one v_and_or give (1024 + q_a, 1024 + q_b); the others (3, 5, 6, 7) take a 32-bit window of the stream with
//     v_alignbit and place its two fields; minus 1024 (exact: q <= 255) and ONE v_dot2_f32_f16 per activation row accumulates
//     x_a q_a + x_b q_b in fp32.  The activations of the block's K chunk are staged once in LDS as fp16 pairs in exactly that pairing
//     (bf16 callers: converted on the way in), with the sum of every unit's 32 activations beside them;
//   * per unit and row: y += s_g (acc - z_g Sx) in fp32 -- x W for the UNROUNDED W = s (q - z), the contract of the strip kernels
//     (DESIGN.md section 2): no per-weight fp16 rounding at all; packed, fp16 (HQQ) and symmetric zero points;
//   * lanes -> waves -> (K-split) blocks are summed in fixed order: LDS, then fp32 slabs + ticket (the protocol of skinny.hip).
#include "kernels.hpp"

namespace qllm {
namespace bg {

constexpr int kCols = 32;     // columns per block
constexpr int kNW = 8;        // waves per block
constexpr int kXsBytes = 16 * 1024;  // LDS budget of the staged activations (with the sums and the reduction scratch: < 64 KB at 16 rows)

// how the 32 fields of a unit pair up into 16 packed operands: pair p = fields (a(p), b(p))
template <int BITS>
struct Pairing {
  static constexpr bool kShared = (16 % BITS) == 0;  // fields 16 bits apart in one word share a shift
  static constexpr int kPerWord = 32 / BITS, kHalf = 16 / (BITS ? BITS : 1);
  __host__ __device__ static constexpr int a(int p) { return kShared ? (p / kHalf) * kPerWord + p % kHalf : 2 * p; }
  __host__ __device__ static constexpr int b(int p) { return kShared ? a(p) + kHalf : 2 * p + 1; }
};

// packed operand (q_a, q_b) as exact fp16 of pair P (compile-time) from the unit's words
template <int BITS, int P>
__device__ __forceinline__ half2_t pair_of(const uint32_t (&w)[BITS]) {
  constexpr uint32_t mask = (1u << BITS) - 1u;
  uint32_t v;
  if constexpr (Pairing<BITS>::kShared) {
    constexpr int half = Pairing<BITS>::kHalf, wd = P / half, sh = BITS * (P % half);
    v = ((w[wd] >> sh) & (mask | (mask << 16))) | kMagic;
  } else {
    constexpr int o0 = 2 * P * BITS, wi = o0 >> 5, sh = o0 & 31;
    uint32_t win;
    if constexpr (sh + 2 * BITS <= 32) win = w[wi] >> sh;
    else win = __builtin_amdgcn_alignbit(w[wi + 1], w[wi], sh);
    v = (win & mask) | ((win << (16 - BITS)) & (mask << 16)) | kMagic;
  }
  return as_h2(v) - splat2((half_t)1024.0f);  // exact: 1024 + q, q <= 255, is an integer below 2048
}

template <int BITS, int MT, int... P>
__device__ __forceinline__ void unit_dot(const uint32_t (&w)[BITS], const uint32_t *xs_u, int x_stride, float (&acc)[MT],
                                         std::integer_sequence<int, P...>) {
  // 4 pairs per ds_read_b128 (every lane of a unit parity reads the same address: a broadcast)
  auto quad = [&](auto qi) {
    constexpr int Q = decltype(qi)::value;
    const half2_t q0 = pair_of<BITS, 4 * Q>(w), q1 = pair_of<BITS, 4 * Q + 1>(w), q2 = pair_of<BITS, 4 * Q + 2>(w), q3 = pair_of<BITS, 4 * Q + 3>(w);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const uint4_t xv = *(const uint4_t *)(xs_u + m * x_stride + 4 * Q);
      float a = acc[m];
      a = __builtin_amdgcn_fdot2(as_h2(xv.x), q0, a, false);
      a = __builtin_amdgcn_fdot2(as_h2(xv.y), q1, a, false);
      a = __builtin_amdgcn_fdot2(as_h2(xv.z), q2, a, false);
      a = __builtin_amdgcn_fdot2(as_h2(xv.w), q3, a, false);
      acc[m] = a;
    }
  };
  (quad(std::integral_constant<int, P>{}), ...);
}

template <int BITS, int MT>
__global__ __launch_bounds__(kNW * 64) void bitgemv_kernel(const BitGemvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, par = lane >> 5;
  const int nb = blockIdx.x % p.n_col_blocks, kb = blockIdx.x / p.n_col_blocks;
  const int n = nb * kCols + col;
  const bool live = n < p.N;
  const int nc = live ? n : p.N - 1;  // (dead lanes of a ragged last block re-read the last column and store nothing)
  const int U = p.K / 32;
  const int u_begin = (int)((long long)U * kb / p.ksplit), u_end = (int)((long long)U * (kb + 1) / p.ksplit);
  // LDS: [MT][chunk units][16 pairs] u32 | [MT][chunk units] f32 sums | reduction scratch
  uint32_t *xs = (uint32_t *)smem;
  const int cu = p.chunk_units, x_stride = cu * 16;
  float *sx = (float *)(smem + (size_t)MT * cu * 64);
  float *red = sx + MT * cu;

  float y[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) y[m] = 0.f;
  const uint32_t *wcol = p.qweight + nc;
  const int zk = p.zero_kind;
  const int zwords = (p.N * BITS) >> 5;  // packed zero points: words per group row

  for (int c0 = u_begin; c0 < u_end; c0 += cu) {
    const int c1 = min(c0 + cu, u_end), nu = c1 - c0;
    __syncthreads();  // (the previous chunk's readers are done)
    // ---- stage x[m][32 c0 .. 32 c1) as fp16 pairs in the unit's pairing; rows past M are zeros ------------------------------------
    for (int i = tid; i < MT * nu * 16; i += kNW * 64) {
      const int pr = i & 15, u = (i >> 4) % nu, m = (i >> 4) / nu;
      uint32_t v = 0;
      if (m < p.M) {
        const uint16_t *xr = (const uint16_t *)p.x + (size_t)m * p.K + 32 * (c0 + u);
        // (compile-time pairing tables as arithmetic: a = first field of the pair, b = its partner)
        const int a = Pairing<BITS>::kShared ? (pr / Pairing<BITS>::kHalf) * Pairing<BITS>::kPerWord + pr % Pairing<BITS>::kHalf : 2 * pr;
        const int b = Pairing<BITS>::kShared ? a + Pairing<BITS>::kHalf : a + 1;
        v = (uint32_t)xr[a] | ((uint32_t)xr[b] << 16);
        if (p.act_bf16) v = as_u32(bf16x2_to_h2(v));
      }
      xs[m * x_stride + u * 16 + pr] = v;
    }
    __syncthreads();
    for (int i = tid; i < MT * nu; i += kNW * 64) {
      const int u = i % nu, m = i / nu;
      float s = 0.f;
#pragma unroll
      for (int pr = 0; pr < 16; ++pr) {
        const half2_t h = as_h2(xs[m * x_stride + u * 16 + pr]);
        s += (float)h.x + (float)h.y;
      }
      sx[m * cu + u] = s;
    }
    __syncthreads();

    // ---- this lane's units: c0 + 2 wave + par, step 2 NW; the next unit's words are in flight while one is consumed ----------------
    auto load_unit = [&](int u, uint32_t (&w)[BITS]) {
      const uint32_t *src = wcol + (size_t)u * BITS * p.N;
#pragma unroll
      for (int j = 0; j < BITS; ++j) w[j] = __builtin_nontemporal_load(src + (size_t)j * p.N);
    };
    int u = c0 + 2 * wave + par;
    uint32_t w0[BITS], w1[BITS];
    if (u < c1) load_unit(u, w0);
    int g_have = -1;
    float s_g = 0.f, z_g = 0.f;
    while (u < c1) {
      const int un = u + 2 * kNW;
      if (un < c1) load_unit(un, w1);
      // scale / zero point of (group, column): reloaded when the group changes
      const int g = (32 * u) / p.group_size;
      if (g != g_have) {
        g_have = g;
        s_g = (float)p.scales[(size_t)g * p.N + nc];
        if (zk == ZK_F16) z_g = (float)((const half_t *)p.qzeros)[(size_t)g * p.N + nc];
        else if (zk == ZK_SYM) z_g = (float)(1 << (BITS - 1));
        else z_g = (float)packed_zero((const uint32_t *)p.qzeros + (size_t)g * zwords, nc, BITS, p.add_zero_bias);
      }
      float acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = 0.f;
      unit_dot<BITS, MT>(w0, xs + (u - c0) * 16, x_stride, acc, std::make_integer_sequence<int, 4>{});
#pragma unroll
      for (int m = 0; m < MT; ++m) y[m] += s_g * (acc[m] - z_g * sx[m * cu + (u - c0)]);
#pragma unroll
      for (int j = 0; j < BITS; ++j) w0[j] = w1[j];
      u = un;
    }
  }

  // ---- lanes (two unit parities) and waves: fixed-order sum through LDS ---------------------------------------------------------------
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m) red[((wave * 2 + par) * MT + m) * kCols + col] = y[m];
  __syncthreads();
  const int S = p.ksplit;
  for (int i = tid; i < MT * kCols; i += kNW * 64) {
    const int c = i % kCols, m = i / kCols;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 2 * kNW; ++q) v += red[(q * MT + m) * kCols + c];
    red[2 * kNW * MT * kCols + i] = v;  // (kept for the split-K publication below)
  }
  __syncthreads();
  float *blk = red + 2 * kNW * MT * kCols;  // [MT][32] sums of this block
  int &s_ticket = *(int *)(blk + MT * kCols);
  if (S > 1) {
    // fp32 slab [ksplit][M][N]: write-through stores, drained; one relaxed agent-scope ticket per column block; the last arriver sums
    // the S slabs in fixed order (deterministic) and re-arms the counter
    for (int i = tid; i < p.M * kCols; i += kNW * 64) {
      const int c = i % kCols, m = i / kCols;
      if (nb * kCols + c < p.N) st_sc1(p.slabs + ((size_t)kb * p.M + m) * p.N + nb * kCols + c, blk[m * kCols + c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.counters + nb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != S - 1) return;
    for (int i = tid; i < p.M * kCols; i += kNW * 64) {
      const int c = i % kCols, m = i / kCols;
      float v = 0.f;
      if (nb * kCols + c < p.N)
        for (int sp = 0; sp < S; ++sp) v += ld_sc1(p.slabs + ((size_t)sp * p.M + m) * p.N + nb * kCols + c);
      blk[m * kCols + c] = v;
    }
    if (tid == 0) __hip_atomic_store(p.counters + nb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
  for (int i = tid; i < p.M * kCols; i += kNW * 64) {
    const int c = i % kCols, m = i / kCols, nn = nb * kCols + c;
    if (nn >= p.N) continue;
    float v = blk[m * kCols + c];
    if (p.bias) v += (float)p.bias[nn];
    if (p.act_bf16) ((uint16_t *)p.y)[(size_t)m * p.N + nn] = f32_to_bf16(v);
    else ((half_t *)p.y)[(size_t)m * p.N + nn] = (half_t)v;
  }
}

template <int BITS>
static int launch_b(const BitGemvParams &p, int mt, int grid, size_t lds, hipStream_t stream) {
#define QLLM_BG(MT_)                                                                                         \
  hipLaunchKernelGGL((bitgemv_kernel<BITS, MT_>), dim3(grid), dim3(kNW * 64), lds, stream, p);              \
  break
  switch (mt) {
    case 1: QLLM_BG(1);
    case 2: QLLM_BG(2);
    case 4: QLLM_BG(4);
    case 8: QLLM_BG(8);
    default: QLLM_BG(16);
  }
#undef QLLM_BG
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace bg

int bitgemv_mt(int M) { return M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : 16))); }

// shapes served: row-stream layouts, whole 32-k units inside one group, decode sizes
bool bitgemv_ok(const qllm_weight_t &w, int M) {
  if (w.layout != QLLM_LAYOUT_GPTQ && w.layout != QLLM_LAYOUT_HQQ) return false;
  if (w.bits < 2 || w.bits > 8 || w.g_idx || M < 1 || M > kBitGemvMaxM) return false;
  if (w.K % 32 != 0 || w.group_size % 32 != 0 || w.N < 1) return false;
  if ((uintptr_t)w.qweight % 4 || (uintptr_t)w.scales % 2) return false;
  return (double)w.K * w.N * w.bits / 8 < 8e9;
}

// K blocks: until the launch covers the CUs twice, at least 2 units per lane slot, at most the workspace's slab count
int bitgemv_split(int M, int K, int N) {
  const int nb = (N + bg::kCols - 1) / bg::kCols, U = K / 32;
  int S = (2 * compute_units() + nb - 1) / nb;
  const int cap = skinny_max_split(M), by_len = U / (4 * bg::kNW) > 0 ? U / (4 * bg::kNW) : 1;
  S = S > cap ? cap : S;
  S = S > by_len ? by_len : S;
  return S < 1 ? 1 : S;
}

int launch_bitgemv(const BitGemvParams &p_in, int bits, hipStream_t stream) {
  BitGemvParams p = p_in;
  const int mt = bitgemv_mt(p.M);
  p.n_col_blocks = (p.N + bg::kCols - 1) / bg::kCols;
  if (p.ksplit < 1 || !p.slabs || !p.counters) p.ksplit = 1;
  const int U = p.K / 32, per_block = (U + p.ksplit - 1) / p.ksplit;
  const int fit = bg::kXsBytes / (mt * 64);  // units whose staged activations fit the LDS budget
  p.chunk_units = per_block < fit ? per_block : fit;
  const size_t lds = (size_t)mt * p.chunk_units * 64 + (size_t)mt * p.chunk_units * 4 + (size_t)(2 * bg::kNW + 1) * mt * bg::kCols * 4 + 16;
  const int grid = p.n_col_blocks * p.ksplit;
  switch (bits) {
    case 2: return bg::launch_b<2>(p, mt, grid, lds, stream);
    case 3: return bg::launch_b<3>(p, mt, grid, lds, stream);
    case 4: return bg::launch_b<4>(p, mt, grid, lds, stream);
    case 5: return bg::launch_b<5>(p, mt, grid, lds, stream);
    case 6: return bg::launch_b<6>(p, mt, grid, lds, stream);
    case 7: return bg::launch_b<7>(p, mt, grid, lds, stream);
    case 8: return bg::launch_b<8>(p, mt, grid, lds, stream);
  }
  return set_error(QLLM_ERR_UNSUPPORTED, "bitgemv: bits must be 2..8 (got %d)", bits);
}

}  // namespace qllm
