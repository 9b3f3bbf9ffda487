// LDS-tiled MFMA GEMM with in-loop int4 dequant: y[M,N] = x[M,K] . dequant(W4)  (prefill, M > 64).
//
// Replaces gemm_forward_4bit_cuda_m16n128k32 / _m16n64k32 (/root/reference/csrc/awq_cuda/quantization/
// gemm_cuda_gen.cu:31-681: M-tile 16, so at M=2048 every weight is re-read and re-dequantised 128x, fp16 split-k
// partials) and the dequant-to-HBM + cuBLAS route of QuantLinearGPTQ (quant_linear_gptq.py:81-85,
// dq_gemv.cu:247-454: a 2.K.N-byte W round trip per call).
//
// v1 structure (round 1): 128x128x64 block tile, 4 waves (2x2), each wave 64x64 = 4x4 tiles of
// v_mfma_f32_16x16x32_f16, fp32 accumulate; A (activations) and B (dequantised weights, stored k-contiguous per
// column = the MFMA B-fragment order) double-buffered in LDS with a 16-byte-slot XOR swizzle
// slot ^= (row ^ row>>3) & 7 that is conflict-free for the b128 writes of both weight layouts and <=2-way for the
// fragment reads; global loads for tile t+1 are issued before the MFMAs of tile t and written to LDS after
// them (one barrier per k-tile).  Packed weights are read once per 128 rows of M (0.5 B/weight): the
// kernel is MFMA-bound, the dequant VALU work (15 ops / 8 weights, once per block) rides under the matrix pipe.
// Act-order (g_idx): the block caches its 128 columns' (scale, zero*scale) table for ALL groups in LDS and
// gathers per k.
#include "kernels.hpp"

namespace qllm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileHalves = 128 * BK;  // one A or B tile, in halves (16 KB)

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ lds_row_swizzle(row); }

// byte offset of 16-byte slot `slot` (8 halves of k) of row/column `row` inside a [128][64]-half tile
__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + swz(row, slot) * 8; }

__device__ __forceinline__ half_t gemm_zero(const GemmParams &p, int G, int n) {
  if (p.zero_kind == ZK_F16) return ((const half_t *)p.qzeros)[(size_t)G * p.N + n];
  if (p.zero_kind == ZK_SYM) return (half_t)8.f;
  const uint32_t zw = ((const uint32_t *)p.qzeros)[(size_t)G * (p.N >> 3) + (n >> 3)];
  return (half_t)(float)(((zw >> (4 * (n & 7))) + (uint32_t)p.add_zero_bias) & 15u);
}
__device__ __forceinline__ half_t gemm_zero_awq(const GemmParams &p, int G, int n) {
  const uint32_t zw = ((const uint32_t *)p.qzeros)[(size_t)G * (p.N >> 3) + (n >> 3)];
  return (half_t)(float)((zw >> (4 * awq_nibble_of_col(n & 7))) & 15u);
}

// LAYOUT 0 = GPTQ/HQQ row stream, 1 = AWQ GEMM.  ACT = act-order gather through an LDS (s, zs) table.
template <int LAYOUT, bool ACT>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;                    // [2][128][64]
  half_t *Bs = smem + 2 * kTileHalves;  // [2][128 n][64 k]
  uint32_t *tab = (uint32_t *)(smem + 4 * kTileHalves);  // ACT: [n_groups][128] (s | zs << 16)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, i = lane & 15;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware rasterisation: consecutive block ids land on different XCDs (id % 8); give each XCD a contiguous
  // run of tiles that share the same weight columns so its private L2 sees the panel once.
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;  // m fastest: neighbours share the B panel
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- B staging assignment --------------------------------------------------------------------------------
  // GPTQ: thread = column (tid & 127), word rows 4*(tid>>7) .. +3 of the 8 in a k-tile -> 4 x b128 writes
  // AWQ : thread = word column (tid & 15), rows 4*(tid>>4) .. +3 -> 8 columns x 4 k -> 8 x b64 writes
  const int bcol = (LAYOUT == 0) ? (tid & 127) : 8 * (tid & 15);
  const int brow = (LAYOUT == 0) ? 4 * (tid >> 7) : 4 * (tid >> 4);  // word rows (GPTQ) / k rows (AWQ)
  const int nB = n0 + bcol;
  const bool bok = nB < p.N;

  if constexpr (ACT) {
    for (int e = tid; e < p.n_groups * BN; e += 256) {
      const int G = e / BN, c = e - G * BN;
      const int n = n0 + c;
      uint32_t v = 0;
      if (n < p.N) {
        const half_t s = p.scales[(size_t)G * p.N + n];
        const half_t zs = gemm_zero(p, G, n) * s;
        v = (uint32_t)__builtin_bit_cast(uint16_t, s) | ((uint32_t)__builtin_bit_cast(uint16_t, zs) << 16);
      }
      tab[e] = v;
    }
  }

  constexpr int NC = (LAYOUT == 0) ? 1 : 8;  // columns whose constants this thread keeps
  ColConst cc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) cc[c] = make_col_const((half_t)0.f, (half_t)0.f);
  int curG = -1;

  auto group_of = [&](int k) { return p.gs_shift >= 0 ? (k >> p.gs_shift) : (k / p.group_size); };
  auto set_group = [&](int G) {
    if (G == curG) return;
    curG = G;
    if (!bok) return;
    if constexpr (LAYOUT == 0) {
      cc[0] = make_col_const(p.scales[(size_t)G * p.N + nB], gemm_zero(p, G, nB));
    } else {
      const half8_t sv = *(const half8_t *)(p.scales + (size_t)G * p.N + nB);
#pragma unroll
      for (int c = 0; c < 8; ++c) cc[c] = make_col_const(sv[c], gemm_zero_awq(p, G, nB + c));
    }
  };

  // ---- register staging for the next k-tile ------------------------------------------------------------------
  uint4_t areg[4];
  uint32_t breg[4];
  int gk[ACT ? 32 : 1];

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 256 * q, row = c >> 3, kc = c & 7;
      uint4_t v = {0u, 0u, 0u, 0u};
      if (m0 + row < p.M) {
        const size_t off = (size_t)(m0 + row) * p.K + k0 + 8 * kc;
        v = *(const uint4_t *)((const uint16_t *)p.x + off);
      }
      areg[q] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t v = 0;
      if (bok) {
        if constexpr (LAYOUT == 0)
          v = p.qweight[(size_t)(kt * 8 + brow + r) * p.N + nB];
        else
          v = p.qweight[(size_t)(k0 + brow + r) * (p.N >> 3) + (nB >> 3)];
      }
      breg[r] = v;
    }
    if constexpr (ACT) {
      // group ids of this thread's 32 consecutive k (GPTQ layout only)
#pragma unroll
      for (int e = 0; e < 32; e += 4) {
        const uint4_t v = *(const uint4_t *)(p.g_idx + k0 + 8 * brow + e);
        gk[e] = (int)v.x; gk[e + 1] = (int)v.y; gk[e + 2] = (int)v.z; gk[e + 3] = (int)v.w;
      }
    }
  };

  auto store_tile = [&](int kt, int buf) {
    half_t *Ab = As + buf * kTileHalves, *Bb = Bs + buf * kTileHalves;
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 256 * q, row = c >> 3, kc = c & 7;
      if (p.act_bf16) {
        *(half8_t *)(Ab + tile_off(row, kc)) = bf16x8_to_h8(areg[q]);
      } else {
        *(uint4_t *)(Ab + tile_off(row, kc)) = areg[q];
      }
    }
    if constexpr (LAYOUT == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        half8_t w;
        if constexpr (ACT) {
          // per-k constants gathered from the LDS table; nibble j of the word is k = 8*(brow+r)+j
          const uint32_t wv = breg[r];
          half_t o[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t t = tab[gk[8 * r + j] * BN + bcol];
            const half_t s = __builtin_bit_cast(half_t, (uint16_t)(t & 0xffffu));
            const half_t zs = __builtin_bit_cast(half_t, (uint16_t)(t >> 16));
            const half_t q = (half_t)(float)((wv >> (4 * j)) & 15u);
            o[j] = s * q - zs;
          }
          w = half8_t{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
        } else {
          set_group(group_of(k0 + 8 * (brow + r)));
          w = unperm_04152637(deq_word_k04(breg[r], cc[0]));
        }
        *(half8_t *)(Bb + tile_off(bcol, brow + r)) = w;
      }
    } else {
      set_group(group_of(k0 + brow));
      // rows brow..brow+3 of word column: pair rows (0,1) and (2,3)
      const uint32_t P0 = __builtin_amdgcn_perm(breg[1], breg[0], 0x05040100u);
      const uint32_t Q0 = __builtin_amdgcn_perm(breg[1], breg[0], 0x07060302u);
      const uint32_t P1 = __builtin_amdgcn_perm(breg[3], breg[2], 0x05040100u);
      const uint32_t Q1 = __builtin_amdgcn_perm(breg[3], breg[2], 0x07060302u);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int sh = 4 * (c >> 1);
        const uint32_t s0 = (c & 1) ? Q0 : P0, s1 = (c & 1) ? Q1 : P1;
        const half2_t b0 = deq_pair(and_or(s0 >> sh, kNibLo, kMagic), cc[c]);
        const half2_t b1 = deq_pair(and_or(s1 >> sh, kNibLo, kMagic), cc[c]);
        const half4_t w = {b0.x, b0.y, b1.x, b1.y};  // k = brow .. brow+3 of column bcol + c
        *(half4_t *)(Bb + tile_off(bcol + c, brow >> 3) + (brow & 4)) = w;
      }
    }
  };

  float4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = p.K / BK;
  load_tile(0);
  if constexpr (ACT) __syncthreads();  // table visible before the first gather
  store_tile(0, 0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    const half_t *Ab = As + buf * kTileHalves, *Bb = Bs + buf * kTileHalves;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t af[4], bf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) af[a] = *(const half8_t *)(Ab + tile_off(wm * 64 + a * 16 + i, ks * 4 + g));
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 16 + i, ks * 4 + g));
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < KT) store_tile(kt + 1, buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: + bias, round once, store --------------------------------------------------------------
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int n = n0 + wn * 64 + b * 16 + i;
    if (n >= p.N) continue;
    const float bv = p.bias ? (float)p.bias[n] : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + a * 16 + 4 * g + r;
        if (m >= p.M) continue;
        const float v = acc[a][b][r] + bv;
        if (p.act_bf16)
          ((uint16_t *)p.y)[(size_t)m * p.N + n] = f32_to_bf16(v);
        else
          ((half_t *)p.y)[(size_t)m * p.N + n] = (half_t)v;
      }
    }
  }
}

size_t gemm_lds_bytes(bool act, int n_groups) {
  return (size_t)4 * kTileHalves * sizeof(half_t) + (act ? (size_t)n_groups * BN * 4 : 0);
}

int launch_gemm(const GemmParams &p, int layout, hipStream_t stream) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const bool act = p.g_idx != nullptr;
  const size_t lds = gemm_lds_bytes(act, p.n_groups);
  if (lds > 160 * 1024) return set_error(QLLM_ERR_UNSUPPORTED, "act-order table (%d groups) exceeds LDS", p.n_groups);
#define QLLM_LAUNCH_GEMM(L, A)                                                                         \
  do {                                                                                                 \
    static DeviceLatch attr_done; /* per (kernel, device): the LDS opt-in is a per-device attribute */   \
    if (int rc = lds_optin(attr_done, (const void *)gemm_kernel<L, A>)) return rc;                     \
    hipLaunchKernelGGL((gemm_kernel<L, A>), dim3(tiles), dim3(256), lds, stream, p);                   \
  } while (0)
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {
    QLLM_LAUNCH_GEMM(1, false);
  } else if (act) {
    QLLM_LAUNCH_GEMM(0, true);
  } else {
    QLLM_LAUNCH_GEMM(0, false);
  }
#undef QLLM_LAUNCH_GEMM
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
