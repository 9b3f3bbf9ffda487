// Prefill GEMM v2: y[M,N] = x[M,K] . dequant(W4) with a 256x256x64 block tile (fp16 activations, trivial groups).
//
// Same job as gemm.hip (which stays as the general path: bf16 activations, act-order, ragged N) with the structure
// the MI355X guide measures as the first big step for MFMA GEMMs (cdna_hip_programming.md section 5):
//   * 8 waves (2 M x 4 N), each 128x64 = 8x4 tiles of v_mfma_f32_16x16x32_f16: 12 fragment reads per 32 MFMAs
//     (the 128x128 kernel needs 8 per 16), and each weight is dequantised once per 256 rows of M;
//   * A (activations) goes global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs, no VALU); the
//     16-byte-slot XOR swizzle is applied on the SOURCE address (the DMA destination is lane-linear);
//   * B: packed words -> registers -> bit-exact fp16 dequant -> ds_write into the k-contiguous [n][64 k] image
//     (the MFMA B-fragment order), same swizzle; the dequant of tile t+1 is split in two halves placed after each
//     32-MFMA sub-step of tile t so the VALU work rides under the other waves' MFMAs;
//   * double-buffered LDS (2 x 64 KB), one barrier per k-tile; XCD-aware block rasterisation;
//   * epilogue through wave-private LDS: 16-byte row-contiguous stores instead of 2-byte scattered ones.
#include <stdlib.h>

#include "kernels.hpp"

namespace qllm {

namespace g2 {
constexpr int BM = 256, BK = 64;
constexpr int kTile = 256 * BK;  // halves per A tile (32 KB); the B tile uses BN * BK of the same-size slot

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row ^ (row >> 3)) & 7); }
__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + swz(row, slot) * 8; }  // in halves

__device__ __forceinline__ half_t zero_gptq(const GemmParams &p, int G, int n) {
  if (p.zero_kind == ZK_F16) return ((const half_t *)p.qzeros)[(size_t)G * p.N + n];
  if (p.zero_kind == ZK_SYM) return (half_t)8.f;
  const uint32_t zw = ((const uint32_t *)p.qzeros)[(size_t)G * (p.N >> 3) + (n >> 3)];
  return (half_t)(float)(((zw >> (4 * (n & 7))) + (uint32_t)p.add_zero_bias) & 15u);
}
__device__ __forceinline__ half_t zero_awq(const GemmParams &p, int G, int n) {
  const uint32_t zw = ((const uint32_t *)p.qzeros)[(size_t)G * (p.N >> 3) + (n >> 3)];
  return (half_t)(float)((zw >> (4 * awq_nibble_of_col(n & 7))) & 15u);
}
}  // namespace g2

// LAYOUT 0 = GPTQ/HQQ row stream, 1 = AWQ GEMM.  Requires: fp16 activations, K % 64 == 0, N % 256 == 0,
// group_size % 8 == 0 (GPTQ) / % 4 == 0 (AWQ), no g_idx.
// BN = 256: waves 2 (M) x 4 (N), 128x64 per wave.  BN = 128: waves 4 x 2, 64x64 per wave -- twice the blocks, for
// problems whose 256x256 tiling leaves CUs idle (M=2048 x N=4096 is only 128 such tiles).
template <int LAYOUT, int BN>
__global__ __launch_bounds__(512) void gemm2_kernel(const GemmParams p) {
  using namespace g2;
  constexpr int AM = (BN == 256) ? 8 : 4;        // 16-row MFMA tiles per wave along M
  constexpr int WROWS = AM * 16;                 // rows per wave
  constexpr int WPT = (64 / 8) * BN / 512;       // B words per thread per k-tile: 4 or 2
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;              // [2][256][64]
  half_t *Bs = smem + 2 * kTile;  // [2][256 n][64 k]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int wm = (BN == 256) ? (wave >> 2) : (wave >> 1);
  const int wn = (BN == 256) ? (wave & 3) : (wave & 1);

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // each XCD (block id % 8) walks a contiguous run of tiles, m fastest: neighbours share the weight panel
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid % tiles_m, tn = bid / tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- A: LDS-DMA.  One wave-instruction fills 8 rows x 128 B; lane l -> row l/8, destination slot l%8, source
  //         chunk (l%8) ^ swizzle(row).  Wave w covers rows [32w, 32w+32) in 4 instructions. ---------------------------
  const int arow = lane >> 3, aslot = lane & 7;
  auto load_a = [&](int kt, int buf) {
    half_t *Ab = As + buf * kTile;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = wave * 32 + q * 8 + arow;
      const int grow = min(m0 + row, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
      const int chunk = aslot ^ ((row ^ (row >> 3)) & 7);
      const half_t *src = (const half_t *)p.x + (size_t)grow * p.K + kt * BK + 8 * chunk;
      __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)src, (lds_void_t *)(Ab + (wave * 32 + q * 8) * BK), 16, 0, 0);
    }
  };

  // ---- B staging assignment -------------------------------------------------------------------------------------
  // GPTQ: thread = column (tid % BN), WPT consecutive word rows of the 8 in a k-tile -> WPT x b128 writes
  // AWQ : thread = word column (tid % (BN/8)) = 8 columns, WPT consecutive k rows -> k pairs, 4-byte writes per column
  const int bcol = (LAYOUT == 0) ? (tid % BN) : 8 * (tid % (BN / 8));
  const int brow = (LAYOUT == 0) ? WPT * (tid / BN) : WPT * (tid / (BN / 8));
  const int nB = n0 + bcol;
  constexpr int NC = (LAYOUT == 0) ? 1 : 8;
  ColConst cc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) cc[c] = make_col_const((half_t)0.f, (half_t)0.f);
  int curG = -1;
  const uint32_t nibmask = nib_mask_vgpr();

  auto group_of = [&](int k) { return p.gs_shift >= 0 ? (k >> p.gs_shift) : (k / p.group_size); };
  auto set_group = [&](int G) {
    if (G == curG) return;
    curG = G;
    if constexpr (LAYOUT == 0) {
      cc[0] = make_col_const(p.scales[(size_t)G * p.N + nB], zero_gptq(p, G, nB));
    } else {
      const half8_t sv = *(const half8_t *)(p.scales + (size_t)G * p.N + nB);
#pragma unroll
      for (int c = 0; c < 8; ++c) cc[c] = make_col_const(sv[c], zero_awq(p, G, nB + c));
    }
  };

  uint32_t breg[WPT];
  auto load_b = [&](int kt) {
#pragma unroll
    for (int r = 0; r < WPT; ++r) {
      if constexpr (LAYOUT == 0)
        breg[r] = p.qweight[(size_t)(kt * 8 + brow + r) * p.N + nB];
      else
        breg[r] = p.qweight[(size_t)(kt * BK + brow + r) * (p.N >> 3) + (nB >> 3)];
    }
  };
  // dequant + LDS write of this thread's 4 words, in two halves (h = 0, 1)
  auto store_b = [&](int kt, int buf, int h) {
    half_t *Bb = Bs + buf * kTile;
    const int k0 = kt * BK;
    if constexpr (LAYOUT == 0) {
#pragma unroll
      for (int r = (WPT / 2) * h; r < (WPT / 2) * (h + 1); ++r) {
        set_group(group_of(k0 + 8 * (brow + r)));
        const half8_t w = unperm_04152637(deq_word_k04(breg[r], cc[0], nibmask));
        *(half8_t *)(Bb + tile_off(bcol, brow + r)) = w;
      }
    } else {
      if (WPT == 2 && h == 1) return;  // two rows = one k pair, done in the first half
      set_group(group_of(k0 + brow));
      const uint32_t P = __builtin_amdgcn_perm(breg[2 * h + 1], breg[2 * h], 0x05040100u);
      const uint32_t Q = __builtin_amdgcn_perm(breg[2 * h + 1], breg[2 * h], 0x07060302u);
      // rows brow+2h, brow+2h+1 -> one k pair per column: 4-byte writes
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int sh = 4 * (c >> 1);
        const uint32_t s0 = (c & 1) ? Q : P;
        const half2_t b0 = deq_pair(and_or(s0 >> sh, nibmask, kMagic), cc[c]);
        *(half2_t *)(Bb + tile_off(bcol + c, brow >> 3) + (brow & 7) + 2 * h) = b0;
      }
    }
  };

  float4_t acc[AM][4];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = p.K / BK;
  load_a(0, 0);
  load_b(0);
  store_b(0, 0, 0);
  store_b(0, 0, 1);
  __syncthreads();  // (waits for the LDS-DMA: hipcc drains vmcnt before the barrier when a DMA is in flight)

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {
      load_a(kt + 1, buf ^ 1);
      load_b(kt + 1);
    }
    const half_t *Ab = As + buf * kTile, *Bb = Bs + buf * kTile;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8_t bf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) bf[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 16 + i, ks * 4 + g));
#pragma unroll
      for (int a = 0; a < AM; ++a) {
        const half8_t af = *(const half8_t *)(Ab + tile_off(wm * WROWS + a * 16 + i, ks * 4 + g));
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[b], acc[a][b], 0, 0, 0);
      }
      if (more) store_b(kt + 1, buf ^ 1, ks);
    }
    __syncthreads();
  }

  // ---- epilogue: + bias, round once, transpose through wave-private LDS, 16-byte row-contiguous stores ---------------
  half_t *ep = smem + wave * (16 * 72);  // 16 rows x 64 cols, row stride 72 halves (144 B: 16-byte aligned, bank-spread)
  float bv[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) bv[b] = p.bias ? (float)p.bias[n0 + wn * 64 + b * 16 + i] : 0.f;
#pragma unroll
  for (int a = 0; a < AM; ++a) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(4 * g + r) * 72 + b * 16 + i] = (half_t)(acc[a][b][r] + bv[b]);
    // 16 rows x 128 B = 128 chunks of 16 B: 2 per lane
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h, row = c >> 3, ch = c & 7;
      const uint4_t v = *(const uint4_t *)(ep + row * 72 + ch * 8);
      const int m = m0 + wm * WROWS + a * 16 + row;
      if (m < p.M) *(uint4_t *)((half_t *)p.y + (size_t)m * p.N + n0 + wn * 64 + ch * 8) = v;
    }
  }
}

bool gemm2_ok(const GemmParams &p, int layout) {
  static const char *e = getenv("QLLM_GEMM2");
  if (e && e[0] == '0') return false;
  if (p.act_bf16 || p.g_idx || p.K % 64 != 0 || p.N % 128 != 0 || p.M < 192) return false;
  if (layout == QLLM_LAYOUT_AWQ_GEMM) return p.group_size % 4 == 0;
  return p.group_size % 8 == 0;
}

template <int LAYOUT, int BN>
static int launch_gemm2_t(const GemmParams &p, hipStream_t stream) {
  using namespace g2;
  static bool attr_done = false;
  if (!attr_done) {
    QLLM_HIP_CHECK(hipFuncSetAttribute((const void *)gemm2_kernel<LAYOUT, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  const size_t lds = (size_t)4 * kTile * sizeof(half_t);  // 128 KB (A 2 x 32 KB, B 2 x <= 32 KB)
  hipLaunchKernelGGL((gemm2_kernel<LAYOUT, BN>), dim3(tiles), dim3(512), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_gemm2(const GemmParams &p, int layout, hipStream_t stream) {
  // 256x256 tiles when they fill the chip well; else 256x128 (twice the blocks)
  static int force_bn = getenv("QLLM_GEMM2_BN") ? atoi(getenv("QLLM_GEMM2_BN")) : 0;
  const int tiles256 = ((p.M + 255) / 256) * (p.N / 256);
  const int rounds = (tiles256 + kNumCU - 1) / kNumCU;
  const bool good256 = (p.N % 256 == 0) && tiles256 >= 0.85 * rounds * kNumCU;
  const int bn = force_bn ? force_bn : (good256 ? 256 : 128);
  if (layout == QLLM_LAYOUT_AWQ_GEMM) return bn == 256 ? launch_gemm2_t<1, 256>(p, stream) : launch_gemm2_t<1, 128>(p, stream);
  return bn == 256 ? launch_gemm2_t<0, 256>(p, stream) : launch_gemm2_t<0, 128>(p, stream);
}

}  // namespace qllm
