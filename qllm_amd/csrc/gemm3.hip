// Prefill GEMM v3: y[M,N] = x[M,K] . dequant(W4), 256x128x64 block tile, WAVE-SPECIALISED: 4 matrix waves + 4 staging waves.
//
// Why: gemm2's eight waves all run the same body -- dequantise, stage, read fragments, MFMA -- and the two waves that share
// a SIMD move through those phases in lockstep between barriers, so the VALU / LDS-store work of one never hides under the
// MFMAs of the other; the matrix pipe is ~40 % busy (profiles/r01_prefill_summary.md).  Here the roles are split per SIMD
// (waves w and w+4 share one):
//   * waves 0-3, "matrix": each owns a 128x64 output tile = 4x2 tiles of v_mfma_f32_32x32x16_f16 (128 accumulator
//     registers).  Their instruction stream is ds_read_b128 + MFMA only: 6 fragment reads per 8 MFMAs (the 64x64-per-wave
//     16x16x32 form needs 8 per 8 MFMA-equivalents), fragments double-buffered one k16 sub-step ahead, the k-tile barrier
//     placed in front of the LAST sub-step's MFMAs so the next tile's first fragment reads are covered too.  One matrix wave
//     per SIMD keeps the pipe fed as long as its fragments arrive: 32 MFMAs x 32 cycles per k-tile against ~30 other issues.
//   * waves 4-7, "staging": activations (plain 16-byte loads -> registers -> ds_write_b128, two register sets, requested two
//     k-tiles ahead) and weights (packed words + raw scale/zero words, two sets; bit-exact 3-op fp16 dequant as everywhere
//     else, common.hpp) into the OTHER LDS stage.  Their VALU and LDS-store work issues beside the matrix wave's MFMAs
//     (separate pipes; the matrix waves run at raised priority).
//   * two LDS stages of (256x64 A + 128x64 B) halves = 96 KB, one workgroup barrier per k-tile:
//         staging, iteration t:  [request tile t+2] [write tile t+1 -> stage (t+1)%2]            barrier #t
//         matrix,  iteration t:  sub-steps 0..2 of tile t (stage t%2), fragments of 3 in registers  barrier #t  [read tile
//                                t+1's first fragments] [MFMAs of sub-step 3]
//     after barrier #t nobody reads stage t%2 any more (the staging waves overwrite it in iteration t+1) and stage (t+1)%2 is
//     complete.  __syncthreads() waits for the issuing wave's LDS operations (lgkmcnt(0)) before the barrier; plain global
//     loads stay in flight across it (counted vmcnt).
//   * LDS rows are 64 halves (128 B) with the eight 16-byte slots XORed by (row >> 1) & 7: conflict-free for the 32x32x16
//     fragment reads (lane l -> row l % 32, slot 2*ks + l / 32: each 16-lane service group of ds_read_b128 sees 8 even and
//     8 odd rows with 8 distinct row>>1 values mod 8), for the activation stores (8 lanes = the 8 slots of one row) and for
//     the GPTQ weight stores (8 consecutive rows at one slot).
//   * epilogue: + bias, one rounding, transposed through wave-private LDS into 16-byte row-contiguous stores.
// Serves what gemm2's 256x128 form serves when no split-K is wanted (M >= 1024 on the Llama shapes); gemm2 keeps the rest.
// Replaces gemm_forward_4bit_cuda_m16n128k32 (/root/reference/csrc/awq_cuda/quantization/gemm_cuda_gen.cu:31-353).
#include <stdlib.h>

#include "kernels.hpp"

namespace qllm {

namespace g3 {
constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kATile = BM * BK, kBTile = BN * BK;  // halves per stage
typedef float float16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + ((slot ^ (row >> 1)) & 7) * 8; }  // in halves
}  // namespace g3

// LAYOUT 0 = GPTQ/HQQ row stream, 1 = AWQ GEMM.  Requires K % 128 == 0 (even number of k-tiles), N % 128 == 0, power-of-two group size >= 32, no g_idx.
template <int LAYOUT, bool BF16>
__global__ __launch_bounds__(512) void gemm3_kernel(const GemmParams p) {
  using namespace g3;
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;               // [2][256][64]
  half_t *Bs = smem + 2 * kATile;  // [2][128 n][64 k]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // each XCD (block id % 8) walks a contiguous run of tiles
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // p.raster 1: n fastest (an XCD's run shares activation rows, which stay in its L2 while the small packed weights stream)
  const int tm = p.raster ? (bid / tiles_n) : (bid % tiles_m);
  const int tn = p.raster ? (bid % tiles_n) : (bid / tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = p.K / BK;

  if (wave >= 4) {
    // ================================================= staging waves ==================================================
    const int t = tid - 256;  // 0..255
    // ---- A: chunk c = t + 256 q: row c / 8, 16-byte k-chunk c % 8 (8 lanes cover one 128-byte row segment) ------------
    uint4_t aset[2][8];
    auto load_a = [&](int kt, uint4_t (&areg)[8]) {
      const int ktc = min(kt, KT - 1);  // past the end: harmless re-read
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q, row = c >> 3, kc = c & 7;
        const int grow = min(m0 + row, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
        areg[q] = *(const uint4_t *)((const half_t *)p.x + (size_t)grow * p.K + ktc * BK + 8 * kc);
      }
    };
    auto store_a = [&](int stage, const uint4_t (&areg)[8]) {
      half_t *Ab = As + stage * kATile;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int c = t + 256 * q, row = c >> 3, kc = c & 7;
        if constexpr (BF16)
          *(half8_t *)(Ab + tile_off(row, kc)) = bf16x8_to_h8(areg[q]);
        else
          *(uint4_t *)(Ab + tile_off(row, kc)) = areg[q];
      }
    };
    // ---- B -------------------------------------------------------------------------------------------------------------
    // GPTQ: thread = column t % 128, word rows 4 (t / 128) .. +3 of the 8 in a k-tile -> 4 x b128 stores
    // AWQ : thread = word column t % 16 (8 columns), k rows 4 (t / 16) .. +3 -> per column one 8-byte store of 4 k
    const int bcol = (LAYOUT == 0) ? (t & 127) : 8 * (t & 15);
    const int brow = (LAYOUT == 0) ? 4 * (t >> 7) : 4 * (t >> 4);
    const int nB = n0 + bcol;
    const uint32_t nibmask = nib_mask_vgpr();
    const int zk = p.zero_kind;
    const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)p.scales : (const uint32_t *)p.qzeros;
    const int zmul = (zk == ZK_PACKED) ? (p.N >> 3) : (p.N >> 1);
    const int zoff = (zk == ZK_PACKED) ? (nB >> 3) : (nB >> 1);
    struct BSet {
      uint32_t w[4];
      half8_t s8;     // AWQ: the 8 columns' scales
      uint32_t sraw;  // GPTQ: the column's scale, raw 16 bits
      uint32_t z;
    };
    BSet bset[2];
    auto load_b = [&](int kt, BSet &bs) {
      const int ktc = min(kt, KT - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (LAYOUT == 0)
          bs.w[r] = p.qweight[(size_t)(ktc * 8 + brow + r) * p.N + nB];
        else
          bs.w[r] = p.qweight[(size_t)(ktc * BK + brow + r) * (p.N >> 3) + (nB >> 3)];
      }
      const int G = (ktc * BK + ((LAYOUT == 0) ? 8 * brow : brow)) >> p.gs_shift;  // one group per thread per k-tile
      if constexpr (LAYOUT == 0)
        bs.sraw = ((const uint16_t *)p.scales)[(size_t)G * p.N + nB];
      else
        bs.s8 = *(const half8_t *)(p.scales + (size_t)G * p.N + nB);
      bs.z = zbase[(size_t)G * zmul + zoff];
    };
    auto store_b = [&](int stage, const BSet &bs) {
      half_t *Bb = Bs + stage * kBTile;
      if constexpr (LAYOUT == 0) {
        const half_t zp = (half_t)(float)(((bs.z >> (4 * (nB & 7))) + (uint32_t)p.add_zero_bias) & 15u);
        const half_t zf = __builtin_bit_cast(half_t, (uint16_t)((nB & 1) ? (bs.z >> 16) : (bs.z & 0xffffu)));
        const half_t sc = __builtin_bit_cast(half_t, (uint16_t)bs.sraw);
        const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)8.f));
#pragma unroll
        for (int r = 0; r < 4; ++r) *(half8_t *)(Bb + tile_off(bcol, brow + r)) = unperm_04152637(deq_word_k04(bs.w[r], cc, nibmask));
      } else {
        // rows brow..brow+3 of 8 interleaved columns: column c of the word sits at nibble awq_nibble_of_col(c).  Two
        // v_perm build, per column pair, the (k0,k1) and (k2,k3) nibble-bearing 16-bit halves side by side.
        const uint32_t P01 = __builtin_amdgcn_perm(bs.w[1], bs.w[0], 0x05040100u), Q01 = __builtin_amdgcn_perm(bs.w[1], bs.w[0], 0x07060302u);
        const uint32_t P23 = __builtin_amdgcn_perm(bs.w[3], bs.w[2], 0x05040100u), Q23 = __builtin_amdgcn_perm(bs.w[3], bs.w[2], 0x07060302u);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int sh = 4 * (c >> 1);
          const half_t z = (half_t)(float)((bs.z >> (4 * awq_nibble_of_col(c))) & 15u);
          const ColConst cc = make_col_const(bs.s8[c], z);
          const half2_t b01 = deq_pair(and_or(((c & 1) ? Q01 : P01) >> sh, nibmask, kMagic), cc);
          const half2_t b23 = deq_pair(and_or(((c & 1) ? Q23 : P23) >> sh, nibmask, kMagic), cc);
          *(uint2_t *)(Bb + tile_off(bcol + c, brow >> 3) + (brow & 7)) = uint2_t{as_u32(b01), as_u32(b23)};
        }
      }
    };

    // Order of one iteration: [write tile kt+1 from its register set] barrier [request tile kt+3 into that set].  A set is
    // requested right after the barrier that frees it and consumed two barriers later (~1.5 k-tiles of flight), and the only
    // vmcnt wait in the loop is the counted one in front of the stores (the younger set's 14 loads stay in flight).  Nothing
    // is conditional around a load (a branch there makes hipcc's counted vmcnt collapse to vmcnt(0)): past the last tile the
    // loads re-read it and the stores fill a stage nobody reads again.
    load_b(0, bset[0]);
    load_a(0, aset[0]);
    load_b(1, bset[1]);
    load_a(1, aset[1]);
    __builtin_amdgcn_sched_barrier(0);
    store_b(0, bset[0]);
    store_a(0, aset[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_b(2, bset[0]);
    load_a(2, aset[0]);
    __syncthreads();  // prologue barrier: stage 0 holds tile 0
    for (int kt = 0; kt < KT; kt += 2) {
      store_b(1, bset[1]);
      store_a(1, aset[1]);
      __syncthreads();  // barrier #kt
      load_b(kt + 3, bset[1]);
      load_a(kt + 3, aset[1]);
      __builtin_amdgcn_sched_barrier(0);
      store_b(0, bset[0]);  // (KT is even -- gemm3_ok -- so the two halves need no branch between them)
      store_a(0, aset[0]);
      __syncthreads();  // barrier #kt+1
      load_b(kt + 4, bset[0]);
      load_a(kt + 4, aset[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }

  // =================================================== matrix waves ===================================================
  const int wm = wave >> 1, wn = wave & 1;  // 2 (M) x 2 (N): rows wm*128.., columns wn*64..
  const int fr = lane & 31, fs = lane >> 5;  // fragment row (A: m, B: n) and k half of the 16-wide sub-step
  float16_t acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  half8_t fa0[4], fb0[2], fa1[4], fb1[2];
  auto read_frags = [&](int stage, int ks, half8_t (&fa)[4], half8_t (&fb)[2]) {
    const half_t *Ab = As + stage * kATile, *Bb = Bs + stage * kBTile;
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 32 + fr, ks * 2 + fs));
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = *(const half8_t *)(Ab + tile_off(wm * 128 + a * 32 + fr, ks * 2 + fs));
  };
  auto mfma_all = [&](const half8_t (&fa)[4], const half8_t (&fb)[2]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
  };

  __builtin_amdgcn_s_setprio(2);  // the matrix wave outranks its SIMD's staging wave for issue slots
  __syncthreads();                // prologue barrier
  read_frags(0, 0, fa0, fb0);
  // The issue order is pinned (sched_barrier): left alone, hipcc sinks every fragment read to just above its first use
  // (fewest live registers) and the lone matrix wave of the SIMD then sits out each LDS round trip with an idle matrix pipe.
#define G3_SB() __builtin_amdgcn_sched_barrier(0)
  for (int kt = 0; kt < KT; ++kt) {
    const int st = kt & 1;
    read_frags(st, 1, fa1, fb1);
    G3_SB();
    mfma_all(fa0, fb0);  // sub-step 0
    G3_SB();
    read_frags(st, 2, fa0, fb0);
    G3_SB();
    mfma_all(fa1, fb1);  // sub-step 1
    G3_SB();
    read_frags(st, 3, fa1, fb1);
    G3_SB();
    mfma_all(fa0, fb0);  // sub-step 2
    G3_SB();
    __syncthreads();     // barrier #kt: stage st^1 complete, stage st free (its last fragments are in registers)
    read_frags(st ^ 1, 0, fa0, fb0);  // past the last tile: a stage nobody uses
    G3_SB();
    mfma_all(fa1, fb1);  // sub-step 3
    G3_SB();
  }
#undef G3_SB
  __builtin_amdgcn_s_setprio(0);

  // ---- epilogue: + bias, round once, transpose through wave-private LDS, 16-byte row-contiguous stores ---------------------
  // C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  (All fragment reads that
  // matter completed before the last barrier; the stray ones above only fill registers.)
  half_t *ep = smem + wave * (32 * 72);  // 32 rows x 64 cols, row stride 72 halves (144 B)
  float bv[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) bv[b] = p.bias ? (float)p.bias[n0 + wn * 64 + b * 32 + fr] : 0.f;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fs;
        const float v = acc[a][b][r] + bv[b];
        if constexpr (BF16)
          ((uint16_t *)ep)[row * 72 + b * 32 + fr] = f32_to_bf16(v);
        else
          ep[row * 72 + b * 32 + fr] = (half_t)v;
      }
    // 32 rows x 128 B = 256 chunks of 16 B: 4 per lane (wave-private region: no barrier, the wave's own LDS ops are ordered)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int c = lane + 64 * h, row = c >> 3, ch = c & 7;
      const uint4_t v = *(const uint4_t *)(ep + row * 72 + ch * 8);
      const int m = m0 + wm * 128 + a * 32 + row;
      if (m < p.M) *(uint4_t *)((half_t *)p.y + (size_t)m * p.N + n0 + wn * 64 + ch * 8) = v;
    }
  }
}

bool gemm3_ok(const GemmParams &p, int layout) {
  (void)layout;
  static const int on = getenv("QLLM_GEMM3") ? atoi(getenv("QLLM_GEMM3")) : 1;
  static const int min_m = getenv("QLLM_GEMM3_MIN_M") ? atoi(getenv("QLLM_GEMM3_MIN_M")) : 1024;
  if (!on || p.g_idx || p.K % 128 != 0 || p.N % 128 != 0 || p.M < min_m) return false;  // K % 128: an even number of k-tiles
  return p.group_size % 32 == 0 && p.gs_shift >= 5;
}

template <int LAYOUT, bool BF16>
static int launch_gemm3_b(const GemmParams &p, hipStream_t stream) {
  using namespace g3;
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)gemm3_kernel<LAYOUT, BF16>)) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  const size_t lds = (size_t)2 * (kATile + kBTile) * sizeof(half_t);  // 96 KB
  hipLaunchKernelGGL((gemm3_kernel<LAYOUT, BF16>), dim3(tiles), dim3(512), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_gemm3(const GemmParams &p_in, int layout, hipStream_t stream) {
  GemmParams p = p_in;
  static int raster = getenv("QLLM_GEMM2_RASTER") ? atoi(getenv("QLLM_GEMM2_RASTER")) : 1;
  p.raster = raster;
  if (layout == QLLM_LAYOUT_AWQ_GEMM) return p.act_bf16 ? launch_gemm3_b<1, true>(p, stream) : launch_gemm3_b<1, false>(p, stream);
  return p.act_bf16 ? launch_gemm3_b<0, true>(p, stream) : launch_gemm3_b<0, false>(p, stream);
}

}  // namespace qllm
