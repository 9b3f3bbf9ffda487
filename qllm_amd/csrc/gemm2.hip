// Prefill GEMM v2: y[M,N] = x[M,K] . dequant(W4) with a 256x256x64 block tile (fp16 activations, trivial groups).
//
// Same job as gemm.hip (which stays as the general path: bf16 activations, act-order, ragged N) with the structure
// the MI355X guide measures as the first big step for MFMA GEMMs (cdna_hip_programming.md section 5):
//   * 8 waves (2 M x 4 N), each 128x64 = 8x4 tiles of v_mfma_f32_16x16x32_f16: 12 fragment reads per 32 MFMAs
//     (the 128x128 kernel needs 8 per 16), and each weight is dequantised once per 256 rows of M;
//   * every global load is an ORDINARY load into registers (A: 4 x 16 B per thread, issued at the top of a k-tile and
//     written to the other LDS buffer after its MFMAs; packed B words: two register sets, loaded TWO k-tiles ahead).
//     Weights stream from HBM (~2 us latency) while a k-tile is ~0.4-0.9 us of MFMA: with one tile of look-ahead the
//     whole block stalls on its weight panel every step.  LDS-DMA for A (tried first) makes hipcc drain vmcnt(0) at
//     every barrier, which caps the look-ahead at one tile; plain loads survive the barrier (counted vmcnt);
//   * B: packed words -> registers -> bit-exact fp16 dequant -> ds_write into the k-contiguous [n][64 k] image
//     (the MFMA B-fragment order), same swizzle; the dequant of tile t+1 is split in two halves placed after each
//     32-MFMA sub-step of tile t so the VALU work rides under the other waves' MFMAs;
//   * double-buffered LDS (2 x 64 KB), one barrier per k-tile; XCD-aware block rasterisation;
//   * epilogue through wave-private LDS: 16-byte row-contiguous stores instead of 2-byte scattered ones.
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace qllm {

namespace g2 {
constexpr int BM = 256, BK = 64;
constexpr int kTile = 256 * BK;  // halves per A tile (32 KB); the B tile uses BN * BK of the same-size slot

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ lds_row_swizzle(row); }
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

// LAYOUT 0 = GPTQ/HQQ row stream, 1 = AWQ GEMM.  Requires: K % 64 == 0, N % 128 == 0 (bf16 activations are converted on load),
// group_size % 8 == 0 (GPTQ) / % 4 == 0 (AWQ), no g_idx.
// BN = 256: waves 2 (M) x 4 (N), 128x64 per wave.  BN = 128: waves 4 x 2, 64x64 per wave -- twice the blocks, for
// problems whose 256x256 tiling leaves CUs idle (M=2048 x N=4096 is only 128 such tiles).
template <int LAYOUT, int BN, bool BF16>
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
  // split-K (p.split_k > 1, problems with fewer tiles than CUs): S consecutive block ids share an output tile and own
  // consecutive K ranges; each writes its fp32 partial tile to a slab, the last to arrive sums them in fixed order
  const int S = p.split_k;
  const int nblk = tiles_m * tiles_n * S;
  int bid = blockIdx.x;
  {  // each XCD (block id % 8) walks a contiguous run of tiles, m fastest: neighbours share the weight panel
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ksplit = bid % S;
  bid /= S;
  const int tile_id = bid;
  const int KT_all = p.K / BK;
  const int KT_per = (KT_all + S - 1) / S;
  const int KT0 = ksplit * KT_per;                       // first k-tile of this block
  const int KT = max(0, min(KT_all, KT0 + KT_per) - KT0);  // its number of k-tiles (0: contributes a zero partial)
  // p.raster 0: m fastest (an XCD's run shares weight panels); 1: n fastest (an XCD's run shares activation rows:
  // 256 rows x K x 2 B = 2 MB per row block stays in its 4 MB L2 while the small packed weights stream)
  const int tm = p.raster ? (bid / tiles_n) : (bid % tiles_m);
  const int tn = p.raster ? (bid % tiles_n) : (bid / tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- A: registers -> LDS.  chunk c = tid + 512 q: row c/8, 16-byte k-chunk c%8 (8 lanes cover one 128-byte row) -----
  // two register sets like B (set (t & 1) = k-tile t) when the register budget allows (BN = 128): the activation
  // tile is then requested TWO k-tiles before it is written to LDS; with BN = 256 (250 VGPRs) one set, one tile ahead.
  constexpr int ASETS = (BN == 128) ? 2 : 1;
  uint4_t aset[ASETS][4];
  auto load_a = [&](int kt, uint4_t (&areg)[4]) {
    const int ktc = min(KT0 + min(kt, max(KT, 1) - 1), KT_all - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 512 * q, row = c >> 3, kc = c & 7;
      const int grow = min(m0 + row, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
      areg[q] = *(const uint4_t *)((const half_t *)p.x + (size_t)grow * p.K + ktc * BK + 8 * kc);
    }
  };
  auto store_a = [&](int buf, const uint4_t (&areg)[4], int q0 = 0, int q1 = 4) {
    half_t *Ab = As + buf * kTile;
#pragma unroll
    for (int q = q0; q < q1; ++q) {
      const int c = tid + 512 * q, row = c >> 3, kc = c & 7;
      if constexpr (BF16)  // bf16 activations are converted to fp16 (RNE) on their way into LDS
        *(half8_t *)(Ab + tile_off(row, kc)) = bf16x8_to_h8(areg[q]);
      else
        *(uint4_t *)(Ab + tile_off(row, kc)) = areg[q];
    }
  };

  // ---- B staging assignment -------------------------------------------------------------------------------------
  // GPTQ: thread = column (tid % BN), WPT consecutive word rows of the 8 in a k-tile -> WPT x b128 writes
  // AWQ : thread = word column (tid % (BN/8)) = 8 columns, WPT consecutive k rows -> k pairs, 4-byte writes per column
  const int bcol = (LAYOUT == 0) ? (tid % BN) : 8 * (tid % (BN / 8));
  const int brow = (LAYOUT == 0) ? WPT * (tid / BN) : WPT * (tid / (BN / 8));
  const int nB = n0 + bcol;
  const uint32_t nibmask = nib_mask_vgpr();
  auto group_of = [&](int k) { return k >> p.gs_shift; };  // power-of-two group sizes only (gemm2_ok): no branch
  // zero points, branch-free addressing (see strip.hip): packed -> word (G, n/8); fp16 -> dword holding half (G, n)
  const int zk = p.zero_kind;
  const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)p.scales : (const uint32_t *)p.qzeros;
  // p.sm (row-stream layouts): the same words stored strip-major (native layout): word (r, n) at ((n / 16) * K/8 + r) * 16 + n % 16,
  // scale / zero group rows hold the strip's 16 columns only
  const bool sm = (LAYOUT == 0) && p.sm;
  const int Gn = (p.K + (1 << p.gs_shift) - 1) >> p.gs_shift;
  const int ncs = sm ? (nB & 15) : nB;
  const int zmul = sm ? ((zk == ZK_PACKED) ? 2 : 8) : ((zk == ZK_PACKED) ? (p.N >> 3) : (p.N >> 1));
  const int zoff = ((zk == ZK_PACKED) ? (ncs >> 3) : (ncs >> 1)) + (sm ? (nB >> 4) * Gn * zmul : 0);
  const int wstride = sm ? 16 : p.N;                                         // words per packed row
  const size_t wcol = sm ? (size_t)(nB >> 4) * (size_t)(p.K >> 3) * 16 + ncs : (size_t)nB;
  const int sstride = sm ? 16 : p.N;                                         // halves per group row of the scale table
  const size_t scol = sm ? (size_t)(nB >> 4) * Gn * 16 + ncs : (size_t)nB;

  // One register set = everything this thread needs to dequantise its share of one k-tile: WPT packed words plus the
  // RAW scale / zero words of their group (a thread's rows span <= 32 k: one group, group_size % 32 == 0).  Two sets:
  // set (t & 1) belongs to k-tile t and is loaded two tiles ahead; nothing is converted before it is needed, so no
  // load is ever waited for early.
  struct BSet {
    uint32_t w[WPT];
    half8_t s8;     // AWQ: the 8 columns' scales
    uint32_t sraw;  // GPTQ: the column's scale, raw 16 bits in its own register (no read-modify-write of a packed reg)
    uint32_t z;
  };
  BSet bset[2];
  auto load_b = [&](int kt, BSet &bs) {
    const int ktc = min(KT0 + min(kt, max(KT, 1) - 1), KT_all - 1);  // past the end: harmless re-read, never used
#pragma unroll
    for (int r = 0; r < WPT; ++r) {
      if constexpr (LAYOUT == 0)
        bs.w[r] = p.qweight[(size_t)(ktc * 8 + brow + r) * wstride + wcol];
      else
        bs.w[r] = p.qweight[(size_t)(ktc * BK + brow + r) * (p.N >> 3) + (nB >> 3)];
    }
    const int G = group_of(ktc * BK + ((LAYOUT == 0) ? 8 * brow : brow));
    if constexpr (LAYOUT == 0) {
      bs.sraw = ((const uint16_t *)p.scales)[(size_t)G * sstride + scol];
    } else {
      bs.s8 = *(const half8_t *)(p.scales + (size_t)G * p.N + nB);
    }
    bs.z = zbase[(size_t)G * zmul + zoff];
  };
  // dequant + LDS write of this thread's words, in two halves (h = 0, 1)
  auto store_b = [&](int kt, int buf, int h, const BSet &bs) {
    half_t *Bb = Bs + buf * kTile;
    if constexpr (LAYOUT == 0) {
      const half_t zp = (half_t)(float)(((bs.z >> (4 * (nB & 7))) + (uint32_t)p.add_zero_bias) & 15u);
      const half_t zf = __builtin_bit_cast(half_t, (uint16_t)((nB & 1) ? (bs.z >> 16) : (bs.z & 0xffffu)));
      const half_t sc = __builtin_bit_cast(half_t, (uint16_t)bs.sraw);
      const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)8.f));
#pragma unroll
      for (int r = (WPT / 2) * h; r < (WPT / 2) * (h + 1); ++r) {
        const half8_t w = unperm_04152637(deq_word_k04(bs.w[r], cc, nibmask));
        *(half8_t *)(Bb + tile_off(bcol, brow + r)) = w;
      }
    } else {
      if (WPT == 2 && h == 1) return;  // two rows = one k pair, done in the first half
      const uint32_t P = __builtin_amdgcn_perm(bs.w[2 * h + 1], bs.w[2 * h], 0x05040100u);
      const uint32_t Q = __builtin_amdgcn_perm(bs.w[2 * h + 1], bs.w[2 * h], 0x07060302u);
      // rows brow+2h, brow+2h+1 -> one k pair per column: 4-byte writes
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int sh = 4 * (c >> 1);
        const uint32_t s0 = (c & 1) ? Q : P;
        const half_t z = (half_t)(float)((bs.z >> (4 * awq_nibble_of_col(c))) & 15u);
        const ColConst cc = make_col_const(bs.s8[c], z);
        const half2_t b0 = deq_pair(and_or(s0 >> sh, nibmask, kMagic), cc);
        *(half2_t *)(Bb + tile_off(bcol + c, brow >> 3) + (brow & 7) + 2 * h) = b0;
      }
    }
    (void)kt;
  };

  float4_t acc[AM][4];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};

  // prologue: tile 0 staged, tile 1's words in flight
  load_a(0, aset[0]);
  load_b(0, bset[0]);
  load_b(1, bset[1]);
  if (ASETS == 2) load_a(1, aset[ASETS - 1]);
  store_b(0, 0, 0, bset[0]);
  store_b(0, 0, 1, bset[0]);
  store_a(0, aset[0]);
  __syncthreads();

  // one k-tile: MFMAs on buffer `buf`; meanwhile load A(kt+ASETS) and B(kt+2), dequantise B(kt+1) into the other buffer.
  // EARLY = true: the dequant + LDS stores of tile kt+1 come BEFORE this tile's MFMAs instead of between/after them.
  // Waves w and w+4 share a SIMD; running waves 4-7 with EARLY (two fully separate straight-line bodies, joined only at
  // the barrier, so hipcc's counted vmcnt waits survive) puts one wave of every SIMD in its VALU phase while the other
  // is in its MFMA phase.  Needs the tile-(kt+1) words to be resident already, i.e. two register sets (BN = 128).
  auto mfma_sub = [&](const half_t *Ab, const half_t *Bb, int ks) {
    half8_t bf[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bf[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 16 + i, ks * 4 + g));
#pragma unroll
    for (int a = 0; a < AM; ++a) {
      const half8_t af = *(const half8_t *)(Ab + tile_off(wm * WROWS + a * 16 + i, ks * 4 + g));
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[b], acc[a][b], 0, 0, 0);
    }
  };
  auto k_tile = [&](int kt, BSet &b_next, BSet &b_free, uint4_t (&a_next)[4], uint4_t (&a_free)[4], auto early_tag) {
    constexpr bool EARLY = decltype(early_tag)::value;
    const int buf = kt & 1;
    // no conditionals in here: a branch around a load makes hipcc's counted vmcnt collapse to vmcnt(0) at the join.
    // Past the last tile the loads re-read it and the stores fill a buffer nobody reads again.
    // ASETS == 2: a_free held tile kt (already in LDS) -> request tile kt+2; a_next holds tile kt+1 (requested one tile ago)
    // ASETS == 1: a_free == a_next: request tile kt+1 now, write it at the end of this tile
    load_a(kt + ASETS, a_free);
    load_b(kt + 2, b_free);  // b_free held tile kt (dequantised during tile kt-1): refill it ~1.5 tiles ahead of use
    const half_t *Ab = As + buf * kTile, *Bb = Bs + buf * kTile;
    if constexpr (EARLY) {
      store_b(kt + 1, buf ^ 1, 0, b_next);
      store_b(kt + 1, buf ^ 1, 1, b_next);
      store_a(buf ^ 1, a_next);
      mfma_sub(Ab, Bb, 0);
      mfma_sub(Ab, Bb, 1);
    } else {
      // Order of the LDS stores relative to the two MFMA sub-steps is a COMPILE-TIME choice: a runtime switch between
      // the two orders cost 2.7x (826 -> 305 TFLOP/s, branch joins collapse the counted waits).  Measured at M=2048:
      // AWQ-layout tiles gain 3-5% with both sub-steps first, GPTQ tiles lose 3-5%.
      if constexpr (LAYOUT == QLLM_LAYOUT_AWQ_GEMM) {
        mfma_sub(Ab, Bb, 0);
        mfma_sub(Ab, Bb, 1);
        store_b(kt + 1, buf ^ 1, 0, b_next);
        store_b(kt + 1, buf ^ 1, 1, b_next);
        store_a(buf ^ 1, a_next);
      } else {
        mfma_sub(Ab, Bb, 0);
        store_b(kt + 1, buf ^ 1, 0, b_next);
        mfma_sub(Ab, Bb, 1);
        store_b(kt + 1, buf ^ 1, 1, b_next);
        store_a(buf ^ 1, a_next);
      }
    }
    __syncthreads();
  };
  using TagF = std::integral_constant<bool, false>;
  // (Tried three times: running waves 4-7 out of phase so that the two waves of every SIMD alternate VALU and MFMA phases.
  //  With the plain body (EARLY = true for waves 4-7): 771 vs 808 TFLOP/s.  With the fragment-pipelined body (waves 4-7
  //  store tile kt+2 during the second MFMA block of tile kt, loads three tiles ahead): two loop bodies in one kernel need
  //  > 256 registers -- 42-64 spilled with two activation sets, 23-46 with one -- and ran 734/725/815 (GPTQ) and
  //  494/452/517 (AWQ) against 862/866/977 and 806/804/882 in lockstep.  Not kept.)
  if constexpr (BN == 128) {
    // Fragment-pipelined body.  The plain body above starts every k-tile with "barrier -> 8 ds_read_b128 -> wait -> MFMA":
    // all 8 waves sit out the LDS round trip together, then the stores, then the barrier -- the MFMA pipe idles 2/3 of
    // the time.  Here the fragments of sub-step 1 are read BEFORE sub-step 0's MFMAs and those of the next tile's
    // sub-step 0 right after the barrier, before sub-step 1's MFMAs: every LDS read has 16 MFMAs (>= 256 cycles) of
    // cover, and the dequant + LDS stores of tile kt+1 are cut in four pieces placed between the 4-MFMA rows of sub-step 0.
    half8_t fa0[AM], fb0[4], fa1[AM], fb1[4];
    auto read_frags = [&](int buf, int ks, half8_t (&fa)[AM], half8_t (&fb)[4]) {
      const half_t *Ab = As + buf * kTile, *Bb = Bs + buf * kTile;
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 16 + i, ks * 4 + g));
#pragma unroll
      for (int a = 0; a < AM; ++a) fa[a] = *(const half8_t *)(Ab + tile_off(wm * WROWS + a * 16 + i, ks * 4 + g));
    };
    auto mfma_rows = [&](const half8_t (&fa)[AM], const half8_t (&fb)[4], int a0, int a1) {
#pragma unroll
      for (int a = a0; a < a1; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
    };
    // Measured at M=2048 (TFLOP/s, 4096x4096 / 4096x11008 / 11008x4096; plain body = 835/849/949 GPTQ, 794/795/881 AWQ):
    // GPTQ tiles are best with hipcc free to interleave the pieces (870/881/976; pinned 826/838/926); AWQ tiles, whose
    // dequant is 8 scalar columns per word, with the pieces pinned by sched_barriers (801/814/876; free 784/795/869).
#define G2_SB()                                                                          \
  do {                                                                                   \
    if constexpr (LAYOUT == QLLM_LAYOUT_AWQ_GEMM) __builtin_amdgcn_sched_barrier(0);     \
  } while (0)
    auto k_tile_p = [&](int kt, BSet &b_next, BSet &b_free, uint4_t (&a_next)[4], uint4_t (&a_free)[4]) {
      const int buf = kt & 1;
      read_frags(buf, 1, fa1, fb1);
      load_a(kt + ASETS, a_free);
      load_b(kt + 2, b_free);
      G2_SB();
      mfma_rows(fa0, fb0, 0, 1);
      store_b(kt + 1, buf ^ 1, 0, b_next);
      G2_SB();
      mfma_rows(fa0, fb0, 1, 2);
      store_b(kt + 1, buf ^ 1, 1, b_next);
      G2_SB();
      mfma_rows(fa0, fb0, 2, 3);
      store_a(buf ^ 1, a_next, 0, 2);
      G2_SB();
      mfma_rows(fa0, fb0, 3, 4);
      store_a(buf ^ 1, a_next, 2, 4);
      G2_SB();
      __syncthreads();
      read_frags(buf ^ 1, 0, fa0, fb0);  // past the last tile: reads a buffer nobody uses
      G2_SB();
      mfma_rows(fa1, fb1, 0, AM);
      G2_SB();
    };
#undef G2_SB
    read_frags(0, 0, fa0, fb0);
    for (int kt = 0; kt < KT; kt += 2) {
      k_tile_p(kt, bset[1], bset[0], aset[ASETS - 1], aset[0]);
      if (kt + 1 < KT) k_tile_p(kt + 1, bset[0], bset[1], aset[0], aset[ASETS - 1]);
    }
    __syncthreads();  // the epilogue reuses the tile buffers: the stray fragment reads above must have landed
  } else {
    for (int kt = 0; kt < KT; kt += 2) {
      k_tile(kt, bset[1], bset[0], aset[ASETS - 1], aset[0], TagF{});
      if (kt + 1 < KT) k_tile(kt + 1, bset[0], bset[1], aset[0], aset[ASETS - 1], TagF{});
    }
  }

  // ---- split-K: publish the fp32 partial tile; the last block to arrive sums the S partials in fixed order --------------
  // Protocol as in skinny.hip (cdna_hip_programming.md, in-launch split-K): write-through (sc1) stores, every storing wave
  // drains them, one relaxed agent-scope ticket per block, the last arriver reads with sc1 loads and re-arms the counter.
  // Slab element (tile, split, wave, register r, lane): every store / load instruction of a wave covers 256 contiguous bytes.
  if (S > 1) {
    // (the ticket lives in the dynamic segment, past the epilogue's wave-private regions: a static __shared__ word on top of
    //  the 160 KB dynamic maximum makes hipFuncSetAttribute fail)
    int &s_ticket = *(int *)(smem + 8 * (16 * 72));
    float *slab = p.slabs + ((size_t)tile_id * S + ksplit) * (size_t)(BM * BN) + (size_t)wave * (AM * 16 * 64) + lane;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) st_sc1(slab + ((a * 4 + b) * 4 + r) * 64, acc[a][b][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != S - 1) return;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S; ++s) {
      const float *src = p.slabs + ((size_t)tile_id * S + s) * (size_t)(BM * BN) + (size_t)wave * (AM * 16 * 64) + lane;
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][r] += ld_sc1(src + ((a * 4 + b) * 4 + r) * 64);
    }
    if (tid == 0) __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
      for (int r = 0; r < 4; ++r) {
        const float v = acc[a][b][r] + bv[b];
        if constexpr (BF16)
          ((uint16_t *)ep)[(4 * g + r) * 72 + b * 16 + i] = f32_to_bf16(v);
        else
          ep[(4 * g + r) * 72 + b * 16 + i] = (half_t)v;
      }
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
  if (!knob("QLLM_GEMM2", 1)) return false;
  // below 192 rows a 256-row tile wastes > 25 % of its MFMAs -- but with split-K the k-loop length, not the MFMA count, sets
  // the time at these sizes.  Measured (us per linear, 128x128 kernel -> this one; profiles/r02_mid_m.md): M = 96..160 on
  // 4096x4096 100 -> 27.5, 4096x11008 103 -> 40, 11008x4096 258 -> 41.
  // 33..64 rows (round 3, profiles/r03_mid_m.md): on the 11008-wide Llama shapes this kernel (40 us) beats both the four-row-tile
  // strips (42-59 us) and the split-K decode kernel (48-53 us); the dispatcher sends it only what the strips do not take
  const int min_m = knob("QLLM_GEMM2_MIN_M", 33);
  if (p.g_idx || p.K % 64 != 0 || p.N % 128 != 0 || p.M < min_m) return false;
  return p.group_size % 32 == 0 && p.gs_shift >= 0;  // one group per thread per k-tile; power-of-two group size
}

template <int LAYOUT, int BN, bool BF16>
static int launch_gemm2_b(const GemmParams &p, hipStream_t stream) {
  using namespace g2;
  static DeviceLatch attr_done;  // per (kernel, device): the LDS opt-in is a per-device attribute
  if (int rc = lds_optin(attr_done, (const void *)gemm2_kernel<LAYOUT, BN, BF16>)) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN) * p.split_k;
  const size_t lds = (size_t)4 * kTile * sizeof(half_t);  // 128 KB (A 2 x 32 KB, B 2 x <= 32 KB)
  hipLaunchKernelGGL((gemm2_kernel<LAYOUT, BN, BF16>), dim3(tiles), dim3(512), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

template <int LAYOUT, int BN>
static int launch_gemm2_t(const GemmParams &p, hipStream_t stream) {
  return p.act_bf16 ? launch_gemm2_b<LAYOUT, BN, true>(p, stream) : launch_gemm2_b<LAYOUT, BN, false>(p, stream);
}

// split-K factor for a problem whose 256x128 tiling leaves CUs idle: the largest S <= 8 that still fits one round of blocks
// (tiles * S <= CUs) and leaves every block >= 8 k-tiles; 1 otherwise.  A block's k-loop runs ~0.9 us per k-tile whatever the
// occupancy, so M <= 1024 on 4096-wide layers is bound by the loop length, not by throughput.
int gemm2_split_k(int M, int N, int K) {
  if (!knob("QLLM_GEMM2_SPLITK", 1)) return 1;
  if (N % 128 != 0) return 1;
  const int tiles = ((M + 255) / 256) * (N / 128), kt = K / 64;
  int s = 1;
  while (s < 8 && tiles * (s * 2) <= compute_units() && kt / (s * 2) >= 8) s *= 2;
  return s;
}
size_t gemm2_slab_bytes(int M, int N, int S) { return S > 1 ? (size_t)((M + 255) / 256) * (N / 128) * S * 256 * 128 * sizeof(float) : 0; }

// 256x256 tiles when they fill the chip well; else 256x128 (twice the blocks)
int gemm2_tile_n(int M, int N, int split_k) {
  const int force_bn = knob("QLLM_GEMM2_BN", 0);
  if (split_k > 1) return 128;  // the slab layout is the 256x128 tile's
  const int tiles256 = ((M + 255) / 256) * (N / 256);
  const int rounds = (tiles256 + compute_units() - 1) / compute_units();
  const bool good256 = (N % 256 == 0) && tiles256 >= 0.85 * rounds * compute_units();
  return force_bn ? force_bn : (good256 ? 256 : 128);
}

int launch_gemm2(const GemmParams &p_in, int layout, hipStream_t stream) {
  GemmParams p = p_in;
  if (p.split_k < 1) p.split_k = 1;
  const int raster = knob("QLLM_GEMM2_RASTER", 1);  // measured +2-3 %
  p.raster = raster;
  const int stagger = knob("QLLM_GEMM2_STAGGER", 0);  // measured: 771 vs 808 TFLOP/s with it on
  p.stagger = stagger;
  const int bn = gemm2_tile_n(p.M, p.N, p.split_k);
  if (layout == QLLM_LAYOUT_AWQ_GEMM) return bn == 256 ? launch_gemm2_t<1, 256>(p, stream) : launch_gemm2_t<1, 128>(p, stream);
  return bn == 256 ? launch_gemm2_t<0, 256>(p, stream) : launch_gemm2_t<0, 128>(p, stream);
}

}  // namespace qllm
