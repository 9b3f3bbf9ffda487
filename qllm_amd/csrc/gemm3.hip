// Prefill GEMM v3: y[M,N] = x[M,K] . dequant(W4), 256x128x64 block tile, WAVE-SPECIALISED: 4 matrix waves + 4 dequant waves.
//
// Why: gemm2's eight waves all run the same body -- dequantise, stage, read fragments, MFMA -- and the two waves that share
// a SIMD move through those phases in lockstep between barriers, so the VALU / LDS-store work of one never hides under the
// MFMAs of the other; the matrix pipe is ~40 % busy (profiles/r01_prefill_summary.md).  Here the roles are split per SIMD
// (waves w and w+4 share one):
//   * waves 0-3, "matrix": each owns a 128x64 output tile = 4x2 tiles of v_mfma_f32_32x32x16_f16 (128 accumulator
//     registers).  Instruction stream: ds_read_b128 + MFMA (6 fragment reads per 8 MFMAs; fragments double-buffered one k16
//     sub-step ahead; the k-tile barrier sits in front of the LAST sub-step's MFMAs so the next tile's first fragment reads
//     are covered too) plus the ACTIVATION tile, which needs no arithmetic: 8 LDS-DMA pieces per wave and k-tile
//     (buffer_load_dwordx4 ... lds: 1 KB = 8 rows x 128 B each, no registers, no ds_write), one behind every four MFMAs,
//     requested two k-tiles ahead into a 3-deep ring and retired with a counted vmcnt(8) in front of the barrier.  These waves issue no other vector
//     memory operation, so the hand-placed vmcnt counts exactly the DMA pieces.
//   * waves 4-7, "dequant": packed weight words + raw scale/zero words through buffer loads (per-lane offsets fixed, the
//     k-tile advance is a scalar offset: no address VALU), two register sets requested ~1.5 k-tiles ahead, the bit-exact
//     3-op fp16 dequant (common.hpp) and ds_write_b128 into the other B stage.  (First version: these waves also staged the
//     activations through registers -- 12 ds_write_b128 + ~130 VALU per k-tile in ONE wave per SIMD -- and were the
//     bottleneck: 748 TFLOP/s against gemm2's 846 on 2048x4096x4096.)
//   * LDS: A 3 x 32 KB + B 2 x 16 KB = 128 KB, one workgroup barrier per k-tile:
//         dequant, iteration t:  [write B tile t+1 -> stage (t+1)%2]                                      barrier #t
//         matrix,  iteration t:  sub-steps 0..2 of tile t, fragments of 3 in registers, vmcnt(8)          barrier #t
//                                [request A tile t+3 -> ring slot t%3] [read tile t+1's first fragments] [MFMAs of sub-step 3]
//     after barrier #t nobody reads B stage t%2 / A slot t%3 any more, and B stage (t+1)%2 / A slot (t+1)%3 are complete
//     (every matrix wave waited for its own DMA pieces of tile t+1 before arriving).  The matrix waves use the raw
//     s_barrier with explicit lgkmcnt(0) / vmcnt(8): __syncthreads() would drain the DMA ring (vmcnt(0)).
//   * LDS rows are 64 halves (128 B) with the eight 16-byte slots XORed by lds_row_swizzle(row) (common.hpp; the plain
//     (row >> 1) & 7 of the first version left the staging waves' ds_write_b128 2-way conflicted: 8 consecutive rows hit only 4
//     slots -- 2.2 M of 15.3 M LDS cycles per launch): conflict-free for the 32x32x16
//     fragment reads (lane l -> row l % 32, slot 2*ks + l / 32: each 16-lane service group of ds_read_b128 sees 8 even and
//     8 odd rows with 8 distinct row>>1 values mod 8) and for the GPTQ weight stores (8 consecutive rows at one slot).  An
//     LDS-DMA piece lands lane-linear (lane l -> row l / 8, physical slot l % 8), so the swizzle is applied to the SOURCE:
//     lane l fetches logical chunk (l % 8) ^ lds_row_swizzle(row) of its row (cdna_hip_programming.md rule 21).
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
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + ((slot ^ lds_row_swizzle(row)) & 7) * 8; }  // in halves
}  // namespace g3

// LAYOUT 0 = GPTQ/HQQ row stream (4 bits), 1 = AWQ GEMM, 2 = GPTQ/HQQ row stream with 3-bit weights (bit stream: the 32 k of
// a half k-tile are 3 consecutive word rows of the column).  p.sm (LAYOUT 0 / 2): the same words stored strip-major (the native
// layout, include/qllm_mi355x.h): only the loop-constant per-lane offsets and the row strides change -- a thread's four words are
// 4 x 64 B apart inside its strip instead of 4 x 4N B apart.  fp16 activations.  Requires K % 64 == 0, N % 128 == 0, power-of-two group size >= 32, no g_idx.
// MW: matrix waves, 4 (one per SIMD, 128x64 each) or 8 (two per SIMD, 64x64 each: 8 fragment reads per 8 MFMAs instead of 6,
// but the two waves cover each other's LDS / barrier waits); always 4 dequant waves behind them.
// BF (round 6, LAYOUT 0 only): NATIVE bf16 -- the activation tiles are the caller's bf16 rows as they are (LDS-DMA moves bytes), the
// staging waves dequantise to bf16 (q as fp32 via v_cvt_f32_ubyte, one v_pk_fma_f32 per pair: W = bf16(s q - s z), ONE rounding;
// v_cvt_pk_bf16_f32), the matrix waves run v_mfma_f32_32x32x16_bf16 and the epilogue rounds the fp32 sums to bf16 once.  No conversion
// pre-pass, no fp16 overflow for |x| > 65504 (the reference's shim casts x to fp16: quant_linear_awq.py:29-36).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int LAYOUT, int MW, bool PRIO = true, bool BF = false>
__global__ __launch_bounds__((MW + 4) * 64) void gemm3_kernel(const GemmParams p) {
  static_assert(!BF || LAYOUT == 0, "native bf16: row-stream / strip-major 4-bit layers");
  using namespace g3;
  constexpr int AM = 8 / MW * 2;       // 32-row MFMA tiles per matrix wave along M: 4 or 2
  constexpr int WROWS = AM * 32;       // rows per matrix wave: 128 or 64
  constexpr int NP = 32 / MW;          // activation DMA pieces (8 rows x 128 B) per matrix wave and k-tile: 8 or 4
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;               // [3][256][64]  (LDS-DMA ring)
  half_t *Bs = smem + 3 * kATile;  // [2][128 n][64 k]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Round 6, grouped launches (p.n_prob > 1: layers sharing x -- q/k/v, gate/up -- in ONE grid): the tiles of layer 0 come first, then
  // layer 1's ...; a block looks its layer up from its tile id and takes that layer's pointers and width (P* below).  The rounds of
  // tiles are counted over the whole group: gate/up of Llama-2-7B are 1376 tiles = 5.4 rounds instead of 2 x 2.7 -> 2 x 3.
  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_all = p.n_prob > 1 ? p.total_tiles : tiles_m * (p.N / BN);
  // split-K (p.split_k > 1: fewer tiles than CUs): S consecutive block ids share an output tile and own consecutive K ranges;
  // each publishes its fp32 partial tile to a slab, the last to arrive sums them in fixed order (the protocol of gemm2.hip)
  // Round 6, tail split (p.tail_split > 1, tiles > CUs): the tiles of the whole rounds run unsplit; the tiles of the ragged last round
  // are shared by tail_split blocks each -- two segments of block ids, each walked XCD-contiguously.
  auto xcd_run = [](int b, int n) {  // each XCD (block id % 8) walks a contiguous run of the n blocks
    const int q = n / 8, r = n % 8, xcd = b % 8, idx = b / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  };
  int S, ksplit, tile_id, slab_tile;
  if (p.tail_split > 1) {
    const int F = p.tail_from, bid = blockIdx.x;
    if (bid < F) {
      S = 1;
      ksplit = 0;
      tile_id = slab_tile = xcd_run(bid, F);
    } else {
      const int u = xcd_run(bid - F, (tiles_all - F) * p.tail_split);
      S = p.tail_split;
      ksplit = u % S;
      slab_tile = u / S;
      tile_id = F + slab_tile;
    }
  } else {
    S = p.split_k;
    const int bid = xcd_run(blockIdx.x, tiles_all * S);
    ksplit = bid % S;
    tile_id = slab_tile = bid / S;
  }
  S = __builtin_amdgcn_readfirstlane(S);
  const uint32_t *Pqweight = p.qweight;
  const half_t *Pscales = p.scales, *Pbias = p.bias;
  const void *Pqzeros = p.qzeros;
  void *Py = p.y;
  int PN = p.N, Pzk = p.zero_kind, bid = tile_id;
  if (p.n_prob > 1) {
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kGemm3MaxProb; ++i) pi += (i < p.n_prob && tile_id >= p.prob[i].tile_begin) ? 1 : 0;
    const GemmProb &q = p.prob[pi];
    Pqweight = q.qweight; Pscales = q.scales; Pbias = q.bias; Pqzeros = q.qzeros; Py = q.y;
    PN = q.N; Pzk = q.zero_kind; bid = tile_id - q.tile_begin;
  }
  const int tiles_n = PN / BN;
  // p.raster 1: n fastest (an XCD's run shares activation rows, which stay in its L2 while the small packed weights stream)
  const int tm = p.raster ? (bid / tiles_n) : (bid % tiles_m);
  const int tn = p.raster ? (bid % tiles_n) : (bid / tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = p.K / BK / S;   // this block's k-tiles (launch_gemm3 only splits when K / 64 is a multiple of S)
  const int KT0 = ksplit * KT;  // ... starting at this one

  if (wave >= MW) {
    // ================================================= dequant waves ==================================================
    const int t = tid - MW * 64;  // 0..255
    // ---- B -------------------------------------------------------------------------------------------------------------
    // GPTQ: thread = column t % 128, word rows 4 (t / 128) .. +3 of the 8 in a k-tile -> 4 x b128 stores
    //       (3 bits: word rows 3 (t / 128) .. +2 of the 6 in a k-tile = the same 32 k -> the same 4 stores)
    // AWQ : thread = word column t % 16 (8 columns), k rows 4 (t / 16) .. +3 -> per column one 8-byte store of 4 k
    constexpr bool ROWS = LAYOUT != 1;   // row-stream layouts
    constexpr int WPT = (LAYOUT == 2) ? 3 : 4;  // packed words per thread and k-tile
    const int bcol = ROWS ? (t & 127) : 8 * (t & 15);
    const int brow = ROWS ? 4 * (t >> 7) : 4 * (t >> 4);
    const int nB = n0 + bcol;
    const uint32_t nibmask = nib_mask_vgpr();
    uint32_t himask;
    asm volatile("v_mov_b32 %0, 0x00f000f0" : "=v"(himask));  // (held in a VGPR for the same reason as nibmask: one v_and_or_b32)
    const int zk = Pzk;
    const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)Pscales : (const uint32_t *)Pqzeros;
    // zero points: words per group row (zmul), this column's word inside the row (zoff), and -- strip-major storage (p.sm, row-stream
    // layouts only) -- the word offset of the column's strip (zstrip; group rows then hold the strip's 16 columns only: 2 or 8 words)
    const int Gn = (p.K + (1 << p.gs_shift) - 1) >> p.gs_shift;
    const bool sm = ROWS && p.sm;
    const int ncs = sm ? (nB & 15) : nB;  // column index inside a row of the scale / zero tables
    const int zmul_all = (zk == ZK_PACKED) ? (LAYOUT == 2 ? (PN * 3) >> 5 : (PN >> 3)) : (PN >> 1);  // words per group over all columns
    const int zmul = sm ? ((zk == ZK_PACKED) ? 2 : 8) : zmul_all;
    const int zoff = ((zk == ZK_PACKED) ? (LAYOUT == 2 ? (ncs * 3) >> 5 : (ncs >> 3)) : (ncs >> 1)) + (sm ? (nB >> 4) * Gn * zmul : 0);
    const int zoff2 = (LAYOUT == 2 && zk == ZK_PACKED && (sm ? ((ncs * 3) >> 5) == 0 : zoff + 1 < zmul)) ? 1 : 0;  // packed 3-bit zero points may straddle two words
    struct BSet {
      uint32_t w[4];
      half8_t s8;     // AWQ: the 8 columns' scales
      uint32_t sraw;  // GPTQ: the column's scale, raw 16 bits
      uint32_t z, z2;
    };
    BSet bset[2];
    // buffer loads: per-lane byte offsets are loop constants, the k-tile / group advance is a scalar offset (SALU only)
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)Pqweight, 0, (int)((size_t)p.K * PN / 2), 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void *)Pscales, 0, Gn * PN * 2, 0x00020000);
    // (strip-major: every strip has whole words of its own -- 3-bit zero points take 2 words per 16 columns, not 1.5)
    const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((void *)zbase, 0, (sm ? (PN >> 4) * Gn * zmul : Gn * zmul_all) * 4, 0x00020000);
    const int wrow_bytes = sm ? 64 : (ROWS ? PN * 4 : (PN >> 3) * 4);     // bytes per packed row (strip-major: the strip's 16 words)
    const int ktile_bytes = (LAYOUT == 0 ? 8 : (LAYOUT == 2 ? 6 : BK)) * wrow_bytes;  // packed rows per k-tile: 8 / 6 (3 bits) / 64 (AWQ)
    const int strip_rows = (LAYOUT == 2) ? (p.K * 3) >> 5 : (p.K >> 3);      // word rows of one strip
    const int voff_w = (LAYOUT == 2 ? 3 * (t >> 7) : brow) * wrow_bytes +
                       (sm ? (nB >> 4) * strip_rows * 64 + ncs * 4 : (ROWS ? nB * 4 : (nB >> 3) * 4));
    const int srow_bytes = sm ? 32 : PN * 2;                                // bytes per group row of the scale table
    const int voff_s = sm ? (nB >> 4) * Gn * 32 + ncs * 2 : nB * 2, voff_z = zoff * 4;
    const int krow0 = ROWS ? 8 * brow : brow;                               // this thread's first k inside a k-tile
    auto load_b = [&](int kt, BSet &bs) {
      const int ktc = KT0 + min(kt, KT - 1);
      const int so = ktc * ktile_bytes;
#pragma unroll
      for (int r = 0; r < WPT; ++r) bs.w[r] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, voff_w + r * wrow_bytes, so, 0);
      // one group per thread per k-tile (group_size >= 32); krow0 is per-lane only through brow (0 or 32 k for GPTQ)
      const int G0 = (ktc * BK) >> p.gs_shift, G1 = (ktc * BK + 32) >> p.gs_shift;
      const int G = ROWS ? ((krow0 >= 32) ? G1 : G0) : ((ktc * BK + krow0) >> p.gs_shift);
      if constexpr (ROWS)
        bs.sraw = __builtin_amdgcn_raw_buffer_load_b16(rs_s, voff_s + G * srow_bytes, 0, 0);
      else
        bs.s8 = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rs_s, voff_s + G * srow_bytes, 0, 0));
      bs.z = __builtin_amdgcn_raw_buffer_load_b32(rs_z, voff_z + G * zmul * 4, 0, 0);
      if constexpr (LAYOUT == 2) bs.z2 = __builtin_amdgcn_raw_buffer_load_b32(rs_z, voff_z + zoff2 * 4 + G * zmul * 4, 0, 0);
    };
    auto store_b = [&](int stage, const BSet &bs) {
      half_t *Bb = Bs + stage * kBTile;
      if constexpr (LAYOUT == 2) {
        // 32 k = 96 bits of the column's bit stream in w[0..2]: four 24-bit fields of 8 values each, natural k order
        const uint32_t zfield = (uint32_t)(((((uint64_t)bs.z2) << 32) | bs.z) >> ((3 * ncs) & 31));
        const half_t zp = (half_t)(float)((zfield + (uint32_t)p.add_zero_bias) & 7u);
        const half_t zf = __builtin_bit_cast(half_t, (uint16_t)((nB & 1) ? (bs.z >> 16) : (bs.z & 0xffffu)));
        const half_t sc = __builtin_bit_cast(half_t, (uint16_t)bs.sraw);
        const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)4.f));
        const uint32_t f[4] = {bs.w[0] & 0xffffffu, __builtin_amdgcn_alignbit(bs.w[1], bs.w[0], 24) & 0xffffffu,
                               __builtin_amdgcn_alignbit(bs.w[2], bs.w[1], 16) & 0xffffffu, bs.w[2] >> 8};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          half2_t b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t lo = (f[r] >> (6 * j)) & 7u, hi = (f[r] >> (6 * j + 3)) & 7u;
            b[j] = deq_pair(lo | (hi << 16) | kMagic, cc);
          }
          *(half8_t *)(Bb + tile_off(bcol, brow + r)) = half8_t{b[0].x, b[0].y, b[1].x, b[1].y, b[2].x, b[2].y, b[3].x, b[3].y};
        }
      } else if constexpr (LAYOUT == 0) {
        const half_t zp = (half_t)(float)(((bs.z >> (4 * (nB & 7))) + (uint32_t)p.add_zero_bias) & 15u);
        const half_t zf = __builtin_bit_cast(half_t, (uint16_t)((nB & 1) ? (bs.z >> 16) : (bs.z & 0xffffu)));
        const half_t sc = __builtin_bit_cast(half_t, (uint16_t)bs.sraw);
        if constexpr (BF) {
          // bf16 W: even / odd nibbles as bytes (k = 2 i / 2 i + 1 in byte i), q -> fp32, (q_a, q_b) s - z s in one packed fma, one
          // rounding to bf16; pairs come out in natural k order.  (The empty asm keeps hipcc from splitting the two masks back into
          // one shift + and per nibble.)
          const float sf = (float)sc, zf32 = (zk == ZK_PACKED) ? (float)zp : ((zk == ZK_F16) ? (float)zf : 8.f);
          const float2_t s2 = {sf, sf}, nzs2 = {-(zf32 * sf), -(zf32 * sf)};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            uint32_t e = bs.w[r] & 0x0f0f0f0fu, o = (bs.w[r] >> 4) & 0x0f0f0f0fu;
            asm volatile("" : "+v"(e), "+v"(o));
            uint32_t out[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float2_t q = {(float)((e >> (8 * i)) & 0xffu), (float)((o >> (8 * i)) & 0xffu)};
              out[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(__builtin_elementwise_fma(q, s2, nzs2), bf16x2_t));
            }
            *(uint4_t *)(Bb + tile_off(bcol, brow + r)) = uint4_t{out[0], out[1], out[2], out[3]};
          }
        } else {
          const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)8.f));
          // Round 6: the odd nibbles are taken where they sit (bits 4..7 of each half) under the pattern 0x5400 = 64.0 -- an fp16 in
          // [64, 128) has ulp 1/16, so mantissa bits 4..7 weigh exactly 1, 2, 4, 8: r = 64 + q -- and fma(r, s, -64 s) is q s rounded
          // once, like fma(1024 + q, s, -1024 s) for the even ones: the same three-rounding W bit for bit, one shift per word instead
          // of three (13 VALU per 8 weights instead of 15: the staging waves' issue slots are what the matrix waves wait for).
          const half2_t c64 = splat2((half_t)(-64.0f) * sc);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const uint32_t w0 = bs.w[r], w8 = bs.w[r] >> 8;
            const half2_t b0 = deq_pair(and_or(w0, nibmask, kMagic), cc);                                                   // (k0, k4)
            const half2_t b1 = __builtin_elementwise_fma(as_h2(and_or(w0, himask, 0x54005400u)), cc.s2, c64) - cc.zs2;     // (k1, k5)
            const half2_t b2 = deq_pair(and_or(w8, nibmask, kMagic), cc);                                                   // (k2, k6)
            const half2_t b3 = __builtin_elementwise_fma(as_h2(and_or(w8, himask, 0x54005400u)), cc.s2, c64) - cc.zs2;     // (k3, k7)
            *(half8_t *)(Bb + tile_off(bcol, brow + r)) = half8_t{b0.x, b1.x, b2.x, b3.x, b0.y, b1.y, b2.y, b3.y};
          }
        }
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

    // Order of one iteration: [write B tile kt+1 from its register set] barrier [request tile kt+3 into that set].  A set is
    // requested right after the barrier that frees it and consumed two barriers later (~1.5 k-tiles of flight), and the only
    // vmcnt wait in the loop is the counted one in front of the stores (the younger set's loads stay in flight).  Nothing is
    // conditional around a load (a branch there makes hipcc's counted vmcnt collapse to vmcnt(0)): past the last tile the
    // loads re-read it and the stores fill a stage nobody reads again.
    load_b(0, bset[0]);
    load_b(1, bset[1]);
    __builtin_amdgcn_sched_barrier(0);
    store_b(0, bset[0]);
    __builtin_amdgcn_sched_barrier(0);
    load_b(2, bset[0]);
    __syncthreads();  // prologue barrier: B stage 0 holds tile 0
    for (int kt = 0; kt + 1 < KT; kt += 2) {  // two k-tiles per trip: the register sets alternate by name, no branch between them
      store_b(1, bset[1]);
      __syncthreads();  // barrier #kt
      load_b(kt + 3, bset[1]);
      __builtin_amdgcn_sched_barrier(0);
      store_b(0, bset[0]);
      __syncthreads();  // barrier #kt+1
      load_b(kt + 4, bset[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KT & 1) __syncthreads();  // odd number of k-tiles: barrier #KT-1 (the tile it would publish does not exist)
    __syncthreads();  // matches the matrix waves' barrier in front of their epilogue
    if (S > 1) {          // ... and the two around the split-K ticket
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // =================================================== matrix waves ===================================================
  const int wm = wave >> 1, wn = wave & 1;  // (MW/2) (M) x 2 (N): rows wm*WROWS.., columns wn*64..
  const int fr = lane & 31, fs = lane >> 5;  // fragment row (A: m, B: n) and k half of the 16-wide sub-step
  float16_t acc[AM][2];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  half8_t fa0[AM], fb0[2], fa1[AM], fb1[2];
  auto read_frags = [&](int sa, int sb, int ks, half8_t (&fa)[AM], half8_t (&fb)[2]) {
    const half_t *Ab = As + sa * kATile, *Bb = Bs + sb * kBTile;
#pragma unroll
    for (int b = 0; b < 2; ++b) fb[b] = *(const half8_t *)(Bb + tile_off(wn * 64 + b * 32 + fr, ks * 2 + fs));
#pragma unroll
    for (int a = 0; a < AM; ++a) fa[a] = *(const half8_t *)(Ab + tile_off(wm * WROWS + a * 32 + fr, ks * 2 + fs));
  };
  // ---- activation tile by LDS-DMA: this wave owns rows wave*64 .. +63 of the 256-row tile = 8 pieces of 8 rows x 128 B.
  // Piece q: lane l -> LDS row r = wave*64 + 8q + l/8, physical slot l%8, which holds logical 16-byte chunk (l%8) ^ ((r>>1)&7).
  // Per-lane byte offsets into x are loop constants; the k-tile advance (128 B) is the scalar offset.
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)min((size_t)p.M * p.K * 2, (size_t)0x7fffffff), 0x00020000);
  int voff_x[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int r = wave * (8 * NP) + 8 * q + (lane >> 3);
    const int grow = min(m0 + r, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
    voff_x[q] = grow * p.K * 2 + (((lane & 7) ^ lds_row_swizzle(r)) << 4);
  }
  // one DMA piece (8 rows x 128 B of this wave's 64 rows) of k-tile kt into ring slot `slot`
  // (the builtin's operands are first copied into plain locals: called with template-dependent expressions, the HOST pass of
  //  hipcc silently fails to instantiate the whole kernel -- no diagnostic, just an undefined __device_stub__ at load time)
  const int rows_per_wave = 8 * NP;
  auto dma_piece = [&](int kt, int slot, int q) {
    const int so = (KT0 + min(kt, KT - 1)) * (BK * 2);
    const int vo = voff_x[q];
    lds_void_t *dst = (lds_void_t *)(As + slot * kATile + (wave * rows_per_wave + q * 8) * BK);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
  };
  // half of a sub-step's MFMAs: row tiles [h * AM/2, (h+1) * AM/2)
  auto mfma_half = [&](const half8_t (&fa)[AM], const half8_t (&fb)[2], int h) {
#pragma unroll
    for (int a = h * (AM / 2); a < (h + 1) * (AM / 2); ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if constexpr (BF) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[a]), __builtin_bit_cast(bf16x8_t, fb[b]), acc[a][b], 0, 0, 0);
        else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
      }
  };

  if constexpr (PRIO) __builtin_amdgcn_s_setprio(2);  // the matrix wave outranks its SIMD's dequant wave for issue slots
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(0, 0, q);
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(1, 1, q);
  if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile 0's pieces have landed (tile 1's still in flight)
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();                      // prologue barrier (the dequant waves' __syncthreads)
#pragma unroll
  for (int q = 0; q < NP / 4; ++q) dma_piece(2, 2, q);
  read_frags(0, 0, 0, fa0, fb0);
  // The issue order is pinned (sched_barrier): left alone, hipcc sinks every fragment read to just above its first use
  // (fewest live registers) and the lone matrix wave of the SIMD then sits out each LDS round trip with an idle matrix pipe.
  // A DMA piece costs 60-180 issue cycles (MI355X_MICROARCH.md): one goes behind every four MFMAs (128 cycles of matrix-pipe
  // work), never eight in a row (first version: all eight right after the barrier -- the pipe ran dry behind them).
  // Between barrier #kt-1 and barrier #kt the wave requests the 8 pieces of tile kt+2 (slot (kt+2)%3, freed by barrier
  // #kt-1): pieces 0,1 beside sub-step 3 of tile kt-1, pieces 2..7 beside sub-steps 0..2 of tile kt.  At barrier #kt the
  // pieces of tile kt+1 are older than those 8: vmcnt(8) retires exactly them.
#define G3_SB() __builtin_amdgcn_sched_barrier(0)
  // NP == 4 (one piece per sub-step): the piece goes BETWEEN the two halves of the sub-step's MFMAs, not behind them (round 4,
  // profiles/logs/r04p_time_native.log: +1-2 % on all three shapes; alternating the position between the two matrix waves of a SIMD: none)
#define G3_PIECE_MID(kt_, slot_, q8, q4) { dma_piece(kt_, slot_, NP == 8 ? q8 : q4); G3_SB(); }
#define G3_PIECE_END(kt_, slot_, q8, q4) if constexpr (NP == 8) { dma_piece(kt_, slot_, q8); } G3_SB();
  int sa = 0;  // A ring slot of tile kt (kt % 3)
  for (int kt = 0; kt < KT; ++kt) {
    const int sb = kt & 1;
    const int sa1 = (sa == 2) ? 0 : sa + 1, sa2 = (sa == 0) ? 2 : sa - 1;  // slots of tiles kt+1, kt+2
    // pieces per sub-step: NP / 4 (2 or 1), one behind each half (NP = 8) or behind the first half (NP = 4) of its MFMAs;
    // piece indices: sub-step 3 of the previous tile took [0, NP/4), sub-steps 0..2 take the rest
    read_frags(sa, sb, 1, fa1, fb1);
    G3_SB();
    mfma_half(fa0, fb0, 0); G3_SB();
    G3_PIECE_MID(kt + 2, sa2, 2, 1)
    mfma_half(fa0, fb0, 1); G3_SB();
    G3_PIECE_END(kt + 2, sa2, 3, 1)  // sub-step 0
    read_frags(sa, sb, 2, fa0, fb0);
    G3_SB();
    mfma_half(fa1, fb1, 0); G3_SB();
    G3_PIECE_MID(kt + 2, sa2, 4, 2)
    mfma_half(fa1, fb1, 1); G3_SB();
    G3_PIECE_END(kt + 2, sa2, 5, 2)  // sub-step 1
    read_frags(sa, sb, 3, fa1, fb1);
    G3_SB();
    mfma_half(fa0, fb0, 0); G3_SB();
    G3_PIECE_MID(kt + 2, sa2, 6, 3)
    mfma_half(fa0, fb0, 1); G3_SB();
    G3_PIECE_END(kt + 2, sa2, 7, 3)  // sub-step 2
    // barrier #kt: my fragment reads of tile kt are complete (lgkmcnt(0)) and my DMA pieces of tile kt+1 have landed
    // (vmcnt(NP): only tile kt+2's are still in flight).  After it: B stage sb and A slot sa are free, tile kt+1 is complete.
    if constexpr (NP == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    G3_SB();
    read_frags(sa1, sb ^ 1, 0, fa0, fb0);  // past the last tile: stages nobody uses
    G3_SB();
    mfma_half(fa1, fb1, 0); G3_SB();
    G3_PIECE_MID(kt + 3, sa, 0, 0)
    mfma_half(fa1, fb1, 1); G3_SB();
    G3_PIECE_END(kt + 3, sa, 1, 0)  // sub-step 3
    sa = sa1;
  }
#undef G3_PIECE_MID
#undef G3_PIECE_END
#undef G3_SB
  __builtin_amdgcn_s_setprio(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // stray DMA pieces past the last tile: land before the LDS is reused
  __builtin_amdgcn_s_barrier();                                // (matched by the dequant waves' final barrier)

  // ---- split-K: publish the fp32 partial tile; the last block to arrive sums the S partials in fixed order (deterministic).
  // Write-through (sc1) stores, every storing wave drains them, one relaxed agent-scope ticket per block, the last arriver reads
  // with sc1 loads and re-arms the counter.  Slab element (tile, split, wave, register, lane): 256 contiguous bytes per instruction.
  if (S > 1) {
    int &s_ticket = *(int *)(smem + 24 * 1024);  // past the epilogue's wave-private regions (8 x 4.5 KB)
    constexpr int WREGS = AM * 2 * 16;
    float *slab = p.slabs + ((size_t)slab_tile * S + ksplit) * (size_t)(BM * BN) + (size_t)wave * (WREGS * 64) + lane;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) st_sc1(slab + ((a * 2 + b) * 16 + r) * 64, acc[a][b][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.counters + slab_tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != S - 1) return;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int sp = 0; sp < S; ++sp) {
      const float *src = p.slabs + ((size_t)slab_tile * S + sp) * (size_t)(BM * BN) + (size_t)wave * (WREGS * 64) + lane;
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += ld_sc1(src + ((a * 2 + b) * 16 + r) * 64);
    }
    if (tid == 0) __hip_atomic_store(p.counters + slab_tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- epilogue: + bias, round once, transpose through wave-private LDS, 16-byte row-contiguous stores ---------------------
  // C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).  (All fragment reads that
  // matter completed before the last barrier; the stray ones above only fill registers.)
  half_t *ep = smem + wave * (32 * 72);  // 32 rows x 64 cols, row stride 72 halves (144 B)
  float bv[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) bv[b] = Pbias ? (float)Pbias[n0 + wn * 64 + b * 32 + fr] : 0.f;
#pragma unroll
  for (int a = 0; a < AM; ++a) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fs;
        const float v = acc[a][b][r] + bv[b];
        // bf16 activations (x converted to fp16 by the pre-pass): the result is rounded to fp16 and then to bf16, as the
        // reference's shim does (fp16 kernel output .to(bfloat16), quant_linear_awq.py:29-36, 144-146)
        if constexpr (BF) ((uint16_t *)ep)[row * 72 + b * 32 + fr] = f32_to_bf16(v);  // (native bf16: one rounding of the fp32 sum)
        else if (p.out_bf16) ((uint16_t *)ep)[row * 72 + b * 32 + fr] = f32_to_bf16((float)(half_t)v);
        else ep[row * 72 + b * 32 + fr] = (half_t)v;
      }
    // 32 rows x 128 B = 256 chunks of 16 B: 4 per lane (wave-private region: no barrier, the wave's own LDS ops are ordered)
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int c = lane + 64 * h, row = c >> 3, ch = c & 7;
      const uint4_t v = *(const uint4_t *)(ep + row * 72 + ch * 8);
      const int m = m0 + wm * WROWS + a * 32 + row;
      if (m < p.M) *(uint4_t *)((half_t *)Py + (size_t)m * PN + n0 + wn * 64 + ch * 8) = v;
    }
  }
}

bool gemm3_ok(const GemmParams &p, int layout) {
  const int on = knob("QLLM_GEMM3", 1);
  // 4 bits: from M = 1024 (below it gemm2's split-K form was the measured choice; QLLM_GEMM3_MIN_M moves the line);
  // 3 bits: every prefill size -- the alternative there is the dequant kernel + a dense GEMM
  // round 3 (profiles/r03_mid_m.md, tools/lab/gbench): with split-K this kernel passes gemm2 from M = 384 on the 11008-wide shapes
  // (59 vs 64 us at M = 384 / 512, 110 vs 130 and 89 vs 93 at 768) and from 768 on 4096 x 4096 (41 vs 43; 34.5 vs 33 below)
  const int min_m_env = knob("QLLM_GEMM3_MIN_M", 0);
  const int min_m = min_m_env ? min_m_env : (((size_t)p.K * p.N > (size_t)4096 * 4096) ? 384 : 768);
  const int min_m3 = knob("QLLM_GEMM3_MIN_M_3BIT", 33);  // (33..64: native-layout layers whose strips stop at two row tiles)
  if (!on || p.g_idx || p.K % 64 != 0 || p.N % 128 != 0) return false;
  if (p.M < (layout == kGemm3Rows3Bit ? min_m3 : min_m)) return false;
  // fp16 activations only: the activation tile goes to LDS by DMA, which cannot convert bf16 on the way (callers convert x with
  // launch_bf16_to_f16 first and set out_bf16, or use gemm2); 32-bit byte offsets into x and the packed weights
  if (p.act_bf16 || (size_t)p.M * p.K * 2 >= 0x7fffffffull || (size_t)p.K * p.N / 2 >= 0x7fffffffull) return false;
  return p.group_size % 32 == 0 && p.gs_shift >= 5;
}

// bf16 activations served natively (no conversion pre-pass): the 4-bit row-stream / strip-major layouts (the modules' native copies)
bool gemm3_bf16_native(int layout) { return knob("QLLM_GEMM3_BF16", 1) && layout == QLLM_LAYOUT_GPTQ; }

__global__ __launch_bounds__(256) void bf16_to_f16_kernel(const uint4_t *__restrict__ src, uint4_t *__restrict__ dst, size_t n8) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n8) dst[i] = __builtin_bit_cast(uint4_t, bf16x8_to_h8(src[i]));
}

int launch_bf16_to_f16(const void *src, void *dst, size_t n, hipStream_t stream) {
  const size_t n8 = n / 8;  // callers pass M * K with K % 64 == 0
  hipLaunchKernelGGL(bf16_to_f16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (const uint4_t *)src, (uint4_t *)dst, n8);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

template <int LAYOUT, int MW, bool PRIO = true, bool BF = false>
static int launch_gemm3_b(const GemmParams &p, hipStream_t stream) {
  using namespace g3;
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)gemm3_kernel<LAYOUT, MW, PRIO, BF>)) return rc;
  const int tiles_all = p.n_prob > 1 ? p.total_tiles : ((p.M + BM - 1) / BM) * (p.N / BN);
  const int tiles = p.tail_split > 1 ? p.tail_from + (tiles_all - p.tail_from) * p.tail_split : tiles_all * p.split_k;
  const size_t lds = (size_t)(3 * kATile + 2 * kBTile) * sizeof(half_t);  // 128 KB
  hipLaunchKernelGGL((gemm3_kernel<LAYOUT, MW, PRIO, BF>), dim3(tiles), dim3((MW + 4) * 64), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

// split-K factor for gemm3: gemm2's rule (largest S <= 8 with tiles * S <= CUs and >= 8 k-tiles per block), reduced until every
// block owns the same number of k-tiles
int gemm3_split_k(int M, int N, int K) {
  int s = gemm2_split_k(M, N, K);
  while (s > 1 && (K / 64) % s != 0) s /= 2;
  return s;
}

// Round 6: K-split of the ragged last round.  More tiles than CUs and a last round that fills at most half of them (profiles/
// r06_shape_table.md: Llama-2-13B's 5120-wide layers are 320 tiles = 1.25 rounds and ran at 767 TFLOP/s where the 256-tile layers
// of Llama-2-7B run at 970-1070): the r = tiles mod CUs tiles of that round are shared by TS blocks each -- the largest power of two with
// r * TS <= CUs, whole k-tile counts and >= 16 k-tiles per block (the fix-up moves 2 x 128 KB per block: below that it eats the gain).
// Returns TS (1: none) and the number of unsplit tiles in *tail_from.
int gemm3_tail_split(int M, int N, int K, int *tail_from) { return gemm3_tail_split_tiles(((M + 255) / 256) * (N / 128), K, tail_from); }
int gemm3_tail_split_tiles(int tiles, int K, int *tail_from) {
  const int cus = compute_units(), kt = K / 64;
  *tail_from = tiles;
  if (!knob("QLLM_GEMM3_TAIL", 1) || tiles <= cus || tiles % cus == 0) return 1;
  const int r = tiles % cus;
  int ts = 1;
  while (ts < 8 && r * ts * 2 <= cus && kt % (ts * 2) == 0 && kt / (ts * 2) >= 16) ts *= 2;
  if (ts > 1) *tail_from = tiles - r;
  return ts;
}

int launch_gemm3(const GemmParams &p_in, int layout, hipStream_t stream) {
  GemmParams p = p_in;
  if (p.split_k < 1 || !p.slabs || !p.counters) p.split_k = 1;
  if (p.tail_split < 2 || !p.slabs || !p.counters || p.split_k != 1) p.tail_split = 0;
  const int raster = knob("QLLM_GEMM2_RASTER", 1);
  p.raster = raster;
#ifdef QLLM_LAB
  if (const int g4 = knob("QLLM_GEMM4", -1); g4 >= 0) return launch_gemm4(p, layout, g4, stream);  // (lab) tools/lab/gemm4.hip variants
  // (lab) tools/lab/gemm5.hip: every wave a matrix wave, B fragments dequantised in registers; QLLM_GEMM5 = waves along M
  if (const int g5 = knob("QLLM_GEMM5", 0); g5 > 0 && gemm5_ok(p, layout)) return launch_gemm5(p, g5, stream);
#endif
  const int mw = knob("QLLM_GEMM3_MW", 8);  // measured (profiles/r02_prefill_summary.md): 8 matrix waves 908 / 923 / 1004 TFLOP/s, 4: 873 / 915 / 1003
  const int prio = knob("QLLM_GEMM3_PRIO", 1);
  if (p.native_bf16) {
    if (layout != QLLM_LAYOUT_GPTQ) return set_error(QLLM_ERR_INVALID, "internal: native bf16 serves the row-stream / strip-major 4-bit layouts");
    return launch_gemm3_b<0, 8, true, true>(p, stream);
  }
  if (layout == kGemm3Rows3Bit) return launch_gemm3_b<2, 8>(p, stream);
  if (mw == 4 && !prio) return layout == QLLM_LAYOUT_AWQ_GEMM ? launch_gemm3_b<1, 4, false>(p, stream) : launch_gemm3_b<0, 4, false>(p, stream);
  if (mw == 8) return layout == QLLM_LAYOUT_AWQ_GEMM ? launch_gemm3_b<1, 8>(p, stream) : launch_gemm3_b<0, 8>(p, stream);
  return layout == QLLM_LAYOUT_AWQ_GEMM ? launch_gemm3_b<1, 4>(p, stream) : launch_gemm3_b<0, 4>(p, stream);
}

}  // namespace qllm
