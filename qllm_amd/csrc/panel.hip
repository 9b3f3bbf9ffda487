// Mid-batch "panel" kernel (round 4): y[M, N] = x[M, K] . dequant(W) (4 bits, or 3) for 17 <= M <= 128 (round 5: layers of up to 4096 x 4096 from 33 rows)
// on the strip-major native layout; one layer, or up to 8 layers sharing x (q/k/v, gate/up) in one launch.
//
// Why: between the strip kernels (M <= 32: every block re-reads ALL of x, which grows with M) and the 256-row prefill tiles there
// was a hole -- gemm2 / gemm3 run 27-40 us per Llama-2-7B linear from M = 33 to M = 256 whatever M is, because their B tiles go
// words -> registers -> fp16 dequant -> ds_write -> barrier -> ds_read, a pipeline that tops out near 1.4 TB/s of packed weights
// chip-wide, while the strips stream 3-4.7 TB/s by building B fragments in registers.  This kernel keeps the strips' B side and
// shares the A side:
//   * block = a PANEL of 64 columns (wave w: the 16-column strip 4 panel + w) over a K range: all of K, or one of S splits when
//     the panels alone do not cover the CUs; up to 64 rows the block has EIGHT waves -- waves 4..7 run the same four strips over
//     the second half of the block's K range with A buffers of their own, and the two halves are summed through LDS (half the
//     cross-block splits for the same waves in flight); 16 MT rows (MT = 1, 2, 4 or 8 row tiles of v_mfma_f32_16x16x32_f16);
//   * A: K-tiles of 8 k-steps go into LDS ONCE per block by LDS-DMA (1 KB pieces of 8 rows x 128 B, the
//     XOR-swizzled [k-pair][row tile][16 rows][128 B] image of strip_dma.hpp, a quarter of the pieces per wave), double-buffered:
//     tile t+1 is requested right after the barrier that publishes tile t; all four waves read the same fragments (ds_read_b128,
//     the reads of k-step s+1 issued before the MFMAs of k-step s: the order is pinned);
//   * B: the wave's packed words of tile t+1 (one dword per lane and k-step, 256 contiguous bytes per instruction) and the scale /
//     zero words of its groups are requested at the same point into the other register set; a fragment is the raw (1024 + q |
//     64 + q) fp16 pattern MINUS (bias + z) -- one v_pk_add_f16 per pair on top of the strips' shift + 4 v_and_or: the exact
//     integers q - z, so there is no bias or zero-point correction (no sum-of-x bookkeeping MFMAs) and a group ends with one
//     fp32 fma per accumulator: y += s * sum x (q - z), the strips' unrounded-W contract.  fp16 zero points (HQQ): q first
//     (exact), then q - z (one fp16 rounding, as the reference's own dequant);
//   * split-K: fp32 partial panels through write-through slabs + one ticket per panel, summed by the last arriver in split order
//     (the protocol of gemm2.hip / skinny.hip: deterministic; ~6 us when it is needed: profiles/r04_mid_m.md);
//   * epilogue through LDS: 128-byte row-contiguous stores.
// K % 64 == 0, N % 64 == 0; 4 bits with group size 32 / 64 / 128, or the 3-bit stream with 64 / 128 (up to 64 rows: patterns minus
// 1024 + f z, times 1 / f -- exact q - z as well).  Replaces, for these batch sizes, the dequantise-then-matmul forward
// of /root/reference/qllm/modeling/q_layers/quant_linear_gptq.py:81-85 (and quant_linear_hqq.py, quant_linear_awq.py through the
// native copy).
#include "kernels.hpp"

namespace qllm {

namespace {

constexpr int kPanelWaves = 4;
constexpr int kPanelKTS = 8;  // k-steps per K-tile
typedef __attribute__((address_space(3))) void lds_void_t;

// KH: K halves per block (1 or 2).  KH = 2: eight waves, waves 4..7 run the same four strips over the second half of the block's K
// range with A buffers of their own; the halves are summed through LDS before the epilogue -- half the global splits for the same
// number of waves in flight (the cross-block sum costs ~6 us when it is needed at all: profiles/r04_mid_m.md).
template <int MT, int CPL, int KH, int SPG, int BITS, bool BF16, bool ZF16>
__global__ __launch_bounds__(kPanelWaves * KH * 64, 1) void panel_kernel(const PanelParams p) {
  constexpr int NW = kPanelWaves;             // waves of one K half = strips of the panel / CPL
  constexpr int KTS = kPanelKTS;              // (MT = 4: 64 KB of A buffers -> two blocks per CU; MT = 8: 128 KB)
  constexpr int KP = KTS / 2;                 // k-pairs per tile
  constexpr int NGT = KTS / SPG;              // groups per tile
  constexpr int TILE_BYTES = KP * MT * 2048;  // one A buffer: [k-pair][row tile][16 rows][128 B]
  constexpr int PPW = KP * MT * 2 / NW;       // 1 KB DMA pieces per wave and tile
  constexpr int NMT = MT >= 2 ? MT / 2 : 1;   // distinct row tiles among a wave's pieces
  constexpr int PCOLS = 16 * CPL * NW;        // columns of a panel
  constexpr int EPS = PCOLS + 8;              // epilogue row stride in halves (16-byte aligned, bank-spread)
  static_assert(KTS % SPG == 0 && PPW * NW == KP * MT * 2, "tile geometry");
  static_assert(BITS == 4 || BITS == 3, "4 bits, or the 3-bit stream of the strips (32 k = 3 words per column)");
  constexpr int WR = (BITS == 4) ? 4 : 3;     // word rows per k-step
  constexpr bool Z2 = BITS == 3 && !ZF16;     // packed 3-bit zero points: the field may straddle into a second word
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // A[KH][2][TILE_BYTES]; the epilogue re-uses it

  const int lane = threadIdx.x & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave_all & (NW - 1), kh = wave_all / NW;  // strip of the panel, K half
  const int g = lane >> 4, i = lane & 15;
  const int S = p.split_k;
  const int ksplit = (int)blockIdx.x % S, gpanel = (int)blockIdx.x / S;  // (panel index over all layers of the launch)
  int pi = 0;
  if (p.n_prob > 1) {
#pragma unroll
    for (int q = 1; q < kMaxProblems; ++q)
      if (q < p.n_prob && gpanel >= p.prob[q].panel_begin) pi = q;
  }
  const PanelProblem pr = p.prob[pi];
  const int panel = gpanel - pr.panel_begin;
  const int M = p.M, N = pr.N, T = p.K >> 5;
  // K parts: S splits x KH halves; this wave's k-steps [t0, t1): whole groups and whole pairs.  Every wave of the block runs the same
  // number of tiles (the barriers are block-wide); a part that ends early multiplies zeros.
  constexpr int ALIGN = SPG < 2 ? 2 : SPG;
  const int chunk = ((T + S * KH - 1) / (S * KH) + ALIGN - 1) / ALIGN * ALIGN;
  const int t0 = (ksplit * KH + kh) * chunk, t1 = min(t0 + chunk, T);
  const int tiles = (chunk + KTS - 1) / KTS;
#ifdef QLLM_LAB
  const int abl = p.abl;  // timing-only ablations (QLLM_PANEL_ABL): 1 no activation pieces, 2 no word loads, 4 no compute, 8 no split-K sum
#else
  constexpr int abl = 0;
#endif

  // ---- addressing: raw buffer loads, per-lane byte offset (loop constant) + wave-uniform scalar offset ---------------------
  const int strip0 = (panel * NW + wave) * CPL;  // this wave's first 16-column strip
  const int strip_bytes = T * WR * 64;
  const int gtab = p.n_groups, Gmax = gtab - 1;
  const int zk = pr.zero_kind;
  const int zgroup = (zk == ZK_PACKED) ? 8 : 32;  // zero-point bytes per (strip, group); symmetric layers re-read their scales
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)pr.qweight, 0, (int)min((size_t)(N >> 4) * strip_bytes, (size_t)0x7fffffff), 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void *)pr.scales, 0, (N >> 4) * gtab * 32, 0x00020000);
  const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((zk == ZK_SYM) ? (void *)pr.scales : (void *)pr.qzeros, 0, (N >> 4) * gtab * zgroup, 0x00020000);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)min((size_t)M * p.K * 2, (size_t)0x7fffffff), 0x00020000);
  const int lane_w = (g * 16 + i) * 4, lane_s = i * 2;
  // 3 bits: lane group g needs stream bits [24 g, 24 g + 24) of the k-step's 96 = words {0,0,1,2}[g] and {0,1,2,2}[g], funnel-shifted
  const int lane_w3_lo = ((g == 0 ? 0 : g - 1) * 16 + i) * 4, lane_w3_hi = ((g == 3 ? 2 : g) * 16 + i) * 4;
  const uint32_t shift3 = (uint32_t)((32 - 8 * g) & 31);
  // zero points: the word holding this lane's column -- packed 4-bit: nibble i%8 of word i/8; packed 3-bit: bit 3i of the 64-bit pair
  // (the field may straddle into the second word); fp16: half i%2 of word i/2
  const int zoff = (zk == ZK_PACKED) ? (BITS == 3 ? (i * 3) >> 5 : (i >> 3)) : ((zk == ZK_F16) ? (i >> 1) : 0);
  const int lane_z = zoff * 4;
  const int lane_z2 = (BITS == 3 && zk == ZK_PACKED && zoff == 0) ? 4 : lane_z;
  const uint32_t zsh = (zk == ZK_PACKED) ? (uint32_t)((BITS == 3 ? 3 * i : 4 * i) & 31) : (uint32_t)(16 * (i & 1));
  // A pieces of this wave: q = wave + NW r  ->  half h = wave & 1, row tile (q >> 1) % MT, k-pair (q >> 1) / MT
  const int ph = wave & 1;
  int a_voff[NMT];
#pragma unroll
  for (int u = 0; u < NMT; ++u) {
    const int mt = (MT == 1) ? 0 : ((wave >> 1) + 2 * u) % MT;
    const int r = 8 * ph + (lane >> 3);
    a_voff[u] = min(16 * mt + r, M - 1) * p.K * 2 + (((lane & 7) ^ lds_row_swizzle(r)) << 4);
  }
  int a_rd[2];  // fragment read of k-step parity e: logical chunk 4e + g of row i
#pragma unroll
  for (int e = 0; e < 2; ++e) a_rd[e] = i * 128 + (((4 * e + g) ^ lds_row_swizzle(i)) << 4);

  const uint32_t mask_lo = nib_mask_vgpr(), mask_hi = mask_lo << 4;
  constexpr uint32_t kMagic64 = 0x54005400u;  // (64.0h, 64.0h): a nibble at bits 4-7 of an fp16 in [64, 128) weighs exactly 1

  // 3-bit field masks, derived from the opaque VGPR so hipcc can fuse each (x & m) | magic into one v_and_or_b32 (strip_kernel.hpp)
  const uint32_t m3a = ((mask_lo & 0x7u) << 1) | (mask_lo & 0x00070000u);  // 0x0007000E
  const uint32_t m3b = m3a << 3, m3c = m3a << 6;
  const uint32_t m3d = mask_lo & 0x00000007u, m3e = mask_lo & 0x00070000u;

  uint32_t w[2][KTS][CPL], w_hi[2][BITS == 3 ? KTS : 1][CPL], zr[2][NGT][CPL], zr2[2][Z2 ? NGT : 1][CPL];
  half_t sc[2][NGT][CPL];
  float4_t yacc[MT][CPL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < CPL; ++c) yacc[mt][c] = float4_t{0.f, 0.f, 0.f, 0.f};

  // tile kt: activation pieces -> buffer kt & 1 (k-pairs past the split's range: out of the buffer's range -> zeros, no traffic), the
  // scale / zero words of its groups and its packed words -> register set `set` (addresses clamped into the strip)
  auto request = [&](const int kt, const int set) __attribute__((always_inline)) {
    const int tb = t0 + kt * KTS;
    uint8_t *dstb = smem + (kh * 2 + (kt & 1)) * TILE_BYTES;
    if (!((abl & 1) && kt > 1))
#pragma unroll
    for (int r = 0; r < PPW; ++r) {
      const int q = wave + NW * r;                     // (wave-uniform)
      const int kp = (r * NW / 2 + (wave >> 1)) / MT;  // == (q >> 1) / MT
      const int u = (r * NW / 2 / 2) % NMT;            // index into a_voff: ((q >> 1) % MT - (wave >> 1)) / 2
      const bool live = tb + 2 * kp < t1;
      const int vo = live ? a_voff[u] : 0x7ffffff0;
      const int so = (tb + 2 * kp) * 64;               // byte offset of the k-pair inside a row
      lds_void_t *dst = (lds_void_t *)(dstb + q * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
    }
    if ((abl & 2) && kt > 1) return;
#pragma unroll
    for (int j = 0; j < NGT; ++j) {
      const int G = min((tb + j * SPG) / SPG, Gmax);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int sg = (strip0 + c) * gtab + G;
        sc[set][j][c] = __builtin_bit_cast(half_t, __builtin_amdgcn_raw_buffer_load_b16(rs_s, lane_s, sg * 32, 2));
        zr[set][j][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, lane_z, sg * zgroup, 2);
        if constexpr (Z2) zr2[set][j][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_z, lane_z2, sg * zgroup, 2);
      }
    }
#pragma unroll
    for (int s = 0; s < KTS; ++s)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int so = (strip0 + c) * strip_bytes + min(tb + s, T - 1) * (WR * 64);
        if constexpr (BITS == 4) {
          w[set][s][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, lane_w, so, 2);
        } else {
          w[set][s][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, lane_w3_lo, so, 2);
          w_hi[set][s][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, lane_w3_hi, so, 2);
        }
      }
  };

  // One K-tile.  The issue order is pinned (sched_barrier): left alone, hipcc sinks every fragment read to just above its first use
  // and every MFMA then sits out an LDS round trip.  Fragments of k-step s+1 are read before the MFMAs of k-step s are issued.
  auto compute = [&](const int buf, const int set) __attribute__((always_inline)) {
    const uint8_t *ab = smem + (kh * 2 + buf) * TILE_BYTES;
    float4_t gacc[MT][CPL];
    uint4_t ar[2][MT];
    // minus (bias + f z) of the current group for the four slot pairs (ZF16: minus z).  4 bits: pairs 0 / 2 carry bias 1024, pairs
    // 1 / 3 bias 64, f = 1; 3 bits: bias 1024 everywhere, slot factors (2,1 | 16,8 | 128,64 | 1,1) on slots (k0,k5 | k1,k6 | k2,k7 | k3,k4)
    half2_t nz[CPL][4];
    auto read_a = [&](const int s, uint4_t (&dst)[MT]) __attribute__((always_inline)) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) dst[mt] = *(const uint4_t *)(ab + ((s >> 1) * MT + mt) * 2048 + a_rd[s & 1]);
    };
    read_a(0, ar[0]);
#pragma unroll
    for (int s = 0; s < KTS; ++s) {
      const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
      if (s + 1 < KTS) read_a(s + 1, ar[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int j = s / SPG;
      if (s % SPG == 0) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          uint32_t field;
          if constexpr (Z2) field = (uint32_t)(((((uint64_t)zr2[set][j][c]) << 32) | zr[set][j][c]) >> zsh);
          else field = zr[set][j][c] >> zsh;
          if constexpr (ZF16) {
            const half_t z = __builtin_bit_cast(half_t, (uint16_t)field);
            nz[c][0] = nz[c][1] = nz[c][2] = nz[c][3] = splat2(-z);
          } else {
            const float zf = (zk == ZK_PACKED) ? (float)((field + (uint32_t)p.add_zero_bias) & (uint32_t)((1 << BITS) - 1)) : (float)(1 << (BITS - 1));
            if constexpr (BITS == 4) {
              nz[c][0] = nz[c][2] = splat2((half_t)(-1024.f - zf));  // exact: integers below 2048
              nz[c][1] = nz[c][3] = splat2((half_t)(-64.f - zf));
            } else {
              nz[c][0] = half2_t{(half_t)(-1024.f - 2.f * zf), (half_t)(-1024.f - zf)};
              nz[c][1] = half2_t{(half_t)(-1024.f - 16.f * zf), (half_t)(-1024.f - 8.f * zf)};
              nz[c][2] = half2_t{(half_t)(-1024.f - 128.f * zf), (half_t)(-1024.f - 64.f * zf)};
              nz[c][3] = splat2((half_t)(-1024.f - zf));
            }
          }
        }
      }
      half8_t av[MT];  // (bf16 activations: the tile was converted to fp16 in place when it landed -- convert_tile)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) av[mt] = __builtin_bit_cast(half8_t, ar[s & 1][mt]);
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        half2_t b0, b1, b2, b3;
        uint32_t c0, c1, c2, c3;
        if constexpr (BITS == 4) {
          const uint32_t wv = w[set][s][c], w8 = wv >> 8;
          b0 = as_h2((wv & mask_lo) | kMagic); b1 = as_h2((wv & mask_hi) | kMagic64);
          b2 = as_h2((w8 & mask_lo) | kMagic); b3 = as_h2((w8 & mask_hi) | kMagic64);
          if constexpr (ZF16) {
            const half2_t m1024 = splat2((half_t)-1024.f), m64 = splat2((half_t)-64.f);
            b0 = (b0 + m1024) + nz[c][0]; b1 = (b1 + m64) + nz[c][1]; b2 = (b2 + m1024) + nz[c][2]; b3 = (b3 + m64) + nz[c][3];
          } else {
            b0 = b0 + nz[c][0]; b1 = b1 + nz[c][1]; b2 = b2 + nz[c][2]; b3 = b3 + nz[c][3];
          }
          // registers hold (k0,k4) (k1,k5) (k2,k6) (k3,k7): four v_perm_b32 put them in natural order (k0,k1) (k2,k3) (k4,k5) (k6,k7):
          // the permutation is paid once per k-step and strip here, not once per row tile on the A fragments
          c0 = __builtin_amdgcn_perm(as_u32(b1), as_u32(b0), 0x05040100u); c2 = __builtin_amdgcn_perm(as_u32(b1), as_u32(b0), 0x07060302u);
          c1 = __builtin_amdgcn_perm(as_u32(b3), as_u32(b2), 0x05040100u); c3 = __builtin_amdgcn_perm(as_u32(b3), as_u32(b2), 0x07060302u);
        } else {
          // f: the lane's 8 three-bit values at bits 0,3,..,21 (strip_kernel.hpp): patterns 1024 + (2 q0, q5 | 16 q1, 8 q6 | 128 q2, 64 q7 |
          // q3, q4); minus 1024 + f z: exact f (q - z); times 1 / f (powers of two): exact q - z
          const uint32_t f = __builtin_amdgcn_alignbit(w_hi[set][s][c], w[set][s][c], shift3);
          const uint32_t f1 = f << 1;
          b0 = as_h2((f1 & m3a) | kMagic);
          b1 = as_h2((f1 & m3b) | kMagic);
          b2 = as_h2((f1 & m3c) | kMagic);
          const uint32_t lo34 = ((f >> 9) & m3d) | kMagic;
          b3 = as_h2(((f << 4) & m3e) | lo34);
          const half2_t r0 = {(half_t)0.5f, (half_t)1.f}, r1 = {(half_t)0.0625f, (half_t)0.125f}, r2 = {(half_t)0.0078125f, (half_t)0.015625f};
          if constexpr (ZF16) {
            const half2_t m1024 = splat2((half_t)-1024.f);
            b0 = (b0 + m1024) * r0 + nz[c][0]; b1 = (b1 + m1024) * r1 + nz[c][1]; b2 = (b2 + m1024) * r2 + nz[c][2]; b3 = (b3 + m1024) + nz[c][3];
          } else {
            b0 = (b0 + nz[c][0]) * r0; b1 = (b1 + nz[c][1]) * r1; b2 = (b2 + nz[c][2]) * r2; b3 = b3 + nz[c][3];
          }
          // slots (k0,k5) (k1,k6) (k2,k7) (k3,k4) -> natural order
          c0 = __builtin_amdgcn_perm(as_u32(b1), as_u32(b0), 0x05040100u); c1 = __builtin_amdgcn_perm(as_u32(b3), as_u32(b2), 0x05040100u);
          c2 = __builtin_amdgcn_perm(as_u32(b0), as_u32(b3), 0x07060302u); c3 = __builtin_amdgcn_perm(as_u32(b2), as_u32(b1), 0x07060302u);
        }
        const half8_t bf = __builtin_bit_cast(half8_t, uint4_t{c0, c1, c2, c3});
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) gacc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], bf, (s % SPG == 0) ? zero4 : gacc[mt][c], 0, 0, 0);
      }
      if (s % SPG == SPG - 1) {  // y += scale * sum x (q - z) of the group
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const float sf = (float)sc[set][j][c];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) yacc[mt][c][q] = __builtin_fmaf(sf, gacc[mt][c][q], yacc[mt][c][q]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // bf16 activations: the landed tile is converted to fp16 IN PLACE, once per block (each wave of the K half a quarter of it), instead
  // of once per fragment and wave in the loop (12 VALU x MT fragments x 8 k-steps per wave and tile; the reference's own bf16 path
  // casts x to fp16 as well: quant_linear_awq.py:29-36); one more barrier per tile
  auto convert_tile = [&](const int buf) __attribute__((always_inline)) {
    uint8_t *tb = smem + (kh * 2 + buf) * TILE_BYTES + wave * (TILE_BYTES / NW) + lane * 16;
#pragma unroll
    for (int r = 0; r < TILE_BYTES / NW / 1024; ++r) {
      const uint4_t v = *(const uint4_t *)(tb + r * 1024);
      *(half8_t *)(tb + r * 1024) = bf16x8_to_h8(v);
    }
  };
  // ---- main loop: [tile kt landed] barrier [request tile kt+1 into the buffer / register set tile kt-1 used] compute tile kt ----
  // (requesting the words two tiles ahead with a counted vmcnt, and a group's scale step deferred by a k-step, measured nothing:
  //  profiles/r04_mid_m.md)
  if (tiles > 0) request(0, 0);
  for (int kt = 0; kt < tiles; kt += 2) {  // the register sets alternate by name
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < tiles) request(kt + 1, 1);
    if constexpr (BF16) { convert_tile(0); __syncthreads(); }
    __builtin_amdgcn_sched_barrier(0);
    if (!(abl & 4)) compute(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 1 < tiles) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 2 < tiles) request(kt + 2, 0);
      if constexpr (BF16) { convert_tile(1); __syncthreads(); }
      __builtin_amdgcn_sched_barrier(0);
      if (!(abl & 4)) compute(1, 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // every wave is done with the A buffers: the epilogue re-uses them
  if constexpr (KH == 2) {  // the second K half hands its accumulators to the first through LDS
    float *hs = (float *)smem + (size_t)wave * (MT * CPL * 4 * 64) + lane;
    if (kh == 1) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) hs[((mt * CPL + c) * 4 + r) * 64] = yacc[mt][c][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) yacc[mt][c][r] += hs[((mt * CPL + c) * 4 + r) * 64];
    }
    __syncthreads();
  }

  // ---- split-K: fp32 partial panels through write-through slabs + one ticket per panel; the last arriver sums in split order ----
  if (S > 1 && !(abl & 8)) {
    int &s_ticket = *(int *)(smem + MT * 4096);  // (past the hand-off area of the K halves and the epilogue's staging rows; the A buffers are MT x 16 KB or more)
    constexpr int WREGS = MT * CPL * 4;
    float *slab = p.slabs + ((size_t)gpanel * S + ksplit) * (size_t)(NW * WREGS * 64) + (size_t)wave * (WREGS * 64) + lane;
    if (kh == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) st_sc1(slab + ((mt * CPL + c) * 4 + r) * 64, yacc[mt][c][r]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(p.counters + gpanel, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != S - 1) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int c = 0; c < CPL; ++c) yacc[mt][c] = float4_t{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < S && kh == 0; ++s) {
      const float *src = p.slabs + ((size_t)gpanel * S + s) * (size_t)(NW * WREGS * 64) + (size_t)wave * (WREGS * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) yacc[mt][c][r] += ld_sc1(src + ((mt * CPL + c) * 4 + r) * 64);
    }
    if (threadIdx.x == 0) __hip_atomic_store(p.counters + gpanel, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }

  // ---- epilogue: + bias, round once, [row][panel columns] through LDS, 16-byte row-contiguous stores -----------------------------
  uint16_t *ep = (uint16_t *)smem;
  const int n0 = panel * PCOLS;
  if (kh == 0)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int col = (wave * CPL + c) * 16 + i;
    const float bv = pr.bias ? (float)pr.bias[n0 + col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = yacc[mt][c][r] + bv;
        ep[(16 * mt + 4 * g + r) * EPS + col] = BF16 ? f32_to_bf16(v) : __builtin_bit_cast(uint16_t, (half_t)v);
      }
  }
  __syncthreads();
  constexpr int CPR = PCOLS / 8;  // 16-byte chunks per row
  for (int c = threadIdx.x; c < M * CPR; c += NW * KH * 64) {
    const int row = c / CPR, ch = c - row * CPR;
    *(uint4_t *)((uint16_t *)pr.y + (size_t)row * N + n0 + ch * 8) = *(const uint4_t *)(ep + row * EPS + ch * 8);
  }
}

template <int MT, int KH, int SPG, int BITS, bool BF16>
int launch_z(const PanelParams &p, int grid, hipStream_t stream) {
  const size_t lds = (size_t)KH * 2 * (kPanelKTS / 2) * MT * 2048;
  bool all_f16 = true;  // (the fp16-zero-point form: every layer of the launch)
  for (int i = 0; i < p.n_prob; ++i) all_f16 = all_f16 && p.prob[i].zero_kind == ZK_F16;
  if (all_f16) {
    static DeviceLatch done;
    if (int rc = lds_optin(done, (const void *)panel_kernel<MT, 1, KH, SPG, BITS, BF16, true>)) return rc;
    hipLaunchKernelGGL((panel_kernel<MT, 1, KH, SPG, BITS, BF16, true>), dim3(grid), dim3(kPanelWaves * KH * 64), lds, stream, p);
  } else {
    static DeviceLatch done;
    if (int rc = lds_optin(done, (const void *)panel_kernel<MT, 1, KH, SPG, BITS, BF16, false>)) return rc;
    hipLaunchKernelGGL((panel_kernel<MT, 1, KH, SPG, BITS, BF16, false>), dim3(grid), dim3(kPanelWaves * KH * 64), lds, stream, p);
  }
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

// (two strips per wave -- 128-column panels -- are written and measured: 4096 -> 11008 at M = 64 21.8-22.4 us against 19.7-21.1 with
//  64-column panels, 11008 -> 4096 24.2-24.7 against 21.4-22.2; not built)
template <int MT, int KH, bool BF16>
int launch_g(const PanelParams &p, int grid, hipStream_t stream) {
  if (p.bits == 3) {  // the 3-bit stream of the strips: 64- / 128-wide groups, up to 64 rows
    if constexpr (MT > 4) return set_error(QLLM_ERR_UNSUPPORTED, "internal: 3-bit panels are served up to 64 rows");
    else return p.group_size == 64 ? launch_z<MT, KH, 2, 3, BF16>(p, grid, stream) : launch_z<MT, KH, 4, 3, BF16>(p, grid, stream);
  }
  if (p.group_size == 32) {
    if constexpr (MT > 4) return set_error(QLLM_ERR_UNSUPPORTED, "internal: 32-wide groups are served up to 64 rows");  // (eight row tiles spill there)
    else return launch_z<MT, KH, 1, 4, BF16>(p, grid, stream);
  }
  if (p.group_size == 64) return launch_z<MT, KH, 2, 4, BF16>(p, grid, stream);
  return launch_z<MT, KH, 4, 4, BF16>(p, grid, stream);
}

}  // namespace

// Rows served: from 17 (two row tiles).  The one-row-tile form was round 4's route for 9..16 rows where K >= 2 N; round 5's strips beat
// it there (profiles/r05_batch16.md), so only the lab build (QLLM_PANEL_MIN_M) still instantiates and reaches it.
#ifdef QLLM_LAB
constexpr int kPanelMinRows = 2;
#else
constexpr int kPanelMinRows = 17;
#endif

// whole 64-column panels, whole k-step pairs, the group sizes of the strips (callers: native strip-major 4-bit layers, no g_idx)
bool panel_shape_ok(int M, int K, int N, int group_size, int bits) {
  if (bits == 3) return M >= kPanelMinRows && M <= 64 && K % 64 == 0 && N % 64 == 0 && (group_size == 64 || group_size == 128) && K % group_size == 0 &&
                        (double)K * N * 3 / 8 < 2147483648.0;
  return bits == 4 && M >= kPanelMinRows && M <= 128 && K % 64 == 0 && N % 64 == 0 && ((group_size == 32 && M <= 64) || group_size == 64 || group_size == 128) &&
         K % group_size == 0 && (double)K * N / 2 < 2147483648.0;
}

// row tiles of 16 the kernel is built for: 1, 2, 4, 8
int panel_mt(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 64 ? 4 : 8)); }
// K halves inside a block: two up to 64 rows (128 KB of A buffers), one above (eight row tiles: two halves would not fit)
int panel_kh(int M) { return M <= 64 ? 2 : 1; }
// blocks per panel along K: one block per CU; at least two K-tiles (16 k-steps) per K part (split x half); at most 8
int panel_split_k(int M, int n_panels, int K) {
  const int kh = panel_kh(M);
  int S = compute_units() / (n_panels > 0 ? n_panels : 1);
  S = S < 1 ? 1 : (S > 8 ? 8 : S);
  const int max_s = (K / 32) / 16 / kh;
  if (S > max_s) S = max_s < 1 ? 1 : max_s;
  return S;
}

size_t panel_slab_bytes(int M, int n_panels, int S) { return S > 1 ? (size_t)n_panels * S * kPanelWaves * (M <= 64 ? 4 : 8) * 256 * sizeof(float) : 0; }  // (sized for four row tiles up to 64 rows)

int launch_panel(const PanelParams &p, hipStream_t stream) {
  const int grid = p.n_panels * p.split_k;
#ifdef QLLM_LAB
  if (p.M <= 16) return p.act_bf16 ? launch_g<1, 2, true>(p, grid, stream) : launch_g<1, 2, false>(p, grid, stream);
#else
  if (p.M < kPanelMinRows) return set_error(QLLM_ERR_UNSUPPORTED, "internal: the panel kernel serves 17..128 rows");
#endif
  if (p.M <= 32) return p.act_bf16 ? launch_g<2, 2, true>(p, grid, stream) : launch_g<2, 2, false>(p, grid, stream);
  if (p.M <= 64) return p.act_bf16 ? launch_g<4, 2, true>(p, grid, stream) : launch_g<4, 2, false>(p, grid, stream);
  return p.act_bf16 ? launch_g<8, 1, true>(p, grid, stream) : launch_g<8, 1, false>(p, grid, stream);
}

}  // namespace qllm
