// Full-K strip kernel for small batches (M = 2..32) on the strip-major native layout: y[M, N] = x . dequant(W), no cross-block
// reduction.  Same decomposition as strip_kernel.hpp (block = NW waves = CPL adjacent 16-column strips for ALL of K; wave w owns a
// contiguous chunk of spw k-steps; raw biased fp16 B fragments, one fp32 scale / zero-point step per group; LDS reduction over the
// waves at the end), rebuilt around how the operands reach the matrix core.  What the timeline stamps (tools/lab/cbench --m 16
// --timeline, profiles/r03_batch16.md) showed about this regime:
//
//   * A CU INGESTS ABOUT 55 GB/s here (activations out of L2 + weights out of HBM, whatever the mix and the instruction), and every
//     block pulls the activations of its k range once per column block: with one strip per block that is 4x the weight bytes at
//     M = 16.  The time of a launch is (bytes through the busiest CU) / 55 GB/s plus about 1.5 us of prologue and 2.5 us of tail
//     (the waves of a block are served in order by the CU's memory pipe; the first ones wait at the reduction barrier).  Hence:
//     WIDE BLOCKS -- up to six strips share one activation stream -- chosen by the host so that the whole launch is ONE round of
//     blocks on the CUs (strip_plan: fewest bytes through the busiest CU); the last block of a layer may be ragged.
//   * ACTIVATIONS THROUGH LDS, IN FULL LINES.  A fragment-shaped load (lane (g, i) -> 16 B of row i) touches 16 rows x 64 B per
//     instruction: every 16-lane pass of the texture addresser sees 16 different cache lines.  Here a STAGE -- two k-steps, 16 rows
//     x 64 k = 2 KB per wave and row tile -- arrives as two LDS-DMA pieces of 8 rows x 128 B (a 16-lane pass reads two whole
//     lines), lands in a wave-private XOR-swizzled [16 rows][128 B] image (the source address carries the swizzle:
//     cdna_hip_programming.md rule 21) and is read back as A fragments with ds_read_b128.  No registers are held across the load
//     latency.  (Plain 16-byte loads + ds_write_b128 into the same image measured the same or 2-25 % slower.)
//   * A ROLLING RING OF FOUR STAGES.  The ring slot of a stage is re-requested (activation pieces, the stage's weight words into
//     the registers just consumed, the finished group's next scale / zero) as soon as its two k-steps have been issued to the matrix
//     core: three stages are always in flight behind the one being computed.  Every vector-memory operation of the loop is counted
//     by hand (raw buffer loads and DMA pieces: nothing hipcc may merge or drop), so each stage waits with an exact
//     `s_waitcnt vmcnt(total - own)`.  All addressing is per-lane constant + wave-uniform scalar offset: no vector address
//     arithmetic in the loop.
//   * THE BIAS LEAVES THROUGH THE MATRIX CORE.  B fragments are raw patterns 1024 + q / 64 + q (below); two bookkeeping MFMAs per
//     k-step (shared by the block's strips) against constant fragments give minus sum x bias and Sx = sum x in the accumulator
//     layout: the per-group VALU work of a strip is 6 packed operations + the decode of one scale and one zero point.
//   * k-steps past the wave's chunk or past K: the DMA source offset is pushed out of the buffer's range, the image holds zeros and
//     the stage contributes nothing (the weight / scale addresses are clamped into the strip and cost an L2 hit).
//   Round 5 (profiles/r05_batch16.md; the loop is bound by its vector-memory INSTRUCTIONS, timing-only ablations there):
//   * SCALE / ZERO TABLES IN FRONT OF THE RING.  The scale and zero words of the wave's whole chunk travel once, by LDS-DMA in 256-byte
//     instructions, into per-wave tables behind the rings ([group][strip] cells of 32 bytes) and are read back per group with
//     ds_read_u16 -- they were 96 of a wave's 208 vector-memory instructions and carry almost no bytes (q/k/v 11.6 -> 10.9 us).
//   * 64-WIDE GROUPS FINISH FROM minus-sum-x-bias ACCUMULATORS: both A fragments and the bookkeeping MFMAs of a slot first, the strips'
//     MFMAs then start from -sum x bias, so a group step is two fmas per accumulator.
//   * 3 BITS: ONE WORD LOAD PER FRAGMENT.  The lane's 24-bit field straddles two words for g = 1, 2; each lane loads word (0,1,2,2)[g]
//     and takes the lower word of its pair from lane - 16 with ds_bpermute (configs[3] 3-bit layer 54.6 -> 51.2 us).
//   * blocks of 1, 2, 3, 4 or 6 strips (3 bits: 6 only with fp16 zero points at 64-wide groups, where the ring fits the registers).
//   Lost: the 16-bytes-per-lane layout (tools/lab/attic/native_layout_v2.patch): 8-byte loads per ring slot touch eight lines per
//   instruction, -13..-25 % on the wide launches.
//   Measured dead ends (same file, profiles/r03_batch16.md): K split over adjacent blocks with fp32 partial slabs + arrival ticket
//   (fewer activation bytes per CU, but the fix-up costs 3-4 us: 4096 -> 4096 at M = 16 5.8 -> 10.6 us); two or four small blocks
//   per CU (no change: the CU's ingest rate is the bound, not the blocks' start-up).
//
// K must be a multiple of 64 and spw even (host: strip_plan).  Replaces gemv<half> (/root/reference/csrc/ort_cuda/dq_gemv.cu:41-150)
// and, for the HQQ configuration, the dequantise-then-matmul forward of /root/reference/qllm/modeling/q_layers/quant_linear_hqq.py.
#pragma once
#include <algorithm>

#include "kernels.hpp"

namespace qllm {

// Ring slots of an instantiation: four; three for six 4-bit strips, or four 3-bit strips with packed zero points, of 64-wide groups
// (four slots need more than 256 registers there).
// 32-wide groups carry a scale / zero pair per k-step and strip: three slots for one strip, two for two strips or two row tiles.
template <int CPL, int SPG, int BITS, bool ZF16, int MT = 1>
constexpr int strip_dma_ring() {
  if (SPG == 1) return (CPL >= 2 || MT >= 2) ? 2 : 3;
  return (SPG == 2 && (CPL >= 6 || (BITS == 3 && CPL >= 4 && !ZF16))) ? 3 : 4;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  // (vmcnt is six bits: a wave keeps at most 63 requests in flight, and a wait for "all but N > 63" is written as 63 -- it waits for a
  //  few requests more than needed, never fewer)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}

// NW: waves per block; CPL: adjacent 16-column strips per block (lane (g, i) holds column i of each); SPG: k-steps per group
// (group_size / 32: 1 (4 bits only), 2 or 4); BITS: 4 or 3; BF16: bf16 activations (converted after the fragment read); MT: 16-row tiles (M <= 16 MT);
// ZF16: every layer of the launch has fp16 zero points (HQQ: the native F16Z layout): the zero-point decode of a group is a shift and a
//       conversion instead of the branch-free five-operation form that also serves packed and symmetric zeros (a sixth of the
//       loop's VALU work at 64-wide groups).
// All byte offsets are 32-bit: the host sends layers of 2 GB and more of packed words to the register-A form (strip_plan).
template <int NW, int CPL, int SPG, int BITS, bool BF16, int MT, bool ZF16 = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void strip_dma_kernel(const StripParams p) {
  static_assert(SPG == 1 || SPG == 2 || SPG == 4, "groups of 32, 64 or 128");
  constexpr int NS = strip_dma_ring<CPL, SPG, BITS, ZF16, MT>();  // ring slots (stages of two k-steps)
  static_assert(NS >= 2 && NS <= 4 && (2 * NS) % SPG == 0, "a round of the ring is whole groups");
  constexpr int NG = 2 * NS / SPG;         // groups per round of the ring
  constexpr int TN = 16 * CPL;             // columns per block
  constexpr int WR = (BITS == 4) ? 4 : 3;  // word-rows per k-step
  constexpr int WL = 1;                    // loads per weight fragment (3 bits: the neighbour word comes from the lane 16 below, see b_frag)
  typedef __attribute__((address_space(3))) void lds_void_t;
  // vector-memory operations of one stage request, in issue order: [the scale / zero words of the group(s) that END in this slot,]
  // the DMA pieces, the weight words.  (SPG = 1: every slot holds two groups; SPG = 2: every slot ends one; SPG = 4: the odd ones.)
  constexpr bool Z2 = BITS == 3 && !ZF16;  // packed 3-bit zero points: the field may straddle into a second word
#ifdef QLLM_DMA_ABL  // (timing-only lab variants, tools/rounds5/g07: 1 = no scale / zero loads, 2 = no packed-word loads, 3 = neither)
  constexpr bool NO_SZ = QLLM_DMA_ABL & 1, NO_W = QLLM_DMA_ABL & 2;
#else
  constexpr bool NO_SZ = false, NO_W = false;
#endif
  // (round 5: the scale / zero tables of the wave's whole chunk travel ONCE, by LDS-DMA in 256-byte instructions in front of the ring --
  //  a 2-byte scale load and a 4-byte zero load per strip and group were a third to a half of the loop's vector-memory instructions,
  //  and the texture addresser charges per instruction: profiles/r05_batch16.md)
  constexpr int LX = 2 * MT + (NO_W ? 0 : 2 * CPL * WL);
  constexpr int L_EVEN = LX, L_ODD = LX;  // requests of an even / odd slot
  constexpr int L_ALL = (NS / 2) * (L_EVEN + L_ODD) + (NS % 2) * L_EVEN;
  extern __shared__ __attribute__((aligned(16))) float red[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, i = lane & 15;
  // diagnostics (qllm_debug_timeline): wave 0 of the first, the middle and the last block stamp [entry, ring requested, first stage
  // landed, rounds done, after the reduction barrier, exit] with the 100 MHz clock
  asm volatile("" ::"s"(p.dbg), "s"(p.n_prob), "s"(p.block_begin8[1]), "s"(p.block_begin8[2]), "s"(p.block_begin8[3]));  // (one batch of scalar loads)
  uint64_t *dbg_slot = nullptr;
  if (p.dbg && wave == 0) {
    if (blockIdx.x == 0) dbg_slot = p.dbg;
    else if (blockIdx.x == gridDim.x / 2) dbg_slot = p.dbg + 8;
    else if (blockIdx.x == gridDim.x - 1) dbg_slot = p.dbg + 16;
    if (dbg_slot && lane == 0) dbg_slot[0] = __builtin_amdgcn_s_memrealtime();
  }
  int pi = 0;
  if (p.n_prob > 1) {
#pragma unroll
    for (int q = 1; q < kMaxProblems; ++q)
      if (q < p.n_prob && (int)blockIdx.x >= p.block_begin8[q]) pi = q;
  }
  // (the whole problem record plus the launch scalars pulled in ONE batch of scalar loads: the empty asm "uses" them here, so hipcc
  //  cannot issue them one at a time at first use -- strip_kernel.hpp)
  const StripProblem pr = p.prob[pi];
  asm volatile("" ::"s"(pr.qweight), "s"(pr.scales), "s"(pr.qzeros), "s"(pr.bias), "s"(pr.y), "s"(pr.N), "s"(pr.block_begin), "s"(pr.zero_kind),
               "s"(p.x), "s"(p.M), "s"(p.K), "s"(p.T), "s"(p.spw), "s"(p.n_groups), "s"(p.add_zero_bias));
  const int b = blockIdx.x - pr.block_begin;
  const int N = pr.N, M = p.M, T = p.T;
  const int spw = p.spw;
  const int t0 = wave * spw;
  const int tend = min(t0 + spw, T);  // the wave owns k-steps [t0, tend): whole pairs (t0, spw, T are even)
  const int rounds = (spw + 2 * NS - 1) / (2 * NS);
  const int Gmax = p.n_groups - 1;

  // ---- addressing (strip-major: strip_kernel.hpp, SM).  Every load is a raw buffer load: per-lane byte offset (a loop constant) +
  // wave-uniform scalar offset.  The last block of a layer may hold fewer than CPL strips: the missing ones re-read the layer's last
  // strip and are not stored. --------------------------------------------------------------------------------------------------
  const int strip_bytes = T * WR * 64;  // one strip of packed words
  const int gtab = Gmax + 1;
  const int zk = pr.zero_kind;
  const int zmul = (zk == ZK_PACKED) ? 2 : 8;  // zero-point words per (strip, group)
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)pr.qweight, 0, (int)min((size_t)(N >> 4) * strip_bytes, (size_t)0x7fffffff), 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void *)pr.scales, 0, (N >> 4) * gtab * 32, 0x00020000);
  const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((zk == ZK_SYM) ? (void *)pr.scales : (void *)pr.qzeros, 0,
                                                      (zk == ZK_SYM) ? (N >> 4) * gtab * 32 : (N >> 4) * gtab * zmul * 4, 0x00020000);
  int w_strip[CPL], g_strip[CPL];  // (scalar registers) byte offset of the strip's words; first row of its scale / zero tables
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int strip = min(b * CPL + c, (N >> 4) - 1);
    w_strip[c] = strip * strip_bytes;
    g_strip[c] = strip * gtab;
  }
  const int lane_w = (g * 16 + i) * 4;
  // 3 bits: a column's k-step is 96 bits = words 0..2 of the strip row; lane (g, i) owns the 24 bits from bit 24 g: word 0 | words 0, 1 from
  // bit 24 | words 1, 2 from bit 16 | word 2 from bit 8.  Each lane LOADS one word -- (0, 1, 2, 2) by g -- and takes the lower word of
  // its pair from lane - 16 (g = 1, 2) through ds_bpermute (no memory access: the LDS crossbar): one vector-memory instruction per
  // fragment instead of two.  The loop is bound by its vector-memory instructions (profiles/r05_batch16.md).
  const int lane_w3 = ((g == 3 ? 2 : g) * 16 + i) * 4;
  const int bperm3 = ((g == 1 || g == 2) ? lane - 16 : lane) * 4;
  const uint32_t shift3 = (uint32_t)((32 - 8 * g) & 31);
  // zero points: the word holding this lane's column -- packed 4-bit: nibble i%8 of word i/8; packed 3-bit: bit 3i of the 64-bit
  // pair (the field may straddle into the second word); fp16: half i%2 of word i/2
  // (3 bits without the fp16-zero form, Z2: the lane reads the aligned 8-byte pair holding its field -- the packed row itself, or the
  //  four halves around its fp16 zero point)
  const int lane_z = (BITS == 3 && !ZF16) ? ((zk == ZK_F16) ? ((2 * i) & ~7) : 0)
                                          : ((zk == ZK_PACKED) ? 4 * (i >> 3) : ((zk == ZK_F16) ? 4 * (i >> 1) : 0));
  // decode, branch-free over the zero kind: field = word >> zsh; fp16 bits = ((field + add_zero_bias) & zmask) | zor; z = float(bits) + zadd
  //   packed: the integer v through the 1024 + v pattern (0x6400 | v), zadd = -1024; fp16: the half itself; symmetric: 2^(BITS-1)
  const uint32_t zsh = (BITS == 3 && !ZF16) ? ((zk == ZK_PACKED) ? (uint32_t)(3 * i) : ((zk == ZK_F16) ? (uint32_t)(16 * (i & 3)) : 0u))
                                            : ((zk == ZK_PACKED) ? (uint32_t)((4 * i) & 31) : (uint32_t)(16 * (i & 1)));
  const uint32_t zmask = (zk == ZK_PACKED) ? (uint32_t)((1 << BITS) - 1) : ((zk == ZK_F16) ? 0xffffu : 0u);
  const uint32_t zor = (zk == ZK_F16) ? 0u : ((zk == ZK_PACKED) ? 0x6400u : (0x6400u | (1u << (BITS - 1))));
  const uint32_t zbias = (zk == ZK_PACKED) ? (uint32_t)p.add_zero_bias : 0u;
  const float zadd = (zk == ZK_F16) ? 0.f : -1024.f;

  // ---- the activation ring: NS slots x MT row tiles x [16 rows][128 B] per wave (the reduction buffer re-uses the space) ----------
  uint8_t *xd = (uint8_t *)red + (size_t)wave * (NS * MT * 2048);
  // ... behind the rings (sized for four slots): per wave the scale table [ngw groups][CPL strips][32 B] and the zero-point table
  // ([..][32 B] fp16, [..][8 B] packed), each rounded up to whole 256-byte DMA instructions
  const int ngw = rounds * NG;                                             // groups the wave's rounds touch (dead stages read their rows too)
  const int G0w = t0 / SPG;                                                // the wave's first group (t0 is a multiple of SPG)
  const int sz_zoff = (ngw * CPL * 32 + 255) & ~255;                       // zero table behind the scale table
  uint8_t *szd = (uint8_t *)red + (size_t)NW * (4 * MT * 2048) + (size_t)wave * (2 * sz_zoff);
  const int x_bytes = (int)min((size_t)M * p.K * 2, (size_t)0x7fffffff);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, x_bytes, 0x00020000);
  // piece h of a row tile: lane l -> row 8h + l/8, physical 16-byte slot l%8 <- logical chunk (l%8) ^ swizzle(row)
  int xd_voff[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 8 * h + (lane >> 3);
      xd_voff[mt][h] = min(16 * mt + r, M - 1) * p.K * 2 + (((lane & 7) ^ lds_row_swizzle(r)) << 4);
    }
  // fragment read of k-step parity e: logical chunk 4e + g of row i
  int xd_rd[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) xd_rd[e] = i * 128 + (((4 * e + g) ^ lds_row_swizzle(i)) << 4);

  const uint32_t mask_lo = nib_mask_vgpr();  // 0x000f000f
  const uint32_t mask_hi = mask_lo << 4;
  const uint32_t m3a = ((mask_lo & 0x7u) << 1) | (mask_lo & 0x00070000u);  // 0x0007000E
  const uint32_t m3b = m3a << 3, m3c = m3a << 6;
  const uint32_t m3d = mask_lo & 0x00000007u, m3e = mask_lo & 0x00070000u;
  // B fragments are raw biased fp16 patterns (no per-weight arithmetic):
  //   4 bits: nibbles at bits 0-3 / 16-19 under 0x6400 = 1024 + q; nibbles at bits 4-7 / 20-23 under 0x5400 = 64 + q (the mantissa
  //           bit 4 of an fp16 in [64, 128) weighs exactly 1) -- one shift and four v_and_or per 8 weights, activations unscaled;
  //           fragment slot order (k0,k4 | k1,k5 | k2,k6 | k3,k7);
  //   3 bits: strip_kernel.hpp's patterns: 1024 + q scaled by (2,1 | 16,8 | 128,64 | 1,1) on slots (k0,k5 | k1,k6 | k2,k7 | k3,k4),
  //           the activations staged divided by the same factors.
  // b_nbias: minus the slot biases; b_sum: what turns the staged activations back into sum x (ones; 3 bits: the slot factors).
  constexpr uint32_t kMagic64 = 0x54005400u;  // (64.0h, 64.0h)
  const half8_t b_nbias = (BITS == 4) ? half8_t{(half_t)-1024.f, (half_t)-1024.f, (half_t)-64.f, (half_t)-64.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-64.f, (half_t)-64.f}
                                      : half8_t{(half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f, (half_t)-1024.f};
  const half8_t b_sum = (BITS == 4) ? half8_t{(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f}
                                    : half8_t{(half_t)2.f, (half_t)1.f, (half_t)16.f, (half_t)8.f, (half_t)128.f, (half_t)64.f, (half_t)1.f, (half_t)1.f};

  // ---- ring state: registers -----------------------------------------------------------------------------------------------
  uint32_t w[2 * NS][CPL];
  float4_t yacc[MT][CPL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < CPL; ++c) yacc[mt][c] = float4_t{0.f, 0.f, 0.f, 0.f};
  // the lane's read addresses in the tables for the current round's first group (advanced by a round of groups per round)
  int sz_rd_s = 2 * i;
  int sz_rd_z = sz_zoff + (ZF16 ? 2 * i : lane_z);
  const int sz_step = NG * CPL * 32;

  // request slot u for the round starting at k-step `base`
  auto request = [&](const int base, const int u) __attribute__((always_inline)) {
    const int kp = base + 2 * u;
    const bool live = kp < tend;      // wave-uniform
    const int kc = min(kp, T - 2);    // addresses stay inside the strip
    const int so = 64 * kc;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int vo = live ? xd_voff[mt][h] : 0x7ffffff0;  // dead stage: out of range -> zeros, no memory traffic
        lds_void_t *dst = (lds_void_t *)(xd + ((u * MT + mt) * 2 + h) * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
      }
    const int wrow = (WR * 64) * kc;
#pragma unroll
    for (int e = 0; e < (NO_W ? 0 : 2); ++e) {  // (the second k-step of the pair: + one k-step of words, an immediate offset)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        if constexpr (BITS == 4) {
          w[2 * u + e][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, lane_w + e * (WR * 64), w_strip[c] + wrow, 2);
        } else {
          w[2 * u + e][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, lane_w3 + e * (WR * 64), w_strip[c] + wrow, 2);
        }
      }
    }
  };

  float4_t gacc[MT][CPL], g_sx[MT], g_nb[MT];  // group accumulators: sum x (q + bias) per strip; sum x; minus sum x bias
  // the two k-steps of slot u (+ the group's scale / zero-point step when a group ends here)
  // B fragment of k-step s, strip c (raw biased patterns: see above)
  auto b_frag = [&](const int s, const int c) __attribute__((always_inline)) {
    half2_t b0, b1, b2, b3;
    if constexpr (BITS == 4) {
      const uint32_t wv = w[s][c], w8 = wv >> 8;
      b0 = as_h2((wv & mask_lo) | kMagic); b1 = as_h2((wv & mask_hi) | kMagic64);
      b2 = as_h2((w8 & mask_lo) | kMagic); b3 = as_h2((w8 & mask_hi) | kMagic64);
    } else {
      const uint32_t own = w[s][c];
      const uint32_t below = (uint32_t)__builtin_amdgcn_ds_bpermute(bperm3, (int)own);  // (g = 0, 3: the lane itself)
      const uint32_t f = __builtin_amdgcn_alignbit(own, below, shift3);
      const uint32_t f1 = f << 1;
      b0 = as_h2((f1 & m3a) | kMagic);
      b1 = as_h2((f1 & m3b) | kMagic);
      b2 = as_h2((f1 & m3c) | kMagic);
      const uint32_t lo34 = ((f >> 9) & m3d) | kMagic;
      b3 = as_h2(((f << 4) & m3e) | lo34);
    }
    return half8_t{b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
  };
  // A fragment of slot u, k-step parity e, row tile mt (4 bits: permuted; 3 bits: permuted and divided by the slot factors)
  auto a_frag = [&](const int u, const int e, const int mt) __attribute__((always_inline)) {
    const uint4_t xraw = *(const uint4_t *)(xd + (u * MT + mt) * 2048 + xd_rd[e]);
    half8_t xv;
    if constexpr (BF16) xv = bf16x8_to_h8(xraw); else xv = __builtin_bit_cast(half8_t, xraw);
    if constexpr (BITS == 4) {
      return a_perm_04152637(xv);
    } else {
      const half8_t pv = __builtin_shufflevector(xv, xv, 0, 5, 1, 6, 2, 7, 3, 4);
      const half2_t q0 = half2_t{pv[0], pv[1]} * half2_t{(half_t)0.5f, (half_t)1.f};
      const half2_t q1 = half2_t{pv[2], pv[3]} * half2_t{(half_t)0.0625f, (half_t)0.125f};
      const half2_t q2 = half2_t{pv[4], pv[5]} * half2_t{(half_t)0.0078125f, (half_t)0.015625f};
      return half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, pv[6], pv[7]};
    }
  };
  // scale and zero point of group j of the current round, strip c, as fp32: read from the wave's tables in LDS (cells of 32 bytes,
  // [group][strip]: 16 halves of scales; 16 halves of fp16 zero points, or the 2 packed words of the (group, strip) row in the cell's
  // first 8 bytes).  sz_rd_s / sz_rd_z: the lane's addresses in the round's first cell.
  auto group_consts = [&](const int j, const int c, float &zf, float &sfc) __attribute__((always_inline)) {
    if constexpr (NO_SZ) { zf = 3.5f; sfc = 0.01f; return; }
    sfc = (float)*(const half_t *)(szd + sz_rd_s + (j * CPL + c) * 32);
    if constexpr (ZF16) {
      zf = (float)*(const half_t *)(szd + sz_rd_z + (j * CPL + c) * 32);
    } else {
      uint32_t field;  // branch-free over the zero kind: an aligned word (pair) holding the lane's field, shifted down
      if constexpr (Z2) {
        const uint2_t zz = *(const uint2_t *)(szd + sz_rd_z + (j * CPL + c) * 32);
        field = (uint32_t)(((((uint64_t)zz.y) << 32) | zz.x) >> zsh);
      } else {
        field = *(const uint32_t *)(szd + sz_rd_z + (j * CPL + c) * 32) >> zsh;
      }
      const uint32_t zbits = ((field + zbias) & zmask) | zor;
      zf = (float)__builtin_bit_cast(half_t, (uint16_t)zbits) + zadd;
    }
  };
  // the two k-steps of slot u (+ the group's scale / zero-point step when a group ends here)
  auto compute = [&](const int u) __attribute__((always_inline)) {
    const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SPG == 2) {
      // 64-wide groups: the slot IS a group.  Both A fragments and the bookkeeping sums first; every strip's two MFMAs then START from
      // minus sum x bias, so the accumulator ends as sum x q and the group step is two packed fmas per accumulator pair
      // (round 5: was an add + two fmas, profiles/r05_batch16.md)
      half8_t av[2][MT];
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[e][mt] = a_frag(u, e, mt);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        g_sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt], b_sum, zero4, 0, 0, 0);
        g_nb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[0][mt], b_nbias, zero4, 0, 0, 0);
        g_sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[1][mt], b_sum, g_sx[mt], 0, 0, 0);
        g_nb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[1][mt], b_nbias, g_nb[mt], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          const half8_t bf = b_frag(2 * u + e, c);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            gacc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[e][mt], bf, e == 0 ? g_nb[mt] : gacc[mt][c], 0, 0, 0);
        }
      // y += scale * (sum x q  -  z * Sx)
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        float zf, sfc;
        group_consts(u, c, zf, sfc);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int q = 0; q < 4; ++q) yacc[mt][c][q] = __builtin_fmaf(sfc, __builtin_fmaf(-zf, g_sx[mt][q], gacc[mt][c][q]), yacc[mt][c][q]);
      }
      return;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int s = 2 * u + e;
      half8_t av[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        av[mt] = a_frag(u, e, mt);
        g_sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], b_sum, (s % SPG == 0) ? zero4 : g_sx[mt], 0, 0, 0);
        g_nb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], b_nbias, (s % SPG == 0) ? zero4 : g_nb[mt], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const half8_t bf = b_frag(s, c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          gacc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], bf, (s % SPG == 0) ? zero4 : gacc[mt][c], 0, 0, 0);
        }
      }
      if (s % SPG == SPG - 1) {
        // y += scale * (sum x (q + bias) - sum x bias  -  z * Sx)
        const int j = s / SPG;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          float zf, sfc;
          group_consts(j, c, zf, sfc);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float t = __builtin_fmaf(-zf, g_sx[mt][q], gacc[mt][c][q] + g_nb[mt][q]);
              yacc[mt][c][q] = __builtin_fmaf(sfc, t, yacc[mt][c][q]);
            }
        }
      }
    }
  };

  if constexpr (NO_W) {
#pragma unroll
    for (int s2 = 0; s2 < 2 * NS; ++s2)
#pragma unroll
      for (int c = 0; c < CPL; ++c) w[s2][c] = 0x12345678u * (lane + 1) + s2;
  }
  // ---- prologue: the scale / zero tables of the wave's chunk, then the whole ring; then rounds -------------------------------------
  if constexpr (!NO_SZ) {
    // 256 bytes per instruction: lane l carries 4 bytes of cell l / 8 = (group jj of the wave, strip c); the source row is clamped into
    // the strip's table (rows past K are only ever multiplied by zero activations, but must be finite)
    const int zrow = zmul * 4;  // bytes of a zero-point row in memory: 8 (packed: the cell's first two words) or 32 (fp16)
    const int bytes = ngw * CPL * 32;
    for (int off = 0; off < bytes; off += 256) {
      const int cell = (off >> 5) + (lane >> 3), wb = 4 * (lane & 7);
      const int jj = cell / CPL, c = cell - jj * CPL;
      int gs = g_strip[0];
#pragma unroll
      for (int q = 1; q < CPL; ++q) gs = (c == q) ? g_strip[q] : gs;
      const int row = gs + min(G0w + jj, Gmax);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lds_void_t *)(szd + off), 4, row * 32 + wb, 0, 0, 0);
      if (zk != ZK_SYM) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, (lds_void_t *)(szd + sz_zoff + off), 4, row * zrow + min(wb, zrow - 4), 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    request(t0, u);
    __builtin_amdgcn_sched_barrier(0);  // (the waits below count requests in THIS order)
  }
  if (dbg_slot) {
    if (lane == 0) dbg_slot[1] = __builtin_amdgcn_s_memrealtime();
    wait_vmcnt<L_ALL - L_EVEN>();
    if (lane == 0) dbg_slot[2] = __builtin_amdgcn_s_memrealtime();
  }
  for (int r = 0; r + 1 < rounds; ++r) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      // everything requested after this slot's last request: the other slots, once each
      if (u & 1) wait_vmcnt<L_ALL - L_ODD>();
      else wait_vmcnt<L_ALL - L_EVEN>();
      __builtin_amdgcn_sched_barrier(0);
      compute(u);
      __builtin_amdgcn_sched_barrier(0);
      request(t0 + 2 * NS * (r + 1), u);
      __builtin_amdgcn_sched_barrier(0);
    }
    sz_rd_s += sz_step;
    sz_rd_z += sz_step;
  }
  // last round: nothing is re-requested, the slots behind the one computed are the only ones still in flight
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    if (u == 0) wait_vmcnt<L_ALL - L_EVEN>();
    else if (u == 1) wait_vmcnt<L_ALL - L_EVEN - L_ODD>();
    else if (u == 2 && NS > 3) wait_vmcnt<L_ODD>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    compute(u);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (dbg_slot && lane == 0) dbg_slot[3] = __builtin_amdgcn_s_memrealtime();

  // ---- reduce the NW waves' partials through LDS: red[wave][row][col], over the rings once every wave is done with its own -------
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * g + r;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) red[(wave * M + row) * TN + c * 16 + i] = yacc[mt][c][r];
      }
    }
  __syncthreads();
  if (dbg_slot && lane == 0) dbg_slot[4] = __builtin_amdgcn_s_memrealtime();
  for (int e = threadIdx.x; e < M * TN; e += NW * 64) {
    const int row = e / TN, col = e - row * TN;
    const int nn = b * TN + col;
    if (nn >= N) continue;  // (ragged last block)
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) v += red[(wv * M + row) * TN + col];
    if (pr.bias) v += (float)pr.bias[nn];
    if (BF16)
      ((uint16_t *)pr.y)[(size_t)row * N + nn] = f32_to_bf16(v);
    else
      ((half_t *)pr.y)[(size_t)row * N + nn] = (half_t)v;
  }
  if (dbg_slot && lane == 0) dbg_slot[5] = __builtin_amdgcn_s_memrealtime();
}

// dynamic LDS of a launch: the waves' activation rings (8 KB per wave and row tile); the reduction buffer (nw x M x 16 cpl floats)
// re-uses them
// groups a wave's rounds touch, for the LDS budget: rounds x groups per round with the deepest ring the instantiation may use
inline int strip_dma_table_groups(int spw, int spg, int cpl, int mt) {
  if (spg == 1) {
    const int ns = (cpl >= 2 || mt >= 2) ? 2 : 3;
    return (spw + 2 * ns - 1) / (2 * ns) * (2 * ns);
  }
  const int r4 = (spw + 7) / 8 * 8 / spg, r3 = (spw + 5) / 6 * 6 / spg;  // (64-wide groups run rings of three or four slots)
  return spg == 2 ? (r3 > r4 ? r3 : r4) : r4;
}
inline size_t strip_dma_lds_bytes(int M, int nw, int cpl, int spw, int group_size) {
  const int mt = M > 16 ? 2 : 1, spg = group_size / 32, ngw = strip_dma_table_groups(spw, spg, cpl, mt);
  const size_t ring = (size_t)nw * 4 * mt * 2048  /* (sized for four slots whatever the ring) */, red = (size_t)nw * M * 16 * cpl * sizeof(float);
  const size_t tables = (size_t)nw * 2 * (((size_t)ngw * cpl * 32 + 255) & ~(size_t)255);  // (scales + zero points, fp16 zeros at most)
  return std::max(ring + tables, red);
}

template <int NW, int CPL, int SPG, int BITS, bool BF16, int MT, bool ZF16>
static int launch_strip_dma_z(const StripParams &p, int grid, hipStream_t stream) {
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)strip_dma_kernel<NW, CPL, SPG, BITS, BF16, MT, ZF16>)) return rc;
  hipLaunchKernelGGL((strip_dma_kernel<NW, CPL, SPG, BITS, BF16, MT, ZF16>), dim3(grid), dim3(NW * 64), strip_dma_lds_bytes(p.M, NW, CPL, p.spw, p.group_size), stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

template <int NW, int CPL, int SPG, int BITS, bool BF16, int MT>
static int launch_strip_dma_t(const StripParams &p, int grid, hipStream_t stream) {
  if constexpr (SPG == 2) {  // (the fp16-zero-point form is built for 64-wide groups, HQQ's default, where the group step weighs most)
    bool all_f16 = true;
    for (int i = 0; i < p.n_prob; ++i) all_f16 = all_f16 && p.prob[i].zero_kind == ZK_F16;
    if (all_f16) return launch_strip_dma_z<NW, CPL, SPG, BITS, BF16, MT, true>(p, grid, stream);
  }
  return launch_strip_dma_z<NW, CPL, SPG, BITS, BF16, MT, false>(p, grid, stream);
}

}  // namespace qllm
