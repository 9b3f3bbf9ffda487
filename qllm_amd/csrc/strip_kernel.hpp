// Full-K "strip" decode kernel: y[M<=64, N] = x . dequant(W), NO cross-block reduction.  Kernel template; the instantiations live
// in strip.hip (reference layouts read in place) and strip_sm.hip (the strip-major native layout).
//
// Why: at batch 1 a Llama-2-7B linear is 8.7-23 MB, i.e. 1-3 us of HBM time, the same order as ONE DRAM round trip under load.  A
// split-K kernel (skinny.hip) pays three more dependent round trips after its loads (slab write-through, arrival ticket, slab
// read-back).  Here every block owns a 16-column strip for ALL of K, so the dependency chain is: loads -> dequant+MFMA -> LDS
// reduce -> store.
//
//   * block = NW waves = one strip of 16*CPL columns; wave w owns a contiguous K chunk of spw k-steps (32 k each).
//   * a wave-load is 64 lanes x 4 B = 4 word-rows x 16 columns (lane (g,i): word-row r+g, column n0+i): exactly the B fragment of
//     v_mfma_f32_16x16x32_f16 for 32 consecutive k.  ALL of a wave's loads -- its activation chunk, <= 32 weight dwords per lane,
//     the scale/zero of every group it touches -- are issued back to back before the first wait: straight-line code.
//   * LAYOUTS.  Row-stream (the reference's GPTQ / HQQ state-dict buffers, [K/8][N] words): a strip is K/8 separate 64-byte
//     segments; the two halves of every 128-byte line belong to strips 2j and 2j+1, which the block->strip map places on the same
//     XCD (blocks b and b+8).  Strip-major (SM; the library's native layout, qllm_repack_native): [N/16][K/8][16] words, a strip
//     is ONE contiguous region and every wave-load is 256 contiguous bytes; scales / zero points are stored per strip as well.
//     Pure-read floor of one Llama-2-7B decoder layer's four launches (tools/lab/memlab2.hip, profiles/r03_memlab2.md): 25.96 us
//     in the round-2 row-stream forms, 21.45 us strip-major.
//   * up to 8 layers sharing x run as one launch (q/k/v, gate/up).
//
// Replaces gemv<half> (/root/reference/csrc/ort_cuda/dq_gemv.cu:41-150).
#pragma once
#include "kernels.hpp"

namespace qllm {

// v + (v from the lane selected by the DPP control): folds into one v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// NW: waves per block; CPL: columns per lane (1 -> 16-column strip; 4 -> 64-column strip, 256-byte row segments, 4 MFMAs per
// k-step; row-stream only); MAXS: k-steps (weight loads) per lane per round; SPG: k-steps per quantisation group (group_size /
// 32: 4, 2, or -- strip-major only -- 1); XL: 16-byte activation chunks staged per lane.
//
// Arithmetic (the kernel was VALU-issue-bound with a per-weight fp16 dequant -- SQ_ACTIVE_INST_VALU ~ 0.9 of the SIMD issue
// capacity -- so the per-weight work is cut to the bone):
//     y[m,n] = sum_G s[G,n] * ( sum_{k in G} x[m,k] q[k,n]  -  z[G,n] * sum_{k in G} x[m,k] )
//   * the B fragment is the raw "magic" fp16 pattern: 0x6400 | nibble = 1024+q for nibbles at bits 0-3/16-19 and
//     0x6400 | (nibble<<4) = 1024+16q for nibbles at bits 4-7/20-23 -- 1 shift + 4 v_and_or_b32 per 8 weights, no
//     per-weight fp16 math at all; the x16 on the odd k-slots is undone by staging x/16 in those A-fragment slots;
//   * per group the MFMA accumulator therefore holds 1024*Sx' + sum x q  (Sx' = sum of the staged A values); the
//     correction 1024*Sx' + z*Sx and the scale are applied once per group and column in fp32 (12 VALU per 32-128
//     weights), with Sx, Sx' per (row, group) computed once per wave when x is staged (v_dot2_f32_f16 + 3-4 DPP adds).
//   This evaluates x.W for the UNROUNDED W = s(q-z) in fp32: it differs from the reference's fp16-rounded W path by
//   the rounding noise of W (measured <= 3e-4 relative, tests bound it at 2e-3 against float64 of the reference's W).
// Everything is straight-line: loads are never predicated (addresses are clamped instead and the surplus is
// cancelled by zero activations), so hipcc keeps all of a wave's loads in flight and waits with counted vmcnt.
// BITS: 4, or 3 (bit-stream layout, 32 k = 3 words per column; CPL = 1 only).  For 3 bits the lane assembles its 24-bit
//       field (8 values) from two words with v_alignbit, and the magic patterns put the fields at different bit offsets
//       of the fp16 mantissa: slot scales (2,1 | 16,8 | 128,64 | 1,1) for k-slots (k0,k5 | k1,k6 | k2,k7 | k3,k4), undone by
//       staging x divided by the same factors -- 9 VALU per 8 weights instead of 5.
// RA ("register A", used for M > 4): no activation slab in LDS at all.  Staging costs M*K/8 chunk operations per BLOCK
//       (permute, Sx/Sx' dot products, LDS write) -- at M = 16 four times the main loop's work, repeated by every strip --
//       and the [M][K] slab (128 KB at M=16, K=4096) limits a CU to one block and K to ~5000.  Instead every lane loads
//       its own A fragment (16 B of row i, k-slots 8g..8g+7) straight from L2 for each k-step, permutes/scales it in
//       registers (4 v_perm + 4 v_pk_mul), and Sx / Sx' come out of the matrix core in exactly the accumulator layout
//       the correction needs: two more MFMAs per k-step against constant B fragments (all ones -> Sx' = sum of the
//       staged values; the per-slot multipliers -> Sx).  Rounds are 8 k-steps (8 x 16 B of x + 8 weight loads in flight
//       per lane); LDS holds only the cross-wave reduction buffer.  (Measured dead end: also requesting the NEXT round's
//       weights before computing a round -- two register sets, loop unrolled by two -- was 8-18 % slower at K=11008.)
// MT (RA only): 16-row MFMA tiles per block, M <= 16*MT.  Every B fragment built from a packed word is used MT times, so the
//       per-weight VALU work is amortised over up to 64 rows; each k-step holds MT x 16 B of activations per lane.
// SM: strip-major native layout: qweight [N/16][K*BITS/32][16] words, scales [N/16][K/g][16] halves, zero points
//       [N/16][K/g][2] words (packed 4-bit: nibble i%8 of word i/8; packed 3-bit: bit 3i of the 64-bit pair) or [N/16][K/g][16]
//       halves (fp16).  CPL = 1: one strip per block.  CPL = 4 (register-A form only): a block takes FOUR adjacent strips -- lane
//       (g, i) holds column i of each -- so that the activation fragments a lane loads per k-step (the register-A form's cost:
//       every block re-reads all of x from L2) are shared by 64 columns instead of 16; each wave-load is still 256 contiguous bytes.
// DBG: diagnostics instantiation: wave 0 of the first, the middle and the last block record 100 MHz timestamps
//       [entry, loads issued, x staged, rounds done, after the barrier, exit] into p.dbg (tools/lab/cbench --timeline).
// ONER (strip-major batch-1 slab form): the wave's chunk is exactly ONE round (spw == MAXS, checked by the host): the chunk
//       geometry folds to constants (about a fifth of the instructions in front of the first weight load).
template <int NW, int CPL, int MAXS, int SPG, int XL, int BITS, bool RA = false, bool RA_BF16 = false, int MT = 1, bool SM = false, bool DBG = false,
          bool ONER = false>
// (second launch-bound = minimum waves per SIMD: the 8-wave 64-column slab variant sits right at the 128-register edge
//  that lets two blocks share a CU -- 130 registers halve its occupancy: gate/up 13.7 -> 15.0 us; the strip-major 4- and 8-wave
//  forms are held to 128 registers; forcing 64 / 80 costs 7-55 spilled registers, the natural allocation is 80 / 126)
__global__ __launch_bounds__(NW * 64, (SM && !RA && NW <= 8) ? ((NW == 8 && MAXS == 16 && SPG == 4 && XL <= 2) ? 6 : 4) : ((NW == 8 && CPL == 4 && SPG == 4 && !RA) ? 4 : 1)) void strip_kernel(const StripParams p) {
  static_assert(BITS == 4 || (BITS == 3 && (CPL == 1 || SM)), "3-bit row-stream strips are 16 columns wide");
  static_assert(!RA || MAXS == 8, "register-A rounds are 8 k-steps");
  static_assert(MT == 1 || (RA && CPL == 1), "several row tiles: register-A, 16-column strips");
  static_assert(!SM || CPL == 1 || (RA && MT == 1), "strip-major blocks of several 16-column strips: register-A form, one row tile");
  static_assert(!ONER || (SM && !RA && XL <= 2), "one-round fold: strip-major batch-1 slab form");
  constexpr int NG = MAXS / SPG;   // groups per round (MAXS is a multiple of SPG; rounds start on a group boundary)
  constexpr int TN = 16 * CPL;     // columns per block
  constexpr int GL = 4 * SPG;      // lanes (16-byte chunks) per group in the staging pass: 4, 8 or 16
  constexpr int WR = (BITS == 4) ? 4 : 3;  // word-rows per k-step
  typedef uint32_t wvec_t __attribute__((ext_vector_type(CPL)));
  typedef float float2_t __attribute__((ext_vector_type(2)));
  // dynamic LDS: red[wave][M rows][TN cols] fp32 | per wave: activation chunk, M rows x (32*spw_pad) halves, row
  // stride + 16 B | per wave: (Sx, Sx') float2 per (group, row), 16 rows per group
  extern __shared__ __attribute__((aligned(16))) float red[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, i = lane & 15;
  uint64_t *dbg_slot = nullptr;
  if constexpr (DBG) {
    if (p.dbg && wave == 0) {
      if (blockIdx.x == 0) dbg_slot = p.dbg;
      else if (blockIdx.x == gridDim.x / 2) dbg_slot = p.dbg + 8;
      else if (blockIdx.x == gridDim.x - 1) dbg_slot = p.dbg + 16;
    }
    if (dbg_slot && lane == 0) dbg_slot[0] = __builtin_amdgcn_s_memrealtime();
  }

  int pi = 0;
  if (p.n_prob > 1) {  // (single-layer launches skip the search; q/k/v and gate/up take two or one comparisons, not seven)
    pi = ((int)blockIdx.x >= p.block_begin8[1]) ? 1 : 0;
    if (p.n_prob > 2) {
      pi = ((int)blockIdx.x >= p.block_begin8[2]) ? 2 : pi;
      if (p.n_prob > 3) {
#pragma unroll
        for (int q = 3; q < kMaxProblems; ++q)
          if (q < p.n_prob && (int)blockIdx.x >= p.block_begin8[q]) pi = q;
      }
    }
  }
  // ... and the whole problem record plus the remaining launch scalars pulled in ONE batch: the empty asm "uses" them here, so
  // hipcc must have issued every s_load before this point instead of one at a time at first use
  const StripProblem pr = p.prob[pi];
  asm volatile("" ::"s"(pr.qweight), "s"(pr.scales), "s"(pr.qzeros), "s"(pr.bias), "s"(pr.y), "s"(pr.N), "s"(pr.n_strips),
               "s"(pr.block_begin), "s"(pr.zero_kind), "s"(p.x), "s"(p.M), "s"(p.K), "s"(p.T), "s"(p.spw), "s"(p.group_size),
               "s"(p.add_zero_bias), "s"(p.act_bf16));

  int b = blockIdx.x - pr.block_begin;
  if (!SM && CPL == 1 && (pr.n_strips & 15) == 0) {
    // 64-byte segments: strips 2j and 2j+1 share every 128-byte line -> put them on blocks b and b+8 (same XCD)
    const int x = b & 7, r = b >> 3;
    b = (((r >> 1) << 3) + x) * 2 + (r & 1);
  }
  const int N = pr.N;
  const int n = b * TN + i * CPL;  // this lane's first column (N is a multiple of TN)
  // strip-major lds-slab launches with XL <= 2 are batch-1 launches (host: launch_sm_slab): M folds to a constant
  constexpr bool M1 = SM && !RA && XL <= 2;
  // strip-major lds-slab form: a round's MAXS k-steps are ONE window [tb, tb + MAXS) of the strip, tb = min(t0 + r MAXS, T - MAXS):
  // the window is shifted back into the matrix as a whole instead of clamping every load, so the loads of a round are
  // base + s * 256 bytes (immediate offsets, one address register) and the k-steps of the window the wave does not own are
  // cancelled by zero activations, like the padding always was.  Needs T >= MAXS (host: strip_plan).
  constexpr bool WIN = SM && !RA;
  const int M = M1 ? 1 : p.M;

  const int spw = ONER ? MAXS : p.spw;
  const int t0 = wave * spw;                      // spw is a multiple of SPG: every wave starts on a group boundary
  const int kend = min(32 * (t0 + spw), p.K);     // activations at k >= kend are staged as zero
  const int tend = min(t0 + spw, p.T);            // WIN: the wave owns k-steps [t0, tend)
  const int rounds = ONER ? 1 : (spw + MAXS - 1) / MAXS;
  const int spw_pad = rounds * MAXS;
  const int ngw = spw_pad / SPG;                  // groups in this wave's (padded) chunk

  // ---- 1. activations: this wave's [M][32*spw_pad] chunk -> registers now; LDS after the weight loads are issued ---
  const int xrow = spw_pad * 32 + 8;  // halves per staged row (16 B pad spreads rows over banks)
  half_t *xs = (half_t *)(red + NW * M * TN) + (size_t)wave * M * xrow;
  float2_t *sxs = (float2_t *)((half_t *)(red + NW * M * TN) + (size_t)NW * M * xrow) + (size_t)wave * ngw * 16;
  const int cpr = spw_pad * 4;  // 16-byte chunks per row
  const int xlast = M * cpr - 1;
  uint4_t xa[XL];
  bool xkeep[XL];
  int xdst[XL], sdst[XL];
  // chunk u of this lane: element offset in x, "k is inside the wave's chunk", LDS destination (-1: surplus lane), (Sx,Sx') slot
  auto x_index = [&](int u, uint32_t &off, bool &keep, int &dst, int &sd) {
    const int cu = lane + 64 * u;
    const int c = min(cu, xlast);  // surplus lanes re-read the last chunk and are masked out below
    const int row = (M == 1) ? 0 : c / cpr;
    const int kc = c - row * cpr;
    if constexpr (WIN) {
      const int r = kc / (MAXS * 4), wi = kc - r * (MAXS * 4);  // round, 16-byte chunk inside the round's window
      const int tb = min(t0 + r * MAXS, p.T - MAXS);
      const int t = tb + (wi >> 2);
      off = (uint32_t)(row * p.K + 32 * tb + 8 * wi);
      keep = (t >= t0 + r * MAXS) && (t < tend);
    } else {
      const int k = 32 * t0 + 8 * kc;
      off = (uint32_t)(row * p.K + min(k, p.K - 8));
      keep = (k < kend);
    }
    dst = (cu <= xlast) ? row * xrow + 8 * kc : -1;
    sd = (kc / GL) * 16 + row;
  };
  if constexpr (!RA) {
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      uint32_t off;
      x_index(u, off, xkeep[u], xdst[u], sdst[u]);
      // raw 16 bytes now (fp16 or bf16: same size); bf16 is converted when the chunk is staged -- converting here put a
      // vmcnt(0) between this load and every load after it
      xa[u] = *(const uint4_t *)((const uint16_t *)p.x + off);
    }
  }
  // lanes whose MFMA row is >= M read a valid row: their products only reach output rows that are never stored
  const half_t *xlane = xs + min(i, M - 1) * xrow + 8 * g;

  auto stage_x = [&]() {
    // the staging arithmetic is loop-invariant and speculatable: LLVM hoists it out of the round loop, i.e. ABOVE the loads of
    // round 0, and then waits for x (s_waitcnt vmcnt(1)) before a single weight load has been issued.  Making the raw chunks
    // opaque here pins every use of them below the point where stage_x() is called.
#pragma unroll
    for (int u = 0; u < XL; ++u) asm volatile("" : "+v"(xa[u]));
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      // fragment slot order and per-slot divisors (see the B-fragment construction below):
      //   4 bits: (k0,k4 | k1,k5 | k2,k6 | k3,k7), divisors (1,1 | 16,16 | 1,1 | 16,16)
      //   3 bits: (k0,k5 | k1,k6 | k2,k7 | k3,k4), divisors (2,1 | 16,8 | 128,64 | 1,1)
      half2_t p0, p1, p2, p3, q0, q1, q2, q3;
      const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      half8_t xv = p.act_bf16 ? bf16x8_to_h8(xa[u]) : __builtin_bit_cast(half8_t, xa[u]);
      xv = xkeep[u] ? xv : zero8;
      if constexpr (BITS == 4) {
        const half8_t pv = a_perm_04152637(xv);
        p0 = half2_t{pv[0], pv[1]}; p1 = half2_t{pv[2], pv[3]}; p2 = half2_t{pv[4], pv[5]}; p3 = half2_t{pv[6], pv[7]};
        const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
        q0 = p0; q1 = p1 * sixteenth; q2 = p2; q3 = p3 * sixteenth;
      } else {
        const half8_t pv = __builtin_shufflevector(xv, xv, 0, 5, 1, 6, 2, 7, 3, 4);
        p0 = half2_t{pv[0], pv[1]}; p1 = half2_t{pv[2], pv[3]}; p2 = half2_t{pv[4], pv[5]}; p3 = half2_t{pv[6], pv[7]};
        q0 = p0 * half2_t{(half_t)0.5f, (half_t)1.f};
        q1 = p1 * half2_t{(half_t)0.0625f, (half_t)0.125f};
        q2 = p2 * half2_t{(half_t)0.0078125f, (half_t)0.015625f};
        q3 = p3;
      }
      const half2_t one = {(half_t)1.f, (half_t)1.f};
      float sx = __builtin_amdgcn_fdot2(p0, one, 0.f, false);
      sx = __builtin_amdgcn_fdot2(p1, one, sx, false);
      sx = __builtin_amdgcn_fdot2(p2, one, sx, false);
      sx = __builtin_amdgcn_fdot2(p3, one, sx, false);
      float sxp = __builtin_amdgcn_fdot2(q0, one, 0.f, false);
      sxp = __builtin_amdgcn_fdot2(q1, one, sxp, false);
      sxp = __builtin_amdgcn_fdot2(q2, one, sxp, false);
      sxp = __builtin_amdgcn_fdot2(q3, one, sxp, false);
      // sum over the GL (8 or 16) lanes of the group with DPP adds: xor 1, xor 2 (quad_perm), then row_half_mirror and
      // row_mirror (reversals are as good as xor once the quads are uniform) -- one VALU op per step instead of
      // __shfl_xor's address VALU + ds_bpermute round trip (16 LDS ops per chunk at g128)
      sx = dpp_add<0xB1>(sx); sxp = dpp_add<0xB1>(sxp);
      sx = dpp_add<0x4E>(sx); sxp = dpp_add<0x4E>(sxp);
      if constexpr (GL >= 8) { sx = dpp_add<0x141>(sx); sxp = dpp_add<0x141>(sxp); }  // (GL = 4, 32-wide groups: the quad is the group)
      if constexpr (GL == 16) { sx = dpp_add<0x140>(sx); sxp = dpp_add<0x140>(sxp); }
      if (xdst[u] >= 0) {
        *(half8_t *)(xs + xdst[u]) = half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
        if ((lane & (GL - 1)) == 0) sxs[sdst[u]] = float2_t{sx, sxp};
      }
    }
  };
  if (!RA && XL > 2) stage_x();

  float4_t yacc[MT][CPL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < CPL; ++c) yacc[mt][c] = float4_t{0.f, 0.f, 0.f, 0.f};
  const uint32_t mask_lo = nib_mask_vgpr();  // 0x000f000f
  const uint32_t mask_hi = mask_lo << 4;     // 0x00f000f0
  // 3-bit field masks, derived from the opaque VGPR so hipcc can fuse each (x & m) | magic into one v_and_or_b32
  const uint32_t m3a = ((mask_lo & 0x7u) << 1) | (mask_lo & 0x00070000u);                          // 0x0007000E
  const uint32_t m3b = m3a << 3;                                                                 // 0x00380070
  const uint32_t m3c = m3a << 6;                                                                 // 0x01C00380
  const uint32_t m3d = mask_lo & 0x00000007u;                                                    // 0x00000007
  const uint32_t m3e = mask_lo & 0x00070000u;                                                    // 0x00070000
  const int Gmax = SM ? p.n_groups - 1 : (p.K - 1) / p.group_size;
  const int tmax = p.T - 1;
  // packed words: row stride WS (words) and this lane's column offset inside a row.  Strip-major: the strip is a [rows][16]
  // matrix of its own at word offset b * rows * 16
  const int WS = SM ? 16 : N;
  const size_t strip_words = (size_t)p.T * WR * 16;  // strip-major: words of one strip
  const uint32_t *qw = SM ? pr.qweight + (size_t)b * CPL * strip_words : pr.qweight;
  const int wcol = SM ? i : n;
  const uint32_t lane_off = (uint32_t)(g * WS + wcol);  // word offset of this lane inside a 4-row group
  // 3-bit: word rows {0,0,1,2}[g] / {0,1,2,2}[g] of the 3-row group, and the funnel shift {0,24,16,8}[g]
  const uint32_t lane_off3_lo = (uint32_t)((g == 0 ? 0 : g - 1) * WS + wcol);
  const uint32_t lane_off3_hi = (uint32_t)((g == 3 ? 2 : g) * WS + wcol);
  const uint32_t shift3 = (uint32_t)((32 - 8 * g) & 31);
  // scales: row stride (halves) and this lane's pointer to group 0
  const int SS = SM ? 16 : N;
  const half_t *scp = SM ? pr.scales + (size_t)b * CPL * (size_t)(Gmax + 1) * 16 + i : pr.scales + n;
  // zero points, branch-free addressing: packed -> word (G, n/8) (CPL | 8: one word holds the lane's columns);
  // fp16 -> the dword(s) holding halves (G, n..n+CPL-1); symmetric -> any valid dword (ignored)
  const int zk = pr.zero_kind;
  const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)pr.scales : (const uint32_t *)pr.qzeros;
  // (3-bit packed: column n sits at bit 3n of the group's row of N*3/32 words and may straddle two of them: the lane keeps the
  //  word holding its first bit and the next one -- clamped to the row, where nothing straddles -- and funnel-shifts)
  const int zcol = SM ? i : n;  // column index inside the zero-point row (strip-major rows hold the strip's 16 columns only)
  int zmul, zoff, zoff2;
  if constexpr (SM) {
    zmul = (zk == ZK_PACKED) ? 2 : 8;
    zoff = (zk == ZK_PACKED) ? (BITS == 3 ? (i * 3) >> 5 : (i >> 3)) : (i >> 1);
    zoff2 = (BITS == 3 && zk == ZK_PACKED && zoff == 0) ? 1 : 0;
    zbase += (size_t)b * CPL * (size_t)(Gmax + 1) * zmul;
  } else {
    zmul = (zk == ZK_PACKED) ? (BITS == 3 ? (N * 3) >> 5 : (N >> 3)) : (N >> 1);
    zoff = (zk == ZK_PACKED) ? (BITS == 3 ? (n * 3) >> 5 : (n >> 3)) : (n >> 1);
    zoff2 = (BITS == 3) ? ((zk == ZK_PACKED && zoff + 1 < zmul) ? 1 : 0) : ((zk == ZK_F16 && CPL == 4) ? 1 : 0);
  }
  const uint32_t zsel_p = (zk == ZK_PACKED) ? 0xffffffffu : 0u, zsel_h = (zk == ZK_F16) ? 0xffffffffu : 0u;
  const uint32_t zsel_s = (zk == ZK_SYM) ? __builtin_bit_cast(uint32_t, (float)(1 << (BITS - 1))) : 0u;
  // RA: this lane's activation row (MFMA row i; rows >= M re-read row M-1, their outputs are never stored), k-slot 8g
  const uint16_t *xrow_ra[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) xrow_ra[mt] = (const uint16_t *)p.x + (size_t)min(16 * mt + i, M - 1) * p.K + 8 * g;
  // constant B fragments: all ones, and the per-slot multipliers that undo the staged divisors (4 bits: x16 on the odd
  // pairs; 3 bits: 2,1 | 16,8 | 128,64 | 1,1)
  const half8_t b_ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  const half8_t b_mult = (BITS == 4) ? half8_t{(half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f, (half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f}
                                     : half8_t{(half_t)2.f, (half_t)1.f, (half_t)16.f, (half_t)8.f, (half_t)128.f, (half_t)64.f, (half_t)1.f, (half_t)1.f};

  auto round_body = [&](const int r) __attribute__((always_inline)) {
    const int base = WIN ? min(t0 + r * MAXS, p.T - MAXS) : t0 + r * MAXS;
    // ---- 2. scale / zero of every group this round touches: RAW loads only (tiny; issued first) ----------------------
    const int G0 = base / SPG;
    half_t sc[NG][CPL];
    uint32_t zraw[NG][(SM && CPL > 2) ? CPL : 2];  // strip-major blocks of several strips: one zero-point word per strip
    uint32_t zraw2[(SM && CPL > 1 && BITS == 3) ? NG : 1][CPL];  // ... and, 3 bits packed, the word the field may straddle into
    if constexpr (WIN) {
      // wave-uniform group base + the lane's fixed offset: the loads of the round differ by immediate offsets only
      const half_t *sl = scp + (size_t)G0 * 16;
      const uint32_t *zl = zbase + (size_t)G0 * zmul + zoff;
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        sc[j][0] = sl[j * 16];
        zraw[j][0] = zl[j * zmul];
        zraw[j][1] = (BITS == 3) ? zl[j * zmul + zoff2] : 0u;
      }
    } else {
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int G = min(G0 + j, Gmax);
      if constexpr (SM && CPL > 1) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          sc[j][c] = scp[((size_t)c * (Gmax + 1) + G) * 16];
          zraw[j][c] = zbase[((size_t)c * (Gmax + 1) + G) * zmul + zoff];
          if constexpr (BITS == 3) zraw2[j][c] = zbase[((size_t)c * (Gmax + 1) + G) * zmul + zoff + zoff2];
        }
      } else {
      if constexpr (CPL == 4) {
        const half4_t sv = *(const half4_t *)(scp + (size_t)G * SS);
        sc[j][0] = sv.x; sc[j][1] = sv.y; sc[j][2] = sv.z; sc[j][3] = sv.w;
      } else if constexpr (CPL == 2) {
        const half2_t sv = *(const half2_t *)(scp + (size_t)G * SS);
        sc[j][0] = sv.x; sc[j][1] = sv.y;
      } else {
        sc[j][0] = scp[(size_t)G * SS];
      }
      zraw[j][0] = zbase[(size_t)G * zmul + zoff];
      zraw[j][1] = (CPL == 4 || BITS == 3) ? zbase[(size_t)G * zmul + zoff + zoff2] : 0u;
      }
    }
    }
    // ---- 2b. RA: this round's activation fragments, raw (16 B per k-step; L2-resident, so they land before the weights)
    uint4_t xq[RA ? MAXS : 1][MT];
    if constexpr (RA) {
#pragma unroll
      for (int s = 0; s < MAXS; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xq[s][mt] = *(const uint4_t *)(xrow_ra[mt] + 32 * min(base + s, tmax));
    }
    // ---- 3. every weight load of this round: exactly MAXS loads, rows clamped into the matrix;
    //         address = wave-uniform row base (SALU) + one per-lane 32-bit offset -------------------------------------
    wvec_t w[MAXS];
    wvec_t w_hi[BITS == 3 ? MAXS : 1];
    if constexpr (WIN) {
      const uint32_t *wl = qw + (size_t)base * (WR * 16);  // wave-uniform
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if constexpr (BITS == 4) {
          w[s][0] = __builtin_nontemporal_load(wl + lane_off + s * (WR * 16));
        } else {
          w[s][0] = __builtin_nontemporal_load(wl + lane_off3_lo + s * (WR * 16));
          w_hi[s][0] = __builtin_nontemporal_load(wl + lane_off3_hi + s * (WR * 16));
        }
      }
    } else {
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      const uint32_t *rowp = qw + (size_t)(WR * min(base + s, tmax)) * WS;
      if constexpr (SM && CPL > 1) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if constexpr (BITS == 4) {
            w[s][c] = __builtin_nontemporal_load(rowp + c * strip_words + lane_off);
          } else {
            w[s][c] = __builtin_nontemporal_load(rowp + c * strip_words + lane_off3_lo);
            w_hi[s][c] = __builtin_nontemporal_load(rowp + c * strip_words + lane_off3_hi);
          }
        }
      } else if constexpr (BITS == 4) {
        w[s] = __builtin_nontemporal_load((const wvec_t *)(rowp + lane_off));
      } else {
        // 32 k = 3 words: lane group g needs stream bits [24g, 24g+24) = words {0,0,1,2}[g] and {0,1,2,2}[g]
        w[s][0] = __builtin_nontemporal_load(rowp + lane_off3_lo);
        w_hi[s][0] = __builtin_nontemporal_load(rowp + lane_off3_hi);
      }
    }
    }

    // RA: pin the issue order -- without this hipcc sinks half of the activation loads below the first MFMAs and waits
    // for them with vmcnt(0).  WIN: without it hipcc hoists the first instructions of the activation staging (the bf16
    // conversion) ABOVE the scale / weight loads and waits for x (s_waitcnt vmcnt(1)) before a single weight load has left
    if constexpr (RA || WIN) __builtin_amdgcn_sched_barrier(0);
    if constexpr (DBG) {
      if (r == 0 && dbg_slot && lane == 0) dbg_slot[1] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- 4. first round: activations -> LDS (needs only the OLDEST loads; the weights stay in flight).  For many rows
    //         (XL > 2) the chunk was staged before the weight loads instead, to keep its registers out of this region.
    if (!RA && XL <= 2 && r == 0) stage_x();
    if constexpr (DBG) {
      if (r == 0 && dbg_slot && lane == 0) dbg_slot[2] = __builtin_amdgcn_s_memrealtime();
    }

    // ---- 5. straight-line: raw-magic B fragments -> MFMA; one fp32 correction per group ----------------------------------
    const half_t *xr = xlane + 32 * (r * MAXS);
    const float2_t *sxr = sxs + (size_t)(r * NG) * 16 + 4 * g;  // (Sx, Sx') of rows 4g..4g+3
    float4_t gacc[MT][CPL];
    float4_t g_ones[MT], g_sx[MT];  // RA: 1024-offset sum and plain sum of x, per group
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      half8_t av[MT];
      if constexpr (RA) {
        // k-steps past this wave's chunk (padding of the last round) or past K contribute nothing: zero multipliers
        const bool valid = (r * MAXS + s < spw) && (base + s <= tmax);
        const half_t one = valid ? (half_t)1.f : (half_t)0.f;
        const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          half8_t xv;  // bf16 activations: compile-time variant (a runtime branch per k-step would split the straight-line body)
          if constexpr (RA_BF16) xv = bf16x8_to_h8(xq[s][mt]); else xv = __builtin_bit_cast(half8_t, xq[s][mt]);
          if constexpr (BITS == 4) {
            const half8_t pv = a_perm_04152637(xv);
            const half_t sixteenth = valid ? (half_t)0.0625f : (half_t)0.f;
            const half2_t q0 = half2_t{pv[0], pv[1]} * half2_t{one, one}, q1 = half2_t{pv[2], pv[3]} * half2_t{sixteenth, sixteenth};
            const half2_t q2 = half2_t{pv[4], pv[5]} * half2_t{one, one}, q3 = half2_t{pv[6], pv[7]} * half2_t{sixteenth, sixteenth};
            av[mt] = half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
          } else {
            const half8_t pv = __builtin_shufflevector(xv, xv, 0, 5, 1, 6, 2, 7, 3, 4);
            const half2_t q0 = half2_t{pv[0], pv[1]} * (half2_t{(half_t)0.5f, (half_t)1.f} * half2_t{one, one});
            const half2_t q1 = half2_t{pv[2], pv[3]} * (half2_t{(half_t)0.0625f, (half_t)0.125f} * half2_t{one, one});
            const half2_t q2 = half2_t{pv[4], pv[5]} * (half2_t{(half_t)0.0078125f, (half_t)0.015625f} * half2_t{one, one});
            const half2_t q3 = half2_t{pv[6], pv[7]} * half2_t{one, one};
            av[mt] = half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
          }
          g_ones[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], b_ones, (s % SPG == 0) ? zero4 : g_ones[mt], 0, 0, 0);
          g_sx[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], b_mult, (s % SPG == 0) ? zero4 : g_sx[mt], 0, 0, 0);
        }
      } else {
        av[0] = *(const half8_t *)(xr + 32 * s);
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        half2_t b0, b1, b2, b3;
        if constexpr (BITS == 4) {
          const uint32_t wv = w[s][c], w8 = wv >> 8;
          b0 = as_h2((wv & mask_lo) | kMagic); b1 = as_h2((wv & mask_hi) | kMagic);
          b2 = as_h2((w8 & mask_lo) | kMagic); b3 = as_h2((w8 & mask_hi) | kMagic);
        } else {
          // f: the lane's 8 three-bit values at bits 0,3,..,21.  f1 = f << 1 puts q5,q6,q7 at bits 16,19,22 (upper half,
          // offsets 0,3,6) and q0,q1,q2 at bits 1,4,7 (lower half): three v_and_or give (2 q0, q5), (16 q1, 8 q6),
          // (128 q2, 64 q7) on top of 1024; q3,q4 (bits 9,12) are moved to bit 0 / bit 16 separately.
          const uint32_t f = __builtin_amdgcn_alignbit(w_hi[s][c], w[s][c], shift3);
          const uint32_t f1 = f << 1;
          b0 = as_h2((f1 & m3a) | kMagic);
          b1 = as_h2((f1 & m3b) | kMagic);
          b2 = as_h2((f1 & m3c) | kMagic);
          const uint32_t lo34 = ((f >> 9) & m3d) | kMagic;
          b3 = as_h2(((f << 4) & m3e) | lo34);
        }
        const half8_t bf = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4_t cin = (s % SPG == 0) ? float4_t{0.f, 0.f, 0.f, 0.f} : gacc[mt][c];
          gacc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[mt], bf, cin, 0, 0, 0);
        }
      }
      if (s % SPG == SPG - 1) {
        const int j = s / SPG;
        float sxv[MT][4], big[MT][4];
        if constexpr (RA) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) { sxv[mt][q] = g_sx[mt][q]; big[mt][q] = 1024.f * g_ones[mt][q]; }
        } else {
          const float4_t s01 = *(const float4_t *)(sxr + j * 16);      // (Sx,Sx') rows 4g, 4g+1
          const float4_t s23 = *(const float4_t *)(sxr + j * 16 + 2);  // rows 4g+2, 4g+3
          sxv[0][0] = s01[0]; sxv[0][1] = s01[2]; sxv[0][2] = s23[0]; sxv[0][3] = s23[2];
          big[0][0] = 1024.f * s01[1]; big[0][1] = 1024.f * s01[3]; big[0][2] = 1024.f * s23[1]; big[0][3] = 1024.f * s23[3];
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          // this group's scale / zero of column n+c, converted to fp32 here (keeps the raw 16/32-bit words live instead)
          constexpr bool MS = SM && CPL > 1;  // several strips per block: column i of strip c, its own zero-point word
          const uint32_t zfield = (BITS == 3) ? (uint32_t)(((((uint64_t)(MS ? zraw2[MS ? j : 0][c] : zraw[j][1])) << 32) | (MS ? zraw[j][c] : zraw[j][0])) >> ((3 * zcol) & 31))
                                              : (MS ? (zraw[j][c] >> (4 * (zcol & 7))) : (zraw[j][0] >> (4 * ((zcol + c) & 7))));
          const float zp = (float)((zfield + (uint32_t)p.add_zero_bias) & (uint32_t)((1 << BITS) - 1));
          const uint32_t zd = MS ? zraw[j][c] : ((CPL >= 2) ? zraw[j][c >> 1] : zraw[j][0]);
          const bool hi = MS ? (zcol & 1) : ((CPL >= 2) ? (c & 1) : (zcol & 1));
          const float zh = (float)__builtin_bit_cast(half_t, (uint16_t)(hi ? (zd >> 16) : (zd & 0xffffu)));
          // branch-free select of the zero kind: with ?: on the wave-uniform zk hipcc may emit real branches around each
          // conversion (a dozen extra basic blocks per round, which also breaks up the load/MFMA schedule)
          const float zfc = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, zp) & zsel_p) | (__builtin_bit_cast(uint32_t, zh) & zsel_h) | zsel_s);
          const float sfc = (float)sc[j][c];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float corr = __builtin_fmaf(zfc, sxv[mt][q], big[mt][q]);
              yacc[mt][c][q] = __builtin_fmaf(sfc, gacc[mt][c][q] - corr, yacc[mt][c][q]);
            }
        }
      }
    }
  };
  if constexpr (ONER) round_body(0);
  else for (int r = 0; r < rounds; ++r) round_body(r);
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[3] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- 6. reduce the NW waves' partials through LDS: red[wave][row][col] -------------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * g + r;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) red[(wave * M + row) * TN + (SM ? c * 16 + i : i * CPL + c)] = yacc[mt][c][r];
      }
    }
  __syncthreads();
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[4] = __builtin_amdgcn_s_memrealtime();
  }
  for (int e = threadIdx.x; e < M * TN; e += NW * 64) {
    const int row = e / TN, col = e - row * TN;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) v += red[(wv * M + row) * TN + col];
    const int nn = b * TN + col;
    if (pr.bias) v += (float)pr.bias[nn];
    if (p.act_bf16)
      ((uint16_t *)pr.y)[(size_t)row * N + nn] = f32_to_bf16(v);
    else
      ((half_t *)pr.y)[(size_t)row * N + nn] = (half_t)v;
  }
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[5] = __builtin_amdgcn_s_memrealtime();
  }
}

template <int NW, int CPL, int MAXS, int SPG, int XL, int BITS = 4, bool RA = false, bool RA_BF16 = false, int MT = 1, bool SM = false, bool DBG = false,
          bool ONER = false>
static int launch_strip_t(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  // the >64 KB dynamic-LDS opt-in is a per-DEVICE function attribute: latch it per (kernel instantiation, device)
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)strip_kernel<NW, CPL, MAXS, SPG, XL, BITS, RA, RA_BF16, MT, SM, DBG, ONER>)) return rc;
  hipLaunchKernelGGL((strip_kernel<NW, CPL, MAXS, SPG, XL, BITS, RA, RA_BF16, MT, SM, DBG, ONER>), dim3(grid), dim3(NW * 64), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
