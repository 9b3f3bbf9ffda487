// Full-K "strip" decode kernel: y[M<=16, N] = x . dequant(W4), GPTQ/HQQ row-stream layout, NO cross-block reduction.
//
// Why a second decode kernel: at batch 1 a Llama-2-7B linear is 8.7-23 MB, i.e. 1-3 us of HBM time, the same order
// as ONE DRAM round trip under load.  The split-K kernel (skinny.hip) pays three more dependent round trips after
// its loads (slab write-through, arrival ticket, slab read-back).  Here every block owns a 16-column strip for ALL
// of K, so the dependency chain is: loads -> dequant+MFMA -> LDS reduce -> store.
//
//   * block = 16 waves (1024 threads) = one 16-column strip; wave w owns a contiguous K chunk of spw k-steps.
//   * a wave-load is 64 lanes x 4 B = 4 word-rows x 64 B (lane (g,i): word-row r+g, column n0+i): exactly the
//     B fragment of v_mfma_f32_16x16x32_f16 for 32 consecutive k.  ALL of a wave's loads -- its activation chunk
//     (LDS-DMA straight into wave-private LDS, no registers), <= 24 weight dwords per lane, the scale/zero of every
//     group it touches -- are issued back to back before the first wait: straight-line code, no branches in the loop.
//   * the two 64-byte halves of every 128-byte line belong to strips 2j and 2j+1; the block->strip map places them
//     on the same XCD (blocks b and b+8), so the line is fetched into one L2 only (measured: 10.5 -> 7.4 us on the
//     22.5 MB shape; tools/lab/memlab.hip).
//   * dequant is the bit-exact 3-op form (common.hpp); the activation fragment is permuted to the (k0,k4,k1,k5,..)
//     slot order when it is read from LDS.
//   * up to 8 layers sharing x run as one launch (q/k/v, gate/up).
//
// Replaces gemv<half> (/root/reference/csrc/ort_cuda/dq_gemv.cu:41-150).
#include <stdlib.h>

#include "kernels.hpp"

namespace qllm {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

// v + (v from the lane selected by the DPP control): folds into one v_add_f32_dpp
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// any 16-bit field of v equal to 0xFFFF?  (zero-field test on ~v)
__device__ __forceinline__ bool has_ffff16(uint64_t v) {
  const uint64_t t = ~v;
  return ((t - 0x0001000100010001ull) & ~t & 0x8000800080008000ull) != 0;
}
constexpr uint32_t kChainSpinLimit = 8192;  // polls of ~1 us each before a chained link gives up (never hang the device)

// NW: waves per block; CPL: columns per lane (1 -> 16-column strip, 64-byte row segments; 4 -> 64-column strip,
// 256-byte row segments, 4 MFMAs per k-step); MAXS: k-steps (weight loads) per lane per round; SPG: k-steps per
// quantisation group (group_size / 32); XL: 16-byte activation chunks staged per lane.
//
// Arithmetic (the kernel is VALU-issue-bound, measured SQ_ACTIVE_INST_VALU ~ 0.9 of the SIMD issue capacity with the
// per-weight fp16 dequant, so the per-weight work is cut to the bone):
//     y[m,n] = sum_G s[G,n] * ( sum_{k in G} x[m,k] q[k,n]  -  z[G,n] * sum_{k in G} x[m,k] )
//   * the B fragment is the raw "magic" fp16 pattern: 0x6400 | nibble = 1024+q for nibbles at bits 0-3/16-19 and
//     0x6400 | (nibble<<4) = 1024+16q for nibbles at bits 4-7/20-23 -- 1 shift + 4 v_and_or_b32 per 8 weights, no
//     per-weight fp16 math at all; the x16 on the odd k-slots is undone by staging x/16 in those A-fragment slots;
//   * per group the MFMA accumulator therefore holds 1024*Sx' + sum x q  (Sx' = sum of the staged A values); the
//     correction 1024*Sx' + z*Sx and the scale are applied once per group and column in fp32 (12 VALU per 32-128
//     weights), with Sx, Sx' per (row, group) computed once per wave when x is staged (v_dot2_f32_f16 + 3-4 shuffles).
//   This evaluates x.W for the UNROUNDED W = s(q-z) in fp32: it differs from the reference's fp16-rounded W path by
//   the rounding noise of W (measured <= 3e-4 relative, tests bound it at 2e-3 against float64 of the reference's W).
// Everything is straight-line: loads are never predicated (addresses are clamped instead and the surplus is
// cancelled by zero activations), so hipcc keeps all of a wave's loads in flight and waits with counted vmcnt.
// BITS: 4, or 3 (bit-stream layout, 32 k = 3 words per column; CPL = 1 only).  For 3 bits the lane assembles its 24-bit
//       field (8 values) from two words with v_alignbit, and the magic patterns put the fields at different bit offsets
//       of the fp16 mantissa: slot scales (2,1 | 16,8 | 128,64 | 1,1) for k-slots (k0,k5 | k1,k6 | k2,k7 | k3,k4), undone by
//       staging x divided by the same factors -- 9 VALU per 8 weights instead of 5.
// RA ("register A", used for M > 2): no activation slab in LDS at all.  Staging costs M*K/8 chunk operations per BLOCK
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
// CH ("chained", lds-slab form only): the launch is one link of a decode chain whose links alternate between two streams, so
//       that link i+1 is already resident and has ALL of its weight loads in flight while link i still computes (weights never
//       depend on activations).  The hand-off of the small activation vector is in-band: the producer's
//       y buffer is pre-filled with 0xFFFF halves (a NaN no finite result can equal), the producer writes its outputs
//       write-through (sc1, agent scope), and every consumer wave re-reads ITS OWN K chunk of x with L1-bypassing loads until no
//       0xFFFF half is left -- every 16-bit value is its own "ready" tag: no flag, no fence, no atomics, no block barrier
//       (cdna_hip_programming.md Guideline 16, recipe R2 with 2-byte granules).  p.chain bit 0: x is such a buffer (poll);
//       bit 1: publish y that way.  A block is at most half a CU (16 waves x <= 64 registers or 8 waves x <= 128) and the host
//       keeps the grid <= 448, so two adjacent links are always co-resident and a polling link can never keep its producer's
//       blocks off the chip; every poll loop is bounded (gives up after ~10 ms and raises bit 0 of *p.err).
// LW (CH only, exactly two rounds): the SECOND round's weights are requested up front as well -- by LDS-DMA into a wave-private
//       LDS area (no registers; 64 KB per 64-column block), its scale/zero words into registers -- so that ALL of the link's
//       weight bytes are in flight before the activations arrive; round 1 then reads its packed words back with ds_read.
//       (Measured before this existed: a two-round link spent 5-8 us after its input was complete, a one-round link 2.3.)
template <int NW, int CPL, int MAXS, int SPG, int XL, int BITS, bool RA = false, bool RA_BF16 = false, int MT = 1, bool CH = false, bool LW = false>
// (second launch-bound = minimum waves per SIMD: the 8-wave 64-column slab variant sits right at the 128-register edge
//  that lets two blocks share a CU -- 130 registers halve its occupancy: gate/up 13.7 -> 15.0 us)
__global__ __launch_bounds__(NW * 64, CH ? (NW == 8 ? 4 : 8) : ((NW == 8 && CPL == 4 && SPG == 4 && !RA) ? 4 : 1)) void strip_kernel(const StripParams p) {
  static_assert(BITS == 4 || (BITS == 3 && CPL == 1), "3-bit strips are 16 columns wide");
  static_assert(!CH || (!RA && XL <= 4 && BITS == 4 && MT == 1), "chained links: lds-slab form, 4 bits, at most 4 activation chunks per lane");
  static_assert(!LW || (CH && CPL == 4), "LDS-parked second round: chained 64-column strips");
  static_assert(!RA || MAXS == 8, "register-A rounds are 8 k-steps");
  static_assert(MT == 1 || (RA && CPL == 1), "several row tiles: register-A, 16-column strips");
  constexpr int NG = MAXS / SPG;   // groups per round (MAXS is a multiple of SPG; rounds start on a group boundary)
  constexpr int TN = 16 * CPL;     // columns per block
  constexpr int GL = 4 * SPG;      // lanes (16-byte chunks) per group in the staging pass: 8 or 16
  typedef uint32_t wvec_t __attribute__((ext_vector_type(CPL)));
  typedef float float2_t __attribute__((ext_vector_type(2)));
  // dynamic LDS: red[wave][M rows][TN cols] fp32 | per wave: activation chunk, M rows x (32*spw_pad) halves, row
  // stride + 16 B | per wave: (Sx, Sx') float2 per (group, row), 16 rows per group
  extern __shared__ __attribute__((aligned(16))) float red[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, i = lane & 15;
  // diagnostics (CH only, p.dbg != NULL): 100 MHz timestamps of the first and the last block's wave 0 --
  // [entry, weight loads issued, x complete, exit] -- written by lane 0; qllm_debug_timeline() hands out the slots
  uint64_t *dbg_slot = nullptr;
  if constexpr (CH) {
    if (p.dbg && wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) dbg_slot = p.dbg + (blockIdx.x == 0 ? 0 : 4);
    if (dbg_slot && lane == 0) dbg_slot[0] = __builtin_amdgcn_s_memrealtime();
  }

  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProblems; ++q)
    if (q < p.n_prob && (int)blockIdx.x >= p.block_begin8[q]) pi = q;
  // ... and the whole problem record plus the remaining launch scalars pulled in ONE batch: the empty asm "uses" them here, so
  // hipcc must have issued every s_load before this point instead of one at a time at first use
  const StripProblem pr = p.prob[pi];
  asm volatile("" ::"s"(pr.qweight), "s"(pr.scales), "s"(pr.qzeros), "s"(pr.bias), "s"(pr.y), "s"(pr.N), "s"(pr.n_strips),
               "s"(pr.block_begin), "s"(pr.zero_kind), "s"(p.x), "s"(p.M), "s"(p.K), "s"(p.T), "s"(p.spw), "s"(p.group_size),
               "s"(p.add_zero_bias), "s"(p.act_bf16));

  int b = blockIdx.x - pr.block_begin;
  if (CPL == 1 && (pr.n_strips & 15) == 0) {
    // 64-byte segments: strips 2j and 2j+1 share every 128-byte line -> put them on blocks b and b+8 (same XCD)
    const int x = b & 7, r = b >> 3;
    b = (((r >> 1) << 3) + x) * 2 + (r & 1);
  }
  const int N = pr.N;
  const int n = b * TN + i * CPL;  // this lane's first column (N is a multiple of TN)
  const int M = p.M;

  const int t0 = wave * p.spw;                    // spw is a multiple of SPG: every wave starts on a group boundary
  const int kend = min(32 * (t0 + p.spw), p.K);   // activations at k >= kend are staged as zero
  const int rounds = (p.spw + MAXS - 1) / MAXS;
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
    const int k = 32 * t0 + 8 * kc;
    off = (uint32_t)(row * p.K + min(k, p.K - 8));
    keep = (k < kend);
    dst = (cu <= xlast) ? row * xrow + 8 * kc : -1;
    sd = (kc / GL) * 16 + row;
  };
  if constexpr (!CH) {
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

  // CH: this wave's activation chunk, issued AFTER its weight loads.  Chained input: L1-bypassing 8-byte loads, repeated
  // until no half of the chunk is the 0xFFFF "not written yet" pattern (the wave decides as a whole; vmcnt is in-order, so the
  // first pass also waits for the wave's weights -- by then they are needed anyway).
  auto chain_load_x = [&]() {
    uint16_t *xb = (uint16_t *)p.x;
    uint32_t xoff[XL];  // (the index arithmetic sits here, after the weight loads, so it holds no registers while they are issued)
#pragma unroll
    for (int u = 0; u < XL; ++u) x_index(u, xoff[u], xkeep[u], xdst[u], sdst[u]);
    if (p.chain & 1) {
      for (uint32_t spin = 0;; ++spin) {
        bool bad = false;
#pragma unroll
        for (int u = 0; u < XL; ++u) {
          uint64_t *a = (uint64_t *)(xb + xoff[u]);
          const uint64_t lo = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint64_t hi = __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          xa[u] = uint4_t{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
          bad = bad || has_ffff16(lo) || has_ffff16(hi);
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
        if (spin >= kChainSpinLimit) {
          if (lane == 0) atomicOr(p.err, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    } else {
#pragma unroll
      for (int u = 0; u < XL; ++u) xa[u] = *(const uint4_t *)(xb + xoff[u]);
    }
  };

  auto stage_x = [&]() {
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
      sx = dpp_add<0x141>(sx); sxp = dpp_add<0x141>(sxp);
      if constexpr (GL == 16) { sx = dpp_add<0x140>(sx); sxp = dpp_add<0x140>(sxp); }
      if (xdst[u] >= 0) {
        *(half8_t *)(xs + xdst[u]) = half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
        if ((lane & (GL - 1)) == 0) sxs[sdst[u]] = float2_t{sx, sxp};
      }
    }
  };
  if (!RA && !CH && XL > 2) stage_x();

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
  const uint32_t lane_off = (uint32_t)(g * N + n);  // word offset of this lane inside a 4-row group
  // 3-bit: word rows {0,0,1,2}[g] / {0,1,2,2}[g] of the 3-row group, and the funnel shift {0,24,16,8}[g]
  const uint32_t lane_off3_lo = (uint32_t)((g == 0 ? 0 : g - 1) * N + n);
  const uint32_t lane_off3_hi = (uint32_t)((g == 3 ? 2 : g) * N + n);
  const uint32_t shift3 = (uint32_t)((32 - 8 * g) & 31);
  const int Gmax = (p.K - 1) / p.group_size;
  const int tmax = p.T - 1;
  // zero points, branch-free addressing: packed -> word (G, n/8) (CPL | 8: one word holds the lane's columns);
  // fp16 -> the dword(s) holding halves (G, n..n+CPL-1); symmetric -> any valid dword (ignored)
  const int zk = pr.zero_kind;
  const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)pr.scales : (const uint32_t *)pr.qzeros;
  // (3-bit packed: column n sits at bit 3n of the group's row of N*3/32 words and may straddle two of them: the lane keeps the
  //  word holding its first bit and the next one -- clamped to the row, where nothing straddles -- and funnel-shifts)
  const int zmul = (zk == ZK_PACKED) ? (BITS == 3 ? (N * 3) >> 5 : (N >> 3)) : (N >> 1);
  const int zoff = (zk == ZK_PACKED) ? (BITS == 3 ? (n * 3) >> 5 : (n >> 3)) : (n >> 1);
  const int zoff2 = (BITS == 3) ? ((zk == ZK_PACKED && zoff + 1 < zmul) ? 1 : 0) : ((zk == ZK_F16 && CPL == 4) ? 1 : 0);
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

  // LW: round 1's scale / zero words (registers) and the wave's LDS area for its packed words
  half_t sc1[LW ? MAXS / SPG : 1][CPL];
  uint32_t zraw1[LW ? MAXS / SPG : 1][2];
  uint32_t *wlds = (uint32_t *)((half_t *)(red + NW * M * TN) + (size_t)NW * M * xrow) + (size_t)NW * ngw * 16 * 2 + (size_t)wave * (MAXS * 64 * CPL);

  auto round_body = [&](const int r) __attribute__((always_inline)) {
    const int base = t0 + r * MAXS;
    // ---- 2. scale / zero of every group this round touches: RAW loads only (tiny; issued first) ----------------------
    const int G0 = base / SPG;
    half_t sc[NG][CPL];
    uint32_t zraw[NG][2];
    auto load_meta = [&](int Gfirst, half_t (&scx)[NG][CPL], uint32_t (&zx)[NG][2]) {
#pragma unroll
      for (int j = 0; j < NG; ++j) {
        const int G = min(Gfirst + j, Gmax);
        if constexpr (CPL == 4) {
          const half4_t sv = *(const half4_t *)(pr.scales + (size_t)G * N + n);
          scx[j][0] = sv.x; scx[j][1] = sv.y; scx[j][2] = sv.z; scx[j][3] = sv.w;
        } else if constexpr (CPL == 2) {
          const half2_t sv = *(const half2_t *)(pr.scales + (size_t)G * N + n);
          scx[j][0] = sv.x; scx[j][1] = sv.y;
        } else {
          scx[j][0] = pr.scales[(size_t)G * N + n];
        }
        zx[j][0] = zbase[(size_t)G * zmul + zoff];
        zx[j][1] = (CPL == 4 || BITS == 3) ? zbase[(size_t)G * zmul + zoff + zoff2] : 0u;
      }
    };
    if constexpr (LW) {
      if (r == 1) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
#pragma unroll
          for (int c = 0; c < CPL; ++c) sc[j][c] = sc1[j][c];
          zraw[j][0] = zraw1[j][0]; zraw[j][1] = zraw1[j][1];
        }
      } else {
        load_meta(G0, sc, zraw);
      }
    } else {
      load_meta(G0, sc, zraw);
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
    uint32_t w_hi[BITS == 3 ? MAXS : 1];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      if constexpr (LW) {
        if (r == 1) {
          w[s] = *(const wvec_t *)(wlds + (s * 64 + lane) * CPL);  // parked there by this wave's own DMA pieces
          continue;
        }
      }
      if constexpr (BITS == 4) {
        const uint32_t *rowp = pr.qweight + (size_t)(4 * min(base + s, tmax)) * N;
        w[s] = __builtin_nontemporal_load((const wvec_t *)(rowp + lane_off));
      } else {
        // 32 k = 3 words: lane group g needs stream bits [24g, 24g+24) = words {0,0,1,2}[g] and {0,1,2,2}[g]
        const uint32_t *rowp = pr.qweight + (size_t)(3 * min(base + s, tmax)) * N;
        w[s][0] = __builtin_nontemporal_load(rowp + lane_off3_lo);
        w_hi[s] = __builtin_nontemporal_load(rowp + lane_off3_hi);
      }
    }

    if constexpr (LW) {
      if (r == 0) {
        // round 1: packed words by LDS-DMA (16 B per lane and k-step, lane-linear = the order they are read back in),
        // non-temporal like the register loads; scale / zero words of its groups into registers
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
          const uint32_t *rowp = pr.qweight + (size_t)(4 * min(base + MAXS + s, tmax)) * N;
          __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(rowp + lane_off), (lds_void_t *)(wlds + s * 64 * CPL), 16, 0, 2);
        }
        load_meta(G0 + NG, sc1, zraw1);
      }
    }

    // RA: pin the issue order -- without this hipcc sinks half of the activation loads below the first MFMAs and waits
    // for them with vmcnt(0)
    if constexpr (RA) __builtin_amdgcn_sched_barrier(0);

    // ---- 4. first round: activations -> LDS (needs only the OLDEST loads; the weights stay in flight).  For many rows
    //         (XL > 2) the chunk was staged before the weight loads instead, to keep its registers out of this region.
    if constexpr (CH) {
      if (r == 0) {
        if (dbg_slot && lane == 0) dbg_slot[1] = __builtin_amdgcn_s_memrealtime();
        chain_load_x();
        if (dbg_slot && lane == 0) dbg_slot[2] = __builtin_amdgcn_s_memrealtime();
        stage_x();
      }
    }
    if (!RA && !CH && XL <= 2 && r == 0) stage_x();

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
        const bool valid = (r * MAXS + s < p.spw) && (base + s <= tmax);
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
          const uint32_t f = __builtin_amdgcn_alignbit(w_hi[s], w[s][0], shift3);
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
          const uint32_t zfield = (BITS == 3) ? (uint32_t)(((((uint64_t)zraw[j][1]) << 32) | zraw[j][0]) >> ((3 * n) & 31))
                                              : (zraw[j][0] >> (4 * ((n + c) & 7)));
          const float zp = (float)((zfield + (uint32_t)p.add_zero_bias) & (uint32_t)((1 << BITS) - 1));
          const uint32_t zd = (CPL >= 2) ? zraw[j][c >> 1] : zraw[j][0];
          const bool hi = (CPL >= 2) ? (c & 1) : (n & 1);
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
  if constexpr (LW) {
    round_body(0);
    __builtin_amdgcn_sched_barrier(0);  // keep round 1's LDS reads below round 0's arithmetic (32 more live registers otherwise)
    round_body(1);
  } else {
    for (int r = 0; r < rounds; ++r) round_body(r);
  }

  // ---- 6. reduce the NW waves' partials through LDS: red[wave][row][col] -------------------------------------------
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * g + r;
      if (row < M) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) red[(wave * M + row) * TN + i * CPL + c] = yacc[mt][c][r];
      }
    }
  __syncthreads();
  if constexpr (CH) {
    if (p.chain & 2) {
      // publish: two adjacent columns per thread as ONE 4-byte write-through (sc1) store; a result that happens to be the
      // 0xFFFF NaN pattern is rewritten to another NaN (0xFE00 / bf16 0xFFC0) so it cannot be mistaken for "not written"
      for (int e = threadIdx.x; e < M * (TN / 2); e += NW * 64) {
        const int row = e / (TN / 2), col = 2 * (e - row * (TN / 2));
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
          v0 += red[(wv * M + row) * TN + col];
          v1 += red[(wv * M + row) * TN + col + 1];
        }
        const int nn = b * TN + col;
        if (pr.bias) { v0 += (float)pr.bias[nn]; v1 += (float)pr.bias[nn + 1]; }
        uint32_t h0, h1;
        if (p.act_bf16) {
          h0 = f32_to_bf16(v0); h1 = f32_to_bf16(v1);
          h0 = (h0 == 0xffffu) ? 0xffc0u : h0; h1 = (h1 == 0xffffu) ? 0xffc0u : h1;
        } else {
          h0 = __builtin_bit_cast(uint16_t, (half_t)v0); h1 = __builtin_bit_cast(uint16_t, (half_t)v1);
          h0 = (h0 == 0xffffu) ? 0xfe00u : h0; h1 = (h1 == 0xffffu) ? 0xfe00u : h1;
        }
        __hip_atomic_store((uint32_t *)((uint16_t *)pr.y + (size_t)row * N + nn), h0 | (h1 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (dbg_slot && lane == 0) dbg_slot[3] = __builtin_amdgcn_s_memrealtime();
      return;
    }
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
}

template <int NW, int CPL, int MAXS, int SPG, int XL, int BITS = 4, bool RA = false, bool RA_BF16 = false, int MT = 1, bool CH = false, bool LW = false>
static int launch_strip_t(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  // the >64 KB dynamic-LDS opt-in is a per-DEVICE function attribute: latch it per (kernel instantiation, device)
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)strip_kernel<NW, CPL, MAXS, SPG, XL, BITS, RA, RA_BF16, MT, CH, LW>)) return rc;
  hipLaunchKernelGGL((strip_kernel<NW, CPL, MAXS, SPG, XL, BITS, RA, RA_BF16, MT, CH, LW>), dim3(grid), dim3(NW * 64), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

// (waves per block, weight loads per lane per round): 16 waves x 8 or 24, or 8 waves x 16
// (64-column strips always use rounds of 8: 4 dwords per load keep the register budget of a 1024-thread block;
//  register-A launches always use rounds of 8: each k-step also holds 16 B of activations per lane)
// chained links (chain != 0): 16-wave blocks must fit 64 registers -> one round of 8 (short K only); 8-wave blocks (128
// registers): see below
int strip_maxs(int nw, int spw, int cpl, int ra, int chain) {
  if (ra || cpl >= 2) return 8;
  if (chain) {
    if (nw == 16) return 8;
    // 8-wave, 16-column links: the largest round of {24, 16, 8} k-steps that pads the wave's chunk least.  (One round of
    // 32 / 48 loads -- everything in flight before the input arrives -- was measured SLOWER on 11008 -> 4096: the 48-load
    // form spills 12-23 registers and takes 4 us to issue; QLLM_CHAIN_ONE_ROUND=1 selects it for experiments.)
    static const int one_round = getenv("QLLM_CHAIN_ONE_ROUND") ? atoi(getenv("QLLM_CHAIN_ONE_ROUND")) : 0;
    if (one_round) {
      for (int m : {8, 16, 24, 32, 48})
        if (spw <= m) return m;
      return 24;
    }
    int best = 8, best_pad = (spw + 7) / 8 * 8;
    for (int m = 16; m <= 24; m += 8) {
      const int pad = (spw + m - 1) / m * m;
      if (pad <= best_pad) { best = m; best_pad = pad; }
    }
    return best;
  }
  return nw == 8 ? 16 : (spw <= 8 ? 8 : 24);
}
static int strip_spw_pad(int nw, int spw, int cpl, int ra, int chain) { const int m = strip_maxs(nw, spw, cpl, ra, chain); return (spw + m - 1) / m * m; }
static int strip_xl(int nw, int M, int spw, int cpl, int chain) { return (M * strip_spw_pad(nw, spw, cpl, 0, chain) * 4 + 63) / 64; }

template <int SPG, bool BF>
static int launch_strip_ra(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.M > 16) {  // several 16-row tiles per block: 16-column strips, 8-wave blocks (register budget)
    if (p.bits == 3)
      return p.M > 32 ? launch_strip_t<8, 1, 8, SPG, 1, 3, true, BF, 4>(p, grid, lds, stream) : launch_strip_t<8, 1, 8, SPG, 1, 3, true, BF, 2>(p, grid, lds, stream);
    return p.M > 32 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 4>(p, grid, lds, stream) : launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 2>(p, grid, lds, stream);
  }
  if (p.bits == 3) return launch_strip_t<16, 1, 8, SPG, 1, 3, true, BF>(p, grid, lds, stream);
  if (p.cpl == 4)
    return p.nw == 8 ? launch_strip_t<8, 4, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream)
                     : launch_strip_t<16, 4, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);
  if (p.cpl == 2) return launch_strip_t<16, 2, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);
  return p.nw == 8 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream)
                   : launch_strip_t<16, 1, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);
}

template <int SPG>
static int launch_strip_s(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.ra) return p.act_bf16 ? launch_strip_ra<SPG, true>(p, grid, lds, stream) : launch_strip_ra<SPG, false>(p, grid, lds, stream);
  if (p.chain) {  // chained links: 4 bits, <= 4 activation chunks per lane, g128 for the 64-column strips (host planner)
    const bool x2 = strip_xl(p.nw, p.M, p.spw, p.cpl, 1) <= 2;
#define QLLM_CH(NW_, CPL_, MAXS_) \
  return x2 ? launch_strip_t<NW_, CPL_, MAXS_, SPG, 2, 4, false, false, 1, true>(p, grid, lds, stream) \
            : launch_strip_t<NW_, CPL_, MAXS_, SPG, 4, 4, false, false, 1, true>(p, grid, lds, stream)
    if (p.cpl == 4) {
      if constexpr (SPG == 4) {
        if (p.lw)  // exactly two rounds: the second one parked in LDS by DMA
          return x2 ? launch_strip_t<8, 4, 8, SPG, 2, 4, false, false, 1, true, true>(p, grid, lds, stream)
                    : launch_strip_t<8, 4, 8, SPG, 4, 4, false, false, 1, true, true>(p, grid, lds, stream);
        QLLM_CH(8, 4, 8);
      } else {
        return set_error(QLLM_ERR_UNSUPPORTED, "chained 64-column strips: group size 128 only");
      }
    }
    if (p.nw == 16) { QLLM_CH(16, 1, 8); }
    const int m = strip_maxs(8, p.spw, 1, 0, 1);
    if (m == 48) { QLLM_CH(8, 1, 48); }
    if (m == 32) { QLLM_CH(8, 1, 32); }
    if (m == 24) { QLLM_CH(8, 1, 24); }
    if (m == 16) { QLLM_CH(8, 1, 16); }
    QLLM_CH(8, 1, 8);
#undef QLLM_CH
  }
  const bool small_x = strip_xl(p.nw, p.M, p.spw, p.cpl, 0) <= 2;
  if (p.bits == 3) {  // 16-column strips, 16 waves
    if (strip_maxs(16, p.spw, 1, 0, 0) == 8)
      return small_x ? launch_strip_t<16, 1, 8, SPG, 2, 3>(p, grid, lds, stream) : launch_strip_t<16, 1, 8, SPG, 8, 3>(p, grid, lds, stream);
    return small_x ? launch_strip_t<16, 1, 24, SPG, 2, 3>(p, grid, lds, stream) : launch_strip_t<16, 1, 24, SPG, 8, 3>(p, grid, lds, stream);
  }
  if (p.cpl == 4 && p.nw == 8)  // 64-column strips, 8-wave blocks (two co-resident per CU), rounds of 8 k-steps
    return small_x ? launch_strip_t<8, 4, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<8, 4, 8, SPG, 8>(p, grid, lds, stream);
  if (p.cpl == 4)  // 64-column strips: 16 waves, rounds of 8 k-steps
    return small_x ? launch_strip_t<16, 4, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 4, 8, SPG, 8>(p, grid, lds, stream);
  if (p.cpl == 2)  // 32-column strips: 16 waves, rounds of 8 k-steps
    return small_x ? launch_strip_t<16, 2, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 2, 8, SPG, 8>(p, grid, lds, stream);
  if (p.nw == 8)
    return small_x ? launch_strip_t<8, 1, 16, SPG, 2>(p, grid, lds, stream) : launch_strip_t<8, 1, 16, SPG, 8>(p, grid, lds, stream);
  if (strip_maxs(16, p.spw, 1, 0, 0) == 8)
    return small_x ? launch_strip_t<16, 1, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 1, 8, SPG, 8>(p, grid, lds, stream);
  return small_x ? launch_strip_t<16, 1, 24, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 1, 24, SPG, 8>(p, grid, lds, stream);
}

// group sizes the strip kernel serves: 64 and 128 (k-steps per group 2, 4); others use the split-K kernel
bool strip_group_ok(int group_size) { return group_size == 64 || group_size == 128; }

// waves per block: 8-wave blocks (4 per CU) when there are enough strips to need more than one round of 16-wave
// blocks (2 per CU) and K is short enough for one 16-dword round per wave; else 16 waves
int strip_nw(int K, int strips_total) { return (strips_total > 2 * kNumCU && K / 32 <= 8 * 16) ? 8 : 16; }

// k-steps per wave: all of K over nw waves, rounded up to whole groups
int strip_spw(int K, int group_size, int nw) {
  const int T = K / 32, spg = group_size / 32;
  int spw = (T + nw - 1) / nw;
  return (spw + spg - 1) / spg * spg;
}

size_t strip_lds_bytes(int M, int spw, int nw, int cpl, int group_size, int ra, int chain) {
  const size_t red = (size_t)nw * M * 16 * cpl * sizeof(float);
  if (ra) return red;  // register-A: only the cross-wave reduction buffer
  const int pad = strip_spw_pad(nw, spw, cpl, 0, chain);
  const int groups = pad / (group_size / 32);  // groups per wave chunk
  const size_t base = red + (size_t)nw * M * (pad * 32 + 8) * sizeof(half_t) +
                      (size_t)nw * groups * 16 * 8;  // (Sx, Sx') float2 per (group, row), 16 rows per group
  // chained 64-column links with two rounds park the second round's packed words in LDS: 8 k-steps x 64 lanes x 16 B per wave
  return base + (strip_lw(nw, spw, cpl, group_size, chain) ? (size_t)nw * 8 * 64 * 16 : 0);
}

// chained link whose second (and last) round is requested up front through LDS-DMA: 8-wave 64-column strips, g128, 9..16 k-steps
// per wave (K <= 4096)
bool strip_lw(int nw, int spw, int cpl, int group_size, int chain) {
  // measured (profiles/r02_chain_experiments.md): the DMA form costs 36 spilled registers and 4-6 us to issue; the step got
  // 40 % slower.  Kept as an experiment knob, off by default.
  static const int on = getenv("QLLM_CHAIN_LW") ? atoi(getenv("QLLM_CHAIN_LW")) : 0;
  return on && chain && cpl == 4 && nw == 8 && group_size == 128 && spw > 8 && spw <= 16;
}

// columns per lane, from measurements on Llama-2-7B shapes (tools/kbench.py --grouped, us per launch, cpl 1/2/4):
//   q/k/v 12288 cols: 11.7 / 10.6 / 8.5    gate/up 22016 cols: 18.7 / 16.1 / 15.6
//   o 4096 cols: 5.3 / 6.1 / 7.0            down 4096 cols (K=11008): 10.1 / 11.2 / 14.7
// -> 64-column strips (256-byte row segments) as soon as they alone give >= 160 blocks, else 16-column strips.
// Llama-2-70B shapes and their 8-way shards (M = 1, us, cpl 1 / 2 / 4; tools/narrow_ab.py, profiles/r02_narrow_shapes.md):
//   8192 -> 2 x 3584: 18.0 / 10.1 / 11.5    8192 -> 8192: 17.6 / 10.9 / 12.3    3584 -> 8192: 7.5 / 6.5 / 7.4    1024 -> 8192: 7.2 / 5.9 / 5.1
// -> 32-column strips when they give >= 190 blocks and the 64-column ones do not (M <= 4 only: all_mult32 is passed false above).
int strip_cpl(int cols_total, bool all_mult64, bool all_mult32) {
  if (all_mult64 && cols_total / 64 >= 160) return 4;
  if (all_mult32 && cols_total / 32 >= 190) return 2;
  return 1;
}

// activation staging budget: at most 8 sixteen-byte chunks per lane
bool strip_x_ok(int M, int spw, int nw, int cpl, int chain) { return strip_xl(nw, M, spw, cpl, chain) <= (chain ? 4 : 8); }

int launch_strip(const StripParams &p, int grid, hipStream_t stream) {
  const size_t lds = strip_lds_bytes(p.M, p.spw, p.nw, p.cpl, p.group_size, p.ra, p.chain ? 1 : 0);
  if (p.group_size == 64) return launch_strip_s<2>(p, grid, lds, stream);
  return launch_strip_s<4>(p, grid, lds, stream);
}

}  // namespace qllm
