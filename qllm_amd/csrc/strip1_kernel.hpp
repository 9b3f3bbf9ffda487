// Batch-1 decode kernel of the native strip-major layout (round 5): y[1, N] = x . dequant(W) for 4-bit layers with 128-wide groups.
// The same decomposition as strip_kernel.hpp's lds-slab form -- one block = one 16-column strip for ALL of K, its NW waves split K,
// every wave issues all of its loads before its first wait, no cross-block reduction -- rebuilt around what a launch costs BESIDES
// its bytes (profiles/r05_decode_bisect.md: the production kernel ran 1.2-1.5 us per launch above a read + reduce + store skeleton of
// the same launch shape):
//
//   * ONE batch of scalar loads.  The problem (layer) of a grouped launch is blockIdx.y, so the problem record's address depends on
//     nothing that has to be loaded first: header and record leave together (strip_kernel: n_prob -> block_begin8[1..] -> prob[pi]
//     -> late header words: 2 dependent round trips on single launches, 3 on gate/up, 4 on q/k/v).  Groups whose layers differ in
//     width launch max(n_strips) columns of blocks; the surplus blocks leave at once.
//   * ONE scale and ONE zero-point load per lane.  Lane (g, i) fetches group G0 + g of column i (strip_kernel: every lane all four
//     groups of its column) and finishes exactly that group:
//   * one accumulator for all groups.  The A operand's 16 rows are free at batch 1 (one real row): rows 4j..4j+3 carry x on the
//     k-steps of the wave's group j and zeros elsewhere (the lane's ds_read address is chosen per group, once: no instruction in the
//     loop), so after the wave's 16 (24) MFMAs lane (g, i) holds  sum_{k in group g} x_k (1024 + q_k)  of column i in its own
//     accumulator rows -- no accumulator reset per group, no select, and the per-group correction
//         y_g = s_g (acc_g - (z_g Sx_g + 1024 Sx'_g))
//     is ONE fma per lane after the last MFMA: s_g and c_g = s_g (z_g Sx_g + 1024 Sx'_g) are computed while the weights are still
//     in flight (Sx_g / Sx'_g are already in the lane: the 16-lane DPP row reduce of the staging pass leaves group g's sums in
//     every lane of row g, so they never go through LDS).  strip_kernel did the four groups one after the other on 16 lanes
//     behind the last MFMA (~100 instructions and a scalar load in the tail).
//   * chains: the MFMAs alternate between NCH accumulators (k-step s -> chain s % NCH) so that the last k-steps to arrive do not
//     queue behind one dependent chain.
//   * the cross-wave and cross-group sum is one pass: every lane stores its value to red[column][wave][g] and the first 16 lanes
//     of wave 0 read their column's NW x 4 values as float4s.
//
// Arithmetic contract: strip_kernel.hpp's (x . s(q - z) for the unrounded W, fp32): raw 0x6400 | nibble patterns as B fragments, the
// x16 of the odd nibbles undone by staging x / 16 in those A slots, one fp32 correction per group.
//
// LVL (lab builds only; the library instantiates LVL = 4): the bisect of tools/lab/dbisect.hip -- 0: loads + xor + reduce + store
// (the memlab2 skeleton behind this kernel's prologue), 1: + scale / zero loads, 2: + x staged through LDS (permute, Sx / Sx' DPP
// sums) and the A fragments read back, 3: + pattern build and MFMAs, 4: + corrections = the kernel.
//
// Replaces gemv<half> (/root/reference/csrc/ort_cuda/dq_gemv.cu:41-150).
#pragma once
#include "kernels.hpp"

namespace qllm {

template <int CTRL>
__device__ __forceinline__ float dpp_add1(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// bytes of LDS per block: red[16 columns][36 floats >= NW * 4] | per wave: staged x (XL KB) + 256 zero bytes
// floats per column of the reduction buffer: NW x 4 values, rows 16-byte aligned and never a multiple of 32 floats apart (7 and 15
// waves would otherwise put all 16 columns on the same banks)
constexpr int strip1_rs(int nw) { return nw * 4 + 4 + (((nw * 4 + 4) % 32 == 0) ? 4 : 0); }

template <int NW, int MAXS, int MR = 1>
constexpr int strip1_lds_bytes() {
  constexpr int XL = (MAXS * 4 + 63) / 64;
  static_assert(NW * 4 <= 64, "at most 16 waves");
  return 16 * MR * strip1_rs(NW) * 4 + NW * (MR * XL * 1024 + 256);
}

// NW waves x MAXS k-steps cover T (host: NW * MAXS >= T >= MAXS; EXACT: NW * MAXS == T, no masking of a shifted window)
// AR: the row-parallel form (one layer per launch, p.ar_* set): instead of storing y the block pushes its 16 partial outputs, rounded to
//     the activation type like the unfused path's y, into slot [parity][rank] of EVERY peer's staging buffer (comm.hip's layout and
//     protocol) and takes a ticket; the rank's last block publishes the world's flags, waits for the peers' and writes the sum of the
//     slots in rank order -- the o_proj / down_proj launch and its all-reduce are ONE launch (160 kernel boundaries per Llama-2-70B token).
// G64 (round 6): 64-wide groups (HQQ's default; the batch-1 kernel took 128-wide groups only and a g64 layer fell back to the general strip
//     kernel: 35.5 us per Llama-2-7B decoder layer against 26.3).  A group is then TWO k-steps, a pass of 16 k-steps holds EIGHT groups:
//     A rows 2 j, 2 j + 1 carry x on the k-steps of group j, so lane (g, i) finds group 2 g of the pass in accumulator registers 0 / 1
//     and group 2 g + 1 in registers 2 / 3 -- two scales, two zero points, two corrections per lane and pass; the staging pass sums x
//     over the 8 lanes of a group and fetches the neighbouring group's sums with one more DPP move.
// MR = 4 (round 6): batches 2..4 at the price of batch 1.  With 128-wide groups the A operand's rows 4 j .. 4 j + 3 all belong to group j of
//     the pass -- at batch 1 they carry four copies of x.  Here row 4 j + m carries batch row m (rows past M: zeros), so accumulator
//     register m of lane (g, i) is batch row m of column i for group g: the same weight stream, the same MFMAs, four times the staging
//     and four corrections per lane.  (Until then batches 2..4 took strip_dma.hpp: ~1.5 x a batch-1 launch.)
// B3 (round 6): 3-bit layers (configs[3] mixes 3- and 4-bit HQQ layers; at batch 1 the 3-bit ones ran on the general strip kernel: 43 us per
//     Llama-2-7B decoder layer against 28.6 for the 4-bit ones).  The recipe of strip_kernel.hpp inside this kernel's skeleton: a k-step is
//     three word rows of the strip; lane (g, i) needs bits [24 g, 24 g + 24) of column i's 96: two word loads (rows {0,0,1,2}[g] and
//     {0,1,2,2}[g]) + one v_alignbit; fragment slots (k0,k5 | k1,k6 | k2,k7 | k3,k4) taken where the fields sit, their weights
//     (2,1 | 16,8 | 128,64 | 1,1) divided out of the staged activations (exact: powers of two).
template <int NW, int MAXS, bool EXACT, int NCH = 2, int LVL = 4, bool DBG = false, bool AR = false, bool G64 = false, int MR = 1, bool B3 = false>
// (second launch bound = minimum waves per SIMD: 64 registers up to rounds of 24 k-steps -- a CU full of waves -- 128 above)
__global__ __launch_bounds__(NW * 64, (MAXS <= 24 && !G64 && MR == 1 && !B3) ? 8 : 4) void strip1_kernel(const Strip1Params p) {
  static_assert(MAXS % 4 == 0 && MAXS >= 8 && MAXS <= 64, "rounds are whole 128-wide groups, at most 16 of them (round 6: 40 .. 64 k-steps for K up to 32768)");
  static_assert(!(G64 && AR), "the fused all-reduce form is built for 128-wide groups");
  static_assert(MR == 1 || (MR == 4 && !G64 && !AR && !DBG && LVL == 4), "four batch rows: 128-wide groups, the plain form");
  static_assert(!B3 || (MR == 1 && !AR && !DBG && LVL == 4), "3 bits: the plain batch-1 form");
  constexpr int KPG = G64 ? 2 : 4;        // k-steps per group
  constexpr int NG = MAXS / KPG;          // groups per wave
  constexpr int NPASS = (MAXS + 15) / 16; // accumulator sets: one per 16 k-steps (four 128-wide groups / eight 64-wide ones)
  constexpr int GPL = G64 ? 2 : 1;        // groups per lane and pass
  constexpr int XL = (MAXS * 4 + 63) / 64;  // 16-byte activation chunks per lane
  constexpr int RS = strip1_rs(NW);       // floats per column of the reduction buffer (16-byte aligned rows, 2-way banks at most)
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, i = lane & 15;
  const int b = blockIdx.x;
  // the problem record and the header in ONE batch of scalar loads: the empty asm "uses" every word here, so hipcc cannot leave
  // any of them for a second round trip behind the surplus-block test
  const Strip1Problem pr = p.prob[blockIdx.y];
  asm volatile("" ::"s"(pr.qweight), "s"(pr.scales), "s"(pr.qzeros), "s"(pr.bias), "s"(pr.y), "s"(pr.n_strips), "s"(pr.zero_kind), "s"(p.x),
               "s"(p.T), "s"(p.n_groups), "s"(p.add_zero_bias), "s"(p.act_bf16));
  uint64_t *dbg_slot = nullptr;
  if constexpr (DBG) {
    if (p.dbg && wave == 0 && blockIdx.y == 0) {
      if (b == 0) dbg_slot = p.dbg;
      else if (b == (int)gridDim.x / 2) dbg_slot = p.dbg + 8;
      else if (b == (int)gridDim.x - 1) dbg_slot = p.dbg + 16;
    }
    if (dbg_slot && lane == 0) dbg_slot[0] = __builtin_amdgcn_s_memrealtime();
  }
  if (b >= pr.n_strips) return;  // (groups of layers of different widths)

  const int T = p.T;
  const int t0 = wave * MAXS;                              // the wave owns k-steps [t0, min(t0 + MAXS, T))
  const int tb = EXACT ? t0 : min(t0, T - MAXS);           // its window [tb, tb + MAXS): shifted back into the strip at the end of K
  // ---- every load of the wave, back to back: x chunk(s), one scale + one zero word per pass, MAXS weight words -----------------
  uint4_t xa[MR][XL];
  bool xkeep[XL];
#pragma unroll
  for (int u = 0; u < XL; ++u) {
    const int c = lane + 64 * u;                           // 16-byte chunk of the window: k-step tb + c / 4
    const int cc = min(c, MAXS * 4 - 1);
#pragma unroll
    for (int m = 0; m < MR; ++m) {                         // (MR = 4: batch row m of x, row stride K = 32 T halves; rows past M stay zero)
      xa[m][u] = uint4_t{0, 0, 0, 0};
      if (MR == 1 || m < p.M) xa[m][u] = *(const uint4_t *)((const uint16_t *)p.x + (size_t)m * 32 * T + 32 * tb + 8 * cc);
    }
    const int t = tb + (c >> 2);
    xkeep[u] = EXACT ? (c < MAXS * 4) : (c < MAXS * 4 && t >= t0 && t < T);
  }
  const int G0 = tb / KPG;
  const size_t grow = (size_t)b * p.n_groups + G0;         // first group row of the wave in the strip's scale / zero tables
  const int zk = pr.zero_kind;
  const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)pr.scales : (const uint32_t *)pr.qzeros;
  const int zmul = (zk == ZK_PACKED) ? 2 : 8;              // dwords per group row
  const int zoff = (zk == ZK_PACKED) ? (i >> 3) : (i >> 1);
  half_t sc[NPASS][GPL];
  uint32_t zraw[NPASS][GPL], zraw2[B3 ? NPASS : 1][GPL];  // (zraw2: packed 3-bit zero points are bit 3 i of the group row's 64-bit pair)
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
    for (int h = 0; h < GPL; ++h) {
      // the lane's group(s) of the pass: 4 ps + g (128-wide), 8 ps + 2 g + h (64-wide)
      const int gj = min((G64 ? 8 * ps + 2 * g + h : 4 * ps + g), NG - 1);  // (lanes past the last group re-read it; their sums of x are zero)
      sc[ps][h] = pr.scales[(grow + gj) * 16 + i];
      if constexpr (B3) {
        // packed: both words of the pair; fp16: the dword holding the half; symmetric: any valid dword
        zraw[ps][h] = zbase[(grow + gj) * zmul + ((zk == ZK_PACKED) ? 0 : zoff)];
        zraw2[ps][h] = zbase[(grow + gj) * zmul + ((zk == ZK_PACKED) ? 1 : zoff)];
      } else {
        zraw[ps][h] = zbase[(grow + gj) * zmul + zoff];
      }
    }
  uint32_t w[MAXS], wh[B3 ? MAXS : 1];
  if constexpr (B3) {
    // the strip is [3 T word rows][16]: k-step t = rows 3 t .. 3 t + 2; the lane's 24-bit field straddles rows {0,0,1,2}[g] / {0,1,2,2}[g]
    const uint32_t *wl3 = pr.qweight + ((size_t)b * T + tb) * 48 + i;
    const int lo_off = (g == 0 ? 0 : g - 1) * 16, hi_off = (g == 3 ? 2 : g) * 16;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
      w[s] = __builtin_nontemporal_load(wl3 + s * 48 + lo_off);
      wh[s] = __builtin_nontemporal_load(wl3 + s * 48 + hi_off);
    }
  } else {
    const uint32_t *wl = pr.qweight + ((size_t)b * T + tb) * 64 + lane;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) w[s] = __builtin_nontemporal_load(wl + s * 64);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[1] = __builtin_amdgcn_s_memrealtime();
  }

  // wave-private LDS: staged activations (XL KB: chunk c at 16 c) and 256 zero bytes
  char *wbase = (char *)lds + 16 * MR * RS * 4 + wave * (MR * XL * 1024 + 256);   // (MR = 4: batch row m's chunks at + m XL KB)
  char *zeros = wbase + MR * XL * 1024;
  uint32_t fold = 0;  // (LVL < 4: keeps the loaded values alive)

  // ---- activations -> LDS (needs only the OLDEST loads; the weights stay in flight) ---------------------------------------------
  float sx[MR][XL], sxp[MR][XL], sx2[XL], sxp2[XL];  // (sx2 / sxp2: the lane's second group of the pass, 64-wide groups only)
  if constexpr (LVL >= 2) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int u = 0; u < XL; ++u) asm volatile("" : "+v"(xa[m][u]));  // (pins the staging below the weight loads: strip_kernel.hpp)
    *(uint32_t *)(zeros + 4 * lane) = 0u;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int u = 0; u < XL; ++u) {
      const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
      half8_t xv = p.act_bf16 ? bf16x8_to_h8(xa[m][u]) : __builtin_bit_cast(half8_t, xa[m][u]);
      xv = xkeep[u] ? xv : zero8;
      // fragment slot order and per-slot divisors (the B-fragment construction below):
      //   4 bits: (k0,k4 | k1,k5 | k2,k6 | k3,k7), divisors (1,1 | 16,16 | 1,1 | 16,16);  3 bits: (k0,k5 | k1,k6 | k2,k7 | k3,k4), (2,1 | 16,8 | 128,64 | 1,1)
      const half8_t pv = B3 ? __builtin_shufflevector(xv, xv, 0, 5, 1, 6, 2, 7, 3, 4) : a_perm_04152637(xv);
      const half2_t p0 = {pv[0], pv[1]}, p1 = {pv[2], pv[3]}, p2 = {pv[4], pv[5]}, p3 = {pv[6], pv[7]};
      const half2_t d0 = B3 ? half2_t{(half_t)0.5f, (half_t)1.f} : half2_t{(half_t)1.f, (half_t)1.f};
      const half2_t d1 = B3 ? half2_t{(half_t)0.0625f, (half_t)0.125f} : half2_t{(half_t)0.0625f, (half_t)0.0625f};
      const half2_t d2 = B3 ? half2_t{(half_t)0.0078125f, (half_t)0.015625f} : half2_t{(half_t)1.f, (half_t)1.f};
      const half2_t d3 = B3 ? half2_t{(half_t)1.f, (half_t)1.f} : half2_t{(half_t)0.0625f, (half_t)0.0625f};
      const half2_t q0 = B3 ? p0 * d0 : p0, q1 = p1 * d1, q2 = B3 ? p2 * d2 : p2, q3 = B3 ? p3 : p3 * d3;
      const half2_t one = {(half_t)1.f, (half_t)1.f};
      float a = __builtin_amdgcn_fdot2(p0, one, 0.f, false);
      a = __builtin_amdgcn_fdot2(p1, one, a, false);
      a = __builtin_amdgcn_fdot2(p2, one, a, false);
      a = __builtin_amdgcn_fdot2(p3, one, a, false);
      float c = __builtin_amdgcn_fdot2(q0, one, 0.f, false);
      c = __builtin_amdgcn_fdot2(q1, one, c, false);
      c = __builtin_amdgcn_fdot2(q2, one, c, false);
      c = __builtin_amdgcn_fdot2(q3, one, c, false);
      // all-reduce over the 16 lanes of the row (= the 128 k of one group): xor 1, xor 2, half-row mirror, row mirror
      a = dpp_add1<0xB1>(a); c = dpp_add1<0xB1>(c);
      a = dpp_add1<0x4E>(a); c = dpp_add1<0x4E>(c);
      a = dpp_add1<0x141>(a); c = dpp_add1<0x141>(c);
      if constexpr (G64) {
        // 64-wide groups: the 8-lane halves of the row are two groups (2 g, 2 g + 1 of the pass); every lane needs both
        const float ao = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x140, 0xF, 0xF, true));
        const float co = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x140, 0xF, 0xF, true));
        const bool lo = (lane & 8) == 0;
        sx[m][u] = lo ? a : ao; sxp[m][u] = lo ? c : co;      // group 8 u + 2 g
        sx2[u] = lo ? ao : a; sxp2[u] = lo ? co : c;    // group 8 u + 2 g + 1
      } else {
        a = dpp_add1<0x140>(a); c = dpp_add1<0x140>(c);
        sx[m][u] = a; sxp[m][u] = c;   // lane (g, i): sums of group 4 u + g of the wave (batch row m)
      }
      *(half8_t *)(wbase + m * (XL * 1024) + 16 * (lane + 64 * u)) = half8_t{q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, q3.x, q3.y};
    }
  } else {
#pragma unroll
    for (int u = 0; u < XL; ++u) { fold ^= xa[0][u].x ^ xa[0][u].y ^ xa[0][u].z ^ xa[0][u].w; sx[0][u] = sxp[0][u] = sx2[u] = sxp2[u] = 0.f; }
  }
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[2] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- per lane: scale and correction of ITS group (pass ps: group 4 ps + g), while the weights are in flight ------------------
  constexpr int NCF = (MR > 1) ? MR : GPL;   // corrections per lane and pass: its group(s), or its four batch rows
  float sf[NPASS][GPL], cf[NPASS][NCF];
  if constexpr (LVL >= 4) {
    const uint32_t zsel_p = (zk == ZK_PACKED) ? 0xffffffffu : 0u, zsel_h = (zk == ZK_F16) ? 0xffffffffu : 0u;
    const uint32_t zsel_s = (zk == ZK_SYM) ? __builtin_bit_cast(uint32_t, B3 ? 4.0f : 8.0f) : 0u;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int h = 0; h < GPL; ++h) {
        const uint32_t zr = zraw[ps][h];
        float zp;
        if constexpr (B3) zp = (float)(((uint32_t)(((((uint64_t)zraw2[ps][h]) << 32) | zr) >> (3 * i)) + (uint32_t)p.add_zero_bias) & 7u);
        else zp = (float)(((zr >> (4 * (i & 7))) + (uint32_t)p.add_zero_bias) & 15u);
        const float zh = (float)__builtin_bit_cast(half_t, (uint16_t)((i & 1) ? (zr >> 16) : (zr & 0xffffu)));
        const float zf = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, zp) & zsel_p) | (__builtin_bit_cast(uint32_t, zh) & zsel_h) | zsel_s);
        sf[ps][h] = (float)sc[ps][h];
        if constexpr (MR > 1) {
#pragma unroll
          for (int m = 0; m < MR; ++m) cf[ps][m] = sf[ps][0] * __builtin_fmaf(zf, sx[m][ps], 1024.f * sxp[m][ps]);
        } else {
          cf[ps][h] = sf[ps][h] * __builtin_fmaf(zf, h ? sx2[ps] : sx[0][ps], 1024.f * (h ? sxp2[ps] : sxp[0][ps]));
        }
      }
  } else if constexpr (LVL >= 1) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int h = 0; h < GPL; ++h) fold ^= zraw[ps][h] ^ (uint32_t)__builtin_bit_cast(uint16_t, sc[ps][h]);
  }

  // ---- A fragment addresses: lane (g, i) is row i of the A operand; rows 4 j .. 4 j + 3 belong to group j of the pass -----------
  // k-step s reads at byte offset 64 s (an immediate): from the staged chunk if the row belongs to the k-step's group, else from
  // the zero block (whose address is pre-biased by the group's first offset)
  uint32_t a_addr[NG];
  const uint32_t xs_lane = (uint32_t)(wbase - (char *)lds) + 16 * g, z_lane = (uint32_t)(zeros - (char *)lds) + 16 * g;
#pragma unroll
  for (int j = 0; j < NG; ++j)   // (the zero block is read at + 64 s for the k-steps s of group j: bias its address by the group's first offset)
    a_addr[j] = (G64 ? ((i >> 1) == (j & 7)) : ((i >> 2) == (j & 3))) ? xs_lane + (MR > 1 ? (i & 3) * (XL * 1024) : 0) : z_lane - 64 * KPG * j;

  float4_t acc[NPASS][NCH];
  const uint32_t mask_lo = nib_mask_vgpr();  // 0x000f000f
  const uint32_t mask_hi = mask_lo << 4;     // 0x00f000f0
  // 3-bit field masks, derived from the opaque VGPR so hipcc can fuse each (x & m) | magic into one v_and_or_b32
  const uint32_t m3a = ((mask_lo & 0x7u) << 1) | (mask_lo & 0x00070000u);  // 0x0007000E
  const uint32_t m3b = m3a << 3, m3c = m3a << 6;                           // 0x00380070, 0x01C00380
  const uint32_t m3d = mask_lo & 0x00000007u, m3e = mask_lo & 0x00070000u;
  const uint32_t shift3 = (uint32_t)((32 - 8 * g) & 31);                   // funnel shift {0, 24, 16, 8}[g]
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int j = s / KPG, ps = s >> 4, ch = s % NCH;
    if constexpr (LVL >= 2) {
      const half8_t av = *(const half8_t *)((const char *)lds + a_addr[j] + 64 * s);
      if constexpr (LVL >= 3) {
        half2_t b0, b1, b2, b3;
        if constexpr (B3) {
          // f: the lane's 8 three-bit values at bits 0, 3, .., 21.  f1 = f << 1 puts q5, q6, q7 at bits 16, 19, 22 (upper half, offsets
          // 0, 3, 6) and q0, q1, q2 at bits 1, 4, 7 (lower half): three v_and_or give (2 q0, q5), (16 q1, 8 q6), (128 q2, 64 q7) on top of
          // 1024; q3, q4 (bits 9, 12) are moved to bit 0 / bit 16 separately
          const uint32_t f = __builtin_amdgcn_alignbit(wh[s], w[s], shift3), f1 = f << 1;
          b0 = as_h2((f1 & m3a) | kMagic);
          b1 = as_h2((f1 & m3b) | kMagic);
          b2 = as_h2((f1 & m3c) | kMagic);
          b3 = as_h2(((f << 4) & m3e) | (((f >> 9) & m3d) | kMagic));
        } else {
          const uint32_t wv = w[s], w8 = wv >> 8;
          b0 = as_h2((wv & mask_lo) | kMagic); b1 = as_h2((wv & mask_hi) | kMagic);
          b2 = as_h2((w8 & mask_lo) | kMagic); b3 = as_h2((w8 & mask_hi) | kMagic);
        }
        const half8_t bf = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
        const bool first = (s - 16 * ps) < NCH;  // the chain's first k-step of this pass
        const float4_t cin = first ? float4_t{0.f, 0.f, 0.f, 0.f} : acc[ps][ch];
        acc[ps][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bf, cin, 0, 0, 0);
      } else {
        const uint4_t ar = __builtin_bit_cast(uint4_t, av);
        fold ^= w[s] ^ ar.x ^ ar.y ^ ar.z ^ ar.w;
      }
    } else {
      fold ^= w[s];
    }
  }
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[3] = __builtin_amdgcn_s_memrealtime();
  }

  // ---- the lane's value: its group's partial of column i; then one pass over [column][wave][g] -------------------------------------
  float val[MR];
  if constexpr (LVL >= 3) {
#pragma unroll
    for (int m = 0; m < MR; ++m) val[m] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if constexpr (MR > 1) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {   // (register m of the lane's accumulators: batch row m of its group)
          float a = acc[ps][0][m];
#pragma unroll
          for (int ch = 1; ch < NCH; ++ch)
            if (16 * ps + ch < MAXS) a += acc[ps][ch][m];
          val[m] += __builtin_fmaf(sf[ps][0], a, -cf[ps][m]);
        }
      } else {
        float a = acc[ps][0][0], a2 = acc[ps][0][2];   // (rows 4 g, 4 g + 2 of the lane's column: its first / second group of the pass)
#pragma unroll
        for (int ch = 1; ch < NCH; ++ch)
          if (16 * ps + ch < MAXS) { a += acc[ps][ch][0]; a2 += acc[ps][ch][2]; }
        if constexpr (LVL >= 4) {
          val[0] += __builtin_fmaf(sf[ps][0], a, -cf[ps][0]);
          if constexpr (G64) val[0] += __builtin_fmaf(sf[ps][1], a2, -cf[ps][1]);
        } else val[0] += a;
      }
    }
    if constexpr (LVL < 4) val[0] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, val[0]) ^ fold);
  } else {
    val[0] = __builtin_bit_cast(float, fold);
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) lds[(m * 16 + i) * RS + wave * 4 + g] = val[m];
  __syncthreads();
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[4] = __builtin_amdgcn_s_memrealtime();
  }
  float vfin = 0.f;  // (lanes 0..15 of wave 0: the block's 16 outputs; MR = 4: threads 16 m .. 16 m + 15 hold batch row m)
  if (threadIdx.x < 16 * MR) {
    const float4_t *row = (const float4_t *)(lds + threadIdx.x * RS);
    float4_t t[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) t[q] = row[q];
    float v;
    if constexpr (LVL >= 3) {
#pragma unroll
      for (int st = 1; st < NW; st *= 2)
#pragma unroll
        for (int q = 0; q + st < NW; q += 2 * st) t[q] = t[q] + t[q + st];
      v = (t[0][0] + t[0][1]) + (t[0][2] + t[0][3]);
    } else {
      uint32_t f = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) { const uint4_t r = __builtin_bit_cast(uint4_t, t[q]); f ^= r.x ^ r.y ^ r.z ^ r.w; }
      v = (float)(f & 1023u);
    }
    const int n = b * 16 + (threadIdx.x & 15), mrow = threadIdx.x >> 4;
    if (pr.bias) v += (float)pr.bias[n];
    vfin = v;
    if constexpr (!AR) {
      if (MR == 1 || mrow < p.M) {
        const size_t at = (size_t)mrow * pr.n_strips * 16 + n;   // (y is [M][N], N = 16 n_strips)
        if (p.act_bf16) ((uint16_t *)pr.y)[at] = f32_to_bf16(v);
        else ((half_t *)pr.y)[at] = (half_t)v;
      }
    }
  }
  if constexpr (AR) {
    // ---- push: the 16 partial outputs as two 16-byte system-scope stores per peer (lanes 0 and 1 of wave 0, through LDS) ------------
    const int world = p.ar_world, rank = p.ar_rank;
    const size_t slot = p.ar_slot_bytes, payload = 2 * (size_t)world * slot;
    CommCtl *own = (CommCtl *)((char *)p.ar_peers[rank] + payload);
    // (every block reads the epoch before it takes its ticket; the last block bumps it after the last ticket)
    const uint32_t epoch = __hip_atomic_load(&own->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int parity = (int)(epoch & 1u);
    uint16_t *stage = (uint16_t *)(lds + 16 * RS);  // (the first wave's staging area: its activations are long consumed)
    int *s_last = (int *)(lds + 16 * RS) + 16;
    if (wave == 0) {
      if (lane < 16) stage[lane] = p.act_bf16 ? f32_to_bf16(vfin) : __builtin_bit_cast(uint16_t, (half_t)vfin);
      if (lane < 2) {
        const uint4_t chunk = *(const uint4_t *)(stage + 8 * lane);
        for (int q = 0; q < world; ++q) {
          char *dst = (char *)p.ar_peers[q] + ((size_t)parity * world + rank) * slot + ((size_t)b * 16 + 8 * lane) * 2;
          store16_sys(dst, chunk);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left ...
      if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // ... (system scope) before the ticket says so
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t ticket = __hip_atomic_fetch_add(&own->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = (ticket == gridDim.x - 1) ? 1 : 0;
      }
    }
    __syncthreads();
    if (*s_last == 0) return;
    // ---- the rank's last block: every block's slice is in every peer's slot.  Publish, wait for the world, sum in rank order ----------
    if (threadIdx.x == 0) __hip_atomic_store(&own->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm (graph replay)
    if ((int)threadIdx.x < world) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other blocks' tickets (and the stores in front of them)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      CommCtl *peer = (CommCtl *)((char *)p.ar_peers[threadIdx.x] + payload);
      __hip_atomic_store(&peer->flag[parity][rank], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      unsigned spins = 0;
      while (__hip_atomic_load(&own->flag[parity][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 26)) {  // a peer never arrived (seconds): report instead of hanging the GPU
          if (p.ar_status) *p.ar_status = 1;
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    const char *mine = (const char *)p.ar_peers[rank] + (size_t)parity * world * slot;
    const int n16 = (int)gridDim.x * 2;  // 16-byte chunks of the output row (16 columns per block)
    for (int c = threadIdx.x; c < n16; c += NW * 64) {
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < world; ++r) {
        const uint4_t v = load16_sys(mine + (size_t)r * slot + (size_t)c * 16);
        const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (p.act_bf16) {
            a8[2 * j] += __builtin_bit_cast(float, wds[j] << 16);
            a8[2 * j + 1] += __builtin_bit_cast(float, wds[j] & 0xffff0000u);
          } else {
            const half2_t h = as_h2(wds[j]);
            a8[2 * j] += (float)h.x;
            a8[2 * j + 1] += (float)h.y;
          }
        }
      }
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (p.act_bf16) o[j] = (uint32_t)f32_to_bf16(a8[2 * j]) | ((uint32_t)f32_to_bf16(a8[2 * j + 1]) << 16);
        else o[j] = as_u32(half2_t{(half_t)a8[2 * j], (half_t)a8[2 * j + 1]});
      }
      *((uint4_t *)pr.y + c) = uint4_t{o[0], o[1], o[2], o[3]};
    }
    if (threadIdx.x == 0) __hip_atomic_store(&own->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (DBG) {
    if (dbg_slot && lane == 0) dbg_slot[5] = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace qllm
