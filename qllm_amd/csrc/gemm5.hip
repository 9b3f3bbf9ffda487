// Prefill GEMM v5 (round 5): y[M,N] = x[M,K] . dequant(W4), 256x128x64 block tile, EVERY wave a matrix wave, B fragments built in
// REGISTERS.  The decomposition gemm3 / gemm4 did not test (profiles/r04_prefill_lab.md: every variant there kept dequant waves ->
// ds_write -> barrier -> matrix waves; B stores + the dequant waves' arithmetic + their barrier waits cost 12-27 of 76 us):
//
//   * 4 WM waves as WM (M) x 4 (N): wave tile (256 / WM) x 32 = AM x 1 tiles of v_mfma_f32_32x32x16_f16 (AM = 8 / WM).  The B
//     operand of that MFMA for lane l is 8 consecutive k of ONE column (column l % 32, k half l / 32) -- exactly one packed word of
//     the row-stream / strip-major layouts.  Each wave loads its own words (one dword per lane per 16-k sub-step, requested two
//     k-tiles ahead into two register sets), applies the bit-exact dequant of common.hpp in registers
//     (1 shift + 4 v_and_or, 4 x (v_pk_fma_f16 + v_pk_add_f16), 4 v_perm for the natural k order: 17 VALU per fragment, used by AM
//     MFMAs) and feeds the matrix core: NO dequant waves, NO B tile in LDS (no ds_write, no B ds_read), NO producer / consumer
//     coupling -- the one barrier per k-tile only recycles the activation ring.  The price: the WM waves of a column quarter dequant
//     the same words (WM = 2: twice).
//   * A tiles exactly as gemm3: LDS-DMA pieces (8 rows x 128 B, XOR-swizzled on the source side) into a 3-deep ring, requested two
//     k-tiles ahead by the matrix waves themselves, fragments by ds_read_b128 one sub-step ahead.
//   * every vector-memory operation of a wave is counted: per k-tile it issues, in a fixed order, NP DMA pieces + 4 packed words +
//     1 scale + 1 zero word (all for k-tile t + 2), so ONE s_waitcnt vmcnt(N) per k-tile retires exactly the operations of tile t + 1.
//   * the issue order is pinned by hand (sched_barrier): one MFMA, then ~4 VALU of the next fragment's dequant and one fragment
//     read, so that the two waves of a SIMD alternate between a matrix burst and a VALU burst.
// W is bit-identical to gemm3's (same dequant ops, same k order inside the MFMAs): GEMM3_CASES pass unchanged.
// Serves: 4-bit row-stream (GPTQ / HQQ) and native strip-major layers, group size >= 64, K % 64 == 0, N % 128 == 0, no split-K
// (tiles >= CUs), fp16 activations (bf16 through gemm3's conversion pre-pass + out_bf16).
// Replaces gemm_forward_4bit_cuda_m16n128k32 (/root/reference/csrc/awq_cuda/quantization/gemm_cuda_gen.cu:31-353) + dequantize.cuh:15-78.
#include <stdlib.h>

#include "kernels.hpp"

namespace qllm {

namespace g5 {
constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kATile = BM * BK;  // halves per ring slot
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + ((slot ^ lds_row_swizzle(row)) & 7) * 8; }  // in halves
}  // namespace g5

#define G5_SB() __builtin_amdgcn_sched_barrier(0)

template <int WM>
__global__ __launch_bounds__(WM * 256) void gemm5_kernel(const GemmParams p) {
  using namespace g5;
  constexpr int NWV = 4 * WM;          // waves
  constexpr int AM = 8 / WM;           // 32-row MFMA tiles per wave along M: 8 or 4
  constexpr int WROWS = AM * 32;       // rows per wave: 256 or 128
  constexpr int NP = 32 / NWV;         // activation DMA pieces per wave and k-tile: 8 or 4
  constexpr int PPS = NP / 4;          // ... per sub-step: 2 or 1
  constexpr int VM_TILE = NP + 6;      // vector-memory operations a wave issues per k-tile
  constexpr int VM_WAIT = VM_TILE - (PPS + 2);  // ... of which those of sub-steps 0..2 are younger than tile t+1's when it is needed
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;  // [3][256][64]  (LDS-DMA ring)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int nblk = tiles_m * tiles_n;
  int bid = blockIdx.x;
  {  // each XCD (block id % 8) walks a contiguous run of tiles
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = p.raster ? (bid / tiles_n) : (bid % tiles_m);
  const int tn = p.raster ? (bid % tiles_n) : (bid / tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = p.K / BK;

  const int wm = wave >> 2, wn = wave & 3;   // WM (M) x 4 (N): rows wm * WROWS.., columns wn * 32..
  const int fr = lane & 31, fs = lane >> 5;  // fragment row (A: m, B: n) and k half of the 16-wide sub-step
  const int nB = n0 + wn * 32 + fr;          // this lane's column

  // ---- packed words, scales, zero points: per-lane byte offsets are loop constants, the k-tile / group advance is scalar ------------
  const int zk = p.zero_kind;
  const uint32_t *zbase = (zk == ZK_SYM) ? (const uint32_t *)p.scales : (const uint32_t *)p.qzeros;
  const int Gn = p.n_groups;
  const bool sm = p.sm;
  const int ncs = sm ? (nB & 15) : nB;  // column index inside a row of the scale / zero tables
  const int zmul_all = (zk == ZK_PACKED) ? (p.N >> 3) : (p.N >> 1);  // words per group over all columns
  const int zmul = sm ? ((zk == ZK_PACKED) ? 2 : 8) : zmul_all;
  const int zoff = ((zk == ZK_PACKED) ? (ncs >> 3) : (ncs >> 1)) + (sm ? (nB >> 4) * Gn * zmul : 0);
  const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.qweight, 0, (int)((size_t)p.K * p.N / 2), 0x00020000);
  const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void *)p.scales, 0, Gn * p.N * 2, 0x00020000);
  const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((void *)zbase, 0, (sm ? (p.N >> 4) * Gn * zmul : Gn * zmul_all) * 4, 0x00020000);
  const int wrow_bytes = sm ? 64 : p.N * 4;  // bytes per packed word row (strip-major: the strip's 16 words)
  const int ktile_bytes = 8 * wrow_bytes;    // 8 word rows per k-tile
  // word row 8 kt + 2 ks + fs of column nB
  int voff_w[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) voff_w[ks] = (2 * ks + fs) * wrow_bytes + (sm ? (nB >> 4) * (p.K >> 3) * 64 + ncs * 4 : nB * 4);
  const int srow_bytes = sm ? 32 : p.N * 2;
  const int voff_s = sm ? (nB >> 4) * Gn * 32 + ncs * 2 : nB * 2, voff_z = zoff * 4;
  const uint32_t nibmask = nib_mask_vgpr();
  struct BSet {
    uint32_t w[4];
    uint32_t sraw, z;
  };
  BSet bset[2];
  auto load_word = [&](int kt, BSet &bs, int ks) {
    const int so = min(kt, KT - 1) * ktile_bytes;
    bs.w[ks] = __builtin_amdgcn_raw_buffer_load_b32(rs_w, voff_w[ks], so, 0);
  };
  auto load_scale = [&](int kt, BSet &bs) {
    const int G = (min(kt, KT - 1) * BK) >> p.gs_shift;
    bs.sraw = __builtin_amdgcn_raw_buffer_load_b16(rs_s, voff_s + G * srow_bytes, 0, 0);
  };
  auto load_zero = [&](int kt, BSet &bs) {
    const int G = (min(kt, KT - 1) * BK) >> p.gs_shift;
    bs.z = __builtin_amdgcn_raw_buffer_load_b32(rs_z, voff_z + G * zmul * 4, 0, 0);
  };
  auto col_const = [&](const BSet &bs) {
    const half_t zp = (half_t)(float)(((bs.z >> (4 * (nB & 7))) + (uint32_t)p.add_zero_bias) & 15u);
    const half_t zf = __builtin_bit_cast(half_t, (uint16_t)((nB & 1) ? (bs.z >> 16) : (bs.z & 0xffffu)));
    const half_t sc = __builtin_bit_cast(half_t, (uint16_t)bs.sraw);
    return make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)8.f));
  };

  // ---- activation tile by LDS-DMA: this wave owns rows wave * 8 NP .. of the 256-row tile = NP pieces of 8 rows x 128 B -----------
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)min((size_t)p.M * p.K * 2, (size_t)0x7fffffff), 0x00020000);
  int voff_x[NP];
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const int r = wave * (8 * NP) + 8 * q + (lane >> 3);
    const int grow = min(m0 + r, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
    voff_x[q] = grow * p.K * 2 + (((lane & 7) ^ lds_row_swizzle(r)) << 4);
  }
  const int rows_per_wave = 8 * NP;
  auto dma_piece = [&](int kt, int slot, int q) {
    const int so = min(kt, KT - 1) * (BK * 2);
    const int vo = voff_x[q];
    lds_void_t *dst = (lds_void_t *)(As + slot * kATile + (wave * rows_per_wave + q * 8) * BK);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
  };

  float16_t acc[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8_t fa0[AM], fa1[AM], fb0, fb1;
  auto read_frag = [&](int sa, int ks, int a, half8_t (&fa)[AM]) {
    fa[a] = *(const half8_t *)(As + sa * kATile + tile_off(wm * WROWS + a * 32 + fr, ks * 2 + fs));
  };

  // ---- the dequant of one word in four stages of 4-5 VALU, so that they can be placed between MFMAs --------------------------------
  uint32_t e0, e1, e2, e3;   // extraction: magic pairs (k0,k4) (k1,k5) (k2,k6) (k3,k7)
  half2_t d0, d1, d2, d3;    // dequantised pairs, same order
  auto stage_extract = [&](uint32_t w) {
    const uint32_t w4 = w >> 4;
    e0 = and_or(w, nibmask, kMagic); e1 = and_or(w4, nibmask, kMagic);
    e2 = and_or(w >> 8, nibmask, kMagic); e3 = and_or(w4 >> 8, nibmask, kMagic);
  };
  auto stage_deq01 = [&](const ColConst &cc) { d0 = deq_pair(e0, cc); d1 = deq_pair(e1, cc); };
  auto stage_deq23 = [&](const ColConst &cc) { d2 = deq_pair(e2, cc); d3 = deq_pair(e3, cc); };
  auto stage_perm = [&](half8_t &fb) {  // (k0,k4,k1,k5,k2,k6,k3,k7) -> natural k order
    const uint32_t a = as_u32(d0), b = as_u32(d1), c = as_u32(d2), d = as_u32(d3);
    const uint32_t r0 = __builtin_amdgcn_perm(b, a, 0x05040100u);  // (k0, k1)
    const uint32_t r1 = __builtin_amdgcn_perm(d, c, 0x05040100u);  // (k2, k3)
    const uint32_t r2 = __builtin_amdgcn_perm(b, a, 0x07060302u);  // (k4, k5)
    const uint32_t r3 = __builtin_amdgcn_perm(d, c, 0x07060302u);  // (k6, k7)
    fb = __builtin_bit_cast(half8_t, uint4_t{r0, r1, r2, r3});
  };

  // ---- prologue: tiles 0 and 1 requested, tile 0 landed, first fragments built ---------------------------------------------------------
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(0, 0, q);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) load_word(0, bset[0], ks);
  load_scale(0, bset[0]);
  load_zero(0, bset[0]);
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(1, 1, q);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) load_word(1, bset[1], ks);
  load_scale(1, bset[1]);
  load_zero(1, bset[1]);
  G5_SB();
  if constexpr (VM_TILE == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");  // tile 0's operations have completed (tile 1's in flight)
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  ColConst cc = col_const(bset[0]);
#pragma unroll
  for (int a = 0; a < AM; ++a) read_frag(0, 0, a, fa0);
  stage_extract(bset[0].w[0]);
  stage_deq01(cc);
  stage_deq23(cc);
  stage_perm(fb0);
  G5_SB();

  // One sub-step: the AM MFMAs of (fa_c, fb_c) with, between them, the reads of the next sub-step's A fragments, the four dequant
  // stages of its B fragment (word wn_, constants ccn_) and this sub-step's vector-memory requests.
  // kt_: the tile being computed; sub-step KS_; SA_N: ring slot of the next fragments; KS_N: their sub-step.
#define G5_SUBSTEP(fa_c, fb_c, fa_n, fb_n, SA_N, KS_N, wn_, ccn_, kt_, set_, KS_, SLOT_REQ)                                              \
  {                                                                                                                                     \
    _Pragma("unroll") for (int a = 0; a < AM; ++a) {                                                                                    \
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_c[a], fb_c, acc[a], 0, 0, 0);                                                  \
      G5_SB();                                                                                                                          \
      read_frag(SA_N, KS_N, a, fa_n);                                                                                                   \
      if (a == 0) stage_extract(wn_);                                                                                                   \
      if (a == 1 * (AM / 4)) stage_deq01(ccn_);                                                                                         \
      if (a == 2 * (AM / 4)) stage_deq23(ccn_);                                                                                         \
      if (a == 3 * (AM / 4)) stage_perm(fb_n);                                                                                          \
      if (a == 0) dma_piece((kt_) + 2, SLOT_REQ, PPS * (KS_));                                                                          \
      if (PPS == 2 && a == AM / 2) dma_piece((kt_) + 2, SLOT_REQ, PPS * (KS_) + 1);                                                     \
      if (a == AM - 1) {                                                                                                                \
        load_word((kt_) + 2, set_, KS_);                                                                                                \
        if ((KS_) == 2) load_scale((kt_) + 2, set_);                                                                                    \
        if ((KS_) == 3) load_zero((kt_) + 2, set_);                                                                                     \
      }                                                                                                                                 \
      G5_SB();                                                                                                                          \
    }                                                                                                                                   \
  }
  // One k-tile: tile kt in ring slot SA (register set CUR), next tile in slot SA1 (set NXT), requests for tile kt + 2 into slot SA2 /
  // set CUR (each word of CUR is re-requested after the sub-step that consumed it).
#define G5_TILE(kt_, CUR, NXT, SA, SA1, SA2)                                                                                            \
  {                                                                                                                                     \
    G5_SUBSTEP(fa0, fb0, fa1, fb1, SA, 1, CUR.w[1], cc, kt_, CUR, 0, SA2)                                                               \
    G5_SUBSTEP(fa1, fb1, fa0, fb0, SA, 2, CUR.w[2], cc, kt_, CUR, 1, SA2)                                                               \
    G5_SUBSTEP(fa0, fb0, fa1, fb1, SA, 3, CUR.w[3], cc, kt_, CUR, 2, SA2)                                                               \
    /* barrier #kt: my fragment reads of tile kt are complete, my requests for tile kt+1 have landed (only those issued during    */   \
    /* sub-steps 0..2 of this tile are younger).  After it: ring slot SA is free for tile kt+3, tile kt+1 is complete.           */   \
    if constexpr (VM_WAIT == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");                                           \
    else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");                                                                    \
    __builtin_amdgcn_s_barrier();                                                                                                       \
    G5_SB();                                                                                                                            \
    ccn = col_const(NXT);                                                                                                               \
    G5_SB();                                                                                                                            \
    G5_SUBSTEP(fa1, fb1, fa0, fb0, SA1, 0, NXT.w[0], ccn, kt_, CUR, 3, SA2)                                                             \
    cc = ccn;                                                                                                                           \
  }

  ColConst ccn = cc;
  // three tiles per trip would keep the ring slots static; two keep the register sets static: the slots rotate in scalars
  int sa = 0;
  for (int kt = 0; kt < KT; kt += 2) {
    const int sa1 = (sa == 2) ? 0 : sa + 1, sa2 = (sa == 0) ? 2 : sa - 1;
    G5_TILE(kt, bset[0], bset[1], sa, sa1, sa2)
    G5_TILE(kt + 1, bset[1], bset[0], sa1, sa2, sa)
    sa = sa2;
  }
#undef G5_TILE
#undef G5_SUBSTEP
  __builtin_amdgcn_s_setprio(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // stray requests past the last tile: land before the LDS is reused
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: + bias, round once, transpose through wave-private LDS, 16-byte row-contiguous stores -------------------------------
  // C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  half_t *ep = smem + wave * (32 * 40);  // 32 rows x 32 cols, row stride 40 halves (80 B)
  const float bv = p.bias ? (float)p.bias[nB] : 0.f;
#pragma unroll
  for (int a = 0; a < AM; ++a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * fs;
      const float v = acc[a][r] + bv;
      if (p.out_bf16) ((uint16_t *)ep)[row * 40 + fr] = f32_to_bf16((float)(half_t)v);
      else ep[row * 40 + fr] = (half_t)v;
    }
    // 32 rows x 64 B = 128 chunks of 16 B: 2 per lane
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = lane + 64 * h, row = c >> 2, ch = c & 3;
      const uint4_t v = *(const uint4_t *)(ep + row * 40 + ch * 8);
      const int m = m0 + wm * WROWS + a * 32 + row;
      if (m < p.M) *(uint4_t *)((half_t *)p.y + (size_t)m * p.N + n0 + wn * 32 + ch * 8) = v;
    }
  }
}
#undef G5_SB

bool gemm5_ok(const GemmParams &p, int layout) {
  // what gemm3 serves unsplit, minus AWQ in place (its words hold 8 columns of one k: not a B fragment), 3 bits and 32-wide groups
  if (layout != QLLM_LAYOUT_GPTQ || p.g_idx || p.act_bf16 || p.split_k > 1) return false;
  if (p.K % 128 != 0 || p.N % 128 != 0 || p.gs_shift < 6 || p.group_size % 64 != 0) return false;
  if ((size_t)p.M * p.K * 2 >= 0x7fffffffull || (size_t)p.K * p.N / 2 >= 0x7fffffffull) return false;
  return true;
}

template <int WM>
static int launch_gemm5_t(const GemmParams &p, hipStream_t stream) {
  using namespace g5;
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)gemm5_kernel<WM>)) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  const size_t lds = (size_t)(3 * kATile) * sizeof(half_t);  // 96 KB
  hipLaunchKernelGGL((gemm5_kernel<WM>), dim3(tiles), dim3(WM * 256), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_gemm5(const GemmParams &p, int wm, hipStream_t stream) {
  return wm == 1 ? launch_gemm5_t<1>(p, stream) : launch_gemm5_t<2>(p, stream);
}

}  // namespace qllm
