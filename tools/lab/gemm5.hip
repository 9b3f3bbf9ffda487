// Prefill GEMM v5 (round 5): y[M,N] = x[M,K] . dequant(W4), 256x128x64 block tile, EVERY wave a matrix wave, B fragments built in
// REGISTERS.  The decomposition gemm3 / gemm4 did not test (profiles/r04_prefill_lab.md: every variant there kept dequant waves ->
// ds_write -> barrier -> matrix waves; B stores + the dequant waves' arithmetic + their barrier waits cost 12-27 of 76 us):
//
//   * 4 WM waves as WM (M) x 4 (N): wave tile (256 / WM) x 32 = AM x 1 tiles of v_mfma_f32_32x32x16_f16 (AM = 8 / WM).  The B
//     operand of that MFMA for lane l is 8 consecutive k of ONE column (column l % 32, k half l / 32) -- exactly one packed word of
//     the row-stream / strip-major layouts.  Each wave loads its own words (one dword per lane per 16-k sub-step, requested three
//     k-tiles ahead into four register sets), applies the bit-exact dequant of common.hpp in registers
//     (1 shift + 4 v_and_or, 4 x (v_pk_fma_f16 + v_pk_add_f16), 4 v_perm for the natural k order: 17 VALU per fragment, used by AM
//     MFMAs) and feeds the matrix core: NO dequant waves, NO B tile in LDS (no ds_write, no B ds_read), NO producer / consumer
//     coupling -- the one barrier per k-tile only recycles the activation ring.  The price: the WM waves of a column quarter dequant
//     the same words (WM = 2: twice).
//   * A tiles exactly as gemm3: LDS-DMA pieces (8 rows x 128 B, XOR-swizzled on the source side) into a 4-deep ring, requested three
//     k-tiles ahead by the matrix waves themselves, fragments by ds_read_b128 one sub-step ahead.
//   * every vector-memory operation of a wave is counted: per k-tile it issues, in a fixed order, NP DMA pieces + 4 packed words +
//     1 scale + 1 zero word (all for k-tile t + 3), so ONE s_waitcnt vmcnt(N) per k-tile retires exactly the operations of tile t + 1.
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
constexpr int kATile = BM * BK;  // halves per ring slot (32 KB)
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
__device__ __forceinline__ int tile_off(int row, int slot) { return row * BK + ((slot ^ lds_row_swizzle(row)) & 7) * 8; }  // in halves
}  // namespace g5

#define G5_SB() __builtin_amdgcn_sched_barrier(0)

template <int N>
__device__ __forceinline__ void g5_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N>
__device__ __forceinline__ void g5_wait_vm_lgkm() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory"); }

// ABL (lab builds, timing only -- results are garbage): 1 no activation DMA, 2 no dequant arithmetic (raw words as fragments), 4 no A
// fragment reads in the loop, 8 no packed-word / scale / zero loads
template <int WM, int ABL = 0>
__global__ __launch_bounds__(WM * 256) void gemm5_kernel(const GemmParams p) {
  using namespace g5;
  constexpr int NWV = 4 * WM;          // waves
  constexpr int AM = 8 / WM;           // 32-row MFMA tiles per wave along M: 8 or 4
  constexpr int WROWS = AM * 32;       // rows per wave: 256 or 128
  constexpr int NP = 32 / NWV;         // activation DMA pieces per wave and k-tile: 8 or 4
  constexpr int PPS = NP / 4;          // ... per sub-step: 2 or 1
  constexpr bool NO_DMA = ABL & 1, NO_DQ = ABL & 2, NO_READ = ABL & 4, NO_LOAD = ABL & 8;
  // per k-tile a wave issues NP activation pieces, ONE 1 KB piece of packed words (its 32 columns x 8 word rows: v3 -- four 4-byte
  // loads per lane cost the wave more than a DMA piece, r05_prefill_lab.md) and the scale / zero words
  constexpr int VM_TILE = (NO_DMA ? 0 : NP) + (NO_LOAD ? 0 : 3);      // vector-memory operations a wave issues per k-tile
  // at barrier #t the operations of tile t + 1 (requested during tile t - 2: THREE tiles ahead -- two left the last requests of a
  // batch 3 sub-steps of flight, less than an HBM round trip: 68.6 -> 52 us with the word loads ablated, profiles/r05_prefill_lab.md)
  // are older than tile t - 1's whole batch and the requests of this tile's sub-steps 0..2
  constexpr int VM_WAIT = 2 * VM_TILE - ((NO_DMA ? 0 : PPS) + (NO_LOAD ? 0 : 1));
  extern __shared__ __attribute__((aligned(16))) half_t smem[];
  half_t *As = smem;  // [4][256][64]  (LDS-DMA ring: FOUR slots, so that the slot of every fragment read is a compile-time constant
                      //  in a loop unrolled four k-tiles deep -- two register sets x ... -- and folds into the ds_read's offset field)

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
  const int KT = p.K / BK;  // a multiple of 4 (gemm5_ok)

  const int wm = wave >> 2, wn = wave & 3;   // WM (M) x 4 (N): rows wm * WROWS.., columns wn * 32..
  const int fr = lane & 31, fs = lane >> 5;  // fragment row (A: m, B: n) and k half of the 16-wide sub-step
  const int nB = n0 + wn * 32 + fr;          // this lane's column

  // ---- packed words, scales, zero points: per-lane byte offsets are loop constants, the k-tile / group advance is scalar.  Nothing is
  // clamped: requests past the last k-tile (never consumed) read the neighbouring rows / strips or, past the end of a buffer, zeros
  // (raw buffer bounds) -------------------------------------------------------------------------------------------------------------------
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
  // packed words of a k-tile for this wave: 8 word rows x its 32 columns = 1 KB, ONE LDS-DMA piece into the image [row][32 columns]
  // (lane l carries row l / 8, columns 4 (l % 8) .. + 3: 16 contiguous bytes in the row-stream layouts and -- inside a 16-column
  // strip -- in the strip-major one), four slots per wave behind the activation ring; the lane's word of sub-step ks is
  // image[2 ks + fs][fr]: read with ds_read_b32 at the k-tile barrier
  const int nW0 = n0 + wn * 32 + 4 * (lane & 7);                      // first of the lane's four columns
  const int voff_wd = sm ? (nW0 >> 4) * (p.K >> 3) * 64 + (lane >> 3) * 64 + (nW0 & 15) * 4 : (lane >> 3) * p.N * 4 + nW0 * 4;
  const int ktile_bytes = sm ? 512 : 8 * p.N * 4;                     // advance of a k-tile (strip-major: 8 rows x 64 B of the strip)
  uint32_t *Wl = (uint32_t *)(smem + 4 * kATile) + wave * 1024;       // this wave's four 1 KB slots
  const int w_rd = fs * 32 + fr;                                      // word of sub-step 0 in a slot (sub-step ks: + 64 ks)
  const int srow_bytes = sm ? 32 : p.N * 2;
  const int voff_s = sm ? (nB >> 4) * Gn * 32 + ncs * 2 : nB * 2, voff_z = zoff * 4;
  const uint32_t mask_lo = nib_mask_vgpr(), mask_hi = mask_lo << 4;
  struct BSet {
    uint32_t w[4];
    uint32_t sraw, z;
  };
  BSet bset[4];
  auto dma_words = [&](int kt, int slot) {
    lds_void_t *dst = (lds_void_t *)(Wl + slot * 256);
    if constexpr (!NO_LOAD) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dst, 16, voff_wd, kt * ktile_bytes, 0, 0);
  };
  auto read_words = [&](int slot, BSet &bs) {
    if constexpr (!NO_LOAD) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bs.w[ks] = Wl[slot * 256 + w_rd + 64 * ks];
    }
  };
  auto load_scale = [&](int kt, BSet &bs) {
    if constexpr (!NO_LOAD) bs.sraw = __builtin_amdgcn_raw_buffer_load_b16(rs_s, voff_s, ((kt * BK) >> p.gs_shift) * srow_bytes, 0);
  };
  auto load_zero = [&](int kt, BSet &bs) {
    if constexpr (!NO_LOAD) bs.z = __builtin_amdgcn_raw_buffer_load_b32(rs_z, voff_z, ((kt * BK) >> p.gs_shift) * zmul * 4, 0);
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
    const int so = kt * (BK * 2);
    const int vo = voff_x[q];
    lds_void_t *dst = (lds_void_t *)(As + slot * kATile + (wave * rows_per_wave + q * 8) * BK);
    if constexpr (!NO_DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
  };

  // ---- A fragment addresses (halves): row = wm WROWS + 32 a + fr; the XOR swizzle depends on a only through its parity, and the ring
  // slot and a / 2 are compile-time constants at every read: 2 x 4 address registers per ring half (slots 0-1 / 2-3: the ds_read
  // offset field holds 64 KB), everything else in the instruction's immediate -- no address arithmetic in the loop
  int va[2][2][4];
#pragma unroll
  for (int pa = 0; pa < 2; ++pa)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      va[0][pa][ks] = tile_off(wm * WROWS + pa * 32 + fr, ks * 2 + fs);
      va[1][pa][ks] = va[0][pa][ks] + 2 * kATile;
    }

  float16_t acc[AM];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  half8_t fa0[AM], fa1[AM], fb0, fb1;
#define G5_READ(fa, SLOT, KS, a) fa[a] = *(const half8_t *)(As + (va[(SLOT) >> 1][(a) & 1][KS] + ((SLOT) & 1) * kATile + ((a) >> 1) * (64 * BK)))

  // ---- the dequant of one word in eight steps of 2-3 VALU, placed between MFMAs.  Extraction without the odd nibbles' shift: pairs 1, 3
  // are 1024 + 16 q; fma(1024 + 16 q, s, -1024 s) = 16 fp16(q s) exactly (a power-of-two multiple rounds like its base), and the second
  // op, fma(t, 1/16, -zs), rounds fp16(q s) - fp16(z s) once: the reference's W bit for bit (common.hpp, deq_pair, does the same for
  // pairs 0, 2 with an add as second op) --------------------------------------------------------------------------------------------------
  uint32_t w8, e0, e1, e2, e3;
  half2_t d0, d1, d2, d3;
  uint32_t r0, r1, r2, r3;
  const half2_t k16 = {(half_t)0.0625f, (half_t)0.0625f};
  auto dq_step = [&](int st, uint32_t w, const ColConst &c, half8_t &fb) {
    if constexpr (NO_DQ) {
      if (st == 7) fb = __builtin_bit_cast(half8_t, uint4_t{w, w, w, w});
      return;
    }
    switch (st) {
      case 0: w8 = w >> 8; e0 = and_or(w, mask_lo, kMagic); e1 = and_or(w, mask_hi, kMagic); break;
      case 1: e2 = and_or(w8, mask_lo, kMagic); e3 = and_or(w8, mask_hi, kMagic); break;
      // (the two ops of a pair sit in different steps: back to back they cost a hazard s_nop each)
      case 2: d0 = __builtin_elementwise_fma(as_h2(e0), c.s2, c.c2); d1 = __builtin_elementwise_fma(as_h2(e1), c.s2, c.c2); break;
      case 3: d0 = d0 - c.zs2; d1 = __builtin_elementwise_fma(d1, k16, -c.zs2); break;
      case 4: d2 = __builtin_elementwise_fma(as_h2(e2), c.s2, c.c2); d3 = __builtin_elementwise_fma(as_h2(e3), c.s2, c.c2); break;
      case 5: d2 = d2 - c.zs2; d3 = __builtin_elementwise_fma(d3, k16, -c.zs2); break;
      case 6: r0 = __builtin_amdgcn_perm(as_u32(d1), as_u32(d0), 0x05040100u); r1 = __builtin_amdgcn_perm(as_u32(d3), as_u32(d2), 0x05040100u); break;
      default:
        r2 = __builtin_amdgcn_perm(as_u32(d1), as_u32(d0), 0x07060302u); r3 = __builtin_amdgcn_perm(as_u32(d3), as_u32(d2), 0x07060302u);
        fb = __builtin_bit_cast(half8_t, uint4_t{r0, r1, r2, r3});
    }
  };

  // ---- prologue: tiles 0 and 1 requested, tile 0 landed, first fragments built ---------------------------------------------------------
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(0, 0, q);
  dma_words(0, 0);
  load_scale(0, bset[0]);
  load_zero(0, bset[0]);
  G5_SB();  // (the batches stay in this order: hipcc's own vmcnt for a register is the minimum over the paths into the loop)
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(1, 1, q);
  dma_words(1, 1);
  load_scale(1, bset[1]);
  load_zero(1, bset[1]);
  G5_SB();
#pragma unroll
  for (int q = 0; q < NP; ++q) dma_piece(2, 2, q);
  dma_words(2, 2);
  load_scale(2, bset[2]);
  load_zero(2, bset[2]);
  G5_SB();
  g5_wait_vm<2 * VM_TILE>();  // tile 0's operations have completed (tiles 1, 2 in flight)
  __builtin_amdgcn_s_barrier();
  read_words(0, bset[0]);
  if constexpr (NO_LOAD) {
    bset[0] = BSet{{0x12345678u, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u}, 0x2000u, 0x77777777u};
    bset[1] = bset[0]; bset[2] = bset[0]; bset[3] = bset[0];
  }
  ColConst cc = col_const(bset[0]);
#pragma unroll
  for (int a = 0; a < AM; ++a) G5_READ(fa0, 0, 0, a);
  if constexpr (NO_READ) {
#pragma unroll
    for (int a = 0; a < AM; ++a) G5_READ(fa1, 0, 1, a);
  }
#pragma unroll
  for (int st = 0; st < 8; ++st) dq_step(st, bset[0].w[0], cc, fb0);
  G5_SB();

  // One sub-step: the AM MFMAs of (fa_c, fb_c) with, between them, the reads of the next sub-step's A fragments (slot SA_N, sub-step
  // KS_N), the dequant steps of its B fragment (word wn_, constants ccn_) and this sub-step's vector-memory requests (tile kt_ + 2:
  // DMA pieces into slot SLOT_REQ, word KS_ of register set set_, scale / zero behind sub-steps 2 / 3).
#define G5_SUBSTEP(fa_c, fb_c, fa_n, fb_n, SA_N, KS_N, wn_, ccn_, kt_, set_, KS_, SLOT_REQ)                                              \
  {                                                                                                                                     \
    _Pragma("unroll") for (int a = 0; a < AM; ++a) {                                                                                    \
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa_c[a], fb_c, acc[a], 0, 0, 0);                                                  \
      G5_SB();                                                                                                                          \
      if constexpr (!NO_READ) G5_READ(fa_n, SA_N, KS_N, a);                                                                             \
      if constexpr (AM == 8) dq_step(a, wn_, ccn_, fb_n);                                                                               \
      else { dq_step(2 * a, wn_, ccn_, fb_n); dq_step(2 * a + 1, wn_, ccn_, fb_n); }                                                    \
      if (a == AM / 8) dma_piece((kt_) + 3, SLOT_REQ, PPS * (KS_));                                                                     \
      if (PPS == 2 && a == 5) dma_piece((kt_) + 3, SLOT_REQ, PPS * (KS_) + 1);                                                          \
      if (a == AM / 2 - 1 && (KS_) == 0) dma_words((kt_) + 3, SLOT_REQ);                                                                \
      if (a == AM - 1 && (KS_) == 2) load_scale((kt_) + 3, set_);                                                                       \
      if (a == AM - 1 && (KS_) == 3) load_zero((kt_) + 3, set_);                                                                        \
      G5_SB();                                                                                                                          \
    }                                                                                                                                   \
  }
  // One k-tile: tile kt in ring slot SA (register set CUR), next tile in slot SA1 (set NXT), requests for tile kt + 3 into slot SA3 /
  // set REQ (= the slot and set of tile kt - 1: all read).
#define G5_TILE(kt_, CUR, NXT, REQ, SA, SA1, SA3)                                                                                       \
  {                                                                                                                                     \
    G5_SUBSTEP(fa0, fb0, fa1, fb1, SA, 1, CUR.w[1], cc, kt_, REQ, 0, SA3)                                                               \
    G5_SUBSTEP(fa1, fb1, fa0, fb0, SA, 2, CUR.w[2], cc, kt_, REQ, 1, SA3)                                                               \
    G5_SUBSTEP(fa0, fb0, fa1, fb1, SA, 3, CUR.w[3], cc, kt_, REQ, 2, SA3)                                                               \
    /* barrier #kt: my fragment reads of tile kt are complete, my requests for tile kt+1 have landed (only those issued during    */   \
    /* sub-steps 0..2 of this tile are younger).  After it: ring slot SA is free, tile kt+1 is complete.                          */   \
    g5_wait_vm_lgkm<VM_WAIT>();                                                                                                         \
    __builtin_amdgcn_s_barrier();                                                                                                       \
    G5_SB();                                                                                                                            \
    read_words(SA1, NXT);                                                                                                               \
    ccn = col_const(NXT);                                                                                                               \
    G5_SB();                                                                                                                            \
    G5_SUBSTEP(fa1, fb1, fa0, fb0, SA1, 0, NXT.w[0], ccn, kt_, REQ, 3, SA3)                                                             \
    cc = ccn;                                                                                                                           \
  }

  ColConst ccn = cc;
  for (int kt = 0; kt < KT; kt += 4) {
    G5_TILE(kt, bset[0], bset[1], bset[3], 0, 1, 3)
    G5_TILE(kt + 1, bset[1], bset[2], bset[0], 1, 2, 0)
    G5_TILE(kt + 2, bset[2], bset[3], bset[1], 2, 3, 1)
    G5_TILE(kt + 3, bset[3], bset[0], bset[2], 3, 0, 2)
  }
#undef G5_TILE
#undef G5_SUBSTEP
#undef G5_READ
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
  if (p.K % 256 != 0 || p.N % 128 != 0 || p.gs_shift < 6 || p.group_size % 64 != 0) return false;  // (the loop runs four k-tiles per trip)
  if ((size_t)p.M * p.K * 2 >= 0x7fffffffull || (size_t)p.K * p.N / 2 >= 0x7fffffffull) return false;
  return true;
}

template <int WM, int ABL = 0>
static int launch_gemm5_t(const GemmParams &p, hipStream_t stream) {
  using namespace g5;
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)gemm5_kernel<WM, ABL>)) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  const size_t lds = (size_t)(4 * kATile) * sizeof(half_t) + (size_t)WM * 4 * 4096;  // 128 KB + the waves' packed-word slots
  hipLaunchKernelGGL((gemm5_kernel<WM, ABL>), dim3(tiles), dim3(WM * 256), lds, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_gemm5(const GemmParams &p, int wm, hipStream_t stream) {
#ifdef QLLM_LAB
  switch (wm == 1 ? knob("QLLM_G5_ABL", 0) : 0) {  // (timing-only ablations of the one-wave-per-SIMD form: profiles/r05_prefill_lab.md)
    case 1: return launch_gemm5_t<1, 1>(p, stream);
    case 2: return launch_gemm5_t<1, 2>(p, stream);
    case 3: return launch_gemm5_t<1, 3>(p, stream);
    case 4: return launch_gemm5_t<1, 4>(p, stream);
    case 8: return launch_gemm5_t<1, 8>(p, stream);
    case 10: return launch_gemm5_t<1, 10>(p, stream);
    case 11: return launch_gemm5_t<1, 11>(p, stream);
    case 15: return launch_gemm5_t<1, 15>(p, stream);
    default: break;
  }
#endif
  return wm == 1 ? launch_gemm5_t<1>(p, stream) : launch_gemm5_t<2>(p, stream);
}

}  // namespace qllm
