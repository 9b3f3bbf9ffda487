// Prefill GEMM v4: y[M,N] = x[M,K] . dequant(W), 256x128x64 block tile, MATRIX waves + PRODUCER waves, everything that enters
// the CU arrives by LDS-DMA.
//
// What gemm3 (the round-2/3 prefill kernel) left on the table (profiles/r03_prefill_summary.md: matrix pipe 50 % busy): its
// matrix waves issued the activation tile's DMA pieces themselves -- a piece costs the issuing wave 60-180 cycles beside
// MFMAs (MI355X_MICROARCH.md, LDS-DMA piece issue cost), 8 (4) of them per 32 (16) MFMAs -- and their fragment addresses were
// recomputed by VALU in every sub-step.  Here:
//   * matrix waves (MW = 4: one per SIMD, 128x64 outputs each, or 8: two per SIMD, 64x64 each) issue NOTHING but ds_read_b128
//     and v_mfma_f32_32x32x16_f16: fragment addresses are one VGPR per k16 sub-step (per-lane swizzle folded in once per
//     kernel, ring slot added once per k-tile) + immediates; reads of sub-step s+1 are slotted one behind each MFMA of
//     sub-step s (never a burst), so a read is >= 2 MFMAs (64 cycles of matrix-pipe work) old when its fragment is used.
//   * producer waves (4, one per SIMD) own ALL global traffic, and all of it is LDS-DMA (buffer_load ... lds), so no global
//     value ever sits in a register and every wait is a hand-counted s_waitcnt vmcnt(N) (the compiler sees no VGPR load to
//     wait for: cdna_hip_programming.md 5, ".s-level traps" (b)):
//       - the ACTIVATION tile: 8 pieces of 8 rows x 128 B per wave and k-tile, requested two k-tiles ahead into a 3-deep ring,
//         XOR-swizzled on the source address (rule 21);
//       - the wave's own PACKED words of the weight tile (1 KB), its columns' scales and zero points (raw, 2 small pieces),
//         requested four k-tiles ahead into a 4-deep ring that only this wave reads back (no cross-wave hand-off, so its own
//         vmcnt is the only ordering needed); then ds_read -> the bit-exact 3-op fp16 dequant (common.hpp) -> ds_write_b128
//         into the double-buffered B tile.
//   * one workgroup barrier per k-tile, placed INSIDE the matrix waves' last sub-step (after BP of its MFMAs, so the pipe has
//     queued work while the barrier resolves):
//         producer, iteration t:  [request A tile t+2, raw tile t+4] vmcnt(14) [raw t+1 -> dequant -> B stage (t+1)%2]   barrier #t
//         matrix,   iteration t:  sub-steps 0..2 of tile t, BP MFMAs of sub-step 3, lgkmcnt(0)                          barrier #t
//                                 [rest of sub-step 3 beside the reads of tile t+1's first fragments]
//     after barrier #t: A slot t%3 and B stage t%2 are free, A tile t+1 and B tile t+1 are complete.
//     vmcnt(14): a producer's queue is ... A(t+1) raw(t+3) | A(t+2) raw(t+4): everything up to A(t+1) has landed when at most
//     3 + 8 + 3 requests are outstanding (loads complete in order); raw(t+1) is older still (two k-tiles of flight).
//   * LDS: A 3 x 32 KB + B 2 x 16 KB + raw 4 x 4 x 1.5 KB = 152 KB: one block per CU.  Tile rows are 64 halves (128 B) with the
//     eight 16-byte slots XORed by g4::swz(row) -- a function of row % 32 only, so a lane's swizzle is the same for every
//     32-row fragment it reads; conflict-free for the 32x32x16 fragment reads, the dequant stores (8 consecutive rows at one
//     slot) and (inherently) the lane-linear DMA pieces (tools/lab/bank_sim.py).
// Layouts: GPTQ / HQQ row stream (4 and 3 bits) and AWQ in place, and the native strip-major storage of the row-stream
// layouts (p.sm).  fp16 activations (bf16 through launch_bf16_to_f16 + out_bf16, as gemm3).  Split-K as gemm3.
// Requires K % 64 == 0, N % 128 == 0, power-of-two group size >= 32, no g_idx.
// Replaces gemm_forward_4bit_cuda_m16n128k32 (/root/reference/csrc/awq_cuda/quantization/gemm_cuda_gen.cu:31-353).
#include "kernels.hpp"

namespace qllm {

namespace g4 {
constexpr int BM = 256, BN = 128, BK = 64;
constexpr int kATileB = BM * BK * 2, kBTileB = BN * BK * 2;  // bytes per stage
constexpr int kRing = 3, kRawSlots = 5;
constexpr int kRawW = 1024, kRawS = 256, kRawZ = 256, kRawWave = kRawW + kRawS + kRawZ;
constexpr int kRawSlotB = 4 * kRawWave;  // (8 dequant waves: two of them share a 1536-byte region, 768 B each)
constexpr int kAOff = 0, kBOff = kRing * kATileB, kRawOff = kBOff + 2 * kBTileB;
constexpr int kLdsBytes = kRawOff + kRawSlots * kRawSlotB;  // 161792
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) | (((row ^ (row >> 4)) & 1) << 2); }
__device__ __forceinline__ int tile_off_b(int row, int slot) { return row * 128 + (((slot ^ swz(row)) & 7) << 4); }  // bytes
}  // namespace g4

#define G4_SB() __builtin_amdgcn_sched_barrier(0)
#ifdef QLLM_LAB
#define G4_STAMP(slot) do { if (dbg && lane == 0) dbg[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#define G4_RSTAMP(slot) do { if (dbg && lane == 0) dbg[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define G4_STAMP(slot) do { } while (0)
#define G4_RSTAMP(slot) do { } while (0)
#endif

// One k16 sub-step of a matrix wave: MFMAs [I0, I1) of the AM x 2 (order a0b0 a0b1 a1b0 a1b1 ...) on the A fragments `a` and the
// B fragments `bc`, and behind them the fragment reads of the NEXT sub-step.  A fragments are replaced IN PLACE (a_j is dead
// behind MFMA 2j+1: its successor is read right there, into the same registers), B fragments alternate between two sets.
// R0: the first MFMA that may carry reads (the ones in front of it belong to a k-tile whose stages are not yet published):
//   behind MFMA R0: b0' and every a_j' already dead; behind R0+1: b1'; behind every later odd MFMA 2j+1: a_j'.
template <int AM, int I0, int I1, int R0, bool READS>
__device__ __forceinline__ void g4_substep(g4::float16_t (&acc)[AM][2], half8_t (&a)[AM], const half8_t (&bc)[2], half8_t (&bn)[2], const char *pa,
                                           const char *pb) {
#pragma unroll
  for (int i = I0; i < I1; ++i) {
    acc[i >> 1][i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i >> 1], bc[i & 1], acc[i >> 1][i & 1], 0, 0, 0);
    if (READS && i >= R0) {
      if (i == R0) {
        bn[0] = *(const half8_t *)pb;
#pragma unroll
        for (int j = 0; j < AM; ++j)
          if (2 * j + 1 < R0) a[j] = *(const half8_t *)(pa + j * 4096);
      }
      if (i == R0 + 1) bn[1] = *(const half8_t *)(pb + 4096);
      if ((i & 1) && i >= R0) a[i >> 1] = *(const half8_t *)(pa + (i >> 1) * 4096);
    }
    G4_SB();
  }
}

// LAYOUT 0 = GPTQ / HQQ row stream (4 bits), 1 = AWQ GEMM, 2 = row stream with 3-bit weights; p.sm: strip-major storage.
// Waves: MW matrix (4 or 8) + DW dequant (4, or 8 for LAYOUT 0: half a k-tile's words per thread) + LD activation loaders (4, or
// 0: the dequant waves request the activation pieces too).
template <int LAYOUT, int MW, int DW, int LD>
__global__ __launch_bounds__((MW + DW + (LD > 0 ? LD : 0)) * 64) void gemm4_kernel(const GemmParams p) {
  static_assert(DW == 4 || (DW == 8 && LAYOUT == 0), "8 dequant waves: 4-bit row-stream layouts only");
  constexpr int NLW = LD > 0 ? LD : (LD < 0 ? MW : DW);  // waves that request the activation tile (LD < 0: the matrix waves themselves)
  constexpr int NPA = 32 / NLW;      // ... pieces of 8 rows each per k-tile
  using namespace g4;
  constexpr int AM = 8 / MW * 2;   // 32-row MFMA tiles per matrix wave along M: 4 or 2
  constexpr int WROWS = AM * 32;   // rows per matrix wave: 128 or 64
  constexpr int BP = AM;           // MFMAs of a k-tile's last sub-step issued in front of its barrier: 4 of 8, 2 of 4
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  char *const smem = smem4;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  const int S = p.split_k;  // > 1: S consecutive block ids share an output tile and own consecutive K ranges (gemm3's protocol)
  const int nblk = tiles_m * tiles_n * S;
  int bid = blockIdx.x;
  {  // each XCD (block id % 8) walks a contiguous run of tiles
    const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ksplit = bid % S;
  bid /= S;
  const int tile_id = bid;
  const int tm = p.raster ? (bid / tiles_n) : (bid % tiles_m);
  const int tn = p.raster ? (bid % tiles_n) : (bid / tiles_m);
  const int m0 = tm * BM, n0 = tn * BN;
  const int KT = p.K / BK / S;
  const int KT0 = ksplit * KT;
#ifdef QLLM_LAB
  // timeline (lab): matrix wave 0 -> slots 0..7, dequant wave 0 -> slots 8..13, loader wave 0 -> slots 14, 15 of this block's 16
  uint64_t *const dbg = !p.dbg ? nullptr : (wave == 0 ? p.dbg + 16 * (size_t)blockIdx.x : (wave == MW ? p.dbg + 16 * (size_t)blockIdx.x + 8 :
                        ((LD && wave == MW + DW) ? p.dbg + 16 * (size_t)blockIdx.x + 14 : nullptr)));
#endif

#ifdef QLLM_LAB
  const int abl = p.prio >> 16;  // lab ablations (timing only, results wrong): 1 no LDS-write wait in the dequant waves, 2 no B stores,
                                 // 4 no dequant arithmetic, 8 no raw-weight requests, 16 no activation requests
#else
  constexpr int abl = 0;
#endif
  // issue priority of this wave's role (p.prio: matrix | dequant << 4 | loader << 8; s_setprio takes an immediate)
  auto set_prio = [&](int v) {
    if (v == 1) __builtin_amdgcn_s_setprio(1);
    else if (v == 2) __builtin_amdgcn_s_setprio(2);
    else if (v == 3) __builtin_amdgcn_s_setprio(3);
  };

  // ---- activation pieces (loader waves, or the dequant waves when LD == 0): wave wl of the NLW owns rows wl*8*NPA .. of the tile
  // = NPA pieces of 8 rows x 128 B per k-tile; lane l -> row 8q + l/8, physical slot l%8, which holds logical chunk
  // (l%8) ^ swz(row) (the swizzle on the SOURCE address).
  const int wl = LD > 0 ? (wave - MW - DW) : (LD < 0 ? wave : wave - MW);
  const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, (int)min((size_t)p.M * p.K * 2, (size_t)0x7fffffff), 0x00020000);
  int voff_x[NPA];
  if (LD < 0 ? wave < MW : wave >= MW + (LD ? DW : 0)) {
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      const int r = wl * (8 * NPA) + 8 * q + (lane >> 3);
      const int grow = min(m0 + r, p.M - 1);  // rows past M re-read the last row; their outputs are never stored
      voff_x[q] = grow * p.K * 2 + (((lane & 7) ^ swz(r)) << 4);
    }
  }
  // (the builtin's operands go through plain locals: with template-dependent expressions in the call the HOST pass of hipcc
  //  silently fails to instantiate the kernel -- gemm3.hip)
  auto dma_a_piece = [&](int kt, int slot, int q) {  // (q must be a compile-time constant at the call site)
    if (abl & 16) return;
    const int so = (KT0 + min(kt, KT - 1)) * (BK * 2);
    const int vo = voff_x[q];
    lds_void_t *dst = (lds_void_t *)(smem + kAOff + slot * kATileB + (wl * (8 * NPA) + q * 8) * 128);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
  };
  auto dma_a = [&](int kt, int slot) {
    const int so = (KT0 + min(kt, KT - 1)) * (BK * 2);
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
      if (abl & 16) break;
      const int vo = voff_x[q];
      lds_void_t *dst = (lds_void_t *)(smem + kAOff + slot * kATileB + (wl * (8 * NPA) + q * 8) * 128);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, dst, 16, vo, so, 0, 0);
    }
  };

  if (LD > 0 && wave >= MW + DW) {
    // ============================================ activation loader waves =============================================
    // Tile kt+2 is requested behind barrier #kt-1 (which freed its ring slot); vmcnt(NPA) in front of barrier #kt: everything
    // but those NPA pieces, i.e. tile kt+1, has landed.
    set_prio((p.prio >> 8) & 3);
    dma_a(0, 0);
    dma_a(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPA) : "memory");
    __builtin_amdgcn_s_barrier();  // prologue barrier
    int sa2 = 2;
#ifdef QLLM_LAB
    uint64_t busy = 0, t_it = dbg ? __builtin_amdgcn_s_memtime() : 0;
#endif
    for (int kt = 0; kt < KT; ++kt) {
      dma_a(kt + 2, sa2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPA) : "memory");
#ifdef QLLM_LAB
      if (dbg) busy += __builtin_amdgcn_s_memtime() - t_it;
#endif
      __builtin_amdgcn_s_barrier();  // barrier #kt
#ifdef QLLM_LAB
      if (dbg) t_it = __builtin_amdgcn_s_memtime();
#endif
      sa2 = (sa2 == 2) ? 0 : sa2 + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stray pieces past the last tile land before the LDS is reused
#ifdef QLLM_LAB
    if (dbg && lane == 0) dbg[0] = busy;
#endif
    __builtin_amdgcn_s_barrier();  // matches the matrix waves' barrier in front of their epilogue
    if (S > 1) {                   // ... and the two around the split-K ticket
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  if (wave >= MW) {
    // ================================================= dequant waves ==================================================
    const int w = wave - MW;  // 0..DW-1
    const int l = lane;
    constexpr bool ROWS = LAYOUT != 1;
    constexpr int WPT = (LAYOUT == 2) ? 3 : 16 / DW;  // packed words per thread and k-tile: 4 (3 bits: 3), or 2 with 8 dequant waves
    const int zk = p.zero_kind;
    const bool sm = ROWS && p.sm;
    const int Gn = (p.K + (1 << p.gs_shift) - 1) >> p.gs_shift;
    const uint32_t nibmask = nib_mask_vgpr();
    set_prio((p.prio >> 4) & 3);
    G4_STAMP(1);

    // ---- raw weight pieces: what this wave's 64 threads dequantise in one k-tile
    //   row stream : columns c0 .. c0+63 (thread = column), 32 k = word rows r0 .. r0+3 (3 bits: +2) of the k-tile's 8 (6);
    //                with 8 dequant waves 16 k = word rows r0, r0+1
    //   AWQ        : all 16 word columns, k rows 16 w .. +15 (thread = word column l%16, k rows 4 (l/16) .. +3)
    const int c0 = n0 + 64 * (w & 1);
    const int wrows_tile = (LAYOUT == 0) ? 8 : (LAYOUT == 2 ? 6 : BK);  // packed rows per k-tile
    const int r0 = ROWS ? (WPT * (w >> 1)) : 16 * w;
    const int strip_rows = (LAYOUT == 2) ? (p.K * 3) >> 5 : (p.K >> 3);
    int voff_w, step_w;  // per-lane byte offset of the W piece; bytes per k-tile
    if constexpr (ROWS) {
      if (sm) {  // image [4 strips][WPT rows][16 words] (two word rows per thread: lanes 32..63 repeat lanes 0..31)
        if constexpr (WPT == 2) voff_w = (((c0 >> 4) + ((l >> 3) & 3)) * strip_rows + r0 + ((l >> 2) & 1)) * 64 + (l & 3) * 16;
        else voff_w = (((c0 >> 4) + (l >> 4)) * strip_rows + r0 + min((l >> 2) & 3, WPT - 1)) * 64 + (l & 3) * 16;
        step_w = wrows_tile * 64;
      } else {   // image [WPT rows][64 words]
        voff_w = ((r0 + (WPT == 2 ? ((l >> 4) & 1) : min(l >> 4, WPT - 1))) * p.N + c0 + 4 * (l & 15)) * 4;
        step_w = wrows_tile * p.N * 4;
      }
    } else {
      voff_w = ((r0 + (l >> 2)) * (p.N >> 3) + (n0 >> 3) + 4 * (l & 3)) * 4;
      step_w = BK * (p.N >> 3) * 4;
    }
    // scales: one group row per wave and k-tile (group_size >= 32; the wave's k range is 32 (16) rows)
    const int kofs = ROWS ? (DW == 8 ? 16 : 32) * (w >> 1) : 16 * w;
    int voff_s, step_s;  // per-lane byte offset inside a group row; bytes per group row
    if (sm) {
      voff_s = ((c0 >> 4) + ((l >> 3) & 3)) * Gn * 32 + (l & 7) * 4;
      step_s = 32;
    } else {
      voff_s = ROWS ? c0 * 2 + 4 * (l & 31) : n0 * 2 + 4 * l;
      step_s = p.N * 2;
    }
    int voff_z, step_z;
    const void *zbase = p.qzeros;
    size_t zbytes;
    if (zk == ZK_PACKED) {
      if (sm) {
        voff_z = ((c0 >> 4) + ((l >> 1) & 3)) * Gn * 8 + (l & 1) * 4;
        step_z = 8;
        zbytes = (size_t)(p.N >> 4) * Gn * 8;
      } else if constexpr (LAYOUT == 2) {
        voff_z = ((c0 * 3) >> 5) * 4 + 4 * (l & 7);
        step_z = ((p.N * 3) >> 5) * 4;
        zbytes = (size_t)Gn * step_z;
      } else {
        voff_z = ROWS ? (c0 >> 3) * 4 + 4 * (l & 7) : (n0 >> 3) * 4 + 4 * (l & 15);
        step_z = (p.N >> 3) * 4;
        zbytes = (size_t)Gn * step_z;
      }
    } else {  // fp16 zero points: stored like the scales; symmetric: the scales again (never read back)
      voff_z = voff_s;
      step_z = step_s;
      zbytes = (size_t)Gn * p.N * 2;
      if (zk == ZK_SYM) zbase = p.scales;
    }
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.qweight, 0, (int)((size_t)p.K * p.N * (LAYOUT == 2 ? 3 : 4) / 8), 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void *)p.scales, 0, Gn * p.N * 2, 0x00020000);
    const auto rs_z = __builtin_amdgcn_make_buffer_rsrc((void *)zbase, 0, (int)zbytes, 0x00020000);

    // this wave's region of a raw ring slot: [packed words RW][scales RS][zero points RS].  With 8 dequant waves a region is half
    // the size and the three pieces are requested by lanes 0..31 only (an LDS-DMA lands at base + lane * size for the ACTIVE lanes)
    constexpr int RW = DW == 8 ? 512 : kRawW, RS = DW == 8 ? 128 : kRawS, kRegion = RW + 2 * RS;
    static_assert(kRegion * DW == kRawSlotB, "raw ring slot = the dequant waves' regions");
    auto dma_raw = [&](int kt, int slot) {
      const int ktc = KT0 + min(kt, KT - 1);
      const int G = (ktc * BK + kofs) >> p.gs_shift;
      char *base = smem + kRawOff + slot * kRawSlotB + w * kRegion;
      const int so_w = ktc * step_w, so_s = G * step_s, so_z = G * step_z;
      const int vw = voff_w, vs = voff_s, vz = voff_z;
      lds_void_t *dw = (lds_void_t *)base, *ds = (lds_void_t *)(base + RW), *dz = (lds_void_t *)(base + RW + RS);
      if (abl & 8) return;
      if (DW == 8 && l >= 32) return;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, dw, 16, vw, so_w, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, ds, 4, vs, so_s, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, dz, 4, vz, so_z, 0, 0);
    };

    // ---- raw tile (LDS) -> registers -> B stage ------------------------------------------------------------------------
    struct Raw {
      uint32_t wd[WPT];
      uint32_t s;    // row stream: the column's scale, raw 16 bits
      half8_t s8;    // AWQ: the 8 columns' scales
      uint32_t z, z2;
    };
    const int rb_lo = ROWS ? (sm ? (l >> 4) * (64 * (WPT == 2 ? 2 : 4)) + (l & 15) * 4 : l * 4) : (4 * (l >> 4)) * 64 + (l & 15) * 4;
    const int rb_rs = ROWS ? (sm ? 64 : 256) : 64;  // bytes between a thread's consecutive words
    // 3-bit packed zero points: field of column ci inside the wave's (strip's) word run
    const int ci3 = sm ? (l & 15) : l, zi3 = (sm ? (l >> 4) * 2 : 0) + ((3 * ci3) >> 5);
    auto read_raw = [&](int slot, Raw &r) {
      const char *rw = smem + kRawOff + slot * kRawSlotB + w * kRegion;
#pragma unroll
      for (int i = 0; i < WPT; ++i) r.wd[i] = *(const uint32_t *)(rw + rb_lo + i * rb_rs);
      if constexpr (LAYOUT == 1) {
        r.s8 = *(const half8_t *)(rw + RW + (l & 15) * 16);
        r.z = *(const uint32_t *)(rw + RW + RS + (l & 15) * 4);
      } else {
        r.s = *(const uint16_t *)(rw + RW + l * 2);
        if (zk == ZK_PACKED) {
          if constexpr (LAYOUT == 2) {
            r.z = *(const uint32_t *)(rw + RW + RS + zi3 * 4);
            r.z2 = *(const uint32_t *)(rw + RW + RS + zi3 * 4 + 4);
          } else {
            r.z = *(const uint32_t *)(rw + RW + RS + (l >> 3) * 4);
          }
        } else {
          r.z = *(const uint16_t *)(rw + RW + RS + l * 2);  // fp16 zero point (symmetric: unused)
        }
      }
    };
    auto produce_b = [&](int stage, const Raw &r) {
      char *Bb = smem + kBOff + stage * kBTileB;
      if (abl & 6) {  // (lab) 4: stores of raw words, no arithmetic; 2: no stores at all
        if (!(abl & 2)) {
#pragma unroll
          for (int i = 0; i < WPT; ++i) *(uint4_t *)(Bb + tile_off_b(64 * (w & 1) + l, WPT * (w >> 1) + i)) = uint4_t{r.wd[0], r.wd[1], r.wd[0], r.s};
        } else {
          asm volatile("" ::"v"(r.wd[0]), "v"(r.wd[1]), "v"(r.s), "v"(r.z));
        }
        return;
      }
      if constexpr (LAYOUT == 0) {
        const half_t zp = (half_t)(float)(((r.z >> (4 * (l & 7))) + (uint32_t)p.add_zero_bias) & 15u);
        const half_t zf = __builtin_bit_cast(half_t, (uint16_t)r.z);
        const half_t sc = __builtin_bit_cast(half_t, (uint16_t)r.s);
        const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)8.f));
        const int bcol = 64 * (w & 1) + l, ch0 = WPT * (w >> 1);
#pragma unroll
        for (int i = 0; i < WPT; ++i) *(half8_t *)(Bb + tile_off_b(bcol, ch0 + i)) = unperm_04152637(deq_word_k04(r.wd[i], cc, nibmask));
      } else if constexpr (LAYOUT == 2) {
        // 32 k = 96 bits of the column's bit stream in wd[0..2]: four 24-bit fields of 8 values each, natural k order
        const uint32_t zfield = (uint32_t)(((((uint64_t)r.z2) << 32) | r.z) >> ((3 * ci3) & 31));
        const half_t zp = (half_t)(float)((zfield + (uint32_t)p.add_zero_bias) & 7u);
        const half_t zf = __builtin_bit_cast(half_t, (uint16_t)r.z);
        const half_t sc = __builtin_bit_cast(half_t, (uint16_t)r.s);
        const ColConst cc = make_col_const(sc, (zk == ZK_PACKED) ? zp : ((zk == ZK_F16) ? zf : (half_t)4.f));
        const uint32_t f[4] = {r.wd[0] & 0xffffffu, __builtin_amdgcn_alignbit(r.wd[1], r.wd[0], 24) & 0xffffffu,
                               __builtin_amdgcn_alignbit(r.wd[2], r.wd[1], 16) & 0xffffffu, r.wd[2] >> 8};
        const int bcol = 64 * (w & 1) + l, ch0 = 4 * (w >> 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          half2_t b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t lo = (f[i] >> (6 * j)) & 7u, hi = (f[i] >> (6 * j + 3)) & 7u;
            b[j] = deq_pair(lo | (hi << 16) | kMagic, cc);
          }
          *(half8_t *)(Bb + tile_off_b(bcol, ch0 + i)) = half8_t{b[0].x, b[0].y, b[1].x, b[1].y, b[2].x, b[2].y, b[3].x, b[3].y};
        }
      } else {
        // rows 4 kq .. +3 of 8 interleaved columns: column c of the word sits at nibble awq_nibble_of_col(c).  Two v_perm
        // build, per column pair, the (k0,k1) and (k2,k3) nibble-bearing 16-bit halves side by side (as gemm3).
        const int wc = l & 15, kq = l >> 4;
        const uint32_t P01 = __builtin_amdgcn_perm(r.wd[1], r.wd[0], 0x05040100u), Q01 = __builtin_amdgcn_perm(r.wd[1], r.wd[0], 0x07060302u);
        const uint32_t P23 = __builtin_amdgcn_perm(r.wd[3], r.wd[2], 0x05040100u), Q23 = __builtin_amdgcn_perm(r.wd[3], r.wd[2], 0x07060302u);
        const int krow = 16 * w + 4 * kq;  // first k of this thread inside the k-tile
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int sh = 4 * (c >> 1);
          const half_t z = (half_t)(float)((r.z >> (4 * awq_nibble_of_col(c))) & 15u);
          const ColConst cc = make_col_const(r.s8[c], z);
          const half2_t b01 = deq_pair(and_or(((c & 1) ? Q01 : P01) >> sh, nibmask, kMagic), cc);
          const half2_t b23 = deq_pair(and_or(((c & 1) ? Q23 : P23) >> sh, nibmask, kMagic), cc);
          *(uint2_t *)(Bb + tile_off_b(8 * wc + c, krow >> 3) + (krow & 7) * 2) = uint2_t{as_u32(b01), as_u32(b23)};
        }
      }
    };

    // Request queue of this wave (loads complete in order), 3 pieces per raw tile [8 per activation tile]:
    //     LD > 0:  raw0 .. raw4, raw5 | raw6 | raw7 | ...                 LD == 0:  raw0 .. raw4 A0 A1 raw5 | A2 raw6 | A3 raw7 | ...
    // Iteration kt turns the registers read one iteration earlier (raw tile kt+1) into B tile kt+1, and reads raw tile kt+2 from the
    // ring FIRST, so that the LDS round trip hides under the dequant arithmetic.  Waits: raw tile kt+2 has landed when at most the
    // requests behind it are outstanding -- 4 raw tiles = 12 (LD > 0); with the activation pieces in the queue at least 12 + 2 NPA
    // requests follow it in every iteration (raw3 raw4 A0 A1 raw5 A2 raw6 in iteration 0, more later) -- and A tile kt+1 when only
    // raw(kt+5), A(kt+2), raw(kt+6) = 6 + NPA are.  A raw tile has >= 2.5 k-tiles of flight; its ring slot (5 deep) is
    // re-requested one iteration after it was read.
    constexpr int kWaitRaw = LD != 0 ? 12 : 12 + 2 * NPA, kWaitA = 6 + NPA, kWaitA0 = 3 + NPA;
    Raw cur, nxt;
#pragma unroll
    for (int j = 0; j < 5; ++j) dma_raw(j, j);
    if constexpr (LD == 0) {
      dma_a(0, 0);
      dma_a(1, 1);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitRaw) : "memory");  // raw tile 0 is here
    G4_STAMP(2);
    read_raw(0, cur);
    produce_b(0, cur);
    dma_raw(5, 0);  // (slot 0 is free: produce_b consumed the registers read from it)
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitRaw) : "memory");  // raw tile 1 is here
    read_raw(1, cur);
    if constexpr (LD == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitA0) : "memory");  // A tile 0 is here (A1, raw5 may be in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G4_STAMP(3);
    __builtin_amdgcn_s_barrier();  // prologue barrier
    int sa2 = 2;                   // A ring slot of tile kt+2
    int rs_new = 1, rs_rd = 2;     // ring slots of raw tiles kt+6 and kt+2
#ifdef QLLM_LAB
    uint64_t busy = 0, t_it = dbg ? __builtin_amdgcn_s_memtime() : 0;
    uint64_t c_issue = 0, c_wraw = 0, c_deq = 0;  // cycles spent requesting / waiting for the raw tile / reading + dequantising + storing
#define G4_LAP(acc_) do { if (dbg) { const uint64_t t_ = __builtin_amdgcn_s_memtime(); acc_ += t_ - t_lap; t_lap = t_; } } while (0)
#else
#define G4_LAP(acc_) do { } while (0)
#endif
    for (int kt = 0; kt < KT; ++kt) {
#ifdef QLLM_LAB
      uint64_t t_lap = t_it;
#endif
      if constexpr (LD == 0) dma_a(kt + 2, sa2);
      dma_raw(kt + 6, rs_new);
      G4_LAP(c_issue);
      asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitRaw) : "memory");  // raw tile kt+2 has landed
      G4_LAP(c_wraw);
      read_raw(rs_rd, nxt);
      produce_b((kt + 1) & 1, cur);
      G4_LAP(c_deq);  // (the stamp waits for lgkmcnt(0): the stores' completion is in here)
      if constexpr (LD == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(kWaitA) : "memory");  // A tile kt+1 has landed
      if (!(abl & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef QLLM_LAB
      if (dbg) busy += __builtin_amdgcn_s_memtime() - t_it;  // request + wait + dequant + store of this iteration
#endif
      __builtin_amdgcn_s_barrier();  // barrier #kt
#ifdef QLLM_LAB
      if (dbg) t_it = __builtin_amdgcn_s_memtime();
#endif
      cur = nxt;
      sa2 = (sa2 == 2) ? 0 : sa2 + 1;
      rs_new = (rs_new == kRawSlots - 1) ? 0 : rs_new + 1;
      rs_rd = (rs_rd == kRawSlots - 1) ? 0 : rs_rd + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stray pieces past the last tile land before the LDS is reused
#ifdef QLLM_LAB
    if (dbg && lane == 0) {
      dbg[4] = busy;
      dbg[0] = c_issue;
      dbg[1] = c_wraw;  // (overwrites the entry stamp: read slot 2 - slot 1 only in runs without the lap counters)
      dbg[5] = c_deq;
    }
#endif
    __builtin_amdgcn_s_barrier();  // matches the matrix waves' barrier in front of their epilogue
    if (S > 1) {                   // ... and the two around the split-K ticket
      __syncthreads();
      __syncthreads();
    }
    return;
  }

  // =================================================== matrix waves ===================================================
  G4_RSTAMP(0);
  G4_STAMP(1);
  const int wm = wave >> 1, wn = wave & 1;   // (MW/2) (M) x 2 (N): rows wm*WROWS.., columns wn*64..
  const int fr = lane & 31, fs = lane >> 5;  // fragment row (A: m, B: n) and k half of the 16-wide sub-step
  float16_t acc[AM][2];
#pragma unroll
  for (int a = 0; a < AM; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // per-lane fragment offsets of the four sub-steps inside a tile (the swizzle depends on row % 32 = fr only, so A and B
  // fragments share them); further 32-row fragments are +4096 B immediates, tile bases are wave-uniform.
  // off(ks) = fr * 128 + (((2 ks + fs) ^ swz(fr)) << 4) = off0 ^ (ks << 5): ONE register, one v_xad_u32 per address
  const int off0 = fr * 128 + ((fs ^ swz(fr)) << 4);
  auto off = [&](int ks) { return off0 ^ (ks << 5); };
  const char *const baseA = smem + kAOff + wm * WROWS * 128, *const baseB = smem + kBOff + wn * 64 * 128;
  half8_t fa[AM], fb0[2], fb1[2];
  set_prio(p.prio & 3);
  if constexpr (LD < 0) {  // the matrix waves request the activation tiles themselves (NPA pieces per wave and k-tile)
    dma_a(0, 0);
    dma_a(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPA) : "memory");  // tile 0 has landed
  }
  __builtin_amdgcn_s_barrier();   // prologue barrier: A tile 0 and B tile 0 are complete
  G4_STAMP(2);
#pragma unroll
  for (int j = 0; j < 2; ++j) fb0[j] = *(const half8_t *)(baseB + off(0) + j * 4096);
#pragma unroll
  for (int j = 0; j < AM; ++j) fa[j] = *(const half8_t *)(baseA + off(0) + j * 4096);
  // The loop is rotated so that its header sits right behind a barrier (where nothing is in flight): hipcc's wait-count pass
  // merges the in-order LDS counter conservatively at a loop header and would otherwise wait for the newest fragment reads
  // (lgkmcnt(0)) in front of every k-tile's first MFMA.
  //   body(kt) = [rest of tile kt-1's last sub-step beside the reads of tile kt's first fragments] [sub-steps 0..2 of tile kt]
  //              [the first BP MFMAs of its last sub-step] lgkmcnt(0) barrier #kt
  constexpr int NM = AM * 2;
#ifdef QLLM_LAB
  uint64_t bwait = 0;
#endif
  // LD < 0: a quarter of the wave's pieces of A tile kt+2 (ring slot (kt+2) % 3, free since barrier #kt-1) behind each sub-step
  constexpr int NQ = NPA / 4 > 0 ? NPA / 4 : 1;
  auto pieces = [&](int kt, int sa, int part) {
    if constexpr (LD < 0) {
      const int s2 = (sa == 0) ? 2 : sa - 1;  // slot of tile kt+2 = slot of tile kt-1
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        if (part == 0) dma_a_piece(kt + 2, s2, 0 * NQ + j);
        else if (part == 1) dma_a_piece(kt + 2, s2, 1 * NQ + j);
        else if (part == 2) dma_a_piece(kt + 2, s2, 2 * NQ + j);
        else dma_a_piece(kt + 2, s2, 3 * NQ + j);
      }
      G4_SB();
    }
  };
  auto head = [&](int kt, int sa) {  // sub-steps 0..2 of tile kt + the first BP MFMAs of sub-step 3 + barrier #kt
    const char *ta = baseA + sa * kATileB, *tb = baseB + (kt & 1) * kBTileB;
    pieces(kt, sa, 0);
    g4_substep<AM, 0, NM, 0, true>(acc, fa, fb0, fb1, ta + off(1), tb + off(1));
    pieces(kt, sa, 1);
    g4_substep<AM, 0, NM, 0, true>(acc, fa, fb1, fb0, ta + off(2), tb + off(2));
    pieces(kt, sa, 2);
    g4_substep<AM, 0, NM, 0, true>(acc, fa, fb0, fb1, ta + off(3), tb + off(3));
    pieces(kt, sa, 3);
    g4_substep<AM, 0, BP, 0, false>(acc, fa, fb1, fb0, ta, tb);
    if constexpr (LD < 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NPA) : "memory");  // my pieces of A tile kt+1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment read of this k-tile has returned: its stages are free
#ifdef QLLM_LAB
    const uint64_t t_bar = dbg ? __builtin_amdgcn_s_memtime() : 0;
#endif
    __builtin_amdgcn_s_barrier();
#ifdef QLLM_LAB
    if (dbg) bwait += __builtin_amdgcn_s_memtime() - t_bar;
#endif
    G4_SB();
  };
  head(0, 0);
  int sa = 1;
  for (int kt = 1; kt < KT; ++kt) {
    g4_substep<AM, BP, NM, BP, true>(acc, fa, fb1, fb0, baseA + sa * kATileB + off(0), baseB + (kt & 1) * kBTileB + off(0));
    head(kt, sa);
    sa = (sa == 2) ? 0 : sa + 1;
  }
  g4_substep<AM, BP, NM, BP, false>(acc, fa, fb1, fb0, smem, smem);
  __builtin_amdgcn_s_setprio(0);
  if constexpr (LD < 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stray pieces past the last tile
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  G4_STAMP(3);
  __builtin_amdgcn_s_barrier();  // every wave is done with the tiles (and the producers' stray DMA pieces have landed)
  G4_STAMP(4);

  // ---- split-K: publish the fp32 partial tile; the last block to arrive sums the S partials in fixed order (gemm3's protocol)
  if (S > 1) {
    int &s_ticket = *(int *)(smem + 48 * 1024);  // past the epilogue's wave-private regions (8 x 4.5 KB)
    constexpr int WREGS = AM * 2 * 16;
    float *slab = p.slabs + ((size_t)tile_id * S + ksplit) * (size_t)(BM * BN) + (size_t)wave * (WREGS * 64) + lane;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) st_sc1(slab + ((a * 2 + b) * 16 + r) * 64, acc[a][b][r]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(p.counters + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != S - 1) return;
#pragma unroll
    for (int a = 0; a < AM; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    for (int sp = 0; sp < S; ++sp) {
      const float *src = p.slabs + ((size_t)tile_id * S + sp) * (size_t)(BM * BN) + (size_t)wave * (WREGS * 64) + lane;
#pragma unroll
      for (int a = 0; a < AM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[a][b][r] += ld_sc1(src + ((a * 2 + b) * 16 + r) * 64);
    }
    if (tid == 0) __hip_atomic_store(p.counters + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- epilogue: + bias, round once, transpose through wave-private LDS, 16-byte row-contiguous stores ---------------------
  // C/D layout of 32x32 tiles: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
  half_t *ep = (half_t *)smem + wave * (32 * 72);  // 32 rows x 64 cols, row stride 72 halves (144 B)
  float bv[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) bv[b] = p.bias ? (float)p.bias[n0 + wn * 64 + b * 32 + fr] : 0.f;
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
        if (p.out_bf16) ((uint16_t *)ep)[row * 72 + b * 32 + fr] = f32_to_bf16((float)(half_t)v);
        else ep[row * 72 + b * 32 + fr] = (half_t)v;
      }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int c = lane + 64 * h, row = c >> 3, ch = c & 7;
      const uint4_t v = *(const uint4_t *)(ep + row * 72 + ch * 8);
      const int m = m0 + wm * WROWS + a * 32 + row;
      if (m < p.M) *(uint4_t *)((half_t *)p.y + (size_t)m * p.N + n0 + wn * 64 + ch * 8) = v;
    }
  }
  G4_STAMP(5);
  G4_RSTAMP(6);
#ifdef QLLM_LAB
  if (dbg && lane == 0) dbg[7] = bwait;  // cycles between arriving at and leaving the k-tile barriers (incl. ~2 stamp reads each)
#endif
}
#undef G4_SB

template <int LAYOUT, int MW, int DW, int LD>
static int launch_gemm4_b(const GemmParams &p, hipStream_t stream) {
  using namespace g4;
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)gemm4_kernel<LAYOUT, MW, DW, LD>)) return rc;
  const int tiles = ((p.M + BM - 1) / BM) * (p.N / BN) * p.split_k;
  hipLaunchKernelGGL((gemm4_kernel<LAYOUT, MW, DW, LD>), dim3(tiles), dim3((MW + DW + (LD > 0 ? LD : 0)) * 64), (size_t)kLdsBytes, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

// variant: 0 = 4 matrix waves (128x64 each, one per SIMD) + 4 dequant-and-load waves; 1 = 8 matrix waves (64x64 each, two per
// SIMD) + 4 dequant + 4 loader waves; 2 = 8 matrix + 4 dequant-and-load waves; 3 = 8 matrix + 8 dequant-and-load waves (4-bit
// row-stream layouts; the others run variant 1)
int launch_gemm4(const GemmParams &p_in, int layout, int variant, hipStream_t stream) {
  GemmParams p = p_in;
  // issue priorities: matrix | dequant << 4 | loader << 8
  p.prio = knob("QLLM_G4_PRIO_M", 2) | (knob("QLLM_G4_PRIO_D", 0) << 4) | (knob("QLLM_G4_PRIO_L", 0) << 8) | (knob("QLLM_G4_ABLATE", 0) << 16);
  const int L = (layout == kGemm3Rows3Bit) ? 2 : (layout == QLLM_LAYOUT_AWQ_GEMM ? 1 : 0);
  if (L == 0 && variant == 3) return launch_gemm4_b<0, 8, 8, 0>(p, stream);
  if (L == 0 && variant == 4) return launch_gemm4_b<0, 8, 4, -1>(p, stream);  // the matrix waves request the activation tiles
  if (L == 0 && variant == 5) return launch_gemm4_b<0, 8, 8, -1>(p, stream);
  if (L == 0 && variant == 6) return launch_gemm4_b<0, 4, 4, -1>(p, stream);
#define G4_CASE(LL)                                                                     \
  if (L == LL) {                                                                        \
    if (variant == 1 || variant == 3) return launch_gemm4_b<LL, 8, 4, 4>(p, stream);    \
    if (variant == 2) return launch_gemm4_b<LL, 8, 4, 0>(p, stream);                    \
    return launch_gemm4_b<LL, 4, 4, 0>(p, stream);                                      \
  }
  G4_CASE(0)
  G4_CASE(1)
  G4_CASE(2)
#undef G4_CASE
  return set_error(QLLM_ERR_INVALID, "gemm4: bad layout %d", layout);
}

}  // namespace qllm
