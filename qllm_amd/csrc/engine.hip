// Persistent decode engine: a whole chain of batch-1 quantized linears (e.g. the 128 grouped launches of a Llama-2-7B decode
// step) as ONE launch in which the weight stream never stops for an activation.
//
// Why (measured, profiles/r02_chain_experiments.md): the two-stream chain of strip.hip overlaps a link's launch and first
// weight loads with its predecessor, but vector-memory returns are IN ORDER -- per wave, and (measured with this kernel's first
// versions, profiles/r02_engine.md) in effect per CU: an activation load issued by any wave of a CU whose loader has 60 KB of
// weight loads in flight comes back 2-4 us later, behind them.  So per CU (one 10-wave workgroup per CU, all resident) the
// roles are split and each kind of traffic is issued ONCE:
//   * wave 8, the LOADER: walks the block's strips in program order and streams their packed words through a 6-slot LDS ring
//     (16 KB slabs = 32 columns x 1024 k, whole 128-byte lines, plus one 1 KB piece with the slab's scales and zero words)
//     with LDS-DMA (global_load_lds, non-temporal): no registers, no dependence on any activation -- it runs up to 6 slabs
//     ahead of the arithmetic and only ever waits for a free slot.  A slab is published (LDS flag) when the counted vmcnt says
//     its 17 pieces landed.
//   * wave 9, the COURIER: for every DISTINCT input vector of the block's links (q/k/v share one) it copies the whole vector
//     into one of two LDS buffers, again by LDS-DMA (write-through-coherent loads), re-requesting the 1 KB chunks that still
//     show the 0xFFFF "not written yet" pattern when the input is another link's output (the in-band hand-off of strip.hip).
//     One trip through the CU's memory queue per input vector instead of one per slab and wave.
//   * waves 0-7, the CONSUMERS: each owns one 128-k slice (= one g128 group) of every slab.  No vector-memory traffic at all
//     until a strip's 32 results are stored: packed words, scales / zeros and the activation slice all come from LDS.
//     Arithmetic = the register-A form of strip.hip (raw magic-number B fragments, Sx / Sx' from two bookkeeping MFMAs, one
//     fp32 correction per group); packed words with ds_read_b32 (lane (g,i): word-row 4s+g, column i -- 256 contiguous bytes per
//     wave-read).  After a strip's last slab the eight partial rows meet in an LDS buffer; the last arriver (LDS ticket) sums
//     them in wave order (deterministic), adds the bias and publishes 32 outputs write-through.
// No workgroup barrier after the first one: loader -> consumers through full[slot] (epoch), consumers -> loader through
// done[slot] (cumulative count; for a strip's last slab only after the strip's reduction, so a wave RING strips ahead proves
// the reduction buffer free), courier -> consumers through xready (inputs delivered), consumers -> courier through xdone
// (inputs released), consumers among themselves through the arrival ticket.
// Every block walks the links in program order and a link's input is produced by earlier links only, so the waits are acyclic;
// all spins are bounded (error word, as in strip.hip).  Scope: M = 1, fp16 activations, 4 bits, group size 128, row-stream
// layouts, N % 32 == 0, K % 128 == 0, K <= 11264 (the LDS input buffers) -- the decode step of BASELINE configs[1].
#include <cstddef>
#include "kernels.hpp"

namespace qllm {

namespace eng {
constexpr int NC = 8;                 // consumer waves
constexpr int RING = 6;               // slots
constexpr int SLAB_K = 1024;          // k per slab
constexpr int SLAB_COLS = 32;         // columns per strip
constexpr int SLAB_WORDS = (SLAB_K / 8) * SLAB_COLS;  // 4096 packed words = 16 KB
constexpr int META_WORDS = 256;                        // one more 1 KB piece: the slab's 8 groups' scales and zero words
constexpr int SLOT_WORDS = SLAB_WORDS + META_WORDS;    // 17 KB per ring slot
constexpr int RED_BUFS = 8;                            // >= RING (see above)
constexpr int XBUF_HALVES = 11264;                     // one input vector, 22 chunks of 1 KB
constexpr int THREADS = (NC + 2) * 64;
constexpr uint32_t kSpinLimit = 1u << 16;  // LDS polls of ~0.1 us before a wait gives up (then every later wait of the wave gives up at once)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // flag words: explicit LDS pointers (a generic pointer would turn
typedef __attribute__((address_space(3))) float lds_f32;      //  every poll into a flat_load on the vector-memory counter)
typedef __attribute__((address_space(3))) const uint4_t lds_cu128;
typedef __attribute__((address_space(3))) uint4_t lds_u128;
typedef __attribute__((address_space(1))) const uint32_t g_cu32;
typedef __attribute__((address_space(1))) const uint16_t g_cu16;
typedef __attribute__((address_space(1))) const uint8_t g_cu8;
typedef __attribute__((address_space(1))) const half_t g_ch;
typedef __attribute__((address_space(1))) uint32_t g_u32;

__device__ __forceinline__ bool has_ffff16(uint64_t v) {
  const uint64_t t = ~v;
  return ((t - 0x0001000100010001ull) & ~t & 0x8000800080008000ull) != 0;
}
__device__ __forceinline__ uint32_t lds_load(lds_u32 *p) { return *(volatile lds_u32 *)p; }
// flag updates as bare DS instructions: as C++ atomics the compiler puts an s_waitcnt vmcnt(0) in front of each, which makes the
// wave that has just stored a strip's results sit out the store's acknowledgement (~1 us) at its next slab
__device__ __forceinline__ void lds_inc(lds_u32 *p) { asm volatile("ds_add_u32 %0, %1" ::"v"(p), "v"(1u) : "memory"); }
__device__ __forceinline__ uint32_t lds_inc_rtn(lds_u32 *p) {
  uint32_t r;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(p), "v"(1u) : "memory");
  return r;
}
}  // namespace eng

// DBG: diagnostics build (tools/engine_timeline.py): 100 MHz timestamps per (block, link) -- [0] consumer wave 0 reaches the
// link, [1] has its input, [2] finished its last strip of the link, [3] the courier delivered the link's input (only stamped
// for the first link of those that share it) -- at dbg[(block * n_links + link) * 4], then per block at
// dbg[(NB * n_links + block) * 4]: ticks wave 0 waited for inputs (), ticks it waited for
// slabs, its total ticks, ticks the loader waited for free slots; then at dbg[(NB * n_links + NB + block) * 4]: wave 0's ticks
// in the LDS read batches, in the arithmetic, in the strip reductions, in the once-per-input rewrites.
template <bool DBG>
__global__ __launch_bounds__(eng::THREADS) void engine_kernel(const EngineLink *__restrict__ links, int n_links, uint32_t *err, uint64_t *dbg) {
  using namespace eng;
  auto now = [&]() -> uint64_t { return DBG ? __builtin_amdgcn_s_memrealtime() : 0; };
  const uint64_t t_entry = now();
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  lds_u32 *ring = (lds_u32 *)smem;                          // [RING][SLOT_WORDS]
  lds_u32 *xbuf = ring + RING * SLOT_WORDS;                 // [2][XBUF_HALVES] halves
  lds_u32 *full = xbuf + XBUF_HALVES;                       // [8]  epoch of the slab a slot holds (0 = none yet)
  lds_u32 *done = full + 8;                                 // [8]  consumer completions, cumulative
  lds_u32 *arrive = done + 8;                               // [RED_BUFS] arrival tickets
  lds_u32 *xready = arrive + 8;                             // inputs the courier has delivered so far
  lds_u32 *xdone = xready + 1;                              // releases of inputs by consumer waves so far
  lds_f32 *red = (lds_f32 *)(full + 32);                    // [RED_BUFS][NC][SLAB_COLS]
  lds_u32 *ltab = (lds_u32 *)(red + RED_BUFS * NC * SLAB_COLS);  // [n_links] strip0 mod NB | slabs << 8 | n_strips << 16

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NB = gridDim.x, bid = blockIdx.x;
  if (threadIdx.x < 32) full[threadIdx.x] = 0;  // all flag words
  for (int l = threadIdx.x; l < n_links; l += blockDim.x)
    ltab[l] = (uint32_t)(links[l].strip0 % NB) | ((uint32_t)links[l].slabs << 8) | ((uint32_t)links[l].n_strips << 16);
  __syncthreads();  // the only workgroup barrier

  // strip0 mod NB | slabs << 8 | n_strips << 16 of link l, from LDS (wave-uniform; read straight from the descriptor array the
  // compiler makes these VECTOR loads)
  auto packed = [&](int l) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)ltab[l]); };
  auto first_t = [&](uint32_t pk) { const int t = bid - (int)(pk & 255u); return t < 0 ? t + NB : t; };
  auto LK = [&](int l) -> const EngineLink & { return links[__builtin_amdgcn_readfirstlane(l)]; };

  if (wave == NC) {
    // =================================================== loader ========================================================
    int q = 0;  // slabs issued so far
    bool dead = false;
    uint64_t ld_wait = 0;
    const int lrow = lane >> 3, lchunk = (lane & 7) * 4;  // piece p: word-row p*8 + lrow, words lchunk..lchunk+3 of its 32
    for (int l = 0; l < n_links; ++l) {
      const EngineLink &L = links[l];
      const int n_strips = L.n_strips, N = L.N, rows = L.K >> 3, slabs = L.slabs, zk = L.zero_kind, Gmax = (L.K >> 7) - 1;
      // 17th piece of a slab, 16 B per lane: lanes 0-31 the scales of its 8 groups x 32 columns (group lane/4, 8 halves each),
      // lanes 32-63 the zero words: packed -> lane 32+w = the 4 words (32 columns) of group w; fp16 -> like the scales.
      // (packed: lanes 40-63 and symmetric layers re-read a valid address; nobody reads those bytes back)
      const int mg = (lane < 32) ? (lane >> 2) : ((zk == ZK_F16) ? ((lane - 32) >> 2) : min(lane - 32, 7));
      g_cu16 *scl = (g_cu16 *)L.scales;
      g_cu32 *zw = (zk == ZK_SYM) ? (g_cu32 *)L.scales : (g_cu32 *)L.qzeros;
      int t = (bid - L.strip0) % NB;
      if (t < 0) t += NB;
      for (; t < n_strips; t += NB) {
        g_cu32 *W = (g_cu32 *)L.qweight + t * SLAB_COLS + lchunk;
        const int n0 = t * SLAB_COLS;
        for (int sl = 0; sl < slabs; ++sl, ++q) {
          const int slot = q % RING;
          const uint32_t need = (uint32_t)(NC * (q / RING));  // completions of the slot's previous uses
          const uint64_t tw = now();
          for (uint32_t spin = 0; !dead && lds_load(done + slot) < need; ++spin) {
            __builtin_amdgcn_s_sleep(1);
            if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 2u); dead = true; }
          }
          if (DBG) ld_wait += now() - tw;
          lds_u32 *dst = ring + slot * SLOT_WORDS;
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const int r = min(sl * (SLAB_K / 8) + p * 8 + lrow, rows - 1);  // rows past K: a harmless re-read (slice skipped)
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(W + (size_t)r * N), (lds_void_t *)(dst + p * 256), 16, 0, 2);
          }
          {
            const int G = min(sl * 8 + mg, Gmax);
            gbl_cvoid_t *src;
            if (lane < 32) src = (gbl_cvoid_t *)(scl + (size_t)G * N + n0 + 8 * (lane & 3));
            else if (zk == ZK_F16) src = (gbl_cvoid_t *)((g_cu16 *)zw + (size_t)G * N + n0 + 8 * (lane & 3));
            else if (zk == ZK_PACKED) src = (gbl_cvoid_t *)(zw + (size_t)G * (N >> 3) + (n0 >> 3));
            else src = (gbl_cvoid_t *)scl;
            __builtin_amdgcn_global_load_lds(src, (lds_void_t *)(dst + SLAB_WORDS), 16, 0, 2);
          }
          if (q > 0) {  // the previous slab's 17 pieces are the oldest outstanding: landed once at most 17 remain
            asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
            if (lane == 0) *(volatile lds_u32 *)(full + (q - 1) % RING) = (uint32_t)((q - 1) / RING + 1);
          }
        }
      }
    }
    if (q > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *(volatile lds_u32 *)(full + (q - 1) % RING) = (uint32_t)((q - 1) / RING + 1);
    }
    if (DBG && lane == 0) dbg[((size_t)NB * n_links + bid) * 4 + 3] = ld_wait;
    return;
  }

  if (wave == NC + 1) {
    // =================================================== courier =======================================================
    int seq = 0;
    bool dead = false;
    const void *prev_x = nullptr;
    for (int l = 0; l < n_links; ++l) {
      const uint32_t pk = packed(l);
      if (first_t(pk) >= (int)(pk >> 16)) continue;  // no strip of this link here
      const EngineLink &L = LK(l);
      const void *xp = (const void *)L.x;
      if (xp == prev_x) continue;  // same vector as the previous link's: already delivered
      prev_x = xp;
      // buffer seq & 1 is free once every consumer wave has released input seq - 2 (they release in order)
      const uint32_t need = seq >= 2 ? (uint32_t)(NC * (seq - 1)) : 0u;
      for (uint32_t spin = 0; !dead && lds_load(xdone) < need; ++spin) {
        __builtin_amdgcn_s_sleep(1);
        if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 8u); dead = true; }
      }
      const int bytes = L.K * 2, chunks = (bytes + 1023) >> 10;
      const bool poll = L.x_poll != 0;
      lds_u32 *xb = xbuf + (seq & 1) * (XBUF_HALVES / 2);
      g_cu8 *src = (g_cu8 *)xp;
      uint32_t pending = (chunks >= 32) ? 0xffffffffu : ((1u << chunks) - 1u);  // (chunks <= 22)
      for (uint32_t spin = 0; pending != 0; ++spin) {
        for (int c = 0; c < chunks; ++c)
          if ((pending >> c) & 1u)  // lanes past the vector's end re-read its last 16 bytes (never a stale "complete")
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(src + min(c * 1024 + lane * 16, bytes - 16)), (lds_void_t *)(xb + c * 256), 16, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!poll) break;
        for (int c = 0; c < chunks; ++c) {
          if (!((pending >> c) & 1u)) continue;
          const uint4_t v = *(volatile lds_cu128 *)((lds_cu128 *)(xb + c * 256) + lane);
          const bool bad = has_ffff16(((uint64_t)v.y << 32) | v.x) || has_ffff16(((uint64_t)v.w << 32) | v.z);
          if (__builtin_amdgcn_ballot_w64(bad) == 0) pending &= ~(1u << c);
        }
        if (pending == 0 || dead) break;
        if (spin > (kSpinLimit >> 4)) { if (lane == 0) atomicOr(err, 1u); dead = true; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++seq;
      if (lane == 0) *(volatile lds_u32 *)xready = (uint32_t)seq;
      if (DBG && lane == 0) dbg[((size_t)bid * n_links + l) * 4 + 3] = now();
    }
    return;
  }

  // ===================================================== consumers ======================================================
  const int g = lane >> 4, i = lane & 15;
  const uint32_t mask_lo = nib_mask_vgpr(), mask_hi = mask_lo << 4;
  const half8_t b_ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  const half8_t b_mult = {(half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f, (half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f};
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  int q = 0, sidx = 0, cseq = -1;
  bool dead = false;
  uint64_t c_wait_x = 0, c_wait_full = 0, c_reads = 0, c_math = 0, c_finish = 0, c_pre = 0;
  const void *prev_x = nullptr;
  for (int l = 0; l < n_links; ++l) {
    const uint32_t pk = packed(l);
    const int n_strips = (int)(pk >> 16), slabs = (int)((pk >> 8) & 255u), t0 = first_t(pk);
    if (t0 >= n_strips) continue;
    const EngineLink &L = LK(l);
    const int K = L.K, zk = L.zero_kind;
    const uint64_t t_in = now();
    {
      const void *xp = (const void *)L.x;
      if (xp != prev_x) {  // next input vector: release the previous one, wait for the courier
        prev_x = xp;
        if (cseq >= 0 && lane == 0) lds_inc(xdone);
        ++cseq;
        for (uint32_t spin = 0; !dead && lds_load(xready) < (uint32_t)(cseq + 1); ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 16u); dead = true; }
        }
      }
    }
    if (DBG) {
      const uint64_t t_x = now();
      c_wait_x += t_x - t_in;
      if (wave == 0 && lane == 0) {
        dbg[((size_t)bid * n_links + l) * 4 + 0] = t_in;
        dbg[((size_t)bid * n_links + l) * 4 + 1] = t_x;
      }
    }
    lds_cu128 *xv = (lds_cu128 *)(xbuf + (cseq & 1) * (XBUF_HALVES / 2));  // 8 halves per element
    const uint32_t zsel_p = (zk == ZK_PACKED) ? 0xffffffffu : 0u, zsel_h = (zk == ZK_F16) ? 0xffffffffu : 0u;
    const uint32_t zsel_s = (zk == ZK_SYM) ? __builtin_bit_cast(uint32_t, 8.0f) : 0u;
    const uint32_t zbias = (uint32_t)L.add_zero_bias;
    for (int t = t0; t < n_strips; t += NB) {
      float yacc[2] = {0.f, 0.f};
      for (int sl = 0; sl < slabs; ++sl, ++q) {
        const bool valid = sl * SLAB_K + wave * 128 < K;
        const bool last = sl == slabs - 1;
        // ---- the slab: wait until the loader has published it, take this wave's words (and its group's scale / zero), hand it back
        const int slot = q % RING;
        const uint32_t epoch = (uint32_t)(q / RING + 1);
        const uint64_t t_w = now();
        for (uint32_t spin = 0; !dead && lds_load(full + slot) < epoch; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 4u); dead = true; }
        }
        const uint64_t t_r = now();
        if (DBG) c_wait_full += t_r - t_w;
        uint32_t w[4][2], zraw[2];
        uint16_t sraw[2], zraw_h[2];
        uint4_t xq[4];
        {
          lds_u32 *base = ring + slot * SLOT_WORDS;
          lds_u32 *src = base + (wave * 16 + g) * SLAB_COLS + i;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            w[s][0] = src[(4 * s) * SLAB_COLS];
            w[s][1] = src[(4 * s) * SLAB_COLS + 16];
          }
          // meta piece: scales of group `wave` at halves [wave*32, +32); zero words: packed at word (32 + wave)*4 + col/8,
          // fp16 at halves 256 + wave*32 + col
          __attribute__((address_space(3))) const uint16_t *mh = (__attribute__((address_space(3))) const uint16_t *)(base + SLAB_WORDS);
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 16 * c + i;
            sraw[c] = mh[wave * 32 + col];
            zraw_h[c] = mh[256 + wave * 32 + col];                          // (both forms are inside the piece: read both,
            zraw[c] = base[SLAB_WORDS + (32 + wave) * 4 + (col >> 3)];    //  pick later -- a branch here would split the read batch)
          }
          // the activation slice: k = sl*1024 + wave*128 + 32 s + 8 g .. + 8 (a wave past K reads a valid address, unused)
          const int e0 = (valid ? sl * (SLAB_K / 8) + wave * 16 : 0) + g;
#pragma unroll
          for (int s = 0; s < 4; ++s) xq[s] = xv[e0 + 4 * s];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // everything is in registers: the slot may be refilled
        if (!last && lane == 0) lds_inc(done + slot);
        const uint64_t t_c = now();
        if (DBG) c_reads += t_c - t_r;
        if (valid) {
          // ---- arithmetic: register-A form (strip.hip), one group.  Every row of the 16-row tile is the same x: row 0 = acc[0]
          float4_t gacc[2] = {zero4, zero4}, g_ones = zero4, g_sx = zero4;
          const half2_t sixteenth = {(half_t)0.0625f, (half_t)0.0625f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const half8_t pv = a_perm_04152637(__builtin_bit_cast(half8_t, xq[s]));
            const half2_t q1 = half2_t{pv[2], pv[3]} * sixteenth, q3 = half2_t{pv[6], pv[7]} * sixteenth;
            const half8_t av = {pv[0], pv[1], q1.x, q1.y, pv[4], pv[5], q3.x, q3.y};
            g_ones = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b_ones, g_ones, 0, 0, 0);
            g_sx = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b_mult, g_sx, 0, 0, 0);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const uint32_t wv = w[s][c], w8 = wv >> 8;
              const half2_t b0 = as_h2((wv & mask_lo) | kMagic), b1 = as_h2((wv & mask_hi) | kMagic);
              const half2_t b2 = as_h2((w8 & mask_lo) | kMagic), b3 = as_h2((w8 & mask_hi) | kMagic);
              const half8_t bf = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
              gacc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bf, gacc[c], 0, 0, 0);
            }
          }
          const float sx = g_sx[0], s1024 = 1024.f * g_ones[0];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = 16 * c + i;  // (strip-relative: n0 is a multiple of 32, so col & 7 is the nibble index)
            const float zp = (float)(((zraw[c] >> (4 * (col & 7))) + zbias) & 15u);
            const float zh = (float)__builtin_bit_cast(half_t, zraw_h[c]);
            const float zfc = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, zp) & zsel_p) | (__builtin_bit_cast(uint32_t, zh) & zsel_h) | zsel_s);
            const float sfc = (float)__builtin_bit_cast(half_t, sraw[c]);
            const float corr = __builtin_fmaf(zfc, sx, s1024);
            yacc[c] = __builtin_fmaf(sfc, gacc[c][0] - corr, yacc[c]);
          }
        }
        if (DBG) {
          if (valid) asm volatile("s_nop 0" ::"v"(yacc[0]), "v"(yacc[1]));  // (the stamp below must come after the arithmetic)
          c_math += now() - t_c;
        }
        const uint64_t t_f = now();
        if (last) {
          // ---- strip complete: row 0 of the two 16-column tiles -> this wave's line of the reduction buffer; last arriver finishes
          const int n0 = t * SLAB_COLS;
          const int buf = sidx % RED_BUFS;
          lds_f32 *rb = red + (buf * NC) * SLAB_COLS;
          if (g == 0) {
            rb[wave * SLAB_COLS + i] = yacc[0];
            rb[wave * SLAB_COLS + 16 + i] = yacc[1];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          uint32_t ticket = 0;
          if (lane == 0) ticket = lds_inc_rtn(arrive + buf);
          ticket = __builtin_amdgcn_readfirstlane(ticket);
          if (ticket == NC - 1) {
            g_ch *bias = (g_ch *)L.bias;
            if (lane < 16) {  // two adjacent columns per lane: one 4-byte write-through store
              float v0 = 0.f, v1 = 0.f;
#pragma unroll
              for (int wv = 0; wv < NC; ++wv) {
                v0 += rb[wv * SLAB_COLS + 2 * lane];
                v1 += rb[wv * SLAB_COLS + 2 * lane + 1];
              }
              const int nn = n0 + 2 * lane;
              if (bias) { v0 += (float)bias[nn]; v1 += (float)bias[nn + 1]; }
              uint32_t h0 = __builtin_bit_cast(uint16_t, (half_t)v0), h1 = __builtin_bit_cast(uint16_t, (half_t)v1);
              h0 = (h0 == 0xffffu) ? 0xfe00u : h0;
              h1 = (h1 == 0xffffu) ? 0xfe00u : h1;
              __hip_atomic_store((g_u32 *)(L.y + nn), h0 | (h1 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile lds_u32 *)(arrive + buf) = 0;
          }
          // the strip's last slot goes back only now: whoever later finds itself RING strips ahead knows this buffer is free
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == 0) lds_inc(done + slot);
          ++sidx;
          if (DBG) c_finish += now() - t_f;
        }
      }
    }
    if (DBG && wave == 0 && lane == 0) dbg[((size_t)bid * n_links + l) * 4 + 2] = now();
  }
  if (DBG && wave == 0 && lane == 0) {
    uint64_t *o = dbg + ((size_t)NB * n_links + bid) * 4;
    o[0] = c_wait_x;
    o[1] = c_wait_full;
    o[2] = now() - t_entry;
    uint64_t *o2 = dbg + ((size_t)NB * n_links + NB) * 4 + (size_t)bid * 4;  // second per-block table
    o2[0] = c_reads;
    o2[1] = c_math;
    o2[2] = c_finish;
    o2[3] = c_pre;
  }
}

bool engine_link_ok(const qllm_weight_t &w, int M, int act_dtype) {
  if (M != 1 || act_dtype != QLLM_F16 || w.bits != 4 || w.group_size != 128 || w.layout == QLLM_LAYOUT_AWQ_GEMM || w.g_idx) return false;
  if (w.N % 32 != 0 || w.K % 128 != 0 || w.K < 128 || w.K > eng::XBUF_HALVES) return false;
  if (w.N / 32 > 65535 || (w.K + eng::SLAB_K - 1) / eng::SLAB_K > 255) return false;  // (the kernel's packed link table)
  return ((uintptr_t)w.qweight % 16 == 0) && ((uintptr_t)w.scales % 2 == 0);
}

size_t engine_lds_bytes() {
  return (size_t)(eng::RING * eng::SLOT_WORDS + eng::XBUF_HALVES + 32) * 4 + (size_t)eng::RED_BUFS * eng::NC * eng::SLAB_COLS * 4 +
         (size_t)kEngineMaxLinks * 4;
}

int launch_engine(const EngineLink *links_dev, int n_links, uint32_t *err, int grid, hipStream_t stream, uint64_t *dbg) {
  static DeviceLatch attr_done, attr_done_dbg;
  if (n_links < 1 || n_links > kEngineMaxLinks || grid < 1 || grid > 256) return QLLM_ERR_INVALID;
  if (dbg) {  // diagnostics build: the caller's buffer holds (grid * n_links + grid) * 4 u64
    if (int rc = lds_optin(attr_done_dbg, (const void *)engine_kernel<true>)) return rc;
    hipLaunchKernelGGL(engine_kernel<true>, dim3(grid), dim3(eng::THREADS), engine_lds_bytes(), stream, links_dev, n_links, err, dbg);
    QLLM_HIP_CHECK(hipGetLastError());
    return QLLM_OK;
  }
  if (int rc = lds_optin(attr_done, (const void *)engine_kernel<false>)) return rc;
  hipLaunchKernelGGL(engine_kernel<false>, dim3(grid), dim3(eng::THREADS), engine_lds_bytes(), stream, links_dev, n_links, err, (uint64_t *)nullptr);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
