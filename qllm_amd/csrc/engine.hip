// Persistent decode engine: a whole chain of batch-1 quantized linears (e.g. the 128 grouped launches of a Llama-2-7B decode
// step) as ONE launch in which the weight stream never stops for an activation.
//
// Why (measured, profiles/r02_chain_experiments.md): the two-stream chain of strip.hip overlaps a link's launch and first
// weight loads with its predecessor, but a wave's loads return IN ORDER -- once a wave has a weight prefetch in flight it
// cannot observe its (younger) activation load before the whole prefetch has landed, and the registers of a co-resident
// link hold only half of a 25-45 MB link anyway.  So the roles are split, per CU (one 9-wave workgroup per CU, all resident):
//   * wave 8, the LOADER: walks the block's strips in program order and streams their packed words through an 8-slot LDS ring
//     (16 KB slabs = 32 columns x 1024 k, whole 128-byte lines) with LDS-DMA (global_load_lds, non-temporal): no registers, no
//     dependence on any activation -- it runs up to 8 slabs (128 KB per CU, ~5 us of the chip's stream) ahead of the arithmetic
//     and only ever waits for a free slot.  A slab is published (LDS flag) when the counted vmcnt says its 16 pieces landed.
//   * waves 0-7, the CONSUMERS: each owns one 128-k slice (= one g128 group) of every slab.  Their only vector-memory traffic is
//     the activation slice (4 x 16 B per lane, polled through the 0xFFFF in-band hand-off when the input is another link's output)
//     and the slice's scale / zero words, so nothing older ever sits in front of an activation load.  Arithmetic = the
//     register-A form of strip.hip (raw magic-number B fragments, Sx / Sx' from two bookkeeping MFMAs, one fp32 correction per
//     group); packed words come from the ring with ds_read_b32 (lane (g,i): word-row 4s+g, column i -- 256 contiguous bytes per
//     wave-read).  After a strip's last slab the eight partial rows meet in a 4-deep LDS buffer; the last arriver (LDS ticket)
//     sums them in wave order (deterministic), adds the bias and publishes 32 outputs write-through.
// No workgroup barrier after the first one: loader -> consumers through full[slot] (epoch), consumers -> loader through
// done[slot] (cumulative count), consumers among themselves through the arrival ticket.  A consumer is never more than 8 slabs
// (= 2 strips of K = 4096) ahead of the slowest one, hence the 4 reduction buffers.
// Every block walks the links in program order and a link's input is produced by earlier links only, so the waits are acyclic;
// all spins are bounded (error word, as in strip.hip).  Scope of this first version: M = 1, fp16 activations, 4 bits, group
// size 128, row-stream layouts, N % 32 == 0, K % 128 == 0 -- the decode step of BASELINE configs[1].
#include "kernels.hpp"

namespace qllm {

namespace eng {
constexpr int NC = 8;                 // consumer waves
constexpr int RING = 8;               // slots
constexpr int SLAB_K = 1024;          // k per slab
constexpr int SLAB_COLS = 32;         // columns per strip
constexpr int SLAB_WORDS = (SLAB_K / 8) * SLAB_COLS;  // 4096 words = 16 KB
constexpr int RED_BUFS = 4;
constexpr uint32_t kSpinLimit = 1u << 16;  // LDS polls of ~0.1 us before a wait gives up (then every later wait of the wave gives up at once)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32;   // flag words: explicit LDS pointers (a generic pointer would turn
typedef __attribute__((address_space(3))) float lds_f32;      //  every poll into a flat_load on the vector-memory counter)
typedef __attribute__((address_space(1))) const uint32_t g_cu32;
typedef __attribute__((address_space(1))) const uint16_t g_cu16;
typedef __attribute__((address_space(1))) const half_t g_ch;
typedef __attribute__((address_space(1))) uint64_t g_u64;
typedef __attribute__((address_space(1))) uint32_t g_u32;

__device__ __forceinline__ bool has_ffff16(uint64_t v) {
  const uint64_t t = ~v;
  return ((t - 0x0001000100010001ull) & ~t & 0x8000800080008000ull) != 0;
}
__device__ __forceinline__ uint32_t lds_load(lds_u32 *p) { return *(volatile lds_u32 *)p; }
}  // namespace eng

__global__ __launch_bounds__(576) void engine_kernel(const EngineLink *__restrict__ links, int n_links, uint32_t *err) {
  using namespace eng;
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  lds_u32 *ring = (lds_u32 *)smem;                          // [RING][SLAB_WORDS]
  lds_u32 *full = ring + RING * SLAB_WORDS;                 // [RING]  epoch of the slab a slot holds (0 = none yet)
  lds_u32 *done = full + RING;                              // [RING]  consumer completions, cumulative
  lds_u32 *arrive = done + RING;                            // [RED_BUFS] arrival tickets
  lds_f32 *red = (lds_f32 *)(ring + RING * SLAB_WORDS + 32);  // [RED_BUFS][NC][SLAB_COLS]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NB = gridDim.x, bid = blockIdx.x;
  if (threadIdx.x < 32) ring[RING * SLAB_WORDS + threadIdx.x] = 0;  // full / done / arrive
  __syncthreads();  // the only workgroup barrier

  if (wave == NC) {
    // =================================================== loader ========================================================
    int q = 0;  // slabs issued so far
    bool dead = false;
    const int lrow = lane >> 3, lchunk = (lane & 7) * 4;  // piece p: word-row p*8 + lrow, words lchunk..lchunk+3 of its 32
    for (int l = 0; l < n_links; ++l) {
      const EngineLink &L = links[l];
      const int n_strips = L.n_strips, N = L.N, rows = L.K >> 3, slabs = L.slabs;
      int t = (bid - L.strip0) % NB;
      if (t < 0) t += NB;
      for (; t < n_strips; t += NB) {
        g_cu32 *W = (g_cu32 *)L.qweight + t * SLAB_COLS + lchunk;
        for (int sl = 0; sl < slabs; ++sl, ++q) {
          const int slot = q % RING;
          const uint32_t need = (uint32_t)(NC * (q / RING));  // completions of the slot's previous uses
          for (uint32_t spin = 0; !dead && lds_load(done + slot) < need; ++spin) {
            __builtin_amdgcn_s_sleep(1);
            if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 2u); dead = true; }
          }
          lds_u32 *dst = ring + slot * SLAB_WORDS;
#pragma unroll
          for (int p = 0; p < 16; ++p) {
            const int r = min(sl * (SLAB_K / 8) + p * 8 + lrow, rows - 1);  // rows past K: a harmless re-read (slice skipped)
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t *)(W + (size_t)r * N), (lds_void_t *)(dst + p * 256), 16, 0, 2);
          }
          if (q > 0) {  // the previous slab's 16 pieces are the oldest outstanding: landed once at most 16 remain
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (lane == 0) *(volatile lds_u32 *)(full + (q - 1) % RING) = (uint32_t)((q - 1) / RING + 1);
          }
        }
      }
    }
    if (q > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) *(volatile lds_u32 *)(full + (q - 1) % RING) = (uint32_t)((q - 1) / RING + 1);
    }
    return;
  }

  // ===================================================== consumers ======================================================
  const int g = lane >> 4, i = lane & 15;
  const uint32_t mask_lo = nib_mask_vgpr(), mask_hi = mask_lo << 4;
  const half8_t b_ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
  const half8_t b_mult = {(half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f, (half_t)1.f, (half_t)1.f, (half_t)16.f, (half_t)16.f};
  const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  int q = 0, sidx = 0;
  bool dead = false;
  for (int l = 0; l < n_links; ++l) {
    const EngineLink &L = links[l];
    const int n_strips = L.n_strips, N = L.N, K = L.K, slabs = L.slabs, zk = L.zero_kind;
    g_cu16 *xb = (g_cu16 *)L.x;
    g_ch *scales = (g_ch *)L.scales;
    g_ch *bias = (g_ch *)L.bias;
    const bool poll = L.x_poll != 0;
    g_cu32 *zbase = (zk == ZK_SYM) ? (g_cu32 *)L.scales : (g_cu32 *)L.qzeros;
    const int zmul = (zk == ZK_PACKED) ? (N >> 3) : (N >> 1);
    const uint32_t zsel_p = (zk == ZK_PACKED) ? 0xffffffffu : 0u, zsel_h = (zk == ZK_F16) ? 0xffffffffu : 0u;
    const uint32_t zsel_s = (zk == ZK_SYM) ? __builtin_bit_cast(uint32_t, 8.0f) : 0u;
    int t = (bid - L.strip0) % NB;
    if (t < 0) t += NB;
    for (; t < n_strips; t += NB, ++sidx) {
      const int n0 = t * SLAB_COLS;
      float4_t yacc[2] = {zero4, zero4};
      for (int sl = 0; sl < slabs; ++sl, ++q) {
        const int slot = q % RING;
        const int kbase = sl * SLAB_K + wave * 128;  // this wave's slice = one group
        const bool valid = kbase < K;                // (K % 128 == 0: a slice is whole or absent)
        uint4_t xq[4];
        half_t sc[2];
        uint32_t zraw[2];
        if (valid) {
          // ---- scale / zero words of the slice's group, and its activations (polled when they are another link's output)
          const int G = kbase >> 7;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int col = n0 + 16 * c + i;
            sc[c] = scales[(size_t)G * N + col];
            zraw[c] = zbase[(size_t)G * zmul + ((zk == ZK_PACKED) ? (col >> 3) : (col >> 1))];
          }
          g_cu16 *xs = xb + kbase + 8 * g;
          if (poll) {
            for (uint32_t spin = 0;; ++spin) {
              bool bad = false;
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                g_u64 *a = (g_u64 *)(xs + 32 * s);
                const uint64_t lo = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint64_t hi = __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xq[s] = uint4_t{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                bad = bad || has_ffff16(lo) || has_ffff16(hi);
              }
              if (__builtin_amdgcn_ballot_w64(bad) == 0 || dead) break;
              if (spin > (kSpinLimit >> 3)) { if (lane == 0) atomicOr(err, 1u); dead = true; break; }
              __builtin_amdgcn_s_sleep(2);
            }
          } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) xq[s] = *(__attribute__((address_space(1))) const uint4_t *)(xs + 32 * s);
          }
        }
        // ---- the slab: wait until the loader has published it, take this wave's 8 words per lane, hand the slot back
        const uint32_t epoch = (uint32_t)(q / RING + 1);
        for (uint32_t spin = 0; !dead && lds_load(full + slot) < epoch; ++spin) {
          __builtin_amdgcn_s_sleep(1);
          if (spin > kSpinLimit) { if (lane == 0) atomicOr(err, 4u); dead = true; }
        }
        uint32_t w[4][2];
        {
          lds_u32 *src = ring + slot * SLAB_WORDS + (wave * 16 + g) * SLAB_COLS + i;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            w[s][0] = src[(4 * s) * SLAB_COLS];
            w[s][1] = src[(4 * s) * SLAB_COLS + 16];
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the words are in registers: the slot may be refilled
        if (lane == 0) __hip_atomic_fetch_add(done + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!valid) continue;
        // ---- arithmetic: register-A form (strip.hip), one group
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
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int col = n0 + 16 * c + i;
          const float zp = (float)(((zraw[c] >> (4 * (col & 7))) + (uint32_t)L.add_zero_bias) & 15u);
          const float zh = (float)__builtin_bit_cast(half_t, (uint16_t)((col & 1) ? (zraw[c] >> 16) : (zraw[c] & 0xffffu)));
          const float zfc = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, zp) & zsel_p) | (__builtin_bit_cast(uint32_t, zh) & zsel_h) | zsel_s);
          const float sfc = (float)sc[c];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float corr = __builtin_fmaf(zfc, g_sx[r], 1024.f * g_ones[r]);
            yacc[c][r] = __builtin_fmaf(sfc, gacc[c][r] - corr, yacc[c][r]);
          }
        }
      }
      // ---- strip complete: row 0 of the two 16-column tiles -> this wave's line of the reduction buffer; last arriver finishes
      const int buf = sidx % RED_BUFS;
      lds_f32 *rb = red + (buf * NC) * SLAB_COLS;
      if (g == 0) {
        rb[wave * SLAB_COLS + i] = yacc[0][0];
        rb[wave * SLAB_COLS + 16 + i] = yacc[1][0];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      uint32_t ticket = 0;
      if (lane == 0) ticket = __hip_atomic_fetch_add(arrive + buf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ticket = __builtin_amdgcn_readfirstlane(ticket);
      if (ticket == NC - 1) {
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
    }
  }
}

bool engine_link_ok(const qllm_weight_t &w, int M, int act_dtype) {
  if (M != 1 || act_dtype != QLLM_F16 || w.bits != 4 || w.group_size != 128 || w.layout == QLLM_LAYOUT_AWQ_GEMM || w.g_idx) return false;
  if (w.N % 32 != 0 || w.K % 128 != 0 || w.K < 128) return false;
  return ((uintptr_t)w.qweight % 16 == 0) && ((uintptr_t)w.scales % 2 == 0);
}

size_t engine_lds_bytes() { return (size_t)(eng::RING * eng::SLAB_WORDS + 32) * 4 + (size_t)eng::RED_BUFS * eng::NC * eng::SLAB_COLS * 4; }

int launch_engine(const EngineLink *links_dev, int n_links, uint32_t *err, int grid, hipStream_t stream) {
  static DeviceLatch attr_done;
  if (int rc = lds_optin(attr_done, (const void *)engine_kernel)) return rc;
  hipLaunchKernelGGL(engine_kernel, dim3(grid), dim3(576), engine_lds_bytes(), stream, links_dev, n_links, err);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // namespace qllm
