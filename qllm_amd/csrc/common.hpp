// Shared host/device helpers for libqllm_mi355x (gfx950 only; wave64; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/qllm_mi355x.h"

namespace qllm {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kNumCU = 256;  // MI355X: the fallback of compute_units() when no device is reachable (pure-host planning)
// CUs of the current device, for launch heuristics only (capi.hip): hipDeviceAttributeMultiprocessorCount, cached; QLLM_NUM_CU
// overrides; kNumCU without a device, so that qllm_plan_describe / qllm_workspace_bytes stay deterministic in CPU-only tests.
int compute_units();

// Tuning knobs of the dispatchers.  Release build: the measured defaults -- the library reads NO environment variable except
// QLLM_NUM_CU (capi.hip) -- unless the caller overrides a planner threshold through qllm_set_knob() (round 6: the thresholds were
// measured on two models' shapes; a deployment with other shapes can move them without rebuilding.  Only the names listed in
// capi.hip's kSettable are accepted, within ranges every built kernel instantiation covers).  Lab build (`make -C qllm_amd/csrc variant
// NAME=lab DEFS=-DQLLM_LAB` -> tools/lab/libqllm_lab.so, loaded with QLLM_MI355X_LIB): QLLM_<NAME> is also read from the environment at
// EVERY call, so one process can A/B variants back to back.
extern int g_knob_overrides;                       // capi.hip: how many overrides are set (0: knob() is its default)
int knob_override(const char *name, int dflt);     // capi.hip
#ifdef QLLM_LAB
int knob(const char *name, int dflt);  // capi.hip
#else
inline int knob(const char *name, int dflt) { return __atomic_load_n(&g_knob_overrides, __ATOMIC_RELAXED) ? knob_override(name, dflt) : dflt; }
#endif

// zero-point representation, decided on the host from (layout, qzeros)
enum ZeroKind : int { ZK_PACKED = 0, ZK_F16 = 1, ZK_SYM = 2 };

// ---- host-side error plumbing (capi.hip owns the storage) ------------------------------------------------
int set_error(int code, const char *fmt, ...);
void clear_error();

#define QLLM_HIP_CHECK(expr)                                                                          \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess)                                                                             \
      return ::qllm::set_error(QLLM_ERR_LAUNCH, "%s failed: %s", #expr, hipGetErrorString(_e));       \
  } while (0)

// Per-device once-flag for per-device function attributes (hipFuncSetAttribute): a lock-free bitmask, one bit per HIP
// device ordinal (<= 64 devices per process).  A racing second caller at worst repeats the (idempotent) attribute call.
struct DeviceLatch {
  std::atomic<uint64_t> bits{0};
  bool test(int dev) const { return dev >= 0 && dev < 64 && (bits.load(std::memory_order_acquire) >> dev) & 1u; }
  void set(int dev) { if (dev >= 0 && dev < 64) bits.fetch_or(1ull << dev, std::memory_order_release); }
};

// opt a kernel into > 64 KB of dynamic LDS on the CURRENT device, once per (kernel, device)
inline int lds_optin(DeviceLatch &latch, const void *kernel_fn) {
  int dev = 0;
  QLLM_HIP_CHECK(hipGetDevice(&dev));
  if (!latch.test(dev)) {
    QLLM_HIP_CHECK(hipFuncSetAttribute(kernel_fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    latch.set(dev);
  }
  return QLLM_OK;
}

// ---- device helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t as_u32(half2_t v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ half2_t as_h2(uint32_t v) { return __builtin_bit_cast(half2_t, v); }
__device__ __forceinline__ half2_t splat2(half_t v) { return half2_t{v, v}; }

// (a & mask) | orv  -> one v_and_or_b32.  gfx950 VOP3 takes a single constant-bus operand, so with two literals
// hipcc splits it into v_and + v_or; holding the mask in a VGPR the compiler cannot constant-fold lets it select the
// fused op itself (no per-op inline asm, hence no hazard s_nops).
__device__ __forceinline__ uint32_t nib_mask_vgpr() {
  uint32_t m;
  asm volatile("v_mov_b32 %0, 0x000f000f" : "=v"(m));
  return m;
}
__device__ __forceinline__ uint32_t and_or(uint32_t a, uint32_t mask, uint32_t orv) { return (a & mask) | orv; }

// fp16 "1024 + q" magic: low nibble of each 16-bit half -> exact fp16 (1024+q).
constexpr uint32_t kMagic = 0x64006400u;  // (1024.0h, 1024.0h)
constexpr uint32_t kNibLo = 0x000f000fu;

// Per-column dequant constants for one quantisation group:
//   s2  = (s, s)
//   c2  = (-1024 s, -1024 s)      exact (power-of-two scaling)
//   zs2 = (fp16(z*s), fp16(z*s))  one rounding, as the reference's `zeros * scales`
// dequant of a magic pair r = (1024+qa, 1024+qb):
//   t = fma(r, s2, c2)  == fp16(q*s) exactly once-rounded   ((1024+q)s - 1024s = qs in exact arithmetic)
//   w = t - zs2         one rounding                        == DequantizeLinearBlockWise bit for bit
struct ColConst {
  half2_t s2, c2, zs2;
};

__device__ __forceinline__ ColConst make_col_const(half_t s, half_t z) {
  ColConst c;
  half_t zs = z * s;  // v_mul_f16: one rounding (contraction is off for this library)
  c.s2 = splat2(s);
  c.c2 = splat2((half_t)(-1024.0f) * s);
  c.zs2 = splat2(zs);
  return c;
}

__device__ __forceinline__ half2_t deq_pair(uint32_t magic_pair, const ColConst &c) {
  half2_t t = __builtin_elementwise_fma(as_h2(magic_pair), c.s2, c.c2);
  return t - c.zs2;
}

// Dequantise one GPTQ-style word (8 consecutive-k 4-bit values of one column) into a B fragment whose
// register r holds k-slots (r, r+4):  {(k0,k4),(k1,k5),(k2,k6),(k3,k7)}.  The matching A fragment must be
// permuted with a_perm_04152637().
__device__ __forceinline__ half8_t deq_word_k04(uint32_t w, const ColConst &c, uint32_t mask = kNibLo) {
  half2_t b0 = deq_pair(and_or(w, mask, kMagic), c);
  half2_t b1 = deq_pair(and_or(w >> 4, mask, kMagic), c);
  half2_t b2 = deq_pair(and_or(w >> 8, mask, kMagic), c);
  half2_t b3 = deq_pair(and_or(w >> 12, mask, kMagic), c);
  return half8_t{b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};
}

__device__ __forceinline__ half8_t a_perm_04152637(half8_t a) {
  return __builtin_shufflevector(a, a, 0, 4, 1, 5, 2, 6, 3, 7);
}
// inverse: fragment in (k0,k4,k1,k5,k2,k6,k3,k7) order -> natural k0..k7
__device__ __forceinline__ half8_t unperm_04152637(half8_t b) {
  return __builtin_shufflevector(b, b, 0, 2, 4, 6, 1, 3, 5, 7);
}

// bf16 (as raw u16 pairs in a u32) -> fp16 with round-to-nearest-even (what x.to(float16) does)
__device__ __forceinline__ half2_t bf16x2_to_h2(uint32_t v) {
  float lo = __builtin_bit_cast(float, v << 16);
  float hi = __builtin_bit_cast(float, v & 0xffff0000u);
  return half2_t{(half_t)lo, (half_t)hi};
}
__device__ __forceinline__ half8_t bf16x8_to_h8(uint4_t v) {
  half2_t a = bf16x2_to_h2(v.x), b = bf16x2_to_h2(v.y), c = bf16x2_to_h2(v.z), d = bf16x2_to_h2(v.w);
  return half8_t{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}
// fp32 -> bf16 RNE (NaN-preserving enough for our finite outputs)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// write-through (sc1) 4-byte store / L1-bypassing (sc1) load: the split-K slab protocol
// (cdna_hip_programming.md section 5, "in-launch split-K reduction", sc1 variant).
__device__ __forceinline__ void st_sc1(float *p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// extract packed zero `col` of row `zrow` (bit stream along N, `bits` wide), apply AutoGPTQ offset
__device__ __forceinline__ int packed_zero(const uint32_t *zrow, int col, int bits, int add_zero_bias) {
  const uint32_t mask = (1u << bits) - 1u;
  const int bit0 = col * bits;
  const int w = bit0 >> 5, off = bit0 & 31;
  uint64_t v = zrow[w];
  if (off + bits > 32) v |= (uint64_t)zrow[w + 1] << 32;
  return (int)(((uint32_t)(v >> off) + (uint32_t)add_zero_bias) & mask);
}

// AWQ nibble position of natural column c within its 8-column word: ORDER = {0,2,4,6,1,3,5,7}
__device__ __host__ __forceinline__ int awq_nibble_of_col(int c) { return ((c & 1) << 2) | (c >> 1); }
__device__ __host__ __forceinline__ int awq_col_of_nibble(int p) { return ((p & 3) << 1) | (p >> 2); }

// XOR swizzle of the eight 16-byte k-slots of a 128-byte tile row ([rows][64 halves] MFMA operand tiles in LDS).
// ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 (MI355X_MICROARCH.md, LDS):
// a fragment read (lane (g, i) -> row i, slot c + g) therefore mixes rows {0-3, 12-15} at slot c with rows {4-11} at
// slot c + 1 in ONE group, and the obvious (row ^ row >> 3) & 7 puts them 2-way on every bank (PMC: SQ_LDS_BANK_CONFLICT
// = 36% of SQ_LDS_IDX_ACTIVE in the prefill kernel).  This map is conflict-free for those reads, for 16-byte stores of
// 8 consecutive rows at one slot (GPTQ dequant), of 8 slots of one row (activations), and 2-way (free) for the AWQ
// 4-byte column stores; tools/lab/bank_sim.py enumerates all four patterns.
__device__ __forceinline__ int lds_row_swizzle(int row) {
  return (((row >> 1) ^ (row >> 4)) & 3) | (((row ^ (row >> 3)) & 1) << 2);
}

}  // namespace qllm
