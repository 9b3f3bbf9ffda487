// Strip-major (native layout) instantiations of the full-K strip decode kernel (strip_kernel.hpp): lds-slab form, M <= 4.
// The register-A forms (M >= 5) of the same layout are in strip_sm_ra.hip.
//
// Waves per block by K (as many of a launch's strips as possible should be resident at once, so that the launch's bytes are
// requested in its first microsecond: the 8-wave g128 batch-1 form takes 74 registers = three blocks per CU):
//   K <= 1024: 4 waves x one round of 8;  K <= 4096: 8 waves x one round of 16 k-steps (QLLM_SM_NW4=1: 4 waves x one round of
//   32; measured slower, 31.2 vs 30.2 us per Llama-2-7B layer);  K <= 8192: 8 waves x one round of 32;  longer: 16 waves x rounds
//   of 24 (Llama-2-7B down_proj, K = 11008: one round)
#include <stdlib.h>

#include "strip_kernel.hpp"

namespace qllm {

// waves per block of the lds-slab form, or 0: use the register-A form (host planner, capi.hip)
int strip_sm_nw(int K, int M, int group_size, int bits) {
  const int nw4 = knob("QLLM_SM_NW4", 0);
  const int T = K / 32;
  if (bits == 3) return T <= 128 ? 16 : 0;            // 16 waves x one round of 8 (longer chunks: register-A)
  if (group_size == 32) {  // a scale / zero pair per k-step: batch 1, and only when a wave's chunk is exactly one round of 8 k-steps
    if (M > 1) return 0;
    return ((T + 3) / 4 == 8) ? 4 : (((T + 15) / 16 == 8) ? 16 : 0);   // K = 928..1024 / 3616..4096; else register-A
  }
  if (group_size == 64) {                             // twice the scale / zero registers per round: short rounds, batch 1 only
    if (M > 1) return 0;
    return T <= 32 ? 4 : (T <= 128 ? 8 : 0);
  }
  if (T <= 32) return 4;
  if (T <= 128) return (nw4 && M == 1) ? 4 : 8;   // (measured: 16-wave blocks of one round of 8 lose here -- 4096 x 11008 6.45 -> 7.11 us)
  if (T <= 256 && M == 1) return 8;                   // one round of 32 (the M = 2..4 staging does not fit beside it: 16 waves)
  return 16;
}

template <int SPG, int BITS, bool DBG>
static int launch_sm_slab(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  const bool small_x = p.M == 1 && strip_xl(p.nw, 1, p.spw, 1, 1) <= 2;  // the XL = 2 instantiations are batch-1 kernels (M folded)
  const int maxs = strip_maxs(p.nw, p.spw, 1, 0, 1);
  // instantiations are limited to the forms the planner reaches AND that do not spill (tests/test_kernel_resources_cpu.py)
  // batch-1 forms: the one-round instantiation when the wave chunk is exactly one round (every Llama-class shape)
#define QLLM_SM1(NW_, MAXS_)                                                                                                   \
  return (p.spw == MAXS_) ? launch_strip_t<NW_, 1, MAXS_, SPG, 2, BITS, false, false, 1, true, DBG, true>(p, grid, lds, stream) \
                          : launch_strip_t<NW_, 1, MAXS_, SPG, 2, BITS, false, false, 1, true, DBG, false>(p, grid, lds, stream)
#define QLLM_SM8(NW_, MAXS_) return launch_strip_t<NW_, 1, MAXS_, SPG, 8, BITS, false, false, 1, true, DBG>(p, grid, lds, stream)
  if constexpr (BITS == 3) {
    if (p.nw == 16 && maxs == 8) { if (small_x) { QLLM_SM1(16, 8); } else { QLLM_SM8(16, 8); } }
  } else if constexpr (SPG == 1) {
    // 32-wide groups: only the one-round forms (48 registers; the general-round forms spill at the 128 a 16-wave block may use)
    if (small_x && p.spw == 8 && maxs == 8) {
      if (p.nw == 4) return launch_strip_t<4, 1, 8, SPG, 2, BITS, false, false, 1, true, DBG, true>(p, grid, lds, stream);
      if (p.nw == 16) return launch_strip_t<16, 1, 8, SPG, 2, BITS, false, false, 1, true, DBG, true>(p, grid, lds, stream);
    }
  } else if constexpr (SPG == 2) {
    if (small_x) {
      if (p.nw == 4 && maxs == 8) { QLLM_SM1(4, 8); }
      if (p.nw == 8 && maxs == 16) { QLLM_SM1(8, 16); }
    }
  } else {
    if (small_x) {
      if (p.nw == 4 && maxs == 8) { QLLM_SM1(4, 8); }
      if (p.nw == 4 && maxs == 32) { QLLM_SM1(4, 32); }
      if (p.nw == 8 && maxs == 16) { QLLM_SM1(8, 16); }
      if (p.nw == 8 && maxs == 32) { QLLM_SM1(8, 32); }
      if (p.nw == 16 && maxs == 8) { QLLM_SM1(16, 8); }
      if (p.nw == 16 && maxs == 24) { QLLM_SM1(16, 24); }
    } else {
      if (p.nw == 4 && maxs == 8) { QLLM_SM8(4, 8); }
      if (p.nw == 8 && maxs == 16) { QLLM_SM8(8, 16); }
      if (p.nw == 16 && maxs == 8) { QLLM_SM8(16, 8); }
      if (p.nw == 16 && maxs == 24) { QLLM_SM8(16, 24); }
    }
  }
#undef QLLM_SM1
#undef QLLM_SM8
  return set_error(QLLM_ERR_UNSUPPORTED, "internal: no strip-major slab instantiation for nw=%d round=%d g=%d bits=%d M=%d", p.nw, maxs, p.group_size,
                   BITS, p.M);
}

int launch_strip_sm(const StripParams &p, int grid, hipStream_t stream) {
  if (p.ra) return launch_strip_sm_ra(p, grid, stream);
  const size_t lds = strip_lds_bytes(p.M, p.spw, p.nw, 1, p.group_size, 0, 1);
  if (p.bits == 3 && p.group_size == 32) return set_error(QLLM_ERR_UNSUPPORTED, "3-bit strips: group sizes 64 and 128");
  if (p.bits == 3) return p.group_size == 64 ? launch_sm_slab<2, 3, false>(p, grid, lds, stream) : launch_sm_slab<4, 3, false>(p, grid, lds, stream);
  if (p.dbg) {  // diagnostics instantiation (timeline stamps): g128 only
    if (p.group_size == 128) return launch_sm_slab<4, 4, true>(p, grid, lds, stream);
    return set_error(QLLM_ERR_UNSUPPORTED, "timeline diagnostics: group size 128 only");
  }
  if (p.group_size == 32) return launch_sm_slab<1, 4, false>(p, grid, lds, stream);
  return p.group_size == 64 ? launch_sm_slab<2, 4, false>(p, grid, lds, stream) : launch_sm_slab<4, 4, false>(p, grid, lds, stream);
}

}  // namespace qllm
