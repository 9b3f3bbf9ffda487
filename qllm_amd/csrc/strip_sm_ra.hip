// Strip-major (native layout) instantiations for more than four rows: strip_dma.hpp (own translation units) serves M = 2..32 (K a multiple of 64), and the
// register-A forms of strip_kernel.hpp for what that kernel does not take (M = 33..64, other K, long-K g64 / 3-bit chunks at batch
// 1..4, QLLM_RA_XD=0): 16-wave blocks of one 16-column strip, 8-wave blocks for 2 / 4 row tiles or 2 / 4 adjacent strips.
#include "strip_kernel.hpp"

namespace qllm {

template <int SPG, bool BF>
static int launch_sm_ra(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.M > 16) {
    if (p.bits == 3) {  // (four 3-bit row tiles need 256+ registers: the planner stops at two)
      if (p.M > 32) return set_error(QLLM_ERR_UNSUPPORTED, "internal: 3-bit strip-major strips serve M <= 32");
      return launch_strip_t<8, 1, 8, SPG, 1, 3, true, BF, 2, true>(p, grid, lds, stream);
    }
    return p.M > 32 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 4, true>(p, grid, lds, stream) : launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 2, true>(p, grid, lds, stream);
  }
  if (p.bits == 3 && p.cpl == 2) return launch_strip_t<8, 2, 8, SPG, 1, 3, true, BF, 1, true>(p, grid, lds, stream);  // (four 3-bit strips spill)
  if (p.bits == 3) {
    if constexpr (SPG == 2 && BF)  // (this one spills 7 registers: not built; callers stream the reference layout in place)
      return set_error(QLLM_ERR_UNSUPPORTED, "3-bit g64 native-layout layers with bf16 activations: no strip-major kernel");
    else
      return launch_strip_t<16, 1, 8, SPG, 1, 3, true, BF, 1, true>(p, grid, lds, stream);
  }
  if (p.cpl == 4) return launch_strip_t<8, 4, 8, SPG, 1, 4, true, BF, 1, true>(p, grid, lds, stream);  // four strips per 8-wave block
  return p.nw == 8 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 1, true>(p, grid, lds, stream)
                   : launch_strip_t<16, 1, 8, SPG, 1, 4, true, BF, 1, true>(p, grid, lds, stream);
}

// 32-wide groups (4 bits): one strip per block only -- a scale / zero pair per k-step and strip leaves no registers for four
template <bool BF>
static int launch_sm_ra_g32(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.bits != 4 || p.cpl != 1) return set_error(QLLM_ERR_UNSUPPORTED, "internal: g32 strip-major strips are 4-bit, one strip per block");
  if (p.M > 32) return launch_strip_t<8, 1, 8, 1, 1, 4, true, BF, 4, true>(p, grid, lds, stream);
  if (p.M > 16) return launch_strip_t<8, 1, 8, 1, 1, 4, true, BF, 2, true>(p, grid, lds, stream);
  return p.nw == 8 ? launch_strip_t<8, 1, 8, 1, 1, 4, true, BF, 1, true>(p, grid, lds, stream)
                   : launch_strip_t<16, 1, 8, 1, 1, 4, true, BF, 1, true>(p, grid, lds, stream);
}

int launch_strip_sm_ra(const StripParams &p, int grid, hipStream_t stream) {
  if (p.ra == 2) return p.group_size == 32 ? launch_strip_dma_g32(p, grid, stream) : (p.group_size == 64 ? launch_strip_dma_g64(p, grid, stream) : launch_strip_dma_g128(p, grid, stream));
  const size_t lds = strip_lds_bytes(p.M, p.spw, p.nw, p.cpl, p.group_size, 1, 1);
  if (p.group_size == 32) return p.act_bf16 ? launch_sm_ra_g32<true>(p, grid, lds, stream) : launch_sm_ra_g32<false>(p, grid, lds, stream);
  if (p.group_size == 64) return p.act_bf16 ? launch_sm_ra<2, true>(p, grid, lds, stream) : launch_sm_ra<2, false>(p, grid, lds, stream);
  return p.act_bf16 ? launch_sm_ra<4, true>(p, grid, lds, stream) : launch_sm_ra<4, false>(p, grid, lds, stream);
}

}  // namespace qllm
