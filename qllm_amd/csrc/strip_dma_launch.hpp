// Host-side instantiation switch of strip_dma.hpp, shared by its two translation units (one per group size, so that they compile in
// parallel: every instantiation is a 500-instruction straight-line loop).
#pragma once
#include "strip_dma.hpp"

namespace qllm {

// M = 5..32 with the activations staged through LDS by DMA (strip_dma.hpp; host planner: ra == 2): 16-wave blocks of one strip,
// 8-wave blocks of 1 / 2 / 3 / 4 / 6 strips (3 bits: 1 / 2 / 3 / 4, and 6 at 64-wide groups when every layer has fp16 zero points),
// 8-wave blocks of one strip and two row tiles
template <int SPG, bool BF>
static int launch_sm_dma(const StripParams &p, int grid, hipStream_t stream) {
  if (p.M > 16) {
    if (p.cpl != 1 || p.nw != 8) return set_error(QLLM_ERR_UNSUPPORTED, "internal: two row tiles take 8-wave blocks of one strip");
    if (p.bits == 3) return launch_strip_dma_t<8, 1, SPG, 3, BF, 2>(p, grid, stream);
    return launch_strip_dma_t<8, 1, SPG, 4, BF, 2>(p, grid, stream);
  }
  if (p.cpl == 1 && p.nw == 16) return p.bits == 3 ? launch_strip_dma_t<16, 1, SPG, 3, BF, 1>(p, grid, stream) : launch_strip_dma_t<16, 1, SPG, 4, BF, 1>(p, grid, stream);
  if (p.nw != 8) return set_error(QLLM_ERR_UNSUPPORTED, "internal: blocks of several strips are 8 waves");
  if (p.bits == 3) {
    switch (p.cpl) {
      case 1: return launch_strip_dma_t<8, 1, SPG, 3, BF, 1>(p, grid, stream);
      case 2: return launch_strip_dma_t<8, 2, SPG, 3, BF, 1>(p, grid, stream);
      case 3: return launch_strip_dma_t<8, 3, SPG, 3, BF, 1>(p, grid, stream);
      case 4: return launch_strip_dma_t<8, 4, SPG, 3, BF, 1>(p, grid, stream);
      case 6:
        if constexpr (SPG == 2) {  // (HQQ 3-bit gate/up at batch 2..16: one round of blocks instead of two -- the planner checks the zero kinds)
          bool all_f16 = true;
          for (int i = 0; i < p.n_prob; ++i) all_f16 = all_f16 && p.prob[i].zero_kind == ZK_F16;
          if (all_f16) return launch_strip_dma_z<8, 6, SPG, 3, BF, 1, true>(p, grid, stream);
        }
        break;
    }
  } else {
    switch (p.cpl) {
      case 1: return launch_strip_dma_t<8, 1, SPG, 4, BF, 1>(p, grid, stream);
      case 2: return launch_strip_dma_t<8, 2, SPG, 4, BF, 1>(p, grid, stream);
      case 3: return launch_strip_dma_t<8, 3, SPG, 4, BF, 1>(p, grid, stream);
      case 4: return launch_strip_dma_t<8, 4, SPG, 4, BF, 1>(p, grid, stream);
      case 6: return launch_strip_dma_t<8, 6, SPG, 4, BF, 1>(p, grid, stream);
    }
  }
  return set_error(QLLM_ERR_UNSUPPORTED, "internal: no %d-bit block of %d strips", p.bits, p.cpl);
}

}  // namespace qllm
