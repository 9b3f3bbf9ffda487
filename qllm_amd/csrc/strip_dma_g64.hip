// strip_dma.hpp instantiations for 64-wide groups (k-steps per group = 2), incl. the fp16-zero-point forms (HQQ).
#include "strip_dma_launch.hpp"

namespace qllm {

int launch_strip_dma_g64(const StripParams &p, int grid, hipStream_t stream) {
  return p.act_bf16 ? launch_sm_dma<2, true>(p, grid, stream) : launch_sm_dma<2, false>(p, grid, stream);
}

}  // namespace qllm
