// strip_dma.hpp instantiations for 128-wide groups (k-steps per group = 4).
#include "strip_dma_launch.hpp"

namespace qllm {

int launch_strip_dma_g128(const StripParams &p, int grid, hipStream_t stream) {
  return p.act_bf16 ? launch_sm_dma<4, true>(p, grid, stream) : launch_sm_dma<4, false>(p, grid, stream);
}

}  // namespace qllm
