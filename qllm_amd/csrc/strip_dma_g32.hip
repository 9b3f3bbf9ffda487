// strip_dma.hpp instantiations for 32-wide groups (one k-step per group; 4 bits): every slot of the ring carries the scale / zero
// words of two groups, so blocks stop at two strips and the ring is shorter (strip_dma_ring).  (Round 4: the "32g" GPTQ checkpoints at batch 2..32.)
#include "strip_dma.hpp"

namespace qllm {

template <bool BF>
static int launch_g32(const StripParams &p, int grid, hipStream_t stream) {
  if (p.bits != 4) return set_error(QLLM_ERR_UNSUPPORTED, "internal: g32 strips are 4-bit");
  if (p.M > 16) {
    if (p.cpl != 1 || p.nw != 8) return set_error(QLLM_ERR_UNSUPPORTED, "internal: two row tiles take 8-wave blocks of one strip");
    return launch_strip_dma_t<8, 1, 1, 4, BF, 2>(p, grid, stream);
  }
  if (p.cpl == 1 && p.nw == 16) return launch_strip_dma_t<16, 1, 1, 4, BF, 1>(p, grid, stream);
  if (p.nw != 8) return set_error(QLLM_ERR_UNSUPPORTED, "internal: blocks of several strips are 8 waves");
  switch (p.cpl) {
    case 1: return launch_strip_dma_t<8, 1, 1, 4, BF, 1>(p, grid, stream);
    case 2: return launch_strip_dma_t<8, 2, 1, 4, BF, 1>(p, grid, stream);
  }
  return set_error(QLLM_ERR_UNSUPPORTED, "internal: no g32 block of %d strips", p.cpl);
}

int launch_strip_dma_g32(const StripParams &p, int grid, hipStream_t stream) {
  return p.act_bf16 ? launch_g32<true>(p, grid, stream) : launch_g32<false>(p, grid, stream);
}

}  // namespace qllm
