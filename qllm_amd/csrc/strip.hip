// Full-K strip decode kernel, host side: planning helpers shared by both layouts, and the instantiations that read the
// reference's row-stream layouts (GPTQ / HQQ state-dict buffers) in place.  The kernel template is strip_kernel.hpp; the
// strip-major (native layout) instantiations are in strip_sm.hip.
#include <stdlib.h>

#include <algorithm>

#include "strip_kernel.hpp"

namespace qllm {

// (waves per block, weight loads per lane per round)
//   row-stream: 16 waves x 8 or 24, or 8 waves x 16; 64-/32-column strips always use rounds of 8 (4 / 2 dwords per load keep the
//               register budget of a 1024-thread block)
//   strip-major (sm): 4 waves x 8 or 32, 8 waves x 16 or 32, 16 waves x 8 or 24
//   register-A launches always use rounds of 8: each k-step also holds 16 B of activations per lane
int strip_maxs(int nw, int spw, int cpl, int ra, int sm) {
  if (ra || cpl >= 2) return 8;
  if (sm) {
    if (nw == 4) return spw <= 8 ? 8 : 32;
    if (nw == 8) return spw <= 16 ? 16 : 32;
    return spw <= 8 ? 8 : 24;
  }
  return nw == 8 ? 16 : (spw <= 8 ? 8 : 24);
}
static int strip_spw_pad(int nw, int spw, int cpl, int ra, int sm) { const int m = strip_maxs(nw, spw, cpl, ra, sm); return (spw + m - 1) / m * m; }
int strip_xl(int nw, int M, int spw, int cpl, int sm) { return (M * strip_spw_pad(nw, spw, cpl, 0, sm) * 4 + 63) / 64; }

template <int SPG, bool BF>
static int launch_strip_ra(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.M > 16) {  // several 16-row tiles per block: 16-column strips, 8-wave blocks (register budget)
    if (p.bits == 3)
      return p.M > 32 ? launch_strip_t<8, 1, 8, SPG, 1, 3, true, BF, 4>(p, grid, lds, stream) : launch_strip_t<8, 1, 8, SPG, 1, 3, true, BF, 2>(p, grid, lds, stream);
    return p.M > 32 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 4>(p, grid, lds, stream) : launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF, 2>(p, grid, lds, stream);
  }
  if (p.bits == 3) return launch_strip_t<16, 1, 8, SPG, 1, 3, true, BF>(p, grid, lds, stream);
  if (p.cpl == 4) return launch_strip_t<8, 4, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);  // (64-column strips: 8-wave blocks only)
  if (p.cpl == 2) return launch_strip_t<16, 2, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);
  return p.nw == 8 ? launch_strip_t<8, 1, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream)
                   : launch_strip_t<16, 1, 8, SPG, 1, 4, true, BF>(p, grid, lds, stream);
}

template <int SPG>
static int launch_strip_s(const StripParams &p, int grid, size_t lds, hipStream_t stream) {
  if (p.ra) return p.act_bf16 ? launch_strip_ra<SPG, true>(p, grid, lds, stream) : launch_strip_ra<SPG, false>(p, grid, lds, stream);
  const bool small_x = strip_xl(p.nw, p.M, p.spw, p.cpl, 0) <= 2;
  // (the host planner sends every wave chunk longer than 8 k-steps of a 3-bit or g64 layer to the register-A form: their one-round
  //  slab instantiations -- 24 loads + 12 scale / zero pairs per lane -- spill 15-46 registers and are not built)
  if (p.bits == 3) {  // 16-column strips, 16 waves, one round of 8
    if (strip_maxs(16, p.spw, 1, 0, 0) != 8) return set_error(QLLM_ERR_UNSUPPORTED, "internal: 3-bit slab strips take chunks of <= 8 k-steps");
    return small_x ? launch_strip_t<16, 1, 8, SPG, 2, 3>(p, grid, lds, stream) : launch_strip_t<16, 1, 8, SPG, 8, 3>(p, grid, lds, stream);
  }
  if (p.cpl == 4)  // 64-column strips, 8-wave blocks (two co-resident per CU), rounds of 8 k-steps
    return small_x ? launch_strip_t<8, 4, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<8, 4, 8, SPG, 8>(p, grid, lds, stream);
  if (p.cpl == 2)  // 32-column strips: 16 waves, rounds of 8 k-steps
    return small_x ? launch_strip_t<16, 2, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 2, 8, SPG, 8>(p, grid, lds, stream);
  if (p.nw == 8)
    return small_x ? launch_strip_t<8, 1, 16, SPG, 2>(p, grid, lds, stream) : launch_strip_t<8, 1, 16, SPG, 8>(p, grid, lds, stream);
  if (strip_maxs(16, p.spw, 1, 0, 0) == 8)
    return small_x ? launch_strip_t<16, 1, 8, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 1, 8, SPG, 8>(p, grid, lds, stream);
  if constexpr (SPG == 4)
    return small_x ? launch_strip_t<16, 1, 24, SPG, 2>(p, grid, lds, stream) : launch_strip_t<16, 1, 24, SPG, 8>(p, grid, lds, stream);
  else
    return set_error(QLLM_ERR_UNSUPPORTED, "internal: g64 slab strips take chunks of <= 8 k-steps");
}

// group sizes the strip kernel serves: 64 and 128 (k-steps per group 2, 4) in every layout, 32 (one k-step per group; 4 bits) in the
// native strip-major layout only (round 4: the TheBloke-style "32g" GPTQ checkpoints); others use the split-K kernel
bool strip_group_ok(int group_size, bool strip_major, int bits) {
  return group_size == 64 || group_size == 128 || (group_size == 32 && strip_major && bits == 4);
}

// row-stream waves per block: 8-wave blocks (4 per CU) when there are enough strips to need more than one round of 16-wave
// blocks (2 per CU) and K is short enough for one 16-dword round per wave; else 16 waves
int strip_nw(int K, int strips_total, int compute_units) { return (strips_total > 2 * compute_units && K / 32 <= 8 * 16) ? 8 : 16; }

// k-steps per wave: all of K over nw waves, rounded up to whole groups
int strip_spw(int K, int group_size, int nw) {
  const int T = K / 32, spg = group_size / 32;
  int spw = (T + nw - 1) / nw;
  return (spw + spg - 1) / spg * spg;
}

size_t strip_lds_bytes(int M, int spw, int nw, int cpl, int group_size, int ra, int sm) {
  const size_t red = (size_t)nw * M * 16 * cpl * sizeof(float);
  // register-A: only the cross-wave reduction buffer; ra == 2 (strip_dma.hpp): the waves' activation rings, 8 KB per 16-row tile,
  // which the reduction buffer re-uses
  // (+ behind them, round 5, the waves' scale / zero-point tables: 2 x groups x strips x 32 bytes, in whole 256-byte pieces)
  if (ra == 2) {
    const int spg = group_size / 32, mt = M > 16 ? 2 : 1;
    // (strip_dma.hpp, strip_dma_table_groups: rounds x groups per round, the larger of the rings the form may run)
    const int ns1 = (cpl >= 2 || mt >= 2) ? 2 : 3, r4 = (spw + 7) / 8 * 8 / spg, r3 = (spw + 5) / 6 * 6 / spg;
    const int ngw = spg == 1 ? (spw + 2 * ns1 - 1) / (2 * ns1) * (2 * ns1) : (spg == 2 ? std::max(r3, r4) : r4);
    return std::max(red, (size_t)nw * (M > 16 ? 2 : 1) * 8192 + (size_t)nw * 2 * (((size_t)ngw * cpl * 32 + 255) & ~(size_t)255));
  }
  if (ra) return red;
  const int pad = strip_spw_pad(nw, spw, cpl, 0, sm);
  const int groups = pad / (group_size / 32);  // groups per wave chunk
  return red + (size_t)nw * M * (pad * 32 + 8) * sizeof(half_t) +
         (size_t)nw * groups * 16 * 8;  // (Sx, Sx') float2 per (group, row), 16 rows per group
}

// row-stream columns per lane, from measurements on Llama-2-7B shapes (tools/kbench.py --grouped, us per launch, cpl 1/2/4):
//   q/k/v 12288 cols: 11.7 / 10.6 / 8.5    gate/up 22016 cols: 18.7 / 16.1 / 15.6
//   o 4096 cols: 5.3 / 6.1 / 7.0            down 4096 cols (K=11008): 10.1 / 11.2 / 14.7
// -> 64-column strips (256-byte row segments) as soon as they alone give >= 5/8 of the CUs a block, else 16-column strips.
// Llama-2-70B shapes and their 8-way shards (M = 1, us, cpl 1 / 2 / 4; tools/narrow_ab.py, profiles/r02_narrow_shapes.md):
//   8192 -> 2 x 3584: 18.0 / 10.1 / 11.5    8192 -> 8192: 17.6 / 10.9 / 12.3    3584 -> 8192: 7.5 / 6.5 / 7.4    1024 -> 8192: 7.2 / 5.9 / 5.1
// -> 32-column strips when they give about 3/4 of the CUs a block and the 64-column ones do not (M <= 4 only).
// (the thresholds were measured as 160 and 190 blocks on 256 CUs; they scale with the device's CU count)
int strip_cpl(int cols_total, bool all_mult64, bool all_mult32, int compute_units) {
  if (all_mult64 && cols_total / 64 >= compute_units * 5 / 8) return 4;
  if (all_mult32 && cols_total / 32 >= compute_units * 190 / 256) return 2;
  return 1;
}

// activation staging budget: at most 8 sixteen-byte chunks per lane
bool strip_x_ok(int M, int spw, int nw, int cpl, int sm) { return strip_xl(nw, M, spw, cpl, sm) <= 8; }

int launch_strip(const StripParams &p, int grid, hipStream_t stream) {
  if (p.sm) return launch_strip_sm(p, grid, stream);
  const size_t lds = strip_lds_bytes(p.M, p.spw, p.nw, p.cpl, p.group_size, p.ra, 0);
  if (p.group_size == 64) return launch_strip_s<2>(p, grid, lds, stream);
  return launch_strip_s<4>(p, grid, lds, stream);
}

}  // namespace qllm
