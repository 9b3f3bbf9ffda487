// Host side of the batch-1 native-layout decode kernel (strip1_kernel.hpp): the shape table and the instantiations.
#include "strip1_kernel.hpp"

namespace qllm {

// (waves per block, k-steps per wave) for T = K / 32 k-steps, or false: not served (the general strip kernel takes the call).
// One round per wave: NW * MAXS >= T >= MAXS.  Measured choices: profiles/r05_decode_bisect.md.
// `blocks`: 16-column strips of the launch (all layers); `cus`: compute units.
bool strip1_shape(int K, int blocks, int cus, int *nw, int *maxs) {
  if (K % 128 != 0) return false;  // (whole 16-byte windows of x per four k-steps; 64-wide groups: the same table)
  const int T = K / 32;
  if (T < 8) return false;
  int w, m;
  if (T <= 32) { w = 4; m = 8; }
  else if (T <= 64) { w = 4; m = 16; }
  // one block per CU at most (o_proj): four waves x 32 k-steps, one per SIMD -- 3.54 vs 3.78 us (profiles/r05_decode_bisect.md); wider
  // launches lose with it (gate/up 9.89 vs 9.57)
  else if (T <= 128 && blocks <= cus && knob("QLLM_S1_NW4", 1)) { w = 4; m = 32; }
  else if (T <= 128) { w = 8; m = 16; }
  else if (T <= 192) { w = 8; m = 24; }
  else if (T <= 256) { if (knob("QLLM_S1_T256_NW16", 0)) { w = 16; m = 16; } else { w = 8; m = 32; } }
  else if (T <= 384) { w = 16; m = 24; }
  else if (T <= 512) { w = 16; m = 32; }
  // round 6 (profiles/r06_shape_table.md: Qwen2-7B's down_proj, K = 18944, sat on the general strip kernel at 0.33 of 8 TB/s): rounds of
  // 40 .. 64 k-steps, three / four accumulator sets -- K up to 32768 (Llama-2-70B's unsharded down_proj is K = 28672)
  else if (T <= 640) { w = 16; m = 40; }
  else if (T <= 768) { w = 16; m = 48; }
  else if (T <= 896) { w = 16; m = 56; }
  else if (T <= 1024) { w = 16; m = 64; }
  else return false;
  // No dead waves where an odd wave count is built: K = 11008 is 344 k-steps = 14 1/3 rounds of 24 -- the sixteenth wave of a 16 x 24
  // block owns nothing (it re-reads its neighbour's words and multiplies zeros); K = 3584 (a Llama-2-70B TP = 8 down_proj shard) is
  // exactly 7 x 16.  The cross-wave sum is a pairwise tree over NW values: the dead wave contributed an exact zero, same bits.
  const int need = (T + m - 1) / m;
  if (need < w && knob("QLLM_S1_ODD", 1) && ((need == 7 && m == 16) || (need == 15 && m == 24))) w = need;
  *nw = w;
  *maxs = m;
  return true;
}

template <int NW, int MAXS, bool EXACT, bool DBG, bool G64 = false, int MR = 1, bool B3 = false>
static int launch_t(const Strip1Params &p, dim3 grid, hipStream_t stream) {
  constexpr int lds_bytes = strip1_lds_bytes<NW, MAXS, MR>();
  static_assert(lds_bytes <= 160 * 1024, "LDS of one CU");
  if constexpr (lds_bytes > 64 * 1024) {  // (rounds of 56 / 64 k-steps x 16 waves, or four batch rows: the staged activations)
    static DeviceLatch attr_done;
    if (int rc = lds_optin(attr_done, (const void *)strip1_kernel<NW, MAXS, EXACT, 2, 4, DBG, false, G64, MR, B3>)) return rc;
  }
  hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, DBG, false, G64, MR, B3>), grid, dim3(NW * 64), lds_bytes, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

template <int NW, int MAXS>
static int launch_e(const Strip1Params &p, dim3 grid, hipStream_t stream) {
  if (p.bits3) {  // 3-bit layers (round 6): batch 1, 64- and 128-wide groups, rounds of up to 32 k-steps (two word loads per k-step)
    if constexpr (MAXS <= 32) {
      if (p.group64)
        return (NW * MAXS == p.T) ? launch_t<NW, MAXS, true, false, true, 1, true>(p, grid, stream) : launch_t<NW, MAXS, false, false, true, 1, true>(p, grid, stream);
      return (NW * MAXS == p.T) ? launch_t<NW, MAXS, true, false, false, 1, true>(p, grid, stream) : launch_t<NW, MAXS, false, false, false, 1, true>(p, grid, stream);
    } else {
      return set_error(QLLM_ERR_UNSUPPORTED, "internal: 3-bit layers on the batch-1 kernel stop at K = 16384");
    }
  }
  if (p.M > 1) {  // batches 2..4 (round 6): the four-row forms, 128-wide groups, rounds of up to 32 k-steps (LDS: four staged rows per wave)
    if constexpr (MAXS <= 32)
      return (NW * MAXS == p.T) ? launch_t<NW, MAXS, true, false, false, 4>(p, grid, stream) : launch_t<NW, MAXS, false, false, false, 4>(p, grid, stream);
    else
      return set_error(QLLM_ERR_UNSUPPORTED, "internal: four batch rows on the batch-1 kernel stop at K = 16384");
  }
  if (p.group64) {  // 64-wide groups (round 6); rounds of up to 48 k-steps (56 spills a register with sixteen group addresses more)
    if constexpr (MAXS <= 48)
      return (NW * MAXS == p.T) ? launch_t<NW, MAXS, true, false, true>(p, grid, stream) : launch_t<NW, MAXS, false, false, true>(p, grid, stream);
    else
      return set_error(QLLM_ERR_UNSUPPORTED, "internal: 64-wide groups on the batch-1 kernel stop at K = 24576");
  }
  return (NW * MAXS == p.T) ? launch_t<NW, MAXS, true, false>(p, grid, stream) : launch_t<NW, MAXS, false, false>(p, grid, stream);
}

template <int NW, int MAXS>
static int launch_ar(const Strip1Params &p, int n_strips, hipStream_t stream) {
  constexpr int lds_bytes = strip1_lds_bytes<NW, MAXS>();
  if (NW * MAXS == p.T) hipLaunchKernelGGL((strip1_kernel<NW, MAXS, true, 2, 4, false, true>), dim3(n_strips, 1), dim3(NW * 64), lds_bytes, stream, p);
  else hipLaunchKernelGGL((strip1_kernel<NW, MAXS, false, 2, 4, false, true>), dim3(n_strips, 1), dim3(NW * 64), lds_bytes, stream, p);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

// the row-parallel layer fused with its one-shot all-reduce: the forms of the Llama-class row-parallel shards (K / world per rank)
int launch_strip1_allreduce(const Strip1Params &p, int nw, int maxs, int n_strips, hipStream_t stream) {
  if (nw == 4 && maxs == 8) return launch_ar<4, 8>(p, n_strips, stream);
  if (nw == 4 && maxs == 16) return launch_ar<4, 16>(p, n_strips, stream);
  if (nw == 4 && maxs == 32) return launch_ar<4, 32>(p, n_strips, stream);
  if (nw == 8 && maxs == 16) return launch_ar<8, 16>(p, n_strips, stream);
  if (nw == 7 && maxs == 16) return launch_ar<7, 16>(p, n_strips, stream);
  if (nw == 8 && maxs == 24) return launch_ar<8, 24>(p, n_strips, stream);
  if (nw == 8 && maxs == 32) return launch_ar<8, 32>(p, n_strips, stream);
  if (nw == 16 && maxs == 16) return launch_ar<16, 16>(p, n_strips, stream);
  if (nw == 16 && maxs == 24) return launch_ar<16, 24>(p, n_strips, stream);
  if (nw == 15 && maxs == 24) return launch_ar<15, 24>(p, n_strips, stream);
  if (nw == 16 && maxs == 32) return launch_ar<16, 32>(p, n_strips, stream);   // K per rank up to 16384 (Llama-2-70B down_proj at TP = 2)
  return set_error(QLLM_ERR_UNSUPPORTED, "fused all-reduce: no batch-1 instantiation for nw=%d round=%d", nw, maxs);
}

int launch_strip1(const Strip1Params &p, int nw, int maxs, int n_prob, int max_strips, hipStream_t stream) {
  const dim3 grid(max_strips, n_prob);
  if (p.dbg && !p.group64 && p.M == 1 && !p.bits3) {  // diagnostics instantiations (timeline stamps): the two Llama-2-7B forms
    if (nw == 8 && maxs == 16 && p.T == 128) return launch_t<8, 16, true, true>(p, grid, stream);
    if (nw == 15 && maxs == 24 && p.T < 360) return launch_t<15, 24, false, true>(p, grid, stream);
  }
  if (nw == 4 && maxs == 8) return launch_e<4, 8>(p, grid, stream);
  if (nw == 4 && maxs == 16) return launch_e<4, 16>(p, grid, stream);
  if (nw == 4 && maxs == 32) return launch_e<4, 32>(p, grid, stream);
  if (nw == 8 && maxs == 16) return launch_e<8, 16>(p, grid, stream);
  if (nw == 7 && maxs == 16) return launch_e<7, 16>(p, grid, stream);
  if (nw == 8 && maxs == 24) return launch_e<8, 24>(p, grid, stream);
  if (nw == 8 && maxs == 32) return launch_e<8, 32>(p, grid, stream);
  if (nw == 16 && maxs == 16) return launch_e<16, 16>(p, grid, stream);
  if (nw == 16 && maxs == 24) return launch_e<16, 24>(p, grid, stream);
  if (nw == 15 && maxs == 24) return launch_e<15, 24>(p, grid, stream);
  if (nw == 16 && maxs == 32) return launch_e<16, 32>(p, grid, stream);
  if (nw == 16 && maxs == 40) return launch_e<16, 40>(p, grid, stream);
  if (nw == 16 && maxs == 48) return launch_e<16, 48>(p, grid, stream);
  if (nw == 16 && maxs == 56) return launch_e<16, 56>(p, grid, stream);
  if (nw == 16 && maxs == 64) return launch_e<16, 64>(p, grid, stream);
  return set_error(QLLM_ERR_UNSUPPORTED, "internal: no batch-1 instantiation for nw=%d round=%d", nw, maxs);
}

}  // namespace qllm
