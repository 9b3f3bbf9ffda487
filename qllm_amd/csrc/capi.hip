// C-ABI entry points of libqllm_mi355x.so: argument validation (the reference's TORCH_CHECK / invalid_argument
// sites: /root/reference/csrc/ort_cuda/ort_ops.cc:64-73,99-107; csrc/awq_cuda/quantization/gemm_cuda_gen.cu:1128-1135),
// kernel selection, workspace carving.  No allocation, no host sync, no global mutable state beyond thread-local
// error text and a few env-derived tuning constants read once.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "kernels.hpp"

namespace qllm {

// ---- error plumbing ----------------------------------------------------------------------------------------
static thread_local char g_err[512] = {0};

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
void clear_error() { g_err[0] = 0; }

// ---- tuning knobs: the measured defaults, qllm_set_knob() overrides, and -- lab builds only -- the environment (common.hpp) ---------
static int env_int(const char *name, int dflt) {
  const char *v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
// the planner thresholds a caller may move in a RELEASE build, with the range every built kernel instantiation covers
struct Settable {
  const char *name;
  int lo, hi;
  int value, set;
};
static Settable kSettable[] = {
    {"QLLM_STRIP1_3BIT", 0, 1, 0, 0},          // 0: 3-bit layers at batch 1 on the general strip kernel
    {"QLLM_STRIP1_MAX_M", 1, 4, 0, 0},         // 1: batches 2..4 on strip_dma instead of the batch-1 kernel's four-row forms
    {"QLLM_STRIP1", 0, 2, 0, 0},               // 0: batch-1 calls on the general strip kernel (the round-4 path); 2: only 128-wide groups on the batch-1 kernel
    {"QLLM_PANEL", 0, 1, 0, 0},                // 0: no panel kernel (strips to 32 rows, the 256-row tiles above)
    {"QLLM_PANEL_MIN_M", 17, 129, 0, 0},       // single layers: rows from which the panel kernel serves (no form below 17 rows)
    {"QLLM_PANEL_GROUP_MIN_M", 17, 129, 0, 0}, // sibling groups: rows from which ONE panel launch serves the group
    {"QLLM_GEMM2", 0, 1, 0, 0},                // 0: no 256-row register-staged tiles (the 128 x 128 kernel instead)
    {"QLLM_GEMM3", 0, 1, 0, 0},                // 0: no wave-specialised prefill kernel
    {"QLLM_GEMM2_MIN_M", 33, 1 << 30, 0, 0},   // rows from which the 256-row tiles serve what the strips leave alone
    {"QLLM_GEMM3_MIN_M", 0, 1 << 30, 0, 0},    // rows from which gemm3 takes over from gemm2 (0: the measured line, 384 / 768)
    {"QLLM_GEMM2_SPLITK", 0, 1, 0, 0},         // 0: never split K over blocks in the tile GEMMs
    {"QLLM_GEMM3_TAIL", 0, 1, 0, 0},           // 0: no K-split of the ragged last round of tiles (gemm3.hip, round 6)
    {"QLLM_GEMM3_GROUP", 0, 1, 0, 0},          // 0: no grouped launches of the prefill kernel (q/k/v, gate/up run layer by layer from 129 rows)
    {"QLLM_GEMM3_BF16", 0, 1, 0, 0},           // 0: bf16 prefill through the fp16 conversion pre-pass (the reference's shim) instead of bf16 MFMA
    {"QLLM_SKINNY_MAX_M", 0, 64, 0, 0},        // rows up to which the split-K decode kernel serves the reference layouts in place
    {"QLLM_STRIP_MIN", 0, 1 << 20, 0, 0},      // fewest 16-column strips the full-K strip kernels take (0: never)
    {"QLLM_BITGEMV", 0, 1, 0, 0},              // 0: 2 / 5 / 6 / 7 / 8-bit decode calls refused (callers then dequantise + GEMM: the reference's branch)
};
int g_knob_overrides = 0;
int knob_override(const char *name, int dflt) {
  for (const Settable &k : kSettable)
    if (k.set && strcmp(k.name, name) == 0) return k.value;
  return dflt;
}
#ifdef QLLM_LAB
int knob(const char *name, int dflt) { return g_knob_overrides ? knob_override(name, env_int(name, dflt)) : env_int(name, dflt); }
#endif
// CUs of the current device (launch heuristics only); 256 (MI355X) when no device is reachable, so that the pure-host
// planners (qllm_plan_describe, qllm_workspace_bytes) stay deterministic without a GPU.  QLLM_NUM_CU overrides.
int compute_units() {
  static int v = [] {
    int e = env_int("QLLM_NUM_CU", 0);
    if (e > 0) return e;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) return n;
    (void)hipGetLastError();
    return kNumCU;
  }();
  return v;
}
static int skinny_target_waves() {
  const int v = knob("QLLM_SKINNY_WAVES", 2048);
  return v;
}
static int skinny_awq_w(int M) {
  const int v = knob("QLLM_SKINNY_AWQ_W", 1);
  return (M <= 16 && v == 2) ? 2 : 1;
}
static int skinny_max_m() {
  const int v = knob("QLLM_SKINNY_MAX_M", 64);
  return v > 64 ? 64 : v;
}

constexpr size_t kCounterBytes = 16384;  // 4096 column-tile arrival counters at the head of the workspace
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- validation ----------------------------------------------------------------------------------------------
static bool is_native(const qllm_weight_t &w) { return w.layout == QLLM_LAYOUT_NATIVE || w.layout == QLLM_LAYOUT_NATIVE_F16Z; }
static int zero_kind_of(const qllm_weight_t &w) {
  if (w.layout == QLLM_LAYOUT_HQQ || w.layout == QLLM_LAYOUT_NATIVE_F16Z) return ZK_F16;
  if (w.qzeros == nullptr) return ZK_SYM;
  return ZK_PACKED;
}

// shapes the strip-major native layout can hold (include/qllm_mi355x.h)
static int native_shape_ok(int bits, int K, int N, int group_size, bool packed_zeros) {
  if (bits != 4 && bits != 3) return set_error(QLLM_ERR_UNSUPPORTED, "the native layout holds 3- and 4-bit weights (got %d)", bits);
  if (K % 32 != 0 || N % 16 != 0) return set_error(QLLM_ERR_UNSUPPORTED, "the native layout needs K %% 32 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
  if (group_size % 32 != 0) return set_error(QLLM_ERR_UNSUPPORTED, "the native layout needs group_size %% 32 == 0 (got %d)", group_size);
  if (bits == 3 && packed_zeros && N % 32 != 0) return set_error(QLLM_ERR_UNSUPPORTED, "3-bit packed zero points need N %% 32 == 0 (N=%d)", N);
  return QLLM_OK;
}

static int validate_weight(const qllm_weight_t *w) {
  if (!w) return set_error(QLLM_ERR_INVALID, "weight descriptor is NULL");
  if (!w->qweight || !w->scales) return set_error(QLLM_ERR_INVALID, "qweight/scales must not be NULL");
  if (w->layout < QLLM_LAYOUT_GPTQ || w->layout > QLLM_LAYOUT_NATIVE_F16Z) return set_error(QLLM_ERR_INVALID, "unknown layout %d", w->layout);
  if (w->bits < 2 || w->bits > 8) return set_error(QLLM_ERR_INVALID, "bits must be >= 2 and <= 8 (got %d)", w->bits);
  if (w->K <= 0 || w->N <= 0) return set_error(QLLM_ERR_INVALID, "in_features/out_features must be >= 1 (K=%d N=%d)", w->K, w->N);
  if (w->group_size < 8) return set_error(QLLM_ERR_INVALID, "groupsize must be >= 8 (got %d)", w->group_size);
  if (w->layout == QLLM_LAYOUT_AWQ_GEMM) {
    if (w->bits != 4) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout is 4-bit only (got %d)", w->bits);
    if (w->N % 8 != 0) return set_error(QLLM_ERR_INVALID, "OC is not multiple of pack_num = 8 (N=%d)", w->N);
    if (w->K % w->group_size != 0) return set_error(QLLM_ERR_INVALID, "IC is not multiple of group size (K=%d g=%d)", w->K, w->group_size);
    if (!w->qzeros) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout needs qzeros");
    if (w->g_idx) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout has no act-order (g_idx must be NULL)");
  } else if (is_native(*w)) {
    if (int rc = native_shape_ok(w->bits, w->K, w->N, w->group_size, w->layout == QLLM_LAYOUT_NATIVE && w->qzeros)) return rc;
    if (w->K % w->group_size != 0) return set_error(QLLM_ERR_INVALID, "the native layout holds whole groups (K=%d g=%d)", w->K, w->group_size);
    if (w->g_idx) return set_error(QLLM_ERR_INVALID, "the native layout has no act-order (g_idx must be NULL: sort the rows by group first)");
    if (w->layout == QLLM_LAYOUT_NATIVE_F16Z && !w->qzeros) return set_error(QLLM_ERR_INVALID, "NATIVE_F16Z layout needs fp16 qzeros");
  } else {
    if ((w->K * w->bits) % 32 != 0) return set_error(QLLM_ERR_INVALID, "in_features*bits must be a multiple of 32 (K=%d bits=%d)", w->K, w->bits);
    if (w->layout == QLLM_LAYOUT_HQQ && !w->qzeros) return set_error(QLLM_ERR_INVALID, "HQQ layout needs fp16 qzeros");
    if (w->layout == QLLM_LAYOUT_HQQ && w->g_idx) return set_error(QLLM_ERR_INVALID, "HQQ layout has no act-order (g_idx must be NULL)");
    if (w->layout == QLLM_LAYOUT_GPTQ && w->qzeros && (w->N * w->bits) % 32 != 0)
      return set_error(QLLM_ERR_INVALID, "out_features*bits must be a multiple of 32 for packed zeros (N=%d bits=%d)", w->N, w->bits);
  }
  if (w->add_zero_bias < 0 || w->add_zero_bias > 1) return set_error(QLLM_ERR_INVALID, "add_zero_bias must be 0 or 1");
  return QLLM_OK;
}

static bool fused_common_ok(const qllm_weight_t &w) {
  return !is_native(w) && w.bits == 4 && w.N % 8 == 0 && w.group_size % 8 == 0 && ((uintptr_t)w.qweight % 16 == 0) &&
         ((uintptr_t)w.scales % 16 == 0) && (!w.qzeros || (uintptr_t)w.qzeros % 8 == 0);
}
static bool skinny_ok(const qllm_weight_t &w, int M) {
  return M <= skinny_max_m() && fused_common_ok(w) && w.K % 32 == 0 && w.g_idx == nullptr;
}
static int strip_min_strips() {
  // measured (profiles/r02_narrow_shapes.md): even 8-80 strips beat the split-K kernel's three dependent round trips
  // (K=8192, N=1024+128+128: 15.2 -> 9.8 us; K=4096, N=1024: 11.7 -> 4.8 us)
  const int v = knob("QLLM_STRIP_MIN", 8);
  return v;
}
// full-K strip kernel: row-stream layouts, M <= 64 (17..64: several 16-row tiles per block), enough 16-column strips to
// cover the 256 CUs
struct StripPlan {
  int cpl, nw, spw, ra, sm;
  int one_nw, one_maxs;  // != 0: the batch-1 kernel (strip1_kernel.hpp) with this many waves x k-steps per wave
};
// diagnostics: timeline buffer handed to the native-layout decode launches (24 x u64 per launch), see qllm_debug_timeline()
static uint64_t *g_timeline = nullptr;
static int g_timeline_slots = 0, g_timeline_next = 0;

// the mid-batch panel kernel (panel.hip): single native layers from QLLM_PANEL_MIN_M (17) rows to 128.  us per linear, strips -> panel
// (profiles/r04_mid_m.md): M = 32: 23.8 -> 15.9 (4096 x 11008), 20.8 -> 16.0 (11008 x 4096).
// Round 5 (profiles/r05_batch16.md, tools/rounds5/g13_down_sweep.sh), after the strips' scale / zero tables moved to LDS:
//   * 9..16 rows where K >= 2 N (down_proj) went BACK to the strips: 15.0 -> 12.6-14.4 us (HQQ g64 4 bits), 16.5 -> 14.1-15.0 (3 bits),
//     13.5 -> 11.9-13.1 (g128): the split-K fix-up of the panel costs more than the one-strip blocks' x traffic now;
//   * 17..32 rows on layers of up to 4096 x 4096 (o_proj) stay on the two-row-tile strips: 9.1-9.7 -> 7.7-8.9 us (3 bits 10.3 -> 8.2-9.9).
static bool panel_rows_ok(int M, int K, int N) {
  if (M < knob("QLLM_PANEL_MIN_M", 17)) return false;  // (lab builds only: below 17 rows the release library has no panel form)
  if (M <= 32 && K <= 4096 && N <= 4096 && !knob("QLLM_PANEL_SMALL", 0)) return false;
  return true;
}

static bool strip_plan(const qllm_weight_t *w, int n, int M, StripPlan *plan) {
  plan->one_nw = plan->one_maxs = 0;
  // measured (graph replay, us; split-K kernel -> strips with 2 / 4 row tiles): M=32: 4096x4096 28.3 -> 13.4, 4096x11008 54.9 -> 36.3,
  // 11008x4096 50.1 -> 30.3; M=64: 33.8 -> 22.6, 52.5 -> 61.5, 48.1 -> 53.9 -- four row tiles only pay on the small shape
  // M = 33..64 (four row tiles), us per linear, split-K kernel -> strips (profiles/r02_mid_m.md): 4096x4096 26.8/31.4/33.5 ->
  // 17.6/19.8/22.6 at M = 33/48/64; on the 11008-wide shapes the strips lose at M >= 48 (43.9 -> 55.1, 41.2 -> 49.8): small shape only
  const int max_m = knob("QLLM_STRIP_MAX_M", 0);
  const bool sm = is_native(w[0]);
  {
    int cols_all = 0;
    for (int i = 0; i < n; ++i) cols_all += w[i].N;
    // (native layout, round 3, profiles/r03_mid_m.md: four row tiles 17-21 us vs gemm2 27-28 on 4096 x 4096; 46-59 and 43-54 vs
    //  40 on the 11008-wide shapes: the same line as for the reference layouts)
    const int lim = max_m ? max_m : ((w[0].K <= 4096 && cols_all <= 4096 && !(sm && w[0].bits == 3)) ? 64 : 32);
    if (M > lim) return false;
    // (single native 4-bit layers: the panel kernel takes over where it is served -- panel.hip; grouped launches stay here)
    if (n == 1 && sm && knob("QLLM_PANEL", 1) && panel_rows_ok(M, w[0].K, w[0].N) && !w[0].g_idx &&
        panel_shape_ok(M, w[0].K, w[0].N, w[0].group_size, w[0].bits))
      return false;  // (M <= 64 here)
    // (groups from 17 rows: ONE grouped launch of the panel kernel -- q/k/v 21.8-23.2 us on the two-row-tile strips, 25.2 layer by layer;
    //  gate/up 40-43.5 / 29.4.  profiles/r04_mid_m.md)
    if (n > 1 && sm && knob("QLLM_PANEL", 1) && M >= knob("QLLM_PANEL_GROUP_MIN_M", 17)) {
      bool all_ok = true;
      for (int i = 0; i < n; ++i)
        // (the conditions of panel_layers_ok: a group the panel kernel will NOT take -- e.g. fp16 and packed zero points mixed -- stays
        //  on the strips, which decode zero points per layer)
        all_ok = all_ok && !w[i].g_idx && w[i].bits == w[0].bits && is_native(w[i]) && panel_shape_ok(M, w[i].K, w[i].N, w[i].group_size, w[i].bits) &&
                 (zero_kind_of(w[i]) == ZK_F16) == (zero_kind_of(w[0]) == ZK_F16);
      if (all_ok) return false;
    }
  }
  if (M > 64 || strip_min_strips() <= 0) return false;
  const int bits = w[0].bits;
  if (bits != 4 && bits != 3) return false;
  if (!strip_group_ok(w[0].group_size, sm, bits)) return false;
  const bool g32 = w[0].group_size == 32;  // (native layout, 4 bits: one strip per block, lds-slab at batch 1, register-A above)
  for (int i = 0; i < n; ++i) {
    if (w[i].bits != bits || w[i].K % 32 != 0 || w[i].g_idx || is_native(w[i]) != sm) return false;
    // 3-bit: fp16 (HQQ), symmetric, or packed zero points (a column's field may straddle two words of the N*3/32-word row)
    if (bits == 3 && !sm && !(w[i].layout == QLLM_LAYOUT_HQQ || w[i].layout == QLLM_LAYOUT_GPTQ)) return false;
    if (bits == 3 && w[i].layout == QLLM_LAYOUT_GPTQ && w[i].qzeros && w[i].N % 32 != 0) return false;
    if ((uintptr_t)w[i].qweight % 16 || (uintptr_t)w[i].scales % 16 || (w[i].qzeros && (uintptr_t)w[i].qzeros % 8)) return false;
  }
  int cols = 0;
  bool m64 = true, m32 = true;
  for (int i = 0; i < n; ++i) {
    if (w[i].layout == QLLM_LAYOUT_AWQ_GEMM || w[i].N % 16 != 0) return false;
    cols += w[i].N;
    m64 = m64 && (w[i].N % 64 == 0);
    m32 = m32 && (w[i].N % 32 == 0);
  }
  // rows from which the activation slab in LDS (staged once per block: M*K/8 chunk operations) loses to per-lane
  // fragment loads + two bookkeeping MFMAs per k-step (strip_kernel.hpp, RA)
  // measured (graph replay, us; slab -> RA): 4096x4096 M=4 5.3 -> 6.2, M=8 9.6 -> 7.1, M=16 9.9 -> 8.5; 4096x11008 M=4 9.8 -> 9.3,
  // M=8 10.0 -> 9.7, M=16 25.7 -> 11.9; 11008x4096 M=4 13.6 -> 13.0, M=8 20.2 -> 14.6, M=16 26.4 (split-K fallback) -> 18.9
  const int ra_min = knob("QLLM_STRIP_RA_MIN", 5);
  // long K with 64-wide groups or 3 bits: the one-round slab variant (24 loads + 12 scale/zero pairs per lane) spills
  // 17-21 registers in a 16-wave block; the register-A variant (rounds of 8) does not
  const int ra_longk = knob("QLLM_STRIP_RA_LONGK", 1);
  const bool longk = ra_longk && (w[0].group_size <= 64 || bits == 3) && strip_spw(w[0].K, w[0].group_size, 16) > 8;
  const int ra_base = (M >= ra_min) ? 1 : 0;
  if (sm) {
    // strip-major: 16-column strips only (every wave-load is 256 contiguous bytes whatever the width; narrow strips balance best)
    const int strips = cols / 16;
    if (strips < 1) return false;
    // Round 6: batches 2..4 on the batch-1 kernel's four-row forms (128-wide groups, K <= 16384): the A rows that carry four copies of x
    // at batch 1 carry four batch rows -- the weight stream and the MFMAs of a batch-1 launch.  QLLM_STRIP1_MAX_M = 1 keeps them on strip_dma.
    if (M >= 2 && M <= 4 && M <= knob("QLLM_STRIP1_MAX_M", 4) && bits == 4 && w[0].group_size == 128 && w[0].K <= 16384 && knob("QLLM_STRIP1", 1)) {
      int nw1 = 0, maxs1 = 0;
      if (strip1_shape(w[0].K, strips, compute_units(), &nw1, &maxs1) && maxs1 <= 32) {
        plan->cpl = 1; plan->nw = nw1; plan->spw = maxs1; plan->ra = 0; plan->sm = 1;
        plan->one_nw = nw1;
        plan->one_maxs = maxs1;
        return true;
      }
    }
    const int slab_nw = ra_base ? 0 : strip_sm_nw(w[0].K, M, w[0].group_size, bits);  // 0: no slab form for this shape
    // register-A form at M = 5..16, 4 bits, every layer a multiple of 64 wide and enough of them: blocks of four adjacent strips
    // (a register-A block re-reads all of x from L2; 64 columns share it instead of 16)
    const int sm_ra_cpl4 = knob("QLLM_SM_RA_CPL4", 1);
    // (3 bits: two strips -- four need more than 256 registers)
    int cpl = (sm_ra_cpl4 && !g32 && slab_nw == 0 && M >= 5 && M <= 16 && m64 && cols / 64 >= compute_units() / 2) ? (bits == 3 ? 2 : 4) : 1;
    int nw = (M > 16 || cpl > 1) ? 8 : (slab_nw ? slab_nw : 16);
    // strip_dma.hpp (M = 5..32, K a multiple of 64): the activations go through LDS by DMA.  A block pulls the activations of its
    // k range through the CU's memory pipe once, a CU ingests ~55 GB/s here, so the launch costs about (rounds of blocks on the
    // CUs) x (a block's fixed life + the bytes it pulls): blocks of cpl strips share one activation stream -- pick the cpl that minimises that product
    // (profiles/r03_batch16.md).  One strip per block: 16 waves; several: 8 waves (registers).  The last block of a layer may be ragged.
    const int ra_xd = knob("QLLM_RA_XD", 1);
    const int dma_cpl = knob("QLLM_DMA_CPL", 0);
    // From batch 2: the lds-slab forms stage M x K / 8 chunks per block and run one-strip blocks in several rounds on wide launches
    // (4 bits g128 at M = 4: q/k/v 14.3 -> 10.5 us, gate/up 24.7 -> 14.1; 64-wide groups had only the register-A form there: gate/up
    // 25 -> 18; 3 bits: q/k/v 26.6 -> 13.5, gate/up 49 -> 24).  Batch 1 never: the one-round slab forms win (profiles/r03_batch16.md).
    const int dma_min_m = knob("QLLM_DMA_MIN_M", 2);
    const int dma_from = dma_min_m < 2 ? 2 : dma_min_m;
    bool small_enough = true;  // (the kernel's byte offsets are 32-bit)
    for (int i = 0; i < n; ++i) small_enough = small_enough && (double)w[i].K * w[i].N * bits / 8 < 2147483648.0;
    if (ra_xd && small_enough && M >= dma_from && M >= 2 && M <= 32 && w[0].K % 64 == 0) {
      const int cus = compute_units();
      // (round 5: three strips -- q/k/v's 768 strips are 256 blocks of three, every CU busy, instead of 192 of four; six 3-bit strips where
      //  the ring fits the registers: 64-wide groups with fp16 zero points, HQQ -- gate/up's 1376 strips in ONE round of 230 blocks
      //  instead of 344 blocks of four in two.  profiles/r05_batch16.md)
      static const int cands4[] = {1, 2, 3, 4, 6}, cands3[] = {1, 2, 3, 4, 6};
      const int *cands = bits == 4 ? cands4 : cands3;
      bool all_f16z = w[0].group_size == 64;
      for (int i = 0; i < n; ++i) all_f16z = all_f16z && w[i].layout == QLLM_LAYOUT_NATIVE_F16Z && w[i].qzeros;
      const int n_cands = (M > 16) ? 1 : (g32 ? 2 : (bits == 4 ? 5 : (all_f16z ? 5 : 4)));  // (two row tiles: one strip per block; 3 bits: four strips unless fp16 zeros at g64; 32-wide groups: two)
      const double x_bytes = (double)M * w[0].K * 2, strip_bytes = (double)w[0].K * bits * 2 + (double)(w[0].K / w[0].group_size) * 64;
      int best = 0;
      double best_cost = 0;
      // a candidate must fit the LDS: the waves' rings + (round 5) their scale / zero-point tables, which grow with K and the width
      auto dma_fits = [&](int c) {
        const int nw_c = (c == 1 && M <= 16) ? 16 : 8;
        const int spw_c = (strip_spw(w[0].K, w[0].group_size, nw_c) + 1) & ~1;
        return strip_lds_bytes(M, spw_c, nw_c, c, w[0].group_size, 2, 1) <= 156 * 1024;
      };
      for (int ci = 0; ci < n_cands; ++ci) {
        const int c = cands[ci];
        if (!dma_fits(c)) continue;
        int blocks = 0;
        for (int i = 0; i < n; ++i) blocks += (w[i].N / 16 + c - 1) / c;
        // (+ 96 KB per round: the ~1.8 us a block lives before and after its stream, at the CU's ingest rate)
        const double cost = (double)((blocks + cus - 1) / cus) * (96.0 * 1024 + x_bytes + c * strip_bytes);
        if (best == 0 || cost < best_cost * 0.97) { best = c; best_cost = cost; }  // (wider only for a clear gain)
        if (c == dma_cpl) { best = c; break; }                                    // (experiments: QLLM_DMA_CPL forces a width)
      }
      cpl = best ? best : 1;
      // (32-wide groups, one strip per block, short K, few rows: the register-A form's 8-k-step rounds beat the three-slot ring --
      //  4096 x 4096 at M = 4: 6.2 vs 7.7 us, M = 16: 7.8 both; tools/g32_bench.py, profiles/logs/r04r_g32_bench.log)
      const bool g32_ra = g32 && cpl == 1 && M <= 8 && w[0].K <= 4096;
      nw = (cpl == 1 && M <= 16) ? 16 : 8;
      const int spw = (strip_spw(w[0].K, w[0].group_size, nw) + 1) & ~1;  // (the ring's slots are pairs of k-steps; only 32-wide groups can give an odd chunk)
      if (!g32_ra && strip_lds_bytes(M, spw, nw, cpl, w[0].group_size, 2, 1) <= 156 * 1024) {
        plan->cpl = cpl;
        plan->nw = nw;
        plan->spw = spw;
        plan->ra = 2;
        plan->sm = 1;
        return true;
      }
      cpl = 1;  // (does not fit: the register-A form below)
      nw = (M > 16) ? 8 : 16;
    }
    for (int tries = 0; tries < 2; ++tries) {
      const int spw = strip_spw(w[0].K, w[0].group_size, nw);
      const int ra = (slab_nw == 0 || (longk && nw == 16)) ? 1 : 0;
      if (!ra && w[0].K / 32 < strip_maxs(nw, spw, 1, 0, 1)) return false;  // a round's window must fit into the strip
      if ((ra || strip_x_ok(M, spw, nw, 1, 1)) && strip_lds_bytes(M, spw, nw, cpl, w[0].group_size, ra, 1) <= 156 * 1024) {
        plan->cpl = cpl;
        plan->nw = nw;
        plan->spw = spw;
        plan->ra = ra;
        plan->sm = 1;
        // batch 1, 4 bits, 128-wide groups: the specialised kernel (round 5; profiles/r05_decode_bisect.md)
        int nw1 = 0, maxs1 = 0;
        // (round 6: 64-wide groups too -- HQQ's default; QLLM_STRIP1 = 2 keeps them on the general kernel)
        const int s1 = knob("QLLM_STRIP1", 1);
        // (round 6: 3-bit layers too -- batch 1, K <= 16384; QLLM_STRIP1_3BIT = 0 keeps them on the general kernel)
        const bool w4 = bits == 4 && (w[0].group_size == 128 || (w[0].group_size == 64 && s1 != 2 && w[0].K <= 24576));
        const bool w3 = bits == 3 && (w[0].group_size == 128 || w[0].group_size == 64) && w[0].K <= 16384 && knob("QLLM_STRIP1_3BIT", 1);
        if (M == 1 && (w4 || w3) && s1 && strip1_shape(w[0].K, strips, compute_units(), &nw1, &maxs1)) {
          plan->one_nw = nw1;
          plan->one_maxs = maxs1;
        }
        return true;
      }
      if (nw == 16 || M > 16 || cpl > 1) break;
      nw = 16;  // shorter per-wave chunks
    }
    return false;
  }
  const int force_cpl_env = knob("QLLM_STRIP_CPL", 0);
  int force_cpl = force_cpl_env;
  const int cus = compute_units();
  int first = (bits == 3 || M > 16) ? 1 : strip_cpl(cols, m64, m32 && M <= 4, cus);
  if (M > 16) force_cpl = 1;  // several row tiles per block: 16-column strips, 8-wave blocks, register-A
  if (force_cpl == 4 && m64 && bits == 4) first = 4;
  if (force_cpl == 2 && m32 && bits == 4) first = 2;
  if (force_cpl == 1) first = 1;
  // candidates: the measured-best strip width first; if its activation slab does not fit in LDS (many rows), narrower
  // strips with 16 waves (each wave stages a shorter K chunk)
  const int cand_cpl[2] = {first, 1};
  for (int ci = 0; ci < 2; ++ci) {
    const int cpl = cand_cpl[ci];
    if (ci >= 1 && cpl == cand_cpl[ci - 1]) continue;
    const int strips = cols / (16 * cpl);
    if (cpl == 1 && strips < strip_min_strips()) continue;
    // 64-column strips use 128 VGPRs -> 16 waves per CU: 8-wave blocks keep two strips co-resident per CU (one
    // round) instead of 16-wave blocks in two rounds (gate/up 15.6 -> 13.8 us, q/k/v 8.5 -> 8.3 us)
    int nw = cpl == 4 ? 8 : ((cpl == 2 || bits == 3) ? 16 : strip_nw(w[0].K, strips, cus));
    if (M > 16) nw = 8;
    for (int tries = 0; tries < 2; ++tries) {
      const int spw = strip_spw(w[0].K, w[0].group_size, nw);
      const int ra = (ra_base || (longk && cpl == 1 && nw == 16)) ? 1 : 0;
      if ((ra || strip_x_ok(M, spw, nw, cpl, 0)) && strip_lds_bytes(M, spw, nw, cpl, w[0].group_size, ra, 0) <= 156 * 1024) {
        plan->cpl = cpl;
        plan->nw = nw;
        plan->spw = spw;
        plan->ra = ra;
        plan->sm = 0;
        return true;
      }
      if (nw == 16 || M > 16 || cpl == 4) break;  // (64-column strips exist as 8-wave blocks only)
      nw = 16;  // shorter per-wave chunks
    }
  }
  return false;
}
// layouts that may share a grouped launch: reference row-stream (GPTQ / HQQ), AWQ, native
static int layout_family(const qllm_weight_t &w) { return is_native(w) ? 2 : (w.layout == QLLM_LAYOUT_AWQ_GEMM ? 1 : 0); }

static int run_strip1(const StripPlan &pl, const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, hipStream_t stream) {
  Strip1Params p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.T = w[0].K / 32;
  p.n_groups = w[0].K / w[0].group_size;
  p.group64 = w[0].group_size == 64;
  p.M = M;
  p.bits3 = w[0].bits == 3;
  p.add_zero_bias = w[0].add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  p.dbg = (g_timeline && g_timeline_next < g_timeline_slots) ? g_timeline + 24 * (g_timeline_next++) : nullptr;
  int max_strips = 0;
  for (int i = 0; i < n; ++i) {
    Strip1Problem &q = p.prob[i];
    q.qweight = (const uint32_t *)w[i].qweight;
    q.scales = (const half_t *)w[i].scales;
    q.qzeros = w[i].qzeros;
    q.bias = (const half_t *)w[i].bias;
    q.y = y[i];
    q.n_strips = w[i].N / 16;
    q.zero_kind = zero_kind_of(w[i]);
    max_strips = std::max(max_strips, q.n_strips);
  }
  return launch_strip1(p, pl.one_nw, pl.one_maxs, n, max_strips, stream);
}

static int run_strip(const StripPlan &pl, const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, hipStream_t stream) {
  if (pl.one_nw) return run_strip1(pl, w, y, n, x, M, act_dtype, stream);
  StripParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.n_prob = n;
  p.M = M;
  p.K = w[0].K;
  p.T = w[0].K / 32;
  p.cpl = pl.cpl;
  p.bits = w[0].bits;
  p.nw = pl.nw;
  p.spw = pl.spw;
  p.ra = pl.ra;
  p.sm = pl.sm;
  p.n_groups = (w[0].K + w[0].group_size - 1) / w[0].group_size;
  p.group_size = w[0].group_size;
  p.add_zero_bias = w[0].add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  // diagnostics: only the lds-slab 4-bit g128 native-layout form has a timeline instantiation
  p.dbg = (pl.sm && (!pl.ra || pl.ra == 2) && p.bits == 4 && p.group_size == 128 && g_timeline && g_timeline_next < g_timeline_slots)
              ? g_timeline + 24 * (g_timeline_next++) : nullptr;
  int block = 0;
  for (int i = 0; i < n; ++i) {
    StripProblem &q = p.prob[i];
    q.qweight = (const uint32_t *)w[i].qweight;
    q.scales = (const half_t *)w[i].scales;
    q.qzeros = w[i].qzeros;
    q.bias = (const half_t *)w[i].bias;
    q.y = y[i];
    q.N = w[i].N;
    q.n_strips = (w[i].N / 16 + pl.cpl - 1) / pl.cpl;  // (strip_dma.hpp: the last block may be ragged; every other form divides)
    q.block_begin = block;
    p.block_begin8[i] = block;
    q.zero_kind = zero_kind_of(w[i]);
    block += q.n_strips;
  }
  return launch_strip(p, block, stream);
}

static bool gemm_ok(const qllm_weight_t &w) {
  if (!fused_common_ok(w) || w.K % 64 != 0) return false;
  if (w.layout == QLLM_LAYOUT_AWQ_GEMM) return w.group_size % 4 == 0;
  if (w.g_idx) {
    const int groups = (w.K + w.group_size - 1) / w.group_size;
    return (size_t)groups * 128 * 4 + 65536 <= 160 * 1024;
  }
  return true;
}

static int run_skinny(const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, void *workspace,
                      size_t workspace_bytes, hipStream_t stream) {
  const int layout = w[0].layout;
  const int awq_w = skinny_awq_w(M);
  const int tn = skinny_tile_cols(layout, awq_w);
  int tiles_total = 0;
  for (int i = 0; i < n; ++i) tiles_total += (w[i].N + tn - 1) / tn;
  if (tiles_total > (int)(kCounterBytes / sizeof(int))) return set_error(QLLM_ERR_UNSUPPORTED, "too many column tiles (%d)", tiles_total);
  int S, spw;
  skinny_plan(w[0].K, M, tiles_total, skinny_target_waves(), &S, &spw);

  SkinnyParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.n_prob = n;
  p.M = M;
  p.K = w[0].K;
  p.T = w[0].K / 32;
  p.group_size = w[0].group_size;
  p.add_zero_bias = w[0].add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  size_t slab_off = kCounterBytes;
  int tile_off = 0, block = 0;
  for (int i = 0; i < n; ++i) {
    SkinnyProblem &q = p.prob[i];
    q.qweight = (const uint32_t *)w[i].qweight;
    q.scales = (const half_t *)w[i].scales;
    q.qzeros = w[i].qzeros;
    q.bias = (const half_t *)w[i].bias;
    q.y = y[i];
    q.N = w[i].N;
    q.n_tiles = (w[i].N + tn - 1) / tn;
    q.S = S;
    q.spw = spw;
    q.block_begin = block;
    q.zero_kind = zero_kind_of(w[i]);
    q.counters = (int *)workspace + tile_off;
    q.slabs = (float *)((char *)workspace + slab_off);
    tile_off += q.n_tiles;
    block += q.n_tiles * S;
    if (S > 1) slab_off += align_up((size_t)S * M * w[i].N * sizeof(float), 256);
  }
  if (S > 1) {
    if (!workspace) return set_error(QLLM_ERR_WORKSPACE, "split-K needs a workspace (call qllm_workspace_bytes)");
    if (workspace_bytes < slab_off) return set_error(QLLM_ERR_WORKSPACE, "workspace too small: need %zu bytes, have %zu", slab_off, workspace_bytes);
    if ((uintptr_t)workspace % 256 != 0) return set_error(QLLM_ERR_INVALID, "workspace must be 256-byte aligned");
  }
  return launch_skinny(p, layout, awq_w, block, stream);
}

static int check_io(const void *x, const void *y, int M, int act_dtype) {
  if (!x || !y) return set_error(QLLM_ERR_INVALID, "x / y must not be NULL");
  if (M <= 0) return set_error(QLLM_ERR_INVALID, "M must be >= 1 (got %d)", M);
  if (act_dtype != QLLM_F16 && act_dtype != QLLM_BF16) return set_error(QLLM_ERR_INVALID, "act_dtype must be f16 or bf16");
  if ((uintptr_t)x % 16 != 0) return set_error(QLLM_ERR_INVALID, "x must be 16-byte aligned");
  return QLLM_OK;
}

// ---- sub-decisions that depend on the caller's workspace: ONE function each, used by the launch path and by qllm_plan_describe -------
// `ws_bytes`: bytes of a usable (non-NULL, 256-byte aligned) workspace, 0 without one; qllm_plan_describe passes SIZE_MAX / 0.
static size_t usable_ws(const void *workspace, size_t workspace_bytes) {
  return (workspace && (uintptr_t)workspace % 256 == 0) ? workspace_bytes : 0;
}
// gemm3 over K-split blocks when the 256x128 tiling leaves CUs idle and the workspace can hold the partial tiles: the split, or 1
static int gemm3_split_for(int M, int N, int K, size_t ws_bytes) {
  const int S = gemm3_split_k(M, N, K);
  if (S <= 1) return 1;
  const int tiles = ((M + 255) / 256) * (N / 128);
  if (ws_bytes < kCounterBytes + gemm2_slab_bytes(M, N, S) || tiles > (int)(kCounterBytes / sizeof(int))) return 1;
  return S;
}
static int gemm2_split_for(int M, int N, int K, size_t ws_bytes) {
  const int S = gemm2_split_k(M, N, K);
  if (S <= 1) return 1;
  const int tiles = ((M + 255) / 256) * (N / 128);
  if (ws_bytes < kCounterBytes + gemm2_slab_bytes(M, N, S) || tiles > (int)(kCounterBytes / sizeof(int))) return 1;
  return S;
}
static int panel_split_for(int M, int n_panels, int K, size_t ws_bytes) {
  const int S = panel_split_k(M, n_panels, K);
  if (S <= 1 || ws_bytes < kCounterBytes + panel_slab_bytes(M, n_panels, S) || n_panels > (int)(kCounterBytes / sizeof(int))) return 1;
  return S;
}
// gemm3's K-split of the ragged last round (gemm3.hip, round 6): the factor (1: none) and the first split tile
static size_t gemm3_tail_slab_bytes(int tail_tiles, int TS) { return TS > 1 ? (size_t)tail_tiles * TS * 256 * 128 * sizeof(float) : 0; }
static int gemm3_tail_for(int M, int N, int K, size_t ws_bytes, int *tail_from) {
  const int tiles = ((M + 255) / 256) * (N / 128);
  const int TS = gemm3_tail_split(M, N, K, tail_from);
  if (TS <= 1 || ws_bytes < kCounterBytes + gemm3_tail_slab_bytes(tiles - *tail_from, TS) || tiles - *tail_from > (int)(kCounterBytes / sizeof(int))) {
    *tail_from = tiles;
    return 1;
  }
  return TS;
}
static void set_split(GemmParams &p, int S, void *workspace, int tail_from = 0, int tail_split = 0) {
  p.split_k = S;
  p.tail_from = tail_split > 1 ? tail_from : 0;
  p.tail_split = tail_split > 1 ? tail_split : 0;
  const bool slabs = S > 1 || tail_split > 1;
  p.counters = slabs ? (int *)workspace : nullptr;
  p.slabs = slabs ? (float *)((char *)workspace + kCounterBytes) : nullptr;
}

static void fill_gemm_params(GemmParams &p, const qllm_weight_t *w, const void *x, void *y, int M, int act_dtype) {
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.qweight = (const uint32_t *)w->qweight;
  p.scales = (const half_t *)w->scales;
  p.qzeros = w->qzeros;
  p.g_idx = w->g_idx;
  p.bias = (const half_t *)w->bias;
  p.y = y;
  p.M = M;
  p.K = w->K;
  p.N = w->N;
  p.group_size = w->group_size;
  p.gs_shift = ((w->group_size & (w->group_size - 1)) == 0) ? __builtin_ctz((unsigned)w->group_size) : -1;
  p.add_zero_bias = w->add_zero_bias;
  p.zero_kind = zero_kind_of(*w);
  p.act_bf16 = (act_dtype == QLLM_BF16);
  p.n_groups = (w->K + w->group_size - 1) / w->group_size;
  p.sm = is_native(*w) ? 1 : 0;
  p.split_k = 1;
#ifdef QLLM_LAB
  p.dbg = g_timeline;  // (lab: qllm_debug_timeline(buf, n) before a prefill call hands gemm4 its per-block stamp buffer)
#endif
}

// bytes of the fp16 copy of bf16 activations gemm3 reads (0 for fp16 activations), and where it sits in the workspace
static size_t bf16_copy_bytes(int M, int K, int act_bf16) { return act_bf16 ? align_up((size_t)M * K * 2, 256) : 0; }

// the 256-row-tile GEMMs on a row-stream (or strip-major) 4-bit layer / AWQ layer: which of the two kernels, split how
struct TileChoice {
  int kernel;     // 3: the wave-specialised 256x128 kernel (gemm3.hip); 2: gemm2
  int split_k;
  size_t copy_off;  // kernel 3 with bf16 activations: where the fp16 copy of x sits in the workspace
  int tail_from, tail_split;  // kernel 3, more tiles than CUs: K-split of the ragged last round (tail_split > 1)
  int native_bf16;            // kernel 3, bf16 activations: no fp16 copy -- bf16 W and bf16 MFMA (gemm3.hip, round 6)
};
// large M: the wave-specialised 256x128 kernel (no split-K needed: every CU has at least one tile), also where it can split K; gemm2
// below.  bf16 activations: gemm3's activation tiles travel by LDS-DMA, which cannot convert, so x is converted to fp16 once into the
// workspace (the reference's own shim does the same cast, quant_linear_awq.py:29-36) and the epilogue rounds the fp16 result to bf16 --
// which needs room for the copy, else gemm2 (which converts in registers)
static TileChoice choose_tile(const GemmParams &p, int layout, size_t ws_bytes) {
  GemmParams q = p;
  q.act_bf16 = 0;
  if (gemm3_ok(q, layout)) {
    const int S3 = gemm3_split_for(p.M, p.N, p.K, ws_bytes);
    if (gemm2_split_k(p.M, p.N, p.K) == 1 || S3 > 1) {
      const int native = p.act_bf16 && gemm3_bf16_native(layout);
      const size_t copy = native ? 0 : bf16_copy_bytes(p.M, p.K, p.act_bf16);
      const int tiles = ((p.M + 255) / 256) * (p.N / 128);
      int tail_from = tiles;
      // (the slabs of the tail split and the fp16 copy of bf16 activations share the workspace: the copy comes first)
      const int TS = S3 > 1 ? 1 : gemm3_tail_for(p.M, p.N, p.K, ws_bytes > copy ? ws_bytes - copy : 0, &tail_from);
      const size_t used = kCounterBytes + (S3 > 1 ? align_up(gemm2_slab_bytes(p.M, p.N, S3), 256) : align_up(gemm3_tail_slab_bytes(tiles - tail_from, TS), 256));
      if (!copy || ws_bytes >= used + copy) return TileChoice{3, S3, used, tail_from, TS, native};
    }
  }
  return TileChoice{2, gemm2_split_for(p.M, p.N, p.K, ws_bytes), 0, 0, 0, 0};
}

static int run_tile_gemm(GemmParams &p, int layout, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  const TileChoice c = choose_tile(p, layout, usable_ws(workspace, workspace_bytes));
  set_split(p, c.split_k, workspace, c.tail_from, c.tail_split);
  if (c.kernel == 2) return launch_gemm2(p, layout, stream);
  if (c.native_bf16) {
    p.native_bf16 = 1;
    p.act_bf16 = 0;  // (gemm3's own flag means "x was converted": not here)
    return launch_gemm3(p, layout, stream);
  }
  if (p.act_bf16) {
    void *xh = (char *)workspace + c.copy_off;
    if (int rc = launch_bf16_to_f16(p.x, xh, (size_t)p.M * p.K, stream)) return rc;
    p.x = xh;
    p.act_bf16 = 0;
    p.out_bf16 = 1;
  }
  return launch_gemm3(p, layout, stream);
}

// prefill-sized calls (M > 64) on native-layout layers: the same tile GEMMs, their staging waves reading the strip-major words
// every layer of the launch: native strip-major, 4 bits, no act-order, whole 64-column panels; the group: one K / group size
static bool panel_layers_ok(const qllm_weight_t *w, int n, int M) {
  if (!knob("QLLM_PANEL", 1) || n < 1 || n > kMaxProblems) return false;
  for (int i = 0; i < n; ++i) {
    if ((w[i].bits != 4 && w[i].bits != 3) || w[i].bits != w[0].bits || !is_native(w[i]) || w[i].g_idx || w[i].K != w[0].K || w[i].group_size != w[0].group_size) return false;
    if (!panel_shape_ok(M, w[i].K, w[i].N, w[i].group_size, w[i].bits)) return false;
    // (one launch decodes every layer's zero points the same way: fp16 zero points and packed / symmetric ones do not mix -- such a
    //  group goes to the strips, which decide per layer, or runs layer by layer)
    if ((zero_kind_of(w[i]) == ZK_F16) != (zero_kind_of(w[0]) == ZK_F16)) return false;
    if ((uintptr_t)w[i].qweight % 16 || (uintptr_t)w[i].scales % 16 || (w[i].qzeros && (uintptr_t)w[i].qzeros % 8)) return false;
  }
  return true;
}
// 65..128 rows (eight row tiles, one K half per block): only where it pays whatever the activation type -- us per linear, fp16 / bf16,
// against gemm2's 28 / 40 / 41 (+1-2 for bf16): 4096 x 4096 19 / 25; 4096 x 11008 36-38 / 50; 11008 x 4096 35 / 44 -> layers of up to
// 2^25 weights; groups (three launches become one: q/k/v 41 against 84) of up to 16384 columns.  profiles/r04_mid_m.md
static bool panel_serves(const qllm_weight_t *w, const GemmParams &p) {
  if (p.M > 64 && (double)p.K * p.N > 33554432.0) return false;
  return panel_rows_ok(p.M, p.K, p.N) && panel_layers_ok(w, 1, p.M);
}
// grouped launches (q/k/v, gate/up): from 17 rows -- the strips keep the smaller batches (BASELINE configs[3] is tuned there)
static bool panel_group_serves(const qllm_weight_t *w, int n, int M) {
  if (n <= 1 || M < knob("QLLM_PANEL_GROUP_MIN_M", 17) || !panel_layers_ok(w, n, M)) return false;
  if (M > 64) {
    int cols = 0;
    for (int i = 0; i < n; ++i) cols += w[i].N;
    if (cols > 16384) return false;
  }
  return true;
}
static int panels_of(const qllm_weight_t *w, int n) {
  int t = 0;
  for (int i = 0; i < n; ++i) t += w[i].N / 64;
  return t;
}
static int run_panel(const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, void *workspace, size_t workspace_bytes,
                     hipStream_t stream) {
  PanelParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.M = M;
  p.K = w[0].K;
  p.group_size = w[0].group_size;
  p.bits = w[0].bits;
  p.n_groups = (w[0].K + w[0].group_size - 1) / w[0].group_size;
  p.add_zero_bias = w[0].add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  p.n_prob = n;
  p.abl = knob("QLLM_PANEL_ABL", 0);  // (lab builds: timing-only ablations)
  int begin = 0;
  for (int i = 0; i < n; ++i) {
    PanelProblem &q = p.prob[i];
    q.qweight = (const uint32_t *)w[i].qweight;
    q.scales = (const half_t *)w[i].scales;
    q.qzeros = w[i].qzeros;
    q.bias = (const half_t *)w[i].bias;
    q.y = y[i];
    q.N = w[i].N;
    q.zero_kind = zero_kind_of(w[i]);
    q.panel_begin = begin;
    begin += w[i].N / 64;
  }
  p.n_panels = begin;
  // split-K when the panels alone leave CUs idle and the caller's workspace can hold the partial panels (else: no split)
  p.split_k = panel_split_for(M, begin, p.K, usable_ws(workspace, workspace_bytes));
  if (p.split_k > 1) {
    p.counters = (int *)workspace;
    p.slabs = (float *)((char *)workspace + kCounterBytes);
  }
  return launch_panel(p, stream);
}

static bool native_prefill_ok(const qllm_weight_t *w, GemmParams &p) {
  if ((uintptr_t)w->qweight % 16 || (uintptr_t)w->scales % 16 || (w->qzeros && (uintptr_t)w->qzeros % 8)) return false;
  if (panel_serves(w, p)) return true;
  if (w->bits == 3) return gemm3_ok(p, kGemm3Rows3Bit);
  return w->bits == 4 && gemm2_ok(p, QLLM_LAYOUT_GPTQ);
}

// ---- ONE decision per forward call (round 6; round-5 verdict, weak #8: qllm_plan_describe used to restate this order by hand) -------
// decide_single / decide_group are the ONLY place a kernel family is chosen: qllm_linear_forward(_grouped) executes the Decision,
// qllm_plan_describe prints it.  The workspace-dependent sub-choices (split-K, which 256-row-tile kernel) are choose_tile /
// *_split_for above, again shared by both.
enum Route { ROUTE_NONE = 0, ROUTE_STRIP, ROUTE_PANEL, ROUTE_ROWS3, ROUTE_TILE, ROUTE_GEMM, ROUTE_SKINNY, ROUTE_BITGEMV, ROUTE_TILE_GROUP };
struct Decision {
  Route route;
  StripPlan strip;  // ROUTE_STRIP
  int layout;       // ROUTE_TILE / ROUTE_GEMM / ROUTE_ROWS3: the tile kernels' layout selector
  int rc;           // ROUTE_NONE: the status the forward call returns (text in qllm_last_error())
};
static Decision routed(Route r, int layout = 0) {
  Decision d;
  memset(&d, 0, sizeof(d));
  d.route = r;
  d.layout = layout;
  return d;
}
static Decision refused(int rc) {
  Decision d = routed(ROUTE_NONE);
  d.rc = rc;
  return d;
}

// one validated layer, M rows of `act_dtype` activations
static Decision decide_single(const qllm_weight_t *w, int M, int act_dtype) {
  Decision d = routed(ROUTE_STRIP);
  GemmParams p;
  fill_gemm_params(p, w, nullptr, nullptr, M, act_dtype);
  if ((w->bits == 3 || is_native(*w)) && strip_plan(w, 1, M, &d.strip)) return d;
  if (is_native(*w)) {
    // prefill-sized calls (M > 64, and what the strips leave alone) on native-layout layers: the panel kernel, or the tile GEMMs with
    // their staging waves reading the strip-major words
    if (!native_prefill_ok(w, p))
      return refused(set_error(QLLM_ERR_UNSUPPORTED, "native-layout layer: no fused kernel for M=%d K=%d N=%d g=%d bits=%d (decode sizes, and M > 64 with "
                               "K %% 64 == 0, N %% 128 == 0 and a power-of-two group size, are served)", M, w->K, w->N, w->group_size, w->bits));
    if (panel_serves(w, p)) return routed(ROUTE_PANEL);
    if (w->bits == 3) return routed(ROUTE_ROWS3, kGemm3Rows3Bit);
    return routed(ROUTE_TILE, QLLM_LAYOUT_GPTQ);
  }
  if (skinny_ok(*w, M)) {
    if (strip_plan(w, 1, M, &d.strip)) return d;
    // 33..64 rows the strips leave alone (wide shapes): the 256-row-tile GEMM beats split-K here
    if (M > 32 && gemm_ok(*w) && gemm2_ok(p, w->layout)) return routed(ROUTE_TILE, w->layout);
    return routed(ROUTE_SKINNY);
  }
  if (w->bits == 3 && M > 64 && w->layout != QLLM_LAYOUT_AWQ_GEMM && !w->g_idx && (uintptr_t)w->qweight % 16 == 0 &&
      (uintptr_t)w->scales % 16 == 0 && (!w->qzeros || (uintptr_t)w->qzeros % 8 == 0) && gemm3_ok(p, kGemm3Rows3Bit))
    return routed(ROUTE_ROWS3, kGemm3Rows3Bit);  // 3-bit row-stream layers at prefill sizes: gemm3 with 3-bit staging waves (LAYOUT 2)
  if (gemm_ok(*w)) return routed(gemm2_ok(p, w->layout) ? ROUTE_TILE : ROUTE_GEMM, w->layout);
  // 2 / 5 / 6 / 7 / 8 bits (and 3 / 4-bit layers nothing above takes) at decode sizes: the bit-stream matvec (bitgemv.hip, round 6)
  if (knob("QLLM_BITGEMV", 1) && bitgemv_ok(*w, M)) return routed(ROUTE_BITGEMV);
  return refused(set_error(QLLM_ERR_UNSUPPORTED, "no fused kernel for bits=%d K=%d N=%d g=%d layout=%d act_order=%d; use qllm_dequant + GEMM",
                           w->bits, w->K, w->N, w->group_size, w->layout, w->g_idx != nullptr));
}

// n >= 2 validated layers sharing x; INVALID when they cannot share a launch at all, UNSUPPORTED when no grouped kernel takes them
// prefill-sized groups (round 6): q/k/v, gate/up as ONE launch of the 256x128 kernel -- its grid carries the tiles of up to 4 layers, so
// the rounds of CUs are counted over the group (Llama-2-7B gate/up: 1376 tiles = 5.4 rounds, with the last one K-split, instead of
// 2 x 2.7 -> 2 x 3) and the group costs one launch boundary.  4-bit row-stream / strip-major layers of one storage kind, whole tiles,
// at least one tile per CU (no split-K inside a group); bf16 only where the kernel takes it natively.
static int group_tiles(const qllm_weight_t *w, int n, int M) {
  int t = 0;
  for (int i = 0; i < n; ++i) t += ((M + 255) / 256) * (w[i].N / 128);
  return t;
}
static bool tile_group_serves(const qllm_weight_t *w, int n, int M, int act_dtype) {
  if (!knob("QLLM_GEMM3_GROUP", 1) || !knob("QLLM_GEMM3", 1) || n < 2 || n > kGemm3MaxProb || M < 384) return false;
  const int gs = w[0].group_size;
  if (w[0].bits != 4 || w[0].K % 64 != 0 || gs < 32 || (gs & (gs - 1)) != 0) return false;
  if (act_dtype == QLLM_BF16 && !gemm3_bf16_native(QLLM_LAYOUT_GPTQ)) return false;
  for (int i = 0; i < n; ++i) {
    if (w[i].layout == QLLM_LAYOUT_AWQ_GEMM || w[i].g_idx || w[i].N % 128 != 0 || is_native(w[i]) != is_native(w[0])) return false;
    if ((uintptr_t)w[i].qweight % 16 || (uintptr_t)w[i].scales % 16 || (w[i].qzeros && (uintptr_t)w[i].qzeros % 8)) return false;
    if ((size_t)M * w[i].K * 2 >= 0x7fffffffull || (size_t)w[i].K * w[i].N / 2 >= 0x7fffffffull) return false;
  }
  return group_tiles(w, n, M) >= compute_units();
}

static Decision decide_group(const qllm_weight_t *w, int n, int M, int act_dtype) {
  for (int i = 0; i < n; ++i) {
    if (w[i].K != w[0].K || w[i].group_size != w[0].group_size || w[i].bits != w[0].bits || layout_family(w[0]) != layout_family(w[i]) ||
        w[i].add_zero_bias != w[0].add_zero_bias)
      return refused(set_error(QLLM_ERR_INVALID, "grouped weights must agree on K, group_size, bits, layout family and add_zero_bias"));
  }
  if (tile_group_serves(w, n, M, act_dtype)) return routed(ROUTE_TILE_GROUP, QLLM_LAYOUT_GPTQ);  // (prefill sizes: M >= 384)
  for (int i = 0; i < n; ++i) {
    if (w[i].bits == 3 || is_native(w[i])) continue;  // decided as a group by strip_plan below
    if (!skinny_ok(w[i], M))
      return refused(set_error(QLLM_ERR_UNSUPPORTED, "grouped forward needs the decode kernel (4-bit, M<=%d, K%%32==0, no act-order)", skinny_max_m()));
  }
  Decision d = routed(ROUTE_STRIP);
  if (strip_plan(w, n, M, &d.strip)) return d;
  if (panel_group_serves(w, n, M)) return routed(ROUTE_PANEL);
  if (is_native(w[0]))
    return refused(set_error(QLLM_ERR_UNSUPPORTED, "grouped forward: native-layout layers are served for M <= 32 (4 bits: <= 128) with group size 64 / 128 (4 bits: also 32), and 4-bit groups of at least one 256x128 tile per CU from 384 rows (M=%d g=%d)", M, w[0].group_size));
  if (w[0].bits != 4) return refused(set_error(QLLM_ERR_UNSUPPORTED, "grouped forward: no fused kernel for %d-bit weights in this shape", w[0].bits));
  return routed(ROUTE_SKINNY);
}

static int bitgemv_split_for(int M, int K, int N, size_t ws_bytes) {
  const int S = bitgemv_split(M, K, N);
  if (S <= 1 || ws_bytes < kCounterBytes + (size_t)S * M * N * sizeof(float) || (N + 15) / 16 > (int)(kCounterBytes / sizeof(int))) return 1;
  return S;
}
static int run_bitgemv(const qllm_weight_t *w, void *y, const void *x, int M, int act_dtype, void *workspace, size_t workspace_bytes, hipStream_t stream) {
  BitGemvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.qweight = (const uint32_t *)w->qweight;
  p.scales = (const half_t *)w->scales;
  p.qzeros = w->qzeros;
  p.bias = (const half_t *)w->bias;
  p.y = y;
  p.M = M;
  p.K = w->K;
  p.N = w->N;
  p.group_size = w->group_size;
  p.zero_kind = zero_kind_of(*w);
  p.add_zero_bias = w->add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  p.ksplit = bitgemv_split_for(M, w->K, w->N, usable_ws(workspace, workspace_bytes));
  if (p.ksplit > 1) {
    p.counters = (int *)workspace;
    p.slabs = (float *)((char *)workspace + kCounterBytes);
  }
  return launch_bitgemv(p, w->bits, stream);
}

static int tile_group_tail_for(const qllm_weight_t *w, int n, int M, size_t ws_bytes, int *tail_from) {
  const int tiles = group_tiles(w, n, M);
  const int TS = gemm3_tail_split_tiles(tiles, w[0].K, tail_from);
  if (TS <= 1 || ws_bytes < kCounterBytes + gemm3_tail_slab_bytes(tiles - *tail_from, TS) || tiles - *tail_from > (int)(kCounterBytes / sizeof(int))) {
    *tail_from = tiles;
    return 1;
  }
  return TS;
}
static int run_tile_group(const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, void *workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  GemmParams p;
  fill_gemm_params(p, &w[0], x, y[0], M, act_dtype);
  p.n_prob = n;
  int begin = 0;
  for (int i = 0; i < n; ++i) {
    GemmProb &q = p.prob[i];
    q.qweight = (const uint32_t *)w[i].qweight;
    q.scales = (const half_t *)w[i].scales;
    q.qzeros = w[i].qzeros;
    q.bias = (const half_t *)w[i].bias;
    q.y = y[i];
    q.N = w[i].N;
    q.zero_kind = zero_kind_of(w[i]);
    q.tile_begin = begin;
    begin += ((M + 255) / 256) * (w[i].N / 128);
  }
  p.total_tiles = begin;
  int tail_from = begin;
  const int TS = tile_group_tail_for(w, n, M, usable_ws(workspace, workspace_bytes), &tail_from);
  set_split(p, 1, workspace, tail_from, TS);
  if (act_dtype == QLLM_BF16) {
    p.native_bf16 = 1;
    p.act_bf16 = 0;
  }
  return launch_gemm3(p, QLLM_LAYOUT_GPTQ, stream);
}

// the Decision, executed
static int execute(const Decision &d, const qllm_weight_t *w, void *const *y, int n, const void *x, int M, int act_dtype, void *workspace,
                   size_t workspace_bytes, hipStream_t stream) {
  switch (d.route) {
    case ROUTE_STRIP: return run_strip(d.strip, w, y, n, x, M, act_dtype, stream);
    case ROUTE_PANEL: return run_panel(w, y, n, x, M, act_dtype, workspace, workspace_bytes, stream);
    case ROUTE_SKINNY: return run_skinny(w, y, n, x, M, act_dtype, workspace, workspace_bytes, stream);
    case ROUTE_TILE_GROUP: return run_tile_group(w, y, n, x, M, act_dtype, workspace, workspace_bytes, stream);
    case ROUTE_BITGEMV: return run_bitgemv(w, y[0], x, M, act_dtype, workspace, workspace_bytes, stream);
    case ROUTE_ROWS3: case ROUTE_TILE: case ROUTE_GEMM: {
      GemmParams p;
      fill_gemm_params(p, w, x, y[0], M, act_dtype);
      if (d.route == ROUTE_GEMM) return launch_gemm(p, d.layout, stream);
      if (d.route == ROUTE_TILE) return run_tile_gemm(p, d.layout, workspace, workspace_bytes, stream);
      p.g_idx = nullptr;
      set_split(p, gemm3_split_for(p.M, p.N, p.K, usable_ws(workspace, workspace_bytes)), workspace);
      return launch_gemm3(p, kGemm3Rows3Bit, stream);
    }
    default: return d.rc ? d.rc : set_error(QLLM_ERR_INVALID, "internal: empty decision");
  }
}

// the Decision, as text (qllm_plan_describe): `ws_bytes` = SIZE_MAX / 0 for "the caller has / has no workspace"
static void describe(const Decision &d, const qllm_weight_t *w, int n, int M, size_t ws_bytes, char *buf, size_t buflen) {
  const char *sm = is_native(w[0]) ? " layout=strip-major" : "";
  switch (d.route) {
    case ROUTE_STRIP: {
      const StripPlan &pl = d.strip;
      if (pl.one_nw)
        snprintf(buf, buflen, "strip1 nw=%d round=%d%s%s%s grid=strips x %d layout=strip-major", pl.one_nw, pl.one_maxs,
                 pl.one_nw * pl.one_maxs == w[0].K / 32 ? " exact" : "", w[0].group_size == 64 ? (w[0].bits == 3 ? " g64 bits=3" : " g64") : (w[0].bits == 3 ? " bits=3" : ""),
                 M > 1 ? " rows=4" : "", n);
      else
        snprintf(buf, buflen, "strip nw=%d cpl=%d spw=%d form=%s row_tiles=%d%s", pl.nw, pl.cpl, pl.spw,
                 pl.ra == 2 ? "dma-A" : (pl.ra ? "register-A" : "lds-slab"), M > 32 ? 4 : (M > 16 ? 2 : 1), pl.sm ? " layout=strip-major" : "");
      return;
    }
    case ROUTE_PANEL: {
      char layers[24] = "";
      if (n > 1) snprintf(layers, sizeof(layers), " layers=%d", n);
      snprintf(buf, buflen, "panel cols=64 row_tiles=%d k_halves=%d split_k=%d%s%s layout=strip-major", panel_mt(M), panel_kh(M),
               panel_split_for(M, panels_of(w, n), w[0].K, ws_bytes), layers, w[0].bits == 3 ? " bits=3" : "");
      return;
    }
    case ROUTE_ROWS3: {
      const int S = gemm3_split_for(M, w[0].N, w[0].K, ws_bytes);
      if (S > 1) snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 bits=3 split_k=%d%s", S, sm);
      else snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 bits=3%s", sm);
      return;
    }
    case ROUTE_TILE: {
      GemmParams p;
      fill_gemm_params(p, &w[0], nullptr, nullptr, M, QLLM_F16);
      const TileChoice c = choose_tile(p, d.layout, ws_bytes);
      if (c.kernel == 3 && c.tail_split > 1) snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 tail_split=%d%s", c.tail_split, sm);
      else if (c.kernel == 3 && c.split_k > 1) snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 split_k=%d%s", c.split_k, sm);
      else if (c.kernel == 3) snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4%s", sm);
      else snprintf(buf, buflen, "gemm2 tile=256x%d split_k=%d%s", gemm2_tile_n(M, w[0].N, c.split_k), c.split_k, sm);
      return;
    }
    case ROUTE_TILE_GROUP: {
      int tail_from = 0;
      const int TS = tile_group_tail_for(w, n, M, ws_bytes, &tail_from);
      if (TS > 1) snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 layers=%d tail_split=%d%s", n, TS, sm);
      else snprintf(buf, buflen, "gemm3 tile=256x128 matrix-waves=8 staging-waves=4 layers=%d%s", n, sm);
      return;
    }
    case ROUTE_GEMM: snprintf(buf, buflen, "gemm tile=128x128%s", w[0].g_idx ? " act-order-gather" : ""); return;
    case ROUTE_SKINNY: {
      const int awq_w = skinny_awq_w(M), tn = skinny_tile_cols(w[0].layout, awq_w);
      int tiles_total = 0, S, spw;
      for (int i = 0; i < n; ++i) tiles_total += (w[i].N + tn - 1) / tn;
      skinny_plan(w[0].K, M, tiles_total, skinny_target_waves(), &S, &spw);
      snprintf(buf, buflen, "skinny tile_cols=%d split_k=%d spw=%d", tn, S, spw);
      return;
    }
    case ROUTE_BITGEMV:
      snprintf(buf, buflen, "bitgemv bits=%d cols=%d waves=8 split_k=%d", w[0].bits, bitgemv_cols(), bitgemv_split_for(M, w[0].K, w[0].N, ws_bytes));
      return;
    default: snprintf(buf, buflen, "unsupported (%s)", g_err[0] ? g_err : "dequant + GEMM");
  }
}

}  // namespace qllm

using namespace qllm;

extern "C" {

int qllm_abi_version(void) { return QLLM_ABI_VERSION; }

int qllm_is_lab_build(void) {
#ifdef QLLM_LAB
  return 1;
#else
  return 0;
#endif
}

const char *qllm_last_error(void) { return g_err; }

int qllm_set_knob(const char *name, int32_t value) {
  clear_error();
  if (!name) return set_error(QLLM_ERR_INVALID, "qllm_set_knob: name is NULL");
  for (Settable &k : kSettable) {
    if (strcmp(k.name, name) != 0) continue;
    if (value < k.lo || value > k.hi) return set_error(QLLM_ERR_INVALID, "qllm_set_knob: %s takes %d..%d (got %d)", name, k.lo, k.hi, value);
    k.value = value;
    if (!k.set) {
      k.set = 1;
      __atomic_fetch_add(&g_knob_overrides, 1, __ATOMIC_RELAXED);
    }
    return QLLM_OK;
  }
  return set_error(QLLM_ERR_INVALID, "qllm_set_knob: %s is not a settable planner threshold (see include/qllm_mi355x.h)", name);
}

int qllm_get_knob(const char *name, int32_t *value, int32_t *is_set) {
  clear_error();
  if (!name || !value) return set_error(QLLM_ERR_INVALID, "qllm_get_knob: name / value is NULL");
  for (const Settable &k : kSettable)
    if (strcmp(k.name, name) == 0) {
      *value = k.set ? k.value : 0;
      if (is_set) *is_set = k.set;
      return QLLM_OK;
    }
  return set_error(QLLM_ERR_INVALID, "qllm_get_knob: %s is not a settable planner threshold", name);
}

void qllm_reset_knobs(void) {
  for (Settable &k : kSettable) k.set = 0;
  __atomic_store_n(&g_knob_overrides, 0, __ATOMIC_RELAXED);
}

int qllm_device_info(int device, qllm_device_info_t *out) {
  clear_error();
  if (!out) return set_error(QLLM_ERR_INVALID, "out is NULL");
  memset(out, 0, sizeof(*out));
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    (void)hipGetLastError();
    return set_error(QLLM_ERR_DEVICE, "no HIP device %d (count=%d)", device, count);
  }
  hipDeviceProp_t prop;
  QLLM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  size_t n = 0;
  while (prop.gcnArchName[n] && prop.gcnArchName[n] != ':' && n + 1 < sizeof(out->arch)) {
    out->arch[n] = prop.gcnArchName[n];
    ++n;
  }
  out->compute_units = prop.multiProcessorCount;
  out->wavefront_size = prop.warpSize;
  out->lds_bytes_per_cu = (int32_t)prop.maxSharedMemoryPerMultiProcessor;
  out->clock_khz = prop.clockRate;
  out->hbm_bytes = (int64_t)prop.totalGlobalMem;
  if (strncmp(out->arch, "gfx950", 6) != 0) return set_error(QLLM_ERR_DEVICE, "device %d is %s, this library is gfx950-only", device, out->arch);
  return QLLM_OK;
}

size_t qllm_workspace_bytes(const qllm_weight_t *w, int32_t M) { return qllm_workspace_bytes_act(w, M, QLLM_BF16); }

size_t qllm_workspace_bytes_act(const qllm_weight_t *w, int32_t M, int32_t act_dtype) {
  if (!w || M <= 0) return kCounterBytes;
  size_t tiles = 0;
  if (M > 32) {  // the 256-row-tile GEMMs (every M > 64, and 33..64 rows of the shapes the strips leave alone): fp32 partial
    GemmParams p;  // tiles of the split-K forms + the fp16 copy of bf16 activations where the wave-specialised kernel would
    fill_gemm_params(p, w, nullptr, nullptr, M, QLLM_F16);  // serve a bf16 call
    p.g_idx = nullptr;
    const bool g3 = gemm3_ok(p, w->bits == 3 ? kGemm3Rows3Bit : (w->layout == QLLM_LAYOUT_AWQ_GEMM ? QLLM_LAYOUT_AWQ_GEMM : QLLM_LAYOUT_GPTQ));
    int tail_from = 0;
    const int TS = g3 && w->bits != 3 ? gemm3_tail_split(M, w->N, w->K, &tail_from) : 1;  // (the ragged last round's K-split, gemm3.hip)
    const size_t tail = align_up(gemm3_tail_slab_bytes(((M + 255) / 256) * (w->N / 128) - tail_from, TS), 256);
    tiles = std::max(align_up(gemm2_slab_bytes(M, w->N, gemm2_split_k(M, w->N, w->K)), 256), tail) + (g3 ? bf16_copy_bytes(M, w->K, act_dtype == QLLM_BF16) : 0);
    if (M > 64 && M > 128) return kCounterBytes + tiles;
  }
  // the panel kernel's partial panels (single native 4-bit layers, 9..128 rows)
  if (M >= 9 && M <= 128 && w->N % 64 == 0) tiles = std::max(tiles, align_up(panel_slab_bytes(M, w->N / 64, panel_split_k(M, w->N / 64, w->K)), 256));
  if (M > 64) return kCounterBytes + tiles;
  const size_t slabs = align_up((size_t)skinny_max_split(M) * M * w->N * sizeof(float), 256);
  return kCounterBytes + (tiles > slabs ? tiles : slabs);
}

int qllm_workspace_init(void *workspace, size_t bytes, void *stream) {
  clear_error();
  if (!workspace || bytes < kCounterBytes) return set_error(QLLM_ERR_WORKSPACE, "workspace must be at least %zu bytes", kCounterBytes);
  QLLM_HIP_CHECK(hipMemsetAsync(workspace, 0, kCounterBytes, (hipStream_t)stream));
  return QLLM_OK;
}

int qllm_linear_forward_grouped(const qllm_weight_t *w, void *const *y, int32_t n_weights, const void *x, int32_t M,
                                int32_t act_dtype, void *workspace, size_t workspace_bytes, void *stream) {
  clear_error();
  if (!w || !y) return set_error(QLLM_ERR_INVALID, "w / y arrays must not be NULL");
  if (n_weights < 1 || n_weights > kMaxProblems) return set_error(QLLM_ERR_INVALID, "n_weights must be 1..%d (got %d)", kMaxProblems, n_weights);
  if (n_weights == 1) return qllm_linear_forward(&w[0], x, y[0], M, act_dtype, workspace, workspace_bytes, stream);  // (a group of one is a plain call)
  for (int i = 0; i < n_weights; ++i) {
    int rc = validate_weight(&w[i]);
    if (rc) return rc;
    rc = check_io(x, y[i], M, act_dtype);
    if (rc) return rc;
  }
  const Decision d = decide_group(w, n_weights, M, act_dtype);
  return execute(d, w, y, n_weights, x, M, act_dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

int qllm_gather_columns(const void *x, const int32_t *perm, void *out, int32_t M, int32_t K, int32_t act_dtype, void *stream) {
  clear_error();
  if (!x || !perm || !out) return set_error(QLLM_ERR_INVALID, "x / perm / out must not be NULL");
  if (M < 0 || K <= 0) return set_error(QLLM_ERR_INVALID, "M must be >= 0 and K >= 1 (got M=%d K=%d)", M, K);
  if (act_dtype != QLLM_F16 && act_dtype != QLLM_BF16) return set_error(QLLM_ERR_INVALID, "act_dtype must be QLLM_F16 or QLLM_BF16");
  if ((uintptr_t)x % 16 || (uintptr_t)out % 16 || (uintptr_t)perm % 16) return set_error(QLLM_ERR_INVALID, "x / perm / out must be 16-byte aligned");
  if (x == out) return set_error(QLLM_ERR_INVALID, "the gather is not in-place: out must not alias x");
  if (!gather_columns_ok(K)) return set_error(QLLM_ERR_UNSUPPORTED, "column gather serves K %% 8 == 0, K <= 28672 (got %d)", K);
  if (M == 0) return QLLM_OK;
  return launch_gather_columns(x, perm, out, M, K, (hipStream_t)stream);
}

int qllm_convert_bf16_to_f16(const void *src, void *dst, size_t n, void *stream) {
  clear_error();
  if (!src || !dst || n % 8 != 0 || (uintptr_t)src % 16 != 0 || (uintptr_t)dst % 16 != 0)
    return set_error(QLLM_ERR_INVALID, "qllm_convert_bf16_to_f16: 16-byte aligned buffers, n a multiple of 8");
  return n ? launch_bf16_to_f16(src, dst, n, (hipStream_t)stream) : QLLM_OK;
}

int qllm_debug_timeline(void *buf, int32_t n_slots) {  // n_slots x 24 x u64
  clear_error();
  g_timeline = (uint64_t *)buf;
  g_timeline_slots = buf ? n_slots : 0;
  g_timeline_next = 0;
  return QLLM_OK;
}

int qllm_linear_forward(const qllm_weight_t *w, const void *x, void *y, int32_t M, int32_t act_dtype, void *workspace,
                        size_t workspace_bytes, void *stream) {
  clear_error();
  int rc = validate_weight(w);
  if (rc) return rc;
  if (act_dtype == QLLM_F16_IN_BF16_OUT) {
    // x already converted by the caller: only the 256x128 prefill kernel writes bf16 from fp16 inputs (out_bf16) -- the call must be
    // one the regular path hands to that kernel (the same Decision and the same tile choice)
    rc = check_io(x, y, M, QLLM_F16);
    if (rc) return rc;
    if (w->bits != 4 || w->g_idx || M <= 64)
      return set_error(QLLM_ERR_UNSUPPORTED, "QLLM_F16_IN_BF16_OUT: 4-bit prefill calls of the 256x128 kernel only");
    const Decision d = decide_single(w, M, QLLM_F16);
    GemmParams p;
    fill_gemm_params(p, w, x, y, M, QLLM_F16);
    const TileChoice c = d.route == ROUTE_TILE ? choose_tile(p, d.layout, usable_ws(workspace, workspace_bytes)) : TileChoice{0, 1, 0, 0, 0, 0};
    if (c.kernel != 3) return set_error(QLLM_ERR_UNSUPPORTED, "QLLM_F16_IN_BF16_OUT: M=%d K=%d N=%d is not served by the 256x128 prefill kernel", M, w->K, w->N);
    set_split(p, c.split_k, workspace, c.tail_from, c.tail_split);
    p.out_bf16 = 1;
    return launch_gemm3(p, d.layout, (hipStream_t)stream);
  }
  rc = check_io(x, y, M, act_dtype);
  if (rc) return rc;
  const Decision d = decide_single(w, M, act_dtype);
  void *ys[1] = {y};
  return execute(d, w, ys, 1, x, M, act_dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

int qllm_linear_forward_allreduce(const qllm_weight_t *w, const void *x, void *y, int32_t M, int32_t act_dtype, void *const *peers_dev,
                                  int32_t rank, int32_t world, size_t slot_bytes, int32_t *status_dev, void *stream) {
  clear_error();
  int rc = validate_weight(w);
  if (rc) return rc;
  rc = check_io(x, y, M, act_dtype);
  if (rc) return rc;
  if (!peers_dev) return set_error(QLLM_ERR_INVALID, "qllm_linear_forward_allreduce: peers_dev is NULL");
  if (world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world) return set_error(QLLM_ERR_INVALID, "world must be 1..%d and 0 <= rank < world (rank=%d world=%d)", kCommMaxWorld, rank, world);
  StripPlan pl;
  // the batch-1 kernel's shapes only (callers run the layer and qllm_allreduce_oneshot / RCCL separately for everything else)
  if (M != 1 || !is_native(*w) || w->group_size != 128 || !strip_plan(w, 1, 1, &pl) || !pl.one_nw)
    return set_error(QLLM_ERR_UNSUPPORTED, "fused all-reduce: batch-1 calls on native 4-bit layers with 128-wide groups (M=%d bits=%d g=%d layout=%d)", M, w->bits, w->group_size, w->layout);
  if ((size_t)w->N * 2 > slot_bytes || slot_bytes % 16 != 0 || (uintptr_t)y % 16 != 0)
    return set_error(QLLM_ERR_UNSUPPORTED, "fused all-reduce: N * 2 <= slot_bytes, slot_bytes %% 16 == 0, y 16-byte aligned (N=%d slot=%zu)", w->N, slot_bytes);
  Strip1Params p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.T = w->K / 32;
  p.n_groups = w->K / 128;
  p.M = 1;
  p.add_zero_bias = w->add_zero_bias;
  p.act_bf16 = (act_dtype == QLLM_BF16);
  p.prob[0] = Strip1Problem{(const uint32_t *)w->qweight, (const half_t *)w->scales, w->qzeros, (const half_t *)w->bias, y, w->N / 16, zero_kind_of(*w)};
  p.ar_peers = peers_dev;
  p.ar_status = status_dev;
  p.ar_rank = rank;
  p.ar_world = world;
  p.ar_slot_bytes = (uint32_t)slot_bytes;
  return launch_strip1_allreduce(p, pl.one_nw, pl.one_maxs, w->N / 16, (hipStream_t)stream);
}

int qllm_dequant(const qllm_weight_t *w, void *out, int32_t out_dtype, int32_t out_transposed, void *stream) {
  clear_error();
  int rc = validate_weight(w);
  if (rc) return rc;
  if (!out) return set_error(QLLM_ERR_INVALID, "out must not be NULL");
  if (out_dtype != QLLM_F16 && out_dtype != QLLM_BF16) return set_error(QLLM_ERR_INVALID, "out_dtype must be f16 or bf16");
  if (is_native(*w)) return set_error(QLLM_ERR_UNSUPPORTED, "qllm_dequant reads the reference layouts: convert with qllm_unpack_native first");
  return launch_dequant(*w, zero_kind_of(*w), out, out_dtype, out_transposed ? 1 : 0, (hipStream_t)stream);
}

int qllm_ort_gemv(const void *x, const void *qweight, const void *scales, const void *qzeros, const int32_t *g_idx,
                  int32_t groupsize, int32_t bits, int32_t in_features, int32_t add_zero_bias, void *y, int32_t M, int32_t N,
                  int32_t act_dtype, void *workspace, size_t workspace_bytes, void *stream) {
  qllm_weight_t w = {qweight, scales, qzeros, g_idx, nullptr, in_features, N, groupsize, bits, QLLM_LAYOUT_GPTQ, add_zero_bias};
  return qllm_linear_forward(&w, x, y, M, act_dtype, workspace, workspace_bytes, stream);
}

int qllm_ort_dequant(const void *qweight, const void *scales, const void *qzeros, const int32_t *g_idx, int32_t groupsize,
                     int32_t bits, int32_t in_features, int32_t add_zero_bias, void *out_kn, int32_t N, void *stream) {
  qllm_weight_t w = {qweight, scales, qzeros, g_idx, nullptr, in_features, N, groupsize, bits, QLLM_LAYOUT_GPTQ, add_zero_bias};
  return qllm_dequant(&w, out_kn, QLLM_F16, 0, stream);
}

int qllm_awq_gemm_forward(const void *x, const void *qweight, const void *scales, const void *qzeros, int32_t split_k_iters,
                          void *y, int32_t M, int32_t K, int32_t N, int32_t group_size, int32_t act_dtype, void *workspace,
                          size_t workspace_bytes, void *stream) {
  (void)split_k_iters;
  qllm_weight_t w = {qweight, scales, qzeros, nullptr, nullptr, K, N, group_size, 4, QLLM_LAYOUT_AWQ_GEMM, 0};
  return qllm_linear_forward(&w, x, y, M, act_dtype, workspace, workspace_bytes, stream);
}

int qllm_ort_dequantize4bits(const void *qweight, const void *scales, const void *qzeros, int32_t zeros_f16, const int32_t *g_idx,
                             int32_t block_size, int32_t in_features, int32_t out_features, void *out_nk, void *stream) {
  clear_error();
  if (!qweight || !scales || !qzeros || !out_nk) return set_error(QLLM_ERR_INVALID, "qweight / scales / qzeros / out must not be NULL");
  if (in_features <= 0 || out_features <= 0) return set_error(QLLM_ERR_INVALID, "in_features/out_features must be >= 1 (K=%d N=%d)", in_features, out_features);
  if (block_size < 16 || block_size % 16) return set_error(QLLM_ERR_INVALID, "block_size must be a multiple of 16 (got %d)", block_size);
  if (in_features % block_size) return set_error(QLLM_ERR_UNSUPPORTED, "in_features must be a multiple of block_size (K=%d block=%d)", in_features, block_size);
  if ((uintptr_t)qweight % 8 || (uintptr_t)out_nk % 16) return set_error(QLLM_ERR_INVALID, "qweight must be 8-byte and out 16-byte aligned");
  return launch_ort_dequant(qweight, scales, qzeros, zeros_f16 ? 1 : 0, g_idx, block_size, in_features, out_features, out_nk, (hipStream_t)stream);
}

// Which kernel a forward call with these descriptors and M rows (fp16 activations) would run, as text -- no device work.  It asks the
// SAME decision functions the forward entry points execute (decide_single / decide_group, choose_tile, *_split_for) and prints the result.
int qllm_plan_describe(const qllm_weight_t *w, int32_t n_weights, int32_t M, int32_t have_workspace, char *buf, size_t buflen) {
  clear_error();
  if (!w || !buf || buflen < 64) return set_error(QLLM_ERR_INVALID, "w / buf must not be NULL (buf >= 64 bytes)");
  if (n_weights < 1 || n_weights > kMaxProblems) return set_error(QLLM_ERR_INVALID, "n_weights must be 1..%d (got %d)", kMaxProblems, n_weights);
  if (M <= 0) return set_error(QLLM_ERR_INVALID, "M must be >= 1 (got %d)", M);
  for (int i = 0; i < n_weights; ++i) {
    const int rc = validate_weight(&w[i]);
    if (rc) return rc;
  }
  const Decision d = n_weights == 1 ? decide_single(&w[0], M, QLLM_F16) : decide_group(w, n_weights, M, QLLM_F16);
  if (d.route == ROUTE_NONE && d.rc == QLLM_ERR_INVALID) return d.rc;  // (layers that cannot share a launch at all: the forward call's own status)
  describe(d, w, n_weights, M, have_workspace ? (size_t)-1 : 0, buf, buflen);
  clear_error();
  return QLLM_OK;
}

int qllm_native_sizes(const qllm_weight_t *src, size_t *qweight_bytes, size_t *scales_bytes, size_t *qzeros_bytes) {
  clear_error();
  if (!src || !qweight_bytes || !scales_bytes || !qzeros_bytes) return set_error(QLLM_ERR_INVALID, "src / outputs must not be NULL");
  if (src->K <= 0 || src->N <= 0 || src->group_size <= 0) return set_error(QLLM_ERR_INVALID, "bad K/N/group_size (%d/%d/%d)", src->K, src->N, src->group_size);
  const bool f16z = src->layout == QLLM_LAYOUT_HQQ || src->layout == QLLM_LAYOUT_NATIVE_F16Z;
  if (int rc = native_shape_ok(src->bits, src->K, src->N, src->group_size, !f16z && src->qzeros)) return rc;
  const size_t G = (size_t)(src->K + src->group_size - 1) / src->group_size;
  *qweight_bytes = (size_t)src->K * src->bits / 32 * src->N * 4;
  *scales_bytes = G * src->N * 2;
  *qzeros_bytes = !src->qzeros ? 0 : (f16z ? G * src->N * 2 : G * (src->N / 16) * 8);
  return QLLM_OK;
}

int qllm_repack_native(const qllm_weight_t *src, void *qweight_out, void *scales_out, void *qzeros_out, void *stream) {
  clear_error();
  int rc = validate_weight(src);
  if (rc) return rc;
  if (is_native(*src)) return set_error(QLLM_ERR_INVALID, "source is already in the native layout");
  if (src->g_idx) return set_error(QLLM_ERR_UNSUPPORTED, "act-order layers: sort the rows by group first (the native layout has contiguous groups)");
  if (src->K % src->group_size != 0) return set_error(QLLM_ERR_UNSUPPORTED, "the native layout needs K %% group_size == 0 (K=%d g=%d)", src->K, src->group_size);
  if ((rc = native_shape_ok(src->bits, src->K, src->N, src->group_size, src->layout != QLLM_LAYOUT_HQQ && src->qzeros))) return rc;
  if (!qweight_out || !scales_out || (src->qzeros && !qzeros_out)) return set_error(QLLM_ERR_INVALID, "output buffers must not be NULL");
  if ((uintptr_t)qweight_out % 16 || (uintptr_t)scales_out % 16 || (uintptr_t)qzeros_out % 8) return set_error(QLLM_ERR_INVALID, "output buffers must be 16-byte (qzeros: 8-byte) aligned");
  return launch_repack_native(*src, zero_kind_of(*src), qweight_out, scales_out, qzeros_out, (hipStream_t)stream);
}

int qllm_unpack_native(const qllm_weight_t *native, int32_t dst_layout, void *qweight_out, void *scales_out, void *qzeros_out, void *stream) {
  clear_error();
  int rc = validate_weight(native);
  if (rc) return rc;
  if (!is_native(*native)) return set_error(QLLM_ERR_INVALID, "source is not in the native layout");
  if (dst_layout < QLLM_LAYOUT_GPTQ || dst_layout > QLLM_LAYOUT_HQQ) return set_error(QLLM_ERR_INVALID, "dst_layout must be GPTQ, AWQ_GEMM or HQQ");
  if ((dst_layout == QLLM_LAYOUT_HQQ) != (native->layout == QLLM_LAYOUT_NATIVE_F16Z))
    return set_error(QLLM_ERR_INVALID, "fp16 zero points <-> HQQ, packed / no zero points <-> GPTQ / AWQ_GEMM");
  if (dst_layout == QLLM_LAYOUT_AWQ_GEMM && (native->bits != 4 || !native->qzeros)) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout is 4-bit with packed zero points");
  if (!qweight_out || !scales_out || (native->qzeros && !qzeros_out)) return set_error(QLLM_ERR_INVALID, "output buffers must not be NULL");
  return launch_unpack_native(*native, dst_layout, qweight_out, scales_out, qzeros_out, (hipStream_t)stream);
}

int qllm_unpack_qweight(const void *qweight, int32_t layout, int32_t bits, int32_t K, int32_t N, int32_t *q_kn, void *stream) {
  clear_error();
  if (!qweight || !q_kn) return set_error(QLLM_ERR_INVALID, "qweight / q_kn must not be NULL");
  if (bits < 2 || bits > 8 || K <= 0 || N <= 0) return set_error(QLLM_ERR_INVALID, "bad bits/K/N (%d/%d/%d)", bits, K, N);
  if (layout == QLLM_LAYOUT_AWQ_GEMM && (bits != 4 || N % 8)) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout is 4-bit, N%%8==0");
  if (layout != QLLM_LAYOUT_AWQ_GEMM && (K * bits) % 32) return set_error(QLLM_ERR_INVALID, "K*bits must be a multiple of 32");
  return launch_unpack_qweight(qweight, layout, bits, K, N, q_kn, (hipStream_t)stream);
}

int qllm_pack_qweight(const int32_t *q_kn, int32_t layout, int32_t bits, int32_t K, int32_t N, void *qweight, void *stream) {
  clear_error();
  if (!qweight || !q_kn) return set_error(QLLM_ERR_INVALID, "qweight / q_kn must not be NULL");
  if (bits < 2 || bits > 8 || K <= 0 || N <= 0) return set_error(QLLM_ERR_INVALID, "bad bits/K/N (%d/%d/%d)", bits, K, N);
  if (layout == QLLM_LAYOUT_AWQ_GEMM && (bits != 4 || N % 8)) return set_error(QLLM_ERR_INVALID, "AWQ GEMM layout is 4-bit, N%%8==0");
  if (layout != QLLM_LAYOUT_AWQ_GEMM && (K * bits) % 32) return set_error(QLLM_ERR_INVALID, "K*bits must be a multiple of 32");
  return launch_pack_qweight(q_kn, layout, bits, K, N, qweight, (hipStream_t)stream);
}

}  // extern "C"
