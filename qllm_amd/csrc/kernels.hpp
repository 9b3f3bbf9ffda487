// Parameter blocks and host launchers shared by the translation units of libqllm_mi355x.
#pragma once
#include "common.hpp"

namespace qllm {

// ---- dequant.hip ---------------------------------------------------------------------------------------------
int launch_dequant(const qllm_weight_t &w, int zero_kind, void *out, int out_dtype, int out_transposed, hipStream_t stream);
int launch_unpack_qweight(const void *qweight, int layout, int bits, int K, int N, int32_t *q_kn, hipStream_t stream);
int launch_pack_qweight(const int32_t *q_kn, int layout, int bits, int K, int N, void *qweight, hipStream_t stream);

// ---- ortblob.hip (ORT / MatMulNBits blob layout -> W[N,K] fp16) ---------------------------------------------------
int launch_ort_dequant(const void *qweight, const void *scales, const void *qzeros, int zeros_f16, const int32_t *g_idx,
                       int block, int K, int N, void *out, hipStream_t stream);

// ---- skinny.hip ----------------------------------------------------------------------------------------------
constexpr int kMaxProblems = 8;

struct SkinnyProblem {
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const half_t *bias;
  void *y;
  float *slabs;   // [S][M][N] fp32 (only if S > 1)
  int *counters;  // [n_tiles]
  int N;
  int n_tiles;
  int S;            // blocks along K
  int spw;          // k-steps (32 k) per wave
  int block_begin;  // first blockIdx.x of this problem
  int zero_kind;
};

struct SkinnyParams {
  SkinnyProblem prob[kMaxProblems];
  const void *x;
  int n_prob;
  int M, K, T;  // T = K / 32
  int group_size;
  int add_zero_bias;
  int act_bf16;
};

int skinny_tile_cols(int layout, int awq_w);
int skinny_max_split(int M);
void skinny_plan(int K, int M, int tiles_total, int target_waves, int *S_out, int *spw_out);
int launch_skinny(const SkinnyParams &p, int layout, int awq_w, int grid, hipStream_t stream);

// ---- strip.hip -----------------------------------------------------------------------------------------------
struct StripProblem {
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const half_t *bias;
  void *y;
  int N;
  int n_strips;     // blocks of this problem: ceil(N / (16 * cpl))
  int block_begin;  // first blockIdx.x of this problem
  int zero_kind;
};

struct StripParams {
  // the words every wave needs first, in ONE 64-byte line at the head of the argument block (one s_load_dwordx8 instead of eight
  // loads from seven lines in two dependent batches): the first weight load leaves after 2 scalar waits instead of 6; measured
  // +0.7 % on the whole decode step (864 -> 871 tok/s, profiles/logs/r02q_hdr.log)
  int block_begin8[kMaxProblems];
  StripProblem prob[kMaxProblems];
  const void *x;
  int n_prob;
  int M, K, T;  // T = K / 32
  int nw;       // waves per block (4, 8 or 16)
  int cpl;      // columns per lane: 1 (16-column strips), 2 or 4 (64-column strips)
  int bits;     // 4, or 3 (bit-stream layout; cpl = 1, fp16 or symmetric zeros)
  int spw;      // k-steps per wave (nw waves per block cover all of K)
  int ra;       // 0: lds-slab form; 1: "register A" form (A fragments straight from L2, Sx / Sx' from two extra MFMAs); 2: strip_dma.hpp
                // (native layout, M = 5..32: activations through LDS by DMA)
  int group_size;
  int add_zero_bias;
  int act_bf16;
  int n_groups;     // strip-major: K / group_size (rows of the per-strip scale / zero tables)
  int sm;           // 1: strip-major native layout (strip_kernel.hpp, SM): every prob[] pointer is a native-layout buffer
  uint64_t *dbg;    // diagnostics (qllm_debug_timeline): 24 timestamps for this launch (3 blocks x 8), or NULL
};
bool strip_group_ok(int group_size, bool strip_major, int bits);
int strip_nw(int K, int strips_total, int compute_units);
int strip_sm_nw(int K, int M, int group_size, int bits);
int strip_spw(int K, int group_size, int nw);
// `sm`: 0 = row-stream layouts read in place, 1 = strip-major native layout
int strip_maxs(int nw, int spw, int cpl, int ra, int sm);
int strip_xl(int nw, int M, int spw, int cpl, int sm);
size_t strip_lds_bytes(int M, int spw, int nw, int cpl, int group_size, int ra, int sm);
int strip_cpl(int cols_total, bool all_mult64, bool all_mult32, int compute_units);
bool strip_x_ok(int M, int spw, int nw, int cpl, int sm);
int launch_strip(const StripParams &p, int grid, hipStream_t stream);
int launch_strip_sm(const StripParams &p, int grid, hipStream_t stream);     // strip_sm.hip
int launch_strip_sm_ra(const StripParams &p, int grid, hipStream_t stream);  // strip_sm_ra.hip
int launch_strip_dma_g32(const StripParams &p, int grid, hipStream_t stream);   // strip_dma_g32.hip
int launch_strip_dma_g64(const StripParams &p, int grid, hipStream_t stream);   // strip_dma_g64.hip
int launch_strip_dma_g128(const StripParams &p, int grid, hipStream_t stream);  // strip_dma_g128.hip

// ---- strip1.hip (batch 1, native layout, 4 bits, 128-wide groups: strip1_kernel.hpp) -------------------------------------------
struct Strip1Problem {  // 48 bytes
  const uint32_t *qweight;  // native: [N/16][K/8][16] words (3 bits: [N/16][3 K/32][16])
  const half_t *scales;     // native: [N/16][K/g][16]  (g = 128 or 64)
  const void *qzeros;       // native: [N/16][K/128][2] words (packed), [N/16][K/128][16] halves (fp16), NULL (symmetric)
  const half_t *bias;
  void *y;
  int n_strips;             // N / 16
  int zero_kind;
};
struct Strip1Params {
  const void *x;
  int T;          // K / 32
  int n_groups;   // K / group size (128, or 64: group64)
  int group64;    // 1: 64-wide groups (the G64 instantiations, round 6)
  int M;          // batch rows: 1, or 2..4 on the four-row instantiations (MR = 4, round 6; 128-wide groups)
  int bits3;      // 1: 3-bit layers (the B3 instantiations, round 6; batch 1, K <= 16384)
  int add_zero_bias;
  int act_bf16;
  uint64_t *dbg;  // diagnostics (qllm_debug_timeline): 24 timestamps for this launch (3 blocks x 8), or NULL
  Strip1Problem prob[kMaxProblems];
  // row-parallel layer fused with its one-shot all-reduce (comm.hip; the AR instantiations, one layer per launch): every block pushes
  // its 16 partial outputs into every peer's staging slot; the rank's last block publishes the flags, waits for the world and sums
  void *const *ar_peers;   // device array of the world's staging buffers (this rank's at index ar_rank)
  int *ar_status;          // set to 1 if a peer never arrived (nullable)
  int ar_rank, ar_world;
  uint32_t ar_slot_bytes;
};

// ---- bitgemv.hip (round 6): fused decode matvec for the widths the MFMA strips do not take (2, 5, 6, 7, 8 bits; row-stream layouts) -----
constexpr int kBitGemvMaxM = 16;
struct BitGemvParams {
  const void *x;
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const half_t *bias;
  void *y;
  float *slabs;   // K-split: [ksplit][M][N] fp32 partial results
  int *counters;  // K-split: one arrival counter per 32-column block (zero before and after the launch)
  int M, K, N, group_size, zero_kind, add_zero_bias, act_bf16;
  int ksplit;                      // blocks along K (1: none)
  int n_col_blocks, chunk_units;   // set by launch_bitgemv
};
bool bitgemv_ok(const qllm_weight_t &w, int M);
int bitgemv_split(int M, int K, int N);
int bitgemv_cols();
int launch_bitgemv(const BitGemvParams &p, int bits, hipStream_t stream);

// ---- comm.hip: staging buffer of one rank = [2 parities][world][slot_bytes] payload | this control block ----------------------------
constexpr int kCommMaxWorld = 16;
struct CommCtl {
  uint32_t flag[2][kCommMaxWorld];  // flag[parity][src rank] = epoch of the last push
  uint32_t epoch;                   // calls completed by the owner (read and bumped by its own kernels only)
  uint32_t ticket;                  // fused GEMV + all-reduce: blocks of the running launch that have pushed (re-armed by the last one)
};
// 16-byte system-scope (write-through) store / load: what crosses xGMI
__device__ __forceinline__ void store16_sys(void *p, uint4_t v) {
  __hip_atomic_store((uint64_t *)p, ((uint64_t)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store((uint64_t *)p + 1, ((uint64_t)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint4_t load16_sys(const void *p) {
  const uint64_t a = __hip_atomic_load((const uint64_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const uint64_t b = __hip_atomic_load((const uint64_t *)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return uint4_t{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
bool strip1_shape(int K, int blocks, int cus, int *nw, int *maxs);
int launch_strip1(const Strip1Params &p, int nw, int maxs, int n_prob, int max_strips, hipStream_t stream);
int launch_strip1_allreduce(const Strip1Params &p, int nw, int maxs, int n_strips, hipStream_t stream);  // (p.ar_* set; one layer)

// ---- native.hip (reference layouts <-> the strip-major native layout) -------------------------------------------------------
int launch_repack_native(const qllm_weight_t &src, int zero_kind, void *qweight_out, void *scales_out, void *qzeros_out, hipStream_t stream);
int launch_unpack_native(const qllm_weight_t &src, int dst_layout, void *qweight_out, void *scales_out, void *qzeros_out, hipStream_t stream);

// ---- gather.hip (act-order: out[m, k] = x[m, perm[k]], 2-byte elements) ------------------------------------------------------
bool gather_columns_ok(int K);
int launch_gather_columns(const void *x, const int32_t *perm, void *out, int M, int K, hipStream_t stream);

// ---- gemm.hip ------------------------------------------------------------------------------------------------
struct GemmParams {
  const void *x;
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const int32_t *g_idx;
  const half_t *bias;
  void *y;
  int M, K, N, group_size, gs_shift, add_zero_bias, zero_kind, act_bf16, n_groups;
  int raster;   // gemm2 tile order: 0 = m fastest, 1 = n fastest inside an XCD's run
  int stagger;  // gemm2: waves 4-7 run their VALU phase before their MFMA phase
  int out_bf16; // gemm3: y is bf16 (x was converted to fp16 by the pre-pass: the reference's bf16 shim, quant_linear_awq.py:29-36)
  int sm;       // 1: row-stream layout stored strip-major (the native layout): word (r, n) at ((n / 16) * rows + r) * 16 + n % 16
  int split_k;      // gemm2: blocks per output tile along K (1 = none)
  float *slabs;     // gemm2 split-K: [tiles][split_k][256 x 128] fp32 partial tiles
  int *counters;    // gemm2 split-K: one arrival counter per output tile (zero before and after the launch)
  // gemm3, round 6: K-split of the ragged LAST round only.  tail_split > 1: tiles [0, tail_from) run all of K, one block each (whole
  // rounds of CUs); every tile from tail_from on is shared by tail_split blocks (slab / counter index = tile - tail_from), so that the
  // last round is as full as the others and 1 / tail_split as long.  split_k must be 1 then.
  int tail_from, tail_split;
  // gemm3, round 6: layers sharing x in ONE launch (q/k/v, gate/up at prefill sizes).  n_prob > 1: the single-layer pointer fields
  // above are unused; total_tiles = sum of the layers' 256x128 tiles, layer i's tiles are [tile_begin, tile_begin + tiles_m * N / 128)
  int n_prob, total_tiles;
  struct GemmProb {
    const uint32_t *qweight;
    const half_t *scales;
    const void *qzeros;
    const half_t *bias;
    void *y;
    int N, zero_kind, tile_begin, pad_;
  } prob[4];
  int native_bf16;  // gemm3 (round 6): x is bf16 and stays bf16 -- bf16 W, bf16 MFMA, bf16 y (LAYOUT 0; gemm3_bf16_native())
  int prio;         // gemm4: s_setprio values of its wave roles (matrix | dequant << 4 | loader << 8)
  uint64_t *dbg;    // lab builds (-DQLLM_LAB): gemm4 timeline, 16 x u64 per block (tools/lab/g4lab timeline); NULL otherwise
};
int launch_gemm(const GemmParams &p, int layout, hipStream_t stream);

// ---- gemm2.hip (256x256 tile; fp16 activations, trivial groups, N % 256 == 0) -------------------------------------
bool gemm2_ok(const GemmParams &p, int layout);
int gemm2_split_k(int M, int N, int K);
constexpr int kGemm3MaxProb = 4;
using GemmProb = GemmParams::GemmProb;
int gemm3_tail_split(int M, int N, int K, int *tail_from);
int gemm3_tail_split_tiles(int tiles, int K, int *tail_from);
bool gemm3_bf16_native(int layout);  // gemm3.hip: K-split factor of the ragged last round of tiles (1: none)
int gemm2_tile_n(int M, int N, int split_k);
size_t gemm2_slab_bytes(int M, int N, int S);
int launch_gemm2(const GemmParams &p, int layout, hipStream_t stream);

// ---- panel.hip (17 <= M <= 128 on the native layout: 64-column panels, A tiles shared through LDS,
//      B fragments from registers; up to 8 layers sharing x in one launch) ----------------------------------------------------------
struct PanelProblem {
  const uint32_t *qweight;
  const half_t *scales;
  const void *qzeros;
  const half_t *bias;
  void *y;
  int N, zero_kind, panel_begin, pad_;
};
struct PanelParams {
  const void *x;
  int M, K, n_groups, group_size, bits, add_zero_bias, act_bf16, n_prob, split_k, abl, n_panels;
  float *slabs;   // split-K: [panels][split_k][4 waves x row tiles x 4 x 64] fp32 partial panels
  int *counters;  // split-K: one arrival counter per panel (zero before and after the launch)
  PanelProblem prob[kMaxProblems];
};
bool panel_shape_ok(int M, int K, int N, int group_size, int bits);
int panel_mt(int M);
int panel_kh(int M);
int panel_split_k(int M, int n_panels, int K);
size_t panel_slab_bytes(int M, int n_panels, int S);
int launch_panel(const PanelParams &p, hipStream_t stream);

// ---- gemm3.hip (256x128 tile, 4 matrix waves + 4 staging waves; no split-K) ---------------------------------------------
constexpr int kGemm3Rows3Bit = 100;  // `layout` value for launch_gemm3 / gemm3_ok: GPTQ / HQQ row stream with 3-bit weights
bool gemm3_ok(const GemmParams &p, int layout);
int gemm3_split_k(int M, int N, int K);
int launch_gemm3(const GemmParams &p, int layout, hipStream_t stream);
int launch_bf16_to_f16(const void *src, void *dst, size_t n, hipStream_t stream);  // elementwise RNE conversion (gemm3's bf16 pre-pass)

// ---- tools/lab/gemm5.hip (lab builds: 256x128 tile, every wave a matrix wave, B fragments dequantised in registers) ------------------
bool gemm5_ok(const GemmParams &p, int layout);
int launch_gemm5(const GemmParams &p, int wm, hipStream_t stream);

// ---- tools/lab/gemm4.hip (lab builds: 256x128 tile, matrix waves + all-DMA producer waves; same contract as gemm3) ----------------------------------
int launch_gemm4(const GemmParams &p, int layout, int variant, hipStream_t stream);

}  // namespace qllm
