// Weight-streaming MFMA "skinny" kernel: y[M<=64, N] = x . dequant(W4) for decode-sized M.
//
// Replaces the reference's fused matvecs (gemv<half> /root/reference/csrc/ort_cuda/dq_gemv.cu:41-150,
// Gemv_g :461-541) and the M<=16 regime of gemm_forward_4bit_cuda_m16n128k32
// (csrc/awq_cuda/quantization/gemm_cuda_gen.cu:31-353), for the GPTQ / HQQ row-stream layout and the
// AWQ-GEMM interleaved layout, read in place (no relayout, no dequantised W in memory).
//
// Design (HBM-bound; see DESIGN.md section "skinny"):
//   * one wave streams a [K-chunk x TN columns] panel of packed weights straight into VGPRs with 16-byte /
//     8-byte loads whose row segments are whole 128/256-byte lines; nothing is staged through LDS.
//   * a GPTQ word (8 consecutive-k nibbles of one column) IS one lane's B fragment of
//     v_mfma_f32_16x16x32_f16; it is dequantised in registers (v_and_or_b32 + v_pk_fma_f16 + v_pk_add_f16,
//     bit-identical to the reference's fp16 W) and fed to the matrix core.  The k-slot order inside a
//     fragment is (k0,k4,k1,k5,...) -- the matching permutation is applied to the tiny A (activation) fragment.
//     AWQ words hold 8 columns of one k: two rows are byte-permuted into "column-pair" words first
//     (2 v_perm_b32 per 16 weights), after which the same nibble decode yields (k,k+1) pairs.
//   * M is padded to 16-row MFMA tiles (MT tiles); the matrix core is >10x over-provisioned here, so M=1..16
//     cost the same and VALU work is only the dequant.
//   * K is split over the 4 waves of a block (LDS reduction) and over S blocks (fp32 slabs written through with
//     sc1 stores, arrival ticket, last-arriving block sums the slabs in fixed order -> deterministic; the
//     reference's fp16 split-k partials, gemm_cuda_gen.cu:1115,1160, are not reproduced).
//   * up to 8 layers that share x (q/k/v, gate/up) run as ONE launch (block ranges per problem).
#include <stdlib.h>

#include "kernels.hpp"

namespace qllm {

template <int MT>
__device__ __forceinline__ void load_a(const SkinnyParams &p, int t, int g, int i, half8_t (&a)[MT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = 16 * mt + i;
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < p.M) {
      const size_t off = (size_t)row * p.K + 32 * t + 8 * g;
      if (p.act_bf16)
        v = bf16x8_to_h8(*(const uint4_t *)((const uint16_t *)p.x + off));
      else
        v = *(const half8_t *)((const half_t *)p.x + off);
    }
    a[mt] = v;
  }
}

// LAYOUT 0: GPTQ/HQQ row stream, lane owns 4 adjacent columns (one dwordx4 per k-step), TN = 64
// LAYOUT 1: AWQ GEMM, lane owns W words = 8W adjacent columns (8 loads of W dwords per k-step), TN = 128 W
// XLDS: the block's activation slab x[M][its K range] is staged once into LDS (permuted to the fragment's k-slot order for
//       LAYOUT 0) and A fragments are read from there; without it every k-step loads its A fragment from L2, which is
//       latency-bound (measured 2x on the strip kernel).  Used whenever the slab fits.
template <int LAYOUT, int W, int MT, bool XLDS>
__global__ __launch_bounds__(256) void skinny_kernel(const SkinnyParams p) {
  constexpr int CPL = (LAYOUT == 0) ? 4 : 8 * W;  // columns per lane
  constexpr int TN = 16 * CPL;
  extern __shared__ __attribute__((aligned(16))) float red[];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane >> 4, i = lane & 15;

  int pi = 0;
#pragma unroll
  for (int q = 1; q < kMaxProblems; ++q)
    if (q < p.n_prob && (int)blockIdx.x >= p.prob[q].block_begin) pi = q;
  const SkinnyProblem &pr = p.prob[pi];

  const int b = blockIdx.x - pr.block_begin;
  const int ntile = b % pr.n_tiles;
  const int kb = b / pr.n_tiles;
  const int N = pr.N;
  const int col0 = ntile * TN;       // first column of the tile
  const int n_lane = col0 + i * CPL;  // first column of this lane
  const bool col_ok = n_lane < N;     // N % 8 == 0 and CPL | 8 (or CPL = 16 with per-word checks below)

  const int t0 = (kb * 4 + wave) * pr.spw;
  const int t1 = min(t0 + pr.spw, p.T);

  float4_t acc[MT][CPL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int c = 0; c < CPL; ++c) acc[mt][c] = float4_t{0.f, 0.f, 0.f, 0.f};

  ColConst cc[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) cc[c] = make_col_const((half_t)0.f, (half_t)0.f);
  int next_group_k = 0;  // reload constants when the lane's k position reaches this

  auto reload_consts = [&](int kpos) {
    const int G = kpos / p.group_size;
    next_group_k = (G + 1) * p.group_size;
    if (!col_ok) return;
    const half_t *sp = pr.scales + (size_t)G * N + n_lane;
    half_t s[CPL], z[CPL];
    if constexpr (LAYOUT == 0) {
      half4_t sv = *(const half4_t *)sp;
      s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
      if (pr.zero_kind == ZK_PACKED) {
        const uint32_t zw = ((const uint32_t *)pr.qzeros)[(size_t)G * (N >> 3) + (n_lane >> 3)];
        const int sh = (n_lane & 4) * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = (half_t)(float)(((zw >> (sh + 4 * c)) + (uint32_t)p.add_zero_bias) & 15u);
      } else if (pr.zero_kind == ZK_F16) {
        half4_t zv = *(const half4_t *)((const half_t *)pr.qzeros + (size_t)G * N + n_lane);
        z[0] = zv.x; z[1] = zv.y; z[2] = zv.z; z[3] = zv.w;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) z[c] = (half_t)8.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const bool wok = (n_lane + 8 * e) < N;
        half8_t sv = {0, 0, 0, 0, 0, 0, 0, 0};
        uint32_t zw = 0;
        if (wok) {
          sv = *(const half8_t *)(sp + 8 * e);
          zw = ((const uint32_t *)pr.qzeros)[(size_t)G * (N >> 3) + (n_lane >> 3) + e];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s[8 * e + c] = sv[c];
          z[8 * e + c] = (half_t)(float)((zw >> (4 * awq_nibble_of_col(c))) & 15u);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) cc[c] = make_col_const(s[c], z[c]);
  };

  // ---- raw weight registers for one k-step -------------------------------------------------------------
  constexpr int RAW = (LAYOUT == 0) ? 4 : 8 * W;  // dwords per lane per k-step
  struct Step {
    uint32_t w[RAW];
    half8_t a[MT];
  };

  auto load_step = [&](int t, Step &s) {
    if (t < t1) {
      if constexpr (LAYOUT == 0) {
        uint4_t v = {0u, 0u, 0u, 0u};
        if (col_ok) v = __builtin_nontemporal_load((const uint4_t *)(pr.qweight + (size_t)(4 * t + g) * N + n_lane));
        s.w[0] = v.x; s.w[1] = v.y; s.w[2] = v.z; s.w[3] = v.w;
      } else {
        const int nw = N >> 3;
        const uint32_t *base = pr.qweight + (size_t)(32 * t + 8 * g) * nw + (n_lane >> 3);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if constexpr (W == 2) {
            uint2_t v = {0u, 0u};
            if (n_lane + 8 < N) {
              v = __builtin_nontemporal_load((const uint2_t *)(base + (size_t)r * nw));
            } else if (col_ok) {
              v.x = __builtin_nontemporal_load(base + (size_t)r * nw);
            }
            s.w[2 * r] = v.x;
            s.w[2 * r + 1] = v.y;
          } else {
            s.w[r] = col_ok ? __builtin_nontemporal_load(base + (size_t)r * nw) : 0u;
          }
        }
      }
      if constexpr (!XLDS) load_a<MT>(p, t, g, i, s.a);
    }
  };

  // activation slab of this block in LDS: rows M x (4*spw k-steps), row stride + 16 B
  const int kb0 = kb * 4 * pr.spw;                 // first k-step of the block
  const int xrow = 4 * pr.spw * 32 + 8;            // halves per staged row
  half_t *xs = (half_t *)(red + 4 * min(p.M, 16) * TN + 4);
  auto lds_a = [&](int t, int mt) -> half8_t {
    return *(const half8_t *)(xs + min(16 * mt + i, p.M - 1) * xrow + 32 * (t - kb0) + 8 * g);
  };

  auto compute_step = [&](int t, const Step &s) {
    if (t >= t1) return;
    const int kpos = 32 * t + 8 * g;
    if (kpos >= next_group_k) reload_consts(kpos);
    if constexpr (LAYOUT == 0) {
      half8_t ap[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) ap[mt] = XLDS ? lds_a(t, mt) : a_perm_04152637(s.a[mt]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const half8_t bf = deq_word_k04(s.w[c], cc[c]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ap[mt], bf, acc[mt][c], 0, 0, 0);
      }
    } else {
      half8_t an[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) an[mt] = XLDS ? lds_a(t, mt) : s.a[mt];
#pragma unroll
      for (int e = 0; e < W; ++e) {
        uint32_t P[4], Q[4];
#pragma unroll
        for (int rp = 0; rp < 4; ++rp) {
          const uint32_t wa = s.w[(2 * rp) * W + e], wb = s.w[(2 * rp + 1) * W + e];
          P[rp] = __builtin_amdgcn_perm(wb, wa, 0x05040100u);  // (wa.lo16 | wb.lo16 << 16): nibbles 0..3 of both rows
          Q[rp] = __builtin_amdgcn_perm(wb, wa, 0x07060302u);  // (wa.hi16 | wb.hi16 << 16): nibbles 4..7
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int sh = 4 * (c >> 1);
          const ColConst &k = cc[8 * e + c];
          half2_t b0, b1, b2, b3;
          if (c & 1) {
            b0 = deq_pair(and_or(Q[0] >> sh, kNibLo, kMagic), k);
            b1 = deq_pair(and_or(Q[1] >> sh, kNibLo, kMagic), k);
            b2 = deq_pair(and_or(Q[2] >> sh, kNibLo, kMagic), k);
            b3 = deq_pair(and_or(Q[3] >> sh, kNibLo, kMagic), k);
          } else {
            b0 = deq_pair(and_or(P[0] >> sh, kNibLo, kMagic), k);
            b1 = deq_pair(and_or(P[1] >> sh, kNibLo, kMagic), k);
            b2 = deq_pair(and_or(P[2] >> sh, kNibLo, kMagic), k);
            b3 = deq_pair(and_or(P[3] >> sh, kNibLo, kMagic), k);
          }
          const half8_t bf = {b0.x, b0.y, b1.x, b1.y, b2.x, b2.y, b3.x, b3.y};  // natural (k0..k7)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][8 * e + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(an[mt], bf, acc[mt][8 * e + c], 0, 0, 0);
        }
      }
    }
  };

  // ---- main loop: U steps in flight, ping-pong register sets ----------------------------------------------
  constexpr int U = (LAYOUT == 0) ? ((MT == 1) ? 4 : 2) : ((W == 1 && MT == 1) ? 2 : 1);
  Step sa[U], sb[U];
#pragma unroll
  for (int u = 0; u < U; ++u) load_step(t0 + u, sa[u]);
  if constexpr (XLDS) {
    // stage the slab while the first weight loads are in flight; k >= K is staged as zero
    const int cpr = 4 * pr.spw * 4;  // 16-byte chunks per row
    for (int c = threadIdx.x; c < p.M * cpr; c += 256) {
      const int row = c / cpr, kc = c - row * cpr;
      const int k = 32 * kb0 + 8 * kc;
      half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (k < p.K) {
        const size_t off = (size_t)row * p.K + k;
        if (p.act_bf16)
          v = bf16x8_to_h8(*(const uint4_t *)((const uint16_t *)p.x + off));
        else
          v = *(const half8_t *)((const half_t *)p.x + off);
      }
      *(half8_t *)(xs + row * xrow + 8 * kc) = (LAYOUT == 0) ? a_perm_04152637(v) : v;
    }
    __syncthreads();
  }
  for (int tb = t0; tb < t1; tb += 2 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) load_step(tb + U + u, sb[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) compute_step(tb + u, sa[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) load_step(tb + 2 * U + u, sa[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) compute_step(tb + U + u, sb[u]);
  }

  // ---- in-block reduction over the 4 waves, then output / split-K slab -----------------------------------
  const int M = p.M;
  int *s_flag = (int *)(red + 4 * min(M, 16) * TN);
  const bool bf16_out = p.act_bf16;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int rows = min(M - 16 * mt, 16);  // valid rows of this m-tile
    if (rows <= 0) break;
    if (mt > 0) __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * g + r;
      if (row < rows) {
        float *dst = red + ((size_t)(wave * rows + row) * TN + i * CPL);
#pragma unroll
        for (int c = 0; c < CPL; c += 4)
          *(float4_t *)(dst + c) = float4_t{acc[mt][c][r], acc[mt][c + 1][r], acc[mt][c + 2][r], acc[mt][c + 3][r]};
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < rows * TN; e += 256) {
      const int row = e / TN, col = e - row * TN;
      const int n = col0 + col;
      if (n >= N) continue;
      float v = red[(size_t)(0 * rows + row) * TN + col] + red[(size_t)(1 * rows + row) * TN + col];
      v += red[(size_t)(2 * rows + row) * TN + col];
      v += red[(size_t)(3 * rows + row) * TN + col];
      const int m = 16 * mt + row;
      if (pr.S == 1) {
        if (pr.bias) v += (float)pr.bias[n];
        if (bf16_out)
          ((uint16_t *)pr.y)[(size_t)m * N + n] = f32_to_bf16(v);
        else
          ((half_t *)pr.y)[(size_t)m * N + n] = (half_t)v;
      } else {
        st_sc1(pr.slabs + ((size_t)kb * M + m) * N + n, v);
      }
    }
  }
  if (pr.S == 1) return;

  // publish: every storing wave drains its write-through stores, then ONE lane takes a ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0)
    *s_flag = __hip_atomic_fetch_add(pr.counters + ntile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (*s_flag != pr.S - 1) return;

  // last arriver for this column tile: sum the S slabs in fixed order (deterministic), add bias, write y
  const int tile_cols = min(TN, N - col0);
  for (int e = threadIdx.x; e < M * tile_cols; e += 256) {
    const int m = e / tile_cols, col = e - m * tile_cols;
    const int n = col0 + col;
    float v = 0.f;
    for (int s = 0; s < pr.S; ++s) v += ld_sc1(pr.slabs + ((size_t)s * M + m) * N + n);
    if (pr.bias) v += (float)pr.bias[n];
    if (bf16_out)
      ((uint16_t *)pr.y)[(size_t)m * N + n] = f32_to_bf16(v);
    else
      ((half_t *)pr.y)[(size_t)m * N + n] = (half_t)v;
  }
  if (threadIdx.x == 0) __hip_atomic_store(pr.counters + ntile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- host side -------------------------------------------------------------------------------------------
int skinny_tile_cols(int layout, int awq_w) { return layout == QLLM_LAYOUT_AWQ_GEMM ? 128 * awq_w : 64; }

int skinny_max_split(int M) { return M <= 16 ? 16 : (M <= 32 ? 8 : 4); }

// number of K-blocks S and k-steps per wave for a launch whose problems total `tiles_total` column tiles
void skinny_plan(int K, int M, int tiles_total, int target_waves, int *S_out, int *spw_out) {
  const int T = K / 32;
  int wpt = (target_waves + tiles_total - 1) / tiles_total;  // waves per column tile along K
  int S = (wpt + 3) / 4;
  const int cap = skinny_max_split(M);
  if (S > cap) S = cap;
  const int by_len = T / 8 > 0 ? T / 8 : 1;  // keep >= 2 k-steps per wave
  if (S > by_len) S = by_len;
  if (S < 1) S = 1;
  int spw = (T + 4 * S - 1) / (4 * S);
  // drop K-blocks that would be entirely empty
  while (S > 1 && (S - 1) * 4 * spw >= T) --S;
  *S_out = S;
  *spw_out = spw;
}

template <int LAYOUT, int W, bool XLDS>
static int launch_mt(const SkinnyParams &p, int mt, int grid, size_t lds, hipStream_t stream) {
#define QLLM_SK(MT_)                                                                                              \
  do {                                                                                                            \
    static DeviceLatch attr_done; /* per (kernel, device): the LDS opt-in is a per-device attribute */              \
    if (int rc = lds_optin(attr_done, (const void *)skinny_kernel<LAYOUT, W, MT_, XLDS>)) return rc;              \
    hipLaunchKernelGGL((skinny_kernel<LAYOUT, W, MT_, XLDS>), dim3(grid), dim3(256), lds, stream, p);             \
  } while (0)
  if constexpr (W == 2) {  // the 16-columns-per-lane AWQ variant only exists for one M-tile (register budget)
    QLLM_SK(1);
  } else {
    switch (mt) {
      case 1: QLLM_SK(1); break;
      case 2: QLLM_SK(2); break;
      default: QLLM_SK(4); break;
    }
  }
#undef QLLM_SK
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

int launch_skinny(const SkinnyParams &p, int layout, int awq_w, int grid, hipStream_t stream) {
  const int mt = p.M <= 16 ? 1 : (p.M <= 32 ? 2 : 4);
  const int tn = skinny_tile_cols(layout, awq_w);
  const size_t red_bytes = (size_t)4 * (p.M < 16 ? p.M : 16) * tn * sizeof(float) + 16;
  int spw = 0;
  for (int i = 0; i < p.n_prob; ++i) spw = p.prob[i].spw > spw ? p.prob[i].spw : spw;
  const size_t x_bytes = (size_t)p.M * (4 * spw * 32 + 8) * sizeof(half_t);
  const bool no_xlds = !knob("QLLM_SKINNY_XLDS", 1);
  const bool xlds = !no_xlds && red_bytes + x_bytes <= 144 * 1024;
  const size_t lds = red_bytes + (xlds ? x_bytes : 0);
  if (layout == QLLM_LAYOUT_AWQ_GEMM) {
    if (awq_w == 2 && mt != 1) return set_error(QLLM_ERR_INVALID, "internal: AWQ W=2 needs M <= 16");
    if (awq_w == 2) return xlds ? launch_mt<1, 2, true>(p, mt, grid, lds, stream) : launch_mt<1, 2, false>(p, mt, grid, lds, stream);
    return xlds ? launch_mt<1, 1, true>(p, mt, grid, lds, stream) : launch_mt<1, 1, false>(p, mt, grid, lds, stream);
  }
  return xlds ? launch_mt<0, 1, true>(p, mt, grid, lds, stream) : launch_mt<0, 1, false>(p, mt, grid, lds, stream);
}

}  // namespace qllm
