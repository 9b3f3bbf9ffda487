// Chained decode links, round 6 prototype (developer tool, not part of the library): the Llama-2-7B decode step at batch 1 on the native
// strip-major layout, driven three ways on the same buffers --
//   P4  the product's launch structure: q/k/v grouped, o, gate/up grouped, down on ONE stream (4 dependent launches per layer);
//   P5  the same kernels with gate and up as two launches (5 per layer), one stream;
//   C5  the CHAINED form (strip1_kernel<..., CH>): the same five launches alternate between TWO streams with no edge between them
//       inside the step; a link may be resident before its input exists -- it issues its weight loads, then polls its input in band
//       (the activation arena is armed with 0xFFFF halves by one memset at the head of the step; outputs are stored write-through).
//       A link enters when its same-stream predecessor (two links back) has finished, i.e. while the link in front of it still runs:
//       its weights stream through the launch boundary and the ramp of the plain form.
//   C5s the chained kernels on one stream (every poll succeeds at once): what the chained instantiations cost serialised.
// 32 layers of distinct weights (3.4 GB: nothing is served from the Infinity Cache), one hipGraph per step, replayed.
// Every intermediate vector of C5 must equal P4's bit for bit.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I qllm_amd/csrc -I tools/lab -o tools/lab/chainlab tools/lab/chainlab.hip \
//        -L tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "strip1_lab_kernel.hpp"

using namespace qllm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __host__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ void fill_words(uint32_t *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = hash32((uint32_t)i * 2654435761u + seed);
}
__global__ void fill_scales(_Float16 *p, size_t n, uint32_t seed, float base) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (_Float16)(((hash32((uint32_t)i + seed) & 0xffff) / 65536.f * 0.4f + 0.8f) * base);
}
__global__ void fill_x(_Float16 *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float s = 0;
    for (int j = 0; j < 4; ++j) s += (hash32((uint32_t)i * 4 + j + seed) & 0xffff) / 65536.f - 0.5f;
    p[i] = (_Float16)(s * 1.732f);
  }
}

struct Lin { const uint32_t *w; const _Float16 *s; const uint32_t *z; int K, N; const uint32_t *sz; };
// LABV bit 1: scale and zero point of a (strip, group, column) as one dword {fp16 scale | zero << 16}, from the layer's own tables
__global__ void fuse_sz(const _Float16 *s, const uint32_t *z, uint32_t *sz, size_t n) {
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // e = (strip * G + g) * 16 + i
  if (e >= n) return;
  const size_t row = e >> 4;
  const int i = (int)(e & 15);
  const uint32_t zn = (z[row * 2 + (i >> 3)] >> (4 * (i & 7))) & 15u;
  uint16_t sb;
  memcpy(&sb, &s[e], 2);
  sz[e] = (uint32_t)sb | (zn << 16);
}
static Lin make_native(int K, int N, uint32_t seed) {
  const size_t qw = (size_t)K / 8 * N, G = K / 128;
  uint32_t *w, *z;
  _Float16 *s;
  CK(hipMalloc(&w, qw * 4));
  CK(hipMalloc(&z, G * (N / 16) * 8));
  CK(hipMalloc(&s, G * N * 2));
  fill_words<<<(qw + 255) / 256, 256>>>(w, qw, seed);
  fill_words<<<(G * (N / 16) * 2 + 255) / 256, 256>>>(z, G * (N / 16) * 2, seed ^ 0x9e3779b9u);
  fill_scales<<<(G * N + 255) / 256, 256>>>(s, G * N, seed ^ 0x1234567u, 1.f / (sqrtf((float)K) * 6.5f));
  uint32_t *sz;
  CK(hipMalloc(&sz, G * N * 4));
  fuse_sz<<<(G * N + 255) / 256, 256>>>(s, z, sz, G * N);
  return Lin{w, s, z, K, N, sz};
}

constexpr int H = 4096, I = 11008;
struct Layer { Lin q, k, v, o, gate, up, down; };
// activation arena of one step: per layer q k v o gate up down (halves)
constexpr size_t kPerLayer = 3 * H + H + 2 * I + H;
struct Acts { _Float16 *q, *k, *v, *o, *gate, *up, *down; };
static Acts acts_of(_Float16 *arena, int l) {
  _Float16 *b = arena + (size_t)l * kPerLayer;
  return Acts{b, b + H, b + 2 * H, b + 3 * H, b + 4 * H, b + 4 * H + I, b + 4 * H + 2 * I};
}

static int g_cus = 256;
static int *g_err;
static uint32_t g_spin_limit = 40000;

static uint64_t *g_dbg = nullptr;   // --timeline: 24 x u64 per link (3 blocks x 8 stamps)
static int g_link = 0;
static int g_labv = 0;   // plain forms: the LABV variant of strip1_kernel to launch (0 = the product kernel)
template <int NW, int MAXS, bool EXACT, bool CH>
static void launch_t(Strip1ParamsLab &p, dim3 grid, hipStream_t st) {
  constexpr int lds_bytes = strip1_lds_bytes<NW, MAXS>();
  if (g_dbg) {
    p.dbg = g_dbg + 24 * (g_link++);
    hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, true, false, CH>), grid, dim3(NW * 64), lds_bytes, st, p);
    return;
  }
  if constexpr (!CH) {
    switch (g_labv) {
      case 1: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 1>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 2: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 2>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 3: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 3>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 4: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 4>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 6: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 6>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 7: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 7>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 8: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 8>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      case 9: hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, false, 9>), grid, dim3(NW * 64), lds_bytes, st, p); return;
      default: break;
    }
  }
  hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, 2, 4, false, false, CH>), grid, dim3(NW * 64), lds_bytes, st, p);
}

// one launch: `n` layers sharing x (K), outputs y[i]
template <bool CH>
static void launch(const Lin *const *L, _Float16 *const *y, int n, const _Float16 *x, hipStream_t st, int force_nw = 0) {
  if (st == nullptr) { if (g_dbg) ++g_link; return; }  // (this link belongs to the other stream's graph)
  Strip1ParamsLab p;
  memset(&p, 0, sizeof(p));
  const int K = L[0]->K;
  p.x = x;
  p.T = K / 32;
  p.n_groups = K / 128;
  p.ch_spin_limit = g_spin_limit;
  p.ch_err = g_err;
  int max_strips = 0, strips = 0;
  for (int i = 0; i < n; ++i) {
    p.prob[i] = Strip1Problem{L[i]->w, (g_labv & 2) && !CH ? (const half_t *)L[i]->sz : (const half_t *)L[i]->s, L[i]->z, nullptr, y[i], L[i]->N / 16, ZK_PACKED};
    max_strips = std::max(max_strips, L[i]->N / 16);
    strips += L[i]->N / 16;
  }
  const dim3 grid(max_strips, n);
  if (K == 4096) {
    if ((strips <= g_cus && force_nw == 0) || force_nw == 4) launch_t<4, 32, true, CH>(p, grid, st);
    else launch_t<8, 16, true, CH>(p, grid, st);
  } else if (K == 11008) {
    launch_t<15, 24, false, CH>(p, grid, st);
  } else {
    printf("no form for K=%d\n", K);
    exit(1);
  }
}

enum Form { P4, P5, C5, C5S, C7, P7 };
static const char *form_name(Form f) {
  switch (f) {
    case P4: return "P4  plain, 4 launches per layer, one stream (the product's structure)";
    case P5: return "P5  plain, 5 launches per layer (gate | up), one stream";
    case P7: return "P7  plain, 7 launches per layer, one stream";
    case C5: return "C5  chained, 5 links per layer, two streams";
    case C5S: return "C5s chained kernels, 5 links per layer, ONE stream (serialised)";
    case C7: return "C7  chained, 7 links per layer, two streams";
  }
  return "?";
}

// issue one decode step (32 layers) in `form` onto streams sa / sb (sb unused by the plain forms)
static int g_only = 0;  // 0: every link; 1 / 2: only the links of stream A / B (two graphs, one per stream)
static void issue_step(Form f, const std::vector<Layer> &Ls, const _Float16 *h0, _Float16 *arena, hipStream_t sa, hipStream_t sb) {
  const bool chained = (f == C5 || f == C5S || f == C7);
  hipStream_t s2 = (f == C5 || f == C7) ? sb : sa;
  int turn = 0;
  auto next = [&]() -> hipStream_t { hipStream_t s = (turn & 1) ? s2 : sa; const int which = (turn & 1) ? 2 : 1; ++turn; return (g_only && g_only != which) ? nullptr : s; };
  const _Float16 *h = h0;
  for (size_t l = 0; l < Ls.size(); ++l) {
    const Layer &Y = Ls[l];
    const Acts a = acts_of(arena, (int)l);
    if (f == P4) {
      { const Lin *w[3] = {&Y.q, &Y.k, &Y.v}; _Float16 *y[3] = {a.q, a.k, a.v}; launch<false>(w, y, 3, h, sa); }
      { const Lin *w[1] = {&Y.o}; _Float16 *y[1] = {a.o}; launch<false>(w, y, 1, a.q, sa); }
      { const Lin *w[2] = {&Y.gate, &Y.up}; _Float16 *y[2] = {a.gate, a.up}; launch<false>(w, y, 2, a.o, sa); }
      { const Lin *w[1] = {&Y.down}; _Float16 *y[1] = {a.down}; launch<false>(w, y, 1, a.gate, sa); }
    } else if (f == P5 || f == C5 || f == C5S) {
      auto L1 = [&](const Lin *w0, _Float16 *y0, const _Float16 *x, hipStream_t st) {
        const Lin *w[1] = {w0}; _Float16 *y[1] = {y0};
        if (chained) launch<true>(w, y, 1, x, st); else launch<false>(w, y, 1, x, st);
      };
      { const Lin *w[3] = {&Y.q, &Y.k, &Y.v}; _Float16 *y[3] = {a.q, a.k, a.v};
        hipStream_t st = next();
        if (chained) launch<true>(w, y, 3, h, st); else launch<false>(w, y, 3, h, st); }
      L1(&Y.o, a.o, a.q, next());
      L1(&Y.gate, a.gate, a.o, next());
      L1(&Y.up, a.up, a.o, next());
      L1(&Y.down, a.down, a.gate, next());
    } else {  // 7 launches / links
      auto L1 = [&](const Lin *w0, _Float16 *y0, const _Float16 *x, hipStream_t st) {
        const Lin *w[1] = {w0}; _Float16 *y[1] = {y0};
        if (chained) launch<true>(w, y, 1, x, st, 8); else launch<false>(w, y, 1, x, st, 8);
      };
      L1(&Y.q, a.q, h, next());
      L1(&Y.k, a.k, h, next());
      L1(&Y.v, a.v, h, next());
      L1(&Y.o, a.o, a.q, next());
      L1(&Y.gate, a.gate, a.o, next());
      L1(&Y.up, a.up, a.o, next());
      L1(&Y.down, a.down, a.gate, next());
    }
    h = a.down;
  }
}

struct StepGraph { hipGraph_t g; hipGraphExec_t ge; };
static StepGraph capture_step(Form f, const std::vector<Layer> &Ls, const _Float16 *h0, _Float16 *arena, hipStream_t sa, hipStream_t sb,
                              hipEvent_t fork, hipEvent_t join) {
  const bool two = (f == C5 || f == C7);
  const bool chained = two || f == C5S;
  StepGraph s;
  CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
  if (chained) CK(hipMemsetAsync(arena, 0xFF, Ls.size() * kPerLayer * 2, sa));   // arm: "not written yet"
  if (two) { CK(hipEventRecord(fork, sa)); CK(hipStreamWaitEvent(sb, fork, 0)); }
  issue_step(f, Ls, h0, arena, sa, sb);
  if (two) { CK(hipEventRecord(join, sb)); CK(hipStreamWaitEvent(sa, join, 0)); }
  CK(hipStreamEndCapture(sa, &s.g));
  CK(hipGraphInstantiate(&s.ge, s.g, nullptr, nullptr, 0));
  return s;
}

static float time_replays(StepGraph &s, hipStream_t st, int replays, int reps = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(s.ge, st));
  CK(hipStreamSynchronize(st));
  float best = 1e30f;
  for (int rep = 0; rep < reps; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(s.ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms * 1e3f / replays);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best;
}

int main(int argc, char **argv) {
  int layers = 32, replays = 20;
  bool do7 = false, timeline = false, variants = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--layers")) layers = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--replays")) replays = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--spin")) g_spin_limit = (uint32_t)atoi(argv[++i]);
    else if (!strcmp(argv[i], "--seven")) do7 = true;
    else if (!strcmp(argv[i], "--timeline")) timeline = true;
    else if (!strcmp(argv[i], "--variants")) variants = true;
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs; %d layers, %d replays (best of 3)\n", prop.gcnArchName, g_cus, layers, replays);
  std::vector<Layer> Ls;
  for (int l = 0; l < layers; ++l) {
    const uint32_t s = 1000 + 97 * l;
    Ls.push_back(Layer{make_native(H, H, s), make_native(H, H, s + 1), make_native(H, H, s + 2), make_native(H, H, s + 3),
                       make_native(H, I, s + 4), make_native(H, I, s + 5), make_native(I, H, s + 6)});
  }
  _Float16 *h0, *arena_ref, *arena;
  const size_t arena_halves = (size_t)layers * kPerLayer;
  CK(hipMalloc(&h0, H * 2));
  CK(hipMalloc(&arena_ref, arena_halves * 2));
  CK(hipMalloc(&arena, arena_halves * 2));
  CK(hipMalloc(&g_err, 4));
  CK(hipMemset(g_err, 0, 4));
  fill_x<<<(H + 255) / 256, 256>>>(h0, H, 77);
  CK(hipDeviceSynchronize());
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t fork, join;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

  // reference: the plain step, eagerly
  CK(hipMemset(arena_ref, 0, arena_halves * 2));
  issue_step(P4, Ls, h0, arena_ref, sa, sb);
  CK(hipStreamSynchronize(sa));
  std::vector<uint16_t> ref(arena_halves), got(arena_halves);
  CK(hipMemcpy(ref.data(), arena_ref, arena_halves * 2, hipMemcpyDeviceToHost));
  {
    double mx = 0; int bad = 0;
    for (size_t i = arena_halves - H; i < arena_halves; ++i) { _Float16 v; memcpy(&v, &ref[i], 2); if (!(fabs((double)v) < 1e4)) ++bad; mx = fmax(mx, fabs((double)v)); }
    printf("reference step: last layer's output max |y| = %.3f, non-finite %d\n", mx, bad);
  }

  if (timeline) {
    // in-kernel stamps (100 MHz s_memrealtime, wave 0 of the first / middle / last block of each link) of one step: plain P5 on one
    // stream, then the chained step as two graphs
    const int links = layers * 5;
    CK(hipMalloc(&g_dbg, (size_t)links * 24 * 8));
    std::vector<uint64_t> st((size_t)links * 24);
    const char *names[5] = {"qkv", "o", "gate", "up", "down"};
    for (int mode = 0; mode < 2; ++mode) {
      CK(hipMemset(g_dbg, 0, (size_t)links * 24 * 8));
      CK(hipMemset(g_err, 0, 4));
      if (mode == 0) {
        for (int rep = 0; rep < 2; ++rep) { g_link = 0; issue_step(P5, Ls, h0, arena, sa, sb); }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamSynchronize(sa));
        g_link = 0;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        issue_step(P5, Ls, h0, arena, sa, sb);
        CK(hipStreamEndCapture(sa, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, sa));
        CK(hipStreamSynchronize(sa));
      } else {
        hipGraph_t ga, gb; hipGraphExec_t gea, geb;
        g_only = 1; g_link = 0;
        CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
        issue_step(C5, Ls, h0, arena, sa, sb);
        CK(hipStreamEndCapture(sa, &ga));
        g_only = 2; g_link = 0;
        CK(hipStreamBeginCapture(sb, hipStreamCaptureModeGlobal));
        issue_step(C5, Ls, h0, arena, sa, sb);
        CK(hipStreamEndCapture(sb, &gb));
        g_only = 0;
        CK(hipGraphInstantiate(&gea, ga, nullptr, nullptr, 0));
        CK(hipGraphInstantiate(&geb, gb, nullptr, nullptr, 0));
        hipEvent_t armed, eb;
        CK(hipEventCreateWithFlags(&armed, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
        CK(hipEventRecord(eb, sb));
        for (int i = 0; i < 3; ++i) {
          CK(hipStreamWaitEvent(sa, eb, 0));
          CK(hipMemsetAsync(arena, 0xFF, arena_halves * 2, sa));
          CK(hipEventRecord(armed, sa));
          CK(hipStreamWaitEvent(sb, armed, 0));
          CK(hipGraphLaunch(gea, sa));
          CK(hipGraphLaunch(geb, sb));
          CK(hipEventRecord(eb, sb));
        }
        CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      }
      CK(hipMemcpy(st.data(), g_dbg, st.size() * 8, hipMemcpyDeviceToHost));
      int err = 0;
      CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
      const int l0 = std::min(8, layers - 3);
      const uint64_t base = st[(size_t)(l0 * 5) * 24];
      printf("\n%s -- us from the entry of layer %d's qkv (block 0 | last block): entry issued x_ok(polls) mfma exit   [timeouts %d]\n",
             mode == 0 ? "P5 plain, one stream" : "C5g chained, two graphs", l0, err);
      for (int k = l0 * 5; k < (l0 + 3) * 5; ++k) {
        const uint64_t *b0 = &st[(size_t)k * 24], *bl = &st[(size_t)k * 24 + 16];
        auto us = [&](uint64_t v) { return v ? ((double)v - (double)base) / 100.0 : -1.0; };
        printf("  L%-2d %-5s | %7.2f %7.2f %7.2f(%3llu) xst %7.2f %7.2f bar %7.2f %7.2f | %7.2f %7.2f %7.2f(%3llu) xst %7.2f %7.2f bar %7.2f %7.2f\n", k / 5, names[k % 5],
               us(b0[0]), us(b0[1]), us(b0[6]), (unsigned long long)b0[7], us(b0[2]), us(b0[3]), us(b0[4]), us(b0[5]),
               us(bl[0]), us(bl[1]), us(bl[6]), (unsigned long long)bl[7], us(bl[2]), us(bl[3]), us(bl[4]), us(bl[5]));
      }
    }
    return 0;
  }
  if (variants) {
    // the product's launch structure (P4) with the LABV variants of the batch-1 kernel, interleaved rounds, same buffers
    const int vs[] = {0, 8, 2, 9, 1};
    const char *vn[] = {"product kernel", "scale by DWORD load (same layout)", "scale + zero as ONE dword (new layout)", "scale by dword load + y write-through",
                        "y write-through (sc1)"};
    std::vector<StepGraph> gs;
    for (int v : vs) { g_labv = v; gs.push_back(capture_step(P4, Ls, h0, arena, sa, sb, fork, join)); }
    g_labv = 0;
    for (size_t i = 0; i < gs.size(); ++i) {
      CK(hipMemset(arena, 0, arena_halves * 2));
      CK(hipDeviceSynchronize());
      CK(hipGraphLaunch(gs[i].ge, sa));
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff = 0, first = arena_halves;
      for (size_t k = 0; k < arena_halves; ++k) if (got[k] != ref[k]) { if (first == arena_halves) first = k; ++diff; }
      printf("variant %d (%s): mismatches vs the product kernel %zu", vs[i], vn[i], diff);
      if (diff) {
        printf("  first at %zu (layer %zu, offset %zu):", first, first / kPerLayer, first % kPerLayer);
        for (size_t k = first; k < first + 6 && k < arena_halves; ++k) printf(" %04x/%04x", got[k], ref[k]);
      }
      printf("\n");
    }
    const int rounds = 10;
    std::vector<std::vector<float>> t(gs.size());
    for (int round = 0; round < rounds; ++round) {
      printf("round %2d:", round);
      for (size_t i = 0; i < gs.size(); ++i) {
        const float us = time_replays(gs[i], sa, replays, 1) / layers;
        t[i].push_back(us);
        printf(" %6.2f", us);
      }
      printf("\n");
      fflush(stdout);
    }
    for (size_t i = 0; i < gs.size(); ++i) {
      std::sort(t[i].begin(), t[i].end());
      printf("P4 + %-40s min %6.2f  median %6.2f  max %6.2f us per layer  (median: %7.1f tok/s)\n", vn[i], t[i].front(), t[i][rounds / 2], t[i].back(),
             1e6 / (t[i][rounds / 2] * 32));
    }
    return 0;
  }
  std::vector<Form> forms = {P4, P5, C5S, C5};
  if (do7) { forms.push_back(P7); forms.push_back(C7); }
  for (int round = 0; round < 2; ++round) {
    for (Form f : forms) {
      StepGraph s = capture_step(f, Ls, h0, arena, sa, sb, fork, join);
      CK(hipMemset(arena, 0, arena_halves * 2));
      CK(hipMemset(g_err, 0, 4));
      CK(hipGraphLaunch(s.ge, sa));
      CK(hipStreamSynchronize(sa));
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff = 0;
      for (size_t i = 0; i < arena_halves; ++i) diff += got[i] != ref[i];
      const float us = time_replays(s, sa, replays);
      int err = 0;
      CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
      // (after the timed replays: the arena must still equal the reference)
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff2 = 0;
      for (size_t i = 0; i < arena_halves; ++i) diff2 += got[i] != ref[i];
      const double bytes = 3369484288.0 * layers / 32;
      printf("%-72s %8.2f us per layer  %7.1f tok/s at 32 layers  %.4f of 8 TB/s   mismatches %zu / %zu  timeouts %d\n", form_name(f), us / layers,
             1e6 / (us / layers * 32), bytes / (us * 1e-6) / 8e12, diff, diff2, err);
      fflush(stdout);
      CK(hipGraphExecDestroy(s.ge));
      CK(hipGraphDestroy(s.g));
    }
    // ---- the chained step as TWO graphs, one per stream (each a plain chain of launches), launched side by side; the streams meet
    //      once per step through events (eager) ---------------------------------------------------------------------------------
    for (Form f : {C5, C7}) {
      if (f == C7 && !do7) continue;
      hipGraph_t ga, gb; hipGraphExec_t gea, geb;
      g_only = 1;
      CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
      CK(hipMemsetAsync(arena, 0xFF, arena_halves * 2, sa));
      issue_step(f, Ls, h0, arena, sa, sb);
      CK(hipStreamEndCapture(sa, &ga));
      g_only = 2;
      CK(hipStreamBeginCapture(sb, hipStreamCaptureModeGlobal));
      issue_step(f, Ls, h0, arena, sa, sb);
      CK(hipStreamEndCapture(sb, &gb));
      g_only = 0;
      CK(hipGraphInstantiate(&gea, ga, nullptr, nullptr, 0));
      CK(hipGraphInstantiate(&geb, gb, nullptr, nullptr, 0));
      hipEvent_t ea, eb, armed;
      CK(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb, hipEventDisableTiming));
      CK(hipEventCreateWithFlags(&armed, hipEventDisableTiming));
      hipGraph_t gm; hipGraphExec_t gem;   // the arming memset as its own tiny graph? no: plain async memset on sa, then both graphs
      (void)gm; (void)gem;
      auto step2 = [&]() {
        // B's links of this step must not start before A's arming memset: the memset is the head of graph A, so B waits for an event
        // recorded on A right AFTER a separate memset; graph A then re-arms nothing itself
        CK(hipStreamWaitEvent(sa, eb, 0));          // A: the previous step's B links are done (they read what this step re-arms)
        CK(hipGraphLaunch(gea, sa));                // (head: the arming memset)
        CK(hipEventRecord(ea, sa));
        CK(hipStreamWaitEvent(sb, ea, 0));          // !! orders B behind ALL of A -- replaced below by the split form
        CK(hipGraphLaunch(geb, sb));
        CK(hipEventRecord(eb, sb));
      };
      (void)step2;
      // split form: memset eagerly on sa, event, then the two graphs side by side (graph A captured WITHOUT the memset)
      CK(hipGraphExecDestroy(gea)); CK(hipGraphDestroy(ga));
      g_only = 1;
      CK(hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal));
      issue_step(f, Ls, h0, arena, sa, sb);
      CK(hipStreamEndCapture(sa, &ga));
      g_only = 0;
      CK(hipGraphInstantiate(&gea, ga, nullptr, nullptr, 0));
      auto step = [&]() {
        CK(hipStreamWaitEvent(sa, eb, 0));                         // the previous step's B links have read what is re-armed now
        CK(hipMemsetAsync(arena, 0xFF, arena_halves * 2, sa));
        CK(hipEventRecord(armed, sa));
        CK(hipStreamWaitEvent(sb, armed, 0));
        CK(hipGraphLaunch(gea, sa));
        CK(hipGraphLaunch(geb, sb));
        CK(hipEventRecord(eb, sb));
      };
      CK(hipEventRecord(eb, sb));
      CK(hipMemset(g_err, 0, 4));
      for (int i = 0; i < 3; ++i) step();
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff = 0;
      for (size_t i = 0; i < arena_halves; ++i) diff += got[i] != ref[i];
      float best = 1e30f;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < replays; ++i) step();
        CK(hipStreamWaitEvent(sa, eb, 0));
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / replays);
      }
      int err = 0;
      CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff2 = 0;
      for (size_t i = 0; i < arena_halves; ++i) diff2 += got[i] != ref[i];
      printf("%-72s %8.2f us per layer  %7.1f tok/s at 32 layers  %.4f of 8 TB/s   mismatches %zu / %zu  timeouts %d\n",
             f == C5 ? "C5g chained, 5 links per layer, TWO graphs side by side" : "C7g chained, 7 links per layer, TWO graphs side by side", best / layers,
             1e6 / (best / layers * 32), 3369484288.0 * layers / 32 / (best * 1e-6) / 8e12, diff, diff2, err);
      fflush(stdout);
      CK(hipGraphExecDestroy(gea)); CK(hipGraphExecDestroy(geb)); CK(hipGraphDestroy(ga)); CK(hipGraphDestroy(gb));
    }
    // ---- eager, two streams (no graph): the protocol itself --------------------------------------------------------------------
    {
      CK(hipMemset(g_err, 0, 4));
      hipEvent_t e0, e1, ej, ef;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
      auto step = [&]() {
        CK(hipMemsetAsync(arena, 0xFF, arena_halves * 2, sa));
        CK(hipEventRecord(ef, sa)); CK(hipStreamWaitEvent(sb, ef, 0));
        issue_step(C5, Ls, h0, arena, sa, sb);
        CK(hipEventRecord(ej, sb)); CK(hipStreamWaitEvent(sa, ej, 0));
      };
      for (int i = 0; i < 2; ++i) step();
      CK(hipStreamSynchronize(sa));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, sa));
        for (int i = 0; i < 5; ++i) step();
        CK(hipEventRecord(e1, sa));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / 5);
      }
      int err = 0;
      CK(hipMemcpy(&err, g_err, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(got.data(), arena, arena_halves * 2, hipMemcpyDeviceToHost));
      size_t diff2 = 0;
      for (size_t i = 0; i < arena_halves; ++i) diff2 += got[i] != ref[i];
      printf("%-72s %8.2f us per layer  %7.1f tok/s at 32 layers  (host-paced)  mismatches %zu  timeouts %d\n", "C5e chained, 5 links per layer, two streams, EAGER launches", best / layers,
             1e6 / (best / layers * 32), diff2, err);
      fflush(stdout);
    }
  }
  return 0;
}
