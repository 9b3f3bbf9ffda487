// Decode bisect, round 5 (developer tool, not part of the library): the four launches of a Llama-2-7B decoder layer at batch 1 --
// q/k/v (3 layers in one launch), o_proj, gate/up (2), down_proj -- on the native strip-major layout, timed per launch shape and per
// layer for
//   * the batch-1 kernel of strip1_kernel.hpp at every bisect level (LVL 0: loads + xor + reduce + store, the memlab2 skeleton
//     behind the kernel's own prologue; 1: + scale / zero loads; 2: + x staged through LDS and the A fragments read back; 3: +
//     pattern build and MFMAs; 4: + corrections = the product kernel) and with 1 / 2 / 4 accumulator chains,
//   * the library's two paths through the C ABI: the general strip kernel (QLLM_STRIP1=0, round 3/4's production path) and the
//     batch-1 kernel (QLLM_STRIP1=1) -- needs the LAB build of the library (knobs re-read at every call),
//   * the Llama-2-70B TP = 8 shard shapes (BASELINE configs[4]) through the C ABI, both paths, and both K = 8192 forms.
// Every timing is a hipGraph of one launch per rotating layer copy (32 copies: 3.4 GB, nothing is served from the 256 MB
// Infinity Cache), replayed; microseconds per launch = graph time / launches.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I qllm_amd/csrc -o tools/lab/dbisect tools/lab/dbisect.hip \
//        -L tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <functional>
#include <string>
#include <vector>

#include "strip1_kernel.hpp"

using namespace qllm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define QK(x) do { int r_ = (x); if (r_ != 0) { printf("qllm error %d (%s) at %s:%d\n", r_, qllm_last_error(), __FILE__, __LINE__); exit(1); } } while (0)

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

struct Lin {  // one layer in the native layout (random integers: the layout is a permutation, any words are a valid layer)
  qllm_weight_t d{};
};
static Lin make_native(int K, int N, uint32_t seed) {
  Lin L;
  const size_t qw = (size_t)K / 8 * N, G = K / 128;
  uint32_t *w, *z;
  _Float16 *s;
  CK(hipMalloc(&w, qw * 4));
  CK(hipMalloc(&z, G * (N / 16) * 8));
  CK(hipMalloc(&s, G * N * 2));
  fill_words<<<(qw + 255) / 256, 256>>>(w, qw, seed);
  fill_words<<<(G * (N / 16) * 2 + 255) / 256, 256>>>(z, G * (N / 16) * 2, seed ^ 0x9e3779b9u);
  fill_scales<<<(G * N + 255) / 256, 256>>>(s, G * N, seed ^ 0x1234567u, 1.f / (sqrtf((float)K) * 6.5f));
  L.d = qllm_weight_t{w, s, z, nullptr, nullptr, K, N, 128, 4, QLLM_LAYOUT_NATIVE, 0};
  return L;
}

struct Launch {  // one launch shape: n layers sharing x, `copies` rotating instances
  const char *name;
  int K, n;
  int N[3];
  std::vector<std::vector<Lin>> inst;  // [copy][layer]
  _Float16 *x;
  _Float16 *y[3];
  double bytes() const {
    double b = 0;
    for (int i = 0; i < n; ++i) b += (double)K * N[i] / 2 + (double)(K / 128) * N[i] * 2.5 + 2.0 * N[i];
    return b + 2.0 * K;
  }
};

static void *g_ws;
static const size_t kWs = 64 << 20;

static Strip1Params params_of(const Launch &L, int copy, int *max_strips) {
  Strip1Params p;
  memset(&p, 0, sizeof(p));
  p.x = L.x;
  p.T = L.K / 32;
  p.n_groups = L.K / 128;
  *max_strips = 0;
  for (int i = 0; i < L.n; ++i) {
    const qllm_weight_t &d = L.inst[copy][i].d;
    p.prob[i] = Strip1Problem{(const uint32_t *)d.qweight, (const half_t *)d.scales, d.qzeros, nullptr, L.y[i], d.N / 16, ZK_PACKED};
    if (d.N / 16 > *max_strips) *max_strips = d.N / 16;
  }
  return p;
}

template <int NW, int MAXS, bool EXACT, int NCH, int LVL>
static void launch_lab(const Launch &L, int copy, hipStream_t st) {
  int ms;
  const Strip1Params p = params_of(L, copy, &ms);
  constexpr int lds_bytes = strip1_lds_bytes<NW, MAXS>();
  hipLaunchKernelGGL((strip1_kernel<NW, MAXS, EXACT, NCH, LVL, false>), dim3(ms, L.n), dim3(NW * 64), lds_bytes, st, p);
}
// the two Llama-2-7B forms: K = 4096 -> 8 waves x 16 exact; K = 11008 -> 16 waves x 24
template <int NCH, int LVL>
static void launch_lvl(const Launch &L, int copy, hipStream_t st) {
  if (L.K == 4096) launch_lab<8, 16, true, NCH, LVL>(L, copy, st);
  else launch_lab<16, 24, false, NCH, LVL>(L, copy, st);
}
// alternative block shapes for K = 4096 (lab only)
template <int NW, int MAXS>
static void launch_alt(const Launch &L, int copy, hipStream_t st) {
  if (L.K == 4096) launch_lab<NW, MAXS, true, 2, 4>(L, copy, st);
  else launch_lab<16, 24, false, 2, 4>(L, copy, st);
}

static void launch_capi(const Launch &L, int copy, hipStream_t st) {
  qllm_weight_t w[3];
  void *y[3];
  for (int i = 0; i < L.n; ++i) { w[i] = L.inst[copy][i].d; y[i] = L.y[i]; }
  if (L.n == 1) QK(qllm_linear_forward(&w[0], L.x, y[0], 1, QLLM_F16, g_ws, kWs, st));
  else QK(qllm_linear_forward_grouped(w, y, L.n, L.x, 1, QLLM_F16, g_ws, kWs, st));
}

typedef std::function<void(const Launch &, int, hipStream_t)> LaunchFn;

// graph of `seq` (one launch per entry, rotating copies), replayed: microseconds per graph
static float time_graph(const std::vector<std::pair<const Launch *, int>> &seq, const LaunchFn &fn, hipStream_t st, int replays) {
  for (auto &e : seq) fn(*e.first, e.second, st);  // eager warm-up (also loads the code objects)
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (auto &e : seq) fn(*e.first, e.second, st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < replays; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms * 1e3f / replays);
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return best;
}

static std::vector<_Float16> fetch(const Launch &L) {
  std::vector<_Float16> out;
  for (int i = 0; i < L.n; ++i) {
    std::vector<_Float16> h(L.N[i]);
    CK(hipMemcpy(h.data(), L.y[i], h.size() * 2, hipMemcpyDeviceToHost));
    out.insert(out.end(), h.begin(), h.end());
  }
  return out;
}
static double max_rel(const std::vector<_Float16> &a, const std::vector<_Float16> &b) {
  double md = 0, mx = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - (double)b[i])); mx = fmax(mx, fabs((double)a[i])); }
  return md / fmax(mx, 1e-30);
}

static Launch make_launch(const char *name, int K, std::vector<int> Ns, int copies, uint32_t seed) {
  Launch L;
  L.name = name;
  L.K = K;
  L.n = (int)Ns.size();
  for (int i = 0; i < L.n; ++i) L.N[i] = Ns[i];
  L.inst.resize(copies);
  for (int c = 0; c < copies; ++c)
    for (int i = 0; i < L.n; ++i) L.inst[c].push_back(make_native(K, Ns[i], seed + 17 * c + i));
  CK(hipMalloc(&L.x, (size_t)K * 2));
  fill_x<<<(K + 255) / 256, 256>>>(L.x, K, seed ^ 0x55u);
  for (int i = 0; i < L.n; ++i) CK(hipMalloc(&L.y[i], (size_t)Ns[i] * 2));
  CK(hipDeviceSynchronize());
  return L;
}

int main(int argc, char **argv) {
  int copies = 32, replays = 10;
  bool tp = true;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--copies")) copies = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--replays")) replays = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--no-tp")) tp = false;
  }
  qllm_device_info_t info;
  QK(qllm_device_info(0, &info));
  printf("device %s, %d CUs; %d rotating copies per launch shape, %d graph replays (best of 3)\n", info.arch, info.compute_units, copies, replays);
  CK(hipMalloc(&g_ws, kWs));
  CK(hipMemset(g_ws, 0, kWs));
  hipStream_t st;
  CK(hipStreamCreate(&st));

  std::vector<Launch> L7;
  L7.push_back(make_launch("q/k/v 4096->3x4096", 4096, {4096, 4096, 4096}, copies, 100));
  L7.push_back(make_launch("o 4096->4096", 4096, {4096}, copies, 200));
  L7.push_back(make_launch("gate/up 4096->2x11008", 4096, {11008, 11008}, copies, 300));
  L7.push_back(make_launch("down 11008->4096", 11008, {4096}, copies, 400));

  struct Variant { std::string name; LaunchFn fn; const char *env; };
  std::vector<Variant> vs;
  vs.push_back({"C ABI, general strip kernel (QLLM_STRIP1=0: rounds 3-4)", launch_capi, "0"});
  vs.push_back({"strip1 LVL0: loads + xor + reduce + store", launch_lvl<2, 0>, nullptr});
  vs.push_back({"strip1 LVL1: + scale / zero loads", launch_lvl<2, 1>, nullptr});
  vs.push_back({"strip1 LVL2: + x staged via LDS, A reads", launch_lvl<2, 2>, nullptr});
  vs.push_back({"strip1 LVL3: + pattern build + MFMA", launch_lvl<2, 3>, nullptr});
  vs.push_back({"strip1 LVL4: + corrections (2 chains)", launch_lvl<2, 4>, nullptr});
  vs.push_back({"strip1 LVL4, 1 chain", launch_lvl<1, 4>, nullptr});
  vs.push_back({"strip1 LVL4, 4 chains", launch_lvl<4, 4>, nullptr});
  vs.push_back({"strip1 LVL4, K=4096 as 4 waves x 32", launch_alt<4, 32>, nullptr});
  vs.push_back({"strip1 LVL4, K=4096 as 16 waves x 8", launch_alt<16, 8>, nullptr});
  vs.push_back({"C ABI, batch-1 kernel (QLLM_STRIP1=1)", launch_capi, "1"});

  // ---- parity: every full variant against the general strip kernel on copy 0 ----------------------------------------------------
  printf("\nparity vs the general strip kernel (max |dy| / max |y|, copy 0):\n");
  for (auto &L : L7) {
    setenv("QLLM_STRIP1", "0", 1);
    launch_capi(L, 0, st);
    CK(hipStreamSynchronize(st));
    const auto ref = fetch(L);
    for (auto &v : vs) {
      if (v.name.find("LVL4") == std::string::npos && v.name.find("batch-1") == std::string::npos) continue;
      if (v.env) setenv("QLLM_STRIP1", v.env, 1);
      for (int i = 0; i < L.n; ++i) CK(hipMemsetAsync(L.y[i], 0xff, (size_t)L.N[i] * 2, st));
      v.fn(L, 0, st);
      CK(hipStreamSynchronize(st));
      const double r = max_rel(ref, fetch(L));
      printf("  %-24s %-52s %.3g %s\n", L.name, v.name.c_str(), r, r <= 1e-3 ? "OK" : "MISMATCH");
    }
  }

  // ---- per launch shape ------------------------------------------------------------------------------------------------------------
  printf("\nus per launch (graph of %d launches, one per rotating copy):\n", copies);
  printf("  %-58s", "variant");
  for (auto &L : L7) printf(" %10.10s", L.name);
  printf("   sum\n");
  std::vector<std::vector<float>> tab(vs.size(), std::vector<float>(L7.size()));
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    if (vs[vi].env) setenv("QLLM_STRIP1", vs[vi].env, 1);
    printf("  %-58s", vs[vi].name.c_str());
    float sum = 0;
    for (size_t li = 0; li < L7.size(); ++li) {
      std::vector<std::pair<const Launch *, int>> seq;
      for (int c = 0; c < copies; ++c) seq.push_back({&L7[li], c});
      const float us = time_graph(seq, vs[vi].fn, st, replays) / copies;
      tab[vi][li] = us;
      sum += us;
      printf(" %10.2f", us);
      fflush(stdout);
    }
    printf(" %6.2f\n", sum);
  }
  printf("  %-58s", "algorithmic MB per launch");
  for (auto &L : L7) printf(" %10.2f", L.bytes() / 1e6);
  printf("\n");

  // ---- per layer: the four launches in sequence, 32 layers in one graph ---------------------------------------------------------
  printf("\nus per decoder layer (graph of %d x 4 launches in model order):\n", copies);
  double layer_bytes = 0;
  for (auto &L : L7) layer_bytes += L.bytes();
  for (size_t vi = 0; vi < vs.size(); ++vi) {
    if (vs[vi].env) setenv("QLLM_STRIP1", vs[vi].env, 1);
    std::vector<std::pair<const Launch *, int>> seq;
    for (int c = 0; c < copies; ++c)
      for (auto &L : L7) seq.push_back({&L, c});
    const float us = time_graph(seq, vs[vi].fn, st, replays) / copies;
    printf("  %-58s %7.2f us per layer   %6.0f tok/s at 32 layers   %.3f of 8 TB/s\n", vs[vi].name.c_str(), us, 1e6 / (us * 32), layer_bytes / us / 1e6 / 8.0);
    fflush(stdout);
  }

  // ---- Llama-2-70B TP = 8 shard shapes through the C ABI ----------------------------------------------------------------------------
  if (tp) {
    std::vector<Launch> L70;
    L70.push_back(make_launch("q/k/v 8192->1024+128+128", 8192, {1024, 128, 128}, 80, 500));
    L70.push_back(make_launch("o 1024->8192", 1024, {8192}, 80, 600));
    L70.push_back(make_launch("gate/up 8192->2x3584", 8192, {3584, 3584}, 80, 700));
    L70.push_back(make_launch("down 3584->8192", 3584, {8192}, 80, 800));
    struct V2 { const char *name; const char *s1; const char *t256; } v2[] = {
        {"general strip kernel (QLLM_STRIP1=0)", "0", "0"}, {"batch-1 kernel, K=8192 as 8 x 32", "1", "0"}, {"batch-1 kernel, K=8192 as 16 x 16", "1", "1"}};
    printf("\nLlama-2-70B TP = 8 shard shapes, us per launch / per layer (80 rotating copies):\n");
    std::vector<std::vector<_Float16>> ref70;
    for (auto &v : v2) {
      setenv("QLLM_STRIP1", v.s1, 1);
      setenv("QLLM_S1_T256_NW16", v.t256, 1);
      printf("  %-42s", v.name);
      float sum = 0;
      for (size_t li = 0; li < L70.size(); ++li) {
        std::vector<std::pair<const Launch *, int>> seq;
        for (int c = 0; c < 80; ++c) seq.push_back({&L70[li], c});
        const float us = time_graph(seq, launch_capi, st, replays) / 80;
        sum += us;
        printf(" %8.2f", us);
        launch_capi(L70[li], 0, st);
        CK(hipStreamSynchronize(st));
        if (ref70.size() <= li) ref70.push_back(fetch(L70[li]));
        else printf("(d=%.1e)", max_rel(ref70[li], fetch(L70[li])));
        fflush(stdout);
      }
      std::vector<std::pair<const Launch *, int>> seq;
      for (int c = 0; c < 80; ++c)
        for (auto &L : L70) seq.push_back({&L, c});
      const float us = time_graph(seq, launch_capi, st, replays) / 80;
      printf("  sum %6.2f  layer graph %6.2f\n", sum, us);
    }
  }
  return 0;
}
