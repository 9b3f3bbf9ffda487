// Decode-step bench of the C ABI without Python / torch (developer tool): the 32 x 4 launches of one Llama-2-7B decode token
// (q/k/v grouped, o, gate/up grouped, down), every launch fed by the previous one, captured into a hipGraph and replayed --
// the same step bench.py times through the modules.  Weights: synthetic, in the reference's GPTQ row-stream layout and/or
// repacked to the library's native strip-major layout (qllm_repack_native); both paths are checked against each other first.
//
// Build:  hipcc --offload-arch=gfx950 -O3 -I include -o tools/lab/cbench tools/lab/cbench.cpp -L qllm_amd -lqllm_mi355x -Wl,-rpath,'$ORIGIN/../../qllm_amd'
// Run:    tools/lab/cbench [--layers 32] [--iters 50] [--layout native|gptq|both] [--timeline] [--m 1]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <initializer_list>
#include <string>
#include <vector>

#include "qllm_mi355x.h"

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
  if (i < n) {  // roughly N(0,1): sum of 4 uniforms
    float s = 0;
    for (int j = 0; j < 4; ++j) s += (hash32((uint32_t)i * 4 + j + seed) & 0xffff) / 65536.f - 0.5f;
    p[i] = (_Float16)(s * 1.732f);
  }
}

// --prefetch experiment (round 5): a side stream reads the packed words of the launch `dist` ahead while the current launch runs, so
// that HBM keeps streaming through the launch boundaries and the next launch finds its words in the Infinity Cache.  Plain
// loads, nothing stored (the never-true store keeps the loads alive).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void prefetch_kernel(const u32x4 *p, size_t n16, uint32_t *sink) {
  uint32_t acc = 0;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
  for (size_t i = tid; i < n16; i += nth * 8) {  // eight independent 16-byte loads per thread in flight
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t idx = i + u * nth;
      v[u] = idx < n16 ? __builtin_nontemporal_load(p + idx) : u32x4{0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x9e3779b9u && sink) *sink = acc;
}

struct Layer {
  qllm_weight_t g{}, n{};  // GPTQ row-stream descriptor, native descriptor
};

static Layer make_layer(int K, int N, int group, uint32_t seed, bool keep_gptq, bool want_native) {
  Layer L;
  const size_t qw = (size_t)K / 8 * N, G = K / group;
  uint32_t *w, *z;
  _Float16 *s;
  CK(hipMalloc(&w, qw * 4));
  CK(hipMalloc(&z, G * (N / 8) * 4));
  CK(hipMalloc(&s, G * N * 2));
  fill_words<<<(qw + 255) / 256, 256>>>(w, qw, seed);
  fill_words<<<(G * (N / 8) + 255) / 256, 256>>>(z, G * (N / 8), seed ^ 0x9e3779b9u);
  fill_scales<<<(G * N + 255) / 256, 256>>>(s, G * N, seed ^ 0x1234567u, 1.f / (sqrtf((float)K) * 6.5f));
  L.g = qllm_weight_t{w, s, z, nullptr, nullptr, K, N, group, 4, QLLM_LAYOUT_GPTQ, 0};
  if (want_native) {
    size_t bw, bs, bz;
    QK(qllm_native_sizes(&L.g, &bw, &bs, &bz));
    void *nw, *ns, *nz;
    CK(hipMalloc(&nw, bw)); CK(hipMalloc(&ns, bs)); CK(hipMalloc(&nz, bz));
    QK(qllm_repack_native(&L.g, nw, ns, nz, nullptr));
    L.n = qllm_weight_t{nw, ns, nz, nullptr, nullptr, K, N, group, 4, QLLM_LAYOUT_NATIVE, 0};
    CK(hipDeviceSynchronize());
    if (!keep_gptq) { CK(hipFree(w)); CK(hipFree(z)); CK(hipFree(s)); L.g.qweight = nullptr; }
  }
  return L;
}

int main(int argc, char **argv) {
  int layers = 32, iters = 50, M = 1;
  int pf_dist = 0, pf_blocks = 256, pf_threads = 256;  // --prefetch <launches ahead> [--pf-blocks n] [--pf-threads n]
  std::string layout = "native";
  bool timeline = false;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--layers")) layers = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--iters")) iters = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--layout")) layout = argv[++i];
    else if (!strcmp(argv[i], "--m")) M = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--timeline")) timeline = true;
    else if (!strcmp(argv[i], "--prefetch")) pf_dist = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pf-blocks")) pf_blocks = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--pf-threads")) pf_threads = atoi(argv[++i]);
  }
  qllm_device_info_t info;
  QK(qllm_device_info(0, &info));
  printf("device %s, %d CUs; layers=%d M=%d layout=%s prefetch=%d launches ahead (%d x %d threads)\n", info.arch, info.compute_units, layers, M, layout.c_str(),
         pf_dist, pf_blocks, pf_threads);
  const int H = 4096, I = 11008, GS = 128;
  const bool both = layout == "both", want_native = both || layout == "native", want_gptq = both || layout == "gptq";
  struct Block { Layer q, k, v, o, gate, up, down; };
  std::vector<Block> blocks(layers);
  for (int l = 0; l < layers; ++l) {
    const uint32_t sd = 1000 * l;
    const bool keep = want_gptq || l == 0;
    blocks[l].q = make_layer(H, H, GS, sd + 1, keep, want_native);
    blocks[l].k = make_layer(H, H, GS, sd + 2, keep, want_native);
    blocks[l].v = make_layer(H, H, GS, sd + 3, keep, want_native);
    blocks[l].o = make_layer(H, H, GS, sd + 4, keep, want_native);
    blocks[l].gate = make_layer(H, I, GS, sd + 5, keep, want_native);
    blocks[l].up = make_layer(H, I, GS, sd + 6, keep, want_native);
    blocks[l].down = make_layer(I, H, GS, sd + 7, keep, want_native);
  }
  _Float16 *h0, *q, *k, *v, *o, *gate, *up, *h1, *ref;
  CK(hipMalloc(&h0, (size_t)M * H * 2)); CK(hipMalloc(&q, (size_t)M * H * 2)); CK(hipMalloc(&k, (size_t)M * H * 2)); CK(hipMalloc(&v, (size_t)M * H * 2));
  CK(hipMalloc(&o, (size_t)M * H * 2)); CK(hipMalloc(&gate, (size_t)M * I * 2)); CK(hipMalloc(&up, (size_t)M * I * 2)); CK(hipMalloc(&h1, (size_t)M * H * 2));
  CK(hipMalloc(&ref, (size_t)M * I * 2));
  fill_x<<<(M * H + 255) / 256, 256>>>(h0, (size_t)M * H, 77);
  void *ws;
  const size_t ws_bytes = 64 << 20;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  CK(hipDeviceSynchronize());

  auto pick = [&](const Layer &L, bool native) -> const qllm_weight_t & { return native ? L.n : L.g; };
  // launch j of the step (4 per layer): the layers whose words it streams
  auto launch_layers = [&](int j) {
    Block &b = blocks[(j / 4) % layers];
    std::vector<const Layer *> v;
    switch (j % 4) {
      case 0: v = {&b.q, &b.k, &b.v}; break;
      case 1: v = {&b.o}; break;
      case 2: v = {&b.gate, &b.up}; break;
      default: v = {&b.down};
    }
    return v;
  };
  hipStream_t st2 = nullptr;
  uint32_t *sink = nullptr;
  std::vector<hipEvent_t> evs;
  if (pf_dist > 0) {
    CK(hipStreamCreate(&st2));
    CK(hipMalloc(&sink, 4));
    evs.resize(4 * layers + 1);
    for (auto &e : evs) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  auto step = [&](bool native, hipStream_t st) {
    const _Float16 *x = h0;
    int j = 0;
    auto before = [&]() {  // in front of launch j: the side stream may start on launch j + dist once launch j - 1 is done
      if (pf_dist > 0 && native) {
        CK(hipEventRecord(evs[j], st));
        CK(hipStreamWaitEvent(st2, evs[j], 0));
        for (const Layer *L : launch_layers((j + pf_dist) % (4 * layers))) {
          const size_t n16 = (size_t)L->n.K * L->n.N / 2 / 16;
          prefetch_kernel<<<pf_blocks, pf_threads, 0, st2>>>((const u32x4 *)L->n.qweight, n16, sink);
        }
      }
      ++j;
    };
    for (int l = 0; l < layers; ++l) {
      Block &b = blocks[l];
      qllm_weight_t w3[3] = {pick(b.q, native), pick(b.k, native), pick(b.v, native)};
      void *y3[3] = {q, k, v};
      before();
      QK(qllm_linear_forward_grouped(w3, y3, 3, x, M, QLLM_F16, ws, ws_bytes, st));
      before();
      QK(qllm_linear_forward(&pick(b.o, native), q, o, M, QLLM_F16, ws, ws_bytes, st));
      qllm_weight_t w2[2] = {pick(b.gate, native), pick(b.up, native)};
      void *y2[2] = {gate, up};
      before();
      QK(qllm_linear_forward_grouped(w2, y2, 2, o, M, QLLM_F16, ws, ws_bytes, st));
      before();
      QK(qllm_linear_forward(&pick(b.down, native), gate, h1, M, QLLM_F16, ws, ws_bytes, st));
      x = h1;
    }
    if (pf_dist > 0 && native) {  // join the side stream (a capture must end with one)
      CK(hipEventRecord(evs[4 * layers], st2));
      CK(hipStreamWaitEvent(st, evs[4 * layers], 0));
    }
  };

  // ---- parity of the two layouts on layer 0's launches (same integers, same x) ----------------------------------------------
  if (want_native) {
    auto cmp = [&](const char *name, const Layer &L, const _Float16 *x, int N) {
      std::vector<_Float16> a((size_t)M * N), b((size_t)M * N);
      QK(qllm_linear_forward(&L.g, x, ref, M, QLLM_F16, ws, ws_bytes, nullptr));
      CK(hipMemcpy(a.data(), ref, a.size() * 2, hipMemcpyDeviceToHost));
      QK(qllm_linear_forward(&L.n, x, ref, M, QLLM_F16, ws, ws_bytes, nullptr));
      CK(hipMemcpy(b.data(), ref, b.size() * 2, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - (double)b[i])); mx = fmax(mx, fabs((double)a[i])); }
      char d1[128], d2[128];
      QK(qllm_plan_describe(&L.g, 1, M, 1, d1, sizeof d1));
      QK(qllm_plan_describe(&L.n, 1, M, 1, d2, sizeof d2));
      printf("parity %-5s max|y_rowstream - y_native| = %.3g (max|y| %.3g)  %s  [%s] vs [%s]\n", name, md, mx, md <= 2e-3 * mx ? "OK" : "MISMATCH", d1, d2);
      return md <= 2e-3 * mx;
    };
    bool ok = cmp("q", blocks[0].q, h0, H);
    ok = cmp("gate", blocks[0].gate, h0, I) && ok;
    fill_x<<<(M * I + 255) / 256, 256>>>(gate, (size_t)M * I, 99);
    ok = cmp("down", blocks[0].down, gate, H) && ok;
    if (!ok) return 2;
  }

  hipStream_t st;
  CK(hipStreamCreate(&st));
  const double bytes_layer = 4.0 * 8732672 + 2.0 * 23455232 + 23455232;  // BASELINE.md, M = 1
  for (int native = 0; native < 2; ++native) {
    if ((native && !want_native) || (!native && !want_gptq)) continue;
    if (timeline && native) {
      const int slots = 4 * 4;  // four layers' launches
      uint64_t *tl;
      CK(hipMalloc(&tl, slots * 24 * 8));
      CK(hipMemset(tl, 0, slots * 24 * 8));
      const int keep_layers = layers;
      for (int rep = 0; rep < 2; ++rep) {  // second pass: warm instruction caches
        QK(qllm_debug_timeline(tl, slots));
        layers = 4 < keep_layers ? 4 : keep_layers;
        step(true, st);
        CK(hipStreamSynchronize(st));
      }
      layers = keep_layers;
      QK(qllm_debug_timeline(nullptr, 0));
      std::vector<uint64_t> t(slots * 24);
      CK(hipMemcpy(t.data(), tl, t.size() * 8, hipMemcpyDeviceToHost));
      const char *names[4] = {"qkv", "o", "gate/up", "down"};
      printf("timeline (eager, us relative to the launch's first-block entry; 100 MHz clock): block: entry issued x_staged rounds_done barrier exit\n");
      for (int s = 4; s < slots; ++s) {  // skip the first layer (cold)
        const uint64_t t0 = t[s * 24];
        printf("  %-8s", names[s % 4]);
        for (int b = 0; b < 3; ++b) {
          printf(" | %s:", b == 0 ? "first" : (b == 1 ? "mid" : "last"));
          for (int e = 0; e < 6; ++e) printf(" %6.2f", ((double)t[s * 24 + b * 8 + e] - (double)t0) / 100.0);
        }
        if (s + 1 < slots) printf("  || next entry +%.2f", ((double)t[(s + 1) * 24] - (double)t0) / 100.0);
        printf("\n");
      }
    }
    // eager warm-up, capture, replay
    step(native, st);
    CK(hipStreamSynchronize(st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    step(native, st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= iters;
      best = fminf(best, ms);
      sum += ms;
    }
    const float ms = sum / 3;
    std::vector<_Float16> out((size_t)M * H);
    CK(hipMemcpy(out.data(), h1, out.size() * 2, hipMemcpyDeviceToHost));
    double amax = 0;
    bool finite = true;
    for (auto v_ : out) { amax = fmax(amax, fabs((double)v_)); finite = finite && isfinite((double)v_); }
    printf("%-7s %d layers: %.4f ms per token (best %.4f) = %.2f us per layer, %.1f tok/s at 32 layers, %.0f GB/s = %.3f of 8 TB/s   (out max %.3g, %s)\n",
           native ? "native" : "gptq", layers, ms, best, ms * 1e3 / layers, 1e3 / (ms * 32 / layers), bytes_layer * layers / ms / 1e6,
           bytes_layer * layers / ms / 1e6 / 8000.0, amax, finite ? "finite" : "NOT FINITE");
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
