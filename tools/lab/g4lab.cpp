// Prefill-kernel lab (developer tool): variants of the 256x128 prefill GEMM through the C ABI of the LAB build of the library
// (tools/lab/libqllm_lab.so: knobs are read from the environment at every call), checked against each other bit for bit and timed
// in interleaved rounds inside ONE process (graph replay over rotating weight sets, HIP events).
//     tools/lab/g4lab check            every layout x variant vs the round-3 kernel (gemm3), small + ragged-M shapes
//     tools/lab/g4lab time [M]         the three Llama-2-7B shapes at M (default 2048), variants interleaved, median of rounds
// Build: hipcc --offload-arch=gfx950 -O3 -I include -o tools/lab/g4lab tools/lab/g4lab.cpp -L tools/lab -lqllm_lab -Wl,-rpath,'$ORIGIN'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
  if (i < n) {
    float s = 0;
    for (int j = 0; j < 4; ++j) s += (hash32((uint32_t)i * 4 + j + seed) & 0xffff) / 65536.f - 0.5f;
    p[i] = (_Float16)(s * 1.732f);
  }
}

enum Kind { GPTQ, GPTQ_SYM, HQQ, AWQ, NATIVE, NATIVE_F16Z, NATIVE_SYM, KINDS };
static const char *kind_name[] = {"gptq", "gptq-sym", "hqq", "awq", "native", "native-f16z", "native-sym"};

struct Layer {
  qllm_weight_t w;
  std::vector<void *> owned;
};

static Layer make_layer(Kind kind, int K, int N, int group, int bits, uint32_t seed) {
  Layer L;
  const size_t G = K / group;
  const bool awq = kind == AWQ, f16z = kind == HQQ || kind == NATIVE_F16Z, sym = kind == GPTQ_SYM || kind == NATIVE_SYM;
  const size_t qw = awq ? (size_t)K * N / 8 : (size_t)K * bits / 32 * N;
  const size_t zwords = f16z ? G * N / 2 : G * ((size_t)N * bits / 32);
  uint32_t *w, *z = nullptr;
  _Float16 *s;
  CK(hipMalloc(&w, qw * 4)); CK(hipMalloc(&s, G * N * 2));
  fill_words<<<(qw + 255) / 256, 256>>>(w, qw, seed);
  if (!sym) {
    CK(hipMalloc(&z, zwords * 4));
    if (f16z) fill_scales<<<(G * N + 255) / 256, 256>>>((_Float16 *)z, G * N, seed ^ 0x9e3779b9u, bits == 3 ? 3.5f : 7.5f);
    else fill_words<<<(zwords + 255) / 256, 256>>>(z, zwords, seed ^ 0x9e3779b9u);
  }
  fill_scales<<<(G * N + 255) / 256, 256>>>(s, G * N, seed ^ 0x1234567u, 1.f / (sqrtf((float)K) * 6.5f));
  const int layout = awq ? QLLM_LAYOUT_AWQ_GEMM : (kind == HQQ || kind == NATIVE_F16Z ? QLLM_LAYOUT_HQQ : QLLM_LAYOUT_GPTQ);
  qllm_weight_t g{w, s, z, nullptr, nullptr, K, N, group, bits, layout, 0};
  CK(hipDeviceSynchronize());
  if (kind < NATIVE) {
    L.w = g;
    L.owned = {w, s, z};
    return L;
  }
  size_t bw, bs, bz;
  QK(qllm_native_sizes(&g, &bw, &bs, &bz));
  void *nw, *ns, *nz = nullptr;
  CK(hipMalloc(&nw, bw)); CK(hipMalloc(&ns, bs));
  if (bz) CK(hipMalloc(&nz, bz));
  QK(qllm_repack_native(&g, nw, ns, nz, nullptr));
  CK(hipDeviceSynchronize());
  CK(hipFree(w)); CK(hipFree(s));
  if (z) CK(hipFree(z));
  L.w = qllm_weight_t{nw, ns, nz, nullptr, nullptr, K, N, group, bits, kind == NATIVE_F16Z ? QLLM_LAYOUT_NATIVE_F16Z : QLLM_LAYOUT_NATIVE, 0};
  L.owned = {nw, ns, nz};
  return L;
}
static void free_layer(Layer &L) {
  for (void *p : L.owned) if (p) CK(hipFree(p));
}

struct Variant { const char *name; const char *g4; const char *mw; const char *pm, *pd, *pl; const char *abl; const char *dw = "4"; const char *g5 = "0"; const char *abl5 = "0"; };
static const Variant variants[] = {
    {"gemm3 mw8", "-1", "8", "2", "0", "0", "0"},
    {"gemm5 2x4 waves (128x32)", "-1", "8", "2", "0", "0", "0", "4", "2"},   // round 5: register-B-fragment kernel (gemm5.hip)
    {"gemm5 1x4 waves (256x32)", "-1", "8", "2", "0", "0", "0", "4", "1"},
    // timing-only ablations of the 1x4 form (results are garbage; check() stops in front of them)
    {"g5 1x4 - A DMA", "-1", "8", "2", "0", "0", "0", "4", "1", "1"},
    {"g5 1x4 - dequant ALU", "-1", "8", "2", "0", "0", "0", "4", "1", "2"},
    {"g5 1x4 - A frag reads", "-1", "8", "2", "0", "0", "0", "4", "1", "4"},
    {"g5 1x4 - word loads", "-1", "8", "2", "0", "0", "0", "4", "1", "8"},
    {"g5 1x4 - DMA - ALU", "-1", "8", "2", "0", "0", "0", "4", "1", "3"},
    {"g5 1x4 - ALU - loads", "-1", "8", "2", "0", "0", "0", "4", "1", "10"},
    {"g5 1x4 - DMA - ALU - loads", "-1", "8", "2", "0", "0", "0", "4", "1", "11"},
    {"g5 1x4 MFMA only", "-1", "8", "2", "0", "0", "0", "4", "1", "15"},
};
static const int NV_CHECK = 3;
static const int NV = sizeof(variants) / sizeof(variants[0]);
static void select_variant(const Variant &v) {
  setenv("QLLM_GEMM4", v.g4, 1);
  setenv("QLLM_GEMM3_MW", v.mw, 1);
  setenv("QLLM_GEMM3_DW", v.dw, 1);
  setenv("QLLM_G4_PRIO_M", v.pm, 1);
  setenv("QLLM_G4_PRIO_D", v.pd, 1);
  setenv("QLLM_G4_PRIO_L", v.pl, 1);
  setenv("QLLM_G4_ABLATE", v.abl, 1);
  setenv("QLLM_GEMM5", v.g5, 1);
  setenv("QLLM_G5_ABL", v.abl5, 1);
}
static bool distinct_kernel(int v) { return v >= 1; }  // (check(): priorities do not change results)

static int check() {
  setenv("QLLM_GEMM3_MIN_M", "65", 1);
  setenv("QLLM_GEMM3_MIN_M_3BIT", "65", 1);
  setenv("QLLM_GEMM2", "1", 1);
  setenv("QLLM_GEMM2_SPLITK", "0", 1);  // unsplit launches: the form gemm5 serves (small shapes would otherwise split K and stay on gemm3)
  void *ws;
  const size_t ws_bytes = 256 << 20;
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  struct Case { int M, K, N, group, bits; } cases[] = {
      {300, 512, 256, 128, 4}, {2048, 4096, 512, 128, 4}, {257, 1024, 384, 64, 4}, {1000, 2048, 1024, 32, 4}, {513, 11008, 256, 128, 4},
      {300, 512, 256, 64, 3}, {1024, 4096, 512, 128, 3}, {130, 1024, 128, 32, 3}, {2048, 1024, 4096, 128, 4}, {128, 8192, 4096, 128, 4}};
  int bad = 0;
  for (auto c : cases) {
    for (int k = 0; k < KINDS; ++k) {
      if (c.bits == 3 && (k == AWQ)) continue;
      Layer L = make_layer((Kind)k, c.K, c.N, c.group, c.bits, 77 + k);
      _Float16 *x, *y0, *y1;
      const size_t ny = (size_t)c.M * c.N;
      CK(hipMalloc(&x, (size_t)c.M * c.K * 2)); CK(hipMalloc(&y0, ny * 2)); CK(hipMalloc(&y1, ny * 2));
      fill_x<<<((size_t)c.M * c.K + 255) / 256, 256>>>(x, (size_t)c.M * c.K, 7);
      select_variant(variants[0]);
      QK(qllm_linear_forward(&L.w, x, y0, c.M, QLLM_F16, ws, ws_bytes, nullptr));
      CK(hipDeviceSynchronize());
      std::vector<uint16_t> h0(ny), h1(ny);
      CK(hipMemcpy(h0.data(), y0, ny * 2, hipMemcpyDeviceToHost));
      char plan[200];
      QK(qllm_plan_describe(&L.w, 1, c.M, 1, plan, sizeof plan));
      for (int v = 1; v < NV_CHECK; ++v) {
        if (!distinct_kernel(v)) continue;
        select_variant(variants[v]);
        CK(hipMemset(y1, 0xff, ny * 2));
        QK(qllm_linear_forward(&L.w, x, y1, c.M, QLLM_F16, ws, ws_bytes, nullptr));
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("FAULT %s at variant %s\n", hipGetErrorString(e), variants[v].name); return 2; }
        CK(hipMemcpy(h1.data(), y1, ny * 2, hipMemcpyDeviceToHost));
        size_t diff = 0, first = 0;
        for (size_t i = 0; i < ny; ++i) if (h0[i] != h1[i]) { if (!diff) first = i; ++diff; }
        printf("%-11s w%d g%-3d M=%4d K=%5d N=%4d  %-24s %s", kind_name[k], c.bits, c.group, c.M, c.K, c.N, variants[v].name, diff ? "MISMATCH" : "bit-exact");
        if (diff) printf("  %zu of %zu differ, first at row %zu col %zu (%04x vs %04x)", diff, ny, first / c.N, first % c.N, h0[first], h1[first]);
        printf("   [%s]\n", plan);
        bad += diff != 0;
      }
      CK(hipFree(x)); CK(hipFree(y0)); CK(hipFree(y1));
      free_layer(L);
    }
  }
  printf(bad ? "CHECK FAILED: %d mismatching runs\n" : "CHECK OK (%d)\n", bad);
  return bad != 0;
}

static int timing(int M, Kind kind) {
  struct Shape { const char *name; int K, N; } shapes[] = {{"4096x4096", 4096, 4096}, {"4096x11008", 4096, 11008}, {"11008x4096", 11008, 4096}};
  const int NSET = 12, ROUNDS = 7;
  void *ws;
  const size_t ws_bytes = 256 << 20;
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  printf("M=%d layout=%s: us per linear, median of %d interleaved rounds (TFLOP/s) [min]\n", M, kind_name[kind], ROUNDS);
  for (auto sh : shapes) {
    std::vector<Layer> sets;
    for (int i = 0; i < NSET; ++i) sets.push_back(make_layer(kind, sh.K, sh.N, 128, 4, 100 * i + sh.K));
    _Float16 *x, *y;
    CK(hipMalloc(&x, (size_t)M * sh.K * 2)); CK(hipMalloc(&y, (size_t)M * sh.N * 2));
    fill_x<<<((size_t)M * sh.K + 255) / 256, 256>>>(x, (size_t)M * sh.K, 7);
    CK(hipDeviceSynchronize());
    std::vector<hipGraphExec_t> ge(NV);
    for (int v = 0; v < NV; ++v) {
      select_variant(variants[v]);
      auto run = [&]() { for (int i = 0; i < NSET; ++i) QK(qllm_linear_forward(&sets[i].w, x, y, M, QLLM_F16, ws, ws_bytes, st)); };
      run();
      CK(hipStreamSynchronize(st));
      hipGraph_t g;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      run();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge[v], g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge[v], st));
      CK(hipStreamSynchronize(st));
    }
    std::vector<std::vector<double>> t(NV);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < ROUNDS; ++r)
      for (int v = 0; v < NV; ++v) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge[v], st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t[v].push_back(ms * 1e3 / 3 / NSET);
      }
    for (int v = 0; v < NV; ++v) {
      std::sort(t[v].begin(), t[v].end());
      const double med = t[v][ROUNDS / 2], mn = t[v][0];
      printf("  %-11s %-24s %8.2f us  %7.1f TFLOP/s   [min %8.2f us %7.1f]\n", sh.name, variants[v].name, med, 2.0 * M * sh.K * sh.N / med / 1e6, mn,
             2.0 * M * sh.K * sh.N / mn / 1e6);
    }
    fflush(stdout);
    for (auto &l : sets) free_layer(l);
    CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}

// one variant, one shape, a handful of eager launches: the target of rocprofv3 --pmc passes
static int prof(int v, int M, int K, int N) {
  void *ws;
  const size_t ws_bytes = 256 << 20;
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  std::vector<Layer> sets;
  for (int i = 0; i < 6; ++i) sets.push_back(make_layer(NATIVE, K, N, 128, 4, 100 * i + K));
  _Float16 *x, *y;
  CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&y, (size_t)M * N * 2));
  fill_x<<<((size_t)M * K + 255) / 256, 256>>>(x, (size_t)M * K, 7);
  select_variant(variants[v]);
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < 6; ++i) QK(qllm_linear_forward(&sets[i].w, x, y, M, QLLM_F16, ws, ws_bytes, nullptr));
  CK(hipDeviceSynchronize());
  printf("prof: variant %s M=%d K=%d N=%d: 18 launches\n", variants[v].name, M, K, N);
  return 0;
}

// in-kernel timeline of ONE launch of a gemm4 variant (lab build stamps, 16 x u64 per block)
static int timeline(int v, int M, int K, int N) {
  void *ws;
  const size_t ws_bytes = 256 << 20;
  CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
  std::vector<Layer> sets;
  for (int i = 0; i < 4; ++i) sets.push_back(make_layer(NATIVE, K, N, 128, 4, 100 * i + K));
  _Float16 *x, *y;
  CK(hipMalloc(&x, (size_t)M * K * 2)); CK(hipMalloc(&y, (size_t)M * N * 2));
  fill_x<<<((size_t)M * K + 255) / 256, 256>>>(x, (size_t)M * K, 7);
  select_variant(variants[v]);
  const int nblk = ((M + 255) / 256) * (N / 128);
  uint64_t *buf;
  CK(hipMalloc(&buf, (size_t)nblk * 16 * 8));
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 3; ++i) QK(qllm_linear_forward(&sets[i].w, x, y, M, QLLM_F16, ws, ws_bytes, nullptr));  // warm, untimed
    CK(hipMemset(buf, 0, (size_t)nblk * 16 * 8));
    CK(hipDeviceSynchronize());
    QK(qllm_debug_timeline(buf, 1));
    QK(qllm_linear_forward(&sets[3].w, x, y, M, QLLM_F16, ws, ws_bytes, nullptr));
    QK(qllm_debug_timeline(nullptr, 0));
    CK(hipDeviceSynchronize());
    std::vector<uint64_t> h((size_t)nblk * 16);
    CK(hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost));
    auto stat = [&](const char *name, auto f) {
      std::vector<double> vals;
      for (int b = 0; b < nblk; ++b) vals.push_back(f(&h[(size_t)b * 16]));
      std::sort(vals.begin(), vals.end());
      printf("    %-46s min %9.0f  p10 %9.0f  median %9.0f  p90 %9.0f  max %9.0f\n", name, vals[0], vals[nblk / 10], vals[nblk / 2], vals[nblk * 9 / 10], vals[nblk - 1]);
    };
    uint64_t r0 = ~0ull, r1 = 0;
    for (int b = 0; b < nblk; ++b) { r0 = std::min(r0, h[(size_t)b * 16]); r1 = std::max(r1, h[(size_t)b * 16 + 6]); }
    printf("timeline %s M=%d K=%d N=%d rep %d: %d blocks, first entry -> last exit %.2f us (100 MHz clock)\n", variants[v].name, M, K, N, rep, nblk, (r1 - r0) / 100.0);
    stat("entry after first block's entry [us]", [&](const uint64_t *d) { return (d[0] - r0) / 100.0; });
    stat("exit after first block's entry [us]", [&](const uint64_t *d) { return (d[6] - r0) / 100.0; });
    stat("block lifetime [us]", [&](const uint64_t *d) { return (d[6] - d[0]) / 100.0; });
    stat("block lifetime [cycles]", [&](const uint64_t *d) { return (double)(d[5] - d[1]); });
    stat("matrix: entry -> prologue barrier passed", [&](const uint64_t *d) { return (double)(d[2] - d[1]); });
    stat("matrix: main loop (64 k-tiles x 1024 MFMA cyc)", [&](const uint64_t *d) { return (double)(d[3] - d[2]); });
    stat("matrix: final barrier", [&](const uint64_t *d) { return (double)(d[4] - d[3]); });
    stat("matrix: epilogue", [&](const uint64_t *d) { return (double)(d[5] - d[4]); });
    stat("matrix: cycles inside k-tile barriers (sum)", [&](const uint64_t *d) { return (double)d[7]; });
    stat("dequant: busy cycles in the loop (sum)", [&](const uint64_t *d) { return (double)d[8 + 4]; });
    stat("dequant:   of it requesting (DMA issue)", [&](const uint64_t *d) { return (double)d[8 + 0]; });
    stat("dequant:   of it waiting for the raw tile", [&](const uint64_t *d) { return (double)d[8 + 1]; });
    stat("dequant:   of it LDS reads + ALU + stores", [&](const uint64_t *d) { return (double)d[8 + 5]; });
    stat("loader: busy cycles in the loop (sum)", [&](const uint64_t *d) { return (double)d[14]; });
  }
  return 0;
}

int main(int argc, char **argv) {
  if (argc >= 2 && !strcmp(argv[1], "check")) return check();
  if (argc >= 3 && !strcmp(argv[1], "timeline")) return timeline(atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 2048, argc >= 5 ? atoi(argv[4]) : 4096, argc >= 6 ? atoi(argv[5]) : 4096);
  if (argc >= 3 && !strcmp(argv[1], "prof")) return prof(atoi(argv[2]), argc >= 4 ? atoi(argv[3]) : 2048, argc >= 5 ? atoi(argv[4]) : 4096, argc >= 6 ? atoi(argv[5]) : 4096);
  if (argc >= 2 && !strcmp(argv[1], "time")) {
    const int M = argc >= 3 ? atoi(argv[2]) : 2048;
    Kind kind = NATIVE;
    if (argc >= 4) for (int k = 0; k < KINDS; ++k) if (!strcmp(argv[3], kind_name[k])) kind = (Kind)k;
    return timing(M, kind);
  }
  printf("usage: g4lab check | time [M] [layout]\n");
  return 1;
}
