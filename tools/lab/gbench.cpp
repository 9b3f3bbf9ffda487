// Per-shape, per-M timing of qllm_linear_forward through the C ABI without Python / torch (developer tool): the three Llama-2-7B
// linears, native strip-major layout (and, with --layout gptq, the reference row stream in place), M from the decode / prefill
// boundary (33) to 2048, hipGraph replay over rotating weight sets (HBM-resident).  The env knobs of the dispatcher are latched at
// first use, so variants are separate invocations:
//     tools/lab/gbench                                     default dispatch
//     QLLM_GEMM3_MIN_M=65 tools/lab/gbench                 the wave-specialised kernel for every M > 64
//     QLLM_STRIP_MAX_M=32 QLLM_GEMM2_MIN_M=33 tools/lab/gbench --m 33 48 64      gemm2 instead of the 4-row-tile strips
//     tools/lab/gbench --cfg3 --bits 3 --m 16              BASELINE configs[3]: HQQ g64 layer, the four launches one by one
// Build: hipcc --offload-arch=gfx950 -O3 -I include -o tools/lab/gbench tools/lab/gbench.cpp -L qllm_amd -lqllm_mi355x -Wl,-rpath,'$ORIGIN/../../qllm_amd'
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

// hqq: HQQ row stream (fp16 zero points [G][N]); else GPTQ (packed zero points)
static qllm_weight_t make_layer(int K, int N, int group, uint32_t seed, bool native, int bits = 4, bool hqq = false) {
  const size_t qw = (size_t)K * bits / 32 * N, G = K / group;
  const size_t zwords = hqq ? G * N / 2 : G * ((size_t)N * bits / 32);
  uint32_t *w, *z;
  _Float16 *s;
  CK(hipMalloc(&w, qw * 4)); CK(hipMalloc(&z, zwords * 4)); CK(hipMalloc(&s, G * N * 2));
  fill_words<<<(qw + 255) / 256, 256>>>(w, qw, seed);
  if (hqq) fill_scales<<<(G * N + 255) / 256, 256>>>((_Float16 *)z, G * N, seed ^ 0x9e3779b9u, 7.5f);
  else fill_words<<<(zwords + 255) / 256, 256>>>(z, zwords, seed ^ 0x9e3779b9u);
  fill_scales<<<(G * N + 255) / 256, 256>>>(s, G * N, seed ^ 0x1234567u, 1.f / (sqrtf((float)K) * 6.5f));
  qllm_weight_t g{w, s, z, nullptr, nullptr, K, N, group, bits, hqq ? QLLM_LAYOUT_HQQ : QLLM_LAYOUT_GPTQ, 0};
  if (!native) return g;
  size_t bw, bs, bz;
  QK(qllm_native_sizes(&g, &bw, &bs, &bz));
  void *nw, *ns, *nz;
  CK(hipMalloc(&nw, bw)); CK(hipMalloc(&ns, bs)); CK(hipMalloc(&nz, bz));
  QK(qllm_repack_native(&g, nw, ns, nz, nullptr));
  CK(hipDeviceSynchronize());
  CK(hipFree(w)); CK(hipFree(z)); CK(hipFree(s));
  return qllm_weight_t{nw, ns, nz, nullptr, nullptr, K, N, group, bits, hqq ? QLLM_LAYOUT_NATIVE_F16Z : QLLM_LAYOUT_NATIVE, 0};
}

int main(int argc, char **argv) {
  std::vector<int> ms = {33, 48, 64, 65, 96, 128, 192, 256, 384, 512, 768, 1024, 2048};
  std::string layout = "native";
  int bf16 = 0, cfg3 = 0, bits = 4, group = 128, hqq = 0;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--m")) { ms.clear(); while (i + 1 < argc && argv[i + 1][0] != '-') ms.push_back(atoi(argv[++i])); }
    else if (!strcmp(argv[i], "--layout")) layout = argv[++i];
    else if (!strcmp(argv[i], "--bf16")) bf16 = 1;
    else if (!strcmp(argv[i], "--cfg3")) { cfg3 = 1; hqq = 1; group = 64; }
    else if (!strcmp(argv[i], "--bits")) bits = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--group")) group = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--gptq")) hqq = 0;
  }
  const bool native = layout == "native";
  if (cfg3) {
    // BASELINE configs[3]-style decoder layer: HQQ g64 fp16 zero points, the four launches of a layer (q/k/v and gate/up grouped),
    // NSET rotating layers, one hipGraph per launch kind: us per launch
    void *ws;
    const size_t ws_bytes = 64 << 20;
    CK(hipMalloc(&ws, ws_bytes)); CK(hipMemset(ws, 0, ws_bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    struct L { const char *name; int K, N, n; };
    std::vector<L> launches = {{"q/k/v", 4096, 4096, 3}, {"o_proj", 4096, 4096, 1}, {"gate/up", 4096, 11008, 2}, {"down", 11008, 4096, 1}};
    // GBENCH_FUSED=1: what a sibling group would cost as ONE layer of the summed width (the strips of adjacent native copies are one array)
    if (getenv("GBENCH_FUSED")) launches = {{"qkv-as-1", 4096, 12288, 1}, {"gu-as-1", 4096, 22016, 1}};
    const int NSET = 20;  // (20 x 10..52 MB per launch kind: far beyond the 256 MB Infinity Cache)
    for (int M : ms) {
      double total = 0, bytes_total = 0;
      for (auto l : launches) {
        std::vector<qllm_weight_t> sets;
        for (int i = 0; i < NSET * l.n; ++i) sets.push_back(make_layer(l.K, l.N, group, 100 * i + l.K + l.N, native, bits, hqq));
        _Float16 *x; void *ys[8];
        CK(hipMalloc(&x, (size_t)M * l.K * 2));
        for (int j = 0; j < l.n; ++j) CK(hipMalloc(&ys[j], (size_t)M * l.N * 2));
        fill_x<<<((size_t)M * l.K + 255) / 256, 256>>>(x, (size_t)M * l.K, 7);
        CK(hipDeviceSynchronize());
        auto run = [&]() { for (int i = 0; i < NSET; ++i) QK(qllm_linear_forward_grouped(&sets[i * l.n], ys, l.n, x, M, bf16 ? QLLM_BF16 : QLLM_F16, ws, ws_bytes, st)); };
        run();
        CK(hipStreamSynchronize(st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        run();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = 20;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms_;
        CK(hipEventElapsedTime(&ms_, e0, e1));
        const double us = ms_ * 1e3 / iters / NSET;
        const double G = l.K / group;
        const double bytes = l.n * ((double)l.K * l.N * bits / 8 + G * l.N * 2 * 2 + (double)M * l.N * 2) + (double)M * l.K * 2;
        char plan[200];
        QK(qllm_plan_describe(&sets[0], l.n, M, 1, plan, sizeof plan));
        printf("  %-8s M=%3d  %7.2f us  %5.2f TB/s  [%s]\n", l.name, M, us, bytes / us / 1e6, plan);
        fflush(stdout);
        total += us; bytes_total += bytes;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        CK(hipFree(x));
        for (int j = 0; j < l.n; ++j) CK(hipFree(ys[j]));
        for (auto &w : sets) { CK(hipFree((void *)w.qweight)); CK(hipFree((void *)w.scales)); CK(hipFree((void *)w.qzeros)); }
      }
      printf("  layer    M=%3d  %7.2f us  %5.2f TB/s = %.3f of 8 TB/s (w%d g%d %s)\n", M, total, bytes_total / total / 1e6, bytes_total / total / 8e6, bits, group, hqq ? "hqq" : "gptq");
    }
    return 0;
  }
  struct Shape { const char *name; int K, N; } shapes[] = {{"4096x4096", 4096, 4096}, {"4096x11008", 4096, 11008}, {"11008x4096", 11008, 4096}};
  const int NSET = 12;  // 12 x 8.4 .. 22.5 MB: beyond the L2s, rotating
  void *ws;
  const size_t ws_bytes = 256 << 20;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  printf("layout=%s act=%s; us per linear (TFLOP/s) [plan]\n", layout.c_str(), bf16 ? "bf16" : "f16");
  for (auto sh : shapes) {
    std::vector<qllm_weight_t> sets;
    for (int i = 0; i < NSET; ++i) sets.push_back(make_layer(sh.K, sh.N, 128, 100 * i + sh.K, native));
    for (int M : ms) {
      _Float16 *x, *y;
      CK(hipMalloc(&x, (size_t)M * sh.K * 2)); CK(hipMalloc(&y, (size_t)M * sh.N * 2));
      fill_x<<<((size_t)M * sh.K + 255) / 256, 256>>>(x, (size_t)M * sh.K, 7);
      CK(hipDeviceSynchronize());
      auto run = [&]() { for (int i = 0; i < NSET; ++i) QK(qllm_linear_forward(&sets[i], x, y, M, bf16 ? QLLM_BF16 : QLLM_F16, ws, ws_bytes, st)); };
      run();
      CK(hipStreamSynchronize(st));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      run();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 2; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const int iters = M >= 1024 ? 5 : 20;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms_;
      CK(hipEventElapsedTime(&ms_, e0, e1));
      const double us = ms_ * 1e3 / iters / NSET;
      char plan[160];
      QK(qllm_plan_describe(&sets[0], 1, M, 1, plan, sizeof plan));
      printf("  %-11s M=%5d  %8.2f us  %7.1f TFLOP/s  [%s]\n", sh.name, M, us, 2.0 * M * sh.K * sh.N / us / 1e6, plan);
      fflush(stdout);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
      CK(hipFree(x)); CK(hipFree(y));
    }
    for (auto &w : sets) { CK(hipFree((void *)w.qweight)); CK(hipFree((void *)w.scales)); CK(hipFree((void *)w.qzeros)); }
  }
  return 0;
}
