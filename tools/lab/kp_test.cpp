// Does kernarg preloading (user SGPRs filled by the dispatcher instead of an s_load round trip at wave start) shorten a short
// dependent kernel on this box?  (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
extern "C" __global__ void kp_plain(const float*, float*, const float*, int, int, int, int);
extern "C" __global__ void kp_preload(const float*, float*, const float*, int, int, int, int);
int main() {
  const int n = 1 << 20;
  float *x, *y, *w;
  CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&w, n * 4));
  CK(hipMemset(x, 0, n * 4)); CK(hipMemset(w, 0, n * 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 3; ++rep)
    for (int which = 0; which < 2; ++which) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int i = 0; i < 256; ++i) {
        if (which) kp_preload<<<256, 64, 0, st>>>(x, y, w, n, 3, i, 2 * i); else kp_plain<<<256, 64, 0, st>>>(x, y, w, n, 3, i, 2 * i);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-8s %.3f us per launch\n", which ? "preload" : "plain", ms * 1e3 / (20 * 256));
    }
  return 0;
}
