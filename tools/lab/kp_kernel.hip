// kernarg-preload micro-test, kernel side: compiled twice (with and without -mllvm -amdgpu-kernarg-preload-count=12)
#include <hip/hip_runtime.h>
#ifndef KNAME
#define KNAME kp_kernel
#endif
extern "C" __global__ void KNAME(const float* x, float* y, const float* w, int n, int m, int k0, int k1) {
  // a dependent chain like the decode kernel's start: arguments -> address -> load -> use -> store
  const int i = blockIdx.x * 64 + threadIdx.x;
  float v = x[(i + k0) % n] * m + w[(i * 33 + k1) % n];
  y[i % n] = v;
}
