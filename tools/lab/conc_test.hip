// Do two kernels on two streams run concurrently (eager and inside a captured graph), and does a relaxed agent-scope poll on every
// XCD see a flag stored by one block of another kernel?  (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void poller(const uint32_t *flag, uint32_t *seen, uint64_t *when, uint32_t limit) {
  uint32_t spin = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
    __builtin_amdgcn_s_sleep(8);
    if (++spin > limit) break;
  }
  if (threadIdx.x == 0) { seen[blockIdx.x] = spin <= limit; when[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0; }
}
__global__ void setter(uint32_t *flag, int busy) {
  // a little work first, so that the poller is certainly resident and spinning
  uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)busy) {}
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  uint32_t *flag, *seen; uint64_t *when;
  CK(hipMalloc(&flag, 256)); CK(hipMalloc(&seen, 256 * 4)); CK(hipMalloc(&when, 256 * 8));
  hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
  hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  const uint32_t limit = 20000;  // ~ 10 ms
  auto report = [&](const char *name) {
    uint32_t h[256]; uint64_t w[256];
    CK(hipMemcpy(h, seen, sizeof h, hipMemcpyDeviceToHost)); CK(hipMemcpy(w, when, sizeof w, hipMemcpyDeviceToHost));
    int n = 0; uint64_t mx = 0, mn = ~0ull;
    for (int i = 0; i < 256; ++i) { n += h[i]; if (w[i] > mx) mx = w[i]; if (w[i] < mn) mn = w[i]; }
    printf("%-44s %3d / 256 pollers saw the flag; poll time %.1f .. %.1f us\n", name, n, mn / 100.0, mx / 100.0);
  };
  auto seq = [&](hipStream_t sa, hipStream_t sb) {
    CK(hipMemsetAsync(flag, 0, 4, sa));
    CK(hipEventRecord(fork, sa));
    CK(hipStreamWaitEvent(sb, fork, 0));
    poller<<<256, 64, 0, sb>>>(flag, seen, when, limit);      // side stream: spins
    setter<<<256, 256, 0, sa>>>(flag, 2000);                    // main stream: 20 us of work, then the flag
    CK(hipEventRecord(join, sb));
    CK(hipStreamWaitEvent(sa, join, 0));
  };
  seq(a, b); CK(hipDeviceSynchronize()); report("eager, poller on side stream:");
  // the other order: poller on the main stream after the setter was enqueued on the side stream
  CK(hipMemsetAsync(flag, 0, 4, a)); CK(hipEventRecord(fork, a)); CK(hipStreamWaitEvent(b, fork, 0));
  setter<<<256, 256, 0, b>>>(flag, 2000);
  poller<<<256, 64, 0, a>>>(flag, seen, when, limit);
  CK(hipEventRecord(join, b)); CK(hipStreamWaitEvent(a, join, 0));
  CK(hipDeviceSynchronize()); report("eager, setter enqueued first on side stream:");
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
  seq(a, b);
  CK(hipStreamEndCapture(a, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; ++i) { CK(hipGraphLaunch(ge, a)); CK(hipDeviceSynchronize()); report("graph replay, poller on captured side stream:"); }
  {  // captured with the setter FIRST (main stream), the poller second (side stream)
    hipGraph_t g2; hipGraphExec_t ge2;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeGlobal));
    CK(hipMemsetAsync(flag, 0, 4, a));
    CK(hipEventRecord(fork, a));
    CK(hipStreamWaitEvent(b, fork, 0));
    setter<<<256, 256, 0, a>>>(flag, 2000);
    poller<<<256, 64, 0, b>>>(flag, seen, when, limit);
    CK(hipEventRecord(join, b));
    CK(hipStreamWaitEvent(a, join, 0));
    CK(hipStreamEndCapture(a, &g2));
    CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int i = 0; i < 2; ++i) { CK(hipGraphLaunch(ge2, a)); CK(hipDeviceSynchronize()); report("graph replay, setter captured first:"); }
  }
  {  // explicit graph API: two kernel nodes with the memset as their only common parent
    hipGraph_t g3; hipGraphExec_t ge3;
    CK(hipGraphCreate(&g3, 0));
    hipGraphNode_t nm, np, ns;
    hipMemsetParams mp = {};
    mp.dst = flag; mp.value = 0; mp.elementSize = 4; mp.width = 1; mp.height = 1; mp.pitch = 4;
    CK(hipGraphAddMemsetNode(&nm, g3, nullptr, 0, &mp));
    uint32_t lim = limit; int busy = 2000;
    void *pargs[] = {&flag, &seen, &when, &lim};
    hipKernelNodeParams kp = {};
    kp.func = (void *)poller; kp.gridDim = dim3(256); kp.blockDim = dim3(64); kp.kernelParams = pargs;
    void *sargs[] = {&flag, &busy};
    hipKernelNodeParams ks = {};
    ks.func = (void *)setter; ks.gridDim = dim3(256); ks.blockDim = dim3(256); ks.kernelParams = sargs;
    CK(hipGraphAddKernelNode(&ns, g3, &nm, 1, &ks));
    CK(hipGraphAddKernelNode(&np, g3, &nm, 1, &kp));
    CK(hipGraphInstantiate(&ge3, g3, nullptr, nullptr, 0));
    for (int i = 0; i < 2; ++i) { CK(hipGraphLaunch(ge3, a)); CK(hipDeviceSynchronize()); report("explicit graph, two children of the memset:"); }
  }
  // same stream: must time out (sanity of the test itself)
  CK(hipMemsetAsync(flag, 0, 4, a));
  poller<<<256, 64, 0, a>>>(flag, seen, when, 2000);
  setter<<<256, 256, 0, a>>>(flag, 100);
  CK(hipDeviceSynchronize()); report("same stream (must time out):");
  return 0;
}
