// Does a "touch one dword per 128-byte line" kernel leave its lines in the L2 a following strip-reader kernel hits?  And which
// XCD does block b of a launch run on?  (developer tool, round 3 prefetcher post-mortem)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

template <int S, int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void reader(const uint32_t* __restrict__ w, uint32_t* out, int R, uint32_t* xcc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t* base = w + (size_t)blockIdx.x * R * 16 + lane;
  uint32_t v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) v[s] = NT ? __builtin_nontemporal_load(base + (size_t)(wave * S + s) * 64) : base[(size_t)(wave * S + s) * 64];
  uint32_t acc = 0;
#pragma unroll
  for (int s = 0; s < S; ++s) acc ^= v[s];
  if (acc == 0x12345678u) out[0] = acc;
  if (xcc && threadIdx.x == 0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id & 0xf; }
}
// block p touches strips p, p + grid, ...: one dword per line; MODE 0: LDS-DMA, 1: plain register loads
template <int MODE, int STRIDE = 128>
__global__ __launch_bounds__(64) void toucher(const char* __restrict__ w, uint32_t strip_bytes, int strips, uint32_t* out, uint32_t* xcc) {
  __shared__ uint32_t scratch[64];
  uint32_t acc = 0;
  for (int s = blockIdx.x; s < strips; s += gridDim.x) {
    const char* base = w + (size_t)s * strip_bytes;
    for (uint32_t l0 = 0; l0 < strip_bytes / STRIDE; l0 += 64) {
      const char* a = base + (size_t)(l0 + threadIdx.x) * STRIDE;
      if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)a, (lds_void_t*)scratch, 4, 0, 0);
      else acc ^= *(const uint32_t*)a;
    }
  }
  if (MODE == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc = scratch[threadIdx.x]; }
  if (acc == 0x12345678u) out[0] = acc;
  if (xcc && threadIdx.x == 0) { uint32_t id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[blockIdx.x] = id & 0xf; }
}

int main() {
  const int R = 512, N = 12288, strips = N / 16;  // the q/k/v launch: 768 strips of 32 KB
  const size_t bytes = (size_t)R * N * 4;
  const int NB = 24;
  std::vector<uint32_t*> bufs(NB);
  for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
  uint32_t *out, *xr, *xt;
  CK(hipMalloc(&out, 4096)); CK(hipMalloc(&xr, 4096 * 4)); CK(hipMalloc(&xt, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_reader = [&](int touch_mode, bool nt, const char* name) {
    float tot = 0;
    const int IT = 40;
    for (int i = 0; i < IT + 3; ++i) {
      uint32_t* b = bufs[i % NB];
      if (touch_mode == 0) toucher<0><<<256, 64>>>((const char*)b, R * 64, strips, out, nullptr);
      if (touch_mode == 1) toucher<1><<<256, 64>>>((const char*)b, R * 64, strips, out, nullptr);
      if (touch_mode == 2) toucher<0, 64><<<256, 64>>>((const char*)b, R * 64, strips, out, nullptr);
      if (touch_mode == 3) toucher<0, 32><<<256, 64>>>((const char*)b, R * 64, strips, out, nullptr);
      if (touch_mode == 4) toucher<0, 16><<<256, 64>>>((const char*)b, R * 64, strips, out, nullptr);
      CK(hipEventRecord(e0));
      if (nt) reader<16, 8, true><<<strips, 512>>>(b, out, R, nullptr); else reader<16, 8, false><<<strips, 512>>>(b, out, R, nullptr);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 3) tot += ms;
    }
    printf("%-64s %6.2f us\n", name, tot * 1e3 / IT);
  };
  time_reader(-1, true, "reader (nt loads), HBM-cold:");
  time_reader(-1, false, "reader (plain loads), HBM-cold:");
  time_reader(0, true, "reader (nt) right after LDS-DMA toucher of the same buffer:");
  time_reader(1, true, "reader (nt) right after plain-load toucher of the same buffer:");
  time_reader(0, false, "reader (plain) right after LDS-DMA toucher:");
  time_reader(2, true, "reader (nt) after LDS-DMA toucher, one dword per 64 bytes:");
  time_reader(3, true, "reader (nt) after LDS-DMA toucher, one dword per 32 bytes:");
  time_reader(4, true, "reader (nt) after LDS-DMA toucher, one dword per 16 bytes:");
  time_reader(3, false, "reader (plain) after LDS-DMA toucher, one dword per 32 bytes:");
  // toucher alone
  {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 40; ++i) toucher<0, 32><<<256, 64>>>((const char*)bufs[i % NB], R * 64, strips, out, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-64s %6.2f us  (%.0f GB/s)\n", "LDS-DMA toucher (32-byte stride) alone (256 waves):", ms * 1e3 / 40, bytes / (ms / 40) / 1e6);
  }
  // XCD placement
  reader<16, 8, true><<<strips, 512>>>(bufs[0], out, R, xr);
  toucher<0><<<256, 64>>>((const char*)bufs[1], R * 64, strips, out, xt);
  CK(hipDeviceSynchronize());
  std::vector<uint32_t> hr(strips), ht(256);
  CK(hipMemcpy(hr.data(), xr, strips * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht.data(), xt, 256 * 4, hipMemcpyDeviceToHost));
  int okr = 0, okt = 0;
  for (int i = 0; i < strips; ++i) okr += (hr[i] == (uint32_t)(i % 8));
  for (int i = 0; i < 256; ++i) okt += (ht[i] == (uint32_t)(i % 8));
  printf("XCC id == block %% 8: reader %d / %d, toucher %d / 256;  reader blocks 0..15:", okr, strips, okt);
  for (int i = 0; i < 16; ++i) printf(" %u", hr[i]);
  printf("\n");
  return 0;
}
