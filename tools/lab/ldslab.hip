// LDS contention lab, round 4 (developer tool): do LDS-DMA writes (buffer_load ... lds), ds_read_b128 fragment reads and ds_write_b128
// tile stores share one LDS pipeline, and what does each cost the others?  One block per CU: ND waves stream 1 KB DMA pieces
// (8 rows x 128 B, L2-resident window), NR waves issue conflict-free ds_read_b128 in batches of 16, NWR waves issue ds_write_b128.
// Each role has a fixed amount of work and records its own elapsed shader cycles (s_memtime); the host prints bytes per cycle per role,
// alone and together.   Build: hipcc --offload-arch=gfx950 -O3 -o ldslab ldslab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int kWindow = 2 << 20, kRow = 8192;

template <int ND, int NR, int NWR>
__global__ __launch_bounds__((ND + NR + NWR) * 64) void k(const char* __restrict__ buf, int n_dma, int n_rd, int n_wr, uint64_t* stamps, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // [64 KB read area][ND x 4 KB DMA rings][NWR x 1 KB write areas]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  if (wave < ND) {
    const char* win = buf + (size_t)(blockIdx.x & 7) * kWindow;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)win, 0, kWindow, 0x00020000);
    const int base = (lane >> 3) * kRow + (lane & 7) * 16;
    char* mine = lds + 65536 + wave * 4096;
    for (int i = 0; i < n_dma; i += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (blockIdx.x >> 3) * 37 + wave + (i + u) * ND;
        const int so = (j & 31) * 8 * kRow + ((j >> 5) & 63) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(mine + u * 1024), 16, base, so, 0, 0);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (wave < ND + NR) {
    const char* src = lds + ((wave - ND) & 3) * 16384 + lane * 16;
    for (int i = 0; i < n_rd; i += 16) {
      u4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = *(const u4*)(src + u * 1024);
#pragma unroll
      for (int u = 0; u < 16; ++u) acc ^= v[u].x ^ v[u].w;
      asm volatile("" ::: "memory");
    }
  } else {
    char* dst = lds + 65536 + ND * 4096 + (wave - ND - NR) * 1024 + lane * 16;
    u4 v = {(uint32_t)lane, 1, 2, 3};
    for (int i = 0; i < n_wr; i += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { *(u4*)dst = v; v.x += 1; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) stamps[blockIdx.x * 32 + wave] = t1 - t0;
  if (acc == 0x12345678u) out[0] = acc;
}

template <int ND, int NR, int NWR>
void run(const char* buf, uint64_t* st_d, uint32_t* out, int n_dma, int n_rd, int n_wr, const char* what) {
  const size_t lds = 65536 + ND * 4096 + NWR * 1024;
  CK(hipFuncSetAttribute((const void*)k<ND, NR, NWR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint64_t st[256 * 32];
  double best[3] = {1e18, 1e18, 1e18};
  float ms_best = 1e9f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) {
    CK(hipEventRecord(e0));
    k<ND, NR, NWR><<<256, (ND + NR + NWR) * 64, lds>>>(buf, n_dma, n_rd, n_wr, st_d, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_best = std::min(ms_best, ms);
    CK(hipMemcpy(st, st_d, sizeof st, hipMemcpyDeviceToHost));
    double r[3] = {0, 0, 0};  // mean over blocks of the slowest wave of each role
    for (int b = 0; b < 256; ++b) {
      uint64_t m[3] = {0, 0, 0};
      for (int w = 0; w < ND + NR + NWR; ++w) { const int role = w < ND ? 0 : w < ND + NR ? 1 : 2; m[role] = std::max(m[role], st[b * 32 + w]); }
      for (int q = 0; q < 3; ++q) r[q] += (double)m[q] / 256;
    }
    if (it) for (int q = 0; q < 3; ++q) best[q] = std::min(best[q], r[q]);
  }
  printf("%-40s", what);
  if (ND) printf("  DMA %5.1f B/clk (%6.0f clk)", (double)n_dma * ND * 1024 / best[0], best[0]);
  if (NR) printf("  read %5.1f B/clk (%6.0f clk)", (double)n_rd * NR * 1024 / best[1], best[1]);
  if (NWR) printf("  write %5.1f B/clk (%6.0f clk)", (double)n_wr * NWR * 1024 / best[2], best[2]);
  printf("   kernel %.1f us\n", ms_best * 1e3);
}

int main() {
  char* buf; uint32_t* out; uint64_t* st_d;
  CK(hipMalloc(&buf, (size_t)8 * kWindow)); CK(hipMemset(buf, 1, (size_t)8 * kWindow)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&st_d, 256 * 32 * 8));
  // per-role work sized so that every role alone runs ~50-100 us
  const int D = 4096, R = 16384, W = 4096;
  run<4, 0, 0>(buf, st_d, out, D, 0, 0, "4 DMA waves alone");
  run<8, 0, 0>(buf, st_d, out, D / 2, 0, 0, "8 DMA waves alone");
  run<0, 8, 0>(buf, st_d, out, 0, R, 0, "8 read waves alone");
  run<0, 4, 0>(buf, st_d, out, 0, R * 2, 0, "4 read waves alone");
  run<0, 0, 4>(buf, st_d, out, 0, 0, W, "4 write waves alone");
  run<4, 8, 0>(buf, st_d, out, D, R, 0, "4 DMA + 8 read");
  run<4, 8, 0>(buf, st_d, out, D, R * 2, 0, "4 DMA + 8 read (reads outlast DMA)");
  run<4, 8, 0>(buf, st_d, out, D * 2, R, 0, "4 DMA + 8 read (DMA outlasts reads)");
  run<0, 8, 4>(buf, st_d, out, 0, R, W, "8 read + 4 write");
  run<4, 0, 4>(buf, st_d, out, D, 0, W, "4 DMA + 4 write");
  run<4, 8, 4>(buf, st_d, out, D, R, W, "4 DMA + 8 read + 4 write");
  return 0;
}
