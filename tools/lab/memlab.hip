// Memory-access-pattern lab for the decode matvec (developer tool, not part of the library).
// Build: hipcc --offload-arch=gfx950 -O3 -o memlab memlab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

// V0: flat streaming read, 16 B per lane, U loads in flight, grid covers the buffer once
template <int U>
__global__ __launch_bounds__(256) void stream16(const u4* __restrict__ p, uint32_t* out, size_t n16) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  u4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n16) ? __builtin_nontemporal_load(p + i + u * 256) : u4{0, 0, 0, 0};
  uint32_t a = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) a ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  if (a == 0x12345678u) out[0] = a;
}

// strip pattern: matrix [R rows][N words]; block = NW waves owns CW words of every row (CW*4 bytes per row segment);
// lane (g,i): VEC words at row r+g, col c0 + i*VEC.  ROWS_PER_INSTR = 4.  Each wave: S steps, all issued up front.
template <int VEC, int S, int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void strip(const uint32_t* __restrict__ w, uint32_t* out, int R, int N, int ksplit, int pair) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  int b = blockIdx.x;
  const int strips = N / (16 * VEC);
  int kb = b / strips; b = b % strips;
  if (pair && (strips & 15) == 0) { int x = b & 7, r = b >> 3; b = (((r >> 1) << 3) + x) * 2 + (r & 1); }
  const int rows_per_block = R / ksplit;
  const int r0 = kb * rows_per_block + wave * (rows_per_block / NW);
  const uint32_t* base = w + (size_t)(r0 + g) * N + b * 16 * VEC + i * VEC;
  uint32_t acc = 0;
  for (int s0 = 0; s0 < rows_per_block / NW / 4; s0 += S) {
    uint32_t v[S][VEC];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const uint32_t* q = base + (size_t)(4 * (s0 + s)) * N;
      if constexpr (VEC == 4) { u4 t = NT ? __builtin_nontemporal_load((const u4*)q) : *(const u4*)q; v[s][0] = t.x; v[s][1] = t.y; v[s][2] = t.z; v[s][3] = t.w; }
      else if constexpr (VEC == 2) { u2 t = NT ? __builtin_nontemporal_load((const u2*)q) : *(const u2*)q; v[s][0] = t.x; v[s][1] = t.y; }
      else { v[s][0] = NT ? __builtin_nontemporal_load(q) : *q; }
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc ^= v[s][e];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
float timeit(F f, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  const int NB = 40;  // rotating buffers
  struct Shape { int R, N; } shapes[] = {{512, 4096}, {512, 11008}, {1376, 4096}};
  uint32_t* out; CK(hipMalloc(&out, 64));
  for (auto sh : shapes) {
    size_t words = (size_t)sh.R * sh.N, bytes = words * 4;
    std::vector<uint32_t*> bufs(NB);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    printf("== R=%d N=%d  %.2f MB\n", sh.R, sh.N, bytes / 1e6);
    auto rep = [&](const char* name, float us) { printf("  %-34s %7.2f us  %7.1f GB/s\n", name, us, bytes / us / 1e3); };
    {
      size_t n16 = bytes / 16; int grid = (int)((n16 + 256 * 8 - 1) / (256 * 8));
      rep("stream16 U=8 (256thr)", timeit([&](int i) { stream16<8><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, 200));
      grid = (int)((n16 + 256 * 4 - 1) / (256 * 4));
      rep("stream16 U=4 (256thr)", timeit([&](int i) { stream16<4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, 200));
    }
    const int R = sh.R, N = sh.N;
    if (R == 512) {
      rep("strip64  dword  S=8  NW=16 nt pair", timeit([&](int i) { strip<1, 8, 16, true><<<N / 16, 1024>>>(bufs[i % NB], out, R, N, 1, 1); }, 200));
      rep("strip64  dword  S=8  NW=16 nt nopair", timeit([&](int i) { strip<1, 8, 16, true><<<N / 16, 1024>>>(bufs[i % NB], out, R, N, 1, 0); }, 200));
      rep("strip64  dword  S=8  NW=16 plain", timeit([&](int i) { strip<1, 8, 16, false><<<N / 16, 1024>>>(bufs[i % NB], out, R, N, 1, 1); }, 200));
      rep("strip64  dword  S=16 NW=8  nt", timeit([&](int i) { strip<1, 16, 8, true><<<N / 16, 512>>>(bufs[i % NB], out, R, N, 1, 1); }, 200));
      rep("strip64  dword  S=8  NW=8 ks2 nt", timeit([&](int i) { strip<1, 8, 8, true><<<N / 16 * 2, 512>>>(bufs[i % NB], out, R, N, 2, 1); }, 200));
      rep("strip64  dword  S=8  NW=4 ks4 nt", timeit([&](int i) { strip<1, 8, 4, true><<<N / 16 * 4, 256>>>(bufs[i % NB], out, R, N, 4, 1); }, 200));
      rep("strip128 dwordx2 S=8 NW=8 ks2 nt", timeit([&](int i) { strip<2, 8, 8, true><<<N / 32 * 2, 512>>>(bufs[i % NB], out, R, N, 2, 0); }, 200));
      rep("strip128 dwordx2 S=8 NW=16 ks1 nt", timeit([&](int i) { strip<2, 8, 16, true><<<N / 32, 1024>>>(bufs[i % NB], out, R, N, 1, 0); }, 200));
      rep("strip256 dwordx4 S=8 NW=4 ks4 nt", timeit([&](int i) { strip<4, 8, 4, true><<<N / 64 * 4, 256>>>(bufs[i % NB], out, R, N, 4, 0); }, 200));
      rep("strip256 dwordx4 S=4 NW=4 ks8 nt", timeit([&](int i) { strip<4, 4, 4, true><<<N / 64 * 8, 256>>>(bufs[i % NB], out, R, N, 8, 0); }, 200));
      rep("strip256 dwordx4 S=8 NW=16 ks1 nt", timeit([&](int i) { strip<4, 8, 16, true><<<N / 64, 1024>>>(bufs[i % NB], out, R, N, 1, 0); }, 200));
    } else {
      // R = 1376 = 16 waves * 86 rows -> not a multiple of 4*S; use NW=4/ks split so rows_per_wave is a multiple of 4
      rep("strip64  dword  S=43 NW=8 nt (K=11008)", timeit([&](int i) { strip<1, 43, 8, true><<<N / 16, 512>>>(bufs[i % NB], out, R, N, 1, 1); }, 200));
      rep("strip256 dwordx4 S=43 NW=8 ks1 (64 blk)", timeit([&](int i) { strip<4, 43, 8, true><<<N / 64, 512>>>(bufs[i % NB], out, R, N, 1, 0); }, 200));
      rep("strip256 dwordx4 S=43 NW=2 ks4", timeit([&](int i) { strip<4, 43, 2, true><<<N / 64 * 4, 128>>>(bufs[i % NB], out, R, N, 4, 0); }, 200));
    }
    for (auto b : bufs) CK(hipFree(b));
  }
  return 0;
}
