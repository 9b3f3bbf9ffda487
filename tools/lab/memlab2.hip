// Memory-access-pattern lab, round 3 (developer tool, not part of the library): what one decode launch can stream when the packed
// words are laid out strip-major ([N/16][K/8][16] words: a 16-column strip is one contiguous region) instead of row-stream
// ([K/8][N]: a strip is K/8 separate 64-byte segments).  Pure reads, no arithmetic: the floor a real kernel can approach.
// Build: hipcc --offload-arch=gfx950 -O3 -o memlab2 memlab2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// row-stream pattern of the round-2 kernel: lane (g,i) reads VEC words at row 4t+g, column c0 + i*VEC; all S loads up front
template <int VEC, int S, int NW>
__global__ __launch_bounds__(NW * 64) void rowstream(const uint32_t* __restrict__ w, uint32_t* out, int R, int N, int pair) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
  int b = blockIdx.x;
  const int strips = N / (16 * VEC);
  if (pair && (strips & 15) == 0) { int x = b & 7, r = b >> 3; b = (((r >> 1) << 3) + x) * 2 + (r & 1); }
  const int T = R / 4;
  uint32_t v[S][VEC];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = min(wave * S + s, T - 1);
    const uint32_t* q = w + (size_t)(4 * t + g) * N + b * 16 * VEC + i * VEC;
    if constexpr (VEC == 4) { u4 x = __builtin_nontemporal_load((const u4*)q); v[s][0] = x.x; v[s][1] = x.y; v[s][2] = x.z; v[s][3] = x.w; }
    else v[s][0] = __builtin_nontemporal_load(q);
  }
  uint32_t acc = 0;
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc ^= v[s][e];
  if (acc == 0x12345678u) out[0] = acc;
}

// strip-major pattern: strip s = words [s*R*W, (s+1)*R*W), row r of the strip = W words; lane (g,i): VEC words at row 4t+g,
// column i*VEC (W = 16*VEC): one wave-load = 4 rows x W words = 256*VEC contiguous bytes; all S loads up front
template <int VEC, int S, int NW, bool NT>
__global__ __launch_bounds__(NW * 64) void stripmajor(const uint32_t* __restrict__ w, uint32_t* out, int R) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int W = 16 * VEC;
  const int T = R / 4;
  const uint32_t* base = w + (size_t)blockIdx.x * R * W + lane * VEC;
  uint32_t v[S][VEC];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = min(wave * S + s, T - 1);
    const uint32_t* q = base + (size_t)t * 4 * W;
    if constexpr (VEC == 4) { u4 x = NT ? __builtin_nontemporal_load((const u4*)q) : *(const u4*)q; v[s][0] = x.x; v[s][1] = x.y; v[s][2] = x.z; v[s][3] = x.w; }
    else v[s][0] = NT ? __builtin_nontemporal_load(q) : *q;
  }
  uint32_t acc = 0;
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc ^= v[s][e];
  if (acc == 0x12345678u) out[0] = acc;
}

// the same with an epilogue like the real kernel's: partials through LDS, one barrier, 16 outputs stored per block; and a prologue
// that reads the block's activation vector (R*8 halves, L2) before anything else is used
template <int S, int NW>
__global__ __launch_bounds__(NW * 64) void stripmajor_epi(const uint32_t* __restrict__ w, const uint32_t* __restrict__ x, uint16_t* y, int R) {
  __shared__ uint32_t red[NW * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = R / 4;
  const uint32_t* base = w + (size_t)blockIdx.x * R * 16 + lane;
  // activation chunk of this wave: S k-steps x 32 halves = S*16 words: lanes < S*4 load 16 B each
  u4 xa = {0, 0, 0, 0};
  if (lane < S * 4) xa = *(const u4*)(x + (size_t)min(wave * S * 16 + lane * 4, R * 4 - 4));
  uint32_t v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int t = min(wave * S + s, T - 1);
    v[s] = __builtin_nontemporal_load(base + (size_t)t * 64);
  }
  uint32_t acc = xa.x ^ xa.y ^ xa.z ^ xa.w;
#pragma unroll
  for (int s = 0; s < S; ++s) acc ^= v[s];
  acc ^= __shfl_xor(acc, 16); acc ^= __shfl_xor(acc, 32);
  if (lane < 16) red[wave * 16 + lane] = acc;
  __syncthreads();
  if (threadIdx.x < 16) {
    uint32_t a = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) a ^= red[q * 16 + threadIdx.x];
    y[blockIdx.x * 16 + threadIdx.x] = (uint16_t)a;
  }
}

// flat: block b reads the b-th contiguous chunk of the buffer, 16 B per lane, U loads per lane all in flight
template <int U, int NW>
__global__ __launch_bounds__(NW * 64) void flat(const u4* __restrict__ p, uint32_t* out, size_t n16) {
  const size_t per_block = (size_t)NW * 64 * U;
  const size_t i0 = (size_t)blockIdx.x * per_block + (threadIdx.x >> 6) * 64 * U + (threadIdx.x & 63);
  u4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + min(i0 + (size_t)u * 64, n16 - 1));
  uint32_t a = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) a ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  if (a == 0x12345678u) out[0] = a;
}

// persistent flavour: grid = one block per CU x BPC; block b walks strips b, b+grid, ... with the NEXT strip's loads issued before
// the current strip is consumed (two register sets)
template <int S, int NW>
__global__ __launch_bounds__(NW * 64) void stripmajor_loop(const uint32_t* __restrict__ w, uint32_t* out, int R, int strips) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int T = R / 4;
  uint32_t acc = 0;
  uint32_t v[2][S];
  auto issue = [&](int st, uint32_t (&vv)[S]) {
    const uint32_t* base = w + (size_t)st * R * 16 + lane;
#pragma unroll
    for (int s = 0; s < S; ++s) vv[s] = __builtin_nontemporal_load(base + (size_t)min(wave * S + s, T - 1) * 64);
  };
  int st = blockIdx.x;
  if (st < strips) issue(st, v[0]);
  for (; st < strips; st += 2 * gridDim.x) {
    if (st + (int)gridDim.x < strips) issue(st + gridDim.x, v[1]);
#pragma unroll
    for (int s = 0; s < S; ++s) acc ^= v[0][s];
    if (st + 2 * (int)gridDim.x < strips) issue(st + 2 * gridDim.x, v[0]);
    if (st + (int)gridDim.x < strips) {
#pragma unroll
      for (int s = 0; s < S; ++s) acc ^= v[1][s];
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
float timeit(F f, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) f(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f(i);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3f / iters;
}

int main() {
  const size_t kFootprint = 700ull << 20;  // rotating copies beyond the 256 MB Infinity Cache
  struct Shape { const char* name; int R, N; } shapes[] = {{"o 4096->4096", 512, 4096}, {"qkv 4096->12288", 512, 12288}, {"gate/up 4096->22016", 512, 22016}, {"down 11008->4096", 1376, 4096}};
  uint32_t* out; CK(hipMalloc(&out, 1 << 20));
  uint32_t* xbuf; CK(hipMalloc(&xbuf, 1 << 20)); CK(hipMemset(xbuf, 1, 1 << 20));
  const int IT = 300;
  for (auto sh : shapes) {
    size_t words = (size_t)sh.R * sh.N, bytes = words * 4;
    const int NB = (int)std::min<size_t>(48, std::max<size_t>(2, kFootprint / bytes));
    std::vector<uint32_t*> bufs(NB);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
    printf("== %s  R=%d N=%d  %.2f MB  (%d rotating copies)\n", sh.name, sh.R, sh.N, bytes / 1e6, NB);
    auto rep = [&](const char* name, float us) { printf("  %-52s %7.2f us  %7.1f GB/s\n", name, us, bytes / us / 1e3); fflush(stdout); };
    const int R = sh.R, N = sh.N;
    const size_t n16 = bytes / 16;
    rep("flat stream16 U=8 256thr (grid covers buffer)", timeit([&](int i) { flat<8, 4><<<(int)((n16 + 2047) / 2048), 256>>>((const u4*)bufs[i % NB], out, n16); }, IT));
    if (R == 512) {
      rep("rowstream cpl=1 NW=16 S=8 pair (r02 o_proj form)", timeit([&](int i) { rowstream<1, 8, 16><<<N / 16, 1024>>>(bufs[i % NB], out, R, N, 1); }, IT));
      rep("rowstream cpl=4 NW=8 S=16 (r02 qkv/gate-up form)", timeit([&](int i) { rowstream<4, 16, 8><<<N / 64, 512>>>(bufs[i % NB], out, R, N, 0); }, IT));
      rep("stripmajor W=16 NW=16 S=8  nt", timeit([&](int i) { stripmajor<1, 8, 16, true><<<N / 16, 1024>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=16 S=8  plain", timeit([&](int i) { stripmajor<1, 8, 16, false><<<N / 16, 1024>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=8  S=16 nt", timeit([&](int i) { stripmajor<1, 16, 8, true><<<N / 16, 512>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=4  S=32 nt", timeit([&](int i) { stripmajor<1, 32, 4, true><<<N / 16, 256>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=2  S=64 nt", timeit([&](int i) { stripmajor<1, 64, 2, true><<<N / 16, 128>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=64 NW=8  S=16 nt (dwordx4)", timeit([&](int i) { stripmajor<4, 16, 8, true><<<N / 64, 512>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=64 NW=16 S=8  nt (dwordx4)", timeit([&](int i) { stripmajor<4, 8, 16, true><<<N / 64, 1024>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=16 S=8 + x load + LDS reduce + store", timeit([&](int i) { stripmajor_epi<8, 16><<<N / 16, 1024>>>(bufs[i % NB], xbuf, (uint16_t*)out, R); }, IT));
      rep("stripmajor W=16 NW=8 S=16 + x load + LDS reduce + store", timeit([&](int i) { stripmajor_epi<16, 8><<<N / 16, 512>>>(bufs[i % NB], xbuf, (uint16_t*)out, R); }, IT));
      rep("stripmajor loop grid=256 NW=16 S=8 (2 register sets)", timeit([&](int i) { stripmajor_loop<8, 16><<<256, 1024>>>(bufs[i % NB], out, R, N / 16); }, IT));
      rep("stripmajor loop grid=512 NW=8 S=16 (2 register sets)", timeit([&](int i) { stripmajor_loop<16, 8><<<512, 512>>>(bufs[i % NB], out, R, N / 16); }, IT));
      rep("stripmajor loop grid=1024 NW=4 S=32 (2 register sets)", timeit([&](int i) { stripmajor_loop<32, 4><<<1024, 256>>>(bufs[i % NB], out, R, N / 16); }, IT));
      rep("L2/MALL-warm: stripmajor W=16 NW=16 S=8 nt, ONE buffer", timeit([&](int i) { stripmajor<1, 8, 16, true><<<N / 16, 1024>>>(bufs[0], out, R); }, IT));
      rep("L2/MALL-warm: stripmajor W=16 NW=16 S=8 plain, ONE buffer", timeit([&](int i) { stripmajor<1, 8, 16, false><<<N / 16, 1024>>>(bufs[0], out, R); }, IT));
    } else {
      rep("rowstream cpl=1 NW=16 S=22 pair (r02 down form)", timeit([&](int i) { rowstream<1, 22, 16><<<N / 16, 1024>>>(bufs[i % NB], out, R, N, 1); }, IT));
      rep("stripmajor W=16 NW=16 S=22 nt", timeit([&](int i) { stripmajor<1, 22, 16, true><<<N / 16, 1024>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=8  S=43 nt", timeit([&](int i) { stripmajor<1, 43, 8, true><<<N / 16, 512>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=64 NW=16 S=22 nt (dwordx4, 64 blocks)", timeit([&](int i) { stripmajor<4, 22, 16, true><<<N / 64, 1024>>>(bufs[i % NB], out, R); }, IT));
      rep("stripmajor W=16 NW=16 S=22 + x load + LDS reduce + store", timeit([&](int i) { stripmajor_epi<22, 16><<<N / 16, 1024>>>(bufs[i % NB], xbuf, (uint16_t*)out, R); }, IT));
      rep("L2/MALL-warm: stripmajor W=16 NW=16 S=22 nt, ONE buffer", timeit([&](int i) { stripmajor<1, 22, 16, true><<<N / 16, 1024>>>(bufs[0], out, R); }, IT));
    }
    // flat contiguous split over exactly 256 / 512 / 1024 blocks (the "any layout" bound for this byte count)
    {
      const size_t per256 = (n16 + 255) / 256;  // 16-byte units per block at grid 256
      auto flat_grid = [&](int grid, int nw, const char* name) {
        const size_t per_lane = (n16 + (size_t)grid * nw * 64 - 1) / ((size_t)grid * nw * 64);
        float us = -1.f;
        if (per_lane <= 2) us = timeit([&](int i) { if (nw == 16) flat<2, 16><<<grid, 1024>>>((const u4*)bufs[i % NB], out, n16); else if (nw == 8) flat<2, 8><<<grid, 512>>>((const u4*)bufs[i % NB], out, n16); else flat<2, 4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, IT);
        else if (per_lane <= 4) us = timeit([&](int i) { if (nw == 16) flat<4, 16><<<grid, 1024>>>((const u4*)bufs[i % NB], out, n16); else if (nw == 8) flat<4, 8><<<grid, 512>>>((const u4*)bufs[i % NB], out, n16); else flat<4, 4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, IT);
        else if (per_lane <= 6) us = timeit([&](int i) { if (nw == 16) flat<6, 16><<<grid, 1024>>>((const u4*)bufs[i % NB], out, n16); else if (nw == 8) flat<6, 8><<<grid, 512>>>((const u4*)bufs[i % NB], out, n16); else flat<6, 4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, IT);
        else if (per_lane <= 12) us = timeit([&](int i) { if (nw == 16) flat<12, 16><<<grid, 1024>>>((const u4*)bufs[i % NB], out, n16); else if (nw == 8) flat<12, 8><<<grid, 512>>>((const u4*)bufs[i % NB], out, n16); else flat<12, 4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, IT);
        else if (per_lane <= 24) us = timeit([&](int i) { if (nw == 16) flat<24, 16><<<grid, 1024>>>((const u4*)bufs[i % NB], out, n16); else if (nw == 8) flat<24, 8><<<grid, 512>>>((const u4*)bufs[i % NB], out, n16); else flat<24, 4><<<grid, 256>>>((const u4*)bufs[i % NB], out, n16); }, IT);
        if (us > 0) rep(name, us); else printf("  %-52s (per-lane loads %zu: skipped)\n", name, per_lane);
      };
      (void)per256;
      flat_grid(256, 16, "flat split grid=256  NW=16 (dwordx4, all in flight)");
      flat_grid(512, 8, "flat split grid=512  NW=8");
      flat_grid(1024, 4, "flat split grid=1024 NW=4");
      flat_grid(512, 16, "flat split grid=512  NW=16");
      flat_grid(2048, 4, "flat split grid=2048 NW=4");
    }
    for (auto b : bufs) CK(hipFree(b));
  }
  // one decoder layer as four dependent launches (q/k/v, o, gate/up, down), rotating over 6 layers' worth of buffers
  {
    struct L { int R, N; } ls[4] = {{512, 12288}, {512, 4096}, {512, 22016}, {1376, 4096}};
    const int NL = 8;
    std::vector<uint32_t*> bufs(4 * NL);
    size_t layer_bytes = 0;
    for (int l = 0; l < NL; ++l)
      for (int j = 0; j < 4; ++j) { size_t b = (size_t)ls[j].R * ls[j].N * 4; CK(hipMalloc(&bufs[l * 4 + j], b)); CK(hipMemset(bufs[l * 4 + j], 1, b)); if (l == 0) layer_bytes += b; }
    auto rep = [&](const char* name, float us) { printf("  %-52s %7.2f us per layer  %7.1f GB/s\n", name, us, layer_bytes / us / 1e3); fflush(stdout); };
    printf("== one decoder layer = 4 launches, %.1f MB of packed words, %d rotating layers (%.0f MB)\n", layer_bytes / 1e6, NL, NL * layer_bytes / 1e6);
    rep("r02 forms: rowstream cpl4/cpl1/cpl4/cpl1", timeit([&](int i) {
      uint32_t** b = &bufs[(i % NL) * 4];
      rowstream<4, 16, 8><<<12288 / 64, 512>>>(b[0], out, 512, 12288, 0);
      rowstream<1, 8, 16><<<4096 / 16, 1024>>>(b[1], out, 512, 4096, 1);
      rowstream<4, 16, 8><<<22016 / 64, 512>>>(b[2], out, 512, 22016, 0);
      rowstream<1, 22, 16><<<4096 / 16, 1024>>>(b[3], out, 1376, 4096, 1);
    }, 200));
    rep("stripmajor W=16 NW=16 everywhere", timeit([&](int i) {
      uint32_t** b = &bufs[(i % NL) * 4];
      stripmajor<1, 8, 16, true><<<12288 / 16, 1024>>>(b[0], out, 512);
      stripmajor<1, 8, 16, true><<<4096 / 16, 1024>>>(b[1], out, 512);
      stripmajor<1, 8, 16, true><<<22016 / 16, 1024>>>(b[2], out, 512);
      stripmajor<1, 22, 16, true><<<4096 / 16, 1024>>>(b[3], out, 1376);
    }, 200));
    rep("stripmajor W=16 NW=8 (NW=16 for down)", timeit([&](int i) {
      uint32_t** b = &bufs[(i % NL) * 4];
      stripmajor<1, 16, 8, true><<<12288 / 16, 512>>>(b[0], out, 512);
      stripmajor<1, 16, 8, true><<<4096 / 16, 512>>>(b[1], out, 512);
      stripmajor<1, 16, 8, true><<<22016 / 16, 512>>>(b[2], out, 512);
      stripmajor<1, 22, 16, true><<<4096 / 16, 1024>>>(b[3], out, 1376);
    }, 200));
    rep("stripmajor W=16 NW=4 (NW=16 for down)", timeit([&](int i) {
      uint32_t** b = &bufs[(i % NL) * 4];
      stripmajor<1, 32, 4, true><<<12288 / 16, 256>>>(b[0], out, 512);
      stripmajor<1, 32, 4, true><<<4096 / 16, 256>>>(b[1], out, 512);
      stripmajor<1, 32, 4, true><<<22016 / 16, 256>>>(b[2], out, 512);
      stripmajor<1, 22, 16, true><<<4096 / 16, 1024>>>(b[3], out, 1376);
    }, 200));
    rep("stripmajor + x load + reduce + store, NW=16", timeit([&](int i) {
      uint32_t** b = &bufs[(i % NL) * 4];
      stripmajor_epi<8, 16><<<12288 / 16, 1024>>>(b[0], xbuf, (uint16_t*)out, 512);
      stripmajor_epi<8, 16><<<4096 / 16, 1024>>>(b[1], xbuf, (uint16_t*)out, 512);
      stripmajor_epi<8, 16><<<22016 / 16, 1024>>>(b[2], xbuf, (uint16_t*)out, 512);
      stripmajor_epi<22, 16><<<4096 / 16, 1024>>>(b[3], xbuf, (uint16_t*)out, 1376);
    }, 200));
    // the same four launches captured into a graph and replayed (the bench's form)
    {
      hipStream_t st; CK(hipStreamCreate(&st));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int l = 0; l < NL; ++l) {
        uint32_t** b = &bufs[l * 4];
        stripmajor_epi<8, 16><<<12288 / 16, 1024, 0, st>>>(b[0], xbuf, (uint16_t*)out, 512);
        stripmajor_epi<8, 16><<<4096 / 16, 1024, 0, st>>>(b[1], xbuf, (uint16_t*)out, 512);
        stripmajor_epi<8, 16><<<22016 / 16, 1024, 0, st>>>(b[2], xbuf, (uint16_t*)out, 512);
        stripmajor_epi<22, 16><<<4096 / 16, 1024, 0, st>>>(b[3], xbuf, (uint16_t*)out, 1376);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      rep("  ... the same as a hipGraph of 8 layers, replayed", ms * 1e3f / 50 / NL);
    }
  }
  return 0;
}
