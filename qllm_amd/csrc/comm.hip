// One-shot all-reduce for decode-sized tensors over peer-mapped staging buffers (tensor-parallel o_proj / down_proj at M = 1:
// [1, 8192] fp16 = 16 KB, two per Llama-2-70B layer, 160 per token).
//
// A ring or tree all-reduce of 16 KB is latency-bound: RCCL's small-message path costs ~10-20 us per call, the same order as a
// rank's whole shard step (22.8 us per layer, bench.py extra.tp_shard_decode_m1).  MI355X's xGMI is a full mesh (7 links per GPU),
// so every rank can WRITE its vector straight into every peer's memory in one hop and sum locally:
//     push    each rank stores its n elements into slot [parity][rank] of every peer's staging buffer (its own included), 16-byte
//             system-scope (write-through) stores, then one system-scope release + a flag word per peer: flag[parity][rank] = epoch
//     wait    poll the world flags of the own buffer until all carry this call's epoch (relaxed system-scope loads, then one acquire)
//     reduce  y = sum over ranks of slot[parity][r], fp32, in rank order: every rank computes the bit-identical result
// The epoch lives in device memory of each rank (it advances in lockstep: every rank makes the same sequence of calls) and is read
// and bumped by the kernel itself, so a call captured into a hipGraph replays correctly.  Two parities alternate: a rank may enter
// call c+1 and overwrite slot[parity(c+1)] while a slow peer still reduces call c from slot[parity(c)]; it cannot reach call c+2
// before every peer has pushed call c+1, i.e. finished reading call c.  One workgroup; staging memory is fine-grained (uncached
// for peers) device memory allocated HERE (qllm_comm_alloc: the one explicit allocation entry point of the library) and exported /
// imported as HIP IPC handles by the host side (qllm_amd/comm.py), one process per GPU.
// Net-new relative to the reference, which has no distributed code ("RCCL LL/one-shot or a custom IPC-mapped
// P2P kernel").
#include <string.h>

#include "kernels.hpp"

namespace qllm {

namespace {

constexpr int kCommThreads = 1024;

template <bool BF16>
__global__ __launch_bounds__(kCommThreads) void allreduce_oneshot_kernel(void *const *__restrict__ peers, int rank, int world, void *x_inout,
                                                                         int n, size_t slot_bytes, int *status) {
  const size_t payload = 2 * (size_t)world * slot_bytes;
  CommCtl *own = (CommCtl *)((char *)peers[rank] + payload);
  const uint32_t epoch = __hip_atomic_load(&own->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  const int parity = (int)(epoch & 1u);
  const int n16 = n / 8;  // 16-byte chunks (n is a multiple of 8 elements: checked by the host)
  // ---- push
  for (int c = threadIdx.x; c < n16; c += kCommThreads) {
    const uint4_t v = *((const uint4_t *)x_inout + c);
    for (int p = 0; p < world; ++p) {
      char *dst = (char *)peers[p] + ((size_t)parity * world + rank) * slot_bytes;
      store16_sys(dst + (size_t)c * 16, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores ...
  __syncthreads();                                  // ... before any flag is written
  if (threadIdx.x < world) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CommCtl *peer = (CommCtl *)((char *)peers[threadIdx.x] + payload);
    __hip_atomic_store(&peer->flag[parity][rank], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- wait: lane r polls the flag of source rank r in the own buffer
    unsigned spins = 0;
    while (__hip_atomic_load(&own->flag[parity][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 26)) {  // a peer never arrived (seconds): report instead of hanging the GPU
        if (status) *status = 1;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  __syncthreads();
  // ---- reduce (fp32, rank order)
  const char *mine = (const char *)peers[rank] + (size_t)parity * world * slot_bytes;
  for (int c = threadIdx.x; c < n16; c += kCommThreads) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) {
      const uint4_t v = load16_sys(mine + (size_t)r * slot_bytes + (size_t)c * 16);
      const uint32_t wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (BF16) {
          acc[2 * j] += __builtin_bit_cast(float, wds[j] << 16);
          acc[2 * j + 1] += __builtin_bit_cast(float, wds[j] & 0xffff0000u);
        } else {
          const half2_t h = as_h2(wds[j]);
          acc[2 * j] += (float)h.x;
          acc[2 * j + 1] += (float)h.y;
        }
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (BF16) o[j] = (uint32_t)f32_to_bf16(acc[2 * j]) | ((uint32_t)f32_to_bf16(acc[2 * j + 1]) << 16);
      else o[j] = as_u32(half2_t{(half_t)acc[2 * j], (half_t)acc[2 * j + 1]});
    }
    *((uint4_t *)x_inout + c) = uint4_t{o[0], o[1], o[2], o[3]};
  }
  if (threadIdx.x == 0) __hip_atomic_store(&own->epoch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

}  // namespace qllm

using namespace qllm;

extern "C" {

size_t qllm_comm_buffer_bytes(int32_t world, size_t slot_bytes) { return 2 * (size_t)world * slot_bytes + sizeof(CommCtl); }

int qllm_comm_alloc(size_t bytes, void **ptr) {
  clear_error();
  if (!ptr || bytes == 0) return set_error(QLLM_ERR_INVALID, "qllm_comm_alloc: ptr is NULL or bytes == 0");
  QLLM_HIP_CHECK(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
  QLLM_HIP_CHECK(hipMemset(*ptr, 0, bytes));
  QLLM_HIP_CHECK(hipDeviceSynchronize());
  return QLLM_OK;
}

int qllm_comm_free(void *ptr) {
  clear_error();
  if (ptr) QLLM_HIP_CHECK(hipFree(ptr));
  return QLLM_OK;
}

int qllm_comm_export(void *ptr, void *handle64) {
  clear_error();
  if (!ptr || !handle64) return set_error(QLLM_ERR_INVALID, "qllm_comm_export: NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "HIP IPC handle size");
  QLLM_HIP_CHECK(hipIpcGetMemHandle((hipIpcMemHandle_t *)handle64, ptr));
  return QLLM_OK;
}

int qllm_comm_import(const void *handle64, void **ptr) {
  clear_error();
  if (!ptr || !handle64) return set_error(QLLM_ERR_INVALID, "qllm_comm_import: NULL argument");
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof h);
  QLLM_HIP_CHECK(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
  return QLLM_OK;
}

int qllm_comm_close(void *ptr) {
  clear_error();
  if (ptr) QLLM_HIP_CHECK(hipIpcCloseMemHandle(ptr));
  return QLLM_OK;
}

int qllm_allreduce_oneshot(void *const *peers_dev, int32_t rank, int32_t world, void *x_inout, int32_t n, int32_t act_dtype,
                           size_t slot_bytes, int32_t *status_dev, void *stream) {
  clear_error();
  if (!peers_dev || !x_inout) return set_error(QLLM_ERR_INVALID, "qllm_allreduce_oneshot: NULL argument");
  if (world < 1 || world > kCommMaxWorld || rank < 0 || rank >= world) return set_error(QLLM_ERR_INVALID, "world must be 1..%d and 0 <= rank < world (rank=%d world=%d)", kCommMaxWorld, rank, world);
  if (act_dtype != QLLM_F16 && act_dtype != QLLM_BF16) return set_error(QLLM_ERR_INVALID, "act_dtype must be f16 or bf16");
  if (n <= 0 || n % 8 != 0 || (size_t)n * 2 > slot_bytes || slot_bytes % 16 != 0) return set_error(QLLM_ERR_UNSUPPORTED, "n must be a positive multiple of 8 with n * 2 <= slot_bytes (n=%d slot=%zu)", n, slot_bytes);
  if ((uintptr_t)x_inout % 16 != 0) return set_error(QLLM_ERR_INVALID, "x must be 16-byte aligned");
  if (act_dtype == QLLM_BF16)
    hipLaunchKernelGGL(allreduce_oneshot_kernel<true>, dim3(1), dim3(kCommThreads), 0, (hipStream_t)stream, peers_dev, rank, world, x_inout, n, slot_bytes, status_dev);
  else
    hipLaunchKernelGGL(allreduce_oneshot_kernel<false>, dim3(1), dim3(kCommThreads), 0, (hipStream_t)stream, peers_dev, rank, world, x_inout, n, slot_bytes, status_dev);
  QLLM_HIP_CHECK(hipGetLastError());
  return QLLM_OK;
}

}  // extern "C"
