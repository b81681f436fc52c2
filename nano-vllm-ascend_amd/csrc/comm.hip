// One-shot SUM all-reduce over xGMI peer mappings (see include/mi355_nanovllm.h).
//
// Region of one rank (uncached device memory, mapped by every peer through HIP IPC):
//   [0, 4096)            control: epoch[SLICES] u32, then the sticky timeout flag
//   [4096, +flag bytes)  flags[parity 2][source rank world][SLICES] u32  (epoch of the last arrival)
//   [..., +slot bytes)   slots[parity 2][source rank world][max_bytes]
// A launch has SLICES workgroups; workgroup s owns elements [s*chunk, (s+1)*chunk) of the vector in
// every phase, so the only cross-GPU dependency is per slice and is carried by flags[.][r][s].
#include <stdlib.h>
#include <string.h>

#include "mi_common.hpp"

namespace mi {

constexpr int SLICES = 16;
constexpr size_t CTRL_BYTES = 4096;
constexpr uint32_t SPIN_LIMIT = 1u << 22;  // polls of ~0.5 us each before a peer is declared missing

struct CommPtrs {
  uint8_t* region[MI_COMM_MAX_WORLD];
};

struct Layout {
  size_t flags_off, slots_off, slot_stride, total;
};
static Layout layout(int world, size_t max_bytes) {
  Layout l;
  l.flags_off = CTRL_BYTES;
  const size_t flag_bytes = ((size_t)2 * world * SLICES * sizeof(uint32_t) + 255) / 256 * 256;
  l.slots_off = l.flags_off + flag_bytes;
  l.slot_stride = (max_bytes + 255) / 256 * 256;
  l.total = l.slots_off + (size_t)2 * world * l.slot_stride;
  return l;
}

__global__ __launch_bounds__(256) void allreduce_kernel(CommPtrs peers, int rank, int world, size_t flags_off,
                                                        size_t slots_off, size_t slot_stride,
                                                        const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                        int64_t n) {
  const int s = blockIdx.x, tid = threadIdx.x;
  uint8_t* mine = peers.region[rank];
  uint32_t* epoch = reinterpret_cast<uint32_t*>(mine) + s;
  uint32_t* timeout_flag = reinterpret_cast<uint32_t*>(mine) + SLICES;
  const uint32_t e = *epoch + 1;  // written only by this workgroup of the previous launch (stream order)
  const uint32_t par = e & 1u;

  const int64_t vecs = n / 8;  // 16-byte pieces
  const int64_t per = (vecs + SLICES - 1) / SLICES;
  const int64_t v0 = min(vecs, (int64_t)s * per), v1 = min(vecs, v0 + per);

  // (a) push this slice into slot [par][rank] of every rank's region, the farthest peers first
  for (int k = 1; k <= world; ++k) {
    const int q = (rank + k) % world;
    u32x4* dst = reinterpret_cast<u32x4*>(peers.region[q] + slots_off + ((size_t)par * world + rank) * slot_stride);
    for (int64_t v = v0 + tid; v < v1; v += 256) dst[v] = reinterpret_cast<const u32x4*>(in)[v];
  }
  __threadfence_system();  // the slice is visible at system scope before its flag
  __syncthreads();
  // (b) publish, (c) wait for the same slice of every source
  if (tid < world) {
    uint32_t* theirs = reinterpret_cast<uint32_t*>(peers.region[tid] + flags_off) + ((size_t)par * world + rank) * SLICES + s;
    __hip_atomic_store(theirs, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* arrive = reinterpret_cast<const uint32_t*>(mine + flags_off) + ((size_t)par * world + tid) * SLICES + s;
    uint32_t spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      if (++spins > SPIN_LIMIT) {
        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: later slot reads are not served from a stale cache line
  // (d) sum the `world` slots in rank order
  const uint8_t* slots = mine + slots_off + (size_t)par * world * slot_stride;
  for (int64_t v = v0 + tid; v < v1; v += 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < world; ++r) {
      const u32x4 x = reinterpret_cast<const u32x4*>(slots + (size_t)r * slot_stride)[v];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += lo_bf(x[i]);
        acc[2 * i + 1] += hi_bf(x[i]);
      }
    }
    u32x4 y;
#pragma unroll
    for (int i = 0; i < 4; ++i) y[i] = pack_bf(acc[2 * i], acc[2 * i + 1]);
    reinterpret_cast<u32x4*>(out)[v] = y;
  }
  if (tid == 0) *epoch = e;
}

}  // namespace mi

using namespace mi;

struct mi_comm {
  int rank, world;
  size_t max_bytes;
  Layout lay;
  CommPtrs ptrs;
};

extern "C" size_t mi_comm_region_bytes(int world, size_t max_bytes) {
  if (world < 1 || world > MI_COMM_MAX_WORLD) return 0;
  return layout(world, max_bytes).total;
}

extern "C" int mi_comm_region_alloc(size_t bytes, void** region, void* ipc_handle) {
  if (!region || !ipc_handle || bytes == 0) return MI_EINVAL;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) return MI_ERUNTIME;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(p);
    return MI_ERUNTIME;
  }
  hipIpcMemHandle_t h;
  static_assert(sizeof(h) <= MI_IPC_HANDLE_BYTES, "IPC handle does not fit the ABI's buffer");
  if (hipIpcGetMemHandle(&h, p) != hipSuccess) {
    (void)hipFree(p);
    return MI_ERUNTIME;
  }
  memset(ipc_handle, 0, MI_IPC_HANDLE_BYTES);
  memcpy(ipc_handle, &h, sizeof(h));
  *region = p;
  return MI_OK;
}

extern "C" int mi_comm_region_open(const void* ipc_handle, void** region) {
  if (!ipc_handle || !region) return MI_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, ipc_handle, sizeof(h));
  void* p = nullptr;
  if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) return MI_ERUNTIME;
  *region = p;
  return MI_OK;
}

extern "C" int mi_comm_region_close(void* region) {
  if (!region) return MI_EINVAL;
  return hipIpcCloseMemHandle(region) == hipSuccess ? MI_OK : MI_ERUNTIME;
}

extern "C" int mi_comm_region_free(void* region) {
  if (!region) return MI_EINVAL;
  return hipFree(region) == hipSuccess ? MI_OK : MI_ERUNTIME;
}

extern "C" int mi_comm_create(int rank, int world, void* const* regions, size_t max_bytes, mi_comm** out) {
  if (!regions || !out || world < 1 || world > MI_COMM_MAX_WORLD || rank < 0 || rank >= world || max_bytes == 0)
    return MI_EINVAL;
  mi_comm* c = static_cast<mi_comm*>(calloc(1, sizeof(mi_comm)));
  if (!c) return MI_ERUNTIME;
  c->rank = rank;
  c->world = world;
  c->max_bytes = max_bytes;
  c->lay = layout(world, max_bytes);
  for (int r = 0; r < world; ++r) {
    if (!regions[r]) {
      free(c);
      return MI_EINVAL;
    }
    c->ptrs.region[r] = static_cast<uint8_t*>(regions[r]);
  }
  *out = c;
  return MI_OK;
}

extern "C" int mi_comm_destroy(mi_comm* comm) {
  if (!comm) return MI_EINVAL;
  free(comm);
  return MI_OK;
}

extern "C" int mi_allreduce_sum_bf16(mi_comm* comm, const mi_bf16* in, mi_bf16* out, int64_t n, mi_stream stream) {
  if (!comm || !in || !out || n < 0) return MI_EINVAL;
  if (n % 8 || (size_t)n * 2 > comm->max_bytes) return MI_EUNSUPPORTED;
  if (!aligned16(in) || !aligned16(out)) return MI_EINVAL;
  if (n == 0) return MI_OK;
  hipLaunchKernelGGL(allreduce_kernel, dim3(SLICES), dim3(256), 0, S(stream), comm->ptrs, comm->rank, comm->world,
                     comm->lay.flags_off, comm->lay.slots_off, comm->lay.slot_stride, in, out, n);
  return check_launch();
}

extern "C" int mi_comm_status(mi_comm* comm, int* timed_out) {
  if (!comm || !timed_out) return MI_EINVAL;
  uint32_t v = 0;
  if (hipMemcpy(&v, comm->ptrs.region[comm->rank] + SLICES * sizeof(uint32_t), sizeof(v), hipMemcpyDeviceToHost) !=
      hipSuccess)
    return MI_ERUNTIME;
  *timed_out = (int)v;
  return MI_OK;
}
