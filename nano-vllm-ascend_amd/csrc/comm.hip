// One-shot SUM all-reduce over xGMI peer mappings (see include/mi355_nanovllm.h).
//
// Region of one rank (uncached device memory, mapped by every peer through HIP IPC):
//   [0, 4096)            control: epoch u32, finished-workgroup counter u32, sticky timeout flag u32,
//                        spin limit u32 (polls before a peer is declared missing), then what the first give-up was
//                        waiting for: epoch, slice, peer, flag value seen
//   [4096, +flag bytes)  flags[parity 2][source rank world][MAX_SLICES], ONE 128-byte line per flag (u32 epoch of the
//                        last arrival in its first word: flags of neighbouring slices are written by waves on different
//                        XCDs, i.e. through different L2s - they do not share a line)
//   [..., +slot bytes)   slots[parity 2][source rank world][max_bytes]
// A launch cuts the vector into slices (up to 64 element ranges for the plain all-reduce, one row each
// for the fused add+RMSNorm), one single-wave workgroup per slice; workgroup s owns slice s in every phase, so the only cross-GPU dependency
// is per slice and is carried by flags[.][r][s].  The epoch is one number per communicator: every
// workgroup reads it when it starts, the last one to finish advances it - launches are stream
// ordered, so the next launch (or graph replay) sees the new value.
#include <stdlib.h>
#include <string.h>

#include "mi_common.hpp"

namespace mi {

constexpr int MAX_SLICES = 512;  // flag slots per (parity, source): rows of the fused kernel (= the largest decode batch)
constexpr int FLAG_STRIDE = 32;   // u32 words between two flags: a 128-byte line each
constexpr size_t CTRL_BYTES = 4096;
constexpr uint32_t DEFAULT_SPIN_LIMIT = 1u << 26;  // polls of ~1 us each: about a minute

struct CommPtrs {
  uint8_t* region[MI_COMM_MAX_WORLD];
};

struct Layout {
  size_t flags_off, slots_off, slot_stride, total;
};
static Layout layout(int world, size_t max_bytes) {
  Layout l;
  l.flags_off = CTRL_BYTES;
  const size_t flag_bytes = ((size_t)2 * world * MAX_SLICES * FLAG_STRIDE * sizeof(uint32_t) + 255) / 256 * 256;
  l.slots_off = l.flags_off + flag_bytes;
  l.slot_stride = (max_bytes + 255) / 256 * 256;
  l.total = l.slots_off + (size_t)2 * world * l.slot_stride;
  return l;
}

struct CommGeom {
  CommPtrs peers;
  int rank, world;
  size_t flags_off, slots_off, slot_stride;
};

__device__ __forceinline__ uint32_t comm_epoch(const CommGeom& g) {
  return *reinterpret_cast<const volatile uint32_t*>(g.peers.region[g.rank]) + 1;
}
__device__ __forceinline__ uint8_t* comm_slot(const CommGeom& g, int owner, uint32_t par, int source) {
  return g.peers.region[owner] + g.slots_off + ((size_t)par * g.world + source) * g.slot_stride;
}
// After this workgroup's pushes: make them visible, publish slice `s` to every rank, wait for the same
// slice of every source.  Lanes 0..world-1 of the first wave do the flag traffic.
__device__ __forceinline__ void comm_publish_and_wait(const CommGeom& g, uint32_t e, int s, int tid, bool sync) {
  __threadfence_system();  // the slice is visible at system scope before its flag
  if (sync) __syncthreads();
  if (tid < g.world) {
    const uint32_t par = e & 1u;
    uint32_t* theirs = reinterpret_cast<uint32_t*>(g.peers.region[tid] + g.flags_off) +
                       (((size_t)par * g.world + g.rank) * MAX_SLICES + s) * FLAG_STRIDE;
    __hip_atomic_store(theirs, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t* arrive = reinterpret_cast<const uint32_t*>(g.peers.region[g.rank] + g.flags_off) +
                             (((size_t)par * g.world + tid) * MAX_SLICES + s) * FLAG_STRIDE;
    // (once an exchange has timed out the step's results are void - the host raises when it reads the sticky flag: the
    // launches still queued behind it do not each wait their full patience for the same missing peer)
    const volatile uint32_t* ctrl = reinterpret_cast<const volatile uint32_t*>(g.peers.region[g.rank]);
    const uint32_t limit = ctrl[2] ? 64u : ctrl[3];
    uint32_t spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
      if (++spins > limit) {
        uint32_t* own = reinterpret_cast<uint32_t*>(g.peers.region[g.rank]);
        // the FIRST give-up of this rank says what it was waiting for (mi_comm_timeout_info): words 4..7 =
        // {epoch, slice, peer, the flag value seen instead}
        if (__hip_atomic_exchange(own + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
          own[4] = e;
          own[5] = (uint32_t)s;
          own[6] = (uint32_t)tid;
          own[7] = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  if (sync) __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: the slot reads below are not served from a stale line
}
// One thread per workgroup, after its last slot read: the last workgroup of the launch advances the epoch.
__device__ __forceinline__ void comm_finish(const CommGeom& g, uint32_t e, int n_workgroups) {
  uint32_t* ctrl = reinterpret_cast<uint32_t*>(g.peers.region[g.rank]);
  if (atomicAdd(ctrl + 1, 1u) == (uint32_t)n_workgroups - 1) {
    ctrl[1] = 0;
    __hip_atomic_store(ctrl, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// plain all-reduce: one wave per slice (same shape as the fused kernel below: a wave pushes, flags,
// waits and reduces its own 16-byte pieces, no workgroup barrier anywhere)
__global__ __launch_bounds__(64) void allreduce_kernel(CommGeom g, const uint16_t* in, uint16_t* out, int64_t n) {
  const int s = blockIdx.x, lane = threadIdx.x, n_slices = gridDim.x;
  const uint32_t e = comm_epoch(g);
  const uint32_t par = e & 1u;

  const int64_t vecs = n / 8;  // 16-byte pieces
  const int64_t per = (vecs + n_slices - 1) / n_slices;
  const int64_t v0 = min(vecs, (int64_t)s * per), v1 = min(vecs, v0 + per);

  // (a) push this slice into slot [par][rank] of every rank's region, the next-higher rank first
  for (int64_t v = v0 + lane; v < v1; v += 64) {
    const u32x4 x = reinterpret_cast<const u32x4*>(in)[v];
    for (int k = 1; k <= g.world; ++k)
      reinterpret_cast<u32x4*>(comm_slot(g, (g.rank + k) % g.world, par, g.rank))[v] = x;
  }
  // (b) publish, (c) wait
  comm_publish_and_wait(g, e, s, lane, false);
  // (d) sum the `world` slots in rank order
  const uint8_t* slots = comm_slot(g, g.rank, par, 0);
  for (int64_t v = v0 + lane; v < v1; v += 64) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < g.world; ++r) {
      const u32x4 x = reinterpret_cast<const u32x4*>(slots + (size_t)r * g.slot_stride)[v];
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
  if (lane == 0) comm_finish(g, e, n_slices);
}

// The sampler's last step under tensor parallelism: every rank holds, per batch row, the best (sampling key, token)
// of ITS vocabulary shard (mi_pick_final_pairs); the ranks exchange these 8-byte pairs - instead of gathering
// [rows][vocab / world] logits to rank 0 (embed_head.py:62-65) - and every rank reduces them the same way (largest
// key, ties to the lowest token id: what the unsharded sampler picks), so every rank ends up with the step's tokens
// on its own device, inside the captured graph.  One wave per 64 rows.
__global__ __launch_bounds__(64) void pick_exchange_kernel(CommGeom g, const uint2* __restrict__ pairs,
                                                           int64_t* __restrict__ tokens, int n) {
  const int s = blockIdx.x, lane = threadIdx.x, row = s * 64 + lane;
  const uint32_t e = comm_epoch(g);
  const uint32_t par = e & 1u;
  if (row < n) {
    const uint2 p = pairs[row];
    const uint64_t v = ((uint64_t)p.x << 32) | p.y;
    for (int k = 1; k <= g.world; ++k)
      reinterpret_cast<uint64_t*>(comm_slot(g, (g.rank + k) % g.world, par, g.rank))[row] = v;
  }
  comm_publish_and_wait(g, e, s, lane, false);
  if (row < n) {
    const uint8_t* slots = comm_slot(g, g.rank, par, 0);
    float best = -INFINITY;
    uint32_t best_c = 0x7fffffffu;
    for (int r = 0; r < g.world; ++r) {
      const uint64_t v = reinterpret_cast<const uint64_t*>(slots + (size_t)r * g.slot_stride)[row];
      const float key = __uint_as_float((uint32_t)(v >> 32));
      const uint32_t c = (uint32_t)v;
      if (key > best || (key == best && c < best_c)) {
        best = key;
        best_c = c;
      }
    }
    tokens[row] = best_c == 0x7fffffffu ? 0 : (int64_t)best_c;
  }
  if (lane == 0) comm_finish(g, e, gridDim.x);
}

// all-reduce + residual add + RMSNorm in one launch: one wave per token row (decode batches, <= MAX_SLICES rows).
// The row's partial sums are pushed to every rank, summed in rank order in fp32 and rounded to bf16 -
// exactly what allreduce_kernel leaves in memory - then the row goes through the arithmetic of
// rmsnorm_kernel<64, VPL, ADD> (elementwise.hip; same operation order, this file is built with
// -ffp-contract=off as well), so the result is bit-identical to the two-launch sequence.
template <int VPL>
__global__ __launch_bounds__(64) void allreduce_add_rmsnorm_kernel(
    CommGeom g, const uint16_t* __restrict__ x, const uint16_t* __restrict__ residual,
    const uint16_t* __restrict__ w, uint16_t* __restrict__ y, uint16_t* __restrict__ residual_out, int rows,
    int cols, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const int nvec = cols >> 3;
  const int64_t off = (int64_t)row * cols;
  const uint32_t e = comm_epoch(g);
  const uint32_t par = e & 1u;

  u32x4 mine[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vec = lane + i * 64;
    if (vec < nvec) mine[i] = *reinterpret_cast<const u32x4*>(x + off + vec * 8);
  }
  for (int k = 1; k <= g.world; ++k) {
    const int q = (g.rank + k) % g.world;
    uint16_t* dst = reinterpret_cast<uint16_t*>(comm_slot(g, q, par, g.rank)) + off;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int vec = lane + i * 64;
      if (vec < nvec) *reinterpret_cast<u32x4*>(dst + vec * 8) = mine[i];
    }
  }
  comm_publish_and_wait(g, e, row, lane, false);

  const uint8_t* slots = comm_slot(g, g.rank, par, 0);
  float v[VPL][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vec = lane + i * 64;
    if (vec < nvec) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < g.world; ++r) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(
            reinterpret_cast<const uint16_t*>(slots + (size_t)r * g.slot_stride) + off + vec * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += lo_bf(p[j]);
          acc[2 * j + 1] += hi_bf(p[j]);
        }
      }
      const u32x4 rr = *reinterpret_cast<const u32x4*>(residual + off + vec * 8);
      u32x4 ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t sum_bf = pack_bf(acc[2 * j], acc[2 * j + 1]);  // the all-reduce's bf16 result
        const float a = lo_bf(sum_bf) + lo_bf(rr[j]);
        const float b = hi_bf(sum_bf) + hi_bf(rr[j]);
        ro[j] = pack_bf(a, b);
        v[i][2 * j] = a;
        v[i][2 * j + 1] = b;
        ss += a * a;
        ss += b * b;
      }
      *reinterpret_cast<u32x4*>(residual_out + off + vec * 8) = ro;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  ss = wave_sum(ss);  // the butterfly 32 .. 1 of rmsnorm_kernel<64, ...>
  const float rs = 1.0f / sqrtf(ss / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vec = lane + i * 64;
    if (vec < nvec) {
      const u32x4 wr = *reinterpret_cast<const u32x4*>(w + vec * 8);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = rbf(v[i][2 * j] * rs) * lo_bf(wr[j]);
        const float b = rbf(v[i][2 * j + 1] * rs) * hi_bf(wr[j]);
        o[j] = pack_bf(a, b);
      }
      *reinterpret_cast<u32x4*>(y + off + vec * 8) = o;
    }
  }
  if (lane == 0) comm_finish(g, e, rows);
}

// The same seam for a few WIDE rows (decode batches of hidden 2048 ... 8192 models: a Qwen3-32B TP-8 rank's 32 x 5120):
// WPR waves per row instead of one, so that a lane has 1 / WPR of the pushes and slot reads in flight (one wave walks a
// 5120-column row in ten dependent-looking 16-byte steps per operand and peer).  The arithmetic after the rank-ordered sum
// is add_rmsnorm_splitk_rows_kernel<0, WPR>'s (elementwise.hip: what mi_add_rmsnorm runs on such rows - thread t holds
// vectors t and t + 64 WPR, the sum of squares goes lane -> wave butterfly -> the waves in order), so the launch is
// bit-identical to mi_allreduce_sum_bf16 followed by mi_add_rmsnorm on the same rows.
template <int WPR>
__global__ __launch_bounds__(WPR * 64) void allreduce_add_rmsnorm_rows_kernel(
    CommGeom g, const uint16_t* __restrict__ x, const uint16_t* __restrict__ residual,
    const uint16_t* __restrict__ w, uint16_t* __restrict__ y, uint16_t* __restrict__ residual_out, int rows,
    int cols, float eps) {
  __shared__ float wave_ss[WPR];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = cols >> 3;
  const int64_t off = (int64_t)row * cols;
  const uint32_t e = comm_epoch(g);
  const uint32_t par = e & 1u;
  constexpr int MAXV = 2;  // vectors per thread: cols <= WPR * 64 * 8 * MAXV

  u32x4 mine[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vec = tid + i * WPR * 64;
    if (vec < nvec) mine[i] = *reinterpret_cast<const u32x4*>(x + off + vec * 8);
  }
  for (int k = 1; k <= g.world; ++k) {
    const int q = (g.rank + k) % g.world;
    uint16_t* dst = reinterpret_cast<uint16_t*>(comm_slot(g, q, par, g.rank)) + off;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int vec = tid + i * WPR * 64;
      if (vec < nvec) *reinterpret_cast<u32x4*>(dst + vec * 8) = mine[i];
    }
  }
  comm_publish_and_wait(g, e, row, tid, true);  // every wave's pushes are fenced, the first wave's lanes do the flags

  const uint8_t* slots = comm_slot(g, g.rank, par, 0);
  float v[MAXV][8];
  u32x4 wr[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vec = tid + i * WPR * 64;
    if (vec < nvec) {
      const u32x4 rr = *reinterpret_cast<const u32x4*>(residual + off + vec * 8);
      wr[i] = *reinterpret_cast<const u32x4*>(w + vec * 8);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < g.world; ++r) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(
            reinterpret_cast<const uint16_t*>(slots + (size_t)r * g.slot_stride) + off + vec * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[2 * j] += lo_bf(p[j]);
          acc[2 * j + 1] += hi_bf(p[j]);
        }
      }
      u32x4 ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t sum_bf = pack_bf(acc[2 * j], acc[2 * j + 1]);  // the all-reduce's bf16 result
        const float a = lo_bf(sum_bf) + lo_bf(rr[j]);
        const float b = hi_bf(sum_bf) + hi_bf(rr[j]);
        ro[j] = pack_bf(a, b);
        v[i][2 * j] = a;
        v[i][2 * j + 1] = b;
        ss += a * a;
        ss += b * b;
      }
      *reinterpret_cast<u32x4*>(residual_out + off + vec * 8) = ro;
    } else {
      wr[i] = u32x4{0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) wave_ss[wave] = ss;
  __syncthreads();
  float tot = wave_ss[0];
#pragma unroll
  for (int wv = 1; wv < WPR; ++wv) tot += wave_ss[wv];
  const float rs = 1.0f / sqrtf(tot / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vec = tid + i * WPR * 64;
    if (vec < nvec) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = rbf(v[i][2 * j] * rs) * lo_bf(wr[i][j]);
        const float b = rbf(v[i][2 * j + 1] * rs) * hi_bf(wr[i][j]);
        o[j] = pack_bf(a, b);
      }
      *reinterpret_cast<u32x4*>(y + off + vec * 8) = o;
    }
  }
  if (tid == 0) comm_finish(g, e, rows);  // (every thread's slot reads are in front of the barrier above)
}

}  // namespace mi

using namespace mi;

struct mi_comm {
  int rank, world;
  size_t max_bytes;
  Layout lay;
  CommPtrs ptrs;
  CommGeom geom() const { return CommGeom{ptrs, rank, world, lay.flags_off, lay.slots_off, lay.slot_stride}; }
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
  const uint32_t limit = DEFAULT_SPIN_LIMIT;
  if (hipMemcpy(static_cast<uint8_t*>(p) + 3 * sizeof(uint32_t), &limit, sizeof(limit), hipMemcpyHostToDevice) !=
      hipSuccess) {
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
  const int64_t vecs = n / 8;
  int slices = (int)((vecs + 127) / 128);  // ~2 KiB per wave, at most MAX_SLICES waves
  if (slices > MAX_SLICES) slices = MAX_SLICES;
  hipLaunchKernelGGL(allreduce_kernel, dim3(slices), dim3(64), 0, S(stream), comm->geom(), in, out, n);
  return check_launch();
}

extern "C" int mi_allreduce_add_rmsnorm(mi_comm* comm, const mi_bf16* x, const mi_bf16* residual,
                                        const mi_bf16* weight, mi_bf16* out, mi_bf16* residual_out, int rows,
                                        int cols, float eps, mi_stream stream) {
  if (!comm || !x || !residual || !weight || !out || !residual_out || rows < 0 || cols <= 0) return MI_EINVAL;
  if (rows > MAX_SLICES || cols % 8 || cols < 512 || cols > 8192 || (size_t)rows * cols * 2 > comm->max_bytes)
    return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(residual) || !aligned16(weight) || !aligned16(out) || !aligned16(residual_out))
    return MI_EINVAL;
  if (rows == 0) return MI_OK;
  const int nvec = cols / 8;
  hipStream_t st = S(stream);
  // a few wide rows: four or eight waves per row - where mi_add_rmsnorm takes its multi-wave kernel, with its arithmetic
  if (rows <= 64 && cols >= 4096 && tuning(MI_TUNE_NORM_WPR) == 4) {
    if (cols <= 4 * 64 * 8 * 2)
      hipLaunchKernelGGL((allreduce_add_rmsnorm_rows_kernel<4>), dim3(rows), dim3(256), 0, st, comm->geom(), x, residual,
                         weight, out, residual_out, rows, cols, eps);
    else
      hipLaunchKernelGGL((allreduce_add_rmsnorm_rows_kernel<8>), dim3(rows), dim3(512), 0, st, comm->geom(), x, residual,
                         weight, out, residual_out, rows, cols, eps);
    return check_launch();
  }
#define LAUNCH_ARN(V)                                                                                           \
  hipLaunchKernelGGL((allreduce_add_rmsnorm_kernel<V>), dim3(rows), dim3(64), 0, st, comm->geom(), x, residual, \
                     weight, out, residual_out, rows, cols, eps)
  if (nvec <= 64) LAUNCH_ARN(1);
  else if (nvec <= 128) LAUNCH_ARN(2);
  else if (nvec <= 256) LAUNCH_ARN(4);
  else if (nvec <= 512) LAUNCH_ARN(8);
  else LAUNCH_ARN(16);
#undef LAUNCH_ARN
  return check_launch();
}

extern "C" int mi_pick_exchange(mi_comm* comm, const void* pairs, int64_t* tokens, int rows, mi_stream stream) {
  if (!comm || !pairs || !tokens || rows < 0) return MI_EINVAL;
  if ((rows + 63) / 64 > MAX_SLICES || (size_t)rows * 8 > comm->max_bytes) return MI_EUNSUPPORTED;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(pick_exchange_kernel, dim3((rows + 63) / 64), dim3(64), 0, S(stream), comm->geom(),
                     static_cast<const uint2*>(pairs), tokens, rows);
  return check_launch();
}

// the sticky timeout flag, copied to (pinned) host memory behind the work already queued on `stream` - the form a
// caller uses that must not synchronise the device (the engine's lookahead loop)
extern "C" int mi_comm_status_async(mi_comm* comm, int* timed_out_host, mi_stream stream) {
  if (!comm || !timed_out_host) return MI_EINVAL;
  if (hipMemcpyAsync(timed_out_host, comm->ptrs.region[comm->rank] + 2 * sizeof(uint32_t), sizeof(uint32_t),
                     hipMemcpyDeviceToHost, S(stream)) != hipSuccess)
    return MI_ERUNTIME;
  return MI_OK;
}

extern "C" int mi_comm_set_spin_limit(mi_comm* comm, uint32_t polls) {
  if (!comm || polls == 0) return MI_EINVAL;
  if (hipMemcpy(comm->ptrs.region[comm->rank] + 3 * sizeof(uint32_t), &polls, sizeof(polls), hipMemcpyHostToDevice) !=
      hipSuccess)
    return MI_ERUNTIME;
  return MI_OK;
}

extern "C" int mi_comm_timeout_info(mi_comm* comm, uint32_t info[4]) {
  if (!comm || !info) return MI_EINVAL;
  if (hipMemcpy(info, comm->ptrs.region[comm->rank] + 4 * sizeof(uint32_t), 4 * sizeof(uint32_t),
                hipMemcpyDeviceToHost) != hipSuccess)
    return MI_ERUNTIME;
  return MI_OK;
}

extern "C" int mi_comm_status(mi_comm* comm, int* timed_out) {
  if (!comm || !timed_out) return MI_EINVAL;
  uint32_t v = 0;
  if (hipMemcpy(&v, comm->ptrs.region[comm->rank] + 2 * sizeof(uint32_t), sizeof(v), hipMemcpyDeviceToHost) !=
      hipSuccess)
    return MI_ERUNTIME;
  *timed_out = (int)v;
  return MI_OK;
}
