// HBM streaming ceilings on this node: read-only and copy, 16 B per lane, several loads in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u32x4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

// contiguous 1 KiB per wave-load, each wave owns a contiguous 16 KiB run (like the KV / weight streams)
template <bool NT>
__global__ __launch_bounds__(256) void read_runs_kernel(const u32x4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16) {
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * 1024 + lane;  // 16 loads x 64 lanes
  if (base + 15 * 64 >= n16) return;
  u32x4 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = NT ? __builtin_nontemporal_load(src + base + u * 64) : src[base + u * 64];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 16; ++u) acc ^= v[u];
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static double timeit(F f, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e-3 / iters;
}

int main() {
  const size_t bytes = (size_t)2 << 30;  // 2 GiB >> 256 MiB Infinity Cache
  const size_t n16 = bytes / 16;
  u32x4 *src, *dst; uint32_t* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&sink, 4096));
  CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    double t1 = timeit([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(grid), dim3(256), 0, 0, src, sink, n16); }, 5);
    double t2 = timeit([&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(grid), dim3(256), 0, 0, src, sink, n16); }, 5);
    double t3 = timeit([&] { hipLaunchKernelGGL((read_kernel<8, true>), dim3(grid), dim3(256), 0, 0, src, sink, n16); }, 5);
    double t4 = timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, src, dst, n16); }, 5);
    printf("grid %5d: read u4 %6.0f GB/s | read u8 %6.0f | read u8 nt %6.0f | copy (r+w) %6.0f GB/s\n", grid,
           bytes / t1 / 1e9, bytes / t2 / 1e9, bytes / t3 / 1e9, 2.0 * bytes / t4 / 1e9);
  }
  const int waves = (int)(n16 / 1024);
  double t5 = timeit([&] { hipLaunchKernelGGL((read_runs_kernel<false>), dim3(waves / 4), dim3(256), 0, 0, src, sink, n16); }, 5);
  double t6 = timeit([&] { hipLaunchKernelGGL((read_runs_kernel<true>), dim3(waves / 4), dim3(256), 0, 0, src, sink, n16); }, 5);
  printf("one 16 KiB run per wave (no loop, %d workgroups): plain %6.0f GB/s | nt %6.0f GB/s\n", waves / 4, bytes / t5 / 1e9, bytes / t6 / 1e9);
  // small-launch regime: 128 MiB per launch (one layer's KV), back to back over distinct regions
  const size_t part = (size_t)128 << 20;
  double t7 = timeit([&] { for (int l = 0; l < 16; ++l) hipLaunchKernelGGL((read_runs_kernel<false>), dim3((int)(part / 16 / 1024 / 4)), dim3(256), 0, 0, src + l * (part / 16), sink, part / 16); }, 5);
  printf("128 MiB launches back to back: %6.0f GB/s (%.1f us per launch)\n", 16.0 * part / t7 / 1e9, t7 / 16 * 1e6);
  // the same 128 MiB region every launch: resident in the 256 MiB Infinity Cache after the first pass.  Can the cache
  // feed a launch of this size faster than HBM does (is a K/V prefetch of layer L + 1 under layer
  // L's latency-bound chain worth a second look)?
  for (int nt = 0; nt < 2; ++nt) {
    auto one = [&] {
      if (nt) hipLaunchKernelGGL((read_runs_kernel<true>), dim3((int)(part / 16 / 1024 / 4)), dim3(256), 0, 0, src, sink, part / 16);
      else hipLaunchKernelGGL((read_runs_kernel<false>), dim3((int)(part / 16 / 1024 / 4)), dim3(256), 0, 0, src, sink, part / 16);
    };
    double t8 = timeit([&] { for (int l = 0; l < 16; ++l) one(); }, 5);
    printf("128 MiB launches, SAME region (Infinity-Cache resident), %s: %6.0f GB/s (%.1f us per launch)\n",
           nt ? "nt   " : "plain", 16.0 * part / t8 / 1e9, t8 / 16 * 1e6);
  }
  // 32 MiB: resident in the eight 4 MiB L2s (each XCD sees 1/8 of the workgroups but any address: mostly NOT L2 hits)
  const size_t small = (size_t)32 << 20;
  double t9 = timeit([&] { for (int l = 0; l < 16; ++l) hipLaunchKernelGGL((read_runs_kernel<true>), dim3((int)(small / 16 / 1024 / 4)), dim3(256), 0, 0, src, sink, small / 16); }, 5);
  printf("32 MiB launches, same region: %6.0f GB/s (%.1f us per launch)\n", 16.0 * small / t9 / 1e9, t9 / 16 * 1e6);
  return 0;
}
