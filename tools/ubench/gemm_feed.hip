// Feasibility probe for a one-wave-per-SIMD GEMM main loop (256 x 256 x 64 tile, 4 waves x 512 VGPRs, wave tile
// 128 x 128): how many cycles does a K step take when the single wave of a SIMD has to issue, besides its 64
// v_mfma_f32_32x32x16_bf16, the fragment reads and the operand feed itself?  Timing only - the operands are whatever
// the feed puts into LDS; nothing is checked.
//   MODE 0  MFMAs only (64 per K step and wave)                              -> the matrix pipe's own pace
//   MODE 1  + 32 ds_read_b128 per K step (A and B fragments of the 128 x 128 wave tile)
//   MODE 2  + feed by LDS-DMA: 16 global_load_lds_dwordx4 per wave and K step (64 KiB per workgroup)
//   MODE 3  + feed through registers: 16 global_load_dwordx4 (top of the step) + 16 ds_write_b128 (behind the last MFMAs)
//   MODE 4  MODE 2 with the DMA pieces pinned one behind every fourth MFMA
// One barrier per K step, double-buffered 2 x 64 KiB LDS image; sources are eight 4 MiB streams (one per XCD: L2 hits
// after the first pass, like a GEMM's weight panel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int KSTEP_BYTES = 65536;

template <int MODE>
__global__ __launch_bounds__(256, 1) void feed_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * KSTEP_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const char* stream = src + (size_t)(blockIdx.x & 7) * ((size_t)ksteps * KSTEP_BYTES);
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment read offsets: row l31 of a 32-row block, 128-byte rows, 16-byte chunk (2 kk + hi) ^ swizzle
  int off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) off[kk] = (l31 >> 3) * 1024 + (l31 & 7) * 128 + (((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4);
  u32x4 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = u32x4{0x3f803f80u + (uint32_t)lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    B[i] = u32x4{0x3f803f80u, 0x3f803f80u + (uint32_t)wave, 0x3f803f80u, 0x3f803f80u};
  }
  u32x4 stage[16];
  for (int t = 0; t < ksteps; ++t) {
    const int p = t & 1;
    const char* cur = lds + p * KSTEP_BYTES;
    char* nxt = lds + (p ^ 1) * KSTEP_BYTES;
    const char* g = stream + (size_t)t * KSTEP_BYTES + wave * 16384 + lane * 16;
    if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE == 3) {  // this step's loads first: a whole K step of MFMAs to land in; written to LDS at the END of the step
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g + i * 1024));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          A[i] = *reinterpret_cast<const u32x4*>(cur + (wave >> 1) * 16384 + i * 4096 + off[kk]);
          B[i] = *reinterpret_cast<const u32x4*>(cur + 32768 + (wave & 1) * 16384 + i * 4096 + off[kk]);
        }
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (kk * 4 + i) * 1024),
                                           (__attribute__((address_space(3))) void*)(nxt + wave * 16384 + (kk * 4 + i) * 1024), 16, 0, 0);
      }
      if (MODE == 3 && kk == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[j]), acc[i][j], 0, 0, 0);
          if (MODE == 4) {  // one DMA piece behind every fourth MFMA, pinned there
            if ((i * 4 + j) % 4 == 3) {
              const int q = kk * 4 + i;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * 1024),
                                               (__attribute__((address_space(3))) void*)(nxt + wave * 16384 + q * 1024), 16, 0, 0);
              __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            }
          }
          if (MODE == 3 && kk == 3) {  // the step's sixteen LDS writes, one behind each MFMA of the last k group
            const int q = i * 4 + j;
            *reinterpret_cast<u32x4*>(nxt + wave * 16384 + q * 1024 + lane * 16) = stage[q];
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

template <int MODE>
static double run(const char* src, float* sink, int ksteps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / 5;  // us per launch
}

int main() {
  const int ksteps = 64;
  const size_t bytes = (size_t)8 * ksteps * KSTEP_BYTES;
  char* src; float* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 4096));
  // random bf16-ish data (DVFS: zero operands clock higher)
  uint32_t* h = (uint32_t*)malloc(bytes);
  uint32_t x = 12345;
  for (size_t i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x & 0x807f807fu) | 0x3f003f00u; }
  CK(hipMemcpy(src, h, bytes, hipMemcpyHostToDevice));
  const double flop = 2.0 * 256 * 256 * 64 * ksteps * 256;
  const double t0 = run<0>(src, sink, ksteps), t1 = run<1>(src, sink, ksteps), t2 = run<2>(src, sink, ksteps), t3 = run<3>(src, sink, ksteps);
  const double t4 = run<4>(src, sink, ksteps);
  printf("one wave per SIMD, 256 x 256 x 64 per workgroup and K step, %d K steps, 256 workgroups (us per launch, us per K step, TFLOP/s)\n", ksteps);
  printf("  MFMAs only                         %8.1f  %6.3f  %7.0f\n", t0, t0 / ksteps, flop / t0 / 1e6);
  printf("  + fragment reads                   %8.1f  %6.3f  %7.0f\n", t1, t1 / ksteps, flop / t1 / 1e6);
  printf("  + reads + LDS-DMA feed             %8.1f  %6.3f  %7.0f\n", t2, t2 / ksteps, flop / t2 / 1e6);
  printf("  + reads + register-staged feed     %8.1f  %6.3f  %7.0f\n", t3, t3 / ksteps, flop / t3 / 1e6);
  printf("  + reads + LDS-DMA, 1 per 4 MFMAs   %8.1f  %6.3f  %7.0f\n", t4, t4 / ksteps, flop / t4 / 1e6);
  return 0;
}
