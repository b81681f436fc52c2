// Micro-benchmark of ONE pipeline stage of the 64-column prefill attention kernel (csrc/prefill64_body.inc): 256 workgroups
// of four waves (one per SIMD) run the stage N times on a resident LDS image - no DMA, optional barrier - and report
// shader cycles per stage (s_memtime).  Build + run: tools/ubench/prefill64_stage.sh
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
#ifndef BODY
#define BODY "../../nano-vllm-ascend_amd/csrc/prefill64_body.inc"
#endif
template <bool BARRIER>
__global__ __launch_bounds__(256) void stage_kernel(unsigned long long* cycles, float* sink, int n, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) uint16_t stage[];
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 5, nn = lane & 31;
  for (int i = threadIdx.x; i < 5 * 4 * 2048 / 2; i += 256) reinterpret_cast<uint32_t*>(stage)[i] = 0x3c003c00u + (i & 255);
  __syncthreads();
  const uint32_t lds_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)stage;
  bf16x8 Q[2][8], Qv[2][8]; for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 8; ++b_) Qv[a_][b_] = __builtin_bit_cast(bf16x8, u32x4{0x3c003c00u, (uint32_t)lane, 0x3c003c00u, 0x3c003c00u});
  const uint32_t qslot = lds_base + threadIdx.x * 256;
  for (int cb = 0; cb < 2; ++cb)
    for (int kk = 0; kk < 8; ++kk)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(Q[cb][kk]) : "v"(qslot), "i"((cb * 8 + kk) * 16) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x16 acc[2][4], S0[2], S1[2];
  for (int cb = 0; cb < 2; ++cb) {
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 16; ++i) acc[cb][j][i] = 0.f;
    for (int i = 0; i < 16; ++i) S0[cb][i] = S1[cb][i] = 0.01f * i;
  }
  u32x4 ph[2][2];
  for (int cb = 0; cb < 2; ++cb)
    for (int sg = 0; sg < 2; ++sg) ph[cb][sg] = u32x4{0, 0, 0, 0};
  float m[2] = {1.f, 1.f}, l[2] = {0.f, 0.f};
  const int k_off = (nn >> 4) * 2048 + (hi * 16 + (nn & 15)) * 8;
  const int v_off = 2 * 2048 + (hi * 16 + (nn & 15)) * 8 + (nn >> 4) * 4;
  const uint32_t kbase = lds_base + 2 * k_off, vbase = lds_base + 2 * v_off;
  auto body = [&](f32x16 (&Sc)[2], f32x16 (&Sx)[2], int c, int slot_k, int slot_v) __attribute__((always_inline)) {
    const uint32_t kaddr = kbase + slot_k * 32768, vaddr = vbase + slot_v * 32768;
    u32x4 Kf[8];
    uint64_t Vf[8][2];
    float p[2][16], tq[2][8], lc[2];
#include BODY
    for (int cb = 0; cb < 2; ++cb) {
      l[cb] += lc[cb];
      for (int i = 0; i < 8; ++i) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 v = {static_cast<__bf16>(p[cb][2 * i]), static_cast<__bf16>(p[cb][2 * i + 1])};
        ph[cb][i >> 2][i & 3] = __builtin_bit_cast(uint32_t, v);
      }
    }
  };
  int slot_k = 1, slot_v = 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < n; c += 2) {
    if (BARRIER) asm volatile("s_barrier" ::: "memory");
    body(S0, S1, c, slot_k, slot_v);
    slot_k = slot_k == 4 ? 0 : slot_k + 1;
    slot_v = slot_v == 4 ? 0 : slot_v + 1;
    if (BARRIER) asm volatile("s_barrier" ::: "memory");
    body(S1, S0, c + 1, slot_k, slot_v);
    slot_k = slot_k == 4 ? 0 : slot_k + 1;
    slot_v = slot_v == 4 ? 0 : slot_v + 1;
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = l[0] + l[1];
  for (int cb = 0; cb < 2; ++cb)
    for (int j = 0; j < 4; ++j)
      for (int i = 0; i < 16; ++i) s += acc[cb][j][i];
  for (int i = 0; i < 16; ++i) s += S0[0][i] + S0[1][i];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, wgs = argc > 2 ? atoi(argv[2]) : 256;
  unsigned long long* cyc;
  float* sink;
  hipMalloc(&cyc, wgs * 4 * 8);
  hipMalloc(&sink, wgs * 256 * 4);
  for (int barrier = 0; barrier < 2; ++barrier) {
    auto kern = barrier ? stage_kernel<true> : stage_kernel<false>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t a, b;
      hipEventCreate(&a);
      hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 81920, 0, cyc, sink, n, 0.1275f);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      unsigned long long h[4096];
      hipMemcpy(h, cyc, wgs * 4 * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0, sum = 0;
      for (int i = 0; i < wgs * 4; ++i) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
      printf("barrier %d rep %d: %d stages: avg %.0f  max %.0f s_memtime ticks per stage; launch %.1f us = %.0f ns per stage\n",
             barrier, rep, n, (double)sum / (wgs * 4) / n, (double)mx / n, ms * 1e3, ms * 1e6 / n);
    }
  }
  return 0;
}
