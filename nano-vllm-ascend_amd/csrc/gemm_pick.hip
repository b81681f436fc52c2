// The head GEMM of a decode step with the sampler folded in (embed_head.py:57-61 + sampler.py:9-17 of the
// reference): gemm_skinny_kernel's EPI_PICK epilogue + the per-row reduction of its candidates.
#include "gemm_skinny_kernel.hpp"

namespace mi {

// token of every row from the candidates of EPI_PICK: one 1024-thread workgroup per row.  The candidates were
// written by other CUs a moment ago (every load is a trip to memory), so a thread issues eight loads before it
// compares anything: one exposed latency per 8192 candidates instead of one per candidate.
__global__ __launch_bounds__(1024) void pick_final_kernel(const uint2* __restrict__ cand, int n_groups, int M,
                                                          int64_t* __restrict__ out, uint2* __restrict__ pairs) {
  const int row = blockIdx.x;
  const uint2* p = cand + (int64_t)row * n_groups;
  float best = -INFINITY;
  int best_c = 0x7fffffff;
  for (int base = 0; base < n_groups; base += 8 * 1024) {
    uint2 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gidx = base + i * 1024 + (int)threadIdx.x;
      c[i] = p[min(gidx, n_groups - 1)];  // past the end: the last candidate again (harmless under max)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float k2 = __uint_as_float(c[i].x);
      const int c2 = (int)c[i].y;
      if (k2 > best || (k2 == best && c2 < best_c)) {
        best = k2;
        best_c = c2;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oc = __shfl_xor(best_c, o, 64);
    if (ob > best || (ob == best && oc < best_c)) {
      best = ob;
      best_c = oc;
    }
  }
  __shared__ float sb[16];
  __shared__ int sc[16];
  if ((threadIdx.x & 63) == 0) {
    sb[threadIdx.x >> 6] = best;
    sc[threadIdx.x >> 6] = best_c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (sb[w] > best || (sb[w] == best && sc[w] < best_c)) {
        best = sb[w];
        best_c = sc[w];
      }
    if (pairs) pairs[row] = uint2{__float_as_uint(best), (uint32_t)best_c};  // this rank's best (key, global column)
    else out[row] = best_c == 0x7fffffff ? 0 : best_c;
  }
}

}  // namespace mi

using namespace mi;

// head GEMM with the pick epilogue: the same workgroup geometry as mi_gemm_bf16_packed for this (M, N)
extern "C" int mi_gemm_pick_groups(int M, int N, int K, int fp8_weights) {
  if (M <= 0 || N <= 0 || N % 16) return 0;
  if (!fp8_weights && head_stream_fits(M, N, K)) return head_stream_grid(M);
  return pick_two_tiles(M, N) ? N / 32 : N / 16;
}

static int packed_pick(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K,
                       const float* temperatures, const uint64_t* rng, void* candidates, int col_offset,
                       mi_stream stream) {
  int rc = check_gemm(x, w_packed, y, M, N, K);
  if (rc != MI_OK) return rc;
  if (!rng || !candidates || col_offset < 0) return MI_EINVAL;
  if (M == 0) return MI_OK;
  const PickArgs pk{temperatures, rng, static_cast<uint2*>(candidates), col_offset};
  if (head_stream_fits(M, N, K)) {
    launch_head_stream<true>(x, w_packed, y, M, N, K, pk, S(stream));
    return check_launch();
  }
  GemmArgs a{x, w_packed, nullptr, y, nullptr, M, N, K, 1, S(stream)};
  a.pick = pk;
  return pick_two_tiles(M, N) ? pick_mt<2, 1, EPI_PICK>(a) : pick_mt<1, 1, EPI_PICK>(a);
}

extern "C" int mi_gemm_bf16_packed_pick(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K,
                                        const float* temperatures, const uint64_t* rng, void* candidates,
                                        mi_stream stream) {
  return packed_pick(x, w_packed, y, M, N, K, temperatures, rng, candidates, 0, stream);
}

extern "C" int mi_gemm_bf16_packed_pick_shard(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, int M, int N,
                                              int K, const float* temperatures, const uint64_t* rng,
                                              void* candidates, int col_offset, mi_stream stream) {
  return packed_pick(x, w_packed, y, M, N, K, temperatures, rng, candidates, col_offset, stream);
}

extern "C" int mi_gemm_fp8w_packed_pick(const mi_bf16* x, const uint8_t* w_packed, const float* scale, mi_bf16* y,
                                        int M, int N, int K, const float* temperatures, const uint64_t* rng,
                                        void* candidates, mi_stream stream) {
  int rc = check_gemm(x, w_packed, y, M, N, K);
  if (rc != MI_OK) return rc;
  if (!scale || !aligned16(scale) || !rng || !candidates) return MI_EINVAL;
  if (K % 64) return MI_EUNSUPPORTED;
  if (M == 0) return MI_OK;
  GemmArgs a{x, reinterpret_cast<const uint16_t*>(w_packed), nullptr, y, nullptr, M, N, K, 1, S(stream)};
  a.scale = scale;
  a.pick = PickArgs{temperatures, rng, static_cast<uint2*>(candidates), 0};
  return pick_two_tiles(M, N) ? pick_mt<2, 2, EPI_PICK>(a) : pick_mt<1, 2, EPI_PICK>(a);
}

extern "C" int mi_pick_final(const void* candidates, int n_groups, int rows, int64_t* out, mi_stream stream) {
  if (!candidates || !out || n_groups <= 0 || rows < 0) return MI_EINVAL;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(pick_final_kernel, dim3(rows), dim3(1024), 0, S(stream), static_cast<const uint2*>(candidates),
                     n_groups, rows, out, static_cast<uint2*>(nullptr));
  return check_launch();
}

extern "C" int mi_pick_final_pairs(const void* candidates, int n_groups, int rows, void* pairs, mi_stream stream) {
  if (!candidates || !pairs || n_groups <= 0 || rows < 0) return MI_EINVAL;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(pick_final_kernel, dim3(rows), dim3(1024), 0, S(stream), static_cast<const uint2*>(candidates),
                     n_groups, rows, static_cast<int64_t*>(nullptr), static_cast<uint2*>(pairs));
  return check_launch();
}

