// Types and helpers of the prefill attention kernel (paged_attn.hip), shared with the 64-columns-per-wave experiment
// (tools/ubench/prefill64/, not part of the library).
#pragma once
#include "mi_common.hpp"

namespace mi {

struct KvStrides {
  int64_t block, head, tile;
};
__host__ __device__ inline KvStrides default_strides(int n_kv_heads, int tpb, int tile_elems = MI_KV_TILE_ELEMS) {
  return KvStrides{(int64_t)n_kv_heads * tpb * tile_elems, (int64_t)tpb * tile_elems, tile_elems};
}

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float kDeferMax = 8.0f;  // prefill: rescale the running softmax only when a maximum grows by more than 2^8

// swap the upper half of `x` with the lower half of a copy: both halves then see (own, partner)
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// FUSE_Q: q points at the RAW q heads of the packed qkv rows; q-norm (if q_w) and RoPE are applied while the Q
// operand is loaded (the separate mi_qknorm_rope_store then handles K and V only and never writes q)
struct QPrep {
  const uint16_t* q_w;
  const int64_t* positions;
  const float* cos_sin;
  float eps;
};

}  // namespace mi
