// Shared device/host helpers for the gfx950 kernels of libmi355_nanovllm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mi355_nanovllm.h"

namespace mi {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA 16x16x32 A/B fragment
typedef __attribute__((ext_vector_type(4))) float f32x4;     // one MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

// ---- bf16 <-> f32, round-to-nearest-even (same rule as c10::BFloat16) ------
__device__ __forceinline__ float bf2f(uint16_t h) {
  return __uint_as_float(static_cast<uint32_t>(h) << 16);
}
// float -> bf16 is the hardware conversion (v_cvt_pk_bf16_f32: round to nearest even, the
// rule of c10::BFloat16) through the compiler's native __bf16 casts
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float lo, float hi) {
  const bf16x2 v = {static_cast<__bf16>(lo), static_cast<__bf16>(hi)};
  return __builtin_bit_cast(uint32_t, v);
}
// round a float to the nearest bf16 and return it as float
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ bf16x8 as_frag(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// ---- wave-level reductions (64 lanes) --------------------------------------
// xor_get<O>(v): the value of lane ^ O (O = 1, 2, 4, 8), on the DPP path of the VALU; xor_pair<O>(v, a, b) for O = 16,
// 32: a and b = (own, partner) in some order (v_permlane16_swap / v_permlane32_swap) - enough for commutative
// reductions.  The SAME partners in the SAME order as a `__shfl_xor` butterfly, so sums and maxima keep their bits
// (x + y == y + x), at a few VALU cycles per step: a __shfl_xor is a ds_bpermute - ~100 cycles of LDS crossbar latency
// per dependent step, six steps per wave reduction, on the critical path of every decode-sized norm launch.
// All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int O>
__device__ __forceinline__ float xor_get(float v) {
  static_assert(O == 1 || O == 2 || O == 4 || O == 8, "in-row partners");
  if (O == 1) return dpp_mov<0xB1>(v);                 // quad_perm [1, 0, 3, 2]
  if (O == 2) return dpp_mov<0x4E>(v);                 // quad_perm [2, 3, 0, 1]
  if (O == 4) return dpp_mov<0x1B>(dpp_mov<0x141>(v));  // row_half_mirror (i -> 7 - i), then quad_perm [3, 2, 1, 0]
  return dpp_mov<0x128>(v);                            // row_ror:8
}
template <int O>
__device__ __forceinline__ void xor_pair(float v, float& a, float& b) {
  static_assert(O == 16 || O == 32, "cross-row partners");
  if (O == 16) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
  }
}
template <int O>
__device__ __forceinline__ float xor_sum(float v) {  // v + (value of lane ^ O), the bits of v + __shfl_xor(v, O)
  if constexpr (O >= 16) {
    float a, b;
    xor_pair<O>(v, a, b);
    return a + b;
  } else {
    return v + xor_get<O>(v);
  }
}
template <int O>
__device__ __forceinline__ float xor_max(float v) {
  if constexpr (O >= 16) {
    float a, b;
    xor_pair<O>(v, a, b);
    return fmaxf(a, b);
  } else {
    return fmaxf(v, xor_get<O>(v));
  }
}
__device__ __forceinline__ float wave_sum(float v) {  // the butterfly 32, 16, 8, 4, 2, 1
  v = xor_sum<32>(v);
  v = xor_sum<16>(v);
  v = xor_sum<8>(v);
  v = xor_sum<4>(v);
  v = xor_sum<2>(v);
  return xor_sum<1>(v);
}
// the same sum as a tree that starts with the NEAREST partners (1, 2, ..., 32): lanes 4 j .. 4 j + 3 add up first - the
// grouping the five-launch decode chain's producers can deliver per 16-feature tile (gemm_chain5_kernel.hpp)
__device__ __forceinline__ float wave_sum_up(float v) {
  v = xor_sum<1>(v);
  v = xor_sum<2>(v);
  v = xor_sum<4>(v);
  v = xor_sum<8>(v);
  v = xor_sum<16>(v);
  return xor_sum<32>(v);
}
__device__ __forceinline__ float wave_max(float v) {
  v = xor_max<32>(v);
  v = xor_max<16>(v);
  v = xor_max<8>(v);
  v = xor_max<4>(v);
  v = xor_max<2>(v);
  return xor_max<1>(v);
}

// ---- counter-based random numbers of the sampler (mi_sample, and the pick epilogue of the head GEMM) ----
__device__ __forceinline__ uint32_t mix32(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

__device__ __forceinline__ float gumbel_key(float logit, float inv_t, uint64_t rkey, int col) {
#pragma clang fp contract(off)  // the same bits in every translation unit
  // u in (0,1): 24 random bits, never 0 or 1
  const float u = ((float)(mix32(rkey + (uint64_t)col) >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return logit * inv_t - logf(-logf(u));
}

// key of one sampling row: (seed, step, row) -> 64 bits mixed per column by gumbel_key
__device__ __forceinline__ uint64_t sample_row_key(uint64_t seed, uint64_t step, int row) {
  return seed * 0x9e3779b97f4a7c15ull + step * 0xd1342543de82ef95ull + (uint64_t)row * 0x2545f4914f6cdd1dull;
}

// ---- fragment-native KV tile addressing (see include/mi355_nanovllm.h) -----
// element offset of (t, d) inside a 16-token x 128-dim K tile
__host__ __device__ __forceinline__ int k_tile_off(int t, int d) {
  return (d >> 5) * 512 + ((((d >> 3) & 3) * 16 + t) << 3) + (d & 7);
}
// element offset of (t, d) inside a V tile (token-transposed)
__host__ __device__ __forceinline__ int v_tile_off(int t, int d) {
  return (d >> 5) * 512 + ((((t >> 2) * 16) + (d & 15)) << 3) + (((d >> 4) & 1) << 2) + (t & 3);
}
// element offset of the tile holding `slot_in_block` of kv head `h` in block `blk`
__host__ __device__ __forceinline__ int64_t kv_tile_base(int64_t blk, int h, int slot_in_block,
                                                         int n_kv_heads, int tiles_per_block) {
  return ((blk * n_kv_heads + h) * tiles_per_block + (slot_in_block >> 4)) * (int64_t)MI_KV_TILE_ELEMS;
}

// ---- host-side launch bookkeeping ------------------------------------------
int tuning(int knob);        // current value of a mi_tuning_knob (c_api.hip; the library never reads the environment)
int check_launch();          // returns MI_OK or MI_ELAUNCH (records hipGetLastError text)
inline hipStream_t S(mi_stream s) { return reinterpret_cast<hipStream_t>(s); }
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace mi
