// 8-lanes-per-head helpers shared by the elementwise kernels and the fused decode attention:
// q/k RMSNorm + NeoX RoPE with the reference's rounding points, and the scatter of one token's
// K / V head into the fragment-native cache tiles.  The arithmetic helpers carry
// `#pragma clang fp contract(off)` (separate fp32 mul / add roundings, as torch's unfused ops), so
// they give the same bits in translation units built with or without -ffp-contract=off.
#pragma once
#include "mi_common.hpp"

namespace mi {

__device__ __forceinline__ void load16(const uint16_t* p, float (&f)[8]) {
  const u32x4 raw = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = lo_bf(raw[j]);
    f[2 * j + 1] = hi_bf(raw[j]);
  }
}
__device__ __forceinline__ u32x4 pack16(const float (&f)[8]) {
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack_bf(f[2 * j], f[2 * j + 1]);
  return o;
}

// rms-normalise the 128 values held by 8 lanes (16 each) with weight w
__device__ __forceinline__ void head_rmsnorm(float (&a)[8], float (&b)[8], const uint16_t* w, int j,
                                             float eps) {
#pragma clang fp contract(off)
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += a[i] * a[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += b[i] * b[i];
  ss = xor_sum<1>(ss);  // (DPP partners: the bits of the __shfl_xor butterfly 1, 2, 4 - mi_common.hpp)
  ss = xor_sum<2>(ss);
  ss = xor_sum<4>(ss);
  const float rs = 1.0f / sqrtf(ss / 128.0f + eps);
  float wa[8], wb[8];
  load16(w + 8 * j, wa);
  load16(w + 64 + 8 * j, wb);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = rbf(rbf(a[i] * rs) * wa[i]);
    b[i] = rbf(rbf(b[i] * rs) * wb[i]);
  }
}

// NeoX rotation in fp32 (rotary_embedding.py:6-14): separate mul / sub / add
__device__ __forceinline__ void head_rope(float (&a)[8], float (&b)[8], const float* cs, int j) {
#pragma clang fp contract(off)
  const float4 c0 = *reinterpret_cast<const float4*>(cs + 8 * j);
  const float4 c1 = *reinterpret_cast<const float4*>(cs + 8 * j + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(cs + 64 + 8 * j);
  const float4 s1 = *reinterpret_cast<const float4*>(cs + 64 + 8 * j + 4);
  const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x1 = a[i], x2 = b[i];
    a[i] = rbf(x1 * c[i] - x2 * s[i]);
    b[i] = rbf(x2 * c[i] + x1 * s[i]);
  }
}

// ---- the same arithmetic on the MFMA-operand distribution of a head --------------------------
// Four lanes g = 0..3 (lane ids n, n+16, n+32, n+48 of a wavefront) hold one head: lane g owns
// dims 8g + 32kk + e (kk < 4, e < 8) in x[kk][e] - the B fragment of K.Q^T.  Lane g is the union of
// the 8-lane form's lanes j = g (x[0] = its a, x[2] = its b) and j = g + 4 (x[1], x[3]); the sums
// below are associated exactly as head_rmsnorm's (lane sums, then the xor-1/2/4 tree), so both
// forms give identical bits.
__device__ __forceinline__ void head_rmsnorm_frag(float (&x)[4][8], const uint16_t* w, int g, float eps) {
#pragma clang fp contract(off)
  float lo = 0.f, hi = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) lo += x[0][i] * x[0][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) lo += x[2][i] * x[2][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) hi += x[1][i] * x[1][i];
#pragma unroll
  for (int i = 0; i < 8; ++i) hi += x[3][i] * x[3][i];
  lo += __shfl_xor(lo, 16, 64);
  lo += __shfl_xor(lo, 32, 64);
  hi += __shfl_xor(hi, 16, 64);
  hi += __shfl_xor(hi, 32, 64);
  const float ss = lo + hi;
  const float rs = 1.0f / sqrtf(ss / 128.0f + eps);
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    float wv[8];
    load16(w + 8 * g + 32 * kk, wv);
#pragma unroll
    for (int i = 0; i < 8; ++i) x[kk][i] = rbf(rbf(x[kk][i] * rs) * wv[i]);
  }
}

// NeoX pairs (d, d + 64) = (x[0], x[2]) with cos/sin index 8g + e and (x[1], x[3]) with 32 + 8g + e
__device__ __forceinline__ void head_rope_frag(float (&x)[4][8], const float* cs, int g) {
#pragma clang fp contract(off)
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float* cp = cs + 32 * p + 8 * g;
    const float4 c0 = *reinterpret_cast<const float4*>(cp);
    const float4 c1 = *reinterpret_cast<const float4*>(cp + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(cp + 64);
    const float4 s1 = *reinterpret_cast<const float4*>(cp + 68);
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x1 = x[p][i], x2 = x[p + 2][i];
      x[p][i] = rbf(x1 * c[i] - x2 * s[i]);
      x[p + 2][i] = rbf(x2 * c[i] + x1 * s[i]);
    }
  }
}

// ---- and on the 32x32-MFMA B-operand distribution of the prefill attention's Q --------------------
// Two lanes hi = 0 / 1 (lane ids n and n + 32) hold one head: lane hi owns dims 16 kk + 8 hi + e (kk < 8, e < 8)
// in x[kk][e], i.e. the 8-dim groups 2 kk + hi.  Group j < 8 is the 8-lane form's a of lane j, group 8 + j its b:
// lane hi holds (a, b) = (x[m], x[m + 4]) of the 8-lane lanes j = 2 m + hi.  Sums are associated as head_rmsnorm's
// (per lane a then b, then the xor-1 / 2 / 4 tree: the xor-1 partner sits in the other lane), so the bits agree.
// the table values a lane needs: c[m] / sn[m] = cos / sin of the 8-dim group 2 m + hi (cos_sin row: 64 cos, then 64 sin)
struct RopeRegs32 {
  float4 c[4][2], s[4][2];
};
__device__ __forceinline__ void rope_regs_q32_load(RopeRegs32& r, const float* cs, int hi) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float* cp = cs + 8 * (2 * m + hi);
    r.c[m][0] = *reinterpret_cast<const float4*>(cp);
    r.c[m][1] = *reinterpret_cast<const float4*>(cp + 4);
    r.s[m][0] = *reinterpret_cast<const float4*>(cp + 64);
    r.s[m][1] = *reinterpret_cast<const float4*>(cp + 68);
  }
}
// the arithmetic of head_rmsnorm_rope_q32 on table values that are already in registers (the tile GEMM's qkv
// epilogue requests them a token block ahead)
__device__ __forceinline__ void head_rmsnorm_rope_q32_regs(float (&x)[8][8], const uint16_t* w, const RopeRegs32& r, int hi,
                                                           float eps) {
#pragma clang fp contract(off)
  if (w != nullptr) {
    float t[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += x[m][i] * x[m][i];
#pragma unroll
      for (int i = 0; i < 8; ++i) ss += x[m + 4][i] * x[m + 4][i];
      t[m] = ss + __shfl_xor(ss, 32, 64);  // s_(2m) + s_(2m+1)
    }
    const float ss = (t[0] + t[1]) + (t[2] + t[3]);
    const float rs = 1.0f / sqrtf(ss / 128.0f + eps);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float wa[8], wb[8];
      load16(w + 8 * (2 * m + hi), wa);
      load16(w + 64 + 8 * (2 * m + hi), wb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        x[m][i] = rbf(rbf(x[m][i] * rs) * wa[i]);
        x[m + 4][i] = rbf(rbf(x[m + 4][i] * rs) * wb[i]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float4 c0 = r.c[m][0], c1 = r.c[m][1], s0 = r.s[m][0], s1 = r.s[m][1];
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x1 = x[m][i], x2 = x[m + 4][i];
      x[m][i] = rbf(x1 * c[i] - x2 * sn[i]);
      x[m + 4][i] = rbf(x2 * c[i] + x1 * sn[i]);
    }
  }
}
__device__ __forceinline__ void head_rmsnorm_rope_q32(float (&x)[8][8], const uint16_t* w, const float* cs, int hi,
                                                      float eps) {
  RopeRegs32 r;
  rope_regs_q32_load(r, cs, hi);
  head_rmsnorm_rope_q32_regs(x, w, r, hi, eps);
}

// The same arithmetic once more, on the head held as PACKED bf16 (xp[kk] = dims 16 kk + 8 hi .. + 7, i.e. x[kk][0..7] of
// the form above) and group by group, so that only 16 values are unpacked at a time: the tile GEMM's qkv epilogue
// has ~190 registers for everything.  Every value of x above is a bf16 (the inputs are GEMM outputs rounded to bf16,
// every result is rbf()'d): packing and unpacking are exact, the operations and their order are the ones above.
__device__ __forceinline__ void head_rmsnorm_rope_q32_packed(u32x4 (&xp)[8], const uint16_t* w, const RopeRegs32& r, int hi,
                                                             float eps) {
#pragma clang fp contract(off)
  float rs = 1.0f;
  if (w != nullptr) {
    float t[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = lo_bf(xp[m][i]), hh = hi_bf(xp[m][i]);
        ss += lo * lo;
        ss += hh * hh;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float lo = lo_bf(xp[m + 4][i]), hh = hi_bf(xp[m + 4][i]);
        ss += lo * lo;
        ss += hh * hh;
      }
      t[m] = ss + __shfl_xor(ss, 32, 64);
    }
    const float ss = (t[0] + t[1]) + (t[2] + t[3]);
    rs = 1.0f / sqrtf(ss / 128.0f + eps);
  }
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    float xa[8], xb[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[2 * i] = lo_bf(xp[m][i]);
      xa[2 * i + 1] = hi_bf(xp[m][i]);
      xb[2 * i] = lo_bf(xp[m + 4][i]);
      xb[2 * i + 1] = hi_bf(xp[m + 4][i]);
    }
    if (w != nullptr) {
      float wa[8], wb[8];
      load16(w + 8 * (2 * m + hi), wa);
      load16(w + 64 + 8 * (2 * m + hi), wb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xa[i] = rbf(rbf(xa[i] * rs) * wa[i]);
        xb[i] = rbf(rbf(xb[i] * rs) * wb[i]);
      }
    }
    const float4 c0 = r.c[m][0], c1 = r.c[m][1], s0 = r.s[m][0], s1 = r.s[m][1];
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    float ya[8], yb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x1 = xa[i], x2 = xb[i];
      ya[i] = rbf(x1 * c[i] - x2 * sn[i]);
      yb[i] = rbf(x2 * c[i] + x1 * sn[i]);
    }
    xp[m] = pack16(ya);
    xp[m + 4] = pack16(yb);
  }
}

__device__ __forceinline__ bool resolve_slot(const int32_t* slots, int slot_is_2d, int token,
                                             int block_size, int64_t& blk, int& off) {
  if (slot_is_2d) {
    blk = slots[2 * token];
    off = slots[2 * token + 1];
    return blk >= 0 && off >= 0;
  }
  const int32_t s = slots[token];
  if (s < 0) return false;
  blk = s / block_size;
  off = s % block_size;
  return true;
}

__device__ __forceinline__ void store_k_head(uint16_t* k_cache, int64_t blk, int off, int h, int j,
                                             const float (&a)[8], const float (&b)[8], int n_kv_heads,
                                             int tpb) {
  uint16_t* tile = k_cache + kv_tile_base(blk, h, off, n_kv_heads, tpb);
  const int t = off & 15;
  // chunk j: d = 8j ; chunk j+8: d = 64 + 8j
  *reinterpret_cast<u32x4*>(tile + k_tile_off(t, 8 * j)) = pack16(a);
  *reinterpret_cast<u32x4*>(tile + k_tile_off(t, 64 + 8 * j)) = pack16(b);
}
__device__ __forceinline__ void store_v_head(uint16_t* v_cache, int64_t blk, int off, int h, int j,
                                             const float (&a)[8], const float (&b)[8], int n_kv_heads,
                                             int tpb) {
  uint16_t* tile = v_cache + kv_tile_base(blk, h, off, n_kv_heads, tpb);
  const int t = off & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    tile[v_tile_off(t, 8 * j + i)] = f2bf(a[i]);
    tile[v_tile_off(t, 64 + 8 * j + i)] = f2bf(b[i]);
  }
}


}  // namespace mi
