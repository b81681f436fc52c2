// Elementwise / gather / scatter kernels of the decode path for gfx950.
// All of them are HBM- or latency-bound: 16-byte vector accesses, one rounding
// point per reference rounding point (compiled with -ffp-contract=off so the
// fp32 sequences match the reference's unfused torch ops bit for bit).
#include <stdlib.h>

#include "mi_common.hpp"
#include "kv_store.hpp"

namespace mi {

// ---------------------------------------------------------------------------
// RMSNorm / add+RMSNorm   (reference: layers/layernorm.py:16-38)
//   LPR lanes cooperate on one row, each holding VPL vectors of 8 bf16.
// ---------------------------------------------------------------------------
template <int LPR, int VPL, bool ADD, int PART>  // PART = number of split-K partials (0: x is bf16)
__global__ __launch_bounds__(256) void rmsnorm_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ part, int nsplit, int64_t x_outer_stride, int inner,
    const uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
    uint16_t* __restrict__ y, uint16_t* __restrict__ residual_out,
    int rows, int cols, float eps) {
  constexpr int ROWS_PER_WAVE = WAVE / LPR;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = wave * ROWS_PER_WAVE + lane / LPR;
  const int sub = lane % LPR;
  const bool active = row < rows;
  const int nvec = cols >> 3;

  const int64_t xoff = active ? (int64_t)(row / inner) * x_outer_stride + (int64_t)(row % inner) * cols : 0;
  const int64_t yoff = (int64_t)row * cols;

  float v[VPL][8];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vec = sub + i * LPR;
    if (active && vec < nvec) {
      u32x4 raw;
      if (PART > 0) {
        // x is the bf16 rounding of the split-K GEMM result: all partial loads are issued first,
        // then summed in split order
        f32x4 plo[PART > 0 ? PART : 1], phi[PART > 0 ? PART : 1];
#pragma unroll
        for (int sp = 0; sp < PART; ++sp) {
          const float* pp = part + ((int64_t)sp * rows + row) * cols + vec * 8;
          plo[sp] = *reinterpret_cast<const f32x4*>(pp);
          phi[sp] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
        f32x4 lo = plo[0], hi = phi[0];
#pragma unroll
        for (int sp = 1; sp < PART; ++sp) {
          lo += plo[sp];
          hi += phi[sp];
        }
        raw = u32x4{pack_bf(lo[0], lo[1]), pack_bf(lo[2], lo[3]), pack_bf(hi[0], hi[1]), pack_bf(hi[2], hi[3])};
      } else {
        raw = *reinterpret_cast<const u32x4*>(x + xoff + vec * 8);
      }
      u32x4 rr = {0, 0, 0, 0};
      if (ADD) rr = *reinterpret_cast<const u32x4*>(residual + yoff + vec * 8);
      u32x4 ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = lo_bf(raw[j]), b = hi_bf(raw[j]);
        if (ADD) {
          a = a + lo_bf(rr[j]);
          b = b + hi_bf(rr[j]);
          ro[j] = pack_bf(a, b);
        }
        v[i][2 * j] = a;
        v[i][2 * j + 1] = b;
        ss += a * a;
        ss += b * b;
      }
      if (ADD) *reinterpret_cast<u32x4*>(residual_out + yoff + vec * 8) = ro;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  // the butterfly LPR / 2 .. 1 over the row's lanes (DPP / permlane partners, mi_common.hpp: the __shfl_xor bits)
  if (LPR >= 64) ss = xor_sum<32>(ss);
  if (LPR >= 32) ss = xor_sum<16>(ss);
  if (LPR >= 16) ss = xor_sum<8>(ss);
  if (LPR >= 8) ss = xor_sum<4>(ss);
  if (LPR >= 4) ss = xor_sum<2>(ss);
  if (LPR >= 2) ss = xor_sum<1>(ss);
  const float rs = 1.0f / sqrtf(ss / (float)cols + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int vec = sub + i * LPR;
    if (active && vec < nvec) {
      const u32x4 wr = *reinterpret_cast<const u32x4*>(w + vec * 8);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = rbf(v[i][2 * j] * rs) * lo_bf(wr[j]);
        const float b = rbf(v[i][2 * j + 1] * rs) * hi_bf(wr[j]);
        o[j] = pack_bf(a, b);
      }
      *reinterpret_cast<u32x4*>(y + yoff + vec * 8) = o;
    }
  }
}

// Decode-sized split-K consumer (<= 64 rows): the same arithmetic with WPR waves per row, so a lane
// has 1/WPR of the partial loads in flight (a row of 1024 columns with 4 splits is 16 fragment loads per
// lane for one wave, 4 for four).  Sum of squares: lane partials -> wave shuffle tree -> the WPR wave
// sums added in wave order through LDS.
// PART = 0: x is a finished bf16 tensor (`xb`: mi_add_rmsnorm for a few rows of many columns - hidden 4096 / 5120 models,
// whose rows one wave per row walks in ten dependent-looking 16-byte loads per operand).
template <int PART, int WPR>
__global__ __launch_bounds__(WPR * 64) void add_rmsnorm_splitk_rows_kernel(
    const float* __restrict__ part, const uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
    uint16_t* __restrict__ y, uint16_t* __restrict__ residual_out, int rows, int cols, float eps,
    const uint16_t* __restrict__ xb = nullptr) {
  __shared__ float wave_ss[WPR];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = cols >> 3;
  const int64_t off = (int64_t)row * cols;
  constexpr int MAXV = 2;  // vectors per lane: cols <= WPR * 64 * 8 * MAXV
  float v[MAXV][8];
  u32x4 wr[MAXV];  // the norm weight is fetched with everything else, not behind the barrier
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vec = tid + i * WPR * 64;
    if (vec < nvec) {
      u32x4 raw;
      u32x4 rr;
      if constexpr (PART == 0) {
        raw = *reinterpret_cast<const u32x4*>(xb + off + vec * 8);
        rr = *reinterpret_cast<const u32x4*>(residual + off + vec * 8);
        wr[i] = *reinterpret_cast<const u32x4*>(w + vec * 8);
      } else {
        f32x4 plo[PART > 0 ? PART : 1], phi[PART > 0 ? PART : 1];
#pragma unroll
        for (int sp = 0; sp < PART; ++sp) {
          const float* pp = part + ((int64_t)sp * rows + row) * cols + vec * 8;
          plo[sp] = *reinterpret_cast<const f32x4*>(pp);
          phi[sp] = *reinterpret_cast<const f32x4*>(pp + 4);
        }
        rr = *reinterpret_cast<const u32x4*>(residual + off + vec * 8);
        wr[i] = *reinterpret_cast<const u32x4*>(w + vec * 8);
        f32x4 lo = plo[0], hi = phi[0];
#pragma unroll
        for (int sp = 1; sp < PART; ++sp) {
          lo += plo[sp];
          hi += phi[sp];
        }
        raw = u32x4{pack_bf(lo[0], lo[1]), pack_bf(lo[2], lo[3]), pack_bf(hi[0], hi[1]), pack_bf(hi[2], hi[3])};
      }
      u32x4 ro;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = lo_bf(raw[j]) + lo_bf(rr[j]);
        const float b = hi_bf(raw[j]) + hi_bf(rr[j]);
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
}

// The same for rows of at most WPR * 64 * 4 columns (hidden 1024 with four waves): FOUR columns per thread, so
// that every thread of the workgroup has work - PART + 2 loads per lane, all issued before the first use.
// Same arithmetic and summation order per element; the sum of squares adds the lanes' partial sums in a
// different grouping than the 8-wide form (both are fp32 sums of the same squares).
// STAMP (mi_add_rmsnorm_splitk_ex, tools/chain_timeline.py): every wave records s_memrealtime (the chip-wide 100 MHz
// clock) at seven points - entry / loads issued / data arrived / sum of squares in LDS / barrier passed / stores issued /
// stores acknowledged - into stamps[row][wave][8].  A separate instantiation; the product kernel carries no stamp code.
template <int PART, int WPR, bool STAMP = false>
__global__ __launch_bounds__(WPR * 64) void add_rmsnorm_splitk_rows4_kernel(
    const float* __restrict__ part, const uint16_t* __restrict__ residual, const uint16_t* __restrict__ w,
    uint16_t* __restrict__ y, uint16_t* __restrict__ residual_out, int rows, int cols, float eps,
    unsigned long long* __restrict__ stamps = nullptr) {
  unsigned long long ts[8] = {};
#define MI_NSTAMP(i)                                    \
  do {                                                  \
    if constexpr (STAMP) {                              \
      __builtin_amdgcn_sched_barrier(0);                \
      ts[i] = __builtin_amdgcn_s_memrealtime();         \
      __builtin_amdgcn_sched_barrier(0);                \
    }                                                   \
  } while (0)
  MI_NSTAMP(0);
  __shared__ float wave_ss[WPR];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nvec = cols >> 2;
  const int64_t off = (int64_t)row * cols;
  const bool on = tid < nvec;
  const int vec = on ? tid : 0;
  f32x4 p[PART];
#pragma unroll
  for (int sp = 0; sp < PART; ++sp)
    p[sp] = *reinterpret_cast<const f32x4*>(part + ((int64_t)sp * rows + row) * cols + vec * 4);
  const u32x2 rr = *reinterpret_cast<const u32x2*>(residual + off + vec * 4);
  const u32x2 wr = *reinterpret_cast<const u32x2*>(w + vec * 4);
  if constexpr (STAMP) {
    MI_NSTAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_NSTAMP(2);
  }
  f32x4 s = p[0];
#pragma unroll
  for (int sp = 1; sp < PART; ++sp) s += p[sp];
  const u32x2 raw = {pack_bf(s[0], s[1]), pack_bf(s[2], s[3])};
  float v[4];
  u32x2 ro;
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = lo_bf(raw[j]) + lo_bf(rr[j]);
    const float b = hi_bf(raw[j]) + hi_bf(rr[j]);
    ro[j] = pack_bf(a, b);
    v[2 * j] = a;
    v[2 * j + 1] = b;
    ss += a * a;
    ss += b * b;
  }
  if (on) *reinterpret_cast<u32x2*>(residual_out + off + vec * 4) = ro;
  // nearest partners first: a 16-column tile's four threads add up before anything else, which is the partial sum the
  // five-launch chain's projections emit per tile (gemm_chain5_kernel.hpp: the two chains agree bit for bit)
  ss = wave_sum_up(on ? ss : 0.f);
  if (lane == 0) wave_ss[wave] = ss;
  MI_NSTAMP(3);
  __syncthreads();
  MI_NSTAMP(4);
  float tot = wave_ss[0];
#pragma unroll
  for (int wv = 1; wv < WPR; ++wv) tot += wave_ss[wv];
  const float rs = 1.0f / sqrtf(tot / (float)cols + eps);
  u32x2 o;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = rbf(v[2 * j] * rs) * lo_bf(wr[j]);
    const float b = rbf(v[2 * j + 1] * rs) * hi_bf(wr[j]);
    o[j] = pack_bf(a, b);
  }
  if (on) *reinterpret_cast<u32x2*>(y + off + vec * 4) = o;
  if constexpr (STAMP) {
    MI_NSTAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_NSTAMP(6);
    if (stamps != nullptr && lane == 0) {
      unsigned long long* dst = stamps + ((int64_t)row * WPR + wave) * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = ts[q];
    }
  }
#undef MI_NSTAMP
}

template <bool ADD, int PART = 0>
static int launch_rmsnorm(const uint16_t* x, const float* part, int nsplit, int64_t xs, int inner, const uint16_t* r,
                          const uint16_t* w, uint16_t* y, uint16_t* ro, int rows, int cols,
                          float eps, hipStream_t st) {
  const int nvec = cols / 8;
  auto go = [&](auto lpr_c, auto vpl_c) {
    constexpr int LPR = decltype(lpr_c)::value, VPL = decltype(vpl_c)::value;
    // decode-sized inputs (a few dozen rows): one wave per workgroup so the rows spread over CUs
    const int threads = rows * LPR <= 64 * 64 ? 64 : 256;
    const int rows_per_block = (threads / 64) * (64 / LPR);
    const int grid = (rows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL((rmsnorm_kernel<LPR, VPL, ADD, PART>), dim3(grid), dim3(threads), 0, st, x, part, nsplit, xs,
                       inner, r, w, y, ro, rows, cols, eps);
  };
  using std::integral_constant;
  if (nvec <= 8) go(integral_constant<int, 8>{}, integral_constant<int, 1>{});
  else if (nvec <= 16) go(integral_constant<int, 16>{}, integral_constant<int, 1>{});
  else if (nvec <= 32) go(integral_constant<int, 32>{}, integral_constant<int, 1>{});
  else if (nvec <= 64) go(integral_constant<int, 64>{}, integral_constant<int, 1>{});
  else if (nvec <= 128) go(integral_constant<int, 64>{}, integral_constant<int, 2>{});
  else if (nvec <= 256) go(integral_constant<int, 64>{}, integral_constant<int, 4>{});
  else if (nvec <= 512) go(integral_constant<int, 64>{}, integral_constant<int, 8>{});
  else if (nvec <= 1024) go(integral_constant<int, 64>{}, integral_constant<int, 16>{});
  else return MI_EUNSUPPORTED;
  return check_launch();
}

// ---------------------------------------------------------------------------
// q/k norm + RoPE + KV scatter: 8 lanes per (token, head); lane j owns
// d = 8j..8j+7 and d = 64+8j..64+8j+7, i.e. the NeoX rotation pairs are in-lane
// and each half is one 16-byte chunk of the fragment-native K tile.
// ---------------------------------------------------------------------------
struct HeadSlot {
  int token, head, kind;  // kind 0 = q, 1 = k, 2 = v
};

// MODE 0: fused norm+rope+store from packed qkv
// MODE 1: rope only (q,k separate inputs -> contiguous outputs)
// MODE 2: store only (k,v separate inputs -> caches)
template <int MODE>
__global__ __launch_bounds__(256) void qk_rope_store_kernel(
    const uint16_t* __restrict__ qsrc, int64_t q_stride, const uint16_t* __restrict__ ksrc,
    int64_t k_stride, const uint16_t* __restrict__ vsrc, int64_t v_stride,
    const uint16_t* __restrict__ q_w, const uint16_t* __restrict__ k_w, float eps,
    const int64_t* __restrict__ positions, const float* __restrict__ cos_sin,
    uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_out, uint16_t* __restrict__ k_cache,
    uint16_t* __restrict__ v_cache, const int32_t* __restrict__ slots, int slot_is_2d, int n_tokens,
    int n_q_heads, int n_kv_heads, int block_size, int skip_v) {
  const int heads_per_token = MODE == 0 ? n_q_heads + 2 * n_kv_heads
                              : MODE == 1 ? n_q_heads + n_kv_heads
                                          : 2 * n_kv_heads;
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;  // (token, head) slot
  const int j = threadIdx.x & 7;
  const bool active = gid < (int64_t)n_tokens * heads_per_token;
  const int token = active ? (int)(gid / heads_per_token) : 0;
  int hh = active ? (int)(gid % heads_per_token) : 0;
  int kind;  // 0 q, 1 k, 2 v
  if (MODE == 2) {
    kind = hh < n_kv_heads ? 1 : 2;
    if (kind == 2) hh -= n_kv_heads;
  } else {
    kind = hh < n_q_heads ? 0 : (hh < n_q_heads + n_kv_heads ? 1 : 2);
    if (kind == 1) hh -= n_q_heads;
    if (kind == 2) hh -= n_q_heads + n_kv_heads;
  }
  const uint16_t* src = kind == 0 ? qsrc + token * q_stride
                        : kind == 1 ? ksrc + token * k_stride
                                    : vsrc + token * v_stride;
  src += hh * 128;
  if (kind == 2 && skip_v) return;  // V goes through v_store_tiles_kernel (whole 8-lane group exits)
  float a[8], b[8];
  if (active) {
    load16(src + 8 * j, a);
    load16(src + 64 + 8 * j, b);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = b[i] = 0.f;
  }
  if (MODE == 0 && kind != 2) {
    const uint16_t* w = kind == 0 ? q_w : k_w;
    if (w != nullptr) head_rmsnorm(a, b, w, j, eps);  // shuffles: all 8 lanes of the group run it
  }
  if (MODE != 2 && kind != 2 && active) {
    const int64_t pos = positions[token];
    head_rope(a, b, cos_sin + pos * 128, j);
  }
  if (!active) return;
  if (MODE != 2 && kind == 0) {
    uint16_t* dst = q_out + ((int64_t)token * n_q_heads + hh) * 128;
    *reinterpret_cast<u32x4*>(dst + 8 * j) = pack16(a);
    *reinterpret_cast<u32x4*>(dst + 64 + 8 * j) = pack16(b);
    return;
  }
  if (MODE == 1) {  // k -> contiguous output
    uint16_t* dst = k_out + ((int64_t)token * n_kv_heads + hh) * 128;
    *reinterpret_cast<u32x4*>(dst + 8 * j) = pack16(a);
    *reinterpret_cast<u32x4*>(dst + 64 + 8 * j) = pack16(b);
    return;
  }
  int64_t blk;
  int off;
  if (!resolve_slot(slots, slot_is_2d, token, block_size, blk, off)) return;
  const int tpb = block_size >> 4;
  if (kind == 1) store_k_head(k_cache, blk, off, hh, j, a, b, n_kv_heads, tpb);
  else store_v_head(v_cache, blk, off, hh, j, a, b, n_kv_heads, tpb);
}

// V scatter for many tokens (prefill): one workgroup per (16 consecutive tokens, kv head).  When the
// 16 tokens fill one aligned cache tile (the common case: a sequence's tokens occupy consecutive
// slots from a block boundary) the tile is transposed through LDS and written as one coalesced
// 4 KiB run; otherwise it falls back to element scatter.  Pure copy either way.
__global__ __launch_bounds__(128) void v_store_tiles_kernel(const uint16_t* __restrict__ vsrc, int64_t v_stride,
                                                            uint16_t* __restrict__ v_cache,
                                                            const int32_t* __restrict__ slots, int n_tokens,
                                                            int n_kv_heads, int block_size) {
  __shared__ uint16_t sm[16][128 + 8];
  __shared__ int sm_slot[16];
  const int t0 = blockIdx.x * 16, h = blockIdx.y, tid = threadIdx.x;
  const int n_here = min(16, n_tokens - t0);
  if (tid < 16) sm_slot[tid] = tid < n_here ? slots[t0 + tid] : -2;
  // coalesced read of 16 rows x 256 B
  for (int c = tid; c < 256; c += 128) {
    const int tk = c >> 4, ch = c & 15;
    u32x4 v = {0, 0, 0, 0};
    if (tk < n_here) v = *reinterpret_cast<const u32x4*>(vsrc + (int64_t)(t0 + tk) * v_stride + h * 128 + ch * 8);
    *reinterpret_cast<u32x4*>(&sm[tk][ch * 8]) = v;
  }
  __syncthreads();
  const int s0 = sm_slot[0];
  bool tile_ok = s0 >= 0 && (s0 & 15) == 0;
#pragma unroll
  for (int i = 1; i < 16; ++i) tile_ok = tile_ok && sm_slot[i] == s0 + i;
  const int tpb = block_size >> 4;
  if (tile_ok) {
    uint16_t* tile = v_cache + kv_tile_base(s0 / block_size, h, s0 % block_size, n_kv_heads, tpb);
    for (int oc = tid; oc < 256; oc += 128) {  // output chunk (jp, g4, n): 4 tokens x {d, d+16}
      const int jp = oc >> 6, g4 = (oc >> 4) & 3, nn = oc & 15;
      const int d0 = jp * 32 + nn;
      u32x4 o;
      o[0] = (uint32_t)sm[4 * g4 + 0][d0] | ((uint32_t)sm[4 * g4 + 1][d0] << 16);
      o[1] = (uint32_t)sm[4 * g4 + 2][d0] | ((uint32_t)sm[4 * g4 + 3][d0] << 16);
      o[2] = (uint32_t)sm[4 * g4 + 0][d0 + 16] | ((uint32_t)sm[4 * g4 + 1][d0 + 16] << 16);
      o[3] = (uint32_t)sm[4 * g4 + 2][d0 + 16] | ((uint32_t)sm[4 * g4 + 3][d0 + 16] << 16);
      *reinterpret_cast<u32x4*>(tile + oc * 8) = o;
    }
  } else {
    for (int e = tid; e < 16 * 128; e += 128) {
      const int tk = e >> 7, d = e & 127;
      const int sl = sm_slot[tk];
      if (sl < 0) continue;
      uint16_t* tile = v_cache + kv_tile_base(sl / block_size, h, sl % block_size, n_kv_heads, tpb);
      tile[v_tile_off(sl & 15, d)] = sm[tk][d];
    }
  }
}

// Prefill form of the fused q/k-norm + RoPE + KV store: one workgroup per (16 consecutive tokens,
// kv head) handles that head's K rows, V rows and its G query heads.  16 groups of 8 lanes, one token
// each, run the same per-head arithmetic as qk_rope_store_kernel (bit-identical results); K and V
// are assembled as whole 4 KiB cache tiles in LDS and leave as one coalesced run when the 16 tokens
// fill an aligned tile (always, for a sequence that starts on a block boundary) - the per-token form
// writes each K row as sixteen 16-byte pieces 256 bytes apart.  Q rows go to q_out (contiguous rows).
__global__ __launch_bounds__(128) void qkv_prefill_store_kernel(
    const uint16_t* __restrict__ qkv, int64_t row_stride, const uint16_t* __restrict__ q_w,
    const uint16_t* __restrict__ k_w, float eps, const int64_t* __restrict__ positions,
    const float* __restrict__ cos_sin, uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache,
    uint16_t* __restrict__ v_cache, const int32_t* __restrict__ slots, int n_tokens, int n_q_heads,
    int n_kv_heads, int block_size) {
  __shared__ __attribute__((aligned(16))) uint16_t k_img[2048];  // the K tile as stored
  __shared__ __attribute__((aligned(16))) uint16_t sm[16][128 + 8];
  __shared__ int sm_slot[16];
  const int t0 = blockIdx.x * 16, h = blockIdx.y, tid = threadIdx.x;
  const int tk = tid >> 3, j = tid & 7;
  const int n_here = min(16, n_tokens - t0);
  const int G = n_q_heads / n_kv_heads;
  const bool live = tk < n_here;
  const int token = live ? t0 + tk : t0;  // idle groups shadow a valid token (shuffles stay defined)
  if (tid < 16) sm_slot[tid] = tid < n_here ? slots[t0 + tid] : -2;
  const uint16_t* row = qkv + (int64_t)token * row_stride;
  const float* cs = cos_sin + positions[token] * 128;

  // ---- K: norm + RoPE, into the tile image
  float a[8], b[8];
  load16(row + (n_q_heads + h) * 128 + 8 * j, a);
  load16(row + (n_q_heads + h) * 128 + 64 + 8 * j, b);
  if (k_w != nullptr) head_rmsnorm(a, b, k_w, j, eps);
  head_rope(a, b, cs, j);
  *reinterpret_cast<u32x4*>(&k_img[k_tile_off(tk, 8 * j)]) = pack16(a);
  *reinterpret_cast<u32x4*>(&k_img[k_tile_off(tk, 64 + 8 * j)]) = pack16(b);
  // ---- V: plain rows into the transpose buffer
  const uint16_t* vrow = row + (n_q_heads + n_kv_heads + h) * 128;
  {
    u32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
    if (live) {
      v0 = *reinterpret_cast<const u32x4*>(vrow + 8 * j);
      v1 = *reinterpret_cast<const u32x4*>(vrow + 64 + 8 * j);
    }
    *reinterpret_cast<u32x4*>(&sm[tk][8 * j]) = v0;
    *reinterpret_cast<u32x4*>(&sm[tk][64 + 8 * j]) = v1;
  }
  __syncthreads();
  const int s0 = sm_slot[0];
  bool tile_ok = s0 >= 0 && (s0 & 15) == 0;
#pragma unroll
  for (int i = 1; i < 16; ++i) tile_ok = tile_ok && sm_slot[i] == s0 + i;
  const int tpb = block_size >> 4;
  if (tile_ok) {
    const int64_t base = kv_tile_base(s0 / block_size, h, s0 % block_size, n_kv_heads, tpb);
    for (int oc = tid; oc < 256; oc += 128) {
      *reinterpret_cast<u32x4*>(k_cache + base + oc * 8) = *reinterpret_cast<const u32x4*>(&k_img[oc * 8]);
      const int jp = oc >> 6, g4 = (oc >> 4) & 3, nn = oc & 15;  // V chunk: 4 tokens x {d, d+16}
      const int d0 = jp * 32 + nn;
      u32x4 o;
      o[0] = (uint32_t)sm[4 * g4 + 0][d0] | ((uint32_t)sm[4 * g4 + 1][d0] << 16);
      o[1] = (uint32_t)sm[4 * g4 + 2][d0] | ((uint32_t)sm[4 * g4 + 3][d0] << 16);
      o[2] = (uint32_t)sm[4 * g4 + 0][d0 + 16] | ((uint32_t)sm[4 * g4 + 1][d0 + 16] << 16);
      o[3] = (uint32_t)sm[4 * g4 + 2][d0 + 16] | ((uint32_t)sm[4 * g4 + 3][d0 + 16] << 16);
      *reinterpret_cast<u32x4*>(v_cache + base + oc * 8) = o;
    }
  } else if (live) {  // ragged: per-token scatter (negative slots are skipped)
    const int sl = sm_slot[tk];
    if (sl >= 0) {
      const int64_t blk = sl / block_size;
      const int off = sl % block_size;
      store_k_head(k_cache, blk, off, h, j, a, b, n_kv_heads, tpb);
      uint16_t* tile = v_cache + kv_tile_base(blk, h, off, n_kv_heads, tpb);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        tile[v_tile_off(off & 15, 8 * j + i)] = sm[tk][8 * j + i];
        tile[v_tile_off(off & 15, 64 + 8 * j + i)] = sm[tk][64 + 8 * j + i];
      }
    }
  }
  // ---- Q: the G heads of this kv head (output rows are contiguous: coalesced as they are);
  //         q_out == nullptr: the attention kernel prepares Q itself (mi_paged_attn_prefill_fused)
  if (q_out == nullptr) return;
  for (int g = 0; g < G; ++g) {
    const int hq = h * G + g;
    load16(row + hq * 128 + 8 * j, a);
    load16(row + hq * 128 + 64 + 8 * j, b);
    if (q_w != nullptr) head_rmsnorm(a, b, q_w, j, eps);
    head_rope(a, b, cs, j);
    if (live) {
      uint16_t* dst = q_out + ((int64_t)token * n_q_heads + hq) * 128;
      *reinterpret_cast<u32x4*>(dst + 8 * j) = pack16(a);
      *reinterpret_cast<u32x4*>(dst + 64 + 8 * j) = pack16(b);
    }
  }
}

// ---------------------------------------------------------------------------
// head_dim 64 (round 4: Llama-3.2-1B, Qwen2-0.5B on the fragment-native attention kernels): the same tile formulas
// (k_tile_off / v_tile_off do not depend on the head width), a 16-token tile is the first 1024 elements = 2 KiB.
// Pure copies of already rotated K rows and of V rows (the q/k-norm + RoPE fusions above are written for 128-wide
// heads; models of this width have no q/k norm and take mi_rope_plain first).
// ---------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void kv_store_frag_kernel(const uint16_t* __restrict__ k, int64_t k_stride,
                                                            const uint16_t* __restrict__ v, int64_t v_stride,
                                                            uint16_t* __restrict__ k_cache, uint16_t* __restrict__ v_cache,
                                                            const int32_t* __restrict__ slots, int slot_is_2d, int n_tokens,
                                                            int n_kv_heads, int block_size, int skip_v) {
  constexpr int GROUPS = D / 8;  // 16-byte groups of a head row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_tokens * n_kv_heads * GROUPS) return;
  const int j = (int)(idx % GROUPS), h = (int)((idx / GROUPS) % n_kv_heads), t = (int)(idx / ((int64_t)GROUPS * n_kv_heads));
  int64_t blk;
  int off;
  if (!resolve_slot(slots, slot_is_2d, t, block_size, blk, off)) return;
  const int64_t base = ((blk * n_kv_heads + h) * (block_size >> 4) + (off >> 4)) * (int64_t)(16 * D);
  const int tk = off & 15;
  *reinterpret_cast<u32x4*>(k_cache + base + k_tile_off(tk, 8 * j)) =
      *reinterpret_cast<const u32x4*>(k + (int64_t)t * k_stride + h * D + 8 * j);
  if (skip_v) return;
  const u32x4 vv = *reinterpret_cast<const u32x4*>(v + (int64_t)t * v_stride + h * D + 8 * j);
  uint16_t* vt = v_cache + base;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    vt[v_tile_off(tk, 8 * j + 2 * i)] = (uint16_t)(vv[i] & 0xffffu);
    vt[v_tile_off(tk, 8 * j + 2 * i + 1)] = (uint16_t)(vv[i] >> 16);
  }
}

// V rows of 16 consecutive tokens that fill one aligned cache tile: transposed through LDS, one coalesced 2 D x 16-byte
// run; anything else: element scatter (v_store_tiles_kernel for any head width)
template <int D>
__global__ __launch_bounds__(128) void v_store_tiles_frag_kernel(const uint16_t* __restrict__ vsrc, int64_t v_stride,
                                                                 uint16_t* __restrict__ v_cache,
                                                                 const int32_t* __restrict__ slots, int n_tokens,
                                                                 int n_kv_heads, int block_size) {
  __shared__ uint16_t sm[16][D + 8];
  __shared__ int sm_slot[16];
  const int t0 = blockIdx.x * 16, h = blockIdx.y, tid = threadIdx.x;
  const int n_here = min(16, n_tokens - t0);
  if (tid < 16) sm_slot[tid] = tid < n_here ? slots[t0 + tid] : -2;
  for (int c = tid; c < 16 * (D / 8); c += 128) {  // coalesced read of 16 rows x 2 D bytes
    const int tk = c / (D / 8), ch = c % (D / 8);
    u32x4 v = {0, 0, 0, 0};
    if (tk < n_here) v = *reinterpret_cast<const u32x4*>(vsrc + (int64_t)(t0 + tk) * v_stride + h * D + ch * 8);
    *reinterpret_cast<u32x4*>(&sm[tk][ch * 8]) = v;
  }
  __syncthreads();
  const int s0 = sm_slot[0];
  bool tile_ok = s0 >= 0 && (s0 & 15) == 0;
#pragma unroll
  for (int i = 1; i < 16; ++i) tile_ok = tile_ok && sm_slot[i] == s0 + i;
  const int tpb = block_size >> 4;
  if (tile_ok) {
    uint16_t* tile = v_cache + (((int64_t)(s0 / block_size) * n_kv_heads + h) * tpb + ((s0 % block_size) >> 4)) * (16 * D);
    for (int oc = tid; oc < 2 * D; oc += 128) {  // output chunk (jp, g4, n): 4 tokens x {d, d + 16}
      const int jp = oc >> 6, g4 = (oc >> 4) & 3, nn = oc & 15;
      const int d0 = jp * 32 + nn;
      u32x4 o;
      o[0] = (uint32_t)sm[4 * g4 + 0][d0] | ((uint32_t)sm[4 * g4 + 1][d0] << 16);
      o[1] = (uint32_t)sm[4 * g4 + 2][d0] | ((uint32_t)sm[4 * g4 + 3][d0] << 16);
      o[2] = (uint32_t)sm[4 * g4 + 0][d0 + 16] | ((uint32_t)sm[4 * g4 + 1][d0 + 16] << 16);
      o[3] = (uint32_t)sm[4 * g4 + 2][d0 + 16] | ((uint32_t)sm[4 * g4 + 3][d0 + 16] << 16);
      *reinterpret_cast<u32x4*>(tile + oc * 8) = o;
    }
  } else {
    for (int e = tid; e < 16 * D; e += 128) {
      const int tk = e / D, d = e % D;
      const int sl = sm_slot[tk];
      if (sl < 0) continue;
      uint16_t* tile = v_cache + (((int64_t)(sl / block_size) * n_kv_heads + h) * tpb + ((sl % block_size) >> 4)) * (16 * D);
      tile[v_tile_off(sl & 15, d)] = sm[tk][d];
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void kv_gather_frag_kernel(const uint16_t* __restrict__ cache, int is_v,
                                                             const int32_t* __restrict__ slot_flat, int n,
                                                             uint16_t* __restrict__ out, int n_kv_heads, int block_size) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * n_kv_heads * D) return;
  const int d = (int)(idx % D), h = (int)((idx / D) % n_kv_heads), i = (int)(idx / ((int64_t)D * n_kv_heads));
  const int32_t s = slot_flat[i];
  if (s < 0) {
    out[idx] = 0;
    return;
  }
  const int off = s % block_size;
  const uint16_t* tile = cache + (((int64_t)(s / block_size) * n_kv_heads + h) * (block_size >> 4) + (off >> 4)) * (16 * D);
  out[idx] = tile[is_v ? v_tile_off(off & 15, d) : k_tile_off(off & 15, d)];
}

// inverse of the scatter, for content checks: out[i][h*128+d] = cache[slot_flat[i]][h][d]
__global__ __launch_bounds__(256) void kv_gather_kernel(const uint16_t* __restrict__ cache, int is_v,
                                                        const int32_t* __restrict__ slot_flat, int n,
                                                        uint16_t* __restrict__ out, int n_kv_heads,
                                                        int block_size) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n * n_kv_heads * 128;
  if (idx >= total) return;
  const int d = idx & 127;
  const int h = (idx >> 7) % n_kv_heads;
  const int i = idx / (128 * n_kv_heads);
  const int32_t s = slot_flat[i];
  if (s < 0) {
    out[idx] = 0;
    return;
  }
  const int64_t blk = s / block_size;
  const int off = s % block_size;
  const uint16_t* tile = cache + kv_tile_base(blk, h, off, n_kv_heads, block_size >> 4);
  out[idx] = tile[is_v ? v_tile_off(off & 15, d) : k_tile_off(off & 15, d)];
}

// ---------------------------------------------------------------------------
// SiluAndMul (activation.py:10-12): bf16(bf16(silu(x)) * y)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ x,
                                                       uint16_t* __restrict__ out, int rows, int inter) {
  const int64_t vec = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int vpr = inter >> 3;
  if (vec >= (int64_t)rows * vpr) return;
  const int r = vec / vpr, c = (vec % vpr) * 8;
  float g[8], u[8];
  load16(x + (int64_t)r * 2 * inter + c, g);
  load16(x + (int64_t)r * 2 * inter + inter + c, u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float s = g[i] / (1.0f + expf(-g[i]));
    g[i] = rbf(s) * u[i];
  }
  *reinterpret_cast<u32x4*>(out + (int64_t)r * inter + c) = pack16(g);
}

// ---------------------------------------------------------------------------
// Row gathers: embedding (embed_head.py:34-42) and last-token select (:58-60)
// ---------------------------------------------------------------------------
template <bool FROM_CU>
__global__ __launch_bounds__(256) void row_gather_kernel(const int64_t* __restrict__ ids,
                                                         const int32_t* __restrict__ cu,
                                                         const uint16_t* __restrict__ w,
                                                         uint16_t* __restrict__ out, int n, int hidden,
                                                         int64_t vocab_start, int64_t vocab_local,
                                                         const int64_t* __restrict__ prev) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= n) return;
  int64_t src;
  bool valid = true;
  if (FROM_CU) {
    src = (int64_t)cu[row + 1] - 1;
  } else {
    // prev != nullptr: `cu` holds, per row, the row of `prev` (the previous step's tokens, still on the
    // device) to take the id from, or -1 for "use ids[row]"
    const int from = prev ? cu[row] : -1;
    src = (from >= 0 ? prev[from] : ids[row]) - vocab_start;
    valid = src >= 0 && src < vocab_local;
  }
  const int nvec = hidden >> 3;
  for (int v = lane; v < nvec; v += 64) {
    u32x4 val = {0, 0, 0, 0};
    if (valid) val = *reinterpret_cast<const u32x4*>(w + src * hidden + v * 8);
    *reinterpret_cast<u32x4*>(out + (int64_t)row * hidden + v * 8) = val;
  }
}

// ---------------------------------------------------------------------------
// argmax / Gumbel-max sampling over [rows][vocab] bf16 logits (sampler.py:9-17)
// one 1024-thread workgroup per row; key = (value, lowest index)
// ---------------------------------------------------------------------------
// scan one row: 4 independent 16-byte loads per thread and iteration are issued before any compare
template <bool NOISY>
__device__ __forceinline__ void scan_row(const uint16_t* __restrict__ p, int vocab, float inv_t, uint64_t rkey,
                                         float& best, int& best_i) {
  const int nvec = vocab >> 3;
  const int T = blockDim.x;
  auto consider = [&](const u32x4& raw, int v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = v * 8 + 2 * j;
      float k0 = lo_bf(raw[j]), k1 = hi_bf(raw[j]);
      if (NOISY) {
        k0 = gumbel_key(k0, inv_t, rkey, col);
        k1 = gumbel_key(k1, inv_t, rkey, col + 1);
      }
      if (k0 > best || (k0 == best && col < best_i)) {
        best = k0;
        best_i = col;
      }
      if (k1 > best || (k1 == best && col + 1 < best_i)) {
        best = k1;
        best_i = col + 1;
      }
    }
  };
  int v = threadIdx.x;
  for (; v + 3 * T < nvec; v += 4 * T) {
    u32x4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) r[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (v + u * T) * 8));
#pragma unroll
    for (int u = 0; u < 4; ++u) consider(r[u], v + u * T);
  }
  for (; v < nvec; v += T) consider(*reinterpret_cast<const u32x4*>(p + v * 8), v);
  for (int col = (nvec << 3) + threadIdx.x; col < vocab; col += T) {  // tail (< 8 columns)
    float key = bf2f(p[col]);
    if (NOISY) key = gumbel_key(key, inv_t, rkey, col);
    if (key > best || (key == best && col < best_i)) {
      best = key;
      best_i = col;
    }
  }
}

template <bool SAMPLE>
__global__ __launch_bounds__(1024) void pick_kernel(const uint16_t* __restrict__ logits,
                                                    int64_t row_stride,
                                                    const float* __restrict__ temperatures,
                                                    int64_t* __restrict__ out, int vocab, uint64_t seed,
                                                    uint64_t step) {
  const int row = blockIdx.x;
  const uint16_t* p = logits + (int64_t)row * row_stride;
  float inv_t = 1.0f;
  bool noisy = false;
  if (SAMPLE) {
    const float t = temperatures[row];
    noisy = t > 0.f;
    inv_t = noisy ? 1.0f / t : 1.0f;
  }
  const uint64_t rkey = sample_row_key(seed, step, row);
  float best = -INFINITY;
  int best_i = 0x7fffffff;
  if (SAMPLE && noisy) scan_row<true>(p, vocab, inv_t, rkey, best, best_i);
  else scan_row<false>(p, vocab, inv_t, rkey, best, best_i);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64);
    if (ob > best || (ob == best && oi < best_i)) {
      best = ob;
      best_i = oi;
    }
  }
  __shared__ float sb[16];
  __shared__ int si[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sb[wave] = best;
    si[wave] = best_i;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    for (int wv = 1; wv < nw; ++wv)
      if (sb[wv] > best || (sb[wv] == best && si[wv] < best_i)) {
        best = sb[wv];
        best_i = si[wv];
      }
    out[row] = best_i == 0x7fffffff ? 0 : best_i;
  }
}

}  // namespace mi

// ===========================================================================
// C ABI
// ===========================================================================
using namespace mi;

extern "C" int mi_rmsnorm(const mi_bf16* x, int64_t x_outer_stride, const mi_bf16* w, mi_bf16* y,
                          int outer, int inner, int cols, float eps, mi_stream stream) {
  if (!x || !w || !y || outer < 0 || inner <= 0 || cols <= 0) return MI_EINVAL;
  if (cols % 8 != 0 || x_outer_stride % 8 != 0) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return MI_EINVAL;
  if (outer == 0) return MI_OK;
  return launch_rmsnorm<false>(x, nullptr, 0, x_outer_stride, inner, nullptr, w, y, nullptr, outer * inner, cols,
                               eps, S(stream));
}

extern "C" int mi_add_rmsnorm(const mi_bf16* x, const mi_bf16* residual, const mi_bf16* w, mi_bf16* y,
                              mi_bf16* residual_out, int rows, int cols, float eps, mi_stream stream) {
  if (!x || !residual || !w || !y || !residual_out || rows < 0 || cols <= 0) return MI_EINVAL;
  if (cols % 8 != 0) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(residual) || !aligned16(w) || !aligned16(y) || !aligned16(residual_out))
    return MI_EINVAL;
  if (rows == 0) return MI_OK;
  // a few rows of many columns (decode steps of hidden 2048 ... 8192 models whose row-parallel projections deliver
  // finished rows): four or eight waves per row instead of one (a Qwen3-32B TP-8 rank's two norms: 6.7 -> ... us each,
  // profiles/r05_kbench_32b_norm.txt); the same arithmetic and rounding points, another grouping of the fp32 sum of squares
  if (rows <= 64 && cols > 1024 && cols <= 8 * 64 * 8 * 2 && tuning(MI_TUNE_NORM_WPR) == 4) {
    if (cols <= 4 * 64 * 8 * 2)
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows_kernel<0, 4>), dim3(rows), dim3(256), 0, S(stream), nullptr, residual, w, y,
                         residual_out, rows, cols, eps, x);
    else
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows_kernel<0, 8>), dim3(rows), dim3(512), 0, S(stream), nullptr, residual, w, y,
                         residual_out, rows, cols, eps, x);
    return check_launch();
  }
  return launch_rmsnorm<true>(x, nullptr, 0, (int64_t)cols, 1, residual, w, y, residual_out, rows, cols, eps,
                              S(stream));
}


extern "C" int mi_add_rmsnorm_splitk(const float* partials, int nsplit, const mi_bf16* residual, const mi_bf16* w,
                                     mi_bf16* y, mi_bf16* residual_out, int rows, int cols, float eps,
                                     mi_stream stream) {
  if (!partials || !residual || !w || !y || !residual_out || rows < 0 || cols <= 0 || nsplit < 1) return MI_EINVAL;
  if (cols % 8 != 0) return MI_EUNSUPPORTED;
  if (!aligned16(partials) || !aligned16(residual) || !aligned16(w) || !aligned16(y) || !aligned16(residual_out))
    return MI_EINVAL;
  if (rows == 0) return MI_OK;
  // decode-sized: WPR waves per row (tuning knob MI_TUNE_NORM_WPR = 1 keeps the one-wave-per-row kernel)
  const int wpr = tuning(MI_TUNE_NORM_WPR);
  constexpr bool narrow = true;  // four columns per thread when the row fits (24.2 vs 24.5 us per layer chain)
#define SPLITK_CASE(NS)                                                                                       \
  case NS:                                                                                                    \
    if (rows <= 64 && wpr == 4 && narrow && cols <= 4 * 64 * 4 && cols % 4 == 0) {                            \
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows4_kernel<NS, 4>), dim3(rows), dim3(256), 0, S(stream),        \
                         partials, residual, w, y, residual_out, rows, cols, eps);                            \
      return check_launch();                                                                                  \
    }                                                                                                         \
    if (rows <= 64 && wpr == 4 && cols <= 4 * 64 * 8 * 2) {                                                   \
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows_kernel<NS, 4>), dim3(rows), dim3(256), 0, S(stream),         \
                         partials, residual, w, y, residual_out, rows, cols, eps);                            \
      return check_launch();                                                                                  \
    }                                                                                                         \
    if (rows <= 64 && wpr == 4 && cols <= 8 * 64 * 8 * 2) {                                                   \
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows_kernel<NS, 8>), dim3(rows), dim3(512), 0, S(stream),         \
                         partials, residual, w, y, residual_out, rows, cols, eps);                            \
      return check_launch();                                                                                  \
    }                                                                                                         \
    if (rows <= 64 && wpr == 2 && cols <= 2 * 64 * 8 * 2) {                                                   \
      hipLaunchKernelGGL((add_rmsnorm_splitk_rows_kernel<NS, 2>), dim3(rows), dim3(128), 0, S(stream),         \
                         partials, residual, w, y, residual_out, rows, cols, eps);                            \
      return check_launch();                                                                                  \
    }                                                                                                         \
    return launch_rmsnorm<true, NS>(nullptr, partials, nsplit, (int64_t)cols, 1, residual, w, y, residual_out, rows, \
                                    cols, eps, S(stream))
  switch (nsplit) {
    SPLITK_CASE(1);
    SPLITK_CASE(2);
    SPLITK_CASE(3);
    SPLITK_CASE(4);
    SPLITK_CASE(6);
    SPLITK_CASE(8);
    default: return MI_EUNSUPPORTED;
  }
#undef SPLITK_CASE
}

// Instrumented form of mi_add_rmsnorm_splitk's decode kernel (tools/chain_timeline.py): stamps[rows][4 waves][8]
extern "C" int mi_add_rmsnorm_splitk_ex(const float* partials, int nsplit, const mi_bf16* residual, const mi_bf16* w,
                                        mi_bf16* y, mi_bf16* residual_out, int rows, int cols, float eps,
                                        uint64_t* stamps, mi_stream stream) {
  if (!partials || !residual || !w || !y || !residual_out || !stamps || rows < 0 || cols <= 0) return MI_EINVAL;
  if (!aligned16(partials) || !aligned16(residual) || !aligned16(w) || !aligned16(y) || !aligned16(residual_out))
    return MI_EINVAL;
  if (nsplit != 4 || rows > 64 || cols > 4 * 64 * 4 || cols % 4) return MI_EUNSUPPORTED;  // the decode chain's form only
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL((add_rmsnorm_splitk_rows4_kernel<4, 4, true>), dim3(rows), dim3(256), 0, S(stream), partials, residual,
                     w, y, residual_out, rows, cols, eps, reinterpret_cast<unsigned long long*>(stamps));
  return check_launch();
}

static int heads_grid(int64_t head_slots) { return (int)((head_slots * 8 + 255) / 256); }

extern "C" int mi_rope(const int64_t* positions, const float* cos_sin, const mi_bf16* q,
                       int64_t q_row_stride, int n_q_heads, const mi_bf16* k, int64_t k_row_stride,
                       int n_kv_heads, mi_bf16* q_out, mi_bf16* k_out, int n_tokens, int head_dim,
                       mi_stream stream) {
  if (!positions || !cos_sin || !q || !k || !q_out || !k_out || n_tokens < 0) return MI_EINVAL;
  if (head_dim != MI_HEAD_DIM || q_row_stride % 8 || k_row_stride % 8) return MI_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(q_out) || !aligned16(k_out) || !aligned16(cos_sin))
    return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  const int64_t slots = (int64_t)n_tokens * (n_q_heads + n_kv_heads);
  hipLaunchKernelGGL((qk_rope_store_kernel<1>), dim3(heads_grid(slots)), dim3(256), 0, S(stream), q,
                     q_row_stride, k, k_row_stride, nullptr, (int64_t)0, nullptr, nullptr, 0.f, positions,
                     cos_sin, q_out, k_out, nullptr, nullptr, nullptr, 0, n_tokens, n_q_heads, n_kv_heads,
                     16, 0);
  return check_launch();
}

static int store_common(const mi_bf16* k, const mi_bf16* v, int64_t ks, int64_t vs, mi_bf16* kc,
                        mi_bf16* vc, const int32_t* slots, int is2d, int n, int n_kv_heads, int head_dim,
                        int block_size, mi_stream stream) {
  if (!k || !v || !kc || !vc || !slots || n < 0 || n_kv_heads <= 0) return MI_EINVAL;
  if ((head_dim != MI_HEAD_DIM && head_dim != 64) || block_size <= 0 || block_size % 16 || ks % 8 || vs % 8)
    return MI_EUNSUPPORTED;
  if (!aligned16(k) || !aligned16(v) || !aligned16(kc) || !aligned16(vc)) return MI_EINVAL;
  if (n == 0) return MI_OK;
  const int64_t hs = (int64_t)n * 2 * n_kv_heads;
  const int tiled_v = (!is2d && n >= 64) ? 1 : 0;
  if (head_dim == 64) {  // 2 KiB tiles
    const int64_t items = (int64_t)n * n_kv_heads * 8;
    hipLaunchKernelGGL(kv_store_frag_kernel<64>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, S(stream), k, ks, v,
                       vs, kc, vc, slots, is2d, n, n_kv_heads, block_size, tiled_v);
    int rc64 = check_launch();
    if (rc64 != MI_OK || !tiled_v) return rc64;
    hipLaunchKernelGGL(v_store_tiles_frag_kernel<64>, dim3((n + 15) / 16, n_kv_heads), dim3(128), 0, S(stream), v, vs, vc,
                       slots, n, n_kv_heads, block_size);
    return check_launch();
  }
  hipLaunchKernelGGL((qk_rope_store_kernel<2>), dim3(heads_grid(hs)), dim3(256), 0, S(stream), nullptr,
                     (int64_t)0, k, ks, v, vs, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, kc,
                     vc, slots, is2d, n, 0, n_kv_heads, block_size, tiled_v);
  int rc = check_launch();
  if (rc != MI_OK || !tiled_v) return rc;
  hipLaunchKernelGGL(v_store_tiles_kernel, dim3((n + 15) / 16, n_kv_heads), dim3(128), 0, S(stream), v, vs, vc,
                     slots, n, n_kv_heads, block_size);
  return check_launch();
}

extern "C" int mi_reshape_and_cache(const mi_bf16* k, const mi_bf16* v, int64_t k_row_stride,
                                    int64_t v_row_stride, mi_bf16* k_cache, mi_bf16* v_cache,
                                    const int32_t* slot_flat, int n_tokens, int n_kv_heads, int head_dim,
                                    int block_size, mi_stream stream) {
  return store_common(k, v, k_row_stride, v_row_stride, k_cache, v_cache, slot_flat, 0, n_tokens, n_kv_heads,
                      head_dim, block_size, stream);
}

extern "C" int mi_scatter_update_kv(const mi_bf16* k, const mi_bf16* v, int64_t k_row_stride,
                                    int64_t v_row_stride, mi_bf16* k_cache, mi_bf16* v_cache,
                                    const int32_t* slot_2d, int batch, int n_kv_heads, int head_dim,
                                    int block_size, mi_stream stream) {
  return store_common(k, v, k_row_stride, v_row_stride, k_cache, v_cache, slot_2d, 1, batch, n_kv_heads,
                      head_dim, block_size, stream);
}

extern "C" int mi_kv_cache_gather(const mi_bf16* cache, int is_v, const int32_t* slot_flat, int n,
                                  mi_bf16* out, int n_kv_heads, int head_dim, int block_size,
                                  mi_stream stream) {
  if (!cache || !slot_flat || !out || n < 0 || n_kv_heads <= 0) return MI_EINVAL;
  if ((head_dim != MI_HEAD_DIM && head_dim != 64) || block_size <= 0 || block_size % 16) return MI_EUNSUPPORTED;
  if (n == 0) return MI_OK;
  if (head_dim == 64) {
    const int64_t total64 = (int64_t)n * n_kv_heads * 64;
    hipLaunchKernelGGL(kv_gather_frag_kernel<64>, dim3((int)((total64 + 255) / 256)), dim3(256), 0, S(stream), cache, is_v,
                       slot_flat, n, out, n_kv_heads, block_size);
    return check_launch();
  }
  const int64_t total = (int64_t)n * n_kv_heads * 128;
  hipLaunchKernelGGL(kv_gather_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, S(stream), cache, is_v,
                     slot_flat, n, out, n_kv_heads, block_size);
  return check_launch();
}

extern "C" int mi_qknorm_rope_store(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w,
                                    const mi_bf16* k_w, float eps, const int64_t* positions,
                                    const float* cos_sin, mi_bf16* q_out, mi_bf16* k_cache, mi_bf16* v_cache,
                                    const int32_t* slots, int slot_is_2d, int n_tokens, int n_q_heads,
                                    int n_kv_heads, int head_dim, int block_size, mi_stream stream) {
  if (!qkv || !positions || !cos_sin || !k_cache || !v_cache || !slots || n_tokens < 0) return MI_EINVAL;
  if (!q_out && (slot_is_2d || n_tokens < 64)) return MI_EINVAL;  // K/V-only is the prefill tile kernel's mode
  if ((q_w == nullptr) != (k_w == nullptr)) return MI_EINVAL;
  if (head_dim != MI_HEAD_DIM || block_size <= 0 || block_size % 16 || qkv_row_stride % 8)
    return MI_EUNSUPPORTED;
  if (!aligned16(qkv) || (q_out && !aligned16(q_out)) || !aligned16(k_cache) || !aligned16(v_cache) ||
      !aligned16(cos_sin) || (q_w && (!aligned16(q_w) || !aligned16(k_w))))
    return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  const int64_t hs = (int64_t)n_tokens * (n_q_heads + 2 * n_kv_heads);
  const mi_bf16* ksrc = qkv + (int64_t)n_q_heads * 128;
  const mi_bf16* vsrc = ksrc + (int64_t)n_kv_heads * 128;
  if (!slot_is_2d && n_tokens >= 64) {  // many tokens with flat slots (prefill): whole cache tiles per workgroup
    if (n_q_heads % n_kv_heads) return MI_EUNSUPPORTED;
    hipLaunchKernelGGL(qkv_prefill_store_kernel, dim3((n_tokens + 15) / 16, n_kv_heads), dim3(128), 0, S(stream),
                       qkv, qkv_row_stride, q_w, k_w, eps, positions, cos_sin, q_out, k_cache, v_cache, slots,
                       n_tokens, n_q_heads, n_kv_heads, block_size);
    return check_launch();
  }
  // decode-sized: one wave per workgroup, so the (token, head) groups spread over four times as many CUs
  const int small_blocks = tuning(MI_TUNE_ROPE_BLOCK64);
  const int threads = (n_tokens <= 64 && small_blocks) ? 64 : 256;
  hipLaunchKernelGGL((qk_rope_store_kernel<0>), dim3((unsigned)((hs * 8 + threads - 1) / threads)), dim3(threads), 0,
                     S(stream), qkv, qkv_row_stride, ksrc, qkv_row_stride, vsrc, qkv_row_stride, q_w, k_w, eps,
                     positions, cos_sin, q_out, nullptr, k_cache, v_cache, slots, slot_is_2d, n_tokens, n_q_heads,
                     n_kv_heads, block_size, 0);
  return check_launch();
}

extern "C" int mi_silu_mul(const mi_bf16* x, mi_bf16* out, int rows, int inter, mi_stream stream) {
  if (!x || !out || rows < 0 || inter <= 0) return MI_EINVAL;
  if (inter % 8) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out)) return MI_EINVAL;
  if (rows == 0) return MI_OK;
  const int64_t nvec = (int64_t)rows * (inter / 8);
  hipLaunchKernelGGL(silu_mul_kernel, dim3((int)((nvec + 255) / 256)), dim3(256), 0, S(stream), x, out, rows,
                     inter);
  return check_launch();
}

extern "C" int mi_embedding(const int64_t* ids, const mi_bf16* w, mi_bf16* out, int n_tokens, int hidden,
                            int64_t vocab_start, int64_t vocab_local, mi_stream stream) {
  if (!ids || !w || !out || n_tokens < 0 || hidden <= 0) return MI_EINVAL;
  if (hidden % 8) return MI_EUNSUPPORTED;
  if (!aligned16(w) || !aligned16(out)) return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  hipLaunchKernelGGL((row_gather_kernel<false>), dim3((n_tokens + 3) / 4), dim3(256), 0, S(stream), ids,
                     nullptr, w, out, n_tokens, hidden, vocab_start, vocab_local, (const int64_t*)nullptr);
  return check_launch();
}

extern "C" int mi_embedding_from_prev(const int64_t* ids, const int32_t* src_rows, const int64_t* prev_tokens,
                                      const mi_bf16* w, mi_bf16* out, int n_tokens, int hidden,
                                      int64_t vocab_start, int64_t vocab_local, mi_stream stream) {
  if (!ids || !src_rows || !prev_tokens || !w || !out || n_tokens < 0 || hidden <= 0) return MI_EINVAL;
  if (hidden % 8) return MI_EUNSUPPORTED;
  if (!aligned16(w) || !aligned16(out)) return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  hipLaunchKernelGGL((row_gather_kernel<false>), dim3((n_tokens + 3) / 4), dim3(256), 0, S(stream), ids,
                     src_rows, w, out, n_tokens, hidden, vocab_start, vocab_local, prev_tokens);
  return check_launch();
}

// Step metadata between pinned host memory and the device, as a KERNEL on the step's own stream (one 16-byte load and
// store per thread, every load of the launch in flight at once): an async memcpy of these few tens of KB goes through the
// copy engine and its cross-queue signals, which leaves the device idle for ~25 us between two decode graphs.
namespace mi {
__global__ __launch_bounds__(256) void stage_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nvec) dst[i] = src[i];
}
}  // namespace mi

extern "C" int mi_stage_copy(void* dst, const void* src, int64_t nbytes, mi_stream stream) {
  if (!dst || !src || nbytes < 0) return MI_EINVAL;
  if (nbytes % 16 || !aligned16(dst) || !aligned16(src)) return MI_EUNSUPPORTED;
  if (nbytes == 0) return MI_OK;
  const int64_t nvec = nbytes / 16;
  if (nvec > ((int64_t)1 << 24)) return MI_EUNSUPPORTED;  // staging buffers, not bulk data
  hipLaunchKernelGGL(mi::stage_copy_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, S(stream),
                     static_cast<const mi::u32x4*>(src), static_cast<mi::u32x4*>(dst), nvec);
  return check_launch();
}

extern "C" int mi_gather_last_tokens(const mi_bf16* x, const int32_t* cu_seqlens_q, mi_bf16* out, int n_seqs,
                                     int hidden, mi_stream stream) {
  if (!x || !cu_seqlens_q || !out || n_seqs < 0 || hidden <= 0) return MI_EINVAL;
  if (hidden % 8) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out)) return MI_EINVAL;
  if (n_seqs == 0) return MI_OK;
  hipLaunchKernelGGL((row_gather_kernel<true>), dim3((n_seqs + 3) / 4), dim3(256), 0, S(stream), nullptr,
                     cu_seqlens_q, x, out, n_seqs, hidden, (int64_t)0, (int64_t)0, (const int64_t*)nullptr);
  return check_launch();
}

extern "C" int mi_argmax(const mi_bf16* logits, int64_t row_stride, int64_t* out, int rows, int vocab,
                         mi_stream stream) {
  if (!logits || !out || rows < 0 || vocab <= 0) return MI_EINVAL;
  if (row_stride % 8 || !aligned16(logits)) return MI_EUNSUPPORTED;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL((pick_kernel<false>), dim3(rows), dim3(1024), 0, S(stream), logits, row_stride, nullptr,
                     out, vocab, 0ull, 0ull);
  return check_launch();
}

extern "C" int mi_sample(const mi_bf16* logits, int64_t row_stride, const float* temperatures, int64_t* out,
                         int rows, int vocab, uint64_t seed, uint64_t step, mi_stream stream) {
  if (!logits || !temperatures || !out || rows < 0 || vocab <= 0) return MI_EINVAL;
  if (row_stride % 8 || !aligned16(logits)) return MI_EUNSUPPORTED;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL((pick_kernel<true>), dim3(rows), dim3(1024), 0, S(stream), logits, row_stride,
                     temperatures, out, vocab, seed, step);
  return check_launch();
}
