// The decode chain in FIVE launches per layer: the row-parallel projections (o_proj, down_proj)
// deliver complete rows, add the residual and emit the RMSNorm statistic; the column-parallel consumers (qkv,
// gate_up + SwiGLU) normalise their own K-slice on load.  The two add+RMSNorm launches of the seven-launch layer
// (mi_add_rmsnorm_splitk) disappear.
//
//   producer  gemm_rowstat_kernel   y = bf16(x @ W^T);  s = float(y) + float(residual)  (fp32, un-rounded: the value the
//                                   reference normalises, layernorm.py:29-35);  residual_out = bf16(s);
//                                   stat[row][tile] = sum of s^2 over the tile's 16 features, fixed order
//   consumer  gemm_normed_kernel    rstd[row] from the row's partials (fixed order);  x = bf16(bf16(s * rstd) * w_norm)
//                                   built in the wave's LDS slab for its K-slice only, under weight loads already in flight
//   tail      norm_from_stat_kernel the model's final norm from (s, stat)
//
// No cross-workgroup hand-off inside a launch (on this part one costs more than a kernel boundary):
// a workgroup of the producer owns 16 output features of EIGHT activation rows over the whole K, so a tile's weights are
// read by ceil(M / 8) workgroups (from L2 after the first) and nothing has to be summed across workgroups.
//
// The bits are those of the seven-launch chain at hidden 1024 (tests require equality):
//   * a wave keeps one accumulator per 64-deep sub-slice of K - the MFMA chains of mi_gemm_bf16_packed_splitk's waves -
//     and the epilogue adds them in that kernel's order (the waves of a split in wave order, then the splits in order);
//   * the sum of squares is the tree of add_rmsnorm_splitk_rows4_kernel: four columns per leaf, a butterfly over
//     the 64 leaves of 256 columns starting with the nearest partners (a tile = four adjacent leaves), then the 256-column
//     groups in order.
#pragma once
#include "gemm_skinny_kernel.hpp"

namespace mi {

constexpr int kStatRows = 8;  // activation rows per producer workgroup

// STAMP: s_memrealtime at entry / loads issued / data arrived / sums in LDS / barrier passed / stores issued / stores
// acknowledged into stamps[workgroup][wave][8] (tools/chain_timeline.py); separate instantiations.
#define MI_C5STAMP(i)                                   \
  do {                                                  \
    if constexpr (STAMP) {                              \
      __builtin_amdgcn_sched_barrier(0);                \
      ts[i] = __builtin_amdgcn_s_memrealtime();         \
      __builtin_amdgcn_sched_barrier(0);                \
    }                                                   \
  } while (0)

// the butterfly over a wave's 16-lane groups with the nearest partners first (1, 2, 4, 8): every lane of a group ends
// with the group's sum
__device__ __forceinline__ float group16_sum_up(float v) {
  v = xor_sum<1>(v);
  v = xor_sum<2>(v);
  v = xor_sum<4>(v);
  return xor_sum<8>(v);
}

// One row's sum of squares from its per-tile partials stat_row[nstat] (nstat <= 512): lane l takes tiles l, l + 64, ...
// (all loads issued before the first add), the 16-lane groups add up by butterfly, the four groups in order.
// All 64 lanes active; every lane returns the value.
constexpr int kStatPerLane = 8;
struct StatRegs {
  float v[kStatPerLane];
};
__device__ __forceinline__ StatRegs load_stat(const float* __restrict__ stat_row, int nstat, int lane, bool on) {
  StatRegs p;
#pragma unroll
  for (int j = 0; j < kStatPerLane; ++j) {
    p.v[j] = 0.f;
    if (j * 64 < nstat) {  // uniform
      const int i = lane + 64 * j;
      if (on && i < nstat) p.v[j] = stat_row[i];
    }
  }
  return p;
}
__device__ __forceinline__ float sumsq_tree(const StatRegs& p, int nstat) {
  float v = p.v[0];
#pragma unroll
  for (int j = 1; j < kStatPerLane; ++j)
    if (j * 64 < nstat) v += p.v[j];
  v = group16_sum_up(v);
  float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  t += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return t;
}

// ---------------------------------------------------------------------------------------------------
// Producer.  grid (N / 16, ceil(M / 8)); WAVES waves, wave w owns sub-slices w * SUB .. + SUB - 1 of 64 k each.
//   ksplit: the split-K geometry whose summation order is reproduced (K / 64 sub-slices in ksplit runs).
// ---------------------------------------------------------------------------------------------------
template <int WAVES, int SUB, bool STAMP = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_rowstat_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ residual,
    uint16_t* __restrict__ residual_out, float* __restrict__ s_out, float* __restrict__ stat, int M, int N, int K,
    int ksplit, unsigned long long* __restrict__ stamps = nullptr) {
  unsigned long long ts[8] = {};
  MI_C5STAMP(0);
  __shared__ __attribute__((aligned(16))) float red[WAVES * SUB][64][4];
  __shared__ __attribute__((aligned(16))) float pl[16][64][4];
  __shared__ __attribute__((aligned(16))) uint16_t slabs[WAVES][kStatRows * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int tile = blockIdx.x;
  const int m0 = (int)blockIdx.y * kStatRows;
  const int Mc = min(M - m0, kStatRows);
  const int ktiles = K >> 5;
  const int q0 = wave * SUB;  // first sub-slice

  // the epilogue's operands first: they are on nobody's way and needed at the very end
  const int erow = lane & 15, efg = lane >> 4;
  const bool evalid = wave == 0 && erow < Mc;
  const int64_t eoff = (int64_t)(m0 + (erow < Mc ? erow : 0)) * N + tile * 16 + 4 * efg;
  u32x2 rr = {0, 0};
  if (evalid) rr = *reinterpret_cast<const u32x2*>(residual + eoff);

  u32x4 a[SUB][2], stage[SUB];
  const uint16_t* wp = w + ((int64_t)tile * ktiles + q0 * 2) * 512 + lane * 8;
#pragma unroll
  for (int j = 0; j < SUB; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) a[j][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (j * 2 + s) * 512));
  {
    const int row = lane >> 3, c = lane & 7;
    const uint16_t* xr = x + (int64_t)(m0 + min(row, Mc - 1)) * K + q0 * 64 + 8 * c;
#pragma unroll
    for (int j = 0; j < SUB; ++j) stage[j] = *reinterpret_cast<const u32x4*>(xr + 64 * j);
  }
  if constexpr (STAMP) {
    MI_C5STAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_C5STAMP(2);
  }
  uint16_t* slab = &slabs[wave][0];
#pragma unroll
  for (int j = 0; j < SUB; ++j) {
    {
      const int row = lane >> 3, c = lane & 7;
      *reinterpret_cast<u32x4*>(slab + (row * 8 + (c ^ row)) * 8) = stage[j];
    }
    u32x4 b[2];
    const int row = r & 7;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) b[h2] = *reinterpret_cast<const u32x4*>(slab + (row * 8 + ((4 * h2 + g) ^ row)) * 8);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[j][0]), as_frag(b[0]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[j][1]), as_frag(b[1]), acc, 0, 0, 0);
    *reinterpret_cast<f32x4*>(&red[q0 + j][lane][0]) = acc;
  }
  MI_C5STAMP(3);
  __syncthreads();
  MI_C5STAMP(4);
  // the split-K kernel's order: the sub-slices of a split in order (its waves), then the splits in order
  const int nsub = WAVES * SUB, per = nsub / ksplit;
  if (ksplit > 1) {
    for (int sp = wave; sp < ksplit; sp += WAVES) {
      f32x4 p = *reinterpret_cast<const f32x4*>(&red[sp * per][lane][0]);
      for (int i = 1; i < per; ++i) p += *reinterpret_cast<const f32x4*>(&red[sp * per + i][lane][0]);
      *reinterpret_cast<f32x4*>(&pl[sp][lane][0]) = p;
    }
    __syncthreads();
  }
  if (wave == 0) {
    f32x4 t;
    if (ksplit > 1) {
      t = *reinterpret_cast<const f32x4*>(&pl[0][lane][0]);
      for (int sp = 1; sp < ksplit; ++sp) t += *reinterpret_cast<const f32x4*>(&pl[sp][lane][0]);
    } else {
      t = *reinterpret_cast<const f32x4*>(&red[0][lane][0]);
      for (int i = 1; i < nsub; ++i) t += *reinterpret_cast<const f32x4*>(&red[i][lane][0]);
    }
    // C fragment: lane (g, c) = features 4 g .. 4 g + 3 of activation row c - the element order of
    // add_rmsnorm_splitk_rows4_kernel's thread (four consecutive columns), and its arithmetic
    float ss = 0.f;
    {
#pragma clang fp contract(off)
      const u32x2 raw = {pack_bf(t[0], t[1]), pack_bf(t[2], t[3])};
      f32x4 sv;
      u32x2 ro;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float av = lo_bf(raw[j]) + lo_bf(rr[j]);
        const float bv = hi_bf(raw[j]) + hi_bf(rr[j]);
        ro[j] = pack_bf(av, bv);
        sv[2 * j] = av;
        sv[2 * j + 1] = bv;
        ss += av * av;
        ss += bv * bv;
      }
      if (evalid) {
        *reinterpret_cast<u32x2*>(residual_out + eoff) = ro;
        *reinterpret_cast<f32x4*>(s_out + eoff) = sv;
      }
    }
    // the tile's four leaves: partners 16 and 32 lanes away = leaf ^ 1, leaf ^ 2
    ss = evalid ? ss : 0.f;
    ss = xor_sum<16>(ss);
    ss = xor_sum<32>(ss);
    if (evalid && efg == 0) stat[(int64_t)(m0 + erow) * (N >> 4) + tile] = ss;
  }
  if constexpr (STAMP) {
    MI_C5STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_C5STAMP(6);
    if (stamps != nullptr && lane == 0) {
      const int64_t wg = (int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y;
      unsigned long long* dst = stamps + (wg * WAVES + wave) * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = ts[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Consumer: gemm_skinny_kernel's weight stream, K-slice reduction and epilogues (EPI_NONE / EPI_SILU, bit for bit) with
// the B operand built from (s, stat, w_norm) instead of read.  M <= 64 rows (one row chunk), packed bf16 weights.
// ---------------------------------------------------------------------------------------------------
struct NormArgs {
  const float* s;       // [M][K] fp32: residual + projection, un-rounded
  const float* stat;    // [M][nstat] partial sums of squares
  const uint16_t* w;    // [K] norm weight
  int nstat;
  float eps;
};

template <int MT, int RT, int WAVES, int STEPS, int EPI, bool STAMP = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_normed_kernel(NormArgs nm, const uint16_t* __restrict__ w,
                                                                 uint16_t* __restrict__ y, int M, int N, int K,
                                                                 unsigned long long* __restrict__ stamps = nullptr) {
  static_assert(STEPS % 2 == 0, "x is staged in pairs of k-steps");
  static_assert(EPI == EPI_NONE || EPI == EPI_SILU, "plain or SwiGLU epilogue");
  unsigned long long ts[8] = {};
  MI_C5STAMP(0);
  {  // more than 64 rows: blockIdx.z walks them in chunks of 64, as gemm_skinny_kernel
    const int m0 = (int)blockIdx.z * kSkinnyRows;
    nm.s += (int64_t)m0 * K;
    nm.stat += (int64_t)m0 * nm.nstat;
    y += (int64_t)m0 * (EPI == EPI_SILU ? N >> 1 : N);
    M = min(M - m0, kSkinnyRows);
  }
  extern __shared__ __attribute__((aligned(16))) float red[];
  __shared__ float rs_lds[MT * 16];
  constexpr int SLOT = RT * MT * 1024 > x_slab_bytes(MT) ? RT * MT * 1024 : x_slab_bytes(MT);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  uint16_t* slab = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(red) + wave * SLOT);
  const int kslice = K / WAVES;
  const int kbeg = wave * kslice;
  const int ktiles = K >> 5;

  // the rows whose statistic this wave finishes: wave, wave + WAVES, ...  (requested before anything else: first back)
  constexpr int RPW = (MT * 16 + WAVES - 1) / WAVES;
  StatRegs sp_[RPW];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int row = wave + i * WAVES;
    sp_[i] = load_stat(nm.stat + (int64_t)min(row, M - 1) * nm.nstat, nm.nstat, lane, row < M);
  }

  int tile[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) tile[t] = EPI == EPI_SILU ? (int)blockIdx.x + t * (N >> 5) : (int)blockIdx.x * RT + t;
  f32x4 acc[RT][MT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint16_t* wp[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) wp[t] = w + ((int64_t)tile[t] * ktiles + (kbeg >> 5)) * 512 + lane * 8;

  bool first = true;
  for (int k = 0; k < kslice; k += 32 * STEPS) {
    u32x4 a[RT][STEPS], bfrag[MT][STEPS];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        a[t][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp[t] + ((k >> 5) + s) * 512));
    // s in whole 256-byte row pieces: 16 lanes per row, four rows per instruction
    f32x4 stage[STEPS / 2][MT * 4];
    u32x2 nw[STEPS / 2];
    const int c4 = lane & 15;
#pragma unroll
    for (int sp = 0; sp < STEPS / 2; ++sp) {
      nw[sp] = *reinterpret_cast<const u32x2*>(nm.w + kbeg + k + 64 * sp + 4 * c4);
#pragma unroll
      for (int i = 0; i < MT * 4; ++i) {
        const int row = i * 4 + (lane >> 4);
        stage[sp][i] = *reinterpret_cast<const f32x4*>(nm.s + (int64_t)min(row, M - 1) * K + kbeg + k + 64 * sp + 4 * c4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (first) {  // rstd of every row: the waves' shares through LDS (uniform branch: first iteration only)
      first = false;
      if constexpr (STAMP) MI_C5STAMP(1);
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int row = wave + i * WAVES;
        const float tot = sumsq_tree(sp_[i], nm.nstat);
        if (row < MT * 16 && lane == 0) rs_lds[row] = row < M ? 1.0f / sqrtf(tot / (float)K + nm.eps) : 0.f;
      }
      __syncthreads();
      if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MI_C5STAMP(2);
      }
    }
#pragma unroll
    for (int sp = 0; sp < STEPS / 2; ++sp) {
#pragma unroll
      for (int i = 0; i < MT * 4; ++i) {
        const int row = i * 4 + (lane >> 4);
        const float rs = rs_lds[row];
        const f32x4 sv = stage[sp][i];
        u32x2 o;
        o[0] = pack_bf(rbf(sv[0] * rs) * lo_bf(nw[sp][0]), rbf(sv[1] * rs) * hi_bf(nw[sp][0]));
        o[1] = pack_bf(rbf(sv[2] * rs) * lo_bf(nw[sp][1]), rbf(sv[3] * rs) * hi_bf(nw[sp][1]));
        const int c = c4 >> 1;
        *reinterpret_cast<u32x2*>(slab + (row * 8 + (c ^ (row & 7))) * 8 + 4 * (c4 & 1)) = o;
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int row = 16 * m + r, c = 4 * h2 + g;
          bfrag[m][2 * sp + h2] = *reinterpret_cast<const u32x4*>(slab + (row * 8 + (c ^ (row & 7))) * 8);
        }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[t][s]), as_frag(bfrag[m][s]), acc[t][m], 0, 0, 0);
  }

#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int m = 0; m < MT; ++m)
      *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(red) + wave * SLOT + ((t * MT + m) * 64 + lane) * 16) = acc[t][m];
  MI_C5STAMP(3);
  __syncthreads();
  MI_C5STAMP(4);
  constexpr int ITEMS = (EPI == EPI_SILU ? 1 : RT) * MT * 64;
  for (int item = threadIdx.x; item < ITEMS; item += WAVES * 64) {
    const int l = item & 63, m = (item >> 6) % MT, t = (item >> 6) / MT;
    auto total = [&](int tt) {
      f32x4 s = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + ((tt * MT + m) * 64 + l) * 16);
#pragma unroll
      for (int wv = 1; wv < WAVES; ++wv)
        s += *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + wv * SLOT + ((tt * MT + m) * 64 + l) * 16);
      return s;
    };
    const int row = 16 * m + (l & 15);
    if (row >= M) continue;
    if (EPI == EPI_SILU) {
      const f32x4 gt = total(0), up = total(1);
      const int col = (int)blockIdx.x * 16 + 4 * (l >> 4);
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gb = rbf(gt[i]);
        const float sb = rbf(gb / (1.0f + expf(-gb)));
        o[i] = sb * rbf(up[i]);
      }
      *reinterpret_cast<u32x2*>(y + (int64_t)row * (N >> 1) + col) = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
    } else {
      const f32x4 s = total(t);
      const int col = tile[0] * 16 + t * 16 + 4 * (l >> 4);
      *reinterpret_cast<u32x2*>(y + (int64_t)row * N + col) = u32x2{pack_bf(s[0], s[1]), pack_bf(s[2], s[3])};
    }
  }
  if constexpr (STAMP) {
    MI_C5STAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_C5STAMP(6);
    if (stamps != nullptr && lane == 0) {
      unsigned long long* dst = stamps + (((int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.z) * WAVES + wave) * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = ts[q];
    }
  }
}

// The model's final norm from the last down projection's (s, stat): one workgroup of four waves per row, four columns per
// thread and pass; every wave finishes the row's statistic itself (no barrier).
__global__ __launch_bounds__(256) void norm_from_stat_kernel(const float* __restrict__ s, const float* __restrict__ stat,
                                                             int nstat, const uint16_t* __restrict__ w,
                                                             uint16_t* __restrict__ y, int cols, float eps) {
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const float tot = sumsq_tree(load_stat(stat + (int64_t)row * nstat, nstat, lane, true), nstat);
  const float rs = 1.0f / sqrtf(tot / (float)cols + eps);
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + (int64_t)row * cols + c);
    const u32x2 nw = *reinterpret_cast<const u32x2*>(w + c);
    u32x2 o;
    o[0] = pack_bf(rbf(sv[0] * rs) * lo_bf(nw[0]), rbf(sv[1] * rs) * hi_bf(nw[0]));
    o[1] = pack_bf(rbf(sv[2] * rs) * lo_bf(nw[1]), rbf(sv[3] * rs) * hi_bf(nw[1]));
    *reinterpret_cast<u32x2*>(y + (int64_t)row * cols + c) = o;
  }
}

#undef MI_C5STAMP

}  // namespace mi
