// bf16 MFMA tile GEMM for the compute-bound regime (M > 64 rows of activations: every prefill projection,
// decode batches above 64 sequences).
//
//   y[M][N] = x[M][K] @ w[N][K]^T (+ bias)        (reference: F.linear, linear.py:51,73,150)
//   epilogue 1: y[M][N/2] = SiluAndMul(x @ w^T)    (activation.py:10-12 on MergedColumnParallelLinear, linear.py:73)
//
// Both operands are K-contiguous, so both are staged the same way: whole 128-byte lines by LDS-DMA
// (global_load_lds_dwordx4) into XOR-swizzled 16 KiB half-tiles (layout and index arithmetic: gemm_tile_index.hpp).
//
// Schedule (one workgroup = 8 waves = one 256 x 256 output tile, 129 KiB of LDS, one workgroup per CU):
//   * the K loop is a stream of half-tiles q = 4 * kstep + h; phase p = 4 * kstep + ph consumes the half-tiles
//     q <= p + 1 and issues half-tile p + 5 (two DMA instructions per wave), so five half-tiles are always
//     requested ahead and no wait in the loop is vmcnt(0): after issuing p + 5 a wave waits vmcnt(6), i.e. for its
//     own pieces of every half-tile <= p + 2, and the workgroup barrier behind that wait publishes them to the
//     readers of phase p + 1;
//   * the two waves that share a SIMD (wave w and w + 4) run one barrier apart: while one issues its eight
//     v_mfma_f32_32x32x16_bf16 of a phase, the other one reads the next phase's fragments (ds_read_b128,
//     conflict-free through the swizzle) and issues its DMA - the matrix pipe of a SIMD always has one wave
//     feeding it;
//   * past-the-end half-tiles are fetched into a spare 1 KiB block, so the wait counts are the same in every phase.
// Workgroups are dealt to the XCDs so that one XCD's L2 holds a contiguous run of tiles (feature-fastest: the
// tiles that run together share their activation rows).
//
// Summation order: one fp32 MFMA chain over K per output element (k ascending), rounded to bf16 once -
// the same rounding points as F.linear; the SwiGLU epilogue keeps the reference's three roundings
// (bf16 gate_up output, bf16 silu, bf16 product) exactly as the decode GEMM's epilogue does.
#include <type_traits>

#include "mi_common.hpp"
#include "gemm_tile_index.hpp"

namespace mi {
using namespace gt;

typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { TEPI_NONE = 0, TEPI_SILU = 1 };

struct TileArgs {
  const uint16_t* x;
  const uint16_t* w;
  const uint16_t* bias;
  uint16_t* y;
  int64_t ldx, ldy;
  int M, N, K;
  int tiles_f, tiles_t;
};

// Tuning variants (mi_gemm_bf16_ex; the product entry point uses kDefaultVariant):
//   PF       half-tiles requested ahead of the issuing phase (5 or 6); a wave's wait leaves PF - 2 of them in flight.
//            6 is the most the eight LDS slots allow with every overwrite two barriers behind the last read.
//   V & 1    no s_setprio around the MFMA clusters
//   V & 2    both waves of a SIMD in lockstep (no one-barrier stagger)
//   V & 4    workgroup b takes tile b (no XCD-aware order)
constexpr int kDefaultPF = 5, kDefaultV = 0;

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
  static_assert(N == 6 || N == 8, "vmcnt immediates used by the K loop");
  if (N == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
}
#define MI_GT_BARRIER() asm volatile("s_barrier" ::: "memory")

template <int EPI, bool BIAS, int PF, int V>
__global__ __launch_bounds__(512, 2) void gemm_tile_kernel(const TileArgs a) {
  static_assert(PF == 5 || PF == 6, "prefetch distance");
  constexpr int VM = 2 * (PF - 2);  // DMA instructions that may stay in flight behind a phase's wait
  // ONE LDS object (a second one makes hipcc drain the DMA queue before every fragment read)
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fh = wave >> 2, tq = wave & 3, hi = lane >> 5, l31 = lane & 31;

  // XCD-aware tile order: workgroup b runs on XCD b % 8; XCD x takes the x-th contiguous run of tiles
  const int ntiles = a.tiles_f * a.tiles_t;
  const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
  const int qn = ntiles >> 3, rn = ntiles & 7;
  const int tid = (V & 4) ? (int)blockIdx.x : (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  const int m0 = (tid / a.tiles_f) * TILE_T, n0 = (tid % a.tiles_f) * TILE_F;
  const int KT = a.K / BK;

  // ---- LDS-DMA sources: byte offset of this lane's 16 bytes for (half-tile h, instruction i), K step 0 ----
  uint32_t src_off[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int lr = dma_local_row(wave, i, lane), c = dma_chunk(wave, i, lane), r = tile_row(h, lr);
      if (is_weight_half(h)) {
        // SwiGLU: the tile's feature rows are 128 gate rows (A half 0 of both feature halves) and the 128 up rows
        // that pair with them (A half 1)
        const int row = EPI == TEPI_SILU ? ((r >> 6) & 1) * (a.N >> 1) + (n0 >> 1) + (r >> 7) * 64 + (r & 63) : n0 + r;
        src_off[h][i] = (uint32_t)(((int64_t)min(row, a.N - 1) * a.K + c * 8) * 2);
      } else {
        src_off[h][i] = (uint32_t)(((int64_t)min(m0 + r, a.M - 1) * a.ldx + c * 8) * 2);
      }
    }
  char* const dummy = lds + SLOTS * HALF_BYTES;
  // issue half-tile h of K step kt (kt >= KT: into the spare block)
  auto issue = [&](int h, int kt) __attribute__((always_inline)) {
    const bool live = kt < KT;
    const char* base = reinterpret_cast<const char*>(is_weight_half(h) ? a.w : a.x) + (live ? (int64_t)kt * (BK * 2) : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      char* dst = live ? lds + ((kt & 1) * 4 + h) * HALF_BYTES + dma_block(wave, i) * 1024 : dummy;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + src_off[h][i]),
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  // ---- fragment reads: per-lane byte offsets inside a half-tile ----
  int kx[4];
  {
    const int rowoff = (l31 >> 3) * 1024 + (l31 & 7) * 128, sw = swizzle(l31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) kx[kk] = rowoff + ((frag_chunk(kk, hi) ^ sw) << 4);
  }
  const char* const ldsA = lds + fh * 8192;  // + slot * HALF_BYTES + a * 4096 + kx[kk]
  const char* const ldsB = lds + tq * 4096;  // + slot * HALF_BYTES + kx[kk]

  u32x4 Af[2][4], B0[4], B1[4];
  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;

  auto read_a = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        Af[f][kk] = *reinterpret_cast<const u32x4*>(ldsA + slot * HALF_BYTES + f * 4096 + kx[kk]);
  };
  auto read_b = [&](int slot, u32x4 (&B)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) B[kk] = *reinterpret_cast<const u32x4*>(ldsB + slot * HALF_BYTES + kx[kk]);
  };
  auto mma = [&](int ah, int bh, const u32x4 (&B)[4]) __attribute__((always_inline)) {
    if (!(V & 1)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int f = 0; f < 2; ++f)
        acc[ah * 2 + f][bh] =
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(Af[f][kk]), as_frag(B[kk]), acc[ah * 2 + f][bh], 0, 0, 0);
    if (!(V & 1)) __builtin_amdgcn_s_setprio(0);
  };

  // one K step = four phases; st = kt & 1 as a compile-time constant (the LDS addresses fold into immediates)
  auto kstep = [&](int kt, auto stc) __attribute__((always_inline)) {
    constexpr int st = decltype(stc)::value;
    // phase 0: (A half 0, B half 0)
    read_a(st * 4 + 0);
    read_b(st * 4 + 1, B0);
    issue((0 + PF) & 3, kt + ((0 + PF) >> 2));
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_barrier<VM>();
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 0, B0);
    __builtin_amdgcn_sched_barrier(0);
    MI_GT_BARRIER();
    // phase 1: (A half 0, B half 1)
    read_b(st * 4 + 2, B1);
    issue((1 + PF) & 3, kt + ((1 + PF) >> 2));
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_barrier<VM>();
    __builtin_amdgcn_sched_barrier(0);
    mma(0, 1, B1);
    __builtin_amdgcn_sched_barrier(0);
    MI_GT_BARRIER();
    // phase 2: (A half 1, B half 1)
    read_a(st * 4 + 3);
    issue((2 + PF) & 3, kt + ((2 + PF) >> 2));
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_barrier<VM>();
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 1, B1);
    __builtin_amdgcn_sched_barrier(0);
    MI_GT_BARRIER();
    // phase 3: (A half 1, B half 0)
    issue((3 + PF) & 3, kt + ((3 + PF) >> 2));
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_barrier<VM>();
    __builtin_amdgcn_sched_barrier(0);
    mma(1, 0, B0);
    __builtin_amdgcn_sched_barrier(0);
    MI_GT_BARRIER();
  };

  // prologue: half-tiles 0 .. PF - 1 requested, 0 and 1 landed and published
#pragma unroll
  for (int q = 0; q < PF; ++q) issue(q & 3, q >> 2);
  __builtin_amdgcn_sched_barrier(0);
  wait_vm_barrier<VM>();
  if (!(V & 2) && fh == 1) MI_GT_BARRIER();  // the second wave of every SIMD runs one barrier behind the first
  __builtin_amdgcn_sched_barrier(0);

  int kt = 0;
  for (; kt + 1 < KT; kt += 2) {
    kstep(kt, std::integral_constant<int, 0>{});
    kstep(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < KT) kstep(kt, std::integral_constant<int, 0>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the past-the-end loads must not outlive the workgroup's LDS
  if (!(V & 2) && fh == 0) MI_GT_BARRIER();         // barrier counts match again
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue: lane (hi, l31) holds token acc_token(...), features 8 rq + 4 hi + {0..3} of each 32-row fragment ----
  const int n_out = EPI == TEPI_SILU ? a.N >> 1 : a.N;
#pragma unroll
  for (int bh = 0; bh < 2; ++bh) {
    const int tok = m0 + acc_token(tq, bh, l31);
    if (tok >= a.M) continue;
    uint16_t* yrow = a.y + (int64_t)tok * a.ldy;
#pragma unroll
    for (int j = 0; j < (EPI == TEPI_SILU ? 2 : 4); ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        // tile feature row of register 4 rq: fh * 128 + (j >> 1) * 64 + (j & 1) * 32 + 8 rq + 4 hi
        float o[4];
        int col;
        if (EPI == TEPI_SILU) {
          col = (n0 >> 1) + fh * 64 + j * 32 + 8 * rq + 4 * hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float gb = rbf(acc[j][bh][4 * rq + e]);  // the gate_up GEMM output, rounded to bf16 as the unfused path
            const float sb = rbf(gb / (1.0f + expf(-gb)));
            o[e] = sb * rbf(acc[2 + j][bh][4 * rq + e]);
          }
        } else {
          col = n0 + acc_feature(fh, j >> 1, j & 1, 4 * rq, hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[j][bh][4 * rq + e];
          if (BIAS && col < n_out) {
            const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
            o[0] += lo_bf(bw[0]);
            o[1] += hi_bf(bw[0]);
            o[2] += lo_bf(bw[1]);
            o[3] += hi_bf(bw[1]);
          }
        }
        if (col < n_out) *reinterpret_cast<u32x2*>(yrow + col) = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
      }
  }
}

template <int EPI, bool BIAS, int PF = kDefaultPF, int V = kDefaultV>
static int launch_tile(const TileArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((gemm_tile_kernel<EPI, BIAS, PF, V>), dim3(a.tiles_f * a.tiles_t), dim3(512), 0, st, a);
  return check_launch();
}

static int check_tile_gemm(const void* x, int64_t ldx, const void* w, const void* bias, const void* y, int64_t ldy,
                           int M, int N, int K, int epilogue) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0 || ldx < K) return MI_EINVAL;
  if (epilogue != 0 && epilogue != 1) return MI_EINVAL;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias))) return MI_EINVAL;
  if (K % BK || N % 4 || ldx % 8 || ldy % 4) return MI_EUNSUPPORTED;
  if (epilogue == 1 && (bias || (N / 2) % 128)) return MI_EUNSUPPORTED;
  if (ldy < (epilogue == 1 ? N / 2 : N)) return MI_EINVAL;
  // the DMA sources are 32-bit byte offsets from the operand bases
  if ((int64_t)N * K * 2 >= (int64_t)1 << 32 || (int64_t)M * ldx * 2 >= (int64_t)1 << 32) return MI_EUNSUPPORTED;
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_gemm_bf16(const mi_bf16* x, int64_t ldx, const mi_bf16* w, const mi_bf16* bias, mi_bf16* y,
                            int64_t ldy, int M, int N, int K, int epilogue, mi_stream stream) {
  const int rc = check_tile_gemm(x, ldx, w, bias, y, ldy, M, N, K, epilogue);
  if (rc != MI_OK || M == 0) return rc;
  const TileArgs a{x, w, bias, y, ldx, ldy, M, N, K, (N + TILE_F - 1) / TILE_F, (M + TILE_T - 1) / TILE_T};
  hipStream_t st = S(stream);
  if (epilogue == 1) return launch_tile<TEPI_SILU, false>(a, st);
  return bias ? launch_tile<TEPI_NONE, true>(a, st) : launch_tile<TEPI_NONE, false>(a, st);
}

// tuning entry point (tools/gemm_bench.py): variant = PF * 16 + V, no bias, plain epilogue
extern "C" int mi_gemm_bf16_ex(const mi_bf16* x, int64_t ldx, const mi_bf16* w, mi_bf16* y, int64_t ldy, int M, int N,
                               int K, int variant, mi_stream stream) {
  const int rc = check_tile_gemm(x, ldx, w, nullptr, y, ldy, M, N, K, 0);
  if (rc != MI_OK || M == 0) return rc;
  const TileArgs a{x, w, nullptr, y, ldx, ldy, M, N, K, (N + TILE_F - 1) / TILE_F, (M + TILE_T - 1) / TILE_T};
  hipStream_t st = S(stream);
  switch (variant) {
    case 5 * 16 + 0: return launch_tile<TEPI_NONE, false, 5, 0>(a, st);
    case 5 * 16 + 1: return launch_tile<TEPI_NONE, false, 5, 1>(a, st);
    case 5 * 16 + 2: return launch_tile<TEPI_NONE, false, 5, 2>(a, st);
    case 5 * 16 + 4: return launch_tile<TEPI_NONE, false, 5, 4>(a, st);
    case 6 * 16 + 0: return launch_tile<TEPI_NONE, false, 6, 0>(a, st);
    case 6 * 16 + 1: return launch_tile<TEPI_NONE, false, 6, 1>(a, st);
    default: return MI_EUNSUPPORTED;
  }
}
