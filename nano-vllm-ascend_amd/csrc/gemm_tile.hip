// bf16 MFMA tile GEMM for the compute-bound regime (M > 64 rows of activations: every prefill projection,
// large decode batches).
//
//   y[M][N] = x[M][K] @ w[N][K]^T (+ bias)        (reference: F.linear, linear.py:51,73,150)
//   epilogue 1: y[M][N/2] = SiluAndMul(x @ w^T)    (activation.py:10-12 on MergedColumnParallelLinear, linear.py:73)
//
// Both operands are K-contiguous, so both are staged the same way: whole 128-byte lines by LDS-DMA
// (global_load_lds_dwordx4) into XOR-swizzled 16 KiB half-tiles (layout and index arithmetic: gemm_tile_index.hpp).
//
// Schedule (one workgroup = 8 waves, 129 KiB of LDS, one workgroup per CU, PERSISTENT: workgroup b computes the
// 256 x 256 tiles b, b + grid, ...; the DMA stream runs on across tile seams, so only a workgroup's first tile
// pays the fill latency and a tile's stores overlap the next tile's loads):
//   * a K step t is two phases of sixteen v_mfma_f32_32x32x16_bf16 per wave (512 cycles of matrix pipe each), each
//     phase a read segment R and a matrix segment M between workgroup barriers:
//       R_P0(t): reads A half 0, B half 0, B half 1 of step t (16 ds_read_b128);
//                DMA of B half 1, A half 1 of step t + 1 (4 pieces)
//       M_P0(t): (A0 x B0, A0 x B1), sixteen MFMAs back to back
//       R_P1(t): reads A half 1 of step t (8 ds_read_b128); DMA of A half 0, B half 0 of step t + 2 (4 pieces)
//       M_P1(t): (A1 x B1, A1 x B0)
//     A DMA instruction blocks its wave for ~75 cycles: issued between MFMAs that is 75 cycles of idle matrix pipe
//     (measured: 20-26 % of the kernel), issued by the READING wave it hides under the partner's MFMA cluster.
//     Every piece has at least two segments to land.  No wait in the loop is vmcnt(0): a wave waits vmcnt(8) at the
//     end of R_P0 (A1 of this step landed) and vmcnt(6) at the end of R_P1 (A0, B0, B1 of the next step landed),
//     for its OWN pieces, and for its fragment reads (lgkmcnt(0): a slot just read may be re-filled by the partner
//     group one barrier later); the workgroup barrier behind the wait publishes the pieces to the next readers;
//   * the two waves that share a SIMD (wave w and w + 4) run one barrier apart: while one issues its MFMA cluster,
//     the other reads the next phase's fragments - the matrix pipe of a SIMD always has one wave feeding it;
//   * a workgroup's last tile requests nothing past its end; the waits of its last two K steps count accordingly.
//     The first R_P0 of every tile waits vmcnt(0): the previous tile's stores sit on the same counter.
// Tile order: plain (see tile_of_block).
//
// Summation order: one fp32 MFMA chain over K per output element (k ascending), rounded to bf16 once -
// the same rounding points as F.linear; the SwiGLU epilogue keeps the reference's three roundings
// (bf16 gate_up output, bf16 silu, bf16 product).
#include <type_traits>

#include "mi_common.hpp"
#include "gemm_tile_index.hpp"
#include "kv_store.hpp"

namespace mi {
using namespace gt;

typedef __attribute__((ext_vector_type(16))) float f32x16;

enum { TEPI_NONE = 0, TEPI_SILU = 1, TEPI_PARTIAL = 2, TEPI_QKV = 3 };

struct TileArgs {
  const uint16_t* x;
  const uint16_t* w;
  const uint16_t* bias;
  uint16_t* y;
  int64_t ldx, ldy;
  int M, N, K;
  int tiles_f, tiles_t;
  float* part;  // split-K (gridDim.y > 1): fp32 partial sums [gridDim.y][M][N] instead of y
  // TEPI_QKV (mi_gemm_bf16_qkv_store): the K and V heads of the packed qkv projection go straight to the paged cache
  struct QkvStore {
    const uint16_t* k_w;       // k-norm weight [128] or null
    const int64_t* positions;  // [M]
    const float* cos_sin;      // [max_position][128]
    uint16_t* k_cache;
    uint16_t* v_cache;
    const int32_t* slots;      // [M] flat slots (block * block_size + offset; negative: not stored)
    float eps;
    int n_q_heads, n_kv_heads, block_size;
  } qkv;
};

// Tuning variants of the eight-wave kernel below (mi_gemm_bf16_ex instantiates the combinations it lists - the staggered
// start V & 1024 and the never-waited DMA V & 256 measured no effect and are not instantiated any more; the timing-only
// ablations lived in round 4's EXPERIMENTS=1 build flavour, removed in round 5 - their numbers are in
// profiles/r04_gemm_ablation.txt; the product entry point uses kDefaultV on the four-wave kernels):
//   V & 2    both waves of a SIMD in lockstep (no one-barrier stagger)
//   V & 4    XCD-rectangle tile order instead of the plain one (tile_of_block; measured equal or slower)
//   V & 8    8-byte stores in the epilogue (no v_permlane32_swap widening)
//   V & 16   one tile per workgroup (grid = tiles, not persistent)
//   V & 1024 workgroups start up to 6 us apart (their output bursts interleave)
//   V & 8192 output rows staged through a per-wave 4 KiB LDS buffer and stored as whole 128-byte lines (a CU's store
//            path moves 49 instead of 17 bytes per cycle that way, tools/ubench/store_pattern.hip; with every CU
//            storing at once the burst is HBM-bound either way: 130.8 vs 132.2 us on qkv, 69.5 vs 64.3 on o_proj)
//   V & 512  ABLATION: no output stores
//   V & 32 / 64 / 128   ABLATIONS for timing only (wrong results): no DMA in the K loop / no fragment reads in the K
//            loop / no workgroup barriers in the K loop
constexpr int kDefaultV = 0;
constexpr int kPersistentGrid = 256;  // one workgroup per CU

#define MI_GT_BARRIER() asm volatile("s_barrier" ::: "memory")

// tile of "block" b.  Plain order (the default): tile b, feature tiles fastest.  The dispatcher places workgroup b
// on XCD b % 8 (a speed assumption only) and a persistent workgroup keeps b % 8 - and, the feature-tile count dividing
// 256 for every shape of the path, its feature tile - over its tiles: its 256 x K weight panel stays in the XCD's L2
// while the activation panels stream through.  The alternative (rect): the 32 workgroups an XCD runs together are
// given a GT x GF rectangle of tiles, GT activation + GF weight panels per K step; measured equal or slower
// (tools/gemm_bench.py), kept as a tuning variant.
__device__ __forceinline__ int tile_of_block(int b, int tiles_t, int tiles_f, bool rect) {
  const int ntiles = tiles_t * tiles_f;
  if (!rect) return b;
  const int gf = tiles_f % 8 == 0 ? 8 : (tiles_f % 4 == 0 ? 4 : (tiles_f % 2 == 0 ? 2 : 1)), gt = 32 / gf;
  const int xcd = b & 7, idx = b >> 3;
  if (ntiles % 256 == 0 && tiles_t % gt == 0) {
    const int r = (idx >> 5) * 8 + xcd, local = idx & 31, rc = tiles_f / gf;
    return ((r / rc) * gt + local / gf) * tiles_f + (r % rc) * gf + local % gf;
  }
  // any other grid: XCD x takes the x-th contiguous run of tiles (bijective for every ntiles)
  const int qn = ntiles >> 3, rn = ntiles & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
}

template <int EPI, bool BIAS, int V>
__global__ __launch_bounds__(512, 2) void gemm_tile_kernel(const TileArgs a) {
  // ONE LDS object (a second one makes hipcc drain the DMA queue before every fragment read)
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fh = wave >> 2, tq = wave & 3, hi = lane >> 5, l31 = lane & 31;
  // split-K: blockIdx.y takes the K steps [kbeg, kbeg + KT) of every tile it is given
  const int KT_all = a.K / BK;
  const int KT = EPI == TEPI_PARTIAL ? KT_all / (int)gridDim.y : KT_all;
  const int64_t kbeg_bytes = EPI == TEPI_PARTIAL ? (int64_t)blockIdx.y * KT * (BK * 2) : 0;
  const int ntiles = a.tiles_f * a.tiles_t, stride = (int)gridDim.x;

  // ---- LDS-DMA sources: byte offset (K step 0) of this lane's 16 bytes of piece (half-tile h, instruction i) of
  // the tile at (m0, n0) ----
  auto src_offset = [&](int h, int i, int m0, int n0) __attribute__((always_inline)) -> uint32_t {
    const int lr = dma_local_row(wave, i, lane), c = dma_chunk(wave, i, lane), r = tile_row(h, lr);
    if (is_weight_half(h)) {
      // SwiGLU: the tile's feature rows are 128 gate rows (A half 0 of both feature halves) and the 128 up rows
      // that pair with them (A half 1)
      const int row = EPI == TEPI_SILU ? ((r >> 6) & 1) * (a.N >> 1) + (n0 >> 1) + (r >> 7) * 64 + (r & 63) : n0 + r;
      return (uint32_t)(((int64_t)min(row, a.N - 1) * a.K + c * 8) * 2);
    }
    return (uint32_t)(((int64_t)min(m0 + r, a.M - 1) * a.ldx + c * 8) * 2);
  };
  int blk = (int)blockIdx.x;  // "block id" of the current tile
  auto coords = [&](int b, int& m0_, int& n0_) __attribute__((always_inline)) {
    const int tid = tile_of_block(b, a.tiles_t, a.tiles_f, (V & 4) != 0);
    m0_ = (tid / a.tiles_f) * TILE_T;
    n0_ = (tid % a.tiles_f) * TILE_F;
  };
  int m0, n0, m0_next = 0, n0_next = 0;
  coords(blk, m0, n0);
  bool has_next = blk + stride < ntiles;
  if (has_next) coords(blk + stride, m0_next, n0_next);
  // src_off[h][i] belongs to the tile that half-tile h is CURRENTLY being fetched for: the stream runs two K steps
  // ahead, so near the end of a tile the offsets of the A0 / B halves, then of A1, switch to the next tile
  uint32_t src_off[4][2];
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) src_off[h][i] = src_offset(h, i, m0, n0);

  // The K steps of a workgroup form ONE stream over all its tiles; `par` is the parity of the stream position of
  // the step being computed (it selects the four LDS slots of that step) and simply keeps toggling across tiles.
  int par = 0;
  // issue piece i of half-tile h of step kt_target (< KT) of the current tile into the slots of parity p
  auto issue_cur = [&](int h, int i, int kt_target, int p) __attribute__((always_inline)) {
    if ((V & 32) && kt_target > 1) return;
    const char* src = reinterpret_cast<const char*>(is_weight_half(h) ? a.w : a.x) + kbeg_bytes + (int64_t)kt_target * (BK * 2) + src_off[h][i];
    char* dst = lds + (p * 4 + h) * HALF_BYTES + dma_block(wave, i) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  // the same for step ke (0 or 1) of the workgroup's NEXT tile (src_off[h] already points at it); without a next
  // tile nothing is requested, and the waits of the last two K steps ask for correspondingly fewer pieces in flight
  auto issue_next = [&](int h, int i, int ke, int p) __attribute__((always_inline)) {
    if ((V & 32) || !has_next) return;
    const char* src = reinterpret_cast<const char*>(is_weight_half(h) ? a.w : a.x) + kbeg_bytes + (int64_t)ke * (BK * 2) + src_off[h][i];
    char* dst = lds + (p * 4 + h) * HALF_BYTES + dma_block(wave, i) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };

  // ---- fragment reads: per-lane byte offsets inside a half-tile, plus the parity's 64 KiB ----
  int kxa[4], kxb[4];
  {
    const int rowoff = (l31 >> 3) * 1024 + (l31 & 7) * 128, sw = swizzle(l31);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int kx = rowoff + ((frag_chunk(kk, hi) ^ sw) << 4);
      kxa[kk] = kx + fh * 8192;  // + h * HALF_BYTES + f * 4096 as immediates
      kxb[kk] = kx + tq * 4096;  // + h * HALF_BYTES
    }
  }
  auto flip_parity = [&]() __attribute__((always_inline)) {
    par ^= 1;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      kxa[kk] ^= 4 * HALF_BYTES;
      kxb[kk] ^= 4 * HALF_BYTES;
    }
  };

  u32x4 Af[2][4], B0[4], B1[4];
  f32x16 acc[4][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][b][r] = 0.f;
  };
  zero_acc();

  bool ablate_reads = false;  // V & 64: only the first K step reads its fragments
  auto read_a = [&](int h) __attribute__((always_inline)) {
    if ((V & 64) && ablate_reads) return;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        Af[f][kk] = *reinterpret_cast<const u32x4*>(lds + h * HALF_BYTES + f * 4096 + kxa[kk]);
  };
  auto read_b = [&](int h, u32x4 (&B)[4]) __attribute__((always_inline)) {
    if ((V & 64) && ablate_reads) return;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) B[kk] = *reinterpret_cast<const u32x4*>(lds + h * HALF_BYTES + kxb[kk]);
  };
  // the four MFMAs of k group kk of a phase: A half ah x (B half bx, B half by) - four independent accumulators
  auto mma_k = [&](int ah, int kk, const u32x4 (&Bx)[4], int bx, const u32x4 (&By)[4], int by) __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      acc[ah * 2 + f][bx] =
          __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(Af[f][kk]), as_frag(Bx[kk]), acc[ah * 2 + f][bx], 0, 0, 0);
      acc[ah * 2 + f][by] =
          __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(Af[f][kk]), as_frag(By[kk]), acc[ah * 2 + f][by], 0, 0, 0);
    }
  };

  // One K step.  MODE 0: steps 0 .. KT - 3 (everything it requests belongs to this tile); MODE 1: step KT - 2 (A0, B0
  // of step kt + 2 are the next tile's step 0); MODE 2: step KT - 1 (B1, A1 of the next tile's step 0, A0, B0 of its
  // step 1).  ALL DMA is issued from the read segments, between the ds_reads: a DMA instruction blocks its wave for
  // ~75 cycles, which inside an MFMA cluster is 75 cycles of idle matrix pipe (measured: 20-26 % of the kernel),
  // while a reading wave's partner is computing.
  auto kstep = [&](int kt, auto modec) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
    auto issue_1 = [&](int h, int i) __attribute__((always_inline)) {  // B half 1 / A half 1, one step ahead
      if (MODE == 2) issue_next(h, i, 0, par ^ 1);
      else issue_cur(h, i, kt + 1, par ^ 1);
    };
    auto issue_2 = [&](int h, int i) __attribute__((always_inline)) {  // A half 0 / B half 0, two steps ahead
      if (MODE == 0) issue_cur(h, i, kt + 2, par);
      else issue_next(h, i, MODE - 1, par);
    };
    // R_P0: A half 0, B half 0, B half 1 of this step; DMA of B half 1 and A half 1 of the next step
    if (MODE == 2 && has_next) {  // from here on B half 1 and A half 1 are fetched for the next tile
#pragma unroll
      for (int h = 2; h < 4; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) src_off[h][i] = src_offset(h, i, m0_next, n0_next);
    }
    read_a(0);
    issue_1(2, 0);
    issue_1(2, 1);
    read_b(1, B0);
    issue_1(3, 0);
    read_b(2, B1);
    issue_1(3, 1);
    __builtin_amdgcn_sched_barrier(0);
    // a tile's first wait: the previous tile's stores share the counter with the loads, drain it once
    if (V & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (V & 256) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ablation: DMA never waited for
    else if (kt == 0 || (V & 32) || (MODE == 2 && !has_next))  // (a workgroup's last K step: nothing newer is in flight)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // M_P0
    mma_k(0, 0, B0, 0, B1, 1);
    mma_k(0, 1, B0, 0, B1, 1);
    mma_k(0, 2, B0, 0, B1, 1);
    mma_k(0, 3, B0, 0, B1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(V & 128)) MI_GT_BARRIER();
    // R_P1: A half 1 of this step; DMA of A half 0 and B half 0 two steps ahead
    if ((MODE == 1 || (MODE == 2 && KT == 1)) && has_next) {  // from here on A half 0 and B half 0 are the next tile's
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) src_off[h][i] = src_offset(h, i, m0_next, n0_next);
    }
    read_a(3);
    issue_2(0, 0);
    issue_2(0, 1);
    issue_2(1, 0);
    issue_2(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (V & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (V & 256) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if ((V & 32) || (MODE == 2 && !has_next)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (MODE == 1 && !has_next)  // the last tile's second-to-last step requested nothing here: only A1 of the last step flies
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // M_P1
    mma_k(1, 0, B1, 1, B0, 0);
    mma_k(1, 1, B1, 1, B0, 0);
    mma_k(1, 2, B1, 1, B0, 0);
    mma_k(1, 3, B1, 1, B0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (!(V & 128)) MI_GT_BARRIER();
    flip_parity();
    ablate_reads = true;
  };

  // ---- epilogue of the tile at (m0, n0): lane (hi, l31) holds token acc_token(...), features 8 rq + 4 hi + {0..3} of
  // each 32-row fragment.  Two register quads (rq, rq + 1) are exchanged between the half-waves
  // (v_permlane32_swap) so that every lane stores 8 consecutive features = 16 bytes: half the store instructions
  // for the same bytes.
  const int n_out = EPI == TEPI_SILU ? a.N >> 1 : a.N;
  const bool wide = !(V & 8) && n_out % 8 == 0 && a.ldy % 8 == 0;
  char* const stg = lds + SLOTS * HALF_BYTES + wave * STAGE_BYTES;
  auto epilogue = [&]() __attribute__((always_inline)) {
    if (V & 512) {  // ablation: no output (the accumulators are kept alive by an empty asm)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" ::"v"(acc[j][b]));
      return;
    }
    if (EPI == TEPI_PARTIAL) {  // fp32 sums of this K slice: [slice][token][feature], 16 bytes per lane and register quad
#pragma unroll
      for (int bh = 0; bh < 2; ++bh) {
        const int tok = m0 + acc_token(tq, bh, l31);
        float* prow = a.part + ((int64_t)blockIdx.y * a.M + min(tok, a.M - 1)) * a.N;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int col = n0 + acc_feature(fh, j >> 1, j & 1, 4 * rq, hi);
            if (tok < a.M && col < a.N)
              *reinterpret_cast<f32x4*>(prow + col) =
                  f32x4{acc[j][bh][4 * rq], acc[j][bh][4 * rq + 1], acc[j][bh][4 * rq + 2], acc[j][bh][4 * rq + 3]};
          }
      }
      return;
    }
#pragma unroll
    for (int bh = 0; bh < 2; ++bh) {
      const int tok = m0 + acc_token(tq, bh, l31);
      const bool tok_ok = tok < a.M;
      uint16_t* yrow = a.y + (int64_t)min(tok, a.M - 1) * a.ldy;
#pragma unroll
      for (int j = 0; j < (EPI == TEPI_SILU ? 2 : 4); ++j) {
        // first column of this fragment for hi = 0, rq = 0 (tile feature row fh * 128 + (j >> 1) * 64 + (j & 1) * 32)
        const int col0 = EPI == TEPI_SILU ? (n0 >> 1) + fh * 64 + j * 32 : n0 + acc_feature(fh, j >> 1, j & 1, 0, 0);
        u32x2 pk[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int col = col0 + 8 * rq + 4 * hi;
          float o[4];
          if (EPI == TEPI_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // the gate_up GEMM output rounded to bf16 as the unfused path, then silu in fp32 on the hardware
              // exp2 / reciprocal (v_exp_f32, v_rcp_f32: a few fp32 ulps, i.e. below the bf16 rounding that follows
              // except at ties; the libm expf + IEEE division of mi_silu_mul cost the 256 x 256 tile's epilogue - which
              // nothing overlaps - 15 % of the kernel)
              const float gb = rbf(acc[j][bh][4 * rq + e]);
              const float sb = rbf(gb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb * -1.4426950408889634f)));
              o[e] = sb * rbf(acc[2 + j][bh][4 * rq + e]);
            }
          } else {
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
          pk[rq] = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
        }
        if (wide) {
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            // lanes 32..63 of quad 2 p2 <-> lanes 0..31 of quad 2 p2 + 1: lanes < 32 then hold features
            // 16 p2 .. 16 p2 + 7 of the fragment, lanes >= 32 features 16 p2 + 8 .. 16 p2 + 15
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
            const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
            if (!(V & 8192)) {  // straight from the accumulator layout: 32 rows x 32 bytes per instruction (the default)
              const int col = col0 + 8 * (2 * p2 + hi);
              if (tok_ok && col < n_out) *reinterpret_cast<u32x4*>(yrow + col) = v;
            } else {
              // into the wave's stage: row l31, 16-byte chunk (j & 1) * 4 + 2 p2 + hi of a 128-byte line, XOR-swizzled by
              // the row so that the eight lanes of a ds_write_b128 group hit eight different chunks
              const int chunk = ((j & 1) * 4 + 2 * p2 + hi) ^ (l31 & 7);
              *reinterpret_cast<u32x4*>(stg + l31 * 128 + chunk * 16) = v;
            }
          }
          if ((V & 8192) && (j & 1)) {
            // Whole cache lines to memory.  Stored straight from the accumulator layout an instruction writes 32 bytes
            // of 32 different rows and a CU's store path moves 17 bytes per cycle; as complete 128-byte lines - eight
            // lanes per row, eight rows per instruction - it moves 49 (tools/ubench/store_pattern.hip).  The 32 x 64
            // features of two fragments pass through the wave's private 4 KiB stage; LDS operations of one wave
            // execute in order: no barrier.
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = 8 * r + (lane >> 3), c = lane & 7;
              const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((c ^ (row & 7)) << 4));
              const int t2 = m0 + acc_token(tq, bh, row), col = col0 - 32 + 8 * c;
              if (t2 < a.M && col < n_out) *reinterpret_cast<u32x4*>(a.y + (int64_t)t2 * a.ldy + col) = v;
            }
          }
        } else {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int col = col0 + 8 * rq + 4 * hi;
            if (tok_ok && col < n_out) *reinterpret_cast<u32x2*>(yrow + col) = pk[rq];
          }
        }
      }
    }
  };

  if (V & 1024) {  // workgroups start up to 6 us apart, in four groups per XCD: their output bursts interleave
    for (int d = ((int)blockIdx.x >> 3) & 3; d > 0; --d) __builtin_amdgcn_s_sleep(63);
  }
  // prologue, in the stream's order (R_P1 issues A0, B0 two steps ahead; R_P0 issues B1, A1 one step ahead):
  // A0, B0 of step 0 | B1, A1 of step 0 | A0, B0 of step 1; everything but the six newest pieces - that is A0, B0,
  // B1 of step 0 - landed and published.  (K = 64: "step 1" is the next tile's step 0.)
#pragma unroll
  for (int h = 0; h < 4; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_cur(h, i, 0, 0);
  if (KT > 1) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) issue_cur(h, i, 1, 1);
  } else {
    if (has_next) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) src_off[h][i] = src_offset(h, i, m0_next, n0_next);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) issue_next(h, i, 0, 1);
  }
  __builtin_amdgcn_sched_barrier(0);
  if (KT > 1 || has_next) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");  // one K step, one tile: only A half 1 still flies
  if (!(V & 2) && fh == 1) MI_GT_BARRIER();  // the second wave of every SIMD runs one barrier behind the first
  __builtin_amdgcn_sched_barrier(0);

  for (;;) {
    for (int kt = 0; kt < KT - 2; ++kt) kstep(kt, std::integral_constant<int, 0>{});
    if (KT > 1) kstep(KT - 2, std::integral_constant<int, 1>{});
    kstep(KT - 1, std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    epilogue();
    __builtin_amdgcn_sched_barrier(0);
    if (!has_next) break;
    zero_acc();
    blk += stride;
    m0 = m0_next;
    n0 = n0_next;
    has_next = blk + stride < ntiles;
    if (has_next) coords(blk + stride, m0_next, n0_next);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the past-the-end loads must not outlive the workgroup's LDS
  if (!(V & 2) && fh == 0) MI_GT_BARRIER();         // barrier counts match again
}

#include "gemm_w4_kernel.hpp"
#include "gemm_w16_kernel.hpp"

// ---------------------------------------------------------------------------
// The same GEMM on 128 x 128 tiles, for shapes whose 256 x 256 tiles cannot fill the chip (64 < M <= ~2048 rows:
// large decode batches, short prompts).  Four waves, each 64 features x 64 tokens (2 x 2 accumulators); one K step of
// the tile is two 16 KiB half-tiles in the layout of gemm_tile_index.hpp (A: local row = tile feature row, B: local
// row = tile token row) in a ring of NST stages.  One barrier per K step: "my pieces of step t have landed" (a COUNTED
// wait: the pieces of the NST - 2 steps behind it may still fly) -> barrier (everybody's have, and everybody is done
// reading step t - 1) -> request step t + NST - 1 into the buffer step t - 1 used -> fragments of step t -> sixteen
// MFMAs.  These shapes have at most a workgroup or two per CU and a K loop of 16-80 steps: with one step of
// prefetch (NST = 2, two workgroups per CU) every step waits out most of a DMA round trip (0.75 us per step
// measured); NST = 4 (128 KiB, one workgroup per CU) keeps three steps in flight.
// Same summation order and epilogues as the large kernel: results are bit-identical.
// SwiGLU: the tile's 128 weight rows are, per wave, 32 gate rows (fragment 0) and the 32 up rows that pair with them.
// ---------------------------------------------------------------------------
constexpr int MID_STAGE = 2 * HALF_BYTES;  // (MID_F, MID_T and the index arithmetic: gemm_tile_index.hpp)

template <int EPI, bool BIAS, int NST>
__global__ __launch_bounds__(256, NST == 2 ? 2 : 1) void gemm_mid_kernel(const TileArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[NST * MID_STAGE];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fw = wave >> 1, tw = wave & 1, hi = lane >> 5, l31 = lane & 31;
  const int KT_all = a.K / BK;
  const int KT = EPI == TEPI_PARTIAL ? KT_all / (int)gridDim.y : KT_all;
  const int64_t kbeg_bytes = EPI == TEPI_PARTIAL ? (int64_t)blockIdx.y * KT * (BK * 2) : 0;
  const int n0 = ((int)blockIdx.x % a.tiles_f) * MID_F, m0 = ((int)blockIdx.x / a.tiles_f) * MID_T;

  // LDS-DMA: wave w moves the 1 KiB blocks 4 w .. 4 w + 3 (8 local rows each) of both half-tiles
  const char* srcA[4];
  const char* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int lr = mid_dma_local_row(wave, i, lane), c = mid_dma_chunk(wave, i, lane);
    const int rowA = mid_weight_row(lr, n0, a.N, EPI == TEPI_SILU);
    srcA[i] = reinterpret_cast<const char*>(a.w) + kbeg_bytes + ((int64_t)min(rowA, a.N - 1) * a.K + c * 8) * 2;
    srcB[i] = reinterpret_cast<const char*>(a.x) + kbeg_bytes + ((int64_t)min(m0 + lr, a.M - 1) * a.ldx + c * 8) * 2;
  }
  auto issue = [&](int kt, int p) __attribute__((always_inline)) {
    char* dst = lds + p * MID_STAGE + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + (int64_t)kt * (BK * 2)),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + (int64_t)kt * (BK * 2)),
                                       (__attribute__((address_space(3))) void*)(dst + HALF_BYTES + i * 1024), 16, 0, 0);
    }
  };
  int kx[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
    kx[kk] = (l31 >> 3) * 1024 + (l31 & 7) * 128 + ((frag_chunk(kk, hi) ^ swizzle(l31)) << 4);

  f32x16 acc[2][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][b][r] = 0.f;

#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) issue(s, s);
  int p = 0, pn = NST - 1;  // ring slots of step kt and of step kt + NST - 1
  for (int kt = 0; kt < KT; ++kt) {
    // requested so far: steps <= kt + NST - 2; eight pieces per step and wave, completing in order
    const int behind = min(KT - 1 - kt, NST - 2);
    if (behind >= 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NST - 1 < KT) issue(kt + NST - 1, pn);
    const char* sa = lds + p * MID_STAGE + fw * 8192;
    const char* sb = lds + p * MID_STAGE + HALF_BYTES + tw * 8192;
    u32x4 A[2][4], B[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int f = 0; f < 2; ++f) A[f][kk] = *reinterpret_cast<const u32x4*>(sa + f * 4096 + kx[kk]);
#pragma unroll
      for (int b = 0; b < 2; ++b) B[b][kk] = *reinterpret_cast<const u32x4*>(sb + b * 4096 + kx[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[f][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(A[f][kk]), as_frag(B[b][kk]), acc[f][b], 0, 0, 0);
    p = p + 1 == NST ? 0 : p + 1;
    pn = pn + 1 == NST ? 0 : pn + 1;
  }

  // epilogue: lane (hi, l31) holds token tw * 64 + b * 32 + l31, features 8 rq + 4 hi + {0..3} of each 32-row fragment
  const int n_out = EPI == TEPI_SILU ? a.N >> 1 : a.N;
  const bool wide = n_out % 8 == 0 && a.ldy % 8 == 0;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int tok = m0 + tw * 64 + b * 32 + l31;
    const bool tok_ok = tok < a.M;
    if (EPI == TEPI_PARTIAL) {
      float* prow = a.part + ((int64_t)blockIdx.y * a.M + min(tok, a.M - 1)) * a.N;
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int col = n0 + fw * 64 + f * 32 + 8 * rq + 4 * hi;
          if (tok_ok && col < a.N)
            *reinterpret_cast<f32x4*>(prow + col) =
                f32x4{acc[f][b][4 * rq], acc[f][b][4 * rq + 1], acc[f][b][4 * rq + 2], acc[f][b][4 * rq + 3]};
        }
      continue;
    }
    uint16_t* yrow = a.y + (int64_t)min(tok, a.M - 1) * a.ldy;
#pragma unroll
    for (int f = 0; f < (EPI == TEPI_SILU ? 1 : 2); ++f) {
      const int col0 = EPI == TEPI_SILU ? (n0 >> 1) + fw * 32 : n0 + fw * 64 + f * 32;
      u32x2 pk[4];
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int col = col0 + 8 * rq + 4 * hi;
        float o[4];
        if (EPI == TEPI_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {  // the roundings and the silu form of the large kernel's epilogue
            const float gb = rbf(acc[0][b][4 * rq + e]);
            const float sb = rbf(gb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb * -1.4426950408889634f)));
            o[e] = sb * rbf(acc[1][b][4 * rq + e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[f][b][4 * rq + e];
          if (BIAS && col < n_out) {
            const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
            o[0] += lo_bf(bw[0]);
            o[1] += hi_bf(bw[0]);
            o[2] += lo_bf(bw[1]);
            o[3] += hi_bf(bw[1]);
          }
        }
        pk[rq] = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
      }
      if (wide) {
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
          const int col = col0 + 8 * (2 * p2 + hi);
          if (tok_ok && col < n_out) *reinterpret_cast<u32x4*>(yrow + col) = u32x4{sx[0], sy[0], sx[1], sy[1]};
        }
      } else {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int col = col0 + 8 * rq + 4 * hi;
          if (tok_ok && col < n_out) *reinterpret_cast<u32x2*>(yrow + col) = pk[rq];
        }
      }
    }
  }
}

// sum of the split-K slices in slice order, (+ bias), one rounding to bf16: four features per thread
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int nsplit,
                                                                   const uint16_t* __restrict__ bias,
                                                                   uint16_t* __restrict__ y, int64_t ldy, int M, int N) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = N >> 2;
  if (idx >= (int64_t)M * nq) return;
  const int row = (int)(idx / nq), col = (int)(idx % nq) * 4;
  f32x4 s = *reinterpret_cast<const f32x4*>(part + (int64_t)row * N + col);
  for (int k = 1; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(part + ((int64_t)k * M + row) * N + col);
  if (bias) {
    const u32x2 bw = *reinterpret_cast<const u32x2*>(bias + col);
    s[0] += lo_bf(bw[0]);
    s[1] += hi_bf(bw[0]);
    s[2] += lo_bf(bw[1]);
    s[3] += hi_bf(bw[1]);
  }
  *reinterpret_cast<u32x2*>(y + (int64_t)row * ldy + col) = u32x2{pack_bf(s[0], s[1]), pack_bf(s[2], s[3])};
}

// K slices for a shape whose tiles cannot fill the chip (few token rows x a narrow projection): the smallest
// power of two that brings tiles x slices to ~one workgroup per CU, every slice at least four K steps deep
static int tile_ksplit(int M, int N, int K) {
  const int ntiles = ((M + TILE_T - 1) / TILE_T) * ((N + TILE_F - 1) / TILE_F), kt = K / BK;
  int ks = 1;
  while (ntiles * ks * 2 <= kPersistentGrid && kt % (ks * 2) == 0 && kt / (ks * 2) >= 4 && ks < 16) ks *= 2;
  return ks;
}

// The large-M kernel: gemm_w16_kernel (four waves, v_mfma_f32_16x16x32_bf16), with whole-line output stores
// (kLineStores) wherever the output allows it: full feature tiles, 16-byte aligned rows, 32-bit byte offsets.
// Tuning / comparison variants: V & kDirectStores keeps the direct stores; V & kMfma32 = the same kernel on
// v_mfma_f32_32x32x16_bf16 (gemm_w4_kernel: the first form of round 4, still the qkv-store kernel); V & kEightWaves = the
// eight-wave ping-pong kernel of rounds 2-3.
constexpr int kEightWaves = 1 << 20, kMfma32 = 1 << 19, kLineStores = 32768, kDirectStores = 1 << 18;
template <int EPI, bool BIAS, int V = kDefaultV>
static int launch_tile(const TileArgs& a, hipStream_t st, int ksplit = 1) {
  // persistent workgroups: one per CU; the grid is a multiple of 8, so a workgroup stays in one XCD class
  const int ntiles = a.tiles_f * a.tiles_t;
  const bool persistent = !(V & 16) && ntiles > kPersistentGrid;
  const dim3 grid(persistent ? kPersistentGrid : ntiles, ksplit);
  constexpr int VK = V & ~(kDirectStores | kMfma32 | kEightWaves);  // the kernel's own flags
  const bool lines = EPI != TEPI_PARTIAL && !(V & (kDirectStores | 8 | 512)) && a.N % TILE_F == 0 && a.ldy % 8 == 0 &&
                     (int64_t)a.M * a.ldy * 2 < ((int64_t)1 << 32) && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
  if constexpr ((V & kEightWaves) != 0) {
    hipLaunchKernelGGL((gemm_tile_kernel<EPI, BIAS, VK>), grid, dim3(512), 0, st, a);
  } else if constexpr ((V & kMfma32) != 0) {
    if (lines) hipLaunchKernelGGL((gemm_w4_kernel<EPI, BIAS, VK | kLineStores>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_w4_kernel<EPI, BIAS, VK>), grid, dim3(256), 0, st, a);
  } else {
    if (lines) hipLaunchKernelGGL((gemm_w16_kernel<EPI, BIAS, VK | kLineStores>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_w16_kernel<EPI, BIAS, VK>), grid, dim3(256), 0, st, a);
  }
  return check_launch();
}

// ---- which kernel, how many K slices ----
// The 128-tile kernel takes the shapes whose 256-tiles are too few for the chip; both kernels give the same bits,
// the choice is a matter of time only.  Modelled time of a launch: rounds of workgroups x K steps x time per step
// (measured: tools/gemm_bench.py), plus - with K slices - the fp32 partial sums out and back and the second launch.
constexpr double kMidStepUs = 0.5, kMidPairStepUs = 0.62, kLaunchUs = 3.8, kPartialBytesPerUs = 4.0e6;

// fewer 256-tiles than CUs: the 128-tile kernel
static bool use_mid(int M, int N, int K) {
  return (int64_t)((M + TILE_T - 1) / TILE_T) * ((N + TILE_F - 1) / TILE_F) < kPersistentGrid;
}

static int mid_ksplit(int M, int N, int K) {
  const int tiles = ((M + MID_T - 1) / MID_T) * ((N + MID_F - 1) / MID_F), kt = K / BK;
  double best = 1e30;
  int best_ks = 1;
  for (int ks = 1; ks <= 16; ks *= 2) {
    if (kt % ks || (ks > 1 && kt / ks < 2)) break;
    // one workgroup on a CU takes ~0.5 us per K step (DMA issue and MFMAs of a wave do not overlap), two sharing
    // it ~0.62 us for both
    const int per_cu = (tiles * ks + kPersistentGrid - 1) / kPersistentGrid;
    double t = (per_cu == 1 ? kMidStepUs : (per_cu + 1) / 2 * kMidPairStepUs) * (kt / ks) + kLaunchUs;
    if (ks > 1) t += 2.0 * ks * (double)M * N * sizeof(float) / kPartialBytesPerUs + kLaunchUs;
    if (t < best) {
      best = t;
      best_ks = ks;
    }
  }
  return best_ks;
}

// NST = 0: by workgroup count - at most one workgroup per CU: the deep ring (nothing else hides the DMA round trip);
// more: the two-stage ring, two workgroups per CU (measured, tools/gemm_bench.py GEMM_MID=1: 22 vs 27 us on
// 1024 x 6144 x 1024, 31 vs 34 the other way round on 1024 x 1024 x 3072)
template <int EPI, bool BIAS, int NST = 0>
static int launch_mid(TileArgs a, hipStream_t st, int ksplit = 1) {
  a.tiles_f = (a.N + MID_F - 1) / MID_F;
  a.tiles_t = (a.M + MID_T - 1) / MID_T;
  const dim3 grid(a.tiles_f * a.tiles_t, ksplit);
  if (NST == 0) {
    if ((int64_t)grid.x * grid.y > kPersistentGrid) hipLaunchKernelGGL((gemm_mid_kernel<EPI, BIAS, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_mid_kernel<EPI, BIAS, 4>), grid, dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((gemm_mid_kernel<EPI, BIAS, (NST ? NST : 2)>), grid, dim3(256), 0, st, a);
  }
  return check_launch();
}

static int gemm_ksplit(int M, int N, int K) { return use_mid(M, N, K) ? mid_ksplit(M, N, K) : tile_ksplit(M, N, K); }

static int check_tile_gemm(const void* x, int64_t ldx, const void* w, const void* bias, const void* y, int64_t ldy,
                           int M, int N, int K, int epilogue) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0 || ldx < K) return MI_EINVAL;
  if (epilogue != 0 && epilogue != 1) return MI_EINVAL;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias))) return MI_EINVAL;
  if (K % BK || N % 4 || ldx % 8 || ldy % 4) return MI_EUNSUPPORTED;
  if (epilogue == 1 && (bias || (N / 2) % 128)) return MI_EUNSUPPORTED;  // (the 128-tile kernel alone would take % 64)
  if (ldy < (epilogue == 1 ? N / 2 : N)) return MI_EINVAL;
  // the DMA sources are 32-bit byte offsets from the operand bases
  // (and 0xF0000000 must lie beyond both: the four-wave kernel's "fetch nothing" offset)
  if (((int64_t)N + 256) * K * 2 >= (int64_t)1 << 31 || ((int64_t)M + 256) * ldx * 2 >= (int64_t)1 << 31) return MI_EUNSUPPORTED;
  return MI_OK;
}

}  // namespace mi

using namespace mi;

// The shape contract of mi_gemm_bf16 as a query (the same tests as check_tile_gemm): the largest M one launch takes on
// a [N][K] weight with activation row stride ldx, 0 when the weight shape itself is refused.
extern "C" int64_t mi_gemm_bf16_max_rows(int N, int K, int64_t ldx) {
  if (N <= 0 || K <= 0 || K % BK || N % 4 || ldx < K || ldx % 8) return 0;
  if (((int64_t)N + 256) * K * 2 >= (int64_t)1 << 31) return 0;
  const int64_t m = (((int64_t)1 << 30) - 1) / ldx - 256;  // (M + 256) * ldx * 2 < 2^31
  return m > 0 ? m : 0;
}

extern "C" size_t mi_gemm_bf16_workspace(int M, int N, int K, int epilogue) {
  if (M <= 0 || N <= 0 || K <= 0 || K % BK || epilogue != 0) return 0;
  const int ks = gemm_ksplit(M, N, K);
  return ks > 1 ? (size_t)ks * M * N * sizeof(float) : 0;
}

extern "C" int mi_gemm_bf16(const mi_bf16* x, int64_t ldx, const mi_bf16* w, const mi_bf16* bias, mi_bf16* y,
                            int64_t ldy, int M, int N, int K, int epilogue, void* workspace, size_t ws_bytes,
                            mi_stream stream) {
  const int rc = check_tile_gemm(x, ldx, w, bias, y, ldy, M, N, K, epilogue);
  if (rc != MI_OK || M == 0) return rc;
  TileArgs a{x, w, bias, y, ldx, ldy, M, N, K, (N + TILE_F - 1) / TILE_F, (M + TILE_T - 1) / TILE_T, nullptr};
  hipStream_t st = S(stream);
  const bool mid = use_mid(M, N, K);
  if (epilogue == 1) return mid ? launch_mid<TEPI_SILU, false>(a, st) : launch_tile<TEPI_SILU, false>(a, st);
  const int ks = gemm_ksplit(M, N, K);
  if (ks > 1) {  // two launches: K slices as fp32 partial sums, then their sum (+ bias) rounded once
    if (!workspace || !aligned16(workspace) || ws_bytes < mi_gemm_bf16_workspace(M, N, K, 0)) return MI_EWORKSPACE;
    a.part = static_cast<float*>(workspace);
    const int rc2 = mid ? launch_mid<TEPI_PARTIAL, false>(a, st, ks) : launch_tile<TEPI_PARTIAL, false>(a, st, ks);
    if (rc2 != MI_OK) return rc2;
    const int64_t quads = (int64_t)M * (N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, a.part, ks, bias, y,
                       ldy, M, N);
    return check_launch();
  }
  if (mid) return bias ? launch_mid<TEPI_NONE, true>(a, st) : launch_mid<TEPI_NONE, false>(a, st);
  return bias ? launch_tile<TEPI_NONE, true>(a, st) : launch_tile<TEPI_NONE, false>(a, st);
}

// The packed qkv projection with the K / V heads stored straight into the paged cache (TEPI_QKV, gemm_w4_kernel.hpp)
extern "C" int mi_gemm_bf16_qkv_store(const mi_bf16* x, int64_t ldx, const mi_bf16* w, const mi_bf16* bias, mi_bf16* qkv,
                                      int64_t ldy, int M, int N, int K, const mi_bf16* k_w, float eps,
                                      const int64_t* positions, const float* cos_sin, mi_bf16* k_cache,
                                      mi_bf16* v_cache, const int32_t* slots, int n_q_heads, int n_kv_heads,
                                      int head_dim, int block_size, mi_stream stream) {
  const int rc = check_tile_gemm(x, ldx, w, bias, qkv, ldy, M, N, K, 0);
  if (rc != MI_OK) return rc;
  if (!positions || !cos_sin || !k_cache || !v_cache || !slots || n_q_heads <= 0 || n_kv_heads <= 0) return MI_EINVAL;
  if (!aligned16(k_cache) || !aligned16(v_cache) || !aligned16(slots) || !aligned16(cos_sin) || (k_w && !aligned16(k_w))) return MI_EINVAL;
  // one wave = one head; whole feature tiles and 16-byte aligned output rows (the whole-line stores of the q heads)
  if (head_dim != 128 || N != (n_q_heads + 2 * n_kv_heads) * 128 || N % TILE_F || ldy % 8 || block_size <= 0 || block_size % 16)
    return MI_EUNSUPPORTED;
  if (use_mid(M, N, K) || (int64_t)M * ldy * 2 >= ((int64_t)1 << 32)) return MI_EUNSUPPORTED;  // the large-M kernel only
  if (M == 0) return MI_OK;
  TileArgs a{x, w, bias, qkv, ldx, ldy, M, N, K, N / TILE_F, (M + TILE_T - 1) / TILE_T, nullptr,
             {k_w, positions, cos_sin, k_cache, v_cache, slots, eps, n_q_heads, n_kv_heads, block_size}};
  const int ntiles = a.tiles_f * a.tiles_t;
  const dim3 grid(ntiles > kPersistentGrid ? kPersistentGrid : ntiles);
  // (the 32 x 32 x 16 form of the kernel: the K-head arithmetic is written for its accumulator layout)
  if (bias) hipLaunchKernelGGL((gemm_w4_kernel<TEPI_QKV, true, kLineStores>), grid, dim3(256), 0, S(stream), a);
  else hipLaunchKernelGGL((gemm_w4_kernel<TEPI_QKV, false, kLineStores>), grid, dim3(256), 0, S(stream), a);
  return check_launch();
}

// tuning entry point (tools/gemm_bench.py): variant = schedule flags V, no bias, plain epilogue
extern "C" int mi_gemm_bf16_ex(const mi_bf16* x, int64_t ldx, const mi_bf16* w, mi_bf16* y, int64_t ldy, int M, int N,
                               int K, int variant, mi_stream stream) {
  const int rc = check_tile_gemm(x, ldx, w, nullptr, y, ldy, M, N, K, 0);
  if (rc != MI_OK || M == 0) return rc;
  const TileArgs a{x, w, nullptr, y, ldx, ldy, M, N, K, (N + TILE_F - 1) / TILE_F, (M + TILE_T - 1) / TILE_T, nullptr};
  hipStream_t st = S(stream);
  switch (variant) {
    // the product kernel (four waves, 16 x 16 x 32 MFMAs) and its variants
    case 0: return launch_tile<TEPI_NONE, false, 0>(a, st);
    case 4: return launch_tile<TEPI_NONE, false, 4>(a, st);
    case 8: return launch_tile<TEPI_NONE, false, 8>(a, st);
    case 16: return launch_tile<TEPI_NONE, false, 16>(a, st);
    case kDirectStores: return launch_tile<TEPI_NONE, false, kDirectStores>(a, st);
    // the same kernel on 32 x 32 x 16 MFMAs
    case kMfma32: return launch_tile<TEPI_NONE, false, kMfma32>(a, st);
    case kMfma32 + kDirectStores: return launch_tile<TEPI_NONE, false, kMfma32 + kDirectStores>(a, st);
    // the eight-wave kernel of rounds 2-3 and its variants
    case kEightWaves: return launch_tile<TEPI_NONE, false, kEightWaves>(a, st);
    case kEightWaves + 2: return launch_tile<TEPI_NONE, false, kEightWaves + 2>(a, st);
    case kEightWaves + 16: return launch_tile<TEPI_NONE, false, kEightWaves + 16>(a, st);
    case kEightWaves + 8192: return launch_tile<TEPI_NONE, false, kEightWaves + 8192>(a, st);
    case 65536: return launch_mid<TEPI_NONE, false, 4>(a, st);  // the 128-tile kernel, no K slices, four-stage ring
    case 65538: return launch_mid<TEPI_NONE, false, 2>(a, st);  // ... with a two-stage ring, two workgroups per CU
    case 65539: return launch_mid<TEPI_NONE, false, 3>(a, st);
    default: return MI_EUNSUPPORTED;
  }
}
