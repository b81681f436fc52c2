// Entry points of the five-launch decode chain (gemm_chain5_kernel.hpp): the row-parallel projection with the residual
// add and the RMSNorm statistic in its epilogue, the column-parallel projection that normalises on load, the final norm.
#include "gemm_chain5_kernel.hpp"

using namespace mi;

namespace mi {

struct RowstatArgs {
  const uint16_t *x, *w, *residual;
  uint16_t* residual_out;
  float *s_out, *stat;
  int M, N, K, ksplit;
  hipStream_t st;
  unsigned long long* stamps;
};

// (instrumented kernels exist for the decode chain's own geometries only: sixteen waves, K <= 3072)
template <int WAVES, int SUB>
static bool launch_rowstat(const RowstatArgs& a) {
  const dim3 grid(a.N / 16, (a.M + kStatRows - 1) / kStatRows);
  if (a.stamps) {
    if constexpr (WAVES == 16 && SUB <= 3)
      hipLaunchKernelGGL((gemm_rowstat_kernel<WAVES, SUB, true>), grid, dim3(WAVES * 64), 0, a.st, a.x, a.w, a.residual,
                         a.residual_out, a.s_out, a.stat, a.M, a.N, a.K, a.ksplit, a.stamps);
    else
      return false;
  } else
    hipLaunchKernelGGL((gemm_rowstat_kernel<WAVES, SUB, false>), grid, dim3(WAVES * 64), 0, a.st, a.x, a.w, a.residual,
                       a.residual_out, a.s_out, a.stat, a.M, a.N, a.K, a.ksplit, nullptr);
  return true;
}

template <int WAVES>
static bool rowstat_sub(const RowstatArgs& a, int sub) {
  switch (sub) {
    case 1: return launch_rowstat<WAVES, 1>(a);
    case 2: return launch_rowstat<WAVES, 2>(a);
    case 3: return launch_rowstat<WAVES, 3>(a);
    case 4: return launch_rowstat<WAVES, 4>(a);
    case 5: return launch_rowstat<WAVES, 5>(a);
    case 6: return launch_rowstat<WAVES, 6>(a);
    case 8: return launch_rowstat<WAVES, 8>(a);
    default: return false;
  }
}

// K / 64 sub-slices over 16 waves when that divides (12, 8, 4 otherwise), at most 8 per wave
static int rowstat(const RowstatArgs& a) {
  if (a.K % 64) return MI_EUNSUPPORTED;
  const int nsub = a.K / 64;
  if (a.ksplit < 1 || a.ksplit > 16 || nsub % a.ksplit) return MI_EUNSUPPORTED;
  bool ok = false;
  if (nsub % 16 == 0) ok = rowstat_sub<16>(a, nsub / 16);
  if (!ok && nsub % 12 == 0) ok = rowstat_sub<12>(a, nsub / 12);
  if (!ok && nsub % 8 == 0) ok = rowstat_sub<8>(a, nsub / 8);
  if (!ok && nsub % 4 == 0) ok = rowstat_sub<4>(a, nsub / 4);
  return ok ? check_launch() : MI_EUNSUPPORTED;
}

struct NormedArgs {
  NormArgs nm;
  const uint16_t* w;
  uint16_t* y;
  int M, N, K;
  hipStream_t st;
  unsigned long long* stamps;
};

template <int MT, int RT, int WAVES, int STEPS, int EPI>
static bool launch_normed(const NormedArgs& a) {
  const size_t slot = (size_t)RT * MT * 1024 > (size_t)x_slab_bytes(MT) ? (size_t)RT * MT * 1024 : (size_t)x_slab_bytes(MT);
  const int tiles = a.N / 16;
  const dim3 grid(EPI == EPI_SILU ? tiles / 2 : tiles / RT, 1, (a.M + kSkinnyRows - 1) / kSkinnyRows);
  if (a.stamps) {
    if constexpr (MT == 2 && WAVES == 16 && STEPS == 2)
      hipLaunchKernelGGL((gemm_normed_kernel<MT, RT, WAVES, STEPS, EPI, true>), grid, dim3(WAVES * 64), WAVES * slot, a.st,
                         a.nm, a.w, a.y, a.M, a.N, a.K, a.stamps);
    else
      return false;
  } else
    hipLaunchKernelGGL((gemm_normed_kernel<MT, RT, WAVES, STEPS, EPI, false>), grid, dim3(WAVES * 64), WAVES * slot, a.st,
                       a.nm, a.w, a.y, a.M, a.N, a.K, nullptr);
  return true;
}

// the wave count and K-slice mi_gemm_bf16_packed takes for these shapes (pick_waves: slices of 64 when K / 64 <= 16,
// else of 128, else sixteen waves), so that the sums are the same sums
template <int MT, int RT, int EPI>
static int normed_waves(const NormedArgs& a) {
  const int K = a.K;
  int waves = 0;
  if (K % 64 == 0 && K / 64 <= 16) waves = K / 64;
  else if (K % 128 == 0) waves = K / 128 <= 16 ? K / 128 : (K / 128 == 24 ? 12 : 16);
  if (waves == 0 || K % waves || (K / waves) % 64) return MI_EUNSUPPORTED;
  // sixteen (twelve) waves have 128 (168) registers each: blocks of 64 and at most 32 activation rows; the few-wave
  // geometries take blocks of 128 when the slice allows
  if (waves > 8 && MT > 2) return MI_EUNSUPPORTED;
  const bool four = (K / waves) % 128 == 0 && waves <= 8 && MT * RT <= 4;
#define MI_NORMED_GO(W)                                          \
  case W:                                                        \
    if constexpr (W <= 8 && MT * RT <= 4) {                      \
      if (four) return launch_normed<MT, RT, W, 4, EPI>(a) ? check_launch() : MI_EUNSUPPORTED; \
    }                                                            \
    if constexpr (W <= 8 || MT <= 2)                             \
      return launch_normed<MT, RT, W, 2, EPI>(a) ? check_launch() : MI_EUNSUPPORTED; \
    return MI_EUNSUPPORTED
  switch (waves) {
    MI_NORMED_GO(4);
    MI_NORMED_GO(8);
    MI_NORMED_GO(12);
    MI_NORMED_GO(16);
    default: return MI_EUNSUPPORTED;
  }
#undef MI_NORMED_GO
}

template <int RT, int EPI>
static int normed_mt(const NormedArgs& a) {
  switch ((min(a.M, kSkinnyRows) + 15) / 16) {
    case 1: return normed_waves<1, RT, EPI>(a);
    case 2: return normed_waves<2, RT, EPI>(a);
    case 3: return normed_waves<3, RT, EPI>(a);
    default: return normed_waves<4, RT, EPI>(a);
  }
}

static int rowstat_entry(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* residual, mi_bf16* residual_out,
                         float* s_out, float* stat, int M, int N, int K, int ksplit, uint64_t* stamps, mi_stream stream) {
  if (!x || !w_packed || !residual || !residual_out || !s_out || !stat || M < 0 || N <= 0 || K <= 0) return MI_EINVAL;
  if (K % 64 || N % 16) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w_packed) || !aligned16(residual) || !aligned16(residual_out) || !aligned16(s_out))
    return MI_EINVAL;
  if (M == 0) return MI_OK;
  return rowstat(RowstatArgs{x, w_packed, residual, residual_out, s_out, stat, M, N, K, ksplit, S(stream),
                             reinterpret_cast<unsigned long long*>(stamps)});
}

static int normed_entry(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                        const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K, int epilogue, uint64_t* stamps,
                        mi_stream stream) {
  if (!s || !stat || !norm_w || !w_packed || !y || M < 0 || N <= 0 || K <= 0 || nstat <= 0) return MI_EINVAL;
  if (epilogue != 0 && epilogue != 1) return MI_EINVAL;
  if (M > kSkinnyMaxRows || K % 64 || N % 16 || (epilogue == 1 && N % 32) || nstat > 64 * kStatPerLane)
    return MI_EUNSUPPORTED;
  if (epilogue == 0 && pick_two_tiles(M, N)) return MI_EUNSUPPORTED;  // (mi_gemm_bf16_packed's two-tile form: not built)
  if (!aligned16(s) || !aligned16(w_packed) || !aligned16(y) || (reinterpret_cast<uintptr_t>(norm_w) & 7u)) return MI_EINVAL;
  if (M == 0) return MI_OK;
  const NormedArgs a{NormArgs{s, stat, norm_w, nstat, eps}, w_packed, y, M, N, K, S(stream),
                     reinterpret_cast<unsigned long long*>(stamps)};
  return epilogue == 1 ? normed_mt<2, EPI_SILU>(a) : normed_mt<1, EPI_NONE>(a);
}

}  // namespace mi

extern "C" int mi_gemm_bf16_rowstat(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* residual,
                                    mi_bf16* residual_out, float* s_out, float* stat, int M, int N, int K, int ksplit,
                                    mi_stream stream) {
  return rowstat_entry(x, w_packed, residual, residual_out, s_out, stat, M, N, K, ksplit, nullptr, stream);
}

extern "C" int mi_gemm_bf16_normed(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                                   const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K, int epilogue,
                                   mi_stream stream) {
  return normed_entry(s, stat, nstat, norm_w, eps, w_packed, y, M, N, K, epilogue, nullptr, stream);
}

// Instrumented forms (tools/chain_timeline.py): stamps[workgroups][waves][8], see gemm_chain5_kernel.hpp
extern "C" int mi_gemm_bf16_rowstat_ex(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* residual,
                                       mi_bf16* residual_out, float* s_out, float* stat, int M, int N, int K, int ksplit,
                                       uint64_t* stamps, mi_stream stream) {
  if (!stamps) return MI_EINVAL;
  return rowstat_entry(x, w_packed, residual, residual_out, s_out, stat, M, N, K, ksplit, stamps, stream);
}

extern "C" int mi_gemm_bf16_normed_ex(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                                      const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K, int epilogue,
                                      uint64_t* stamps, mi_stream stream) {
  if (!stamps) return MI_EINVAL;
  return normed_entry(s, stat, nstat, norm_w, eps, w_packed, y, M, N, K, epilogue, stamps, stream);
}

extern "C" int mi_norm_from_stat(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                                 mi_bf16* y, int rows, int cols, mi_stream stream) {
  if (!s || !stat || !norm_w || !y || rows < 0 || cols <= 0 || nstat <= 0) return MI_EINVAL;
  if (cols % 4 || nstat > 64 * kStatPerLane) return MI_EUNSUPPORTED;
  if (!aligned16(s) || (reinterpret_cast<uintptr_t>(norm_w) & 7u) || (reinterpret_cast<uintptr_t>(y) & 7u)) return MI_EINVAL;
  if (rows == 0) return MI_OK;
  hipLaunchKernelGGL(norm_from_stat_kernel, dim3(rows), dim3(256), 0, S(stream), s, stat, nstat, norm_w, y, cols, eps);
  return check_launch();
}
