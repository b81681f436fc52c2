// Weight-streaming bf16 GEMMs for the decode regime (M <= 64 rows of activations).
//
//   y[M][N] = x[M][K] @ w[N][K]^T (+ bias)        (reference: F.linear, linear.py:51,73,150)
//
// Decode linears are a pure HBM scan of the weight matrix (arithmetic intensity
// ~M flop/B), so the kernels are organised around the weight stream:
//   * a workgroup owns RT x 16 consecutive weight rows (output features); its WAVES
//     wavefronts each stream a K-slice of those rows straight into VGPRs as MFMA A
//     fragments (16 B per lane, STEPS k-steps issued before the first use - no LDS round
//     trip for data that is read exactly once);
//   * PACKED weights are stored fragment-native, [N/16][K/32][64 lanes][8 bf16]: one
//     wave-load is one contiguous, fully coalesced 1 KiB of HBM and a wave's whole K-slice
//     is one contiguous run (mi_pack_weight builds it once at load time).  The row-major
//     variant reads the same fragments as 16 rows x 64 B pieces;
//   * x is tiny and L2-resident; each wave reads its K-slice of x as MFMA B fragments
//     (x^T), so C[n][m] accumulates in fp32 on the matrix cores
//     (v_mfma_f32_16x16x32_bf16, MT = ceil(M/16) column tiles);
//   * the K-slices of a workgroup are summed through LDS in a fixed order (deterministic)
//     and rounded to bf16 once - or, for split-K over workgroups, written as fp32 partials
//     that the consumer (mi_add_rmsnorm_splitk) sums in a fixed order before that rounding.
// Epilogue EPI_SILU pairs gate row j with up row j + N/2 in one workgroup and writes
// bf16(bf16(silu(bf16 gate)) * bf16 up): SiluAndMul (activation.py:10-12) on top of
// MergedColumnParallelLinear with the reference's rounding points, minus one launch.
#include "gemm_skinny_kernel.hpp"

using namespace mi;

// (four row tiles per workgroup measured slower for the head: 70.6 vs 64.0 us at 151936 x 1024, 32 rows)

extern "C" int mi_gemm_bf16_skinny(const mi_bf16* x, const mi_bf16* w, const mi_bf16* bias, mi_bf16* y, int M,
                                   int N, int K, mi_stream stream) {
  int rc = check_gemm(x, w, y, M, N, K);
  if (rc != MI_OK) return rc;
  if (bias && !aligned16(bias)) return MI_EINVAL;
  if (M == 0) return MI_OK;
  return pick_mt<1, 0, EPI_NONE>(GemmArgs{x, w, bias, y, nullptr, M, N, K, 1, S(stream)});
}

extern "C" int mi_pack_weight(const mi_bf16* w, mi_bf16* w_packed, int N, int K, mi_stream stream) {
  if (!w || !w_packed || N <= 0 || K <= 0) return MI_EINVAL;
  if (N % 16 || K % 32) return MI_EUNSUPPORTED;
  if (!aligned16(w) || !aligned16(w_packed)) return MI_EINVAL;
  const int64_t chunks = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, S(stream), w,
                     w_packed, N, K);
  return check_launch();
}

extern "C" int mi_gemm_bf16_packed(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* bias, mi_bf16* y,
                                   int M, int N, int K, int epilogue, mi_stream stream) {
  int rc = check_gemm(x, w_packed, y, M, N, K);
  if (rc != MI_OK) return rc;
  if (bias && !aligned16(bias)) return MI_EINVAL;
  if (epilogue != 0 && epilogue != 1) return MI_EINVAL;
  if (epilogue == 1 && (bias || N % 32)) return MI_EUNSUPPORTED;
  if (M == 0) return MI_OK;
  if (epilogue == 0 && !bias && head_stream_fits(M, N, K)) {  // vocabulary-sized N: persistent workgroups
    launch_head_stream<false>(x, w_packed, y, M, N, K, PickArgs{nullptr, nullptr, nullptr, 0}, S(stream));
    return check_launch();
  }
  const GemmArgs a{x, w_packed, bias, y, nullptr, M, N, K, 1, S(stream)};
  if (epilogue == 1) return pick_mt<2, 1, EPI_SILU>(a);
  // two row tiles per workgroup halve the x traffic per weight byte once there are plenty of tiles
  if (N / 16 >= 1024 && (N / 16) % 2 == 0 && M <= 32) return pick_mt<2, 1, EPI_NONE>(a);
  return pick_mt<1, 1, EPI_NONE>(a);
}

extern "C" int mi_pack_weight_rows4(const mi_bf16* w, mi_bf16* w_packed, int N, int K, mi_stream stream) {
  if (!w || !w_packed || N <= 0 || K <= 0) return MI_EINVAL;
  if (N % 4 || K % 32) return MI_EUNSUPPORTED;
  if (!aligned16(w) || !aligned16(w_packed)) return MI_EINVAL;
  const int64_t chunks = (int64_t)N * K / 8;
  hipLaunchKernelGGL(pack_weight_rows4_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, S(stream), w,
                     w_packed, N, K);
  return check_launch();
}

extern "C" int mi_gemm_bf16_rows4(const mi_bf16* x, const mi_bf16* w_packed4, mi_bf16* y, int M, int N, int K,
                                  mi_stream stream) {
  if (!x || !w_packed4 || !y || M < 0 || N <= 0 || K <= 0) return MI_EINVAL;
  if (M > kSkinnyMaxRows || K % 32 || N % 4) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w_packed4) || !aligned16(y)) return MI_EINVAL;
  if (M == 0) return MI_OK;
  switch ((min(M, kSkinnyRows) + 15) / 16) {
    case 1: return rows4_waves<1>(x, w_packed4, y, M, N, K, S(stream));
    case 2: return rows4_waves<2>(x, w_packed4, y, M, N, K, S(stream));
    case 3: return rows4_waves<3>(x, w_packed4, y, M, N, K, S(stream));
    default: return rows4_waves<4>(x, w_packed4, y, M, N, K, S(stream));
  }
}

extern "C" int mi_gemm_bf16_packed_splitk(const mi_bf16* x, const mi_bf16* w_packed, float* partials, int M, int N,
                                          int K, int ksplit, mi_stream stream) {
  int rc = check_gemm(x, w_packed, partials, M, N, K);
  if (rc != MI_OK) return rc;
  if (ksplit < 1 || ksplit > 16 || K % (32 * ksplit)) return MI_EUNSUPPORTED;
  if (M == 0) return MI_OK;
  // (two row tiles per workgroup with twice the K slices - half the x bytes per weight byte - measured slower on the
  // decode chain: 25.0 vs 23.5 us per layer, profiles/r05_chain_ab.txt; fewer waves per CU cost more than the bytes save)
  return pick_mt<1, 1, EPI_PARTIAL>(GemmArgs{x, w_packed, nullptr, nullptr, partials, M, N, K, ksplit, S(stream)});
}

// Instrumented forms of mi_gemm_bf16_packed / mi_gemm_bf16_packed_splitk (tools/chain_timeline.py): the same launch
// with every wave's s_memrealtime stamps in stamps[workgroups][waves][8].  Only the decode chain's own configurations
// (17..32 rows, K / ksplit a multiple of 64 up to 1024: 8, 12 or 16 waves) have an instrumented kernel.
extern "C" int mi_gemm_bf16_packed_ex(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, float* partials, int M,
                                      int N, int K, int epilogue, int ksplit, uint64_t* stamps, mi_stream stream) {
  const void* out = ksplit > 0 ? static_cast<const void*>(partials) : static_cast<const void*>(y);
  int rc = check_gemm(x, w_packed, out, M, N, K);
  if (rc != MI_OK) return rc;
  if (!stamps || (epilogue != 0 && epilogue != 1) || (epilogue == 1 && ksplit > 0)) return MI_EINVAL;
  if (epilogue == 1 && N % 32) return MI_EUNSUPPORTED;
  auto* st = reinterpret_cast<unsigned long long*>(stamps);
  bool ok;
  if (ksplit > 0) {
    if (ksplit > 16 || K % (32 * ksplit)) return MI_EUNSUPPORTED;
    ok = launch_stamped<EPI_PARTIAL>(GemmArgs{x, w_packed, nullptr, nullptr, partials, M, N, K, ksplit, S(stream)}, st);
  } else if (epilogue == 1) {
    ok = launch_stamped<EPI_SILU>(GemmArgs{x, w_packed, nullptr, y, nullptr, M, N, K, 1, S(stream)}, st);
  } else {
    ok = launch_stamped<EPI_NONE>(GemmArgs{x, w_packed, nullptr, y, nullptr, M, N, K, 1, S(stream)}, st);
  }
  return ok ? check_launch() : MI_EUNSUPPORTED;
}

// ---- fp8 (e4m3) weights, bf16 activations ----------------------------------------------------------
namespace mi {
// dst[((tn * K/64 + tk) * 64 + lane) * 16 + 8 h + e] = src[(16 tn + lane % 16) * K + 64 tk + 32 h + 8 (lane / 16) + e]
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const uint8_t* __restrict__ src,
                                                              uint8_t* __restrict__ dst, int N, int K) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte chunk each
  const int64_t total = (int64_t)N * K / 16;
  if (idx >= total) return;
  const int lane = idx & 63;
  const int64_t frag = idx >> 6;
  const int ktiles = K >> 6;
  const int tk = frag % ktiles;
  const int64_t tn = frag / ktiles;
  const uint8_t* row = src + (tn * 16 + (lane & 15)) * K + tk * 64 + (lane >> 4) * 8;
  u32x4 v;
  const u32x2 lo = *reinterpret_cast<const u32x2*>(row), hi = *reinterpret_cast<const u32x2*>(row + 32);
  v[0] = lo[0];
  v[1] = lo[1];
  v[2] = hi[0];
  v[3] = hi[1];
  *reinterpret_cast<u32x4*>(dst + idx * 16) = v;
}
}  // namespace mi

extern "C" int mi_pack_weight_fp8(const uint8_t* w_q, uint8_t* w_packed, int N, int K, mi_stream stream) {
  if (!w_q || !w_packed || N <= 0 || K <= 0) return MI_EINVAL;
  if (N % 16 || K % 64) return MI_EUNSUPPORTED;
  if (!aligned16(w_q) || !aligned16(w_packed)) return MI_EINVAL;
  const int64_t chunks = (int64_t)N * K / 16;
  hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, S(stream), w_q,
                     w_packed, N, K);
  return check_launch();
}

extern "C" int mi_gemm_fp8w_packed(const mi_bf16* x, const uint8_t* w_packed, const float* scale, mi_bf16* y, int M,
                                   int N, int K, int epilogue, mi_stream stream) {
  int rc = check_gemm(x, w_packed, y, M, N, K);
  if (rc != MI_OK) return rc;
  if (!scale || !aligned16(scale)) return MI_EINVAL;
  if (K % 64) return MI_EUNSUPPORTED;
  if (epilogue != 0 && epilogue != 1) return MI_EINVAL;
  if (epilogue == 1 && N % 32) return MI_EUNSUPPORTED;
  if (M == 0) return MI_OK;
  GemmArgs a{x, reinterpret_cast<const uint16_t*>(w_packed), nullptr, y, nullptr, M, N, K, 1, S(stream)};
  a.scale = scale;
  if (epilogue == 1) return pick_mt<2, 2, EPI_SILU>(a);
  if (N / 16 >= 1024 && (N / 16) % 2 == 0 && M <= 32) return pick_mt<2, 2, EPI_NONE>(a);
  return pick_mt<1, 2, EPI_NONE>(a);
}

extern "C" int mi_gemm_fp8w_packed_splitk(const mi_bf16* x, const uint8_t* w_packed, const float* scale,
                                          float* partials, int M, int N, int K, int ksplit, mi_stream stream) {
  int rc = check_gemm(x, w_packed, partials, M, N, K);
  if (rc != MI_OK) return rc;
  if (!scale || !aligned16(scale)) return MI_EINVAL;
  if (ksplit < 1 || ksplit > 16 || K % (64 * ksplit)) return MI_EUNSUPPORTED;
  if (M == 0) return MI_OK;
  GemmArgs a{x, reinterpret_cast<const uint16_t*>(w_packed), nullptr, nullptr, partials, M, N, K, ksplit, S(stream)};
  a.scale = scale;
  return pick_mt<1, 2, EPI_PARTIAL>(a);
}
