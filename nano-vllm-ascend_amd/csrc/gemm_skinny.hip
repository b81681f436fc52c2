// Weight-streaming bf16 GEMM for the decode regime (M <= 64 rows of activations).
//
//   y[M][N] = x[M][K] @ w[N][K]^T (+ bias)        (reference: F.linear, linear.py:51,73,150)
//
// Decode linears are a pure HBM scan of the weight matrix (arithmetic intensity
// ~M flop/B), so the kernel is organised around the weight stream:
//   * one workgroup owns 16 consecutive weight rows (output features); its
//     WAVES wavefronts each stream a K/WAVES slice of those rows straight into
//     VGPRs as MFMA A fragments (16 B per lane, issued 8 deep before the first
//     use - no LDS round trip for data that is read exactly once);
//   * x is tiny and L2-resident; each wave reads its K slice of x as MFMA B
//     fragments (x^T), so C[n][m] accumulates in fp32 on the matrix cores
//     (v_mfma_f32_16x16x32_bf16, MT = ceil(M/16) column tiles);
//   * the K-slices are summed through LDS in a fixed order (deterministic) and
//     rounded to bf16 once.
#include "mi_common.hpp"

namespace mi {

template <int MT, int WAVES, int STEPS, bool BIAS>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ w, const uint16_t* __restrict__ bias,
    uint16_t* __restrict__ y, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [WAVES][MT][256]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int n0 = blockIdx.x * 16;
  const int kslice = K / WAVES;  // multiple of 32 * STEPS (checked on the host)
  const int kbeg = wave * kslice;

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A fragment for k-step s: lane (g, r) <- w[n0 + r][k + 32 s + 8 g .. +8]
  const uint16_t* wp = w + (int64_t)(n0 + r) * K + kbeg + 8 * g;
  // B fragment of column tile m: lane (g, c) <- x[16 m + c][k + 32 s + 8 g .. +8].  Rows >= M are
  // clamped to row M-1: MFMA output columns are independent, the duplicates are never stored.
  const uint16_t* xp[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) xp[m] = x + (int64_t)min(16 * m + r, M - 1) * K + kbeg + 8 * g;

  for (int k = 0; k < kslice; k += 32 * STEPS) {
    u32x4 a[STEPS], bfrag[MT][STEPS];
    // every load of the block is issued before the first MFMA: no branches, no waits in between
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      a[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + k + 32 * s));
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s = 0; s < STEPS; ++s) bfrag[m][s] = *reinterpret_cast<const u32x4*>(xp[m] + k + 32 * s);
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[s]), as_frag(bfrag[m][s]), acc[m], 0, 0, 0);
  }

  // C fragment: lane (g, c) holds y[m-tile col c][n0 + 4 g + i], i = 0..3
#pragma unroll
  for (int m = 0; m < MT; ++m)
    *reinterpret_cast<f32x4*>(red + ((wave * MT + m) * 64 + lane) * 4) = acc[m];
  __syncthreads();
  // each (m-tile, lane) result is finished by one thread, summing K-slices in wave order
  for (int item = threadIdx.x; item < MT * 64; item += WAVES * 64) {
    const int m = item >> 6, l = item & 63;
    f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * MT + m) * 64 + l) * 4);
#pragma unroll
    for (int wv = 1; wv < WAVES; ++wv) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(red + ((wv * MT + m) * 64 + l) * 4);
      s += t;
    }
    const int row = 16 * m + (l & 15);
    const int col = n0 + 4 * (l >> 4);
    if (row < M) {
      if (BIAS) {
        const u32x2 bw = *reinterpret_cast<const u32x2*>(bias + col);
        s[0] += lo_bf(bw[0]);
        s[1] += hi_bf(bw[0]);
        s[2] += lo_bf(bw[1]);
        s[3] += hi_bf(bw[1]);
      }
      u32x2 o;
      o[0] = pack_bf(s[0], s[1]);
      o[1] = pack_bf(s[2], s[3]);
      *reinterpret_cast<u32x2*>(y + (int64_t)row * N + col) = o;
    }
  }
}

template <int MT, int WAVES, int STEPS>
static void launch(const uint16_t* x, const uint16_t* w, const uint16_t* bias, uint16_t* y, int M, int N,
                   int K, hipStream_t st) {
  const size_t lds = (size_t)WAVES * MT * 256 * sizeof(float);
  if (bias)
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, WAVES, STEPS, true>), dim3(N / 16), dim3(WAVES * 64), lds, st,
                       x, w, bias, y, M, N, K);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<MT, WAVES, STEPS, false>), dim3(N / 16), dim3(WAVES * 64), lds, st,
                       x, w, bias, y, M, N, K);
}

// STEPS k-steps (32 * STEPS elements of K) are in flight per wave and loop iteration.
template <int MT, int WAVES>
static bool try_waves(const uint16_t* x, const uint16_t* w, const uint16_t* bias, uint16_t* y, int M, int N,
                      int K, int want_steps, hipStream_t st) {
  if (K % WAVES) return false;
  const int kslice = K / WAVES;
  constexpr int MAXS = MT <= 2 ? 8 : 4;  // register budget: (MT + 1) * STEPS fragments
  if (MAXS >= 8 && want_steps >= 8 && kslice % 256 == 0) return launch<MT, WAVES, 8>(x, w, bias, y, M, N, K, st), true;
  if (want_steps >= 4 && kslice % 128 == 0) return launch<MT, WAVES, 4>(x, w, bias, y, M, N, K, st), true;
  if (want_steps >= 2 && kslice % 64 == 0) return launch<MT, WAVES, 2>(x, w, bias, y, M, N, K, st), true;
  if (kslice % 32 == 0) return launch<MT, WAVES, 1>(x, w, bias, y, M, N, K, st), true;
  return false;
}

template <int MT>
static int pick_waves(const uint16_t* x, const uint16_t* w, const uint16_t* bias, uint16_t* y, int M, int N,
                      int K, hipStream_t st) {
  // Few row tiles (N/16 < ~2 per CU): spread K over many waves so every CU slot holds loads;
  // many row tiles (lm_head): fewer, fatter waves.
  const int tiles = N / 16;
  bool ok = false;
  if (tiles >= 2048) {
    ok = try_waves<MT, 4>(x, w, bias, y, M, N, K, 8, st) || try_waves<MT, 2>(x, w, bias, y, M, N, K, 8, st) ||
         try_waves<MT, 1>(x, w, bias, y, M, N, K, 8, st);
  } else if (K >= 2048) {
    ok = (K % (16 * 128) == 0 && try_waves<MT, 16>(x, w, bias, y, M, N, K, 8, st)) ||
         (K % (12 * 128) == 0 && try_waves<MT, 12>(x, w, bias, y, M, N, K, 8, st)) ||
         try_waves<MT, 8>(x, w, bias, y, M, N, K, 8, st) || try_waves<MT, 4>(x, w, bias, y, M, N, K, 8, st);
  } else {
    ok = (K % (8 * 128) == 0 && try_waves<MT, 8>(x, w, bias, y, M, N, K, 8, st)) ||
         (K % (4 * 128) == 0 && try_waves<MT, 4>(x, w, bias, y, M, N, K, 8, st)) ||
         try_waves<MT, 2>(x, w, bias, y, M, N, K, 8, st);
  }
  if (!ok) ok = try_waves<MT, 1>(x, w, bias, y, M, N, K, 8, st);
  if (!ok) return MI_EUNSUPPORTED;
  return check_launch();
}

}  // namespace mi

using namespace mi;

extern "C" int mi_gemm_bf16_skinny(const mi_bf16* x, const mi_bf16* w, const mi_bf16* bias, mi_bf16* y, int M,
                                   int N, int K, mi_stream stream) {
  if (!x || !w || !y || M < 0 || N <= 0 || K <= 0) return MI_EINVAL;
  if (M > 64 || K % 32 || N % 16) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (bias && !aligned16(bias))) return MI_EINVAL;
  if (M == 0) return MI_OK;
  hipStream_t st = S(stream);
  switch ((M + 15) / 16) {
    case 1: return pick_waves<1>(x, w, bias, y, M, N, K, st);
    case 2: return pick_waves<2>(x, w, bias, y, M, N, K, st);
    case 3: return pick_waves<3>(x, w, bias, y, M, N, K, st);
    default: return pick_waves<4>(x, w, bias, y, M, N, K, st);
  }
}
