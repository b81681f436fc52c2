// EXPERIMENT: the MLP half of a decode layer as ONE persistent launch with in-launch hand-offs
//
//   add + RMSNorm over the o_proj split-K partials  ->  gate_up GEMM + SwiGLU  ->  down GEMM (split-K partials)
//   (mi_add_rmsnorm_splitk                          ->  mi_gemm_bf16_packed(epilogue 1)  ->  mi_gemm_bf16_packed_splitk)
//
// for the bench's shapes only (hidden 1024, intermediate 3072, at most 32 rows, four K slices in and out).  256
// workgroups of 16 waves, all resident.  Every workgroup requests the weight fragments of BOTH its GEMM blocks
// before anything else (they do not depend on the activations: this is what a launch boundary cannot do), then
//   N: workgroup b < rows normalises row b                      -> xn   (write-through stores), counter `norm`
//   G: workgroup b < 192 waits for norm == rows, computes the 16 SwiGLU columns of tile pair b
//                                                                -> act  (write-through stores), counter act[b / 48]
//   D: workgroup b waits for act[b / 64] == 48, computes tile b % 64 over K slice b / 64 -> partials_out
// Hand-offs follow cdna_hip_programming.md Guideline 16: payload stored write-through (sc1) and drained by every
// storing wave, one relaxed agent-scope counter per consumer set, ONE lane polls, ONE agent-scope acquire after the
// match, then plain loads.  Spins are bounded (a give-up raises the error word instead of hanging the GPU).  The last
// workgroup to finish zeroes the counters for the next call (graph replays freeze kernel arguments).
//
// Arithmetic, summation order and rounding points are those of the three launches (same lanes, same K slices per
// wave, waves summed in order): outputs are bit-identical to them, which tests/test_kernels_gpu.py checks.
// Measured against them: tools/kbench.py KBENCH_ONLY=mlp_half, profiles/r03_mlp_half.txt.
#include "gemm_skinny_kernel.hpp"
#include "mi355_nanovllm_experiments.h"

namespace mi {

constexpr int MH_HIDDEN = 1024, MH_INTER = 3072, MH_WAVES = 16, MH_GRID = 256;
constexpr int MH_GU_BLOCKS = MH_INTER / 16;       // 192 tile pairs (gate tile b, up tile b + 192)
constexpr int MH_KS_OUT = 4;                      // K slices of the down projection
constexpr int MH_GU_PER_SLICE = MH_GU_BLOCKS / MH_KS_OUT;  // 48 gate_up blocks produce one K slice of act
constexpr int MH_DOWN_WAVES = 12;                 // 768 / 64

struct MlpHalfArgs {
  const float* parts_in;   // [4][rows][1024]
  const uint16_t* residual;
  const uint16_t* norm_w;
  const uint16_t* w_gu;    // packed [6144][1024]
  const uint16_t* w_down;  // packed [1024][3072]
  uint16_t* residual_out;
  uint16_t* xn;            // scratch [rows][1024]
  uint16_t* act;           // scratch [rows][3072]
  float* parts_out;        // [4][rows][1024]
  uint32_t* sync;          // [0] norm, [1..4] act slices, [5] done, [6] error
  int rows;
  float eps;
};

typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

__device__ __forceinline__ void store_through(uint16_t* p, u32x2 v) {  // 8 bytes, write-through to memory (sc1)
  __hip_atomic_store((gu64*)p, ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ONE lane polls; returns after the agent-scope acquire.  Bounded: ~0.2 s, then the error word is raised.
__device__ __forceinline__ void wait_count(uint32_t* counter, uint32_t target, uint32_t* err) {
  if (threadIdx.x == 0) {
    uint32_t spins = 0;
    while (__hip_atomic_load((gu32*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 22)) {
        __hip_atomic_store((gu32*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ void publish(uint32_t* counter) {  // every storing wave drains, then ONE lane counts
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MT>
__global__ __launch_bounds__(MH_WAVES * 64) void mlp_half_kernel(const MlpHalfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [16 waves][2 tiles x MT][64 lanes][4] fp32 / x slabs
  __shared__ float wave_ss[4];
  constexpr int SLOT = 2 * MT * 1024;  // bytes per wave (>= x_slab_bytes(MT) = MT * 2048)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.rows;
  uint16_t* slab = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(red) + wave * SLOT);

  // ---- the weight fragments of both GEMM blocks: requested first, used after the hand-offs ----
  u32x4 ag[2][2], ad[2];
  const bool gu_on = b < MH_GU_BLOCKS;
  const int dtile = b & 63, dslice = b >> 6;
  if (gu_on) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s)
        ag[t][s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
            a.w_gu + ((int64_t)(b + t * MH_GU_BLOCKS) * (MH_HIDDEN >> 5) + wave * 2 + s) * 512 + lane * 8));
  }
  if (wave < MH_DOWN_WAVES) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      ad[s] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
          a.w_down + ((int64_t)dtile * (MH_INTER >> 5) + dslice * 24 + wave * 2 + s) * 512 + lane * 8));
  }

  // ---- N: add + RMSNorm of row b (the arithmetic of add_rmsnorm_splitk_rows4_kernel<4, 4>, elementwise.hip) ----
  if (b < M) {
#pragma clang fp contract(off)
    float v[4], ss = 0.f;
    u32x2 wr = {0, 0};
    const int64_t off = (int64_t)b * MH_HIDDEN;
    if (tid < 256) {
      f32x4 p[4];
#pragma unroll
      for (int sp = 0; sp < 4; ++sp)
        p[sp] = *reinterpret_cast<const f32x4*>(a.parts_in + ((int64_t)sp * M + b) * MH_HIDDEN + tid * 4);
      const u32x2 rr = *reinterpret_cast<const u32x2*>(a.residual + off + tid * 4);
      wr = *reinterpret_cast<const u32x2*>(a.norm_w + tid * 4);
      f32x4 s = p[0];
#pragma unroll
      for (int sp = 1; sp < 4; ++sp) s += p[sp];
      const u32x2 raw = {pack_bf(s[0], s[1]), pack_bf(s[2], s[3])};
      u32x2 ro;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float x0 = lo_bf(raw[j]) + lo_bf(rr[j]);
        const float x1 = hi_bf(raw[j]) + hi_bf(rr[j]);
        ro[j] = pack_bf(x0, x1);
        v[2 * j] = x0;
        v[2 * j + 1] = x1;
        ss += x0 * x0;
        ss += x1 * x1;
      }
      *reinterpret_cast<u32x2*>(a.residual_out + off + tid * 4) = ro;
      ss = wave_sum(ss);
      if (lane == 0) wave_ss[wave] = ss;
    }
    __syncthreads();
    if (tid < 256) {
      float tot = wave_ss[0];
#pragma unroll
      for (int wv = 1; wv < 4; ++wv) tot += wave_ss[wv];
      const float rs = 1.0f / sqrtf(tot / (float)MH_HIDDEN + a.eps);
      u32x2 o;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float y0 = rbf(v[2 * j] * rs) * lo_bf(wr[j]);
        const float y1 = rbf(v[2 * j + 1] * rs) * hi_bf(wr[j]);
        o[j] = pack_bf(y0, y1);
      }
      store_through(a.xn + off + tid * 4, o);
    }
    publish(a.sync + 0);
  }

  // ---- G: SwiGLU columns 16 b .. 16 b + 15 (gemm_skinny_kernel<MT, 2, 16, 2, 1, EPI_SILU>) ----
  if (gu_on) {
    wait_count(a.sync + 0, (uint32_t)M, a.sync + 6);
    u32x4 stage[1][MT * 2], bfrag[MT][2];
    issue_x_lines<MT, 2>(a.xn, M, MH_HIDDEN, wave * 64, lane, stage);
    x_lines_to_frags<MT, 2>(stage, slab, lane, bfrag);
    f32x4 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[t][m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(ag[t][s]), as_frag(bfrag[m][s]), acc[t][m], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(red) + wave * SLOT + ((t * MT + m) * 64 + lane) * 16) = acc[t][m];
    __syncthreads();
    if (tid < MT * 64) {
      const int l = tid & 63, m = tid >> 6;
      auto total = [&](int tt) {
        f32x4 s = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + ((tt * MT + m) * 64 + l) * 16);
#pragma unroll
        for (int wv = 1; wv < MH_WAVES; ++wv)
          s += *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + wv * SLOT + ((tt * MT + m) * 64 + l) * 16);
        return s;
      };
      const int row = 16 * m + (l & 15);
      if (row < M) {
        const f32x4 gt = total(0), up = total(1);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float gb = rbf(gt[i]);
          const float sb = rbf(gb / (1.0f + expf(-gb)));
          o[i] = sb * rbf(up[i]);
        }
        store_through(a.act + (int64_t)row * MH_INTER + b * 16 + 4 * (l >> 4), u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])});
      }
    }
    publish(a.sync + 1 + b / MH_GU_PER_SLICE);
  }

  // ---- D: down tile dtile over K slice dslice (gemm_skinny_kernel<MT, 1, 12, 2, 1, EPI_PARTIAL>, grid (64, 4)) ----
  wait_count(a.sync + 1 + dslice, (uint32_t)MH_GU_PER_SLICE, a.sync + 6);
  if (wave < MH_DOWN_WAVES) {
    u32x4 stage[1][MT * 2], bfrag[MT][2];
    issue_x_lines<MT, 2>(a.act, M, MH_INTER, dslice * 768 + wave * 64, lane, stage);
    x_lines_to_frags<MT, 2>(stage, slab, lane, bfrag);
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(ad[s]), as_frag(bfrag[m][s]), acc[m], 0, 0, 0);
    // the product kernel's slot size for RT = 1: MT KiB per wave
#pragma unroll
    for (int m = 0; m < MT; ++m)
      *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(red) + wave * SLOT + (m * 64 + lane) * 16) = acc[m];
  }
  __syncthreads();
  if (tid < MT * 64) {
    const int l = tid & 63, m = tid >> 6;
    const int row = 16 * m + (l & 15);
    if (row < M) {
      f32x4 s = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + (m * 64 + l) * 16);
#pragma unroll
      for (int wv = 1; wv < MH_DOWN_WAVES; ++wv)
        s += *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(red) + wv * SLOT + (m * 64 + l) * 16);
      *reinterpret_cast<f32x4*>(a.parts_out + ((int64_t)dslice * M + row) * MH_HIDDEN + dtile * 16 + 4 * (l >> 4)) = s;
    }
  }

  // ---- the last workgroup to get here re-arms the counters (every poll of this call is over by then) ----
  __syncthreads();
  if (tid == 0) {
    const uint32_t n = __hip_atomic_fetch_add((gu32*)(a.sync + 5), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n == (uint32_t)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 6; ++i) __hip_atomic_store((gu32*)(a.sync + i), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace mi

using namespace mi;

extern "C" int mi_mlp_half_fused(const float* partials_in, const mi_bf16* residual, const mi_bf16* norm_w, float eps,
                                 const mi_bf16* w_gate_up_packed, const mi_bf16* w_down_packed, mi_bf16* residual_out,
                                 mi_bf16* xn_scratch, mi_bf16* act_scratch, float* partials_out, uint32_t* sync_words,
                                 int rows, int hidden, int intermediate, mi_stream stream) {
  if (!partials_in || !residual || !norm_w || !w_gate_up_packed || !w_down_packed || !residual_out || !xn_scratch ||
      !act_scratch || !partials_out || !sync_words)
    return MI_EINVAL;
  if (hidden != MH_HIDDEN || intermediate != MH_INTER || rows < 1 || rows > 32) return MI_EUNSUPPORTED;
  if (!aligned16(partials_in) || !aligned16(residual) || !aligned16(norm_w) || !aligned16(w_gate_up_packed) ||
      !aligned16(w_down_packed) || !aligned16(residual_out) || !aligned16(xn_scratch) || !aligned16(act_scratch) ||
      !aligned16(partials_out) || !aligned16(sync_words))
    return MI_EINVAL;
  const MlpHalfArgs a{partials_in, residual, norm_w, w_gate_up_packed, w_down_packed, residual_out, xn_scratch,
                      act_scratch, partials_out, sync_words, rows, eps};
  hipStream_t st = S(stream);
  if (rows <= 16)
    hipLaunchKernelGGL((mlp_half_kernel<1>), dim3(MH_GRID), dim3(MH_WAVES * 64), MH_WAVES * 2 * 1 * 1024, st, a);
  else
    hipLaunchKernelGGL((mlp_half_kernel<2>), dim3(MH_GRID), dim3(MH_WAVES * 64), MH_WAVES * 2 * 2 * 1024, st, a);
  return check_launch();
}
