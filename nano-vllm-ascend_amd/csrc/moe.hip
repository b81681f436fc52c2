// Mixture-of-experts block of the decode path (reference: Qwen3MoeSparseMoeBlock.forward,
// nanovllm/models/qwen3_moe.py:150-185) as five launches over expert-sorted (token, expert) pairs:
//
//   mi_moe_route     router logits -> fp32 softmax, top-k, renormalise, bf16 weights   (:153-161)
//   mi_moe_sort      pairs grouped by expert: offsets, the token and the (token, slot) index of every sorted pair
//   mi_moe_gate_up   per expert: act = SwiGLU(x[tokens] @ W_gate_up[e]^T)   (expert MLP, :118-121)
//   mi_moe_down      per expert: y = act @ W_down[e]^T, rounded to bf16 (the expert's output), written to the
//                    pair's own row t * top_k + slot: y has the SAME layout on every tensor-parallel rank
//                    whatever order the sort's atomics produced, so it can be summed over the ranks
//   mi_moe_combine   out[t] = sum over the token's experts in ascending expert id of bf16(y * w), every
//                    partial sum rounded to bf16 (`index_add_` on a bf16 tensor, :171-184)
//
// The reference loops over the experts in Python and runs three library GEMMs per expert.  Here the two
// grouped GEMMs are ONE launch each: a workgroup owns (expert, 16 output features) - for gate_up the gate
// tile and its up tile - its waves split K and keep their weight fragments (fragment-native layout of
// mi_pack_weight, one slab per expert) in registers while they walk the expert's pairs 16 at a time: in
// decode (a few pairs per expert) every expert's weights are streamed exactly once, straight into VGPRs.
// The token gather happens in the B-fragment loads (row index = pair_token[...]).
// Under tensor parallelism the experts are sharded along their intermediate dimension as the reference
// does (:104-115); the bf16 partial outputs of mi_moe_down are summed over the ranks before mi_moe_combine.
#include "mi_common.hpp"

namespace mi {

// ---------------------------------------------------------------------------------------------------
// routing: one wavefront per token, E <= 512 experts (8 per lane)
// ---------------------------------------------------------------------------------------------------
template <int EPL>  // experts per lane
__global__ __launch_bounds__(256) void moe_route_kernel(const uint16_t* __restrict__ logits, int T, int E, int top_k,
                                                        int32_t* __restrict__ topk_ids, uint16_t* __restrict__ topk_w) {
  const int lane = threadIdx.x & 63;
  const int t = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (t >= T) return;
  float p[EPL];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    const int e = lane + 64 * i;
    p[i] = e < E ? bf2f(logits[(int64_t)t * E + e]) : -INFINITY;
    mx = fmaxf(mx, p[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    p[i] = lane + 64 * i < E ? expf(p[i] - mx) : 0.f;
    sum += p[i];
  }
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < EPL; ++i) {
    p[i] = p[i] / sum;  // softmax(dim=1, dtype=float)
    // non-finite router logits (a NaN upstream): the picks below must still be k DISTINCT valid experts - an id left
    // unwritten or out of range would send mi_moe_sort's counters and the grouped GEMMs out of bounds.  The weights
    // of such a token are NaN (0 / 0), as the reference's would be; nothing changes for finite logits.
    if (!(p[i] == p[i])) p[i] = 0.f;
  }
  // top-k by repeated arg-max; equal probabilities: the lower expert id first
  float sel_p = 0.f;  // lane j < top_k keeps the j-th pick
  int sel_e = 0;
  float total = 0.f;
  for (int j = 0; j < top_k; ++j) {
    float best = -1.f;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < EPL; ++i) {
      const int e = lane + 64 * i;
      if (e < E && (p[i] > best || (p[i] == best && e < best_e))) {
        best = p[i];
        best_e = e;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oe = __shfl_xor(best_e, o, 64);
      if (ob > best || (ob == best && oe < best_e)) {
        best = ob;
        best_e = oe;
      }
    }
    if (lane == j) {
      sel_p = best;
      sel_e = best_e;
    }
    total += best;  // routing_weights.sum(dim=-1): picks in descending order, sequential fp32 adds
#pragma unroll
    for (int i = 0; i < EPL; ++i)
      if (lane + 64 * i == best_e) p[i] = -2.f;  // taken
  }
  // the k picks, ascending by expert id (the order mi_moe_combine adds them in): rank = number of smaller ids
  int rank = 0;
  for (int j = 0; j < top_k; ++j) {
    const int oe = __shfl(sel_e, j, 64);
    rank += (lane < top_k && oe < sel_e) ? 1 : 0;
  }
  if (lane < top_k) {
    topk_ids[(int64_t)t * top_k + rank] = sel_e;
    topk_w[(int64_t)t * top_k + rank] = f2bf(sel_p / total);
  }
}

// ---------------------------------------------------------------------------------------------------
// grouping: one workgroup; pairs (t, slot) -> position in the expert-sorted list
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void moe_sort_kernel(const int32_t* __restrict__ topk_ids, int n_pairs, int top_k, int E,
                                                        int32_t* __restrict__ offsets, int32_t* __restrict__ pair_token,
                                                        int32_t* __restrict__ pair_index) {
  __shared__ int cnt[513], cursor[512];
  for (int e = threadIdx.x; e <= E; e += blockDim.x) cnt[e] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_pairs; i += blockDim.x) atomicAdd(&cnt[topk_ids[i]], 1);
  __syncthreads();
  if (threadIdx.x == 0) {  // exclusive scan (E <= 512)
    int run = 0;
    for (int e = 0; e < E; ++e) {
      const int c = cnt[e];
      offsets[e] = run;
      cursor[e] = run;
      run += c;
    }
    offsets[E] = run;
  }
  __syncthreads();
  // which slot inside an expert's run a pair gets is irrelevant to every result (the pairs of a run are
  // independent MFMA columns, and results are written back by pair index), so the order the atomics happen
  // to produce - different from rank to rank and run to run - is fine
  for (int i = threadIdx.x; i < n_pairs; i += blockDim.x) {
    const int pos = atomicAdd(&cursor[topk_ids[i]], 1);
    pair_token[pos] = i / top_k;
    pair_index[pos] = i;
  }
}

// ---------------------------------------------------------------------------------------------------
// grouped GEMMs.  GATE_UP: y = SwiGLU(x[token] @ [W_gate | W_up][e]^T) with the rounding points of
// mi_gemm_bf16_packed(epilogue 1); else y = bf16(x_pairs @ W[e]^T).
// grid (N / 16 tiles (GATE_UP: gate tiles), E); WAVES waves split K = WAVES * 32 * STEPS.
// ---------------------------------------------------------------------------------------------------
template <int WAVES, int STEPS, bool GATE_UP>
__global__ __launch_bounds__(WAVES * 64) void moe_gemm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ w,
                                                              const int32_t* __restrict__ offsets,
                                                              const int32_t* __restrict__ pair_token,
                                                              uint16_t* __restrict__ y, int N, int K) {
  // GATE_UP: pair_token = token of each sorted pair (input row); result rows in sorted order.
  // down:    pair_token = (token, slot) index of each sorted pair (OUTPUT row); input rows in sorted order.
  constexpr int RT = GATE_UP ? 2 : 1;
  __shared__ __attribute__((aligned(16))) float red[WAVES][RT][64][4];
  const int e = blockIdx.y;
  const int p0 = offsets[e], n_e = offsets[e + 1] - p0;
  if (n_e <= 0) return;  // uniform for the workgroup: nobody chose this expert
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, r = lane & 15;
  const int ktiles = K >> 5, kbeg = wave * 32 * STEPS;
  const int n_rows = GATE_UP ? 2 * N : N;  // weight rows per expert
  const uint16_t* we = w + (int64_t)e * n_rows * K;
  u32x4 a[RT][STEPS];
#pragma unroll
  for (int t = 0; t < RT; ++t) {
    const int tile = (int)blockIdx.x + t * (N >> 4);
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      a[t][s] = __builtin_nontemporal_load(
          reinterpret_cast<const u32x4*>(we + ((int64_t)tile * ktiles + (kbeg >> 5) + s) * 512 + lane * 8));
  }
  for (int m0 = 0; m0 < n_e; m0 += 16) {
    const int pr = p0 + min(m0 + r, n_e - 1);  // pairs past the end shadow the last one (results unused)
    const int64_t row = GATE_UP ? (int64_t)pair_token[pr] : (int64_t)pr;
    const int64_t out_row = GATE_UP ? (int64_t)(p0 + m0 + r) : (int64_t)pair_token[pr];
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 b[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) b[s] = *reinterpret_cast<const u32x4*>(x + row * K + kbeg + 32 * s + 8 * g);
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
#pragma unroll
      for (int t = 0; t < RT; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a[t][s]), as_frag(b[s]), acc[t], 0, 0, 0);
    if (m0 > 0) __syncthreads();  // the previous pass has read red[]
#pragma unroll
    for (int t = 0; t < RT; ++t) *reinterpret_cast<f32x4*>(&red[wave][t][lane][0]) = acc[t];
    __syncthreads();
    if (threadIdx.x < 64) {  // C fragment: lane (g, c) = features 16 tile + 4g .. +3 of pair m0 + c
      f32x4 s0 = *reinterpret_cast<const f32x4*>(&red[0][0][lane][0]);
      f32x4 s1 = GATE_UP ? *reinterpret_cast<const f32x4*>(&red[0][RT - 1][lane][0]) : s0;
#pragma unroll
      for (int wv = 1; wv < WAVES; ++wv) {
        s0 += *reinterpret_cast<const f32x4*>(&red[wv][0][lane][0]);
        if (GATE_UP) s1 += *reinterpret_cast<const f32x4*>(&red[wv][RT - 1][lane][0]);
      }
      if (m0 + r < n_e) {
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (GATE_UP) {
            const float gb = rbf(s0[i]);  // the gate_up GEMM output, rounded to bf16 as the unfused path
            o[i] = rbf(gb / (1.0f + expf(-gb))) * rbf(s1[i]);
          } else {
            o[i] = s0[i];
          }
        }
        *reinterpret_cast<u32x2*>(y + out_row * N + (int)blockIdx.x * 16 + 4 * g) =
            u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
      }
    }
  }
}

// out[t] = (((0 + c_0) + c_1) + ...) with c_j = bf16(y[pair j of t] * w_j), every sum rounded to bf16
__global__ __launch_bounds__(256) void moe_combine_kernel(const uint16_t* __restrict__ y,
                                                          const uint16_t* __restrict__ topk_w, uint16_t* __restrict__ out,
                                                          int T, int top_k, int H) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int vpr = H >> 3;
  if (idx >= (int64_t)T * vpr) return;
  const int t = idx / vpr, c = (idx % vpr) * 8;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int j = 0; j < top_k; ++j) {
    const float wj = bf2f(topk_w[(int64_t)t * top_k + j]);
    const u32x4 v = *reinterpret_cast<const u32x4*>(y + ((int64_t)t * top_k + j) * H + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] = rbf(acc[2 * i] + rbf(lo_bf(v[i]) * wj));
      acc[2 * i + 1] = rbf(acc[2 * i + 1] + rbf(hi_bf(v[i]) * wj));
    }
  }
  *reinterpret_cast<u32x4*>(out + (int64_t)t * H + c) =
      u32x4{pack_bf(acc[0], acc[1]), pack_bf(acc[2], acc[3]), pack_bf(acc[4], acc[5]), pack_bf(acc[6], acc[7])};
}

// contraction lengths launch_moe_gemm has an instantiation for (K = WAVES * 32 * STEPS, see there)
static bool moe_k_supported(int K) {
  if (K <= 0) return false;
  if (K % 64 == 0 && K / 64 <= 16)
    switch (K / 64) {
      case 1: case 2: case 3: case 4: case 6: case 8: case 12: case 16: return true;
      default: break;
    }
  if (K % 128 == 0 && K / 128 <= 16)
    switch (K / 128) {
      case 5: case 10: case 12: case 16: return true;
      default: break;
    }
  if (K % 256 == 0 && K / 256 <= 16)
    switch (K / 256) {
      case 10: case 16: return true;
      default: break;
    }
  return false;
}

template <bool GATE_UP>
static int launch_moe_gemm(const uint16_t* x, const uint16_t* w, const int32_t* offsets, const int32_t* pair_token,
                           uint16_t* y, int N, int K, int E, hipStream_t st) {
  const dim3 grid(N / 16, E);
#define MOE_GO(W, S)                                                                                              \
  do {                                                                                                            \
    hipLaunchKernelGGL((moe_gemm_kernel<W, S, GATE_UP>), grid, dim3(W * 64), 0, st, x, w, offsets, pair_token, y, \
                       N, K);                                                                                     \
    return check_launch();                                                                                        \
  } while (0)
  // K = WAVES * 32 * STEPS: as many waves as K allows at two k-steps each, deeper slices beyond 16 waves
  if (K % 64 == 0 && K / 64 <= 16) {
    switch (K / 64) {
      case 1: MOE_GO(1, 2);
      case 2: MOE_GO(2, 2);
      case 3: MOE_GO(3, 2);
      case 4: MOE_GO(4, 2);
      case 6: MOE_GO(6, 2);
      case 8: MOE_GO(8, 2);
      case 12: MOE_GO(12, 2);
      case 16: MOE_GO(16, 2);
      default: break;
    }
  }
  if (K % 128 == 0 && K / 128 <= 16) {
    switch (K / 128) {
      case 5: MOE_GO(5, 4);
      case 10: MOE_GO(10, 4);
      case 12: MOE_GO(12, 4);
      case 16: MOE_GO(16, 4);
      default: break;
    }
  }
  if (K % 256 == 0 && K / 256 <= 16) {
    switch (K / 256) {
      case 10: MOE_GO(10, 8);
      case 16: MOE_GO(16, 8);
      case 20: break;
      default: break;
    }
  }
#undef MOE_GO
  return MI_EUNSUPPORTED;
}

}  // namespace mi

// MI_OK if mi_moe_gate_up / mi_moe_down take an expert of [2 * inter][hidden] / [hidden][inter] (inter = this rank's
// share of moe_intermediate_size): asked before the model is built, so that an unsupported width is a message at
// start-up and not MI_EUNSUPPORTED inside warm-up or graph capture
extern "C" int mi_moe_shapes_supported(int hidden, int inter) {
  if (hidden <= 0 || inter <= 0) return MI_EINVAL;
  if (hidden % 64 || inter % 64 || hidden % 16) return MI_EUNSUPPORTED;
  return mi::moe_k_supported(hidden) && mi::moe_k_supported(inter) ? MI_OK : MI_EUNSUPPORTED;
}

using namespace mi;

extern "C" int mi_moe_route(const mi_bf16* router_logits, int n_tokens, int n_experts, int top_k, int32_t* topk_ids,
                            mi_bf16* topk_w, mi_stream stream) {
  if (!router_logits || !topk_ids || !topk_w || n_tokens < 0) return MI_EINVAL;
  if (n_experts < 1 || n_experts > 512 || top_k < 1 || top_k > n_experts || top_k > 64) return MI_EUNSUPPORTED;
  if (n_tokens == 0) return MI_OK;
  const dim3 grid((n_tokens + 3) / 4), block(256);
  hipStream_t st = S(stream);
  if (n_experts <= 64) hipLaunchKernelGGL((moe_route_kernel<1>), grid, block, 0, st, router_logits, n_tokens, n_experts, top_k, topk_ids, topk_w);
  else if (n_experts <= 128) hipLaunchKernelGGL((moe_route_kernel<2>), grid, block, 0, st, router_logits, n_tokens, n_experts, top_k, topk_ids, topk_w);
  else if (n_experts <= 256) hipLaunchKernelGGL((moe_route_kernel<4>), grid, block, 0, st, router_logits, n_tokens, n_experts, top_k, topk_ids, topk_w);
  else hipLaunchKernelGGL((moe_route_kernel<8>), grid, block, 0, st, router_logits, n_tokens, n_experts, top_k, topk_ids, topk_w);
  return check_launch();
}

extern "C" int mi_moe_sort(const int32_t* topk_ids, int n_tokens, int top_k, int n_experts, int32_t* expert_offsets,
                           int32_t* pair_token, int32_t* pair_index, mi_stream stream) {
  if (!topk_ids || !expert_offsets || !pair_token || !pair_index || n_tokens < 0) return MI_EINVAL;
  if (n_experts < 1 || n_experts > 512 || top_k < 1) return MI_EUNSUPPORTED;
  hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), 0, S(stream), topk_ids, n_tokens * top_k, top_k, n_experts,
                     expert_offsets, pair_token, pair_index);
  return check_launch();
}

extern "C" int mi_moe_gate_up(const mi_bf16* x, const mi_bf16* w_packed, const int32_t* expert_offsets,
                              const int32_t* pair_token, mi_bf16* act, int n_experts, int hidden, int inter,
                              mi_stream stream) {
  if (!x || !w_packed || !expert_offsets || !pair_token || !act) return MI_EINVAL;
  if (n_experts < 1 || hidden % 64 || inter % 16) return MI_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w_packed) || !aligned16(act)) return MI_EINVAL;
  return launch_moe_gemm<true>(x, w_packed, expert_offsets, pair_token, act, inter, hidden, n_experts, S(stream));
}

extern "C" int mi_moe_down(const mi_bf16* act, const mi_bf16* w_packed, const int32_t* expert_offsets,
                           const int32_t* pair_index, mi_bf16* y, int n_experts, int hidden, int inter,
                           mi_stream stream) {
  if (!act || !w_packed || !expert_offsets || !pair_index || !y) return MI_EINVAL;
  if (n_experts < 1 || inter % 64 || hidden % 16) return MI_EUNSUPPORTED;
  if (!aligned16(act) || !aligned16(w_packed) || !aligned16(y)) return MI_EINVAL;
  return launch_moe_gemm<false>(act, w_packed, expert_offsets, pair_index, y, hidden, inter, n_experts, S(stream));
}

extern "C" int mi_moe_combine(const mi_bf16* y, const mi_bf16* topk_w, mi_bf16* out, int n_tokens, int top_k,
                              int hidden, mi_stream stream) {
  if (!y || !topk_w || !out || n_tokens < 0 || top_k < 1) return MI_EINVAL;
  if (hidden % 8) return MI_EUNSUPPORTED;
  if (!aligned16(y) || !aligned16(out)) return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  const int64_t n = (int64_t)n_tokens * (hidden / 8);
  hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), y, topk_w, out,
                     n_tokens, top_k, hidden);
  return check_launch();
}
