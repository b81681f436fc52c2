// Paged GQA attention for gfx950 over the fragment-native KV cache.
//
// One wavefront attends one 32-token chunk (two 16-token cache tiles) at a time:
//   * 16 fully coalesced 1 KiB loads bring the chunk's K and V tiles straight
//     into VGPRs as MFMA operand fragments (the cache layout is the fragment
//     layout, include/mi355_nanovllm.h) - no LDS staging, no transposes;
//   * S^T = K . Q^T on v_mfma_f32_16x16x32_bf16 with tokens on the M axis and the
//     16 "query columns" of the wave on the N axis (decode: the G q-heads of one
//     kv head; prefill: 16/G consecutive query tokens x G heads), fp32 scores;
//   * wavefront-level online softmax in fp32: a column's 32 scores live in 8
//     registers of 4 lanes, so max/sum need two cross-lane steps;
//   * O^T += V^T . P on the same MFMA shape; the contraction index is a token
//     slot, and because any slot order is legal as long as V and P agree, the
//     score registers feed the P operand without moving between lanes.  P is
//     split into bf16 hi + lo parts (two MFMAs) so the probabilities keep ~16
//     significant bits - the matrix cores are >90 % idle in this HBM-bound
//     kernel, the extra MFMA is free and keeps the result within fp32-softmax
//     accuracy of the oracle.
// Decode splits each sequence's context over `nsplit` workgroups (flash-decoding)
// whose ranges are derived on the device from context_lens, so the launch
// geometry is static and hipGraph-capturable; a second kernel merges the splits.
#include <limits.h>

#include "mi_common.hpp"
#include "kv_store.hpp"
#include "prefill_common.hpp"

namespace mi {

// DB = head_dim / 32: 4 for the 128-wide heads every kernel here was written around, 2 for head_dim 64 (round 4:
// Llama-3.2-1B, Qwen2-0.5B - the first DB * 512 elements of the same tile formulas: a 2 KiB tile per 16 tokens)
template <int DB>
__device__ __forceinline__ void load_tile(const uint16_t* __restrict__ tile, int lane, u32x4 (&f)[DB]) {
#pragma unroll
  for (int i = 0; i < DB; ++i)  // streamed once: non-temporal (measured +10-15 % HBM read bandwidth)
    f[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tile + i * 512 + lane * 8));
}

// S^T = K . Q^T for one chunk, masked and scaled to the log2 domain.
// `limit`: tokens with index < limit are visible to this lane's column.
// `limit_all`: a wave-uniform lower bound of `limit` over the wave's columns.
template <int DB>
__device__ __forceinline__ void score_chunk(const u32x4 (&K0)[DB], const u32x4 (&K1)[DB], const bf16x8 (&Q)[DB],
                                            int tok0, int limit, int limit_all, float scale_log2e, int g,
                                            float (&p)[8]) {
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < DB; ++kk) {
    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(K0[kk]), Q[kk], s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(K1[kk]), Q[kk], s1, 0, 0, 0);
  }
  if (tok0 + 32 <= limit_all) {  // wave-uniform: the whole chunk is visible to every column
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = s0[r] * scale_log2e;
      p[4 + r] = s1[r] * scale_log2e;
    }
  } else {
    const int t0 = tok0 + 4 * g;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = (t0 + r < limit) ? s0[r] * scale_log2e : -INFINITY;
      p[4 + r] = (t0 + 16 + r < limit) ? s1[r] * scale_log2e : -INFINITY;
    }
  }
}

typedef __attribute__((ext_vector_type(4))) short s16x4;  // one MFMA 16x16x16 bf16 operand

// online softmax update + O^T += V^T . P for one chunk.
// X16 = false: one 32-deep MFMA per (dim block, P part); its A operand is assembled from halves of a
//   tile-0 and a tile-1 fragment (register moves).
// X16 = true: two 16-deep MFMAs (v_mfma_f32_16x16x16_bf16), one per tile, whose operands are aligned
//   halves of the registers as loaded - no moves, so nothing touches a fragment between its load and its
//   MFMA (the two-chunks-in-flight loop depends on that: a move right behind the load would wait for it).
template <bool X16, int DB>
__device__ __forceinline__ void accumulate_chunk(float (&p)[8], const u32x4 (&V0)[DB], const u32x4 (&V1)[DB],
                                                 float& m, float& l, f32x4 (&acc)[2 * DB]) {
  float mc = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
  mc = fmaxf(mc, __shfl_xor(mc, 16, 64));
  mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
  const float mn = fmaxf(m, mc);
  const bool dead = mn == -INFINITY;  // nothing visible for this column yet
  const float ms = dead ? 0.0f : mn;
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    p[i] = __builtin_amdgcn_exp2f(p[i] - ms);  // v_exp_f32: arguments are <= 0, tiny results may flush to 0
    ps += p[i];
  }
  if (__any(mn != m)) {  // some column's running maximum moved: rescale (rare after the first chunks)
    const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(m - mn);
    l *= alpha;
#pragma unroll
    for (int j = 0; j < 2 * DB; ++j) acc[j] *= alpha;
    m = mn;
  }
  l += ps;
  u32x4 ph, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ph[i] = pack_bf(p[2 * i], p[2 * i + 1]);
    pl[i] = pack_bf(p[2 * i] - lo_bf(ph[i]), p[2 * i + 1] - hi_bf(ph[i]));
  }
  if (X16) {
    auto half = [](const u32x4& v, int hf) { return __builtin_bit_cast(s16x4, u32x2{v[hf], v[hf + 1]}); };
#pragma unroll
    for (int part = 0; part < 4; ++part) {  // (tile 0, hi), (tile 0, lo), (tile 1, hi), (tile 1, lo)
      const s16x4 pb = half((part & 1) ? pl : ph, (part >> 1) * 2);
#pragma unroll
      for (int j = 0; j < 2 * DB; ++j)  // eight (DB = 4) independent accumulators between two uses of the same one
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(half((part >> 1) ? V1[j >> 1] : V0[j >> 1], (j & 1) * 2), pb,
                                                           acc[j], 0, 0, 0);
    }
  } else {
    const bf16x8 Ph = as_frag(ph), Pl = as_frag(pl);
#pragma unroll
    for (int j = 0; j < 2 * DB; ++j) {
      const int jp = j >> 1, hf = (j & 1) * 2;
      const u32x4 a = {V0[jp][hf], V0[jp][hf + 1], V1[jp][hf], V1[jp][hf + 1]};
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), Ph, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), Pl, acc[j], 0, 0, 0);
    }
  }
}

template <bool X16, int DB>
__device__ __forceinline__ void attend_chunk(const u32x4 (&K0)[DB], const u32x4 (&K1)[DB],
                                             const u32x4 (&V0)[DB], const u32x4 (&V1)[DB],
                                             const bf16x8 (&Q)[DB], int tok0, int limit, int limit_all,
                                             float scale_log2e, int g, float& m, float& l, f32x4 (&acc)[2 * DB]) {
  float p[8];
  score_chunk<DB>(K0, K1, Q, tok0, limit, limit_all, scale_log2e, g, p);
  accumulate_chunk<X16, DB>(p, V0, V1, m, l, acc);
}

// element strides of the KV cache: block id, kv head, 16-token tile inside a block
// (KvStrides / default_strides: prefill_common.hpp)

// ---------------------------------------------------------------------------
// decode: grid (splits, n_kv_heads, batch), WAVES wavefronts per workgroup
//
// A workgroup owns one (sequence, kv head[, split]).  The context is dealt to its waves in
// contiguous runs of 16-token cache tiles (ceil(tiles / WAVES) each, so the slowest wave is at
// most one tile - not one 32-token chunk - behind the mean); a wave walks its run two tiles per
// MFMA chunk (plus a single-tile tail) with all 16 fragment loads of a chunk issued back to back
// and keeps a running (max, sum, O).  (A software-pipelined variant that re-issued K/V loads
// mid-chunk spilled 50 VGPRs at the 128-register budget of a 16-wave workgroup and measured 1.7x
// slower.)  The waves are merged through LDS; with one split per sequence (the batched regime:
// batch * kv_heads >= ~128 workgroups) the workgroup writes the final bf16 row itself and nothing
// else is launched.  With more splits (small batches) fp32 partials go to the workspace and a tiny
// second kernel merges them.  The launch geometry is static (hipGraph); the work adapts to
// context_lens on the device.
// ---------------------------------------------------------------------------
// load one chunk = tiles (t, t2); for a single-tile tail t2 == t (a cache hit) and the caller masks
// the second half through the token limit - no branch, no zero fill
template <int DB>
__device__ __forceinline__ void load_tiles(const uint16_t* __restrict__ cache, const int32_t* __restrict__ table_row,
                                           int t, int t2, int h, KvStrides st, int tpb, int lane,
                                           u32x4 (&T0)[DB], u32x4 (&T1)[DB]) {
  const int blk0 = table_row[t / tpb];
  const int blk1 = table_row[t2 / tpb];
  load_tile<DB>(cache + (int64_t)blk0 * st.block + (int64_t)h * st.head + (int64_t)(t % tpb) * st.tile, lane, T0);
  load_tile<DB>(cache + (int64_t)blk1 * st.block + (int64_t)h * st.head + (int64_t)(t2 % tpb) * st.tile, lane, T1);
}

// The same from resolved tile offsets (elements from the cache base of this kv head): no table read, no stride
// arithmetic between the end of one chunk and the 16 loads of the next
template <int DB>
__device__ __forceinline__ void load_tiles_at(const uint16_t* __restrict__ cache_h, int64_t off0, int64_t off1, int lane,
                                              u32x4 (&T0)[DB], u32x4 (&T1)[DB]) {
  load_tile<DB>(cache_h + off0, lane, T0);
  load_tile<DB>(cache_h + off1, lane, T1);
}

// Operands of the fused step prologue (FUSE): the packed qkv row of QKVParallelLinear is consumed
// directly - q_norm / k_norm / RoPE (qwen3.py:83-88) and the scatter of the new token's K / V row
// (attention.py:32-35) happen inside the attention launch (bit-identical to mi_qknorm_rope_store,
// same helpers).
struct FusedStep {
  const uint16_t* q_w;        // [128] or nullptr (attention_bias models have no q/k norm)
  const uint16_t* k_w;
  const int64_t* positions;   // [batch]
  const float* cos_sin;       // [max_pos][128] fp32
  const int32_t* slots;       // [batch][2] = {block id, offset}
  float eps;
};

// Step prologue of the fused launch, run by ONE wavefront per workgroup.  Lane (g, n) takes head
// hh = n (+16 for a second pass when G = 16) of {the G query heads of kv head h, then its k head}, dims
// 8g + 32kk + e.  The operand loads are ISSUED before the wave's K/V tile loads (vmcnt retires in
// order: they can then be waited for without waiting for the tiles) and CONSUMED behind them, so the
// prologue's arithmetic runs under the tile loads' latency.
template <int G>
struct StepOperands {
  static constexpr int PASSES = (G + 1 + 15) / 16;
  u32x4 x[PASSES][4], w[PASSES][4];
  float4 cs[8];  // cos(8g..), cos(8g+4..), sin(..), sin(..) for the pair (kk0, kk2), then for (kk1, kk3)
  uint32_t v;    // two dims of the V row
};

template <int G>
__device__ __forceinline__ void step_prologue_issue(StepOperands<G>& o, const uint16_t* row, int n_q_heads, int n_kv,
                                                    int h, const FusedStep& fs, const float* cs, int lane) {
  const int g = lane >> 4, n = lane & 15;
#pragma unroll
  for (int ps = 0; ps < StepOperands<G>::PASSES; ++ps) {
    const int hh = min(n + 16 * ps, G);  // lanes beyond the last head shadow the k head (results unused)
    const bool is_k = hh == G;
    const uint16_t* src = row + (int64_t)(is_k ? n_q_heads + h : h * G + hh) * 128 + 8 * g;
    const uint16_t* w = is_k ? fs.k_w : fs.q_w;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      o.x[ps][kk] = *reinterpret_cast<const u32x4*>(src + 32 * kk);
      if (w != nullptr) o.w[ps][kk] = *reinterpret_cast<const u32x4*>(w + 8 * g + 32 * kk);
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float* cp = cs + 32 * p + 8 * g;
    o.cs[4 * p + 0] = *reinterpret_cast<const float4*>(cp);
    o.cs[4 * p + 1] = *reinterpret_cast<const float4*>(cp + 4);
    o.cs[4 * p + 2] = *reinterpret_cast<const float4*>(cp + 64);
    o.cs[4 * p + 3] = *reinterpret_cast<const float4*>(cp + 68);
  }
  o.v = *reinterpret_cast<const uint32_t*>(row + (int64_t)(n_q_heads + n_kv + h) * 128 + 2 * lane);
}

// q/k RMSNorm + RoPE with the rounding points (and the summation order) of head_rmsnorm / head_rope in
// kv_store.hpp; query heads -> sm_q[G][128], the new K / V row -> sm_k / sm_v and (k_tile / v_tile non-null)
// into the cache tile.
template <int G>
__device__ __forceinline__ void step_prologue_finish(const StepOperands<G>& o, bool has_norm, float eps, int lane,
                                                     uint16_t* k_tile, uint16_t* v_tile, int tok, uint16_t* sm_q,
                                                     uint16_t* sm_k, uint16_t* sm_v) {
#pragma clang fp contract(off)
  const int g = lane >> 4, n = lane & 15;
  u32x4 krow[4] = {};
#pragma unroll
  for (int ps = 0; ps < StepOperands<G>::PASSES; ++ps) {
    const int hh = n + 16 * ps;
    float x[4][8];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[kk][2 * j] = lo_bf(o.x[ps][kk][j]);
        x[kk][2 * j + 1] = hi_bf(o.x[ps][kk][j]);
      }
    if (has_norm) {  // head_rmsnorm_frag with the weights already in registers
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
      const float rs = 1.0f / sqrtf((lo + hi) / 128.0f + eps);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          x[kk][2 * j] = rbf(rbf(x[kk][2 * j] * rs) * lo_bf(o.w[ps][kk][j]));
          x[kk][2 * j + 1] = rbf(rbf(x[kk][2 * j + 1] * rs) * hi_bf(o.w[ps][kk][j]));
        }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float c[8] = {o.cs[4 * p].x, o.cs[4 * p].y, o.cs[4 * p].z, o.cs[4 * p].w,
                          o.cs[4 * p + 1].x, o.cs[4 * p + 1].y, o.cs[4 * p + 1].z, o.cs[4 * p + 1].w};
      const float sn[8] = {o.cs[4 * p + 2].x, o.cs[4 * p + 2].y, o.cs[4 * p + 2].z, o.cs[4 * p + 2].w,
                           o.cs[4 * p + 3].x, o.cs[4 * p + 3].y, o.cs[4 * p + 3].z, o.cs[4 * p + 3].w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x1 = x[p][i], x2 = x[p + 2][i];
        x[p][i] = rbf(x1 * c[i] - x2 * sn[i]);
        x[p + 2][i] = rbf(x2 * c[i] + x1 * sn[i]);
      }
    }
    if (hh <= G) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const u32x4 pk = pack16(x[kk]);
        if (hh < G) {
          *reinterpret_cast<u32x4*>(sm_q + hh * 128 + 8 * g + 32 * kk) = pk;
        } else {
          *reinterpret_cast<u32x4*>(sm_k + 8 * g + 32 * kk) = pk;
          krow[kk] = pk;
        }
      }
    }
  }
  *reinterpret_cast<uint32_t*>(sm_v + 2 * lane) = o.v;  // V row: plain copy, two dims per lane
  // the cache stores come LAST: with loads and stores pending on the same counter, every wait would have
  // to be vmcnt(0), i.e. wait for the wave's tile loads as well
  if (k_tile != nullptr && n == G % 16) {  // k_tile_off(t, 32kk + 8g) = kk*512 + (g*16 + t)*8
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<u32x4*>(k_tile + kk * 512 + (g * 16 + tok) * 8) = krow[kk];
  }
  if (v_tile != nullptr) {
    v_tile[v_tile_off(tok, 2 * lane)] = (uint16_t)(o.v & 0xffffu);
    v_tile[v_tile_off(tok, 2 * lane + 1)] = (uint16_t)(o.v >> 16);
  }
}

// Overwrite row `tok` of a loaded K / V tile with the step's token (sm_k / sm_v): the wave that attends
// the context's last tile does not wait for the prologue wave's cache store, it patches its fragments.
//   K fragment i of lane (g, n) = K[token n][32i + 8g .. +7]
//   V fragment i of lane (g, n): dwords {0,1} = V[tokens 4g .. 4g+3][32i + n], {2,3} = V[..][32i + 16 + n]
// `hit` (wave-uniform): this tile is the one that holds the token.  Callers patch BOTH tiles of a chunk
// with their own `hit` instead of branching to one of them: two alternative calls get their common code
// sunk behind a pointer phi, which forces the fragment arrays out of registers into scratch memory.
__device__ __forceinline__ void patch_new_token(u32x4 (&K)[4], u32x4 (&V)[4], const uint16_t* sm_k,
                                                const uint16_t* sm_v, int tok, int g, int n, bool hit) {
  // branch-free bit masks with static register indices (a select over the dword index would be lowered
  // to a dynamically indexed vector, i.e. through scratch memory)
  const uint32_t sh = (tok & 1) * 16, field = 0xffffu << sh;
  const bool mine = hit && g == (tok >> 2);
  const uint32_t m_even = (mine && !(tok & 2)) ? field : 0u, m_odd = (mine && (tok & 2)) ? field : 0u;
  const uint32_t m_k1 = hit && n == tok ? 0xffffffffu : 0u;
  const u32x4 m_k = {m_k1, m_k1, m_k1, m_k1}, m_v = {m_even, m_odd, m_even, m_odd};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32x4 kn = *reinterpret_cast<const u32x4*>(sm_k + 32 * i + 8 * g);
    const uint32_t va = (uint32_t)sm_v[32 * i + n] << sh, vb = (uint32_t)sm_v[32 * i + 16 + n] << sh;
    K[i] = (K[i] & ~m_k) | (kn & m_k);
    V[i] = (V[i] & ~m_v) | (u32x4{va, va, vb, vb} & m_v);
  }
}

// PIPE = false: the round-1 geometry - 16 waves x 128 VGPRs, one chunk per wave in flight (load all,
//   attend, load the next), overlap comes from the 16 waves being in different phases.
// PIPE = true: 8 waves x 256 VGPRs, TWO chunks per wave in flight: chunk i+2 is requested as soon as
//   chunk i has been consumed, so every wave keeps 16-32 KiB of loads outstanding while it computes
//   (the same 256 KiB per CU as the 16-wave form, but no wave ever sits with nothing requested), half
//   the waves to merge, and register room for the fused step prologue.
// STAMP (mi_paged_attn_decode_fused_ex, tools/attn_timeline.py): every wave records s_memrealtime (the chip-wide 100 MHz
// reference clock: s_memtime's cycle counters are not aligned between compute units) at eight points of its life into stamps[workgroup][wave][8] - where a launch's microseconds go (ramp, steady state, merge tail).  The
// instrumented instantiation is a separate kernel; the product kernels carry no stamp code.
template <int G, int WAVES, bool FUSE, bool PIPE, bool STAMP = false, int DB = 4>
__global__ __launch_bounds__(WAVES * 64) void paged_attn_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* kc,
    const uint16_t* vc, const int32_t* __restrict__ block_table, int table_stride,
    const int32_t* __restrict__ ctx_lens, float* __restrict__ part_o, float* __restrict__ part_ml,
    uint16_t* __restrict__ out, int n_q_heads, KvStrides kvs, int tpb, int tpb_shift, int resolve_run, float scale_log2e, FusedStep fs,
    unsigned long long* __restrict__ stamps = nullptr) {
  static_assert(!FUSE || PIPE, "the fused prologue needs the register room of the 8-wave form");
  static_assert(!FUSE || DB == 4, "the fused step prologue is written for 128-wide heads");
  constexpr int D = 32 * DB;  // head_dim
  unsigned long long ts[8] = {};
#define MI_STAMP(i)                                        \
  do {                                                     \
    if constexpr (STAMP) ts[i] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
  MI_STAMP(0);  // entry
  __shared__ __attribute__((aligned(16))) float sm_o[WAVES][G][D];
  __shared__ float sm_m[WAVES][16];
  __shared__ float sm_l[WAVES][16];
  // PIPE: the step's query heads, staged once per workgroup (FUSE: produced by the prologue wave)
  __shared__ __attribute__((aligned(16))) uint16_t sm_q[PIPE ? G : 1][D];
  __shared__ __attribute__((aligned(16))) uint16_t sm_k[FUSE ? 128 : 8], sm_v[FUSE ? 128 : 8];  // the step's new K / V row

  const int split = blockIdx.x, splits = gridDim.x, h = blockIdx.y, b = blockIdx.z;
  const int ctx = max(ctx_lens[b], 0);
  const int n_tiles = (ctx + 15) >> 4;
  // the context is cut into chunks of two 16-token tiles (only the very last chunk may hold one);
  // the chunks are dealt to the WAVES*splits waves as evenly as possible, each wave a contiguous run:
  // run r covers chunks [deal(r), deal(r + 1)), sizes differ by at most one, the longer runs come
  // first (so the last wave of a workgroup - which, when FUSE, also does the step prologue - never has
  // a long one)
  const int n_chunks = (n_tiles + 1) >> 1, n_waves = WAVES * splits;
  const int deal_q = n_chunks / n_waves, deal_r = n_chunks % n_waves;
  auto deal = [&](int r) { return r * deal_q + min(r, deal_r); };
  const int wg_c0 = deal(split * WAVES), wg_c1 = deal((split + 1) * WAVES);
  const int64_t row0 = (int64_t)b * n_q_heads + h * G;  // first q head of this kv head
  if (wg_c0 >= wg_c1) {  // uniform for the workgroup: nothing to attend in this split
    if (splits == 1) {     // empty context (graph padding row): the output row is zero
      for (int idx = threadIdx.x; idx < G * (D / 2); idx += WAVES * 64)
        *reinterpret_cast<uint32_t*>(out + row0 * D + 2 * idx) = 0u;
    } else {
      for (int hn = threadIdx.x; hn < G; hn += WAVES * 64) {
        part_ml[((row0 + hn) * 16 + split) * 2] = -INFINITY;
        part_ml[((row0 + hn) * 16 + split) * 2 + 1] = 0.f;
      }
    }
    return;
  }

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int32_t* table_row = block_table + (int64_t)b * table_stride;
  const int vw = split * WAVES + wave;
  const int t0 = 2 * deal(vw), t1 = min(n_tiles, 2 * deal(vw + 1));
  if constexpr (STAMP) asm volatile("" ::"s"(t0), "s"(t1));  // the context length has arrived
  MI_STAMP(1);

  // FUSE: the token of this step sits at position ctx - 1: row new_tok of tile new_tile.  The last wave of
  // the workgroup (never a long run, see deal()) produces the query heads and that token's K / V row in LDS
  // and - in split 0, which always has work - writes the row into the cache, fire and forget: the
  // wave that attends new_tile patches the row into its fragments from LDS instead of reading it back.
  const int new_tile = (ctx - 1) >> 4, new_tok = (ctx - 1) & 15;
  const bool is_pro = FUSE && wave == WAVES - 1;
  StepOperands<G> pro;
  int32_t blk = 0, off = 0;  // fetched with the other prologue operands, i.e. ahead of the tile loads
  if (is_pro) {
    blk = fs.slots[2 * b];
    off = fs.slots[2 * b + 1];
  }
  auto finish_prologue = [&]() __attribute__((always_inline)) {
    const bool store = split == 0 && blk >= 0 && off >= 0;  // split 0 always has work (the longer runs come first)
    const int64_t tile_w = (int64_t)blk * kvs.block + (int64_t)h * kvs.head + (int64_t)(off >> 4) * kvs.tile;
    step_prologue_finish<G>(pro, fs.q_w != nullptr, fs.eps, lane, store ? const_cast<uint16_t*>(kc) + tile_w : nullptr,
                            store ? const_cast<uint16_t*>(vc) + tile_w : nullptr, off & 15, &sm_q[0][0], sm_k, sm_v);
  };
  if (is_pro)
    step_prologue_issue<G>(pro, q + (int64_t)b * q_stride, n_q_heads, (int)gridDim.y, h, fs,
                           fs.cos_sin + fs.positions[b] * 128, lane);

  if (PIPE && !FUSE && wave == 0) {  // stage the G query heads (contiguous G * 2 D bytes of the q row)
    for (int c = lane; c < G * (D / 8); c += 64)
      *reinterpret_cast<u32x4*>(&sm_q[0][0] + 8 * c) =
          *reinterpret_cast<const u32x4*>(q + (int64_t)b * q_stride + (int64_t)h * G * D + 8 * c);
  }

  float m = -INFINITY, l = 0.f;
  f32x4 acc[2 * DB];
#pragma unroll
  for (int j = 0; j < 2 * DB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // a single-tile chunk loads its tile twice (a cache hit) and masks the duplicate through the limit
  auto second = [&](int t) { return t + 1 < t1 ? t + 1 : t; };
  auto limit_of = [&](int t) { return t + 1 < t1 ? ctx : min(ctx, (t + 1) * 16); };
  auto load_q = [&](bf16x8 (&Q)[DB]) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < DB; ++kk) {
      u32x4 v;
      if (PIPE) v = *reinterpret_cast<const u32x4*>(&sm_q[PIPE && n < G ? n : 0][8 * g + 32 * kk]);
      else  // n >= G reads head 0 (valid memory)
        v = *reinterpret_cast<const u32x4*>(q + (int64_t)b * q_stride + (int64_t)(h * G + (n < G ? n : 0)) * D +
                                            8 * g + 32 * kk);
      if (n >= G) v = u32x4{0, 0, 0, 0};
      Q[kk] = as_frag(v);
    }
  };

  if (PIPE) {
    // Two chunk buffers A and B.  Every path below is straight-line in its loads (no load sits under a
    // condition the following attend does not also sit under), so the compiler's vmcnt bookkeeping is
    // exact: an attend waits for ITS buffer only (s_waitcnt vmcnt(16)) while the other buffer's 16
    // fragment loads stay in flight behind it.
    u32x4 AK0[DB], AK1[DB], AV0[DB], AV1[DB], BK0[DB], BK1[DB], BV0[DB], BV1[DB];
    bf16x8 Q[DB];
    int ta = t0, tb = t0 + 2;
    // resolve_run (MI355_ATTN_RESOLVE=1, off by default - measured slower, see decode_impl): the wave's whole run of
    // block ids is read ONCE by one vector load issued ahead of everything else, lane i keeps the element offset of
    // block (b_first + i), and a chunk's two tile addresses are two pairs of v_readlane and two scalar adds instead
    // of two scalar-cache reads (s_load + lgkmcnt(0)) and 64-bit stride multiplies.  Runs of more than 64 blocks
    // take the table path in any case.
    const uint16_t* const kc_h = kc + (int64_t)h * kvs.head;
    const uint16_t* const vc_h = vc + (int64_t)h * kvs.head;
    const int b_first = tpb_shift >= 0 ? t0 >> tpb_shift : t0 / tpb;
    const int b_last = t1 > t0 ? (tpb_shift >= 0 ? (t1 - 1) >> tpb_shift : (t1 - 1) / tpb) : b_first;
    const bool resolved = resolve_run && b_last - b_first < 64;  // wave-uniform
    uint32_t run_lo = 0, run_hi = 0;
    if (resolved) {
      const int64_t o = (int64_t)table_row[min(b_first + lane, table_stride - 1)] * kvs.block;
      run_lo = (uint32_t)o;
      run_hi = (uint32_t)((uint64_t)o >> 32);
    }
    auto tile_off = [&](int t) __attribute__((always_inline)) -> int64_t {
      const int bi = tpb_shift >= 0 ? t >> tpb_shift : t / tpb;
      const int64_t in_block = tpb == 1 ? 0 : (int64_t)(tpb_shift >= 0 ? t & (tpb - 1) : t % tpb) * kvs.tile;
      if (resolved) {
        const uint32_t lo = __builtin_amdgcn_readlane(run_lo, bi - b_first), hi = __builtin_amdgcn_readlane(run_hi, bi - b_first);
        return (int64_t)(((uint64_t)hi << 32) | lo) + in_block;
      }
      return (int64_t)table_row[bi] * kvs.block + in_block;
    };
    auto load_a = [&]() __attribute__((always_inline)) {
      const int64_t o0 = tile_off(ta), o1 = tile_off(second(ta));
      load_tiles_at<DB>(kc_h, o0, o1, lane, AK0, AK1);
      load_tiles_at<DB>(vc_h, o0, o1, lane, AV0, AV1);
    };
    auto load_b = [&]() __attribute__((always_inline)) {
      const int64_t o0 = tile_off(tb), o1 = tile_off(second(tb));
      load_tiles_at<DB>(kc_h, o0, o1, lane, BK0, BK1);
      load_tiles_at<DB>(vc_h, o0, o1, lane, BV0, BV1);
    };
    auto attend_a = [&]() __attribute__((always_inline)) {
      if constexpr (FUSE) {
        if (ta == new_tile || ta + 1 == new_tile) {  // wave-uniform, true once per (sequence, kv head)
          patch_new_token(AK0, AV0, sm_k, sm_v, new_tok, g, n, ta == new_tile);
          patch_new_token(AK1, AV1, sm_k, sm_v, new_tok, g, n, ta + 1 == new_tile);
        }
      }
      attend_chunk<true, DB>(AK0, AK1, AV0, AV1, Q, ta * 16, limit_of(ta), limit_of(ta), scale_log2e, g, m, l, acc);
    };
    auto attend_b = [&]() __attribute__((always_inline)) {
      if constexpr (FUSE) {
        if (tb == new_tile || tb + 1 == new_tile) {
          patch_new_token(BK0, BV0, sm_k, sm_v, new_tok, g, n, tb == new_tile);
          patch_new_token(BK1, BV1, sm_k, sm_v, new_tok, g, n, tb + 1 == new_tile);
        }
      }
      attend_chunk<true, DB>(BK0, BK1, BV0, BV1, Q, tb * 16, limit_of(tb), limit_of(tb), scale_log2e, g, m, l, acc);
    };
    if (tb < t1) {  // at least two chunks
      load_a();
      if (is_pro) finish_prologue();  // under the latency of buffer A
      load_b();
      MI_STAMP(2);  // both buffers requested (the prologue wave: and its norm + RoPE done)
      __syncthreads();  // sm_q (sm_k, sm_v) complete; this wave's tile loads are in flight behind the barrier
      MI_STAMP(3);  // the step's q / k / v rows are published
      load_q(Q);
      if constexpr (STAMP) {  // the first chunk on its own, so that its end can be stamped
        if (tb + 4 < t1) {
          attend_a();
          MI_STAMP(4);  // first chunk consumed: the first K/V bytes have arrived and been used
          ta += 4;
          load_a();
          attend_b();
          tb += 4;
          load_b();
        }
      }
      while (tb + 4 < t1) {  // both buffers have a successor
        attend_a();
        ta += 4;
        load_a();
        attend_b();
        tb += 4;
        load_b();
      }
      attend_a();
      if (ta + 4 < t1) {  // one more chunk behind B
        ta += 4;
        load_a();
        attend_b();
        attend_a();
      } else {
        attend_b();
      }
    } else if (ta < t1) {  // a single chunk
      load_a();
      if (is_pro) finish_prologue();
      __syncthreads();
      load_q(Q);
      attend_a();
    } else {
      if (is_pro) finish_prologue();
      __syncthreads();
    }
  } else {
    int t = t0;
    u32x4 K0[DB], K1[DB], V0[DB], V1[DB];
    if (t < t1) {
      load_tiles<DB>(kc, table_row, t, second(t), h, kvs, tpb, lane, K0, K1);  // 16 fragment loads back to back
      load_tiles<DB>(vc, table_row, t, second(t), h, kvs, tpb, lane, V0, V1);
      bf16x8 Q[DB];
      load_q(Q);
      while (true) {
        attend_chunk<false, DB>(K0, K1, V0, V1, Q, t * 16, limit_of(t), limit_of(t), scale_log2e, g, m, l, acc);
        t += 2;
        if (t >= t1) break;
        load_tiles<DB>(kc, table_row, t, second(t), h, kvs, tpb, lane, K0, K1);
        load_tiles<DB>(vc, table_row, t, second(t), h, kvs, tpb, lane, V0, V1);
      }
    }
  }
  if constexpr (STAMP) asm volatile("" ::"v"(acc[0]), "v"(acc[2 * DB - 1]));
  MI_STAMP(5);  // this wave's run is attended
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  if (n < G) {
#pragma unroll
    for (int j = 0; j < 2 * DB; ++j) *reinterpret_cast<f32x4*>(&sm_o[wave][n][j * 16 + 4 * g]) = acc[j];
    if (g == 0) {
      sm_m[wave][n] = m;
      sm_l[wave][n] = l;
    }
  }
  __syncthreads();
  MI_STAMP(6);  // every wave of the workgroup has arrived
  // merge the waves (at least one of them had a chunk, so M is finite; empty ones weigh exp2(-inf) = 0)
  for (int idx = threadIdx.x; idx < G * D; idx += WAVES * 64) {
    const int hn = idx / D, d = idx % D;  // (D is a power of two: shifts)
    float M = sm_m[0][hn];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) M = fmaxf(M, sm_m[w][hn]);
    float o = 0.f, L = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
      const float e = exp2f(sm_m[w][hn] - M);
      o += e * sm_o[w][hn][d];
      L += e * sm_l[w][hn];
    }
    if (splits == 1) {
      out[(row0 + hn) * D + d] = f2bf(o / L);
    } else {
      const int64_t slot = (row0 + hn) * 16 + split;
      part_o[slot * D + d] = o;
      if (d == 0) {
        part_ml[slot * 2] = M;
        part_ml[slot * 2 + 1] = L;
      }
    }
  }
  if constexpr (STAMP) {
    MI_STAMP(7);  // merged and stored
    if (stamps != nullptr && lane == 0) {
      unsigned long long* dst = stamps + ((((int64_t)b * gridDim.y + h) * splits + split) * WAVES + wave) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i] = ts[i];
    }
  }
#undef MI_STAMP
}

// merge the splits (small batches only): one wave per (sequence, q head) row, lane = 2 dims
template <int D = 128>
__global__ __launch_bounds__(256) void paged_attn_merge_kernel(const float* __restrict__ part_o,
                                                               const float* __restrict__ part_ml,
                                                               uint16_t* __restrict__ out, int n_rows, int splits) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  float ms = -INFINITY, ls = 0.f;
  if (lane < splits) {
    const float2 ml = *reinterpret_cast<const float2*>(part_ml + (row * 16 + lane) * 2);
    ms = ml.x;
    ls = ml.y;
  }
  const float M = wave_max(ms);
  const float e = (lane < splits && ms != -INFINITY) ? exp2f(ms - M) : 0.f;
  const float den = wave_sum(e * ls);
  float2 acc = {0.f, 0.f};
  for (int s = 0; s < splits; ++s) {
    const float es = __shfl(e, s, 64);
    if (es != 0.f && 2 * lane < D) {  // (es: wave-uniform; empty splits never wrote their partial rows)
      const float2 o = *reinterpret_cast<const float2*>(part_o + (row * 16 + s) * D + 2 * lane);
      acc.x += es * o.x;
      acc.y += es * o.y;
    }
  }
  const float inv = den > 0.f ? 1.0f / den : 0.f;
  if (2 * lane < D) *reinterpret_cast<uint32_t*>(out + row * D + 2 * lane) = pack_bf(acc.x * inv, acc.y * inv);
}

// ---------------------------------------------------------------------------
// prefill: grid ceil(max_q / (4 * 32/G)) * n_kv_heads * n_seqs (XCD-aware order), 4 waves
//
// A workgroup owns 128 MFMA columns of one (sequence, kv head) - 128/G consecutive query tokens
// x G heads, 32 columns per wave - and walks the sequence's KV chunks once.  Every 32-token chunk
// (K0|K1|V0|V1 = 16 KiB of cache tiles) is fetched from HBM/L2 ONCE per workgroup by its 256
// threads (four coalesced 16-byte loads each) into a double-buffered LDS image; chunk c+1 is in
// flight in registers while chunk c is computed, one barrier per chunk.
//
// The math runs on v_mfma_f32_32x32x16_bf16 in the "swapped" form, keys on the M axis:
//   S^T[32 keys][32 cols] = K . Q^T      (8 MFMAs over the 128 dims; the A operand is a
//                                         ds_read_b128 of the cache tile as it is stored)
//   O^T[128 dims][32 cols] += V^T . P^T  (4 dim blocks x 2 key steps; the A operand is two
//                                         ds_read_b64 of the token-transposed V tile as stored)
// so a lane holds, for ITS column, 16 of the chunk's 32 scores (the other 16 sit in lane^32):
// the running max needs 15 in-lane fmax and ONE cross-lane exchange (v_permlane32_swap), the
// running sum none until the end.  The contraction index of the second product is a key slot;
// V's 8-byte pieces are picked so that slot order equals the order in which the scores already
// sit in the lane's registers - P goes from the score registers into the B operand with no lane
// movement at all.  P is split into bf16 hi + lo (two MFMAs) as in decode.
// Four-wave workgroups leave two (independent, differently phased) workgroups per CU, so one
// workgroup's softmax VALU work overlaps the other's MFMA phases.
// The finished O tile is transposed through the (now idle) LDS stage and stored as whole rows.
// ---------------------------------------------------------------------------
// (f32x16, kDeferMax, xor32_max / xor32_sum, QPrep: prefill_common.hpp)

// Probabilities in the second product.  SPLIT_P = false (the default since round 4): P is ONE bf16 per key, as in
// the reference's own CPU statement of this operator (attention_torch_native.py:80,127,139,188-189 keeps the scale, S
// and P in bf16; the fp32 running sum l is taken from the un-rounded exponentials): 16 instead of 24 MFMAs and ~50
// VALU instructions fewer per 32-key chunk in an issue-bound loop.  Bound: |out - fp32 oracle| <= 2^-7 |out| + 1e-4
// (1 bf16 ulp of the output instead of 1/2; tests/test_kernels_gpu.py).  SPLIT_P = true (tuning knob
// MI_TUNE_PREFILL_P_SPLIT, the round-1..3 form): P as bf16 hi + lo, two MFMAs per product, <= 1/2 ulp.

// one hand-issued ds_read_b64 at LDS byte address `addr` + OFF (see PV_READ2 below).  NOT counted by hipcc: the caller
// waits with its own asm s_waitcnt lgkmcnt naming the destination "+v" before the first use.
template <int OFF>
__device__ __forceinline__ void lds_read64_uncounted(uint64_t& dst, uint32_t addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}


// Variant bits VAR (mi_paged_attn_prefill_fused_ex; the product entry points use 0, or PV_SPLIT_P under the tuning knob):
//   PV_EARLY   the first two chunks are requested ahead of the Q preparation instead of behind it (stress test)
//   PV_PAIR    a ring of four buffers and ONE barrier per TWO chunks (measured slower in round 3)
//   PV_SPLIT_P P as bf16 hi + lo (see above)
//   PV_READ2   round 3's V operand reads: the two 8-byte pieces of a fragment through ordinary loads, which hipcc
//              merges into ONE ds_read2st64_b64 - an instruction served in 16-lane groups on 32 banks, where this
//              layout (16-byte lane stride) is a 2-way conflict: 16 LDS cycles per fragment, 38.6 % of all LDS cycles
//              of the kernel (profiles/r03_prefill_attention_pmc.txt: SQ_LDS_BANK_CONFLICT 8.65 M = 8 extra cycles x
//              8 instructions x 16.5 chunks x 8192 waves, to the digit).  Default now: two un-merged ds_read_b64
//              (32-lane groups on 64 banks: the same addresses are conflict-free, 2 cycles each).
//   PV_NOPREF  round 3's block-table read: s_load + s_waitcnt right in front of every chunk request.  Default now: the
//              block id of the NEXT request is read one chunk ahead, so the wait finds it landed.
//   PV_MAXTREE round 5's softmax: the chunk's maximum (a v_max3 tree over the sixteen scores, one v_permlane32_swap) BEFORE
//              the exponentials, to decide whether the reference point has to move.  Default now: the exponentials are
//              taken against the current reference point at once and the lane's sum of them - which the running sum needs
//              anyway - is the test: a sum <= 2^8 bounds every probability by 2^8, the invariant the deferred rescale
//              keeps; only when some lane's sum exceeds it (the first chunk, a maximum that ran away) is the maximum
//              computed, the state rescaled and the chunk's exponentials redone.  ~22 VALU instructions per chunk less
//              in a loop that is bound by its VALU issue (profiles/r05_prefill_attention_pmc.txt).
//   PV_VADDR   round 5's request addressing: a per-lane 64-bit pointer (five v_lshl_add_u64 + the block-table index kept
//              in a VGPR: ~13 VALU per chunk) and global_load_lds.  Default now: the tile's address stays on the scalar
//              unit - a buffer descriptor whose base is the cache block, lane * 16 as the only vector offset - and the
//              request is buffer_load_dwordx4 ... lds.
enum { PV_EARLY = 1, PV_PAIR = 2, PV_SPLIT_P = 4, PV_READ2 = 8, PV_NOPREF = 16, PV_MAXTREE = 32, PV_VADDR = 64 };

// DB = head_dim / 32 (4, or 2 for head_dim 64: plain q rows only).  G = 7 (Qwen2-0.5B, Qwen2.5-7B) runs as a group
// of 8 columns per query token whose eighth column is masked out: 4 tokens x 7 heads per wave.
template <int G, bool FUSE_Q, int VAR = 0, int DB = 4>
__global__ __launch_bounds__(256) void paged_attn_prefill_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, const int32_t* __restrict__ block_table, int table_stride,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ kv_lens, uint16_t* __restrict__ out,
    int n_q_heads, int n_kv_heads, int tpb, int tpb_shift, float scale_log2e, int n_qblocks, int n_pairs, QPrep qp) {
  constexpr bool EARLY = (VAR & PV_EARLY) != 0, PAIR = (VAR & PV_PAIR) != 0, SPLIT_P = (VAR & PV_SPLIT_P) != 0;
  constexpr bool READ2 = (VAR & PV_READ2) != 0, PREF = !(VAR & PV_NOPREF) && !PAIR;
  constexpr bool MAXTREE = (VAR & PV_MAXTREE) != 0, VADDR = (VAR & PV_VADDR) != 0;
  static_assert(DB == 4 || (!FUSE_Q && VAR == 0), "head_dim 64: the default schedule over prepared q rows");
  constexpr int D = 32 * DB, TILE = 512 * DB;  // head_dim; elements of a 16-token cache tile
  constexpr int GP = G == 7 ? 8 : G;  // columns per query token (a power of two)
  constexpr int TQ = 32 / GP;       // query tokens per wave
  constexpr int TQ_WG = 4 * TQ;     // per workgroup
  constexpr int NBUF = PAIR ? 4 : 3;  // LDS ring: chunk c is computed while c+1 and c+2 are landing
  __shared__ __attribute__((aligned(16))) uint16_t stage[NBUF][4][TILE];  // [buffer][K0,K1,V0,V1][tile]

  // Workgroups are dealt to the 8 XCDs round-robin by linear id.  All query blocks of one
  // (sequence, kv head) re-read the same K/V tiles, so they are given ids that are equal mod 8:
  // the pair's K/V (2 x 256 KiB per 1024 tokens) is then served by ONE XCD's L2 instead of being
  // pulled into all eight.  id = ((pair / 8) * n_qblocks + qblock) * 8 + pair % 8.
  const int pair = ((int)blockIdx.x / (8 * n_qblocks)) * 8 + ((int)blockIdx.x & 7);
  const int qblock = ((int)blockIdx.x >> 3) % n_qblocks;
  if (pair >= n_pairs) return;
  const int seq = pair / n_kv_heads, h = pair % n_kv_heads;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const int q_start = cu_q[seq];
  const int q_len = cu_q[seq + 1] - q_start;
  const int kv_len = kv_lens[seq];
  // heaviest (latest) query blocks are dispatched first
  const int wg_qt0 = (n_qblocks - 1 - qblock) * TQ_WG;
  if (wg_qt0 >= q_len) return;  // uniform for the workgroup
  const int shift = kv_len - q_len;
  const int wg_last_pos = shift + min(wg_qt0 + TQ_WG, q_len) - 1;
  const int wg_chunks = (wg_last_pos + 32) >> 5, wg_tiles = (wg_last_pos + 16) >> 4;

  const int qt0 = wg_qt0 + wave * TQ;
  const bool wave_on = qt0 < q_len;
  const int my_qt = qt0 + n / GP, hn = n % GP;
  const bool valid = wave_on && my_qt < q_len && hn < G;
  const int limit = valid ? shift + my_qt + 1 : 1;  // keys [0, limit) are visible to this lane's column
  const int wave_chunks = wave_on ? (shift + min(qt0 + TQ, q_len) - 1 + 32) >> 5 : 0;
  const int limit_all = wave_on && qt0 + TQ <= q_len ? shift + qt0 + 1 : 0;  // earliest column of a full block

  bf16x8 Q[2 * DB];  // B operand of S^T = K . Q^T: column n, dims 16 kk + 8 hi .. +7
  auto prepare_q = [&]() __attribute__((always_inline)) {
    const int row = valid ? my_qt : wg_qt0;  // invalid columns read a valid row and are zeroed
    const uint16_t* qrow = q + (int64_t)(q_start + row) * q_stride + (int64_t)(h * G + (hn < G ? hn : 0)) * D + 8 * hi;
    if constexpr (FUSE_Q) {
      float xq[8][8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) load16(qrow + 16 * kk, xq[kk]);
      head_rmsnorm_rope_q32(xq, qp.q_w, qp.cos_sin + qp.positions[q_start + row] * 128, hi, qp.eps);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        u32x4 v = pack16(xq[kk]);
        if (!valid) v = u32x4{0, 0, 0, 0};
        Q[kk] = as_frag(v);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2 * DB; ++kk) {
        u32x4 v = *reinterpret_cast<const u32x4*>(qrow + 16 * kk);
        if (!valid) v = u32x4{0, 0, 0, 0};
        Q[kk] = as_frag(v);
      }
    }
  };
  float m = -INFINITY, l = 0.f;  // running max (log2 domain) and this lane's share of the running sum
  f32x16 acc[DB];
#pragma unroll
  for (int j = 0; j < DB; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  // cooperative fetch by LDS-DMA (global_load_lds_dwordx4): piece = wave (K0, K1, V0, V1), four
  // 1 KiB rows of the 4 KiB cache tile per wave, each lane's 16 bytes landing at row + 16 * lane -
  // the cache tile is copied as it is stored, no registers and no ds_write involved
  const int32_t* table_row = block_table + (int64_t)seq * table_stride;
  const KvStrides st = default_strides(n_kv_heads, tpb, 16 * D);
  const int piece = wave;
  // this wave's cache (K or V) at this kv head; the tile of a chunk is then one scalar load + one 64-bit multiply-add
  // away (tiles per block are a power of two for every block size but 48, 80, ...: shifts, not divisions - the
  // request of a chunk is ~40 scalar instructions shorter, and the loop is issue-bound)
  const uint16_t* const cache_hs = ((piece & 2) ? vc : kc) + (int64_t)h * st.head;  // (wave-uniform)
  const uint16_t* const cache_h = cache_hs + lane * 8;
  const uint32_t lane16 = lane * 16;
  // the cache tile this wave fetches for chunk c (past the end: the last chunk again, into a buffer nobody reads -
  // keeps the vmcnt arithmetic uniform), and the block-table entry that holds it
  auto tile_of = [&](int c) {
    const int tile0 = 2 * min(c, wg_chunks - 1);
    return (piece & 1) ? ((tile0 + 1 < wg_tiles) ? tile0 + 1 : tile0) : tile0;
  };
  auto block_of = [&](int tile) { return table_row[tpb_shift >= 0 ? tile >> tpb_shift : tile / tpb]; };
  auto issue_from = [&](int c, int blk) {
    const int tile = tile_of(c);
    const int in_block = tpb_shift >= 0 ? tile & (tpb - 1) : tile % tpb;
    const uint16_t* src = cache_h + (int64_t)blk * st.block + (int64_t)in_block * st.tile;
    uint16_t* dst = &stage[min(c, wg_chunks - 1) % NBUF][piece][0];
#pragma unroll
    for (int i = 0; i < DB; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 512 * i),
                                       (__attribute__((address_space(3))) void*)(dst + 512 * i), 16, 0, 0);
  };
  auto issue = [&](int c) { issue_from(c, block_of(tile_of(c))); };

  // LDS element offsets of this lane's operand pieces inside a tile
  //   K (A of the first product): key n (tile n / 16, token n % 16), dims 16 kk + 8 hi .. +7
  const int k_off = (hi * 16 + (n & 15)) * 8;                  // + (kk / 2) * 512 + (kk % 2) * 256
  //   V (A of the second): dim 32 db + n, key slots {4 hi .. +3} and {8 + 4 hi .. +3} of tile s
  const int v_off = (hi * 16 + (n & 15)) * 8 + (n >> 4) * 4;   // + db * 512 (+ 256 for the second piece)

  // The loop's wait is a COUNT: vmcnt(4) = "all but the four newest vector-memory operations of this wave have
  // completed".  That means "chunk c has landed" only if the chunk DMAs are the ONLY vector-memory operations in
  // flight, in issue order.  Made true by construction: the Q operand (ordinary loads, placed by the compiler) is
  // finished and the counter drained to zero before the first chunk request; inside the loop the wave issues no
  // other vector-memory operation (block ids come through the scalar cache: lgkmcnt), and the output stores follow
  // the final vmcnt(0).  EARLY: the two requests go out first and the drain behind the Q preparation waits for
  // them as well - the same invariant at loop entry; measured slower (149-158 vs 145-150 us per launch,
  // profiles/r03_prefill_attention_order.txt).
  if (EARLY) {
    issue(0);
    issue(1);
    prepare_q();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    prepare_q();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(0);
    issue(1);
  }
  // (Scores one chunk ahead of the softmax - the eight dependent score MFMAs of chunk c + 1 in the basic block of
  // chunk c's exponentials, a fourth ring buffer - was built and measured in round 3: 145-147 us against 137-139
  // for this form at 16 x 1024 tokens; removed.)
  auto attend = [&](int c, int buf) __attribute__((always_inline)) {  // buf = c % NBUF
    if (c < wave_chunks) {
      const uint16_t* kt = &stage[buf][n >> 4][k_off];
      f32x16 s;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2 * DB; ++kk) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(kt + (kk >> 1) * 512 + (kk & 1) * 256);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), Q[kk], s, 0, 0, 0);
      }
      // register i of this lane: key c*32 + (i & 3) + 8 (i >> 2) + 4 hi
      const int tok0 = c * 32 + 4 * hi;
      if (tok0 - 4 * hi + 32 > limit_all) {  // wave-uniform: some column does not see the whole chunk
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (tok0 + (i & 3) + 8 * (i >> 2) >= limit) s[i] = -INFINITY;
      }
      float p[16];
      auto chunk_max = [&]() __attribute__((always_inline)) {
        // sixteen scores -> one maximum: nested pairs that the compiler folds into v_max3_f32 (8 instead of 15)
        float mc = fmaxf(fmaxf(fmaxf(fmaxf(s[0], s[1]), s[2]), fmaxf(fmaxf(s[3], s[4]), s[5])),
                         fmaxf(fmaxf(fmaxf(s[6], s[7]), s[8]), fmaxf(fmaxf(s[9], s[10]), s[11])));
        mc = fmaxf(mc, fmaxf(fmaxf(fmaxf(s[12], s[13]), s[14]), s[15]));
        return xor32_max(mc) * scale_log2e;  // scale > 0: the max commutes with it
      };
      auto rescale_to = [&](float mn) __attribute__((always_inline)) {
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        l *= alpha;
#pragma unroll
        for (int j = 0; j < DB; ++j)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[j][i] *= alpha;
        m = mn;
      };
      auto exps = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) p[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], scale_log2e, -m));
        // four independent partial sums instead of one 16-deep dependent add chain
        return ((p[0] + p[4]) + (p[8] + p[12])) + ((p[1] + p[5]) + (p[9] + p[13])) +
               (((p[2] + p[6]) + (p[10] + p[14])) + ((p[3] + p[7]) + (p[11] + p[15])));
      };
      // The reference point m of the exponentials only has to stay within kDeferMax (log2 units) of the true running
      // maximum: probabilities then reach at most 2^kDeferMax, which P - carried as bf16 (hi + lo), 8 (16) significant
      // bits at any magnitude - and the fp32 sums hold without loss.  On random scores some column's maximum moves in
      // most chunks (a rescale in every such chunk ran in ~85 % of them: 64 multiplies + the exp, in an issue-bound
      // loop); it moves by more than 2^8 practically once, at the first chunk.
      float lc;
      if constexpr (MAXTREE) {
        const float mc = chunk_max();
        if (__any(mc > m + kDeferMax)) rescale_to(fmaxf(m, mc));  // (finite from chunk 0 on: key 0 is visible to every column)
        lc = exps();
      } else {
        // no maximum in the common case: a lane's sixteen exponentials against the current reference point add up to at
        // most 2^kDeferMax exactly when none of them is larger - the same invariant.  (m = -inf at the first chunk: the
        // exponentials are inf or NaN, the comparison fails, the slow path sets the reference point.)
        lc = exps();
        if (__any(!(lc <= 256.0f))) {
          rescale_to(fmaxf(m, chunk_max()));
          lc = exps();
        }
      }
      l += lc;
      u32x4 ph[2], pl[2];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t hw = pack_bf(p[2 * i], p[2 * i + 1]);
        ph[i >> 2][i & 3] = hw;
        if (SPLIT_P) pl[i >> 2][i & 3] = pack_bf(p[2 * i] - lo_bf(hw), p[2 * i + 1] - hi_bf(hw));
      }
      if constexpr (READ2) {
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm) {
          const uint16_t* vt = &stage[buf][2 + sgm][v_off];
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const u32x2 a0 = *reinterpret_cast<const u32x2*>(vt + db * 512);
            const u32x2 a1 = *reinterpret_cast<const u32x2*>(vt + db * 512 + 256);
            const u32x4 a = {a0[0], a0[1], a1[0], a1[1]};
            acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(ph[sgm]), acc[db], 0, 0, 0);
            if (SPLIT_P)
              acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(pl[sgm]), acc[db], 0, 0, 0);
          }
        }
      } else {
        // The V operand by hand-issued ds_read_b64 (hipcc would pair two ordinary loads into a ds_read2st64_b64, see
        // PV_READ2; a volatile access loses the LDS address space).  hipcc does not count an asm load (guide 5.7): the
        // data is waited for by the asm s_waitcnt below, which names every destination "+v" so that no consumer can
        // be scheduled above it; LDS operations return in order, so the compiler's own lgkmcnt waits (its K reads of
        // the next chunk) only become more conservative by these.  Both key segments are requested up front - 16
        // reads in flight under the exponentials - and waited for segment by segment (lgkmcnt(8), then 0).
        uint64_t va[2][DB][2];
        const uint32_t vaddr = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)&stage[buf][2][v_off];
        // (a key segment's V tile is TILE * 2 = DB KiB behind the other's; a 32-dim block 1 KiB behind the previous)
#define MI_V_READ(SG, BLK)                                                               \
  lds_read64_uncounted<(SG) * (2 * TILE) + (BLK) * 1024>(va[SG][BLK][0], vaddr);         \
  lds_read64_uncounted<(SG) * (2 * TILE) + (BLK) * 1024 + 512>(va[SG][BLK][1], vaddr)
        MI_V_READ(0, 0); MI_V_READ(0, 1);
        if constexpr (DB == 4) { MI_V_READ(0, 2); MI_V_READ(0, 3); }
        MI_V_READ(1, 0); MI_V_READ(1, 1);
        if constexpr (DB == 4) { MI_V_READ(1, 2); MI_V_READ(1, 3); }
#undef MI_V_READ
#pragma unroll
        for (int sgm = 0; sgm < 2; ++sgm) {
          if constexpr (DB == 4) {
            if (sgm == 0)
              asm volatile("s_waitcnt lgkmcnt(8)"
                           : "+v"(va[0][0][0]), "+v"(va[0][0][1]), "+v"(va[0][1][0]), "+v"(va[0][1][1]), "+v"(va[0][2][0]),
                             "+v"(va[0][2][1]), "+v"(va[0][3][0]), "+v"(va[0][3][1]));
            else
              asm volatile("s_waitcnt lgkmcnt(0)"
                           : "+v"(va[1][0][0]), "+v"(va[1][0][1]), "+v"(va[1][1][0]), "+v"(va[1][1][1]), "+v"(va[1][2][0]),
                             "+v"(va[1][2][1]), "+v"(va[1][3][0]), "+v"(va[1][3][1]));
          } else {
            if (sgm == 0)
              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(va[0][0][0]), "+v"(va[0][0][1]), "+v"(va[0][1][0]), "+v"(va[0][1][1]));
            else
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(va[1][0][0]), "+v"(va[1][0][1]), "+v"(va[1][1][0]), "+v"(va[1][1][1]));
          }
#pragma unroll
          for (int db = 0; db < DB; ++db) {
            const u32x4 a = {(uint32_t)va[sgm][db][0], (uint32_t)(va[sgm][db][0] >> 32), (uint32_t)va[sgm][db][1],
                             (uint32_t)(va[sgm][db][1] >> 32)};
            acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(ph[sgm]), acc[db], 0, 0, 0);
            if (SPLIT_P)
              acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(a), as_frag(pl[sgm]), acc[db], 0, 0, 0);
          }
        }
      }
    }
  };
  if (PAIR) {
    for (int c = 0; c < wg_chunks; c += 2) {
      // this wave's pieces of chunks c and c + 1 (its only requests in flight) have landed; after the barrier
      // everybody's have, and everybody is done with chunks c - 2 and c - 1, whose buffers c + 2 and c + 3 re-use
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      issue(c + 2);
      issue(c + 3);
      attend(c, c % NBUF);
      if (c + 1 < wg_chunks) attend(c + 1, (c + 1) % NBUF);
    }
  } else if (!PREF) {
    for (int c = 0; c < wg_chunks; ++c) {
      // this wave's pieces of chunk c have landed (those of c+1 may still fly); after the barrier
      // everybody's have, and everybody is done reading chunk c-1, whose buffer chunk c+2 re-uses
      asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      issue(c + 2);
      attend(c, c % NBUF);
    }
  } else {
    // The same loop with the request side kept as running state - no division, no modulo, no table read in front
    // of a request (the round-3 form spends ~90 scalar instructions per chunk and wave on them, and the scalar unit
    // is shared by the CU's twelve waves).  Request r (= min(c + 2, wg_chunks - 1), i.e. saturating: past the end
    // the last chunk is requested again into its own slot, which keeps the vmcnt arithmetic uniform) fetches tile
    // rq_t = min(2 r + odd, t_max) = block-table entry rq_bi, tile rq_in of that block, into ring slot rq_slot;
    // rq_blk is the block id, read one iteration AHEAD of its use (a scalar-cache read: lgkmcnt, not vmcnt).
    const int odd = piece & 1, r_last = wg_chunks - 1;
    const int t_max = min(2 * r_last + odd, wg_tiles - 1);
    int rq = min(2, r_last), rq_slot = rq % NBUF, rq_t = min(2 * rq + odd, t_max);
    int rq_bi = tpb_shift >= 0 ? rq_t >> tpb_shift : rq_t / tpb, rq_in = rq_t - rq_bi * tpb;
    int rq_blk = table_row[rq_bi];
    int rd_slot = 0;
    for (int c = 0; c < wg_chunks; ++c) {
      // (every wave requests DB pieces of 1 KiB per chunk: "all but my newest DB" = chunk c has landed)
      if constexpr (DB == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
      if constexpr (VADDR) {
        // (a block is n_kv_heads x tpb tiles: its element stride fits 32 bits, one s_mul_i32 + s_mul_hi_i32)
        const uint16_t* src = cache_h + (int64_t)rq_blk * (int)st.block + rq_in * TILE;
        uint16_t* dst = &stage[0][piece][0] + rq_slot * (4 * TILE);
#pragma unroll
        for (int i = 0; i < DB; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 512 * i),
                                           (__attribute__((address_space(3))) void*)(dst + 512 * i), 16, 0, 0);
      } else {
        // the tile's address on the scalar unit: descriptor base = this head's tile of cache block rq_blk (64-bit scalar
        // arithmetic: a cache may be larger than 4 GiB), the lane's 16 bytes as the vector offset, the tile's DB pieces
        // of 1 KiB as SCALAR offsets (an instruction's immediate offset would move the LDS destination as well)
        const uint16_t* base = cache_hs + (int64_t)rq_blk * (int)st.block + rq_in * TILE;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base), 0, 2 * TILE, 0x00020000);
        uint16_t* dst = &stage[0][piece][0] + rq_slot * (4 * TILE);
#define MI_REQ_PIECE(I)                                                                                                \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 512 * (I)), 16, lane16,   \
                                           1024 * (I), 0, 0)
        MI_REQ_PIECE(0);
        MI_REQ_PIECE(1);
        if constexpr (DB == 4) {
          MI_REQ_PIECE(2);
          MI_REQ_PIECE(3);
        }
#undef MI_REQ_PIECE
      }
      if (rq < r_last) {  // advance the request state to chunk rq + 1 (wave-uniform)
        ++rq;
        rq_slot = rq_slot + 1 == NBUF ? 0 : rq_slot + 1;
        const int t_new = min(rq_t + 2, t_max);
        rq_in += t_new - rq_t;
        rq_t = t_new;
        if (rq_in >= tpb) {
          rq_in -= tpb;
          ++rq_bi;
        }
        if (rq_in >= tpb) {  // tpb == 1: two blocks further
          rq_in -= tpb;
          ++rq_bi;
        }
        if constexpr (!VADDR) rq_bi = __builtin_amdgcn_readfirstlane(rq_bi);  // (wave-uniform: keep the index scalar)
        rq_blk = table_row[rq_bi];  // used by the NEXT iteration: a whole chunk of compute to land in
      }
      attend(c, rd_slot);
      rd_slot = rd_slot + 1 == NBUF ? 0 : rd_slot + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two padding loads
  __syncthreads();
  l = xor32_sum(l);
  // O^T -> rows: each wave transposes its [D dims][32 cols] tile through its own D * 64 bytes of the
  // stage (all reads of the last chunk are behind the final barrier).  Column n's 2 D-byte row is
  // XOR-swizzled in 8-byte pieces so that the 32 lanes of a write hit 32 different bank pairs.
  uint16_t* tile = &stage[0][0][0] + wave * (32 * D);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  constexpr int PMASK = 8 * DB - 1;  // 8-byte pieces per row - 1
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int piece8 = db * 8 + i4 * 2 + hi;  // dims 4 piece8 .. +3  (= 32 db + 8 i4 + 4 hi)
      u32x2 o;
      o[0] = pack_bf(acc[db][4 * i4] * inv, acc[db][4 * i4 + 1] * inv);
      o[1] = pack_bf(acc[db][4 * i4 + 2] * inv, acc[db][4 * i4 + 3] * inv);
      *reinterpret_cast<u32x2*>(tile + n * D + ((piece8 ^ n) & PMASK) * 4) = o;
    }
  // same wave wrote and reads: LDS operations of a wave complete in order
  constexpr int LPR = D / 8;        // lanes that store one row of 2 D bytes (16 bytes each)
  constexpr int CPI = 64 / LPR;     // columns per iteration
#pragma unroll
  for (int it = 0; it < 32 / CPI; ++it) {
    const int col = it * CPI + lane / LPR, dl = lane % LPR;
    const int qt = qt0 + col / GP;
    const u32x2 lo = *reinterpret_cast<const u32x2*>(tile + col * D + (((2 * dl) ^ col) & PMASK) * 4);
    const u32x2 hh = *reinterpret_cast<const u32x2*>(tile + col * D + (((2 * dl + 1) ^ col) & PMASK) * 4);
    if (wave_on && qt < q_len && col % GP < G) {
      uint16_t* op = out + ((int64_t)(q_start + qt) * n_q_heads + h * G + col % GP) * D + 8 * dl;
      *reinterpret_cast<u32x4*>(op) = u32x4{lo[0], lo[1], hh[0], hh[1]};
    }
  }
}

// waves per workgroup: as many as the LDS merge buffer (WAVES * G * 512 B) and the register file allow
static int decode_waves(int G) { return G <= 4 ? 16 : (G <= 8 ? 8 : 4); }

// splits per (sequence, kv head): aim at ~16 waves per CU over the whole launch
static int decode_splits(int batch, int n_kv_heads, int waves) {
  const int base = batch * n_kv_heads * waves;
  int ns = 4096 / (base > 0 ? base : 1);
  if (ns < 1) ns = 1;
  if (ns > 16) ns = 16;
  return ns;
}

}  // namespace mi

using namespace mi;

// [batch*n_q_heads][16 splits][128] fp32 partial outputs + [..][16][2] (max, sum); only touched
// when a sequence is split over several workgroups (small batches).
extern "C" size_t mi_paged_attn_decode_workspace(int batch, int n_q_heads) {
  if (batch <= 0 || n_q_heads <= 0) return 0;
  return (size_t)batch * n_q_heads * 16 * (128 + 2) * sizeof(float);
}

static int check_attn_common(const void* q, const void* kc, const void* vc, const void* bt, int n_q_heads,
                             int n_kv_heads, int head_dim, int block_size, int64_t q_stride) {
  if (!q || !kc || !vc || !bt || n_q_heads <= 0 || n_kv_heads <= 0) return MI_EINVAL;
  if ((head_dim != 128 && head_dim != 64) || block_size <= 0 || block_size % 16 || q_stride % 8) return MI_EUNSUPPORTED;
  if (n_q_heads % n_kv_heads) return MI_EUNSUPPORTED;
  const int G = n_q_heads / n_kv_heads;
  // (7 query heads per kv head: Qwen2-0.5B / Qwen2.5-7B - the plain entry points only, not the fused step forms)
  if (G != 1 && G != 2 && G != 4 && G != 7 && G != 8 && G != 16) return MI_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(kc) || !aligned16(vc)) return MI_EINVAL;
  return MI_OK;
}

static int decode_impl(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache, const mi_bf16* v_cache,
                       const int32_t* block_table, int table_stride, const int32_t* context_lens, mi_bf16* out,
                       void* workspace, size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                       int block_size, float scale, int num_splits, KvStrides kvs, mi_stream stream,
                       const FusedStep* fused = nullptr, unsigned long long* stamps = nullptr) {
  int rc = check_attn_common(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size,
                             q_row_stride);
  if (rc != MI_OK) return rc;
  if (!context_lens || !out || !workspace || batch < 0 || table_stride <= 0) return MI_EINVAL;
  if (!aligned16(out) || !aligned16(workspace)) return MI_EINVAL;
  if (batch == 0) return MI_OK;
  if (ws_bytes < mi_paged_attn_decode_workspace(batch, n_q_heads)) return MI_EWORKSPACE;
  const int G = n_q_heads / n_kv_heads;
  if (fused != nullptr && (head_dim != MI_HEAD_DIM || G == 7)) return MI_EUNSUPPORTED;
  // geometry: 0 = 16 waves, one chunk per wave in flight; 1 = 8 waves, two chunks per wave in flight.
  // The fused entry point always uses 1; the tuning knob MI_TUNE_ATTN_PIPE (default 1) selects it for the plain one.
  const bool pipe = fused != nullptr || head_dim == 64 || G == 7 || tuning(MI_TUNE_ATTN_PIPE) != 0;
  const int waves = pipe ? 8 : decode_waves(G);
  int nsplit = num_splits > 0 ? num_splits : decode_splits(batch, n_kv_heads, pipe ? 16 : waves);
  if (nsplit > 16) nsplit = 16;
  float* part_o = static_cast<float*>(workspace);
  float* part_ml = part_o + (size_t)batch * n_q_heads * 16 * 128;
  const float sl2 = scale * 1.4426950408889634f;
  const int tpb_host = block_size / 16;
  int tpb_shift = -1;  // log2 of the tiles per block, or -1 (division) for block sizes like 48
  for (int sft = 0; sft < 12; ++sft)
    if ((1 << sft) == tpb_host) tpb_shift = sft;
  // MI_TUNE_ATTN_RESOLVE = 1: a wave resolves its whole run of block ids once (vector load + v_readlane).  Measured
  // SLOWER than reading the ids through the scalar cache per chunk (26.07 vs 25.70 us for the fused step at ctx 1100,
  // interleaved rounds, profiles/r03_kbench_attention.txt): the first tile loads then wait for a vector load instead
  // of a scalar one.  Default off.
  const int resolve_run = tuning(MI_TUNE_ATTN_RESOLVE) != 0;
  const dim3 grid(nsplit, n_kv_heads, batch);
  hipStream_t st = S(stream);
  const FusedStep fs = fused ? *fused : FusedStep{};
#define LAUNCH_DEC_AS(GG, WW, FF, PP, DD)                                                                            \
  hipLaunchKernelGGL((paged_attn_decode_kernel<GG, WW, FF, PP, false, DD>), grid, dim3(WW * 64), 0, st, q,            \
                     q_row_stride, k_cache, v_cache, block_table, table_stride, context_lens, part_o, part_ml, out, \
                     n_q_heads, kvs, block_size / 16, tpb_shift, resolve_run, sl2, fs)
#define LAUNCH_DEC(GG, WW)                              \
  if (fused) LAUNCH_DEC_AS(GG, 8, true, true, 4);       \
  else if (pipe) LAUNCH_DEC_AS(GG, 8, false, true, 4);  \
  else LAUNCH_DEC_AS(GG, WW, false, false, 4)
  if (stamps != nullptr) {  // the instrumented kernel: the bench model's geometry only
    if (!fused || G != 2) return MI_EUNSUPPORTED;
    hipLaunchKernelGGL((paged_attn_decode_kernel<2, 8, true, true, true>), grid, dim3(8 * 64), 0, st, q, q_row_stride,
                       k_cache, v_cache, block_table, table_stride, context_lens, part_o, part_ml, out, n_q_heads, kvs,
                       block_size / 16, tpb_shift, resolve_run, sl2, fs, stamps);
    rc = check_launch();
    if (rc != MI_OK || nsplit == 1) return rc;
    hipLaunchKernelGGL(paged_attn_merge_kernel<128>, dim3((batch * n_q_heads + 3) / 4), dim3(256), 0, st, part_o, part_ml,
                       out, batch * n_q_heads, nsplit);
    return check_launch();
  }
  if (head_dim == 64) {  // 2 KiB tiles (DB = 2): the 8-wave two-chunks-in-flight form for every group size
    switch (G) {
      case 1: LAUNCH_DEC_AS(1, 8, false, true, 2); break;
      case 2: LAUNCH_DEC_AS(2, 8, false, true, 2); break;
      case 4: LAUNCH_DEC_AS(4, 8, false, true, 2); break;
      case 7: LAUNCH_DEC_AS(7, 8, false, true, 2); break;
      case 8: LAUNCH_DEC_AS(8, 8, false, true, 2); break;
      default: LAUNCH_DEC_AS(16, 8, false, true, 2); break;
    }
  } else {
    switch (G) {
      case 1: LAUNCH_DEC(1, 16); break;
      case 2: LAUNCH_DEC(2, 16); break;
      case 4: LAUNCH_DEC(4, 16); break;
      case 7: LAUNCH_DEC_AS(7, 8, false, true, 4); break;
      case 8: LAUNCH_DEC(8, 8); break;
      default: LAUNCH_DEC(16, 4); break;
    }
  }
#undef LAUNCH_DEC
#undef LAUNCH_DEC_AS
  rc = check_launch();
  if (rc != MI_OK || nsplit == 1) return rc;
  const int rows = batch * n_q_heads;
  if (head_dim == 64)
    hipLaunchKernelGGL(paged_attn_merge_kernel<64>, dim3((rows + 3) / 4), dim3(256), 0, st, part_o, part_ml, out, rows,
                       nsplit);
  else
    hipLaunchKernelGGL(paged_attn_merge_kernel<128>, dim3((rows + 3) / 4), dim3(256), 0, st, part_o, part_ml, out, rows,
                       nsplit);
  return check_launch();
}

extern "C" int mi_paged_attn_decode(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                    const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                    const int32_t* context_lens, mi_bf16* out, void* workspace,
                                    size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                                    int block_size, float scale, mi_stream stream) {
  return decode_impl(q, q_row_stride, k_cache, v_cache, block_table, table_stride, context_lens, out,
                     workspace, ws_bytes, batch, n_q_heads, n_kv_heads, head_dim, block_size, scale, 0,
                     default_strides(n_kv_heads, block_size > 0 ? block_size / 16 : 1, 16 * head_dim), stream);
}

extern "C" int mi_paged_attn_decode_fused(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w,
                                          const mi_bf16* k_w, float eps, const int64_t* positions,
                                          const float* cos_sin, const int32_t* slot_2d, mi_bf16* k_cache,
                                          mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                          const int32_t* context_lens, mi_bf16* out, void* workspace,
                                          size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                                          int block_size, float scale, mi_stream stream) {
  if (!positions || !cos_sin || !slot_2d) return MI_EINVAL;
  if ((q_w == nullptr) != (k_w == nullptr)) return MI_EINVAL;
  if (!aligned16(cos_sin) || (q_w && (!aligned16(q_w) || !aligned16(k_w)))) return MI_EINVAL;
  const FusedStep fs{q_w, k_w, positions, cos_sin, slot_2d, eps};
  return decode_impl(qkv, qkv_row_stride, k_cache, v_cache, block_table, table_stride, context_lens, out,
                     workspace, ws_bytes, batch, n_q_heads, n_kv_heads, head_dim, block_size, scale, 0,
                     default_strides(n_kv_heads, block_size > 0 ? block_size / 16 : 1), stream, &fs);
}

// instrumented form of mi_paged_attn_decode_fused (tools/attn_timeline.py): stamps[batch * n_kv_heads * splits][8 waves][8]
// receives every wave's s_memrealtime (100 MHz) at entry / context length known / tile loads requested / step rows published /
// first chunk consumed / run attended / workgroup arrived / merged and stored.  Same results as the product kernel.
extern "C" int mi_paged_attn_decode_fused_ex(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w,
                                             const mi_bf16* k_w, float eps, const int64_t* positions,
                                             const float* cos_sin, const int32_t* slot_2d, mi_bf16* k_cache,
                                             mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                             const int32_t* context_lens, mi_bf16* out, void* workspace,
                                             size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                                             int block_size, float scale, uint64_t* stamps, mi_stream stream) {
  if (!positions || !cos_sin || !slot_2d || !stamps) return MI_EINVAL;
  if ((q_w == nullptr) != (k_w == nullptr)) return MI_EINVAL;
  if (!aligned16(cos_sin) || (q_w && (!aligned16(q_w) || !aligned16(k_w)))) return MI_EINVAL;
  const FusedStep fs{q_w, k_w, positions, cos_sin, slot_2d, eps};
  return decode_impl(qkv, qkv_row_stride, k_cache, v_cache, block_table, table_stride, context_lens, out,
                     workspace, ws_bytes, batch, n_q_heads, n_kv_heads, head_dim, block_size, scale, 0,
                     default_strides(n_kv_heads, block_size > 0 ? block_size / 16 : 1), stream, &fs,
                     reinterpret_cast<unsigned long long*>(stamps));
}

// tuning entry point: explicit split count and cache strides (tools/attn_exp.py)
extern "C" int mi_paged_attn_decode_ex(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                       const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                       const int32_t* context_lens, mi_bf16* out, void* workspace,
                                       size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                                       int block_size, float scale, int num_splits, int64_t stride_block,
                                       int64_t stride_head, int64_t stride_tile, mi_stream stream) {
  return decode_impl(q, q_row_stride, k_cache, v_cache, block_table, table_stride, context_lens, out,
                     workspace, ws_bytes, batch, n_q_heads, n_kv_heads, head_dim, block_size, scale, num_splits,
                     KvStrides{stride_block, stride_head, stride_tile}, stream);
}

static int prefill_impl(const mi_bf16* q, int64_t q_row_stride, const QPrep* prep, const mi_bf16* k_cache,
                        const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                        const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs, int max_seqlen_q,
                        mi_bf16* out, int n_q_heads, int n_kv_heads, int head_dim, int block_size, float scale,
                        mi_stream stream, int variant = 0) {
  if (variant < 0) return MI_EUNSUPPORTED;
  // P as one bf16 (default) or as bf16 hi + lo (variant bit / process-wide tuning knob)
  const bool split_p = (variant & PV_SPLIT_P) || tuning(MI_TUNE_PREFILL_P_SPLIT) != 0;
  int rc = check_attn_common(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size,
                             q_row_stride);
  if (rc != MI_OK) return rc;
  if (!cu_seqlens_q || !kv_lens || !out || n_seqs < 0 || max_seqlen_q < 0 || table_stride <= 0)
    return MI_EINVAL;
  if (!aligned16(out)) return MI_EINVAL;
  if (n_seqs == 0 || max_seqlen_q == 0) return MI_OK;
  const int G = n_q_heads / n_kv_heads;
  if ((int64_t)n_kv_heads * (block_size / 16) * MI_KV_TILE_ELEMS > INT32_MAX) return MI_EUNSUPPORTED;  // 32-bit block stride
  if (prep != nullptr && (head_dim != MI_HEAD_DIM || G == 7)) return MI_EUNSUPPORTED;  // fused-Q form: 128-wide, 2^n groups
  const int tq_wg = 4 * (32 / (G == 7 ? 8 : G));  // query tokens per workgroup
  const int n_qblocks = (max_seqlen_q + tq_wg - 1) / tq_wg, n_pairs = n_seqs * n_kv_heads;
  const dim3 grid((unsigned)((n_pairs + 7) / 8 * 8 * n_qblocks));
  const float sl2 = scale * 1.4426950408889634f;
  int tpb_shift = -1;
  for (int sft = 0; sft < 12; ++sft)
    if ((1 << sft) == block_size / 16) tpb_shift = sft;
  hipStream_t st = S(stream);
  const QPrep qp = prep ? *prep : QPrep{nullptr, nullptr, nullptr, 0.f};
#define LAUNCH_PRE_AS(GG, FQ, VV)                                                                                  \
  hipLaunchKernelGGL((paged_attn_prefill_kernel<GG, FQ, VV>), grid, dim3(256), 0, st, q, q_row_stride, k_cache,   \
                     v_cache, block_table, table_stride, cu_seqlens_q, kv_lens, out, n_q_heads, n_kv_heads,       \
                     block_size / 16, tpb_shift, sl2, n_qblocks, n_pairs, qp)
  // every group size: {raw q rows, q prepared in the kernel} x {P as one bf16, P as hi + lo}
#define LAUNCH_PRE(GG)                                                       \
  do {                                                                       \
    if (prep && split_p) LAUNCH_PRE_AS(GG, true, PV_SPLIT_P);                \
    else if (prep) LAUNCH_PRE_AS(GG, true, 0);                               \
    else if (split_p) LAUNCH_PRE_AS(GG, false, PV_SPLIT_P);                  \
    else LAUNCH_PRE_AS(GG, false, 0);                                        \
  } while (0)
  if (variant & ~PV_SPLIT_P) {  // schedule variants: the bench model's group size only (tools/kbench.py, the stress test)
    if (!prep || G != 2) return MI_EUNSUPPORTED;
    switch (variant) {
      case PV_EARLY: LAUNCH_PRE_AS(2, true, PV_EARLY); break;
      case PV_PAIR: LAUNCH_PRE_AS(2, true, PV_PAIR); break;
      // the round-3 forms are the round-3 kernels: with the maximum before the exponentials and per-lane request
      // pointers (PV_MAXTREE | PV_VADDR), as they were measured.  (Merged V reads + P as hi + lo on top of the
      // sum-checked softmax gave wrong, run-to-run different results with this compiler - tools/debug/
      // prefill_variant_determinism.py, variants 12 / 28 of a test build; every form offered here reproduces itself
      // bit for bit, tests/test_kernels_gpu.py::test_prefill_attention_chunk_pipeline_stress.)
      case PV_READ2: LAUNCH_PRE_AS(2, true, PV_READ2 | PV_MAXTREE | PV_VADDR); break;
      case PV_NOPREF: LAUNCH_PRE_AS(2, true, PV_NOPREF | PV_MAXTREE | PV_VADDR); break;
      case PV_READ2 | PV_NOPREF: LAUNCH_PRE_AS(2, true, PV_READ2 | PV_NOPREF | PV_MAXTREE | PV_VADDR); break;
      case PV_SPLIT_P | PV_READ2 | PV_NOPREF:  // round 3
        LAUNCH_PRE_AS(2, true, PV_SPLIT_P | PV_READ2 | PV_NOPREF | PV_MAXTREE | PV_VADDR);
        break;
      case PV_MAXTREE: LAUNCH_PRE_AS(2, true, PV_MAXTREE); break;
      case PV_VADDR: LAUNCH_PRE_AS(2, true, PV_VADDR); break;
      case PV_MAXTREE | PV_VADDR: LAUNCH_PRE_AS(2, true, PV_MAXTREE | PV_VADDR); break;  // round 5
      default: return MI_EUNSUPPORTED;
    }
    return check_launch();
  }
#define LAUNCH_PRE_Q(GG, DD)                                                                                      \
  do {                                                                                                            \
    if (split_p)                                                                                                  \
      hipLaunchKernelGGL((paged_attn_prefill_kernel<GG, false, PV_SPLIT_P * (DD == 4), DD>), grid, dim3(256), 0, st, q, \
                         q_row_stride, k_cache, v_cache, block_table, table_stride, cu_seqlens_q, kv_lens, out,   \
                         n_q_heads, n_kv_heads, block_size / 16, tpb_shift, sl2, n_qblocks, n_pairs, qp);         \
    else                                                                                                          \
      hipLaunchKernelGGL((paged_attn_prefill_kernel<GG, false, 0, DD>), grid, dim3(256), 0, st, q, q_row_stride,   \
                         k_cache, v_cache, block_table, table_stride, cu_seqlens_q, kv_lens, out, n_q_heads,      \
                         n_kv_heads, block_size / 16, tpb_shift, sl2, n_qblocks, n_pairs, qp);                    \
  } while (0)
  if (head_dim == 64) {  // 2 KiB tiles, prepared q rows; P as one bf16 (the hi + lo form exists for 128-wide heads)
    switch (G) {
      case 1: LAUNCH_PRE_Q(1, 2); break;
      case 2: LAUNCH_PRE_Q(2, 2); break;
      case 4: LAUNCH_PRE_Q(4, 2); break;
      case 7: LAUNCH_PRE_Q(7, 2); break;
      case 8: LAUNCH_PRE_Q(8, 2); break;
      default: LAUNCH_PRE_Q(16, 2); break;
    }
  } else {
    switch (G) {
      case 1: LAUNCH_PRE(1); break;
      case 2: LAUNCH_PRE(2); break;
      case 4: LAUNCH_PRE(4); break;
      case 7: LAUNCH_PRE_Q(7, 4); break;
      case 8: LAUNCH_PRE(8); break;
      default: LAUNCH_PRE(16); break;
    }
  }
#undef LAUNCH_PRE_Q
#undef LAUNCH_PRE
#undef LAUNCH_PRE_AS
  return check_launch();
}

extern "C" int mi_paged_attn_prefill(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                     const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                     const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                     int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads,
                                     int head_dim, int block_size, float scale, mi_stream stream) {
  return prefill_impl(q, q_row_stride, nullptr, k_cache, v_cache, block_table, table_stride, cu_seqlens_q, kv_lens,
                      n_seqs, max_seqlen_q, out, n_q_heads, n_kv_heads, head_dim, block_size, scale, stream);
}

extern "C" int mi_paged_attn_prefill_fused(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w, float eps,
                                           const int64_t* positions, const float* cos_sin, const mi_bf16* k_cache,
                                           const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                           const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                           int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads,
                                           int head_dim, int block_size, float scale, mi_stream stream) {
  if (!positions || !cos_sin || !aligned16(cos_sin) || (q_w && !aligned16(q_w))) return MI_EINVAL;
  const QPrep prep{q_w, positions, cos_sin, eps};
  return prefill_impl(qkv, qkv_row_stride, &prep, k_cache, v_cache, block_table, table_stride, cu_seqlens_q, kv_lens,
                      n_seqs, max_seqlen_q, out, n_q_heads, n_kv_heads, head_dim, block_size, scale, stream);
}

// tuning / stress-test form of mi_paged_attn_prefill_fused: variant 1 requests the first two K/V chunks ahead of the
// Q preparation, variant 2 synchronises the workgroup once per two chunks (tests/test_kernels_gpu.py::test_prefill_attention_chunk_pipeline_stress)
extern "C" int mi_paged_attn_prefill_fused_ex(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w, float eps,
                                              const int64_t* positions, const float* cos_sin, const mi_bf16* k_cache,
                                              const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                              const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                              int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads,
                                              int head_dim, int block_size, float scale, int variant, mi_stream stream) {
  if (!positions || !cos_sin || !aligned16(cos_sin) || (q_w && !aligned16(q_w))) return MI_EINVAL;
  const QPrep prep{q_w, positions, cos_sin, eps};
  return prefill_impl(qkv, qkv_row_stride, &prep, k_cache, v_cache, block_table, table_stride, cu_seqlens_q, kv_lens,
                      n_seqs, max_seqlen_q, out, n_q_heads, n_kv_heads, head_dim, block_size, scale, stream, variant);
}
