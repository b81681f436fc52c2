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
#include <stdlib.h>

#include "mi_common.hpp"

namespace mi {

__device__ __forceinline__ void load_tile(const uint16_t* __restrict__ tile, int lane, u32x4 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)  // streamed once: non-temporal (measured +10-15 % HBM read bandwidth)
    f[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(tile + i * 512 + lane * 8));
}

// S^T = K . Q^T for one chunk, masked and scaled to the log2 domain.
// `limit`: tokens with index < limit are visible to this lane's column.
// `limit_all`: a wave-uniform lower bound of `limit` over the wave's columns.
__device__ __forceinline__ void score_chunk(const u32x4 (&K0)[4], const u32x4 (&K1)[4], const bf16x8 (&Q)[4],
                                            int tok0, int limit, int limit_all, float scale_log2e, int g,
                                            float (&p)[8]) {
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
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

// online softmax update + O^T += V^T . P for one chunk
__device__ __forceinline__ void accumulate_chunk(float (&p)[8], const u32x4 (&V0)[4], const u32x4 (&V1)[4],
                                                 float& m, float& l, f32x4 (&acc)[8]) {
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
    for (int j = 0; j < 8; ++j) acc[j] *= alpha;
    m = mn;
  }
  l += ps;
  u32x4 ph, pl;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ph[i] = pack_bf(p[2 * i], p[2 * i + 1]);
    pl[i] = pack_bf(p[2 * i] - lo_bf(ph[i]), p[2 * i + 1] - hi_bf(ph[i]));
  }
  const bf16x8 Ph = as_frag(ph), Pl = as_frag(pl);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int jp = j >> 1, hf = (j & 1) * 2;
    const u32x4 a = {V0[jp][hf], V0[jp][hf + 1], V1[jp][hf], V1[jp][hf + 1]};
    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), Ph, acc[j], 0, 0, 0);
    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(a), Pl, acc[j], 0, 0, 0);
  }
}

__device__ __forceinline__ void attend_chunk(const u32x4 (&K0)[4], const u32x4 (&K1)[4],
                                             const u32x4 (&V0)[4], const u32x4 (&V1)[4],
                                             const bf16x8 (&Q)[4], int tok0, int limit, int limit_all,
                                             float scale_log2e, int g, float& m, float& l, f32x4 (&acc)[8]) {
  float p[8];
  score_chunk(K0, K1, Q, tok0, limit, limit_all, scale_log2e, g, p);
  accumulate_chunk(p, V0, V1, m, l, acc);
}

// element strides of the KV cache: block id, kv head, 16-token tile inside a block
struct KvStrides {
  int64_t block, head, tile;
};
__host__ __device__ inline KvStrides default_strides(int n_kv_heads, int tpb) {
  return KvStrides{(int64_t)n_kv_heads * tpb * MI_KV_TILE_ELEMS, (int64_t)tpb * MI_KV_TILE_ELEMS, MI_KV_TILE_ELEMS};
}

// Issue the 16 fragment loads of chunk `c` (tiles 2c, 2c+1).  `c` and the table row are
// wave-uniform, so the two block ids come from scalar loads and all vector loads are issued
// back to back with no wait in between.  A second tile past the end of the context is
// redirected to the (valid) first tile; its scores are masked by `limit`, so its bytes never
// reach the output (p == 0 exactly) and no branch or zero-fill is needed.
__device__ __forceinline__ void load_chunk(const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
                                           const int32_t* __restrict__ table_row, int c, int n_tiles, int h,
                                           KvStrides st, int tpb, int lane, u32x4 (&K0)[4], u32x4 (&K1)[4],
                                           u32x4 (&V0)[4], u32x4 (&V1)[4]) {
  const int tile0 = 2 * c;
  const int tile1 = (2 * c + 1 < n_tiles) ? 2 * c + 1 : tile0;
  const int blk0 = table_row[tile0 / tpb];
  const int blk1 = table_row[tile1 / tpb];
  const int64_t base0 = (int64_t)blk0 * st.block + (int64_t)h * st.head + (int64_t)(tile0 % tpb) * st.tile;
  const int64_t base1 = (int64_t)blk1 * st.block + (int64_t)h * st.head + (int64_t)(tile1 % tpb) * st.tile;
  load_tile(kc + base0, lane, K0);
  load_tile(kc + base1, lane, K1);
  load_tile(vc + base0, lane, V0);
  load_tile(vc + base1, lane, V1);
}

// one cache (K or V) of chunk `c`: 8 fragment loads
__device__ __forceinline__ void load_chunk_k(const uint16_t* __restrict__ cache, const int32_t* __restrict__ table_row,
                                             int c, int n_tiles, int h, KvStrides st, int tpb, int lane,
                                             u32x4 (&T0)[4], u32x4 (&T1)[4]) {
  const int tile0 = 2 * c;
  const int tile1 = (2 * c + 1 < n_tiles) ? 2 * c + 1 : tile0;
  const int blk0 = table_row[tile0 / tpb];
  const int blk1 = table_row[tile1 / tpb];
  load_tile(cache + (int64_t)blk0 * st.block + (int64_t)h * st.head + (int64_t)(tile0 % tpb) * st.tile, lane, T0);
  load_tile(cache + (int64_t)blk1 * st.block + (int64_t)h * st.head + (int64_t)(tile1 % tpb) * st.tile, lane, T1);
}

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
__device__ __forceinline__ void load_tiles(const uint16_t* __restrict__ cache, const int32_t* __restrict__ table_row,
                                           int t, int t2, int h, KvStrides st, int tpb, int lane,
                                           u32x4 (&T0)[4], u32x4 (&T1)[4]) {
  const int blk0 = table_row[t / tpb];
  const int blk1 = table_row[t2 / tpb];
  load_tile(cache + (int64_t)blk0 * st.block + (int64_t)h * st.head + (int64_t)(t % tpb) * st.tile, lane, T0);
  load_tile(cache + (int64_t)blk1 * st.block + (int64_t)h * st.head + (int64_t)(t2 % tpb) * st.tile, lane, T1);
}

template <int G, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void paged_attn_decode_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, const int32_t* __restrict__ block_table, int table_stride,
    const int32_t* __restrict__ ctx_lens, float* __restrict__ part_o, float* __restrict__ part_ml,
    uint16_t* __restrict__ out, int n_q_heads, KvStrides kvs, int tpb, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) float sm_o[WAVES][G][128];
  __shared__ float sm_m[WAVES][16];
  __shared__ float sm_l[WAVES][16];

  const int split = blockIdx.x, splits = gridDim.x, h = blockIdx.y, b = blockIdx.z;
  const int ctx = max(ctx_lens[b], 0);
  const int n_tiles = (ctx + 15) >> 4;
  // the context is cut into chunks of two 16-token tiles (only the very last chunk may hold one);
  // the chunks are dealt to the WAVES*splits waves as evenly as possible, each wave a contiguous run
  const int n_chunks = (n_tiles + 1) >> 1, n_waves = WAVES * splits;
  const int wg_c0 = split * WAVES * n_chunks / n_waves, wg_c1 = (split + 1) * WAVES * n_chunks / n_waves;
  const int64_t row0 = (int64_t)b * n_q_heads + h * G;  // first q head of this kv head
  if (wg_c0 >= wg_c1) {  // uniform for the workgroup: nothing to attend in this split
    if (splits == 1) {     // empty context (graph padding row): the output row is zero
      for (int idx = threadIdx.x; idx < G * 64; idx += WAVES * 64)
        *reinterpret_cast<uint32_t*>(out + row0 * 128 + 2 * idx) = 0u;
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
  const int t0 = 2 * (vw * n_chunks / n_waves), t1 = min(n_tiles, 2 * ((vw + 1) * n_chunks / n_waves));

  float m = -INFINITY, l = 0.f;
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool has_work = t0 < t1;
  int t = t0;
  u32x4 K0[4], K1[4], V0[4], V1[4];
  if (has_work) {
    const int t2 = t + 1 < t1 ? t + 1 : t;
    load_tiles(kc, table_row, t, t2, h, kvs, tpb, lane, K0, K1);  // 16 fragment loads back to back
    load_tiles(vc, table_row, t, t2, h, kvs, tpb, lane, V0, V1);
  }
  if (has_work) {
    bf16x8 Q[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 v = *reinterpret_cast<const u32x4*>(q + (int64_t)b * q_stride + (int64_t)(h * G + (n < G ? n : 0)) * 128 +
                                                8 * g + 32 * kk);  // n >= G reads head 0 (valid memory)
      if (n >= G) v = u32x4{0, 0, 0, 0};
      Q[kk] = as_frag(v);
    }
    while (true) {
      // a single-tile chunk masks its (duplicate) second half through the token limit
      const int lim = t + 1 < t1 ? ctx : min(ctx, (t + 1) * 16);
      attend_chunk(K0, K1, V0, V1, Q, t * 16, lim, lim, scale_log2e, g, m, l, acc);
      t += 2;
      if (t >= t1) break;
      const int t2 = t + 1 < t1 ? t + 1 : t;
      load_tiles(kc, table_row, t, t2, h, kvs, tpb, lane, K0, K1);
      load_tiles(vc, table_row, t, t2, h, kvs, tpb, lane, V0, V1);
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  if (n < G) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(&sm_o[wave][n][j * 16 + 4 * g]) = acc[j];
    if (g == 0) {
      sm_m[wave][n] = m;
      sm_l[wave][n] = l;
    }
  }
  __syncthreads();
  // merge the waves (at least one of them had a chunk, so M is finite; empty ones weigh exp2(-inf) = 0)
  for (int idx = threadIdx.x; idx < G * 128; idx += WAVES * 64) {
    const int hn = idx >> 7, d = idx & 127;
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
      out[(row0 + hn) * 128 + d] = f2bf(o / L);
    } else {
      const int64_t slot = (row0 + hn) * 16 + split;
      part_o[slot * 128 + d] = o;
      if (d == 0) {
        part_ml[slot * 2] = M;
        part_ml[slot * 2 + 1] = L;
      }
    }
  }
}

// merge the splits (small batches only): one wave per (sequence, q head) row, lane = 2 dims
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
    if (es != 0.f) {  // wave-uniform; empty splits never wrote their partial rows
      const float2 o = *reinterpret_cast<const float2*>(part_o + (row * 16 + s) * 128 + 2 * lane);
      acc.x += es * o.x;
      acc.y += es * o.y;
    }
  }
  const float inv = den > 0.f ? 1.0f / den : 0.f;
  *reinterpret_cast<uint32_t*>(out + row * 128 + 2 * lane) = pack_bf(acc.x * inv, acc.y * inv);
}

// ---------------------------------------------------------------------------
// prefill: grid (ceil(max_q / (8 * 16/G)), n_kv_heads, n_seqs), 8 waves
//
// A workgroup owns 8 consecutive query blocks of one (sequence, kv head) - 128 MFMA columns, i.e.
// 128/G query tokens x G heads - and walks the sequence's KV chunks once: every 32-token chunk
// (K0|K1|V0|V1 = 16 KiB, already in MFMA-fragment order in the paged cache) is fetched from
// HBM/L2 ONCE per workgroup by all 512 threads (two coalesced 16-byte loads each) into a
// double-buffered LDS image and read back by each wave as lane-linear ds_read_b128 fragments
// (conflict-free).  Compared with one wave fetching its own chunk copies this cuts L2 traffic 8x and
// makes the loop MFMA-bound.  Chunk c+1 is in flight in registers while chunk c is computed; one
// barrier per chunk.  Waves whose query block lies beyond the chunk (causal) skip the math only.
// ---------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(512) void paged_attn_prefill_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, const int32_t* __restrict__ block_table, int table_stride,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ kv_lens, uint16_t* __restrict__ out,
    int n_q_heads, int n_kv_heads, int tpb, float scale_log2e) {
  constexpr int TQ = 16 / G;        // query tokens per wave
  constexpr int TQ_WG = 8 * TQ;     // per workgroup
  __shared__ __attribute__((aligned(16))) uint16_t stage[2][4][2048];  // [buffer][K0,K1,V0,V1][tile]

  const int seq = blockIdx.z, h = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int q_start = cu_q[seq];
  const int q_len = cu_q[seq + 1] - q_start;
  const int kv_len = kv_lens[seq];
  // heaviest (latest) query blocks are dispatched first
  const int wg_qt0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * TQ_WG;
  if (wg_qt0 >= q_len) return;  // uniform for the workgroup
  const int shift = kv_len - q_len;
  const int wg_last_pos = shift + min(wg_qt0 + TQ_WG, q_len) - 1;
  const int wg_chunks = (wg_last_pos + 32) >> 5, wg_tiles = (wg_last_pos + 16) >> 4;

  const int qt0 = wg_qt0 + wave * TQ;
  const bool wave_on = qt0 < q_len;
  const int my_qt = qt0 + n / G, hn = n % G;
  const bool valid = wave_on && my_qt < q_len;
  const int limit = valid ? shift + my_qt + 1 : 1;
  const int wave_chunks = wave_on ? (shift + min(qt0 + TQ, q_len) - 1 + 32) >> 5 : 0;
  const int limit_all = wave_on && qt0 + TQ <= q_len ? shift + qt0 + 1 : 0;  // earliest column of a full block

  bf16x8 Q[4];
  {
    const int row = valid ? my_qt : wg_qt0;  // invalid columns read a valid row and are zeroed
    const uint16_t* qp = q + (int64_t)(q_start + row) * q_stride + (int64_t)(h * G + hn) * 128 + 8 * g;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + 32 * kk);
      if (!valid) v = u32x4{0, 0, 0, 0};
      Q[kk] = as_frag(v);
    }
  }
  float m = -INFINITY, l = 0.f;
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // cooperative fetch: thread t moves bytes [16 t', 16 t'+16) and +2 KiB of piece t / 128
  const int32_t* table_row = block_table + (int64_t)seq * table_stride;
  const KvStrides st = default_strides(n_kv_heads, tpb);
  const int piece = threadIdx.x >> 7;            // 0 K0, 1 K1, 2 V0, 3 V1
  const int within = (threadIdx.x & 127) * 8;    // element offset of this thread's first 16 bytes
  auto fetch = [&](int c, u32x4& r0, u32x4& r1) {
    const int tile0 = 2 * c;
    const int tile = (piece & 1) ? ((tile0 + 1 < wg_tiles) ? tile0 + 1 : tile0) : tile0;
    const int blk = table_row[tile / tpb];
    const uint16_t* src = ((piece & 2) ? vc : kc) + (int64_t)blk * st.block + (int64_t)h * st.head +
                          (int64_t)(tile % tpb) * st.tile + within;
    r0 = *reinterpret_cast<const u32x4*>(src);
    r1 = *reinterpret_cast<const u32x4*>(src + 1024);
  };
  auto stash = [&](int buf, const u32x4& r0, const u32x4& r1) {
    *reinterpret_cast<u32x4*>(&stage[buf][piece][within]) = r0;
    *reinterpret_cast<u32x4*>(&stage[buf][piece][within + 1024]) = r1;
  };

  u32x4 r0, r1;
  fetch(0, r0, r1);
  stash(0, r0, r1);
  __syncthreads();
  for (int c = 0; c < wg_chunks; ++c) {
    const bool more = c + 1 < wg_chunks;
    if (more) fetch(c + 1, r0, r1);  // in flight during this chunk's math
    if (c < wave_chunks) {
      const int buf = c & 1;
      u32x4 K0[4], K1[4], V0[4], V1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        K0[i] = *reinterpret_cast<const u32x4*>(&stage[buf][0][i * 512 + lane * 8]);
        K1[i] = *reinterpret_cast<const u32x4*>(&stage[buf][1][i * 512 + lane * 8]);
        V0[i] = *reinterpret_cast<const u32x4*>(&stage[buf][2][i * 512 + lane * 8]);
        V1[i] = *reinterpret_cast<const u32x4*>(&stage[buf][3][i * 512 + lane * 8]);
      }
      attend_chunk(K0, K1, V0, V1, Q, c * 32, limit, limit_all, scale_log2e, g, m, l, acc);
    }
    if (more) stash((c + 1) & 1, r0, r1);
    __syncthreads();
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (!valid) return;
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  uint16_t* op = out + ((int64_t)(q_start + my_qt) * n_q_heads + h * G + hn) * 128 + 4 * g;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32x2 o;
    o[0] = pack_bf(acc[j][0] * inv, acc[j][1] * inv);
    o[1] = pack_bf(acc[j][2] * inv, acc[j][3] * inv);
    *reinterpret_cast<u32x2*>(op + 16 * j) = o;
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
  if (head_dim != MI_HEAD_DIM || block_size <= 0 || block_size % 16 || q_stride % 8) return MI_EUNSUPPORTED;
  if (n_q_heads % n_kv_heads) return MI_EUNSUPPORTED;
  const int G = n_q_heads / n_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16) return MI_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(kc) || !aligned16(vc)) return MI_EINVAL;
  return MI_OK;
}

static int decode_impl(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache, const mi_bf16* v_cache,
                       const int32_t* block_table, int table_stride, const int32_t* context_lens, mi_bf16* out,
                       void* workspace, size_t ws_bytes, int batch, int n_q_heads, int n_kv_heads, int head_dim,
                       int block_size, float scale, int num_splits, KvStrides kvs, mi_stream stream) {
  int rc = check_attn_common(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size,
                             q_row_stride);
  if (rc != MI_OK) return rc;
  if (!context_lens || !out || !workspace || batch < 0 || table_stride <= 0) return MI_EINVAL;
  if (!aligned16(out) || !aligned16(workspace)) return MI_EINVAL;
  if (batch == 0) return MI_OK;
  if (ws_bytes < mi_paged_attn_decode_workspace(batch, n_q_heads)) return MI_EWORKSPACE;
  const int G = n_q_heads / n_kv_heads;
  const int waves = decode_waves(G);
  int nsplit = num_splits > 0 ? num_splits : decode_splits(batch, n_kv_heads, waves);
  if (nsplit > 16) nsplit = 16;
  float* part_o = static_cast<float*>(workspace);
  float* part_ml = part_o + (size_t)batch * n_q_heads * 16 * 128;
  const float sl2 = scale * 1.4426950408889634f;
  const dim3 grid(nsplit, n_kv_heads, batch);
  hipStream_t st = S(stream);
#define LAUNCH_DEC(GG, WW)                                                                                   \
  hipLaunchKernelGGL((paged_attn_decode_kernel<GG, WW>), grid, dim3(WW * 64), 0, st, q, q_row_stride, k_cache, \
                     v_cache, block_table, table_stride, context_lens, part_o, part_ml, out, n_q_heads, kvs, \
                     block_size / 16, sl2)
  switch (G) {
    case 1: LAUNCH_DEC(1, 16); break;
    case 2: LAUNCH_DEC(2, 16); break;
    case 4: LAUNCH_DEC(4, 16); break;
    case 8: LAUNCH_DEC(8, 8); break;
    default: LAUNCH_DEC(16, 4); break;
  }
#undef LAUNCH_DEC
  rc = check_launch();
  if (rc != MI_OK || nsplit == 1) return rc;
  const int rows = batch * n_q_heads;
  hipLaunchKernelGGL(paged_attn_merge_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, part_o, part_ml, out, rows,
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
                     default_strides(n_kv_heads, block_size > 0 ? block_size / 16 : 1), stream);
}

// experiment entry point (not part of the public header yet): explicit split count and cache strides
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

extern "C" int mi_paged_attn_prefill(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                     const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                     const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                     int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads,
                                     int head_dim, int block_size, float scale, mi_stream stream) {
  int rc = check_attn_common(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size,
                             q_row_stride);
  if (rc != MI_OK) return rc;
  if (!cu_seqlens_q || !kv_lens || !out || n_seqs < 0 || max_seqlen_q < 0 || table_stride <= 0)
    return MI_EINVAL;
  if (!aligned16(out)) return MI_EINVAL;
  if (n_seqs == 0 || max_seqlen_q == 0) return MI_OK;
  const int G = n_q_heads / n_kv_heads;
  const int tq_wg = 8 * (16 / G);  // query tokens per workgroup
  const dim3 grid((max_seqlen_q + tq_wg - 1) / tq_wg, n_kv_heads, n_seqs);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = S(stream);
#define LAUNCH_PRE(GG)                                                                                       \
  hipLaunchKernelGGL((paged_attn_prefill_kernel<GG>), grid, dim3(512), 0, st, q, q_row_stride, k_cache,      \
                     v_cache, block_table, table_stride, cu_seqlens_q, kv_lens, out, n_q_heads, n_kv_heads, \
                     block_size / 16, sl2)
  switch (G) {
    case 1: LAUNCH_PRE(1); break;
    case 2: LAUNCH_PRE(2); break;
    case 4: LAUNCH_PRE(4); break;
    case 8: LAUNCH_PRE(8); break;
    default: LAUNCH_PRE(16); break;
  }
#undef LAUNCH_PRE
  return check_launch();
}
