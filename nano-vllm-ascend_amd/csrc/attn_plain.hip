// Paged attention for the head geometries the MFMA-fragment kernels of paged_attn.hip are not built for:
// head_dim 64 (Llama-3.2-1B, Qwen2-0.5B - the reference's README benchmarks them) and GQA group sizes that are not a
// power of two (7: Qwen2-0.5B / Qwen2.5-7B).  Same operator contract (attention.py:22-93), a PLAIN cache layout
//
//     k_cache / v_cache  [num_blocks][n_kv_heads][block_size][head_dim]  bf16
//
// (a token's head row is contiguous: 128 or 256 bytes), and kernels written for coverage first:
//   * mi_kv_store_plain          K1/K2 of SURVEY 2.2: flat slots (prefill) or [block, offset] pairs (decode)
//   * mi_rope_plain              NeoX rotation with the reference's fp32 roundings (rotary_embedding.py:6-14)
//   * mi_paged_attn_decode_plain one workgroup per (sequence, kv head, context split): a wave-load is 1 KiB = 4 or 8 whole
//                                token rows; scores by fp32 dot products (8 dims per lane, shuffle tree over the row's
//                                lanes), an online softmax per (token slot, q head) in registers, merged over slots,
//                                waves and splits at the end.  HBM-bound in intent: ~G x 34 VALU per KiB pair.
//   * mi_paged_attn_prefill_plain flash attention on v_mfma_f32_16x16x32_bf16: a wave owns 16 query rows of one q head,
//                                the workgroup's four waves share the 32-key K / V chunk through LDS (V stored
//                                transposed), P goes from the C layout to the A layout through a wave-private LDS tile.
// fp32 softmax; P enters the second product as bf16 hi + lo (prefill) / unrounded (decode).
#include <stdlib.h>

#include "mi_common.hpp"

namespace mi {

// ---------------------------------------------------------------------------------------------------
// cache writes
// ---------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void kv_store_plain_kernel(
    const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, int64_t k_stride, int64_t v_stride,
    uint16_t* __restrict__ kc, uint16_t* __restrict__ vc, const int32_t* __restrict__ slots, int slots_2d, int n_tokens,
    int n_kv_heads, int head_dim, int block_size) {
  const int vec_per_tok = n_kv_heads * (head_dim >> 3);
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_tokens * vec_per_tok) return;
  const int t = (int)(idx / vec_per_tok), r = (int)(idx % vec_per_tok);
  const int h = r / (head_dim >> 3), c = r % (head_dim >> 3);
  int64_t blk;
  int off;
  if (slots_2d) {
    blk = slots[2 * t];
    off = slots[2 * t + 1];
    if (blk < 0) return;
  } else {
    const int s = slots[t];
    if (s < 0) return;  // the reference's -1 padding / rows whose KV is already cached
    blk = s / block_size;
    off = s % block_size;
  }
  const int64_t dst = ((blk * n_kv_heads + h) * block_size + off) * head_dim + c * 8;
  *reinterpret_cast<u32x4*>(kc + dst) = *reinterpret_cast<const u32x4*>(k + (int64_t)t * k_stride + h * head_dim + c * 8);
  *reinterpret_cast<u32x4*>(vc + dst) = *reinterpret_cast<const u32x4*>(v + (int64_t)t * v_stride + h * head_dim + c * 8);
}

// ---------------------------------------------------------------------------------------------------
// NeoX RoPE: y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin on the halves (x1 | x2) of a head, fp32 with separate
// roundings, cos_sin[pos] = (cos[0..D/2) | sin[0..D/2)).  One thread: 8 pairs.
// ---------------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void rope_plain_kernel(
    const int64_t* __restrict__ positions, const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ k,
    int64_t k_stride, const float* __restrict__ cos_sin, uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_out,
    int n_tokens, int n_q_heads, int n_kv_heads, int head_dim) {
#pragma clang fp contract(off)
  const int half = head_dim >> 1, vph = half >> 3;  // 16-byte vectors per half head
  const int heads = n_q_heads + n_kv_heads;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_tokens * heads * vph) return;
  const int c = (int)(idx % vph);
  const int hh = (int)((idx / vph) % heads), t = (int)(idx / ((int64_t)vph * heads));
  const bool is_q = hh < n_q_heads;
  const int h = is_q ? hh : hh - n_q_heads;
  const uint16_t* src = is_q ? q + (int64_t)t * q_stride + h * head_dim : k + (int64_t)t * k_stride + h * head_dim;
  uint16_t* dst = is_q ? q_out + ((int64_t)t * n_q_heads + h) * head_dim : k_out + ((int64_t)t * n_kv_heads + h) * head_dim;
  const u32x4 r1 = *reinterpret_cast<const u32x4*>(src + c * 8);
  const u32x4 r2 = *reinterpret_cast<const u32x4*>(src + half + c * 8);
  const float* cs = cos_sin + positions[t] * head_dim;
  float y1[8], y2[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x1[2] = {lo_bf(r1[j]), hi_bf(r1[j])}, x2[2] = {lo_bf(r2[j]), hi_bf(r2[j])};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float co = cs[c * 8 + 2 * j + e], si = cs[half + c * 8 + 2 * j + e];
      y1[2 * j + e] = x1[e] * co - x2[e] * si;
      y2[2 * j + e] = x2[e] * co + x1[e] * si;
    }
  }
  *reinterpret_cast<u32x4*>(dst + c * 8) =
      u32x4{pack_bf(y1[0], y1[1]), pack_bf(y1[2], y1[3]), pack_bf(y1[4], y1[5]), pack_bf(y1[6], y1[7])};
  *reinterpret_cast<u32x4*>(dst + half + c * 8) =
      u32x4{pack_bf(y2[0], y2[1]), pack_bf(y2[2], y2[3]), pack_bf(y2[4], y2[5]), pack_bf(y2[6], y2[7])};
}

// ---------------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------------
constexpr float kNegBig = -1.0e30f;  // "no score yet": finite, so that differences of maxima never produce NaN

// Sum over an aligned group of 8 (or 16) lanes, every lane receiving the total, on the DPP path of the VALU (quad
// permutes, then the half-row / row mirror: the partner holds the other half's sum already) - a __shfl_xor is a
// ds_bpermute, ~100 cycles of LDS crossbar latency per step, three or four dependent steps per score.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int LANES>
__device__ __forceinline__ float group_sum(float s) {
  s += dpp_f<0xB1>(s);   // quad_perm [1, 0, 3, 2]
  s += dpp_f<0x4E>(s);   // quad_perm [2, 3, 0, 1]
  s += dpp_f<0x141>(s);  // row_half_mirror: lane i <-> 7 - i of each 8
  if (LANES == 16) s += dpp_f<0x140>(s);  // row_mirror: lane i <-> 15 - i of each 16
  return s;
}

// merge the online-softmax state (m, l, acc[8]) of two partners
__device__ __forceinline__ void merge_state(float& m, float& l, float (&acc)[8], float m2, float l2, const float (&a2)[8]) {
  const float mn = fmaxf(m, m2);
  const float s1 = __builtin_amdgcn_exp2f(m - mn), s2 = __builtin_amdgcn_exp2f(m2 - mn);
  l = l * s1 + l2 * s2;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = acc[i] * s1 + a2[i] * s2;
  m = mn;
}

// grid (nsplit, n_kv_heads, batch), 256 threads.  part: [batch][n_q_heads][nsplit][D + 2] fp32 (m, l, acc) when nsplit > 1
template <int D, int GMAX>
__global__ __launch_bounds__(256) void attn_decode_plain_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int32_t* __restrict__ block_table, int table_stride, const int32_t* __restrict__ ctx_lens,
    uint16_t* __restrict__ out, float* __restrict__ part, int n_q_heads, int n_kv_heads, int G, int block_size,
    int bs_shift, float scale_log2e) {
  constexpr int LPT = D / 8;     // lanes per token row
  constexpr int TPW = 64 / LPT;  // token rows per wave-load
  __shared__ float sm_acc[4][GMAX][D];
  __shared__ float sm_ml[4][GMAX][2];
  const int split = blockIdx.x, nsplit = gridDim.x, h = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = lane / LPT, dl = lane % LPT;  // token slot of the wave-load, 8-dim piece of the row
  const int len = ctx_lens[b];
  // this split's token range: whole multiples of the workgroup's stride (4 * TPW tokens)
  const int per = ((len + nsplit - 1) / nsplit + 4 * TPW - 1) / (4 * TPW) * (4 * TPW);  // (wave w: groups w, w + 4, ...)
  const int t_beg = split * per, t_end = min(len, t_beg + per);

  float qf[GMAX][8], acc[GMAX][8], m[GMAX], l[GMAX];
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    m[g] = kNegBig;
    l[g] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[g][i] = 0.f;
    if (g < G) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(q + (int64_t)b * q_stride + (int64_t)(h * G + g) * D + dl * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qf[g][2 * j] = lo_bf(raw[j]) * scale_log2e;
        qf[g][2 * j + 1] = hi_bf(raw[j]) * scale_log2e;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[g][i] = 0.f;
    }
  }
  const int32_t* table = block_table + (int64_t)b * table_stride;
  // U wave-loads of K and of V in flight: the block ids of all U first (independent loads), then the 2 U row loads -
  // two dependent memory round trips per U token groups instead of per group
  constexpr int U = 4;
  for (int t0 = t_beg + wave * TPW; t0 < t_end; t0 += U * 4 * TPW) {
    int64_t row[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 4 * TPW + slot;
      valid[u] = t < t_end;
      const int tb = bs_shift >= 0 ? t >> bs_shift : t / block_size;
      const int64_t blk = valid[u] ? table[tb] : 0;
      row[u] = ((blk * n_kv_heads + h) * block_size + (t - tb * block_size)) * D + dl * 8;
    }
    u32x4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      kr[u] = vr[u] = u32x4{0, 0, 0, 0};
      if (valid[u]) {
        kr[u] = *reinterpret_cast<const u32x4*>(kc + row[u]);
        vr[u] = *reinterpret_cast<const u32x4*>(vc + row[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (t0 + u * 4 * TPW >= t_end) break;  // wave-uniform: no lane of this group has a token
      float kf[8], vf[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kf[2 * j] = lo_bf(kr[u][j]);
        kf[2 * j + 1] = hi_bf(kr[u][j]);
        vf[2 * j] = lo_bf(vr[u][j]);
        vf[2 * j + 1] = hi_bf(vr[u][j]);
      }
#pragma unroll
      for (int g = 0; g < GMAX; ++g) {
        if (g < G) {
          float s = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) s += qf[g][i] * kf[i];
          s = group_sum<LPT>(s);  // the row's LPT lanes hold the full dot product
          const float mn = valid[u] ? fmaxf(m[g], s) : m[g];
          const float alpha = __builtin_amdgcn_exp2f(m[g] - mn);
          const float p = valid[u] ? __builtin_amdgcn_exp2f(s - mn) : 0.f;
          l[g] = l[g] * alpha + p;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[g][i] = acc[g][i] * alpha + p * vf[i];
          m[g] = mn;
        }
      }
    }
  }
  // the wave's TPW token slots -> one state per (head, 8-dim piece): partners differ in the slot bits of the lane id
#pragma unroll
  for (int g = 0; g < GMAX; ++g) {
    if (g < G) {
#pragma unroll
      for (int o = LPT; o < 64; o <<= 1) {
        const float m2 = __shfl_xor(m[g], o, 64), l2 = __shfl_xor(l[g], o, 64);
        float a2[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a2[i] = __shfl_xor(acc[g][i], o, 64);
        merge_state(m[g], l[g], acc[g], m2, l2, a2);
      }
      if (slot == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sm_acc[wave][g][dl * 8 + i] = acc[g][i];
        if (dl == 0) {
          sm_ml[wave][g][0] = m[g];
          sm_ml[wave][g][1] = l[g];
        }
      }
    }
  }
  __syncthreads();
  // the four waves -> the result: thread (g, d) for g < G, d < D
  for (int item = tid; item < G * D; item += 256) {
    const int g = item / D, d = item % D;
    float mm = sm_ml[0][g][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) mm = fmaxf(mm, sm_ml[w][g][0]);
    float ll = 0.f, aa = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float sc = __builtin_amdgcn_exp2f(sm_ml[w][g][0] - mm);
      ll += sm_ml[w][g][1] * sc;
      aa += sm_acc[w][g][d] * sc;
    }
    const int hq = h * G + g;
    if (nsplit == 1) {
      out[((int64_t)b * n_q_heads + hq) * D + d] = f2bf(ll > 0.f ? aa / ll : 0.f);
    } else {
      float* p = part + (((int64_t)b * n_q_heads + hq) * nsplit + split) * (D + 2);
      p[2 + d] = aa;
      if (d == 0) {
        p[0] = mm;
        p[1] = ll;
      }
    }
  }
}

// grid (n_q_heads, batch), D threads
template <int D>
__global__ __launch_bounds__(D) void attn_merge_plain_kernel(const float* __restrict__ part, uint16_t* __restrict__ out,
                                                             int n_q_heads, int nsplit) {
  const int hq = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const float* p = part + ((int64_t)b * n_q_heads + hq) * nsplit * (D + 2);
  float mm = kNegBig;
  for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, p[s * (D + 2)]);
  float ll = 0.f, aa = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float sc = __builtin_amdgcn_exp2f(p[s * (D + 2)] - mm);
    ll += p[s * (D + 2) + 1] * sc;
    aa += p[s * (D + 2) + 2 + d] * sc;
  }
  out[((int64_t)b * n_q_heads + hq) * D + d] = f2bf(ll > 0.f ? aa / ll : 0.f);
}

static int decode_plain_splits(int batch, int n_kv_heads, int max_blocks_hint) {
  (void)max_blocks_hint;
  const int wgs = batch * n_kv_heads;
  const int target = tuning(MI_TUNE_PLAIN_SPLIT_TARGET);  // 512: measured best of 128 / 256 / 512
  int ns = 1;
  while (wgs * ns < target && ns < 16) ns *= 2;
  return ns;
}

// ---------------------------------------------------------------------------------------------------
// prefill: grid (ceil(max_q / 64), n_q_heads, n_seqs), 256 threads; wave w owns query rows 16 w .. 16 w + 15 of the block
// ---------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_prefill_plain_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc, const uint16_t* __restrict__ vc,
    const int32_t* __restrict__ block_table, int table_stride, const int32_t* __restrict__ cu_q,
    const int32_t* __restrict__ kv_lens, uint16_t* __restrict__ out, int n_q_heads, int n_kv_heads, int block_size,
    float scale_log2e) {
  constexpr int KS = D / 32;   // k-steps of the first product
  constexpr int NT = D / 16;   // 16-dim output tiles of the second
  constexpr int VP = 40;       // padded row of the transposed V tile (32 keys + 8): 80 bytes, 16-byte aligned pieces
  __shared__ __attribute__((aligned(16))) uint16_t sk[32][D + 8];   // K chunk [key][dim]
  __shared__ __attribute__((aligned(16))) uint16_t svt[D][VP];      // V chunk transposed [dim][key]
  __shared__ __attribute__((aligned(16))) uint16_t sp[4][2][16][VP];  // per wave: P tile [hi | lo][row][key]
  const int qb = blockIdx.x, hq = blockIdx.y, seq = blockIdx.z;
  const int G = n_q_heads / n_kv_heads, h = hq / G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, r = lane & 15;
  const int q0 = cu_q[seq], q_len = cu_q[seq + 1] - q0, kv_len = kv_lens[seq];
  const int row0 = qb * 64;
  if (row0 >= q_len) return;  // uniform for the workgroup
  const int shift = kv_len - q_len;  // position of query row i = shift + i
  const int wg_last = shift + min(row0 + 64, q_len) - 1;  // last key any row of the block sees
  const int n_chunks = wg_last / 32 + 1;
  const int32_t* table = block_table + (int64_t)seq * table_stride;

  // A operand of S = Q K^T: lane (g, r) holds Q[row r][32 ks + 8 g .. +7]; rows past the end read row 0 and are never stored
  const int my_row = row0 + wave * 16 + r;
  u32x4 qa[KS];
  {
    const uint16_t* qrow = q + (int64_t)(q0 + (my_row < q_len ? my_row : 0)) * q_stride + (int64_t)hq * D;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qa[ks] = *reinterpret_cast<const u32x4*>(qrow + 32 * ks + 8 * g);
  }
  // C layout: lane (g, c = r) holds rows 4 g + i (i < 4) of column c
  f32x4 o[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] = kNegBig;
    l[i] = 0.f;
  }
  const int wave_last = shift + min(row0 + wave * 16 + 16, q_len) - 1;  // last key this wave's rows see

  for (int c = 0; c < n_chunks; ++c) {
    __syncthreads();  // everybody is done with the previous chunk's tiles
    // stage the chunk: 32 keys x D dims of K and V, 16 bytes per thread and pass
    for (int v = tid; v < 32 * (D / 8); v += 256) {
      const int key = v / (D / 8), piece = v % (D / 8);
      const int t = c * 32 + key;
      u32x4 kr = {0, 0, 0, 0}, vr = {0, 0, 0, 0};
      if (t < kv_len) {
        const int64_t blk = table[t / block_size];
        const int64_t rowoff = ((blk * n_kv_heads + h) * block_size + t % block_size) * D + piece * 8;
        kr = *reinterpret_cast<const u32x4*>(kc + rowoff);
        vr = *reinterpret_cast<const u32x4*>(vc + rowoff);
      }
      *reinterpret_cast<u32x4*>(&sk[key][piece * 8]) = kr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        svt[piece * 8 + 2 * j][key] = (uint16_t)(vr[j] & 0xffffu);
        svt[piece * 8 + 2 * j + 1][key] = (uint16_t)(vr[j] >> 16);
      }
    }
    __syncthreads();
    if (c * 32 > wave_last) continue;  // nothing of this chunk is visible to this wave's rows (wave-uniform)
    // S tiles: keys 16 nt .. 16 nt + 15.  B operand: lane (g, r) holds K[key 16 nt + r][32 ks + 8 g .. +7]
    f32x4 s[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      s[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 kb = *reinterpret_cast<const u32x4*>(&sk[16 * nt + r][32 * ks + 8 * g]);
        s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(qa[ks]), as_frag(kb), s[nt], 0, 0, 0);
      }
    }
    // scale, causal mask, online softmax per row (rows 4 g + i; the 16 lanes of a row group hold its 32 scores)
    float p[2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pos = shift + row0 + wave * 16 + 4 * g + i;  // the row's own position: keys <= pos are visible
      float mx = kNegBig;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int key = c * 32 + 16 * nt + r;
        const float sv = key <= pos && key < kv_len ? s[nt][i] * scale_log2e : kNegBig;
        p[nt][i] = sv;
        mx = fmaxf(mx, sv);
      }
#pragma unroll
      for (int of = 1; of < 16; of <<= 1) mx = fmaxf(mx, __shfl_xor(mx, of, 64));
      const float mn = fmaxf(m[i], mx);
      const float alpha = __builtin_amdgcn_exp2f(m[i] - mn);
      float rs = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const float pv = p[nt][i] > 0.5f * kNegBig ? __builtin_amdgcn_exp2f(p[nt][i] - mn) : 0.f;
        p[nt][i] = pv;
        rs += pv;
      }
#pragma unroll
      for (int of = 1; of < 16; of <<= 1) rs += __shfl_xor(rs, of, 64);
      l[i] = l[i] * alpha + rs;
      m[i] = mn;
#pragma unroll
      for (int n = 0; n < NT; ++n) o[n][i] *= alpha;
    }
    // P as bf16 hi + lo (two MFMAs per tile: fp32-softmax accuracy, as the fragment-native kernels): C layout -> this
    // wave's LDS tiles [row][key] -> A layout: lane (g, r) holds P[row r][keys 8 g .. 8 g + 7]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint16_t hi = f2bf(p[nt][i]);
        sp[wave][0][4 * g + i][16 * nt + r] = hi;
        sp[wave][1][4 * g + i][16 * nt + r] = f2bf(p[nt][i] - bf2f(hi));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes, then its reads (one wave: in order)
    const u32x4 pa = *reinterpret_cast<const u32x4*>(&sp[wave][0][r][8 * g]);
    const u32x4 pl = *reinterpret_cast<const u32x4*>(&sp[wave][1][r][8 * g]);
    // O tiles: dims 16 n .. 16 n + 15.  B operand: lane (g, r) holds V[keys 8 g .. 8 g + 7][dim 16 n + r] = svt[dim][key..]
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const u32x4 vb = *reinterpret_cast<const u32x4*>(&svt[16 * n + r][8 * g]);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(pa), as_frag(vb), o[n], 0, 0, 0);
      o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag(pl), as_frag(vb), o[n], 0, 0, 0);
    }
  }
  // rows 4 g + i, dims 16 n + r
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + wave * 16 + 4 * g + i;
    if (row < q_len) {
      const float inv = l[i] > 0.f ? 1.0f / l[i] : 0.f;
      uint16_t* orow = out + ((int64_t)(q0 + row) * n_q_heads + hq) * D;
#pragma unroll
      for (int n = 0; n < NT; ++n) orow[16 * n + r] = f2bf(o[n][i] * inv);
    }
  }
}

static int check_plain(const void* q, const void* kc, const void* vc, const void* bt, int n_q_heads, int n_kv_heads,
                       int head_dim, int block_size, int64_t q_stride) {
  if (!q || !kc || !vc || !bt || n_q_heads <= 0 || n_kv_heads <= 0 || block_size <= 0) return MI_EINVAL;
  if (head_dim != 64 && head_dim != 128) return MI_EUNSUPPORTED;
  if (n_q_heads % n_kv_heads || n_q_heads / n_kv_heads > 8) return MI_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(kc) || !aligned16(vc) || q_stride % 8) return MI_EINVAL;
  return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" int mi_kv_store_plain(const mi_bf16* k, const mi_bf16* v, int64_t k_row_stride, int64_t v_row_stride,
                                 mi_bf16* k_cache, mi_bf16* v_cache, const int32_t* slots, int slots_2d, int n_tokens,
                                 int n_kv_heads, int head_dim, int block_size, mi_stream stream) {
  if (!k || !v || !k_cache || !v_cache || !slots || n_tokens < 0 || n_kv_heads <= 0 || block_size <= 0) return MI_EINVAL;
  if (head_dim <= 0 || head_dim % 8 || k_row_stride % 8 || v_row_stride % 8) return MI_EUNSUPPORTED;
  if (!aligned16(k) || !aligned16(v) || !aligned16(k_cache) || !aligned16(v_cache)) return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  const int64_t n = (int64_t)n_tokens * n_kv_heads * (head_dim / 8);
  hipLaunchKernelGGL(kv_store_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), k, v,
                     k_row_stride, v_row_stride, k_cache, v_cache, slots, slots_2d, n_tokens, n_kv_heads, head_dim,
                     block_size);
  return check_launch();
}

extern "C" int mi_rope_plain(const int64_t* positions, const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k,
                             int64_t k_row_stride, const float* cos_sin, mi_bf16* q_out, mi_bf16* k_out, int n_tokens,
                             int n_q_heads, int n_kv_heads, int head_dim, mi_stream stream) {
  if (!positions || !q || !k || !cos_sin || !q_out || !k_out || n_tokens < 0 || n_q_heads <= 0 || n_kv_heads <= 0)
    return MI_EINVAL;
  if (head_dim <= 0 || head_dim % 16 || q_row_stride % 8 || k_row_stride % 8) return MI_EUNSUPPORTED;
  if (!aligned16(q) || !aligned16(k) || !aligned16(q_out) || !aligned16(k_out)) return MI_EINVAL;
  if (n_tokens == 0) return MI_OK;
  const int64_t n = (int64_t)n_tokens * (n_q_heads + n_kv_heads) * (head_dim / 16);
  hipLaunchKernelGGL(rope_plain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), positions, q,
                     q_row_stride, k, k_row_stride, cos_sin, q_out, k_out, n_tokens, n_q_heads, n_kv_heads, head_dim);
  return check_launch();
}

extern "C" size_t mi_paged_attn_decode_plain_workspace(int batch, int n_q_heads, int head_dim) {
  if (batch <= 0 || n_q_heads <= 0 || head_dim <= 0) return 0;
  return (size_t)batch * n_q_heads * 16 * (head_dim + 2) * sizeof(float);
}

extern "C" int mi_paged_attn_decode_plain(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                          const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                          const int32_t* context_lens, mi_bf16* out, void* workspace, size_t ws_bytes,
                                          int batch, int n_q_heads, int n_kv_heads, int head_dim, int block_size,
                                          float scale, mi_stream stream) {
  const int rc = check_plain(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size, q_row_stride);
  if (rc != MI_OK) return rc;
  if (!context_lens || !out || batch < 0 || table_stride <= 0) return MI_EINVAL;
  if (batch == 0) return MI_OK;
  const int G = n_q_heads / n_kv_heads;
  const int ns = decode_plain_splits(batch, n_kv_heads, table_stride);
  if (ns > 1 && (!workspace || !aligned16(workspace) ||
                 ws_bytes < mi_paged_attn_decode_plain_workspace(batch, n_q_heads, head_dim)))
    return MI_EWORKSPACE;
  float* part = static_cast<float*>(workspace);
  const dim3 grid(ns, n_kv_heads, batch);
  const float sl2 = scale * 1.4426950408889634f;
  int bs_shift = -1;  // block sizes that are a power of two: shifts instead of per-lane integer divisions
  for (int sft = 0; sft < 16; ++sft)
    if ((1 << sft) == block_size) bs_shift = sft;
  hipStream_t st = S(stream);
#define DEC_PLAIN(DD, GM)                                                                                          \
  hipLaunchKernelGGL((attn_decode_plain_kernel<DD, GM>), grid, dim3(256), 0, st, q, q_row_stride, k_cache, v_cache, \
                     block_table, table_stride, context_lens, out, part, n_q_heads, n_kv_heads, G, block_size, bs_shift, \
                     sl2)
  if (head_dim == 64) {
    if (G <= 4) DEC_PLAIN(64, 4);
    else DEC_PLAIN(64, 8);
  } else {
    if (G <= 4) DEC_PLAIN(128, 4);
    else DEC_PLAIN(128, 8);
  }
#undef DEC_PLAIN
  int rc2 = check_launch();
  if (rc2 != MI_OK || ns == 1) return rc2;
  if (head_dim == 64)
    hipLaunchKernelGGL((attn_merge_plain_kernel<64>), dim3(n_q_heads, batch), dim3(64), 0, st, part, out, n_q_heads, ns);
  else
    hipLaunchKernelGGL((attn_merge_plain_kernel<128>), dim3(n_q_heads, batch), dim3(128), 0, st, part, out, n_q_heads, ns);
  return check_launch();
}

extern "C" int mi_paged_attn_prefill_plain(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                           const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                           const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                           int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads, int head_dim,
                                           int block_size, float scale, mi_stream stream) {
  const int rc = check_plain(q, k_cache, v_cache, block_table, n_q_heads, n_kv_heads, head_dim, block_size, q_row_stride);
  if (rc != MI_OK) return rc;
  if (!cu_seqlens_q || !kv_lens || !out || n_seqs < 0 || max_seqlen_q < 0 || table_stride <= 0) return MI_EINVAL;
  if (n_seqs == 0 || max_seqlen_q == 0) return MI_OK;
  const dim3 grid((max_seqlen_q + 63) / 64, n_q_heads, n_seqs);
  const float sl2 = scale * 1.4426950408889634f;
  if (head_dim == 64)
    hipLaunchKernelGGL((attn_prefill_plain_kernel<64>), grid, dim3(256), 0, S(stream), q, q_row_stride, k_cache, v_cache,
                       block_table, table_stride, cu_seqlens_q, kv_lens, out, n_q_heads, n_kv_heads, block_size, sl2);
  else
    hipLaunchKernelGGL((attn_prefill_plain_kernel<128>), grid, dim3(256), 0, S(stream), q, q_row_stride, k_cache,
                       v_cache, block_table, table_stride, cu_seqlens_q, kv_lens, out, n_q_heads, n_kv_heads, block_size,
                       sl2);
  return check_launch();
}
