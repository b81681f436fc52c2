// EXPERIMENT, NOT PART OF THE LIBRARY (README.md next to this file: slower than the 32-column kernel, and not robust).
// Prefill attention with SIXTY-FOUR query columns per wave (round 6), 128-wide heads.
// Reference operator: /root/reference/nanovllm/layers/attention.py:46-59 (the prefill branch: causal attention of the
// new tokens over the paged cache, block tables and cu_seqlens as given).
//
// paged_attn.hip's prefill kernel gives a wave 32 columns (query tokens x heads of one kv head) and relies on two
// workgroups per CU for overlap: per 32-key chunk a wave reads 8 K and 16 V fragments from LDS for 16 MFMAs, passes one
// s_barrier, and its score MFMAs -> exponentials -> sum -> test -> pack -> PV MFMAs run as one serial chain
// (profiles/r06_prefill_attention_pmc.txt: a wave-chunk takes 3.1 k cycles for 512 cycles of MFMA issue).
// Here a wave owns TWO 32-column blocks and the whole 512-register file (one workgroup of four waves per CU, one wave
// per SIMD - the shape of the guide's 4 x 64 loop):
//   * every K / V fragment read from LDS feeds two MFMAs (one per column block): 32 MFMAs per chunk and barrier
//     for the same 24 LDS reads;
//   * the chain is cut by a software pipeline INSIDE the wave: iteration c issues the score MFMAs of chunk c + 1 and
//     the PV MFMAs of chunk c - 1 (32 independent-of-the-VALU MFMAs, 1024 cycles of matrix issue) in the same basic
//     block as the exponentials of chunk c, so the matrix pipe works while the VALU takes the softmax;
//   * the LDS ring is five chunks deep (80 KiB): V of chunk c - 1 and K of chunk c + 1 are being read while chunks
//     c + 2 and c + 3 are landing; requests stop at the last chunk (the waits count what is really in flight).
// The arithmetic is the 32-column kernel's: S^T = K . Q^T and O^T += V^T . P^T on v_mfma_f32_32x32x16_bf16 with keys
// on the M axis, P as one bf16 per key, the running sum from the un-rounded exponentials, the reference point of
// the exponentials moved only when a lane's sum of them exceeds 2^8 (prefill_common.hpp: kDeferMax) - so a column's
// result is the same function of the same chunk sequence, and the bound tests of the 32-column kernel apply unchanged.
#include <limits.h>

#include <type_traits>

#include "mi_common.hpp"
#include "kv_store.hpp"
#include "prefill_common.hpp"

namespace mi {

constexpr int P64_NBUF = 5;                                 // ring depth, chunks
constexpr int P64_TILE = 2048;                              // elements of a 16-token x 128-dim cache tile
constexpr int P64_RING_BYTES = P64_NBUF * 4 * P64_TILE * 2;  // [slot][K0, K1, V0, V1][tile] = 80 KiB
constexpr int P64_LDS_BYTES = P64_RING_BYTES + 256 * 256;   // + 64 KiB: the prepared Q operand on its way to a[128:191]

typedef __attribute__((address_space(3))) uint16_t lds_u16;

// STAMP (tools/ubench/prefill64_timeline.hip only): the wave sums the shader cycles it spends in each segment of a stage
// (SGPR accumulators; s_memtime) and in its prologue / epilogue, and lane 0 writes them to stamps[workgroup][wave][16]
// at the end: [0] stages, [1] wait + barrier, [2] request, [3] mask, [4] the stage's instruction stream, [5] test + pack,
// [8] Q preparation, [9] first chunk landed, [10] scores of chunk 0, [11] loop, [12] last product + drain, [13] epilogue
#define MI_P64_T(k)                                          \
  do {                                                       \
    if constexpr (STAMP) {                                   \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
      tsum[k] += now_ - tlast;                               \
      tlast = now_;                                          \
    }                                                        \
  } while (0)

template <int G, bool FUSE_Q, bool STAMP = false>
__global__ __launch_bounds__(256) void paged_attn_prefill64_kernel(
    const uint16_t* __restrict__ q, int64_t q_stride, const uint16_t* __restrict__ kc,
    const uint16_t* __restrict__ vc, const int32_t* __restrict__ block_table, int table_stride,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ kv_lens, uint16_t* __restrict__ out,
    int n_q_heads, int n_kv_heads, int tpb, int tpb_shift, float scale_log2e, int n_qblocks, int n_pairs, QPrep qp,
    unsigned long long* __restrict__ stamps = nullptr) {
  unsigned long long tsum[16] = {}, tlast = 0;
  if constexpr (STAMP) tlast = __builtin_amdgcn_s_memtime();
  static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16, "query heads per kv head: a power of two <= 16");
  constexpr int D = 128, TILE = P64_TILE, NBUF = P64_NBUF;
  constexpr int TPC = 32 / G;    // query tokens per column block
  constexpr int TQ = 2 * TPC;    // per wave
  constexpr int TQ_WG = 4 * TQ;  // per workgroup
  extern __shared__ __attribute__((aligned(16))) uint16_t stage[];  // [NBUF][4][TILE]

  // workgroup -> (sequence, kv head, query block): the XCD-aware order of the 32-column kernel (all query blocks of a
  // pair on one XCD's L2, heaviest blocks first)
  const int pair = ((int)blockIdx.x / (8 * n_qblocks)) * 8 + ((int)blockIdx.x & 7);
  const int qblock = ((int)blockIdx.x >> 3) % n_qblocks;
  if (pair >= n_pairs) return;
  const int seq = pair / n_kv_heads, h = pair % n_kv_heads;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, n = lane & 31;
  const int q_start = cu_q[seq];
  const int q_len = cu_q[seq + 1] - q_start;
  const int kv_len = kv_lens[seq];
  const int wg_qt0 = (n_qblocks - 1 - qblock) * TQ_WG;
  if (wg_qt0 >= q_len) return;  // uniform for the workgroup
  const int shift = kv_len - q_len;
  const int wg_last_pos = shift + min(wg_qt0 + TQ_WG, q_len) - 1;
  const int wg_chunks = (wg_last_pos + 32) >> 5, wg_tiles = (wg_last_pos + 16) >> 4;

  const int qt0 = wg_qt0 + wave * TQ;
  const bool wave_on = qt0 < q_len;
  const int hn = n % G;
  int limit[2];  // keys [0, limit) are visible to this lane's column of block cb
  bool valid[2];
  int my_qt[2];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    my_qt[cb] = qt0 + cb * TPC + n / G;
    valid[cb] = wave_on && my_qt[cb] < q_len;
    limit[cb] = valid[cb] ? shift + my_qt[cb] + 1 : 1;
  }
  const int wave_chunks = wave_on ? (shift + min(qt0 + TQ, q_len) - 1 + 32) >> 5 : 0;  // chunks this wave's columns see
  const int limit_all = wave_on && qt0 + TQ <= q_len ? shift + qt0 + 1 : 0;  // earliest column of a full wave

  // cooperative fetch by LDS-DMA: piece = wave (K0, K1, V0, V1), the 4 KiB cache tile as it is stored, 1 KiB per
  // instruction; the tile's address stays on the scalar unit (buffer descriptor per request, lane * 16 the only
  // vector offset)
  const int32_t* table_row = block_table + (int64_t)seq * table_stride;
  const KvStrides st = default_strides(n_kv_heads, tpb, 16 * D);
  const int piece = wave, odd = piece & 1;
  const uint16_t* const cache_hs = ((piece & 2) ? vc : kc) + (int64_t)h * st.head;  // (wave-uniform)
  const uint32_t lane16 = lane * 16;
  auto tile_of = [&](int r) { return min(2 * r + odd, wg_tiles - 1); };  // (an odd tile past the end: the even one again, masked)
  auto block_of = [&](int r) {
    const int tile = tile_of(r);
    return table_row[tpb_shift >= 0 ? tile >> tpb_shift : tile / tpb];
  };
  auto issue = [&](int r, int slot, int blk) {
    const int tile = tile_of(r);
    const int in_block = tpb_shift >= 0 ? tile & (tpb - 1) : tile % tpb;
    const uint16_t* base = cache_hs + (int64_t)blk * (int)st.block + in_block * TILE;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(base), 0, 2 * TILE, 0x00020000);
    uint16_t* dst = stage + (slot * 4 + piece) * TILE;
#define MI_REQ_PIECE(I)                                                                                              \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + 512 * (I)), 16, lane16, \
                                           1024 * (I), 0, 0)
    MI_REQ_PIECE(0);
    MI_REQ_PIECE(1);
    MI_REQ_PIECE(2);
    MI_REQ_PIECE(3);
#undef MI_REQ_PIECE
  };

  // the first three chunks are requested before anything else: they land under the Q preparation
  const int n_first = min(3, wg_chunks);
  for (int r = 0; r < n_first; ++r) issue(r, r, block_of(r));

  // B operand of S^T = K . Q^T: column n of block cb, dims 16 kk + 8 hi .. +7.  Q lives in the ACCUMULATION registers for
  // the whole kernel (a[128:191]; the MFMA reads a B operand from either file): each prepared fragment takes the one
  // road on which the compiler keeps it there - through this lane's 256 bytes of LDS (behind the ring) and back by a
  // ds_read_b128 whose destination is declared "=a".  (An "a" INPUT built from an arch-VGPR value is copied in front of
  // every use: 64 v_accvgpr_write per stage and 64 arch VGPRs lost, measured on the first build.)
  // One column block after the other: only ONE block's eight fragments (and rope registers) are alive at a time.  (Both
  // blocks' loads ahead of any arithmetic - 64 more live registers across the prologue - saved ~10 k cycles of the
  // prologue's dependent round trips and made the register allocator SPILL pinned accumulators inside the stage, i.e.
  // read them right behind the inline-asm MFMA that writes them: wrong results.  This order did not remove the spills
  // either - see README.md next to this file: the kernel is an experiment, not part of the library.)
  bf16x8 Q[2][8];
  const uint32_t lds_base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)stage;
  const uint32_t qslot = lds_base + P64_RING_BYTES + threadIdx.x * 256;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    u32x4 qv[8];
    const int row = valid[cb] ? my_qt[cb] : wg_qt0;  // invalid columns read a valid row and are zeroed
    const uint16_t* qrow = q + (int64_t)(q_start + row) * q_stride + (int64_t)(h * G + hn) * D + 8 * hi;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qv[kk] = *reinterpret_cast<const u32x4*>(qrow + 16 * kk);
    if constexpr (FUSE_Q) {
      RopeRegs32 rr;
      rope_regs_q32_load(rr, qp.cos_sin + qp.positions[q_start + row] * 128, hi);
      head_rmsnorm_rope_q32_packed(qv, qp.q_w, rr, hi, qp.eps);
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if (!valid[cb]) qv[kk] = u32x4{0, 0, 0, 0};
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(qslot), "v"(qv[kk]), "i"((cb * 8 + kk) * 16) : "memory");
    }
  }
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(Q[cb][kk]) : "v"(qslot), "i"((cb * 8 + kk) * 16) : "memory");
  // (LDS operations of a wave complete in order: the reads see the writes.)  The first chunks were requested before
  // the Q preparation: every request of this wave has landed; after the barrier everybody's have.
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  MI_P64_T(8);

  float m[2], l[2] = {0.f, 0.f};  // running reference point (log2 domain) / this lane's share of the sum
  f32x16 acc[2][4];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[cb][j][i] = 0.f;

  // LDS element offsets of this lane's operand pieces inside a tile (the 32-column kernel's)
  //   K (A of the first product): key n (tile n / 16, token n % 16), dims 16 kk + 8 hi .. +7
  const int k_off = (n >> 4) * TILE + (hi * 16 + (n & 15)) * 8;           // + (kk / 2) * 512 + (kk % 2) * 256
  //   V (A of the second): dim 32 db + n, key slots {4 hi .. +3} and {8 + 4 hi .. +3} of segment sg
  const int v_off = 2 * TILE + (hi * 16 + (n & 15)) * 8 + (n >> 4) * 4;   // + sg * TILE + db * 512 (+ 256: second piece)
  const lds_u16* const lds0 = (const lds_u16*)stage;

  f32x16 S0[2], S1[2];  // raw scores: of the chunk whose softmax is next / being produced (the roles alternate)
  u32x4 ph[2][2];       // probabilities of the chunk whose PV product is next: [column block][key segment]
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) ph[cb][sg] = u32x4{0, 0, 0, 0};
  const uint32_t kbase = lds_base + 2 * k_off, vbase = lds_base + 2 * v_off;  // + slot * 32 KiB

  // The matrix instructions are inline asm (the accumulators and Q stay in the accumulation registers, destination =
  // addend, and the ORDER of the stage is tools/gen_prefill64_body.py's): the compiler's hazard recogniser does not see
  // them.  Every consumer of their results is far behind them by construction (scores: a barrier later; O: the rare
  // rescale branch and the epilogue, which wait out the last MFMA's passes explicitly).
#define MI_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")

  // scores of chunk 0 (prologue; the loop produces every later chunk's a stage ahead)
  auto qk_first = [&](f32x16 (&Sx)[2]) __attribute__((always_inline)) {
    u32x4 Kf[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      asm volatile("ds_read_b128 %0, %1" : "=v"(Kf[kk]) : "v"(kbase + (kk >> 1) * 1024 + (kk & 1) * 512));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(Sx[cb]) : "v"(Kf[0]), "a"(Q[cb][0]));
#pragma unroll
      for (int kk = 1; kk < 8; ++kk)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(Sx[cb]) : "v"(Kf[kk]), "a"(Q[cb][kk]));
    }
    MI_MFMA_DRAIN();
  };
  // the PV product of a wave's LAST chunk (nothing left to overlap it with)
  auto pv_last = [&](int slot) __attribute__((always_inline)) {
    const uint32_t vaddr = vbase + slot * (8 * TILE);
    uint64_t Vf[8][2];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int half = 0; half < 2; ++half)
        asm volatile("ds_read_b64 %0, %1" : "=v"(Vf[f][half]) : "v"(vaddr + (f >> 2) * 4096 + (f & 3) * 1024 + half * 512));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const u32x4 va = {(uint32_t)Vf[f][0], (uint32_t)(Vf[f][0] >> 32), (uint32_t)Vf[f][1], (uint32_t)(Vf[f][1] >> 32)};
#define MI_PV_LAST(CB, DB, LO)                                                                                        \
  if (f == 4 * (f >> 2) + DB)                                                                                         \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+{a[" #LO "]}"(acc[CB][DB]) : "v"(va), "v"(ph[CB][f >> 2]))
      MI_PV_LAST(0, 0, 0:15);
      MI_PV_LAST(1, 0, 64:79);
      MI_PV_LAST(0, 1, 16:31);
      MI_PV_LAST(1, 1, 80:95);
      MI_PV_LAST(0, 2, 32:47);
      MI_PV_LAST(1, 2, 96:111);
      MI_PV_LAST(0, 3, 48:63);
      MI_PV_LAST(1, 3, 112:127);
#undef MI_PV_LAST
    }
  };
  // scores of keys a column does not see -> -inf.  (volatile asm: a REAL branch around it - as plain selects the
  // compiler ran these 130 instructions in every stage)
  auto mask_scores = [&](f32x16 (&Sc)[2], int c) __attribute__((always_inline)) {
    const int tok0 = c * 32 + 4 * hi;  // register i: key tok0 + (i & 3) + 8 (i >> 2)
    const float ninf = -INFINITY;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_cndmask_b32 %0, %3, %0, vcc"
                     : "+v"(Sc[cb][i])
                     : "v"(limit[cb]), "v"(tok0 + (i & 3) + 8 * (i >> 2)), "v"(ninf)
                     : "vcc");
  };
  // One pipeline stage: scores of chunk c + 1 (-> Sx) and the PV product of chunk c - 1 on the matrix pipe, the
  // exponentials of chunk c (from Sc) on the VALU - prefill64_body.inc, one basic block.  p = 2^(s * scale - m) against
  // the CURRENT reference point at once; the lane's sum of them is the test (<= 2^8 bounds every probability by 2^8);
  // only when some lane fails it is the chunk's maximum taken, the state rescaled and the exponentials redone (the
  // first chunk; a maximum that ran away).
  auto stage_body = [&](f32x16 (&Sc)[2], f32x16 (&Sx)[2], int c, int slot_k, int slot_v) __attribute__((always_inline)) {
    if (c * 32 + 32 > limit_all) mask_scores(Sc, c);  // wave-uniform: some column does not see the whole chunk
    MI_P64_T(3);
    const uint32_t kaddr = kbase + slot_k * (8 * TILE), vaddr = vbase + slot_v * (8 * TILE);
    u32x4 Kf[8];
    uint64_t Vf[8][2];
    float p[2][16], tq[2][8], lc[2];
#include "prefill64_body.inc"
    MI_P64_T(4);
    if (__any(!(lc[0] <= 256.0f) || !(lc[1] <= 256.0f))) {
      MI_MFMA_DRAIN();
      float alpha[2], tmp;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const f32x16& s = Sc[cb];
        float mc = fmaxf(fmaxf(fmaxf(fmaxf(s[0], s[1]), s[2]), fmaxf(fmaxf(s[3], s[4]), s[5])),
                         fmaxf(fmaxf(fmaxf(s[6], s[7]), s[8]), fmaxf(fmaxf(s[9], s[10]), s[11])));
        mc = fmaxf(mc, fmaxf(fmaxf(fmaxf(s[12], s[13]), s[14]), s[15]));
        const float mn = fmaxf(m[cb], xor32_max(mc) * scale_log2e);  // scale > 0: the max commutes with it
        alpha[cb] = __builtin_amdgcn_exp2f(m[cb] - mn);
        l[cb] *= alpha[cb];
        m[cb] = mn;
#pragma unroll
        for (int i = 0; i < 16; ++i) p[cb][i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[i], scale_log2e, -mn));
        lc[cb] = ((p[cb][0] + p[cb][4]) + (p[cb][8] + p[cb][12])) + ((p[cb][1] + p[cb][5]) + (p[cb][9] + p[cb][13])) +
                 (((p[cb][2] + p[cb][6]) + (p[cb][10] + p[cb][14])) + ((p[cb][3] + p[cb][7]) + (p[cb][11] + p[cb][15])));
      }
      // O *= alpha on the accumulation registers, in place (read - multiply - write per register)
#include "prefill64_rescale.inc"
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      l[cb] += lc[cb];
#pragma unroll
      for (int i = 0; i < 8; ++i) ph[cb][i >> 2][i & 3] = pack_bf(p[cb][2 * i], p[cb][2 * i + 1]);
    }
    if constexpr (STAMP) {
      asm volatile("" ::"v"(ph[0][0]), "v"(ph[0][1]), "v"(ph[1][0]), "v"(ph[1][1]));  // (the pack belongs to this segment)
      MI_P64_T(5);
      tsum[0] += 1;
    }
  };

  // The loop's waits are COUNTS ("all but my newest 4 k requests have completed"): true only because the chunk DMAs are
  // this wave's only vector-memory operations in flight - the Q operand is finished and the counter drained before the
  // first request, block ids come through the scalar cache, the output stores follow the last wait.
  MI_P64_T(9);
  qk_first(S0);
  // The reference point starts at chunk 0's maximum (key 0 is visible to every column: finite), so the first stage
  // takes the common path like every other (from m = -inf it took the rescale branch: ~3 k cycles per workgroup).
  if (32 > limit_all) mask_scores(S0, 0);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const f32x16& sv = S0[cb];
    float mc = fmaxf(fmaxf(fmaxf(fmaxf(sv[0], sv[1]), sv[2]), fmaxf(fmaxf(sv[3], sv[4]), sv[5])),
                     fmaxf(fmaxf(fmaxf(sv[6], sv[7]), sv[8]), fmaxf(fmaxf(sv[9], sv[10]), sv[11])));
    mc = fmaxf(mc, fmaxf(fmaxf(fmaxf(sv[12], sv[13]), sv[14]), sv[15]));
    m[cb] = xor32_max(mc) * scale_log2e;
  }
  MI_P64_T(10);
  int blk_next = 3 < wg_chunks ? block_of(3) : 0;  // block id of the NEXT request: read one iteration ahead of its use
  int slot_v = NBUF - 1, slot_k = 1, slot_r = 3;   // ring slots of chunks c - 1, c + 1, c + 3
  auto iteration = [&](int c, f32x16 (&Sc)[2], f32x16 (&Sx)[2]) __attribute__((always_inline)) {
    // my pieces of chunk c + 1 have landed (those of c + 2 may still fly); after the barrier everybody's have, and
    // everybody is done with iteration c - 1, i.e. with chunk c - 2, whose slot chunk c + 3 takes
    if (c + 2 < wg_chunks) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    MI_P64_T(1);
    if (c + 3 < wg_chunks) {
      issue(c + 3, slot_r, blk_next);
      if (c + 4 < wg_chunks) blk_next = block_of(c + 4);
    }
    MI_P64_T(2);
    // A wave whose columns end before the workgroup's runs ONE stage more than it has chunks (that stage's PV product is
    // its last chunk's; its own chunk is masked out entirely: probabilities of zero) and then only keeps the barriers
    // and its share of the requests.  The loop has ONE stage body (the accumulators are pinned: a skipped stage costs no
    // register traffic).  (chunk 0 has no predecessor: its "PV product" multiplies chunk 0's own V by zeros)
    if (c <= wave_chunks) stage_body(Sc, Sx, c, slot_k, c == 0 ? 0 : slot_v);
    slot_v = slot_v + 1 == NBUF ? 0 : slot_v + 1;
    slot_k = slot_k + 1 == NBUF ? 0 : slot_k + 1;
    slot_r = slot_r + 1 == NBUF ? 0 : slot_r + 1;
  };
  if constexpr (STAMP) tlast = __builtin_amdgcn_s_memtime();
  const unsigned long long t_loop = tlast;
  for (int c = 0; c < wg_chunks; c += 2) {
    iteration(c, S0, S1);
    if (c + 1 < wg_chunks) iteration(c + 1, S1, S0);
  }
  if constexpr (STAMP) tsum[11] = tlast - t_loop;
  pv_last(slot_v);  // (slot_v is now the last chunk's)
  MI_MFMA_DRAIN();
#undef MI_MFMA_DRAIN
  __syncthreads();
  MI_P64_T(12);

  // O^T -> rows: each wave transposes its two [128 dims][32 cols] tiles through its own 16 KiB of the ring (all reads
  // are behind the barrier).  Column n's 256-byte row is XOR-swizzled in 8-byte pieces: the 32 lanes of a write hit
  // 32 different bank pairs.
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const float lt = xor32_sum(l[cb]);
    const float inv = lt > 0.f ? 1.0f / lt : 0.f;
    uint16_t* tile = stage + wave * (64 * D) + cb * (32 * D);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int piece8 = db * 8 + i4 * 2 + hi;  // dims 4 piece8 .. +3  (= 32 db + 8 i4 + 4 hi)
        u32x2 o;
        o[0] = pack_bf(acc[cb][db][4 * i4] * inv, acc[cb][db][4 * i4 + 1] * inv);
        o[1] = pack_bf(acc[cb][db][4 * i4 + 2] * inv, acc[cb][db][4 * i4 + 3] * inv);
        *reinterpret_cast<u32x2*>(tile + n * D + ((piece8 ^ n) & 31) * 4) = o;
      }
    // same wave wrote and reads: LDS operations of a wave complete in order
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int col = it * 4 + (lane >> 4), dl = lane & 15;  // 16 lanes store one row of 256 bytes
      const int qt = qt0 + cb * TPC + col / G;
      const u32x2 lo = *reinterpret_cast<const u32x2*>(tile + col * D + (((2 * dl) ^ col) & 31) * 4);
      const u32x2 hh = *reinterpret_cast<const u32x2*>(tile + col * D + (((2 * dl + 1) ^ col) & 31) * 4);
      if (wave_on && qt < q_len) {
        uint16_t* op = out + ((int64_t)(q_start + qt) * n_q_heads + h * G + col % G) * D + 8 * dl;
        *reinterpret_cast<u32x4*>(op) = u32x4{lo[0], lo[1], hh[0], hh[1]};
      }
    }
  }
  if constexpr (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI_P64_T(13);
    if (stamps != nullptr && lane == 0) {
      unsigned long long* dst = stamps + ((int64_t)blockIdx.x * 4 + wave) * 16;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i] = tsum[i];
    }
  }
}

template <int G, bool FQ>
static int launch64(dim3 grid, hipStream_t st, const uint16_t* q, int64_t q_stride, const uint16_t* kc, const uint16_t* vc,
                    const int32_t* block_table, int table_stride, const int32_t* cu_q, const int32_t* kv_lens,
                    uint16_t* out, int n_q_heads, int n_kv_heads, int tpb, int tpb_shift, float sl2, int n_qblocks,
                    int n_pairs, QPrep qp) {
  auto kern = paged_attn_prefill64_kernel<G, FQ>;
  static bool once = false;  // (80 KiB of dynamic LDS: above the default limit of a launch)
  if (!once) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P64_LDS_BYTES) !=
        hipSuccess)
      return MI_ELAUNCH;
    once = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), P64_LDS_BYTES, st, q, q_stride, kc, vc, block_table, table_stride, cu_q,
                     kv_lens, out, n_q_heads, n_kv_heads, tpb, tpb_shift, sl2, n_qblocks, n_pairs, qp, nullptr);
  return MI_OK;
}

int prefill64_launch(const uint16_t* q, int64_t q_stride, const QPrep* prep, const uint16_t* kc, const uint16_t* vc,
                     const int32_t* block_table, int table_stride, const int32_t* cu_q, const int32_t* kv_lens,
                     int n_seqs, int max_seqlen_q, uint16_t* out, int n_q_heads, int n_kv_heads, int block_size,
                     float scale_log2e, int variant, hipStream_t st) {
  const int G = n_q_heads / n_kv_heads;
  if (G != 1 && G != 2 && G != 4 && G != 8 && G != 16) return MI_EUNSUPPORTED;
  if (variant != 0) return MI_EUNSUPPORTED;
  const int tq_wg = 4 * 2 * (32 / G);
  const int n_qblocks = (max_seqlen_q + tq_wg - 1) / tq_wg, n_pairs = n_seqs * n_kv_heads;
  const dim3 grid((unsigned)((n_pairs + 7) / 8 * 8 * n_qblocks));
  int tpb_shift = -1;
  for (int sft = 0; sft < 12; ++sft)
    if ((1 << sft) == block_size / 16) tpb_shift = sft;
  const QPrep qp = prep ? *prep : QPrep{nullptr, nullptr, nullptr, 0.f};
#define MI_L64(GG)                                                                                                     \
  return prep ? launch64<GG, true>(grid, st, q, q_stride, kc, vc, block_table, table_stride, cu_q, kv_lens, out,        \
                                   n_q_heads, n_kv_heads, block_size / 16, tpb_shift, scale_log2e, n_qblocks, n_pairs,   \
                                   qp)                                                                                   \
              : launch64<GG, false>(grid, st, q, q_stride, kc, vc, block_table, table_stride, cu_q, kv_lens, out,       \
                                    n_q_heads, n_kv_heads, block_size / 16, tpb_shift, scale_log2e, n_qblocks, n_pairs,  \
                                    qp)
  switch (G) {
    case 1: MI_L64(1);
    case 2: MI_L64(2);
    case 4: MI_L64(4);
    case 8: MI_L64(8);
    default: MI_L64(16);
  }
#undef MI_L64
}

}  // namespace mi
