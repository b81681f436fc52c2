// Index arithmetic of the 256 x 256 x 64 MFMA tile GEMM (gemm_tile.hip), kept apart from the kernel so that a
// host program can replay it (tests/test_gemm_tile_index.py compiles this header with g++ and checks that every
// fragment read finds the element the LDS-DMA of some wave put there, and that no ds_read_b128 lane group hits a
// bank twice).  No HIP types in here.
//
// Geometry.  The MFMA A operand carries weight rows ("features", output columns), the B operand activation rows
// ("tokens"): C^T[feature][token], so that a lane ends up with runs of four consecutive features of one token.
//   workgroup tile   256 features x 256 tokens, K step 64
//   8 waves          fh = wave >> 2 (feature half, 128 each; also the ping-pong group), tq = wave & 3 (token quarter, 64 each)
//   wave tile        128 features x 64 tokens = 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16
//   phases           a K step is consumed in four quadrants: (A half 0, B half 0), (A0, B1), (A1, B1), (A1, B0),
//                    A half = 64 features (two 32-row fragments), B half = 32 tokens (one fragment)
//
// LDS.  One K step of the tile (2 x 256 rows x 128 bytes = 64 KiB) is stored as FOUR 16 KiB half-tiles, each holding
// exactly the rows one phase reads, so that a half-tile can be consumed as soon as it has landed:
//   h = 0: A half 0 of both feature halves     (read in phase 0)
//   h = 1: B half 0 of the four token quarters (read in phase 0, kept for phase 3)
//   h = 2: B half 1                            (read in phase 1)
//   h = 3: A half 1                            (read in phase 2)
// A half-tile is 128 "local rows" of 128 bytes (64 bf16 along K), in blocks of 8 rows = 1 KiB = one
// global_load_lds_dwordx4 wave instruction (lane l lands at byte 16 l of the block): 8 lanes fetch one whole
// 128-byte line of a row.  The 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7, applied to the SOURCE
// address (the DMA destination is lane-linear) and to the fragment reads alike.
#pragma once

#ifndef MI_HD
#ifdef __HIPCC__
#define MI_HD __host__ __device__
#else
#define MI_HD
#endif
#endif

namespace mi {
namespace gt {

constexpr int TILE_F = 256, TILE_T = 256, BK = 64;
constexpr int HALF_BYTES = 16384, SLOTS = 8;         // two K steps x four half-tiles
constexpr int STAGE_BYTES = 4096;  // per wave: 32 output rows x 128 bytes on their way to memory as whole cache lines
constexpr int LDS_BYTES = SLOTS * HALF_BYTES + 8 * STAGE_BYTES;  // 160 KiB: everything a CU has

// row of the workgroup tile (feature row for h = 0 / 3, token row for h = 1 / 2) held by local row lr of half-tile h
MI_HD constexpr int tile_row(int h, int lr) {
  return (h == 0 || h == 3) ? (lr >> 6) * 128 + (h == 3 ? 64 : 0) + (lr & 63)
                            : (lr >> 5) * 64 + (h == 2 ? 32 : 0) + (lr & 31);
}
MI_HD constexpr bool is_weight_half(int h) { return h == 0 || h == 3; }
MI_HD constexpr int swizzle(int lr) { return (lr >> 1) & 7; }
// byte offset of 16-byte K chunk c (0..7) of local row lr inside a half-tile
MI_HD constexpr int half_off(int lr, int c) { return (lr >> 3) * 1024 + (lr & 7) * 128 + ((c ^ swizzle(lr)) << 4); }

// ---- LDS-DMA side: wave `wave`, instruction i (0, 1) of a half-tile, lane `lane` ----
MI_HD constexpr int dma_block(int wave, int i) { return wave * 2 + i; }                  // 1 KiB block of the half-tile
MI_HD constexpr int dma_local_row(int wave, int i, int lane) { return dma_block(wave, i) * 8 + (lane >> 3); }
// the K chunk this lane must fetch so that its 16 bytes, landing at 16 * lane, are where half_off() expects them
MI_HD constexpr int dma_chunk(int wave, int i, int lane) { return (lane & 7) ^ swizzle(dma_local_row(wave, i, lane)); }

// ---- fragment side: wave (fh, tq), lane = 32 hi + l31 ----
// A fragment `a` (0, 1) of A half `ah`, k step kk (0..3 of 16): row l31 of the fragment, K chunk 2 kk + hi
MI_HD constexpr int a_half(int ah) { return ah ? 3 : 0; }
MI_HD constexpr int a_local_row(int fh, int a, int l31) { return fh * 64 + a * 32 + l31; }
MI_HD constexpr int b_half(int bh) { return bh ? 2 : 1; }
MI_HD constexpr int b_local_row(int tq, int l31) { return tq * 32 + l31; }
MI_HD constexpr int frag_chunk(int kk, int hi) { return 2 * kk + hi; }

// ---- accumulator side: acc[ah * 2 + a][bh], register r (0..15) of lane (hi, l31) ----
MI_HD constexpr int acc_feature(int fh, int ah, int a, int r, int hi) {
  return fh * 128 + ah * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
}
MI_HD constexpr int acc_token(int tq, int bh, int l31) { return tq * 64 + bh * 32 + l31; }

// ---- the 128 x 128 kernel (gemm_mid_kernel): four waves, fw = wave >> 1 (64 features), tw = wave & 1 (64 tokens);
// a K step = one A half-tile (local row = tile feature row) + one B half-tile (local row = tile token row) ----
constexpr int MID_F = 128, MID_T = 128;
MI_HD constexpr int mid_dma_block(int wave, int i) { return wave * 4 + i; }  // i = 0..3: 1 KiB block of either half-tile
MI_HD constexpr int mid_dma_local_row(int wave, int i, int lane) { return mid_dma_block(wave, i) * 8 + (lane >> 3); }
MI_HD constexpr int mid_dma_chunk(int wave, int i, int lane) { return (lane & 7) ^ swizzle(mid_dma_local_row(wave, i, lane)); }
// weight row of local row lr of the A half-tile of the tile at feature n0: plain, or SwiGLU (each wave's first fragment
// = 32 gate rows, its second the 32 up rows that pair with them; n0 counts gate + up rows, n0 / 2 output columns)
MI_HD constexpr int mid_weight_row(int lr, int n0, int N, bool silu) {
  return silu ? ((lr >> 5) & 1) * (N >> 1) + (n0 >> 1) + (lr >> 6) * 32 + (lr & 31) : n0 + lr;
}
MI_HD constexpr int mid_a_local_row(int fw, int f, int l31) { return fw * 64 + f * 32 + l31; }
MI_HD constexpr int mid_b_local_row(int tw, int b, int l31) { return tw * 64 + b * 32 + l31; }
// accumulator acc[f][b], register r of lane (hi, l31): feature row inside the tile / token inside the tile
MI_HD constexpr int mid_acc_feature(int fw, int f, int r, int hi) { return fw * 64 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }
MI_HD constexpr int mid_acc_token(int tw, int b, int l31) { return tw * 64 + b * 32 + l31; }

// ---- the four-wave 256 x 256 x 64 kernel (gemm_w4_kernel, the product kernel for large M since round 4) ----
// One wave per SIMD: fw = wave >> 1 (128 features), tw = wave & 1 (128 tokens), wave tile 128 x 128 = 4 x 4 accumulators.
// One K step of the tile in LDS is ONE image of 64 PIECES: a piece is what one `buffer_load_dwordx4 ... lds` wave
// instruction lands - 1 KiB = eight 128-byte row slices (lane l -> slice l >> 3, 16-byte chunk l & 7: the eight lanes
// of a slice fetch one whole line of a source row IN ORDER, which is what the address coalescer wants; a chunk
// permutation inside the slice - the XOR swizzle of the kernels above - costs 3-5 % of the GEMM) - followed by 16
// bytes of padding.  Pieces 0..31 hold the A region (256 feature rows), 32..63 the B region (256 token rows); wave w
// moves pieces 16 w .. 16 w + 15, i.e. the 128 rows of one half of one region.
// Which row goes where: row r7 = (b1 b0 x4 x3 x2 x1 x0) of a 128-row half lives in piece q = (b0 x3 x1 x0), slice
// s = (b1 x2 x4).  A fragment read (row l31 of a 32-row block, one chunk for all 16 lanes of a ds_read_b128 lane
// group) then finds its 16 rows in 16 different 16-byte bank slots: slot = (piece + 8 (s & 1) + chunk) mod 16 - the
// padding shifts every piece by one slot - and the two lane-group row sets {x4 ^ x3 ^ x2 = 0 / 1} are each in
// bijection with (x4, x3, x1, x0).  Both maps are additive, row = w4_piece_rows(q) + w4_slice_rows(s): a piece's
// source offset is a per-lane constant plus a scalar.
constexpr int W4_PIECE_BYTES = 1024 + 16, W4_PIECES = 16;  // (W4_PIECES: per wave)
constexpr int W4_STEP_BYTES = 64 * W4_PIECE_BYTES;         // 66 560
constexpr int W4_STAGE_BYTES = 4096;  // per wave: 32 tokens x 64 output features on their way to memory as whole lines
MI_HD constexpr bool w4_wave_is_weight(int wave) { return wave < 2; }
MI_HD constexpr int w4_piece_of_row(int r7) { return 8 * ((r7 >> 5) & 1) + 4 * ((r7 >> 3) & 1) + (r7 & 3); }
MI_HD constexpr int w4_slice_of_row(int r7) { return ((r7 >> 4) & 1) + 2 * ((r7 >> 2) & 1) + 4 * ((r7 >> 6) & 1); }
MI_HD constexpr int w4_piece_rows(int q) { return 32 * ((q >> 3) & 1) + 8 * ((q >> 2) & 1) + (q & 3); }
MI_HD constexpr int w4_slice_rows(int s) { return 16 * (s & 1) + 4 * ((s >> 1) & 1) + 64 * ((s >> 2) & 1); }
// byte offset inside the image of 16-byte chunk c of row r (0..255) of region (0: A, 1: B)
MI_HD constexpr int w4_row_off(int region, int r, int c) {
  return (region * 32 + (r >> 7) * 16 + w4_piece_of_row(r & 127)) * W4_PIECE_BYTES + w4_slice_of_row(r & 127) * 128 + c * 16;
}
// feed: row (inside its region) and chunk fetched by lane `lane` of piece q of wave `wave`; lands at byte 16 * lane of the piece
MI_HD constexpr int w4_dma_row(int wave, int q, int lane) { return (wave & 1) * 128 + w4_piece_rows(q) + w4_slice_rows(lane >> 3); }
MI_HD constexpr int w4_dma_chunk(int lane) { return lane & 7; }
MI_HD constexpr int w4_piece_off(int wave, int q) { return (wave * 16 + q) * W4_PIECE_BYTES; }  // inside the image
// weight row of A-region row r for the tile at feature n0: plain, or SwiGLU (each wave's 128 rows = 64 gate rows, then
// the 64 up rows that pair with them; n0 counts gate + up rows, n0 / 2 output columns)
MI_HD constexpr int w4_weight_row(int r, int n0, int N, bool silu) {
  return silu ? ((r >> 6) & 1) * (N >> 1) + (n0 >> 1) + (r >> 7) * 64 + (r & 63) : n0 + r;
}
// fragment i (0..3) of wave half fw / tw, k group kk: region row, chunk frag_chunk(kk, hi); the kernel's address form:
// per-lane w4_frag_lane(l31, hi) + half * 16 pieces + immediates w4_frag_imm(i, kk)
MI_HD constexpr int w4_a_row(int fw, int i, int l31) { return fw * 128 + i * 32 + l31; }
MI_HD constexpr int w4_b_row(int tw, int j, int l31) { return tw * 128 + j * 32 + l31; }
MI_HD constexpr int w4_frag_lane(int l31, int hi) {
  return (4 * ((l31 >> 3) & 1) + (l31 & 3)) * W4_PIECE_BYTES + (((l31 >> 4) & 1) + 2 * ((l31 >> 2) & 1)) * 128 + hi * 16;
}
MI_HD constexpr int w4_frag_imm(int i, int kk) { return (i & 1) * 8 * W4_PIECE_BYTES + (i >> 1) * 512 + kk * 32; }
// accumulator acc[i][j], register r of lane (hi, l31)
MI_HD constexpr int w4_acc_feature(int fw, int i, int r, int hi) { return fw * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi; }
MI_HD constexpr int w4_acc_token(int tw, int j, int l31) { return tw * 128 + j * 32 + l31; }

// ---- the same four-wave kernel on v_mfma_f32_16x16x32_bf16 (gemm_w16_kernel: the product kernel; the 32 x 32 form above
// remains for the qkv-store epilogue) ----
// Wave tile 128 x 128 = 8 x 8 accumulators of 16 x 16; a K step is two k halves of 32.  A fragment read: lane
// (r16 = lane & 15, q = lane >> 4) reads row r16 of a 16-row fragment, 16-byte chunk 4 * khalf + q.  A ds_read_b128 lane
// group then holds 8 rows at chunk c and the OTHER 8 rows at chunk c + 1 ({0-3, 12-15} vs {4-11}), so the image is
// padded by 32 bytes per 1 KiB piece - a piece shifts the 16-byte bank slots by TWO - and row (b2 b1 b0 x3 x2 x1 x0)
// of a 128-row half lives in piece (b0 x3 x1 x0), slice (b2 b1 x3^x2): slot = 8 x2 + 4 x1 + 2 x0 + chunk (mod 16), even
// for the rows read at chunk c, odd for those read at c + 1.  The row of (piece, slice) is
// w16_piece_rows(q) + w16_slice_rows(s, x3 of the piece): additive but for the one bit x2 = (s & 1) ^ x3.
constexpr int W16_PIECE_BYTES = 1024 + 32;
constexpr int W16_STEP_BYTES = 64 * W16_PIECE_BYTES;  // 67 584
MI_HD constexpr int w16_piece_of_row(int r7) { return 8 * ((r7 >> 4) & 1) + 4 * ((r7 >> 3) & 1) + (r7 & 3); }
MI_HD constexpr int w16_slice_of_row(int r7) { return (((r7 >> 3) ^ (r7 >> 2)) & 1) + 2 * ((r7 >> 5) & 1) + 4 * ((r7 >> 6) & 1); }
MI_HD constexpr int w16_piece_x3(int q) { return (q >> 2) & 1; }
MI_HD constexpr int w16_piece_rows(int q) { return 16 * ((q >> 3) & 1) + 8 * ((q >> 2) & 1) + (q & 3); }
MI_HD constexpr int w16_slice_rows(int s, int x3) { return 4 * ((s & 1) ^ x3) + 32 * ((s >> 1) & 1) + 64 * ((s >> 2) & 1); }
MI_HD constexpr int w16_row_off(int region, int r, int c) {
  return (region * 32 + (r >> 7) * 16 + w16_piece_of_row(r & 127)) * W16_PIECE_BYTES + w16_slice_of_row(r & 127) * 128 + c * 16;
}
MI_HD constexpr int w16_dma_row(int wave, int q, int lane) {
  return (wave & 1) * 128 + w16_piece_rows(q) + w16_slice_rows(lane >> 3, w16_piece_x3(q));
}
MI_HD constexpr int w16_piece_off(int wave, int q) { return (wave * 16 + q) * W16_PIECE_BYTES; }
// fragment f (0..7) of wave half fw / tw, k half kh: the kernel's address form = region + half + per-lane + immediate
MI_HD constexpr int w16_frag_row(int half, int f, int r16) { return half * 128 + f * 16 + r16; }
MI_HD constexpr int w16_frag_chunk(int kh, int q) { return 4 * kh + q; }
MI_HD constexpr int w16_frag_lane(int r16, int q) {
  return (4 * ((r16 >> 3) & 1) + (r16 & 3)) * W16_PIECE_BYTES + (((r16 >> 3) ^ (r16 >> 2)) & 1) * 128 + q * 16;
}
MI_HD constexpr int w16_frag_imm(int f, int kh) { return (f & 1) * 8 * W16_PIECE_BYTES + ((f >> 1) & 1) * 256 + (f >> 2) * 512 + kh * 64; }
// SwiGLU: rows 0..63 of a wave's half are gate rows, 64..127 the up rows that pair with them (fragments f and f + 4)
MI_HD constexpr int w16_weight_row(int r, int n0, int N, bool silu) { return w4_weight_row(r, n0, N, silu); }
// accumulator acc[f][t], register e (0..3) of lane (n = lane & 15, q = lane >> 4): feature 16 f + 4 q + e, token 16 t + n
MI_HD constexpr int w16_acc_feature(int fw, int f, int e, int q) { return fw * 128 + f * 16 + 4 * q + e; }
MI_HD constexpr int w16_acc_token(int tw, int t, int n) { return tw * 128 + t * 16 + n; }

}  // namespace gt
}  // namespace mi
