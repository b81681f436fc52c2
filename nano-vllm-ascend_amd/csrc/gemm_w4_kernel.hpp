// The 256 x 256 x 64 tile GEMM with ONE wave per SIMD (four waves, 512 registers each, wave tile 128 x 128): the
// product kernel for the compute-bound shapes since round 4.  Included by gemm_tile.hip (TileArgs, TEPI_*, namespace mi).
//
// Why this shape.  The chip is power-limited under bf16 MFMA load (1.6-1.75 GHz): what a schedule buys is issue
// cycles per K step, and tools/ubench/gemm_feed.hip measures them for a loop of exactly this form - 2165 cycles per K
// step of 64 MFMAs (2048 of matrix pipe) against 2785 for the same work with the fragment reads one k group ahead and
// ~2700 for the eight-wave ping-pong kernel above (its two waves per SIMD read 1.5 x the LDS bytes and meet at four
// barriers per step).  Three rules make the difference (profiles/r04_gemm_feed_probe.txt):
//   * fragments are read TWO k groups ahead into four register sets (one per k group of a step): by the end of group 1
//     the step's LDS image has been read completely, so ONE wait + barrier per K step - "my reads of this image are
//     back, my pieces of the next step have landed" - both releases the image for step t + 2 and opens the other one;
//   * the feed is `buffer_load_dwordx4 ... lds` with scalar offsets (two SALU per piece, no vector address arithmetic),
//     ONE piece behind every second MFMA of groups 2 and 3: a piece holds its wave's issue for ~60 cycles, an MFMA
//     occupies the pipe for 32, so one per two MFMAs hides and one per MFMA does not (2573 cycles);
//   * a piece then has a whole K step (~2000 cycles) to land before the wait that covers it.
// Rows past M / N are never clamped: they lie beyond the buffer descriptor's range and arrive as zeros.
// LDS image: 64 padded 1 KiB pieces per K step (gemm_tile_index.hpp) - the lanes of a piece fetch whole source lines in
// order, and the padding, not a chunk permutation, keeps the fragment reads off each other's banks.
//
// Persistent like the kernel above: workgroup b computes tiles b, b + grid, ...; the piece stream runs on across tile
// seams (the last two K steps of a tile fetch steps 0 and 1 of the next), a tile's stores overlap the next tile's
// first pieces, and the first wait of the next tile (vmcnt(0), as every wait here) drains them.
// Summation order: one fp32 MFMA chain over K per output element, k ascending - the same bits as every other GEMM here.
#pragma once

template <int EPI, bool BIAS, int V>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(const TileArgs a) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * W4_STEP_BYTES + ((V & 32768) ? 4 * W4_STAGE_BYTES : 0)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fw = wave >> 1, tw = wave & 1, hi = lane >> 5, l31 = lane & 31;
  const int KT_all = a.K / BK;
  const int KT = EPI == TEPI_PARTIAL ? KT_all / (int)gridDim.y : KT_all;
  const uint32_t kbeg_bytes = EPI == TEPI_PARTIAL ? (uint32_t)blockIdx.y * KT * (BK * 2) : 0u;
  const int ntiles = a.tiles_f * a.tiles_t, stride = (int)gridDim.x;

  // ---- the feed: waves 0, 1 move the A region (weight rows), waves 2, 3 the B region (activation rows) ----
  const bool wgt = w4_wave_is_weight(wave);
  const uint32_t ld2 = (uint32_t)(wgt ? a.K : a.ldx) * 2;  // bytes per source row
  // The descriptor covers whole rows only: a lane whose row is past the operand's last one (edge tiles) is out of
  // range and its 16 bytes arrive as zeros.  The range check is made on the VECTOR offset (whether the scalar offset
  // takes part in it differs between ISA generations), so the row goes into the vector offset and only the K step
  // into the scalar one.
  const uint32_t records = (uint32_t)(wgt ? a.N : a.M) * ld2;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wgt ? a.w : a.x), 0, records, 0x00020000);
  // per-lane part of a piece's source offset: row slice lane >> 3 of the piece (gemm_tile_index.hpp: which rows those
  // are), chunk lane & 7 - the eight lanes of a slice fetch one 128-byte line in order.  SwiGLU weights: a wave's 128
  // region rows are 64 gate rows, then the 64 up rows N / 2 further on.
  const int slice_rows = w4_slice_rows(lane >> 3);
  uint32_t voff = (uint32_t)(lane & 7) * 16 +
                  (wgt && EPI == TEPI_SILU ? (uint32_t)((slice_rows & 63) + (slice_rows >> 6) * (a.N >> 1)) : (uint32_t)slice_rows) * ld2;
  if (V & 1024) voff = (uint32_t)lane * 16;  // ablation (wrong results): every piece from the operand's first 64 KiB
  // scalar part: first source row (as a byte offset) of this wave's 128 region rows for the tile being FETCHED, and
  // the K offset of the step being fetched
  uint32_t f_row, f_k = kbeg_bytes;
  int f_kt = 0, f_blk = (int)blockIdx.x;
  auto fetch_tile = [&](int b) __attribute__((always_inline)) {
    // past the workgroup's last tile every row is beyond the descriptor's range: the pieces of the stream's last two
    // steps fetch nothing and complete at once (zeros land in the image nobody reads)
    if (b >= ntiles) {
      f_row = 0xF0000000u;
      return;
    }
    const int tid = tile_of_block(b, a.tiles_t, a.tiles_f, (V & 4) != 0);
    const int m0_ = (tid / a.tiles_f) * TILE_T, n0_ = (tid % a.tiles_f) * TILE_F;
    if (!wgt) f_row = (uint32_t)(m0_ + (wave & 1) * 128) * ld2;
    else if (EPI == TEPI_SILU) f_row = (uint32_t)((n0_ >> 1) + (wave & 1) * 64) * ld2;
    else f_row = (uint32_t)(n0_ + (wave & 1) * 128) * ld2;
  };
  fetch_tile(f_blk);
  char* const pieces = lds + wave * (W4_PIECES * W4_PIECE_BYTES);
  int wbuf = 0;  // image the next step's pieces go to
  auto piece = [&](int q) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(pieces + wbuf * W4_STEP_BYTES + q * W4_PIECE_BYTES), 16,
                                             (V & 1024) ? voff + (uint32_t)(wave * 16 + q) * 1024 : voff + (f_row + (uint32_t)w4_piece_rows(q) * ld2),
                                             (V & 1024) ? 0u : f_k, 0, 0);
  };
  // behind the sixteenth piece of a step: on to the next K step of the fetched tile, or to step 0 of the next tile
  auto fetch_advance = [&]() __attribute__((always_inline)) {
    wbuf ^= 1;
    f_k += BK * 2;
    if (++f_kt == KT) {
      f_kt = 0;
      f_k = kbeg_bytes;
      f_blk += stride;
      fetch_tile(f_blk);
    }
  };

  // ---- fragment reads: ONE per-lane byte offset per operand into the image being read; fragment and k group are
  // immediates (w4_frag_imm) ----
  int offa = w4_frag_lane(l31, hi) + fw * (16 * W4_PIECE_BYTES);
  int offb = w4_frag_lane(l31, hi) + (32 + tw * 16) * W4_PIECE_BYTES;
  int flip = W4_STEP_BYTES;  // to the other image and back
  u32x4 Ra[4][4], Rb[4][4];  // [k group][fragment]
  f32x16 acc[4][4];          // [feature block i][token block j]
  auto read1 = [&](int kk, int q) __attribute__((always_inline)) {  // q-th of the eight fragment reads of k group kk
    if (q < 4) Ra[kk][q] = *reinterpret_cast<const u32x4*>(lds + w4_frag_imm(q, kk) + offa);
    else Rb[kk][q - 4] = *reinterpret_cast<const u32x4*>(lds + w4_frag_imm(q - 4, kk) + offb);
  };
  auto mma1 = [&](int kk, int m, auto first) __attribute__((always_inline)) {
    const int i = m >> 2, j = m & 3;
    if (decltype(first)::value) {  // a tile's first MFMA into this accumulator: C = 0
      f32x16 z;
#pragma unroll
      for (int r = 0; r < 16; ++r) z[r] = 0.f;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(Ra[kk][i]), as_frag(Rb[kk][j]), z, 0, 0, 0);
    } else {
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(Ra[kk][i]), as_frag(Rb[kk][j]), acc[i][j], 0, 0, 0);
    }
  };

  // One K step.  FIRST: step 0 of a tile (group 0 starts the accumulators).  LAST (the qkv-store kernel only): the
  // tile's last step - the k groups 0, 1 of the NEXT tile's first step are not read here but behind the epilogue
  // (read_first_groups): 64 registers its K / V branches need, for ~300 cycles of exposed LDS latency per tile
  // (measured on the plain kernels: 3-5 % of the GEMM - they keep the prefetch).
  constexpr bool kRereadAfterEpilogue = EPI == TEPI_QKV;
  auto kstep = [&](auto first, auto last) __attribute__((always_inline)) {
    using First = decltype(first);
    constexpr bool LAST = kRereadAfterEpilogue && decltype(last)::value;
    // groups 0, 1: MFMAs of k group g, reads of k group g + 2 of the same image
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (g == 0) mma1(g, 2 * q, First{});
        else mma1(g, 2 * q, std::false_type{});
        read1(g + 2, q);
        if (g == 0) mma1(g, 2 * q + 1, First{});
        else mma1(g, 2 * q + 1, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // my reads of this image are back, my pieces of the next step have landed (and a previous tile's stores are out)
    if (V & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else if (V & 256) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ablation: pieces never waited for
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    offa += flip;  // the other image from here on
    offb += flip;
    flip = -flip;
    // groups 2, 3: MFMAs of k group g, reads of k group g - 2 of the NEXT step, the pieces of the step after it
#pragma unroll
    for (int g = 2; g < 4; ++g) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        mma1(g, 2 * q, std::false_type{});
        if (!(V & 64) && !LAST) read1(g - 2, q);
        __builtin_amdgcn_sched_barrier(0);
        mma1(g, 2 * q + 1, std::false_type{});
        if (!(V & 32)) piece((g - 2) * 8 + q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    fetch_advance();
  };

  // ---- epilogue of the tile at (m0, n0): lane (hi, l31) holds token w4_acc_token(tw, j, l31) and features
  // 8 rq + 4 hi + {0..3} of feature block i; two register quads are exchanged between the half-waves
  // (v_permlane32_swap) so that every lane stores 8 consecutive features = 16 bytes.  Measured and not kept
  // (profiles/r04_gemm_w4.txt): rows through a per-wave LDS stage and out as whole 128-byte lines (145 vs 126 us on
  // qkv: the stage's write -> read -> store chain is latency the direct form does not have), non-temporal /
  // write-through stores (179 / 211 / 141 us: 32-byte pieces of a line must merge in the L2) ----
  const int n_out = EPI == TEPI_SILU ? a.N >> 1 : a.N;
  const bool wide = !(V & 8) && n_out % 8 == 0 && a.ldy % 8 == 0;
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    if (V & 512) {  // ablation: no output (the accumulators are kept alive by an empty asm)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int tok = m0 + w4_acc_token(tw, j, l31);
      const bool tok_ok = tok < a.M;
      if (EPI == TEPI_PARTIAL) {  // fp32 sums of this K slice: [slice][token][feature]
        float* prow = a.part + ((int64_t)blockIdx.y * a.M + min(tok, a.M - 1)) * a.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int col = n0 + w4_acc_feature(fw, i, 4 * rq, hi);
            if (tok_ok && col < a.N)
              *reinterpret_cast<f32x4*>(prow + col) =
                  f32x4{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
          }
        continue;
      }
      uint16_t* yrow = a.y + (int64_t)min(tok, a.M - 1) * a.ldy;
#pragma unroll
      for (int i = 0; i < (EPI == TEPI_SILU ? 2 : 4); ++i) {
        const int col0 = EPI == TEPI_SILU ? (n0 >> 1) + fw * 64 + i * 32 : n0 + fw * 128 + i * 32;
        u32x2 pk[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int col = col0 + 8 * rq + 4 * hi;
          float o[4];
          if (EPI == TEPI_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // the roundings and the silu form of gemm_tile_kernel's epilogue
              const float gb = rbf(acc[i][j][4 * rq + e]);
              const float sb = rbf(gb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb * -1.4426950408889634f)));
              o[e] = sb * rbf(acc[i + 2][j][4 * rq + e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * rq + e];
            if (BIAS && col < n_out) {
              const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + col);
              o[0] += lo_bf(bw[0]);
              o[1] += hi_bf(bw[0]);
              o[2] += lo_bf(bw[1]);
              o[3] += hi_bf(bw[1]);
            }
          }
          pk[rq] = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
        }
        if (wide) {
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
            const u32x4 v = {sx[0], sy[0], sx[1], sy[1]};
            const int col = col0 + 8 * (2 * p2 + hi);
            if (tok_ok && col < n_out) *reinterpret_cast<u32x4*>(yrow + col) = v;
          }
        } else {
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            const int col = col0 + 8 * rq + 4 * hi;
            if (tok_ok && col < n_out) *reinterpret_cast<u32x2*>(yrow + col) = pk[rq];
          }
        }
      }
    }
  };


  // ---- the same epilogue with the rows leaving as WHOLE cache lines (V & 32768; the host guarantees full feature
  // tiles and 16-byte aligned rows).  Straight from the accumulator layout a store instruction writes 32 bytes of each
  // of 32 rows, and the CU's store path handles one line per cycle: 32 cycles for 1 KiB.  Through a per-wave LDS stage
  // (32 tokens x 64 output features = 128-byte rows, 16-byte chunks XOR-swizzled by the row) an instruction covers
  // eight lanes per row = 8 full lines.  The stage is written and read with inline assembly: for a C++ access to LDS
  // the compiler first drains the VMEM queue (the pieces in flight might alias it, for all it knows), and with it every
  // earlier store of the epilogue.  LDS operations of one wave execute in order: no barrier; the reads of one slab
  // are in flight while the next one is converted.
  // A slab = token block j x feature blocks (2 u, 2 u + 1); plain: u = 0, 1 (the wave's 128 features), SwiGLU: u = 0
  // (its 64 outputs: gate block i, up block i + 2).
  constexpr int NU = EPI == TEPI_SILU ? 1 : 2, NSLAB = 4 * NU;
  const uint32_t stg_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + 2 * W4_STEP_BYTES) + wave * W4_STAGE_BYTES;
  const uint32_t stg_w = stg_base + (uint32_t)l31 * 128 + (uint32_t)(((l31 & 7) ^ hi) << 4);  // ^ (4 i' + 2 p2) << 4 per chunk
  const uint32_t stg_r = stg_base + (uint32_t)(lane >> 3) * 128 + (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);  // + 8 rows per read
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (uint32_t)a.M * (uint32_t)a.ldy * 2, 0x00020000);
  const uint32_t y_lane = (uint32_t)(lane >> 3) * (uint32_t)a.ldy * 2 + (uint32_t)(lane & 7) * 16;
  auto epilogue_lines = [&](int m0, int n0) __attribute__((always_inline)) {
    u32x4 out[4], back[4];
    // (opaque per tile: otherwise the chunk addresses are hoisted out of the tile loop and live - spilled - through
    // the whole K loop)
    uint32_t w_addr = stg_w;
    asm volatile("" : "+v"(w_addr));
    auto convert = [&](int slab) __attribute__((always_inline)) {
      const int j = slab / NU, u = slab % NU;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * u + ii;
        u32x2 pk[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float o[4];
          if (EPI == TEPI_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float gb = rbf(acc[i][j][4 * rq + e]);
              const float sb = rbf(gb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb * -1.4426950408889634f)));
              o[e] = sb * rbf(acc[i + 2][j][4 * rq + e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[i][j][4 * rq + e];
            if (BIAS) {
              const u32x2 bw = *reinterpret_cast<const u32x2*>(a.bias + n0 + fw * 128 + i * 32 + 8 * rq + 4 * hi);
              o[0] += lo_bf(bw[0]);
              o[1] += hi_bf(bw[0]);
              o[2] += lo_bf(bw[1]);
              o[3] += hi_bf(bw[1]);
            }
          }
          pk[rq] = u32x2{pack_bf(o[0], o[1]), pack_bf(o[2], o[3])};
        }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
          out[2 * ii + p2] = u32x4{sx[0], sy[0], sx[1], sy[1]};  // features 32 ii + 16 p2 + 8 hi .. + 7 of the slab, token l31
        }
      }
    };
    auto stage_in = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int ch = 0; ch < 4; ++ch)  // chunk 2 ch + hi of row l31
        asm volatile("ds_write_b128 %0, %1" ::"v"(w_addr ^ (uint32_t)(ch << 5)), "v"(out[ch]) : "memory");
    };
    auto stage_out = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < 4; ++it)  // rows 8 it + (lane >> 3): (row & 7) = lane >> 3 for every it
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(back[it]) : "v"(stg_r), "n"(it * 1024) : "memory");
    };
    auto store_lines = [&](int slab) __attribute__((always_inline)) {
      const int j = slab / NU, u = slab % NU;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const uint32_t colb = (uint32_t)(EPI == TEPI_SILU ? (n0 >> 1) + fw * 64 : n0 + fw * 128 + u * 64) * 2;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t rowb = (uint32_t)(m0 + tw * 128 + j * 32 + it * 8) * (uint32_t)a.ldy * 2 + colb;
        // (rows past M are beyond the descriptor's range: dropped)
        __builtin_amdgcn_raw_buffer_store_b128(back[it], yrsrc, y_lane + rowb, 0, 0);
      }
    };
    convert(0);
    stage_in();
    stage_out();
#pragma unroll
    for (int slab = 1; slab < NSLAB; ++slab) {
      convert(slab);
      store_lines(slab - 1);
      stage_in();
      stage_out();
    }
    store_lines(NSLAB - 1);
  };

  // ---- TEPI_QKV: the packed qkv projection whose K and V heads never reach the qkv rows (VERDICT r03 item 1b;
  // qwen3.py:79-90 + attention.py:55-58 of the reference: split, k-norm, RoPE, store_kvcache).  head_dim 128: a wave's
  // 128 features are exactly ONE head (q, k or v - wave-uniform).  Q heads: the ordinary whole-line stores (the
  // attention kernel norms and rotates q in its operand load).  K heads: after the epilogue's half-wave exchange the
  // lane pair (hi = 0 / 1) of a token holds the head as 8-dim groups 2 kk + hi - the distribution of
  // head_rmsnorm_rope_q32 (kv_store.hpp), which is bit-identical to the 8-lanes-per-head form of mi_qknorm_rope_store;
  // its 16-byte groups are the cache tile's own chunks: stored straight from the registers, no staging.  V heads: the
  // cache tile is token-transposed (4 tokens x {d, d + 16} per 16-byte chunk): the (d, d + 16) pairs of a token go
  // through the wave's LDS stage as 32-bit words and come back as chunks; four tokens whose slots are not four
  // consecutive ones of a tile (sequence seams, skipped tokens) are scattered element by element.
  auto epilogue_qkv = [&](int m0, int n0) __attribute__((always_inline)) {
    const int head = (n0 >> 7) + fw, nq = a.qkv.n_q_heads, nkv = a.qkv.n_kv_heads, bs = a.qkv.block_size;
    if (head < nq) {
      epilogue_lines(m0, n0);
      return;
    }
    if (head >= nq + 2 * nkv) return;
    const bool is_v = head >= nq + nkv;
    const int hk = head - nq - (is_v ? nkv : 0), tpb = bs >> 4;
    // Everything lane-dependent below is derived from an OPAQUE copy of the lane id: computed here, per tile.  Derived
    // from `lane` the compiler hoists dozens of such values out of the tile loop, they live through the K loop, and
    // the register file of the loop (256 accumulators + 128 fragment registers) starts spilling accumulators.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int hi = lane_e >> 5, l31 = lane_e & 31;
    const uint32_t stage = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + 2 * W4_STEP_BYTES) + wave * W4_STAGE_BYTES;
    auto value = [&](int i, int j, int r) __attribute__((always_inline)) -> float {  // the GEMM output element, un-rounded
      float o = acc[i][j][r];
      if (BIAS) o += bf2f(a.bias[n0 + fw * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi]);
      return o;
    };
    if (!is_v) {
      // The table rows are two dependent global loads away (position, then its cos / sin row) with ONE wave per SIMD to
      // hide them: positions and slots of all four token blocks are requested first, the row of block j + 1 as soon
      // as block j has been rotated.
      int slot[4];
      const float* cs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int tok = m0 + w4_acc_token(tw, j, l31);
        const int tk = tok < a.M ? tok : a.M - 1;
        slot[j] = tok < a.M ? a.qkv.slots[tk] : -1;
        cs[j] = a.qkv.cos_sin + a.qkv.positions[tk] * 128;
      }
      RopeRegs32 rope;  // (ONE set of 64 registers)
      rope_regs_q32_load(rope, cs[0], hi);

#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x4 xp[8];  // the head of token l31 as packed bf16: xp[kk] = dims 16 kk + 8 hi .. + 7
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u32x2 pk[4];
#pragma unroll
          for (int rq = 0; rq < 4; ++rq)
            pk[rq] = u32x2{pack_bf(value(i, j, 4 * rq), value(i, j, 4 * rq + 1)), pack_bf(value(i, j, 4 * rq + 2), value(i, j, 4 * rq + 3))};
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * p2][0], pk[2 * p2 + 1][0], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * p2][1], pk[2 * p2 + 1][1], false, false);
            xp[2 * i + p2] = u32x4{sx[0], sy[0], sx[1], sy[1]};  // dims 32 i + 16 p2 + 8 hi .. + 7 = group 2 (2 i + p2) + hi
          }
        }
        head_rmsnorm_rope_q32_packed(xp, a.qkv.k_w, rope, hi, a.qkv.eps);
        __builtin_amdgcn_sched_barrier(0);
        if (j < 3) rope_regs_q32_load(rope, cs[j + 1], hi);  // under this block's stores and the next one's conversion
        __builtin_amdgcn_sched_barrier(0);
        if (slot[j] >= 0) {
          const int blk = slot[j] / bs, off = slot[j] - blk * bs;
          uint16_t* tile = a.qkv.k_cache + kv_tile_base(blk, hk, off, nkv, tpb);
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) *reinterpret_cast<u32x4*>(tile + k_tile_off(off & 15, 8 * (2 * kk + hi))) = xp[kk];
        }
        __builtin_amdgcn_sched_barrier(0);  // (one token block at a time: interleaved blocks need the registers twice)
      }
      return;
    }
    // V: words (d, d + 16) of token l31 into the stage at [32-dim block][token group l31 >> 2][(d & 15) ^ group][l31 & 3]
    const int tg = l31 >> 2;
    uint32_t w_e[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w_e[e] = stage + (uint32_t)((tg * 16 + ((4 * hi + e) ^ tg)) * 16 + (l31 & 3) * 4);
    const uint32_t r_addr = stage + (uint32_t)lane_e * 16;
    // the slots of the token quadruples this lane_e will store (chunk it * 64 + lane_e of a stage image: token group
    // tgp = (it & 1) * 4 + (lane_e >> 4) of block j), requested up front: one round trip instead of one per image
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    i32x4 sl4[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int tok0 = m0 + tw * 128 + j * 32 + 4 * (h2 * 4 + (lane_e >> 4));
        if (tok0 + 3 < a.M) {
          sl4[j][h2] = *reinterpret_cast<const i32x4*>(a.qkv.slots + tok0);
        } else {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) sl4[j][h2][tt] = tok0 + tt < a.M ? a.qkv.slots[tok0 + tt] : -1;
        }
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {  // the 32-dim blocks 2 u, 2 u + 1 of the head: 256 chunks = the 4 KiB stage
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int rq = 0; rq < 2; ++rq)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              // dims d = 32 (2 u + ii) + 8 rq + 4 hi + e and d + 16, rounded to bf16 as the qkv rows would hold them
              const uint32_t wv = pack_bf(value(2 * u + ii, j, 4 * rq + e), value(2 * u + ii, j, 4 * (rq + 2) + e));
              if (ii == 0 && rq == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(w_e[e]), "v"(wv) : "memory");
              else if (ii == 0) asm volatile("ds_write_b32 %0, %1 offset:128" ::"v"(w_e[e]), "v"(wv) : "memory");
              else if (rq == 0) asm volatile("ds_write_b32 %0, %1 offset:2048" ::"v"(w_e[e]), "v"(wv) : "memory");
              else asm volatile("ds_write_b32 %0, %1 offset:2176" ::"v"(w_e[e]), "v"(wv) : "memory");
            }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 back[4];
        asm volatile("ds_read_b128 %0, %1" : "=v"(back[0]) : "v"(r_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(back[1]) : "v"(r_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(back[2]) : "v"(r_addr) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(back[3]) : "v"(r_addr) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(back[0]), "+v"(back[1]), "+v"(back[2]), "+v"(back[3])::"memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          // chunk it * 64 + lane_e of the stage: 32-dim block it >> 1, token group tgp, stored position lane_e & 15
          const int tgp = (it & 1) * 4 + (lane_e >> 4), d15 = (lane_e & 15) ^ tgp;
          const int d = (2 * u + (it >> 1)) * 32 + d15;  // the chunk holds dims d and d + 16 of tokens tok0 .. tok0 + 3
          const int sl[4] = {sl4[j][it & 1][0], sl4[j][it & 1][1], sl4[j][it & 1][2], sl4[j][it & 1][3]};
          const uint32_t w0 = back[it][0], w1 = back[it][1], w2 = back[it][2], w3 = back[it][3];
          if (sl[0] >= 0 && (sl[0] & 3) == 0 && sl[1] == sl[0] + 1 && sl[2] == sl[0] + 2 && sl[3] == sl[0] + 3) {
            const int blk = sl[0] / bs, off = sl[0] - blk * bs;
            uint16_t* tile = a.qkv.v_cache + kv_tile_base(blk, hk, off, nkv, tpb);
            *reinterpret_cast<u32x4*>(tile + v_tile_off(off & 15, d)) =
                u32x4{(w0 & 0xffffu) | (w1 << 16), (w2 & 0xffffu) | (w3 << 16), (w0 >> 16) | (w1 & 0xffff0000u), (w2 >> 16) | (w3 & 0xffff0000u)};
          } else {
            const uint32_t wt[4] = {w0, w1, w2, w3};
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
              if (sl[tt] >= 0) {
                const int blk = sl[tt] / bs, off = sl[tt] - blk * bs;
                uint16_t* tile = a.qkv.v_cache + kv_tile_base(blk, hk, off, nkv, tpb);
                tile[v_tile_off(off & 15, d)] = (uint16_t)(wt[tt] & 0xffffu);
                tile[v_tile_off(off & 15, d + 16)] = (uint16_t)(wt[tt] >> 16);
              }
          }
        }
      }
    }
  };

  // ---- prologue: steps 0 and 1 of the stream into images 0 and 1; step 0 landed and published; its k groups 0, 1 read ----
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < W4_PIECES; ++i) piece(i);
    fetch_advance();
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  // k groups 0, 1 of the image the fragment offsets point at: the first step of the next tile
  auto read_first_groups = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      read1(0, q);
      read1(1, q);
    }
  };
  read_first_groups();
  // V & 4096 (measurement; round 4 read it with tools/gemm_clock.py, numbers in profiles/r04_gemm_clock.txt): shader cycles and 100 MHz reference ticks of the whole tile loop of
  // every workgroup, written over the first output bytes when everything else is done
  unsigned long long clk0 = 0, ref0 = 0;
  if (V & 4096) {
    clk0 = __builtin_readcyclecounter();
    ref0 = __builtin_amdgcn_s_memrealtime();
  }

  for (int blk = (int)blockIdx.x; blk < ntiles; blk += stride) {
    const int tid = tile_of_block(blk, a.tiles_t, a.tiles_f, (V & 4) != 0);
    const int m0 = (tid / a.tiles_f) * TILE_T, n0 = (tid % a.tiles_f) * TILE_F;
    if constexpr (kRereadAfterEpilogue) {
      if (KT == 1) {
        kstep(std::true_type{}, std::true_type{});
      } else {
        kstep(std::true_type{}, std::false_type{});
        for (int kt = 1; kt < KT - 1; ++kt) kstep(std::false_type{}, std::false_type{});
        kstep(std::false_type{}, std::true_type{});
      }
    } else {
      kstep(std::true_type{}, std::false_type{});
      for (int kt = 1; kt < KT; ++kt) kstep(std::false_type{}, std::false_type{});
    }
    __builtin_amdgcn_sched_barrier(0);
    // A workgroup's last tile: its (empty) past-the-end pieces must not outlive the workgroup's LDS - waited for HERE,
    // not behind the stores: the waves end with their stores in flight
    if (blk + stride >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (EPI == TEPI_QKV) epilogue_qkv(m0, n0);
    else if ((V & 32768) && EPI != TEPI_PARTIAL) epilogue_lines(m0, n0);
    else epilogue(m0, n0);
    __builtin_amdgcn_sched_barrier(0);
    if (kRereadAfterEpilogue && blk + stride < ntiles) read_first_groups();
  }
  if (V & 4096) {
    const unsigned long long clk1 = __builtin_readcyclecounter(), ref1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) {
      float* o = reinterpret_cast<float*>(a.y) + 2 * blockIdx.x;
      o[0] = (float)(clk1 - clk0);
      o[1] = (float)(ref1 - ref0);
    }
  }
}
