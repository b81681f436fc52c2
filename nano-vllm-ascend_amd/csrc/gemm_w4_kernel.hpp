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

  // One K step.  FIRST: step 0 of a tile (group 0 starts the accumulators).
  auto kstep = [&](auto first) __attribute__((always_inline)) {
    using First = decltype(first);
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
        if (!(V & 64)) read1(g - 2, q);
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
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    read1(0, q);
    read1(1, q);
  }

  for (int blk = (int)blockIdx.x; blk < ntiles; blk += stride) {
    const int tid = tile_of_block(blk, a.tiles_t, a.tiles_f, (V & 4) != 0);
    const int m0 = (tid / a.tiles_f) * TILE_T, n0 = (tid % a.tiles_f) * TILE_F;
    kstep(std::true_type{});
    for (int kt = 1; kt < KT; ++kt) kstep(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    // A workgroup's last tile: its (empty) past-the-end pieces must not outlive the workgroup's LDS - waited for HERE,
    // not behind the stores: the waves end with their stores in flight
    if (blk + stride >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((V & 32768) && EPI != TEPI_PARTIAL) epilogue_lines(m0, n0);
    else epilogue(m0, n0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
