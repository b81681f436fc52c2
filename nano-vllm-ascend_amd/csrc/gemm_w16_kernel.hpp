// The 256 x 256 x 64 tile GEMM with one wave per SIMD on v_mfma_f32_16x16x32_bf16: the product kernel for the
// compute-bound shapes.  Included by gemm_tile.hip (TileArgs, TEPI_*, namespace mi).  Schedule, feed and persistence
// are those of gemm_w4_kernel.hpp (read its header first); what differs is the MFMA shape and what follows from it.
//
// Why 16 x 16 x 32.  The chip is power-limited under this load, and the accumulator traffic of a 32 x 32 x 16 MFMA is
// twice that of a 16 x 16 x 32 per flop (16 + 16 instead of 4 + 4 registers per 32 768 / 16 384 flops): the same loop
// clocks 4-6 % higher on the small shape (tools/ubench/gemm_feed.hip, profiles/r04_gemm_feed_probe.txt: bare MFMAs
// 1.243 vs 1.313 us per K step, the whole loop with the qkv GEMM's fetch pattern 1.356 vs 1.422).
//   * wave tile 128 x 128 = 8 x 8 accumulators of 16 x 16 (f32x4 each: 256 registers); a K step is two k halves of 64
//     MFMAs; the fragments of the next half are read during the current one (two register sets of 8 + 8 fragments,
//     128 registers - the same lead of ~1000 cycles as "two k groups ahead" above); ONE wait + barrier per K step
//     behind the first half; the sixteen pieces one behind every fourth MFMA of the second half;
//   * the MFMAs are inline assembly, in place (`v_mfma ... %0, %1, %2, %0`): with the builtin hipcc gives the result
//     another register quad than the addend and copies 64 quads around at the loop's back edge.  What the compiler
//     then no longer knows is handled here: a tile's first MFMA into an accumulator takes the constant 0 as addend
//     (output-only operand), and the epilogue is fenced from the last MFMAs by s_nop (the matrix pipe's result latency
//     is a software-managed hazard);
//   * LDS image: 64 pieces of 1 KiB + 32 bytes (gemm_tile_index.hpp, w16_*; replayed by tests/test_gemm_tile_index.py);
//   * epilogue: a lane holds features 4 q .. 4 q + 3 of token n of every fragment; v_permlane16_swap between the
//     fragments f and f + 1 gives every lane 8 consecutive features = one 16-byte chunk, which goes through the
//     per-wave LDS stage (32 tokens x 64 features) and out as whole 128-byte lines, as above.
// Summation order: one fp32 MFMA chain over K per output element, k ascending - the same bits as every other GEMM here
// (tests/test_gemm_tile_gpu.py holds the two MFMA shapes against each other and against the library).
#pragma once

template <int EPI, bool BIAS, int V>
__global__ __launch_bounds__(256, 1) void gemm_w16_kernel(const TileArgs a) {
  constexpr bool LINES = (V & 32768) != 0 && EPI != TEPI_PARTIAL;
  __shared__ __attribute__((aligned(1024))) char lds[2 * W16_STEP_BYTES + (LINES ? 4 * W4_STAGE_BYTES : 0)];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fw = wave >> 1, tw = wave & 1, q4 = lane >> 4, n16 = lane & 15;
  const int KT_all = a.K / BK;
  const int KT = EPI == TEPI_PARTIAL ? KT_all / (int)gridDim.y : KT_all;
  const uint32_t kbeg_bytes = EPI == TEPI_PARTIAL ? (uint32_t)blockIdx.y * KT * (BK * 2) : 0u;
  const int ntiles = a.tiles_f * a.tiles_t, stride = (int)gridDim.x;

  // ---- the feed (as gemm_w4_kernel): waves 0, 1 the A region (weight rows), waves 2, 3 the B region ----
  const bool wgt = w4_wave_is_weight(wave);
  const uint32_t ld2 = (uint32_t)(wgt ? a.K : a.ldx) * 2;
  const uint32_t records = (uint32_t)(wgt ? a.N : a.M) * ld2;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(wgt ? a.w : a.x), 0, records, 0x00020000);
  // per-lane part of a piece's source offset: the rows of slice lane >> 3 depend on the piece through ONE bit (x3 of the
  // piece, gemm_tile_index.hpp): two variants.  SwiGLU weights: rows 64..127 of a wave's half are the up rows N / 2 on.
  uint32_t voff[2];
#pragma unroll
  for (int x3 = 0; x3 < 2; ++x3) {
    const int rows = w16_slice_rows(lane >> 3, x3);
    voff[x3] = (uint32_t)(lane & 7) * 16 +
               (wgt && EPI == TEPI_SILU ? (uint32_t)((rows & 63) + (rows >> 6) * (a.N >> 1)) : (uint32_t)rows) * ld2;
  }
  uint32_t f_row, f_k = kbeg_bytes;
  int f_kt = 0, f_blk = (int)blockIdx.x;
  auto fetch_tile = [&](int b) __attribute__((always_inline)) {
    if (b >= ntiles) {  // past the workgroup's last tile: every row beyond the descriptor - the pieces fetch nothing
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
  char* const pieces = lds + wave * (W4_PIECES * W16_PIECE_BYTES);
  int wbuf = 0;
  auto piece = [&](int q) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(pieces + wbuf * W16_STEP_BYTES + q * W16_PIECE_BYTES), 16,
                                             voff[w16_piece_x3(q)] + (f_row + (uint32_t)w16_piece_rows(q) * ld2), f_k, 0, 0);
  };
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

  // ---- fragment reads: one per-lane byte offset per operand; fragment and k half are immediates ----
  int offa = w16_frag_lane(n16, q4) + fw * (16 * W16_PIECE_BYTES);
  int offb = w16_frag_lane(n16, q4) + (32 + tw * 16) * W16_PIECE_BYTES;
  int flip = W16_STEP_BYTES;
  u32x4 Ra[2][8], Rb[2][8];  // [k half][fragment]
  f32x4 acc[8][8];           // [feature fragment f][token fragment t]
  // q-th of the sixteen fragment reads of k half kh, in the order the MFMAs (feature fragment major) first need them:
  // A0, B0 .. B7, A1 .. A7
  auto read1 = [&](int kh, int q) __attribute__((always_inline)) {
    if (q == 0 || q > 8) {
      const int f = q == 0 ? 0 : q - 8;
      Ra[kh][f] = *reinterpret_cast<const u32x4*>(lds + w16_frag_imm(f, kh) + offa);
    } else {
      Rb[kh][q - 1] = *reinterpret_cast<const u32x4*>(lds + w16_frag_imm(q - 1, kh) + offb);
    }
  };
  // (two plain lambdas, not one generic one: inline-assembly operands inside a generic lambda do not capture)
  auto mma_first = [&](int kh, int m) __attribute__((always_inline)) {  // a tile's first MFMA into this accumulator: addend 0
    const int f = m >> 3, t = m & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[f][t]) : "v"(Ra[kh][f]), "v"(Rb[kh][t]));
  };
  auto mma_acc = [&](int kh, int m) __attribute__((always_inline)) {
    const int f = m >> 3, t = m & 7;
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[f][t]) : "v"(Ra[kh][f]), "v"(Rb[kh][t]));
  };

  auto kstep = [&](auto first) __attribute__((always_inline)) {
    using First = decltype(first);
    // first half: MFMAs of k half 0, reads of k half 1 of the same image
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if constexpr (First::value) {
        mma_first(0, 4 * g);
        mma_first(0, 4 * g + 1);
        read1(1, g);
        mma_first(0, 4 * g + 2);
        mma_first(0, 4 * g + 3);
      } else {
        mma_acc(0, 4 * g);
        mma_acc(0, 4 * g + 1);
        read1(1, g);
        mma_acc(0, 4 * g + 2);
        mma_acc(0, 4 * g + 3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // my reads of this image are back, my pieces of the next step have landed (and a previous tile's stores are out)
    if (V & 128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    offa += flip;
    offb += flip;
    flip = -flip;
    // second half: k half 1; reads of k half 0 of the NEXT step; the pieces of the step after it
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      mma_acc(1, 4 * g);
      mma_acc(1, 4 * g + 1);
      if (!(V & 64)) read1(0, g);
      __builtin_amdgcn_sched_barrier(0);
      mma_acc(1, 4 * g + 2);
      mma_acc(1, 4 * g + 3);
      if (!(V & 32)) piece(g);
      __builtin_amdgcn_sched_barrier(0);
    }
    fetch_advance();
  };

  // ---- epilogues.  acc[f][t][e] of lane (n16, q4): feature 16 f + 4 q4 + e, token 16 t + n16 of the wave tile ----
  const int n_out = EPI == TEPI_SILU ? a.N >> 1 : a.N;
  // the output value of (fragment f, token fragment t, e): SwiGLU pairs the gate fragment f with the up fragment f + 4
  // (the reference's three roundings, silu on v_exp_f32 / v_rcp_f32 as gemm_tile_kernel's epilogue)
  auto out_value = [&](int f, int t, int e, int n0) __attribute__((always_inline)) -> float {
    if (EPI == TEPI_SILU) {
      const float gb = rbf(acc[f][t][e]);
      const float sb = rbf(gb * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb * -1.4426950408889634f)));
      return sb * rbf(acc[f + 4][t][e]);
    }
    float o = acc[f][t][e];
    if (BIAS) o += bf2f(a.bias[n0 + fw * 128 + 16 * f + 4 * q4 + e]);
    return o;
  };
  constexpr int NF = EPI == TEPI_SILU ? 4 : 8;  // output feature fragments of a wave
  // direct form (edge shapes, split-K partials): four features per lane and store
  auto epilogue = [&](int m0, int n0) __attribute__((always_inline)) {
    if (V & 512) {  // ablation: no output (the accumulators are kept alive by an empty asm)
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("" ::"a"(acc[f][t]));
      return;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int tok = m0 + w16_acc_token(tw, t, n16);
      const bool tok_ok = tok < a.M;
      if (EPI == TEPI_PARTIAL) {  // fp32 sums of this K slice: [slice][token][feature]
        float* prow = a.part + ((int64_t)blockIdx.y * a.M + min(tok, a.M - 1)) * a.N;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
          const int col = n0 + w16_acc_feature(fw, f, 0, q4);
          if (tok_ok && col < a.N) *reinterpret_cast<f32x4*>(prow + col) = acc[f][t];
        }
        continue;
      }
      uint16_t* yrow = a.y + (int64_t)min(tok, a.M - 1) * a.ldy;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int col = (EPI == TEPI_SILU ? (n0 >> 1) + fw * 64 : n0 + fw * 128) + 16 * f + 4 * q4;
        const u32x2 pk = {pack_bf(out_value(f, t, 0, n0), out_value(f, t, 1, n0)), pack_bf(out_value(f, t, 2, n0), out_value(f, t, 3, n0))};
        if (tok_ok && col < n_out) *reinterpret_cast<u32x2*>(yrow + col) = pk;
      }
    }
  };

  // whole-line form (V & 32768; the host guarantees full feature tiles and 16-byte aligned rows): a slab = 32 tokens
  // (token fragments 2 tt, 2 tt + 1) x 64 output features (fragments 4 u .. 4 u + 3) through the wave's 4 KiB stage
  // (128-byte rows, 16-byte chunks XOR-swizzled by the row), written and read with inline assembly (for a C++ access to
  // LDS the compiler drains the VMEM queue first), the reads of one slab in flight while the next one is converted.
  constexpr int NU = NF / 4, NSLAB = 4 * NU;
  const uint32_t stg_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + 2 * W16_STEP_BYTES) + wave * W4_STAGE_BYTES;
  // after the exchange lane (n16, q4) holds chunk 2 * (2 pr + (q4 & 1)) + (q4 >> 1) of row 16 ts + n16 (pr: fragment pair)
  const uint32_t stg_w = stg_base + (uint32_t)n16 * 128 + (uint32_t)(((2 * (q4 & 1) + (q4 >> 1)) ^ (n16 & 7)) << 4);
  const uint32_t stg_r = stg_base + (uint32_t)(lane >> 3) * 128 + (uint32_t)(((lane & 7) ^ (lane >> 3)) << 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (uint32_t)a.M * (uint32_t)a.ldy * 2, 0x00020000);
  const uint32_t y_lane = (uint32_t)(lane >> 3) * (uint32_t)a.ldy * 2 + (uint32_t)(lane & 7) * 16;
  auto epilogue_lines = [&](int m0, int n0) __attribute__((always_inline)) {
    u32x4 out[4], back[4];
    uint32_t w_addr = stg_w;  // (opaque per tile: otherwise the chunk addresses are hoisted out of the tile loop and spilled)
    asm volatile("" : "+v"(w_addr));
    auto convert = [&](int slab) __attribute__((always_inline)) {
      const int tt = slab / NU, u = slab % NU;
#pragma unroll
      for (int ts = 0; ts < 2; ++ts)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int t = 2 * tt + ts, f = 4 * u + 2 * pr;
          uint32_t x0 = pack_bf(out_value(f, t, 0, n0), out_value(f, t, 1, n0)), x1 = pack_bf(out_value(f, t, 2, n0), out_value(f, t, 3, n0));
          uint32_t y0 = pack_bf(out_value(f + 1, t, 0, n0), out_value(f + 1, t, 1, n0)), y1 = pack_bf(out_value(f + 1, t, 2, n0), out_value(f + 1, t, 3, n0));
          // rows of 16 lanes: X' = {X.r0, Y.r0, X.r2, Y.r2}, Y' = {X.r1, Y.r1, X.r3, Y.r3}: row q4 then holds features
          // 8 (q4 >> 1) .. + 7 of fragment f + (q4 & 1)
          const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
          out[2 * ts + pr] = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
    };
    auto stage_in = [&]() __attribute__((always_inline)) {
      asm volatile("ds_write_b128 %0, %1" ::"v"(w_addr), "v"(out[0]) : "memory");
      asm volatile("ds_write_b128 %0, %1" ::"v"(w_addr ^ 64u), "v"(out[1]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(w_addr), "v"(out[2]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:2048" ::"v"(w_addr ^ 64u), "v"(out[3]) : "memory");
    };
    auto stage_out = [&]() __attribute__((always_inline)) {
      asm volatile("ds_read_b128 %0, %1" : "=v"(back[0]) : "v"(stg_r) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(back[1]) : "v"(stg_r) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(back[2]) : "v"(stg_r) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(back[3]) : "v"(stg_r) : "memory");
    };
    auto store_lines = [&](int slab) __attribute__((always_inline)) {
      const int tt = slab / NU, u = slab % NU;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(back[0]), "+v"(back[1]), "+v"(back[2]), "+v"(back[3])::"memory");
      const uint32_t colb = (uint32_t)(EPI == TEPI_SILU ? (n0 >> 1) + fw * 64 : n0 + fw * 128 + u * 64) * 2;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const uint32_t rowb = (uint32_t)(m0 + tw * 128 + tt * 32 + it * 8) * (uint32_t)a.ldy * 2 + colb;
        __builtin_amdgcn_raw_buffer_store_b128(back[it], yrsrc, y_lane + rowb, 0, 0);  // (rows past M: beyond the descriptor)
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

  // ---- prologue: steps 0 and 1 of the stream into images 0 and 1; step 0 landed and published; its k half 0 read ----
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
  for (int q = 0; q < 16; ++q) read1(0, q);
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
    kstep(std::true_type{});
    for (int kt = 1; kt < KT; ++kt) kstep(std::false_type{});
    __builtin_amdgcn_sched_barrier(0);
    // The matrix pipe's results are a software-managed hazard for what reads the accumulators next, and the compiler
    // cannot see inline-assembly MFMAs: 32 cycles of margin, inside an asm statement that "modifies" accumulators -
    // every read of an accumulator the compiler generates for the epilogue (v_accvgpr_read copies are placed by the
    // register allocator, which no scheduling barrier constrains) depends on one of these three statements, and the
    // two empty ones follow the fence in program order.  (An asm statement takes at most 30 operands.)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[1][4]), "+a"(acc[1][5]), "+a"(acc[1][6]), "+a"(acc[1][7]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[2][4]), "+a"(acc[2][5]));
    asm volatile("" : "+a"(acc[2][6]), "+a"(acc[2][7]), "+a"(acc[3][0]), "+a"(acc[3][1]), "+a"(acc[3][2]), "+a"(acc[3][3]), "+a"(acc[3][4]), "+a"(acc[3][5]), "+a"(acc[3][6]), "+a"(acc[3][7]), "+a"(acc[4][0]), "+a"(acc[4][1]), "+a"(acc[4][2]), "+a"(acc[4][3]), "+a"(acc[4][4]), "+a"(acc[4][5]), "+a"(acc[4][6]), "+a"(acc[4][7]), "+a"(acc[5][0]), "+a"(acc[5][1]), "+a"(acc[5][2]));
    asm volatile("" : "+a"(acc[5][3]), "+a"(acc[5][4]), "+a"(acc[5][5]), "+a"(acc[5][6]), "+a"(acc[5][7]), "+a"(acc[6][0]), "+a"(acc[6][1]), "+a"(acc[6][2]), "+a"(acc[6][3]), "+a"(acc[6][4]), "+a"(acc[6][5]), "+a"(acc[6][6]), "+a"(acc[6][7]), "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]), "+a"(acc[7][6]), "+a"(acc[7][7]));
    // A workgroup's last tile: its (empty) past-the-end pieces must not outlive the workgroup's LDS
    if (blk + stride >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (LINES) epilogue_lines(m0, n0);
    else epilogue(m0, n0);
    __builtin_amdgcn_sched_barrier(0);
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
