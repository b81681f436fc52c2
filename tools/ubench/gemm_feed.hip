// Feasibility probe for a one-wave-per-SIMD GEMM main loop (256 x 256 x 64 tile, 4 waves x 512 VGPRs, wave tile
// 128 x 128): how many cycles does a K step take when the single wave of a SIMD has to issue, besides its 64
// v_mfma_f32_32x32x16_bf16, the fragment reads and the operand feed itself?  Timing only - the operands are whatever
// the feed puts into LDS; nothing is checked.
//   MODE 0  MFMAs only (64 per K step and wave)                              -> the matrix pipe's own pace
//   MODE 1  + 32 ds_read_b128 per K step (A and B fragments of the 128 x 128 wave tile)
//   MODE 2  + feed by LDS-DMA: 16 global_load_lds_dwordx4 per wave and K step (64 KiB per workgroup)
//   MODE 3  + feed through registers: 16 global_load_dwordx4 (top of the step) + 16 ds_write_b128 (behind the last MFMAs)
//   MODE 4  MODE 2 with the DMA pieces pinned one behind every fourth MFMA
// One barrier per K step, double-buffered 2 x 64 KiB LDS image; sources are eight 4 MiB streams (one per XCD: L2 hits
// after the first pass, like a GEMM's weight panel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int KSTEP_BYTES = 65536;

template <int MODE>
__global__ __launch_bounds__(256, 1) void feed_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * KSTEP_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const char* stream = src + (size_t)(blockIdx.x & 7) * ((size_t)ksteps * KSTEP_BYTES);
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fragment read offsets: row l31 of a 32-row block, 128-byte rows, 16-byte chunk (2 kk + hi) ^ swizzle
  int off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) off[kk] = (l31 >> 3) * 1024 + (l31 & 7) * 128 + (((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4);
  u32x4 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = u32x4{0x3f803f80u + (uint32_t)lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    B[i] = u32x4{0x3f803f80u, 0x3f803f80u + (uint32_t)wave, 0x3f803f80u, 0x3f803f80u};
  }
  u32x4 stage[16];
  for (int t = 0; t < ksteps; ++t) {
    const int p = t & 1;
    const char* cur = lds + p * KSTEP_BYTES;
    char* nxt = lds + (p ^ 1) * KSTEP_BYTES;
    const char* g = stream + (size_t)t * KSTEP_BYTES + wave * 16384 + lane * 16;
    if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (MODE == 3) {  // this step's loads first: a whole K step of MFMAs to land in; written to LDS at the END of the step
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(g + i * 1024));
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          A[i] = *reinterpret_cast<const u32x4*>(cur + (wave >> 1) * 16384 + i * 4096 + off[kk]);
          B[i] = *reinterpret_cast<const u32x4*>(cur + 32768 + (wave & 1) * 16384 + i * 4096 + off[kk]);
        }
      }
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (kk * 4 + i) * 1024),
                                           (__attribute__((address_space(3))) void*)(nxt + wave * 16384 + (kk * 4 + i) * 1024), 16, 0, 0);
      }
      if (MODE == 3 && kk == 3) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[i]), __builtin_bit_cast(bf16x8, B[j]), acc[i][j], 0, 0, 0);
          if (MODE == 4) {  // one DMA piece behind every fourth MFMA, pinned there
            if ((i * 4 + j) % 4 == 3) {
              const int q = kk * 4 + i;
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + q * 1024),
                                               (__attribute__((address_space(3))) void*)(nxt + wave * 16384 + q * 1024), 16, 0, 0);
              __builtin_amdgcn_sched_group_barrier(0x8, 4, 0);
              __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);
            }
          }
          if (MODE == 3 && kk == 3) {  // the step's sixteen LDS writes, one behind each MFMA of the last k group
            const int q = i * 4 + j;
            *reinterpret_cast<u32x4*>(nxt + wave * 16384 + q * 1024 + lane * 16) = stage[q];
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}


// MODE 5 / 6: the software pipeline a one-wave-per-SIMD kernel needs (the vendor's MT256x256x64 assembly kernel has the
// same shape): fragments double-buffered in REGISTERS - the reads of k group kk + 1 are issued between the MFMAs of
// group kk - and the feed issued as `buffer_load_dwordx4 ... lds` (SGPR offsets, M0 destinations: two SALU per piece,
// no VALU address arithmetic) between the MFMAs of the last group.  One wait + barrier per K step: "my fragment reads
// of this buffer are back, my pieces of the next step have landed" -> the buffer is re-filled two steps ahead.
//   PIPE 0: 16 pieces all in group 3;  PIPE 1: pieces spread 1 per 4 MFMAs over groups 3, 0, 1, 2 is not legal (the
//   buffer is being read) - instead 8 in group 3 and 8 in the next step's group 0 (its reads touch the OTHER buffer...
//   no: group 0 reads k group 1 of the SAME step).  So PIPE 1 = pieces 2 per 2 MFMAs (front-loaded), PIPE 0 = 1 per MFMA.
template <int PIPE>
__global__ __launch_bounds__(256, 1) void pipe_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * KSTEP_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const char* stream = src + (size_t)(blockIdx.x & 7) * ((size_t)ksteps * KSTEP_BYTES);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, ksteps * KSTEP_BYTES, 0x00020000);
  const int voff = wave * 16384 + lane * 16;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int offa[4], offb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int o = (l31 >> 3) * 1024 + (l31 & 7) * 128 + (((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4);
    offa[kk] = o + (wave >> 1) * 16384;
    offb[kk] = o + 32768 + (wave & 1) * 16384;
  }
  u32x4 Ra[2][4], Rb[2][4];
  auto dma = [&](int t, int i, int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + buf * KSTEP_BYTES + wave * 16384 + i * 1024),
                                             16, voff, t * KSTEP_BYTES + i * 1024, 0, 0);
  };
  auto reads = [&](int kk, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      Ra[set][i] = *reinterpret_cast<const u32x4*>(lds + i * 4096 + offa[kk]);
      Rb[set][i] = *reinterpret_cast<const u32x4*>(lds + i * 4096 + offb[kk]);
    }
  };
  auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra[set][i]), __builtin_bit_cast(bf16x8, Rb[set][j]), acc[i][j], 0, 0, 0);
  };
  auto flip = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { offa[kk] ^= KSTEP_BYTES; offb[kk] ^= KSTEP_BYTES; }
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(0, i, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(1, i, 1);
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  reads(0, 0);
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < ksteps; ++t) {
    const int p = t & 1;
    // group 0: MFMAs of k group 0, reads of k group 1 - one read behind each of the first eight MFMAs
    reads(1, 1);
    mma(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    reads(2, 0);
    mma(1);
#pragma unroll
    for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    reads(3, 1);
    mma(0);
#pragma unroll
    for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    __builtin_amdgcn_sched_group_barrier(0x8, 8, 0);
    __builtin_amdgcn_sched_barrier(0);
    // every read of this buffer is back (lgkmcnt), my pieces of step t + 1 have landed (vmcnt) -> barrier -> the buffer
    // is free for step t + 2 and the other one readable
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    flip();
    // group 3 in pinned chunks (the scheduler does not move LDS-DMA under sched_group_barrier: they are chained through M0)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i0 = (2 * q) >> 2, j0 = (2 * q) & 3;
      acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra[1][i0]), __builtin_bit_cast(bf16x8, Rb[1][j0]), acc[i0][j0], 0, 0, 0);
      if (q < 4) Ra[0][q] = *reinterpret_cast<const u32x4*>(lds + q * 4096 + offa[0]);
      else Rb[0][q - 4] = *reinterpret_cast<const u32x4*>(lds + (q - 4) * 4096 + offb[0]);
      if (PIPE == 0 || PIPE == 3) { dma(t + 2, 2 * q, p); }
      __builtin_amdgcn_sched_barrier(0);
      acc[i0][j0 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra[1][i0]), __builtin_bit_cast(bf16x8, Rb[1][j0 + 1]), acc[i0][j0 + 1], 0, 0, 0);
      if (PIPE == 0) { dma(t + 2, 2 * q + 1, p); }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 8; q < 16; ++q) {
      const int i0 = q >> 2, j0 = q & 3;
      acc[i0][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra[1][i0]), __builtin_bit_cast(bf16x8, Rb[1][j0]), acc[i0][j0], 0, 0, 0);
      if (PIPE == 1) { dma(t + 2, 2 * (q - 8), p); dma(t + 2, 2 * (q - 8) + 1, p); }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { sink[1024 + 2 * blockIdx.x] = (float)(c1 - c0); sink[1025 + 2 * blockIdx.x] = (float)(r1 - r0); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}


// pipe2: fragment reads TWO k groups ahead (four register sets, one per k group): the buffer of step t has been read
// completely by the end of group 1, so ONE wait + barrier at the start of group 2 ("my reads of this buffer are back,
// my pieces of step t + 1 have landed") both frees it for step t + 2 and opens the other buffer; the sixteen pieces go
// one behind every second MFMA of groups 2 and 3 (a piece blocks its wave ~60 cycles, an MFMA runs 32) and have a
// whole K step to land.
//   FEED 0: no pieces in the loop;  FEED 1: 16 pieces, 1 per 2 MFMAs over groups 2 and 3;  FEED 2: 16 pieces, 1 per MFMA over group 2
//   FEED 3: FEED 1 with the piece's offset in the VECTOR offset (one v_add per piece) instead of the scalar one
//   FEED 4 / 5 / 6: FEED 3 fetching what the qkv GEMM of the bench fetches (x: 16384 x 1024 bf16 = 32 MiB, w: 4096 x
//           1024 = 8 MiB behind it; waves 0, 1 the weight panel of feature tile blockIdx & 15, waves 2, 3 the activation
//           panel of token tile (blockIdx >> 4) + 16 (t / 16); 8 rows x 128 bytes per piece, rows 2 KiB apart) with the
//           default / nt (aux 2) / sc1 (aux 16) cache policy on the loads
template <int FEED>
__global__ __launch_bounds__(256, 1) void pipe2_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * KSTEP_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  constexpr bool GEMM_LIKE = FEED >= 4;
  const char* stream = GEMM_LIKE ? src : src + (size_t)(blockIdx.x & 7) * ((size_t)ksteps * KSTEP_BYTES);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, GEMM_LIKE ? (40 << 20) : ksteps * KSTEP_BYTES, 0x00020000);
  const int voff = GEMM_LIKE ? (lane >> 3) * 2048 + (lane & 7) * 16 : wave * 16384 + lane * 16;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int offa[4], offb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    const int o = (l31 >> 3) * 1024 + (l31 & 7) * 128 + (((2 * kk + hi) ^ ((l31 >> 1) & 7)) << 4);
    offa[kk] = o + (wave >> 1) * 16384;
    offb[kk] = o + 32768 + (wave & 1) * 16384;
  }
  u32x4 Ra[4][4], Rb[4][4];
  auto dma = [&](int t, int i, int buf) __attribute__((always_inline)) {
    auto dst = (__attribute__((address_space(3))) void*)(lds + buf * KSTEP_BYTES + wave * 16384 + i * 1024);
    if (FEED >= 4) {
      // row base of this wave's 128 rows: weights behind the 32 MiB of activations
      const int tile = wave < 2 ? (int)(blockIdx.x & 15) : (((int)blockIdx.x >> 4) + 16 * (t >> 4)) & 63;
      const uint32_t rows = (wave < 2 ? (32u << 20) : 0u) + (uint32_t)(tile * 256 + (wave & 1) * 128 + i * 8) * 2048u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff + rows, (t & 15) * 128, 0, FEED == 5 ? 2 : (FEED == 6 ? 16 : 0));
    } else if (FEED == 3) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff + (uint32_t)(i * 1024 + (ksteps & 0)), t * KSTEP_BYTES, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, t * KSTEP_BYTES + i * 1024, 0, 0);
    }
  };
  auto read1 = [&](int kk, int q) __attribute__((always_inline)) {  // q-th of the eight fragment reads of k group kk
    if (q < 4) Ra[kk][q] = *reinterpret_cast<const u32x4*>(lds + q * 4096 + offa[kk]);
    else Rb[kk][q - 4] = *reinterpret_cast<const u32x4*>(lds + (q - 4) * 4096 + offb[kk]);
  };
  auto mma1 = [&](int kk, int m) __attribute__((always_inline)) {
    const int i = m >> 2, j = m & 3;
    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ra[kk][i]), __builtin_bit_cast(bf16x8, Rb[kk][j]), acc[i][j], 0, 0, 0);
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(0, i, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(1, i, 1);
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int q = 0; q < 8; ++q) { read1(0, q); read1(1, q); }
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < ksteps; ++t) {
    const int p = t & 1;
    // groups 0, 1: MFMAs of k group g, reads of k group g + 2 of the same buffer
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        mma1(g, 2 * q);
        read1(g + 2, q);
        mma1(g, 2 * q + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the other buffer from here on
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { offa[kk] ^= KSTEP_BYTES; offb[kk] ^= KSTEP_BYTES; }
#pragma unroll
    for (int g = 2; g < 4; ++g) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        mma1(g, 2 * q);
        read1(g - 2, q);
        if (FEED == 2 && g == 2) dma(t + 2, 2 * q, p);
        __builtin_amdgcn_sched_barrier(0);
        mma1(g, 2 * q + 1);
        if (FEED == 1 || FEED >= 3) dma(t + 2, (g - 2) * 8 + q, p);
        if (FEED == 2 && g == 2) dma(t + 2, 2 * q + 1, p);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { sink[1024 + 2 * blockIdx.x] = (float)(c1 - c0); sink[1025 + 2 * blockIdx.x] = (float)(r1 - r0); }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// MFMA shape under the power limit: the same flops per K step as 64 v_mfma_f32_32x32x16_bf16 or 128
// v_mfma_f32_16x16x32_bf16 (half the accumulator register traffic per flop, twice the operand traffic), nothing else
// in the loop; operands are random bf16 patterns loaded once.
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int SHAPE>
__global__ __launch_bounds__(256, 1) void mfma_shape_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  const int lane = threadIdx.x & 63;
  u32x4 A[8], B[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    A[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(i * 64 + lane) * 16 + (size_t)(blockIdx.x & 255) * 65536);
    B[i] = *reinterpret_cast<const u32x4*>(src + (size_t)((8 + i) * 64 + lane) * 16 + (size_t)(blockIdx.x & 255) * 65536);
  }
  float s = 0.f;
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int t = 0; t < ksteps; ++t)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[(i + kk) & 7]), __builtin_bit_cast(bf16x8, B[(j + 2 * kk) & 7]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int t = 0; t < ksteps; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[(i + ks) & 7]), __builtin_bit_cast(bf16x8, B[(j + 3 * ks) & 7]), acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { sink[1024 + 2 * blockIdx.x] = (float)(c1 - c0); sink[1025 + 2 * blockIdx.x] = (float)(r1 - r0); }
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// (a kernel of its own: sharing a function with the 32 x 32 branch left the 64 accumulators shuffled through
// v_accvgpr_mov every iteration)
__global__ __launch_bounds__(256, 1) void mfma16_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  const int lane = threadIdx.x & 63;
  u32x4 A[8], B[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    A[i] = *reinterpret_cast<const u32x4*>(src + (size_t)(i * 64 + lane) * 16 + (size_t)(blockIdx.x & 255) * 65536);
    B[i] = *reinterpret_cast<const u32x4*>(src + (size_t)((8 + i) * 64 + lane) * 16 + (size_t)(blockIdx.x & 255) * 65536);
  }
  f32x4 acc[64];
#pragma unroll
  for (int m = 0; m < 64; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < ksteps; ++t) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 64; ++m)
        // (in place by inline assembly: with the builtin hipcc gives the result a different register quad than the
        // addend and copies 64 quads around at the loop's back edge)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(A[((m >> 3) + ks) & 7]), "v"(B[((m & 7) + 3 * ks) & 7]));
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { sink[1024 + 2 * blockIdx.x] = (float)(c1 - c0); sink[1025 + 2 * blockIdx.x] = (float)(r1 - r0); }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 64; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

// pipe3: the pipe2 loop on v_mfma_f32_16x16x32_bf16 (the vendor kernel's shape; 4.6 % more clock than 32x32x16 under the
// power limit, see below): a K step is two k halves of 64 MFMAs (8 x 8 accumulators of 16 x 16), fragments of the next
// half are read during the current one (two register sets of 16 fragments), one wait + barrier per K step behind the
// first half, the sixteen pieces one behind every fourth MFMA of the second half.  MFMAs in place by inline assembly.
template <int FEED>
__global__ __launch_bounds__(256, 1) void pipe3_kernel(const char* __restrict__ src, float* __restrict__ sink, int ksteps) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * KSTEP_BYTES];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr bool GEMM_LIKE = FEED >= 4;
  const char* stream = GEMM_LIKE ? src : src + (size_t)(blockIdx.x & 7) * ((size_t)ksteps * KSTEP_BYTES);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, GEMM_LIKE ? (40 << 20) : ksteps * KSTEP_BYTES, 0x00020000);
  const int voff = GEMM_LIKE ? (lane >> 3) * 2048 + (lane & 7) * 16 : wave * 16384 + lane * 16;
  f32x4 acc[64];
#pragma unroll
  for (int m = 0; m < 64; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  // conflict-free stand-in addresses: fragment f of half h of operand o at o * 32 KiB + (wave half) * 16 KiB + h * 8 KiB + f * 1 KiB + 16 * lane
  int offa = (wave >> 1) * 16384 + lane * 16, offb = 32768 + (wave & 1) * 16384 + lane * 16;
  u32x4 Ra[2][8], Rb[2][8];
  auto dma = [&](int t, int i, int buf) __attribute__((always_inline)) {
    auto dst = (__attribute__((address_space(3))) void*)(lds + buf * KSTEP_BYTES + wave * 16384 + i * 1024);
    if (FEED >= 4) {
      const int tile = wave < 2 ? (int)(blockIdx.x & 15) : (((int)blockIdx.x >> 4) + 16 * (t >> 4)) & 63;
      const uint32_t rows = (wave < 2 ? (32u << 20) : 0u) + (uint32_t)(tile * 256 + (wave & 1) * 128 + i * 8) * 2048u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff + rows, (t & 15) * 128, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff + (uint32_t)(i * 1024 + (ksteps & 0)), t * KSTEP_BYTES, 0, 0);
    }
  };
  auto read1 = [&](int h, int q) __attribute__((always_inline)) {  // q-th of the sixteen fragment reads of k half h
    if (q < 8) Ra[h][q] = *reinterpret_cast<const u32x4*>(lds + h * 8192 + q * 1024 + offa);
    else Rb[h][q - 8] = *reinterpret_cast<const u32x4*>(lds + h * 8192 + (q - 8) * 1024 + offb);
  };
  auto mma1 = [&](int h, int m) __attribute__((always_inline)) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(Ra[h][m >> 3]), "v"(Rb[h][m & 7]));
  };
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(0, i, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) dma(1, i, 1);
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int q = 0; q < 16; ++q) read1(0, q);
  const uint64_t c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < ksteps; ++t) {
    const int p = t & 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // first half: MFMAs of k half 0, reads of k half 1 of the same image
      mma1(0, 4 * q);
      mma1(0, 4 * q + 1);
      read1(1, q);
      mma1(0, 4 * q + 2);
      mma1(0, 4 * q + 3);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    offa ^= KSTEP_BYTES;
    offb ^= KSTEP_BYTES;
#pragma unroll
    for (int q = 0; q < 16; ++q) {  // second half: k half 1; reads of k half 0 of the NEXT step; the pieces of step t + 2
      mma1(1, 4 * q);
      mma1(1, 4 * q + 1);
      read1(0, q);
      __builtin_amdgcn_sched_barrier(0);
      mma1(1, 4 * q + 2);
      mma1(1, 4 * q + 3);
      if (FEED) dma(t + 2, q, p);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const uint64_t c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { sink[1024 + 2 * blockIdx.x] = (float)(c1 - c0); sink[1025 + 2 * blockIdx.x] = (float)(r1 - r0); }
  float s = 0.f;
#pragma unroll
  for (int m = 0; m < 64; ++m) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

static double g_cycles_per_step, g_mhz;
typedef void (*pipe_fn)(const char*, float*, int);
static double run_fn(pipe_fn fn, const char* src, float* sink, int ksteps);
template <int PIPE>
static double run_pipe(const char* src, float* sink, int ksteps) { return run_fn(pipe_kernel<PIPE>, src, sink, ksteps); }
static double run_fn(pipe_fn fn, const char* src, float* sink, int ksteps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(fn, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  float h[2 * 256]; CK(hipMemcpy(h, sink + 1024, sizeof h, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  g_cycles_per_step = cyc / 256 / ksteps; g_mhz = cyc / rt * 100.0;
  return ms * 1e3 / 5;
}

template <int MODE>
static double run(const char* src, float* sink, int ksteps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(feed_kernel<MODE>, dim3(256), dim3(256), 0, 0, src, sink, ksteps);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / 5;  // us per launch
}

int main() {
  const int ksteps = 64;
  const size_t bytes = (size_t)40 << 20;  // >= 8 streams of ksteps x 64 KiB; the GEMM-like modes: 32 MiB of activations + 8 MiB of weights
  char* src; float* sink;
  CK(hipMalloc(&src, bytes)); CK(hipMalloc(&sink, 8192));
  // random bf16-ish data (DVFS: zero operands clock higher)
  uint32_t* h = (uint32_t*)malloc(bytes);
  uint32_t x = 12345;
  for (size_t i = 0; i < bytes / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (x & 0x807f807fu) | 0x3f003f00u; }
  CK(hipMemcpy(src, h, bytes, hipMemcpyHostToDevice));
  const double flop = 2.0 * 256 * 256 * 64 * ksteps * 256;
  const double t0 = run<0>(src, sink, ksteps), t1 = run<1>(src, sink, ksteps), t2 = run<2>(src, sink, ksteps), t3 = run<3>(src, sink, ksteps);
  const double t4 = run<4>(src, sink, ksteps);
  printf("one wave per SIMD, 256 x 256 x 64 per workgroup and K step, %d K steps, 256 workgroups (us per launch, us per K step, TFLOP/s)\n", ksteps);
  printf("  MFMAs only                         %8.1f  %6.3f  %7.0f\n", t0, t0 / ksteps, flop / t0 / 1e6);
  printf("  + fragment reads                   %8.1f  %6.3f  %7.0f\n", t1, t1 / ksteps, flop / t1 / 1e6);
  printf("  + reads + LDS-DMA feed             %8.1f  %6.3f  %7.0f\n", t2, t2 / ksteps, flop / t2 / 1e6);
  printf("  + reads + register-staged feed     %8.1f  %6.3f  %7.0f\n", t3, t3 / ksteps, flop / t3 / 1e6);
  printf("  + reads + LDS-DMA, 1 per 4 MFMAs   %8.1f  %6.3f  %7.0f\n", t4, t4 / ksteps, flop / t4 / 1e6);
  printf("software-pipelined loop (fragments double-buffered in registers, buffer_load ... lds feed); cycles per K step of the loop itself, shader clock\n");
  double t;
  t = run_pipe<2>(src, sink, ksteps); printf("  reads only, no feed                         %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_pipe<3>(src, sink, ksteps); printf("  + 8 pieces per K step (half the feed)       %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_pipe<0>(src, sink, ksteps); printf("  + 16 pieces, 1 per MFMA over group 3        %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_pipe<1>(src, sink, ksteps); printf("  + 16 pieces, 2 per MFMA behind the reads    %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  printf("reads two k groups ahead (four fragment register sets), one wait + barrier per K step\n");
  t = run_fn(pipe2_kernel<0>, src, sink, ksteps); printf("  reads only, no feed                         %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<1>, src, sink, ksteps); printf("  + 16 pieces, 1 per 2 MFMAs, groups 2 and 3  %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<2>, src, sink, ksteps); printf("  + 16 pieces, 1 per MFMA, group 2            %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<3>, src, sink, ksteps); printf("  1 per 2 MFMAs, offset in the vector offset  %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<4>, src, sink, ksteps); printf("  the qkv GEMM's fetch pattern (40 MiB)       %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<5>, src, sink, ksteps); printf("  ... loads nt                                %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<6>, src, sink, ksteps); printf("  ... loads sc1                               %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  printf("MFMA shape under the power limit (MFMAs only, random operands)\n");
  t = run_fn(mfma_shape_kernel<32>, src, sink, ksteps); printf("  64 x v_mfma_f32_32x32x16_bf16 per K step    %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(mfma16_kernel, src, sink, ksteps); printf("  128 x v_mfma_f32_16x16x32_bf16 per K step   %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  printf("the two-groups-ahead loop on 128 x v_mfma_f32_16x16x32_bf16 per K step (in-place inline assembly), next to its 32 x 32 x 16 twin\n");
  t = run_fn(pipe2_kernel<0>, src, sink, ksteps); printf("  32x32x16: reads only                        %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe3_kernel<0>, src, sink, ksteps); printf("  16x16x32: reads only                        %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<3>, src, sink, ksteps); printf("  32x32x16: + 16 pieces                       %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe3_kernel<1>, src, sink, ksteps); printf("  16x16x32: + 16 pieces                       %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe2_kernel<4>, src, sink, ksteps); printf("  32x32x16: the qkv GEMM's fetch pattern      %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  t = run_fn(pipe3_kernel<4>, src, sink, ksteps); printf("  16x16x32: the qkv GEMM's fetch pattern      %8.1f  %6.3f  %7.0f   %6.0f cyc  %5.0f MHz\n", t, t / ksteps, flop / t / 1e6, g_cycles_per_step, g_mhz);
  return 0;
}
