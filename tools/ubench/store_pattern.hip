// Store-path throughput of one GEMM tile epilogue, by pattern: every workgroup (512 threads, one per CU) writes the
// 256 x 256 bf16 tile of a [M][N] matrix the way gemm_tile.hip's accumulators leave it -
//   pattern 0: row-per-lane: lane (hi, l31) stores 16 bytes of token row l31 (+32), 16 instructions per wave,
//              each touching 32 rows x 32 bytes;
//   pattern 1: whole rows: 16 lanes store one 256-byte row, an instruction covers 4 complete rows.
// Same bytes, same number of store instructions.  Standalone: hipcc --offload-arch=gfx950 -O3 store_pattern.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(uint16_t* __restrict__ y, int64_t ldy, int tiles_f, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fh = wave >> 2, tq = wave & 3, hi = lane >> 5, l31 = lane & 31;
  for (int rep = 0; rep < reps; ++rep) {
    const int tile = blockIdx.x + rep * gridDim.x;
    const int m0 = (tile / tiles_f) * 256, n0 = (tile % tiles_f) * 256;
    const u32x4 v = {(uint32_t)tile, (uint32_t)lane, (uint32_t)wave, 7u};
    if (PATTERN == 0) {
#pragma unroll
      for (int bh = 0; bh < 2; ++bh)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int tok = m0 + tq * 64 + bh * 32 + l31, col = n0 + fh * 128 + j * 32 + 16 * p + 8 * hi;
            *reinterpret_cast<u32x4*>(y + (int64_t)tok * ldy + col) = v;
          }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tok = m0 + tq * 64 + i * 4 + (lane >> 4), col = n0 + fh * 128 + (lane & 15) * 8;
        *reinterpret_cast<u32x4*>(y + (int64_t)tok * ldy + col) = v;
      }
    }
  }
}

int main() {
  const int M = 16384, N = 4096, tiles_f = N / 256, tiles = (M / 256) * tiles_f;
  uint16_t* y;
  hipMalloc(&y, (size_t)M * N * 2);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int pat = 0; pat < 2; ++pat)
    for (int grid : {32, 64, 128, 256, 1024}) {
      const int reps = tiles / grid;
      float best = 1e9f;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(a);
        if (pat == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(grid), dim3(512), 0, 0, y, (int64_t)N, tiles_f, reps);
        else hipLaunchKernelGGL(store_kernel<1>, dim3(grid), dim3(512), 0, 0, y, (int64_t)N, tiles_f, reps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (it && ms < best) best = ms;
      }
      printf("pattern %d (%s) grid %4d: %7.1f us for %.0f MB = %.2f TB/s\n", pat, pat ? "whole rows  " : "row per lane", grid,
             best * 1e3, (double)M * N * 2 / 1e6, (double)M * N * 2 / (best * 1e-3) / 1e12);
    }
  return 0;
}
