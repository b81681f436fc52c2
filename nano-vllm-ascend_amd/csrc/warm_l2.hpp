// L2 warming: workgroups that have nothing else to do pull the packed weights of the launches BEHIND them into the
// L2 of the XCD whose workgroups will read them (used by mi_add_rmsnorm_splitk_warm and mi_warm_l2, elementwise.hip).
#pragma once
#include "mi_common.hpp"

namespace mi {

// A decode-sized launch has a few dozen workgroups of work and a chip of 256 CUs: spare workgroups pull the
// weights of the NEXT launches (the GEMMs this norm feeds) into L2 while the norm's own dependent chain (partials ->
// sum of squares -> barrier -> scale) runs.  L2 is per XCD: a weight tile must land in the L2 of the XCD whose
// workgroup will read it.  The dispatcher deals workgroups to the eight XCDs round-robin by linear block id (a speed
// assumption only - a wrong guess costs the benefit, never the result), and every consumer kernel here maps its
// 16-row weight tile t to a block id = t mod 8 (gemm_skinny_kernel: tile bx, the SwiGLU pair bx and bx + N/32, the
// K slices of tile bx on blockIdx.y with gridDim.x a multiple of 8), so tile t belongs to XCD t % 8, and so do the
// warming workgroups `first` + p with p % 8 == t % 8.
struct WarmArgs {
  const char* base[2];   // packed weights (fragment-native: a 16-row tile is one contiguous run of tile_bytes)
  uint32_t bytes[2];
  uint32_t tile_bytes[2];
  int first;             // block id of the first warming workgroup, a multiple of 8
};

// The lines are requested by LDS-DMA into a scratch block nobody reads: a load with a register destination would
// need its registers kept free until it returns (an `asm` load's destination is invisible to the compiler's liveness).
__device__ __forceinline__ void warm_l2(const WarmArgs& wa, int p, int n_warm, int tid, char* scratch) {
  const int xcd = p & 7, q = p >> 3, nq = n_warm >> 3;
  char* dst = scratch + (tid >> 6) * 1024;  // one 1 KiB landing block per wave (the DMA destination is lane-linear)
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (!wa.base[r]) continue;
    const uint32_t tb = wa.tile_bytes[r], ntiles = wa.bytes[r] / tb;
    if ((uint32_t)xcd >= ntiles) continue;
    const uint32_t mine = (ntiles - xcd + 7) >> 3;            // tiles xcd, xcd + 8, ... of this XCD
    const uint32_t pieces = mine * (tb >> 12);                // in 4 KiB pieces, dealt to the XCD's warming workgroups
    for (uint32_t j = q; j < pieces; j += nq) {
      const uint32_t t = j / (tb >> 12), o = (j % (tb >> 12)) << 12;
      const char* src = wa.base[r] + (size_t)(xcd + 8 * t) * tb + o + tid * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace mi
