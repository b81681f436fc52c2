/*
 * mi355_nanovllm_experiments.h - entry points of MEASURED-AND-LOST experiments.
 *
 * Not part of the product ABI: libmi355_nanovllm.so exports these only when built with
 * `make -C nano-vllm-ascend_amd/csrc EXPERIMENTS=1` (-DMI_EXPERIMENTS); the default build, the default header and
 * the product path do not contain them.  They stay in the tree because DESIGN.md quotes their measurements and
 * their tests reproduce them (tests skip when the symbols are absent):
 *   mi_add_rmsnorm_splitk_warm / mi_warm_l2   L2 warming by the norm launch's idle CUs: decode step 1.52-1.54 ms
 *                                             against 1.50-1.51 without (round 3)
 *   mi_mlp_half_fused                         the MLP half of a decode layer as one persistent launch: 23.6 us
 *                                             against 12.6 us for the three launches (profiles/r03_mlp_half.txt)
 */
#ifndef MI355_NANOVLLM_EXPERIMENTS_H
#define MI355_NANOVLLM_EXPERIMENTS_H

#include "mi355_nanovllm.h"

#ifdef __cplusplus
extern "C" {
#endif

/* mi_add_rmsnorm_splitk for the decode chain (rows <= 64, cols <= 1024) with the CUs the norm leaves idle
 * (a launch of `rows` workgroups on a 256-CU part) pulling the packed weights of the launches BEHIND it - the
 * projections this norm feeds (linear.py:73,150 of the same layer) - into the L2 of the XCD that will read them.
 * warmN: mi_pack_weight output (or NULL), warmN_tile_bytes = 32 * K (one 16-row tile), a multiple of 4096.
 * Results are those of mi_add_rmsnorm_splitk bit for bit; only the timing of the following launches changes. */
int mi_add_rmsnorm_splitk_warm(const float* partials, int nsplit, const mi_bf16* residual,
                               const mi_bf16* w, mi_bf16* y, mi_bf16* residual_out, int rows,
                               int cols, float eps, const void* warm0, size_t warm0_bytes,
                               int warm0_tile_bytes, const void* warm1, size_t warm1_bytes,
                               int warm1_tile_bytes, mi_stream stream);

/* The warming workgroups of mi_add_rmsnorm_splitk_warm as a launch of their own: n_workgroups (a multiple of 8)
 * workgroups pull the given packed weights into L2.  The tensor-parallel decode path queues it on a forked stream
 * beside mi_allreduce_add_rmsnorm (MI355_SEAM_OVERLAP=1, SURVEY 8(f)1: the all-reduce overlapped with the next
 * projection's weight stream). */
int mi_warm_l2(const void* warm0, size_t warm0_bytes, int warm0_tile_bytes, const void* warm1,
               size_t warm1_bytes, int warm1_tile_bytes, int n_workgroups, mi_stream stream);

/* EXPERIMENT, not on the product path (decode chain): the MLP half of a decode layer
 * (layernorm.py:27-38 -> linear.py:73 + activation.py:10-12 -> linear.py:150) as ONE persistent launch of 256
 * workgroups with in-launch hand-offs instead of three launches; bit-identical to
 * mi_add_rmsnorm_splitk + mi_gemm_bf16_packed(epilogue 1) + mi_gemm_bf16_packed_splitk(ksplit 4).
 * hidden 1024, intermediate 3072, 1 <= rows <= 32, partials_in / partials_out [4][rows][1024] fp32;
 * sync_words: 8 x uint32, zeroed once by the caller (word 6 != 0 afterwards: a hand-off timed out). */
int mi_mlp_half_fused(const float* partials_in, const mi_bf16* residual, const mi_bf16* norm_w, float eps,
                      const mi_bf16* w_gate_up_packed, const mi_bf16* w_down_packed,
                      mi_bf16* residual_out, mi_bf16* xn_scratch, mi_bf16* act_scratch,
                      float* partials_out, uint32_t* sync_words, int rows, int hidden, int intermediate,
                      mi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_NANOVLLM_EXPERIMENTS_H */
