/*
 * mi355_nanovllm.h — C ABI of libmi355_nanovllm.so
 *
 * MI355X (gfx950 / CDNA4) replacement for the device-operator surface that the
 * reference (linzm1007/nano-vllm-ascend) reaches through torch_npu / torchair:
 * the paged-KV Qwen3 decode path and the prefill that feeds it.
 *
 * The reference has no FFI of its own: its seam is the Python class surface of
 * nanovllm/layers.  Every entry point below names the reference
 * call site it stands in for (paths relative to the reference checkout).
 *
 * Conventions (all entry points)
 *   - plain pointers + sizes; no torch types.  `mi_bf16` is a raw 16-bit
 *     bfloat16 pattern.  All device pointers must be 16-byte aligned.
 *   - asynchronous on `stream` (a hipStream_t passed as void*), allocates
 *     nothing, never synchronises, safe to call while the stream is being
 *     captured into a hipGraph.
 *   - the caller owns every buffer, including the workspace.
 *   - returns MI_OK (0) or a negative MI_E* code; never throws.  Argument
 *     validation happens on the host before anything is enqueued.
 *   - indices (block ids, slots) are NOT range-checked on the device, matching
 *     the reference ops (op_docs/_op_plugin_docs.py:9656-9657, :7123); negative
 *     slots / block ids are skipped.
 *
 * Paged KV cache layout ("fragment-native")
 *   One layer's K (or V) cache is   [num_blocks][n_kv_heads][block_size/16][2048] bf16,
 *   i.e. one 4 KiB tile per (16 consecutive slots of a block, kv head); head_dim = 128.
 *   Element (t = slot%16, d) of a K tile sits at  (d/32)*512 + (((d%32)/8)*16 + t)*8 + d%8,
 *   element (t, d) of a V tile at                 (d/32)*512 + ((t/4)*16 + d%16)*8 + ((d%32)/16)*4 + t%4
 *   (offsets in bf16 elements).  A wavefront's 1 KiB coalesced load of a tile
 *   quarter is then exactly one MFMA 16x16x32 operand fragment.  Only block ids
 *   and slot indices are part of the parity contract with the reference
 *   (model_runner.py:218 stores [nblk, block, Hkv*D]); `mi_kv_cache_gather`
 *   converts back to that logical view for bit-exact content checks.
 */
#ifndef MI355_NANOVLLM_H
#define MI355_NANOVLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t mi_bf16;
typedef void*    mi_stream; /* hipStream_t */

enum {
  MI_OK            = 0,
  MI_EINVAL        = -1, /* bad argument (null pointer, negative size, misaligned) */
  MI_EUNSUPPORTED  = -2, /* shape outside what the kernels are built for          */
  MI_EWORKSPACE    = -3, /* workspace too small                                   */
  MI_ELAUNCH       = -4, /* hipLaunchKernel reported an error                     */
  MI_ERUNTIME      = -5  /* a HIP runtime call (allocation, IPC) failed           */
};

#define MI_HEAD_DIM        128  /* head_dim of the fragment-native kernels (all Qwen3 sizes); 64: mi_*_plain */
#define MI_KV_TILE_TOKENS  16
#define MI_KV_TILE_ELEMS   2048

/* ---- library info -------------------------------------------------------- */
const char* mi_strerror(int code);
/* "mi355_nanovllm <semver> gfx950" */
const char* mi_version(void);
/* hipGetLastError() text of the most recent MI_ELAUNCH on this thread. */
const char* mi_last_launch_error(void);

/* ---- process-wide tuning knobs ------------------------------------------------
 * The entry points never read the process environment: a kernel choice that is not implied by the arguments is
 * either an explicit parameter of an `_ex` entry point or one of these knobs.  mi_set_tuning() is host-side,
 * takes effect for launches issued after it returns (a captured hipGraph keeps what it was captured with) and is
 * the ONLY hidden state of the library; every default is the measured-best path.  Returns MI_EINVAL for an unknown
 * knob or a value outside its range.  (The Python binding maps its MI355_* A/B environment switches onto these at
 * import time: nanovllm/_C.py.) */
enum mi_tuning_knob {
  MI_TUNE_ATTN_PIPE = 0,          /* mi_paged_attn_decode: 1 (default) = 8 waves x two chunks in flight, 0 = round-1 16-wave form */
  MI_TUNE_ATTN_RESOLVE = 1,       /* decode attention resolves its block-id run at kernel start (default 0: measured slower)    */
  MI_TUNE_NORM_WPR = 2,           /* mi_add_rmsnorm_splitk: waves per row for <= 64 rows: 4 (default), 2, 1                     */
  MI_TUNE_ROPE_BLOCK64 = 3,       /* mi_qknorm_rope_store: one-wave workgroups for <= 64 tokens (default 1)                     */
  MI_TUNE_PLAIN_SPLIT_TARGET = 4, /* mi_paged_attn_decode_plain: workgroups aimed for when contexts are split (default 512)     */
  MI_TUNE_PREFILL_P_SPLIT = 5,    /* prefill attention: probabilities as bf16 hi + lo (1) instead of one bf16 (default 0)       */
  MI_TUNE_GEMM_PIPE = 6,          /* streaming GEMMs: double-buffered K loop for K-slices of two or more blocks (default 1)     */
  MI_TUNE_COUNT = 7
};
int mi_set_tuning(int knob, int value);
/* current value, or INT32_MIN for an unknown knob */
int mi_get_tuning(int knob);

/* ---- KV cache: layout helpers ------------------------------------------- */
/* Element offset of logical (slot_in_block, head, d) inside one block of the
 * K (is_v=0) or V (is_v=1) cache.  Host-side helper used by tests. */
int64_t mi_kv_elem_offset(int is_v, int slot_in_block, int kv_head, int d,
                          int n_kv_heads, int block_size);

/* ---- KV scatter (reference: layers/attention.py:22-35) ------------------- */
/* Prefill scatter, stands in for torch_npu._npu_reshape_and_cache
 * (attention.py:25-30): cache[slot_flat[t]] = k[t], v[t] for every token t.
 * k/v are [n_tokens, n_kv_heads, 128] with row strides given in elements
 * (v is a strided view of the fused qkv output, qwen3.py:79).
 * slot_flat[t] = block_id*block_size + offset (model_runner.py:263-270); <0 skips. */
int mi_reshape_and_cache(const mi_bf16* k, const mi_bf16* v,
                         int64_t k_row_stride, int64_t v_row_stride,
                         mi_bf16* k_cache, mi_bf16* v_cache,
                         const int32_t* slot_flat, int n_tokens,
                         int n_kv_heads, int head_dim, int block_size,
                         mi_stream stream);

/* Decode scatter, stands in for the two torch_npu.scatter_update_ calls
 * (attention.py:32-35): cache[idx[b][0], idx[b][1]] = k[b], v[b].
 * slot_2d is [batch][2] int32 = {block_id, offset} (model_runner.py:301,353). */
int mi_scatter_update_kv(const mi_bf16* k, const mi_bf16* v,
                         int64_t k_row_stride, int64_t v_row_stride,
                         mi_bf16* k_cache, mi_bf16* v_cache,
                         const int32_t* slot_2d, int batch,
                         int n_kv_heads, int head_dim, int block_size,
                         mi_stream stream);

/* Test/debug helper: read rows back from the fragment-native cache into the
 * reference's logical [n, n_kv_heads*128] layout (one row per flat slot). */
int mi_kv_cache_gather(const mi_bf16* cache, int is_v,
                       const int32_t* slot_flat, int n,
                       mi_bf16* out, int n_kv_heads, int head_dim, int block_size,
                       mi_stream stream);

/* ---- paged attention ------------------------------------------------------ */
/* Bytes of workspace mi_paged_attn_decode needs for (batch, n_q_heads). */
size_t mi_paged_attn_decode_workspace(int batch, int n_q_heads);

/* Decode attention, stands in for npu_fused_infer_attention_score_v2 "BNSD"
 * with block_table (attention.py:63-76) and its torchair twin (:79-93).
 * q [batch, n_q_heads, 128] (row stride q_row_stride elements per token),
 * block_table [batch][table_stride] int32 (-1 padded, model_runner.py:231-236),
 * context_lens [batch] int32, includes the token just written (:351);
 * rows with context_len <= 0 (graph padding, :305) produce zeros.
 * out [batch, n_q_heads*128] bf16. */
int mi_paged_attn_decode(const mi_bf16* q, int64_t q_row_stride,
                         const mi_bf16* k_cache, const mi_bf16* v_cache,
                         const int32_t* block_table, int table_stride,
                         const int32_t* context_lens,
                         mi_bf16* out, void* workspace, size_t ws_bytes,
                         int batch, int n_q_heads, int n_kv_heads, int head_dim,
                         int block_size, float scale, mi_stream stream);

/* mi_paged_attn_decode with the step's q_norm / k_norm / RoPE (qwen3.py:83-88) and the scatter of the
 * new token's K / V row (attention.py:32-35) folded into the same launch: reads the packed qkv rows of
 * QKVParallelLinear ([batch][(n_q_heads + 2 n_kv_heads) * 128], linear.py:117-126) directly.  Output and
 * cache contents are bit-identical to mi_qknorm_rope_store followed by mi_paged_attn_decode.
 * The token of row b is attended at position context_lens[b] - 1 and stored at slot_2d[b] = {block id,
 * offset} (model_runner.py:301,353; both name the same slot); rows with context_len <= 0 (graph padding,
 * :303-311) produce zeros and do NOT write their dummy slot (the reserved block is never read).
 * q_w / k_w may be NULL (attention_bias=true models skip the norms, qwen3.py:70).  k_cache / v_cache are
 * read and written. */
int mi_paged_attn_decode_fused(const mi_bf16* qkv, int64_t qkv_row_stride,
                               const mi_bf16* q_w, const mi_bf16* k_w, float eps,
                               const int64_t* positions, const float* cos_sin,
                               const int32_t* slot_2d,
                               mi_bf16* k_cache, mi_bf16* v_cache,
                               const int32_t* block_table, int table_stride,
                               const int32_t* context_lens,
                               mi_bf16* out, void* workspace, size_t ws_bytes,
                               int batch, int n_q_heads, int n_kv_heads, int head_dim,
                               int block_size, float scale, mi_stream stream);

/* Instrumented form of mi_paged_attn_decode_fused (tools/attn_timeline.py; n_q_heads / n_kv_heads == 2 only): the same
 * results, and stamps[batch * n_kv_heads * splits][8 waves][8] (uint64, device memory) receives every wave's
 * s_memrealtime (the chip-wide 100 MHz clock) at: 0 entry, 1 context length known, 2 its first two K/V chunks requested, 3 the step's q / k / v rows
 * published (workgroup barrier), 4 first chunk consumed, 5 its run of the context attended, 6 all waves arrived,
 * 7 merged and stored. */
int mi_paged_attn_decode_fused_ex(const mi_bf16* qkv, int64_t qkv_row_stride,
                                  const mi_bf16* q_w, const mi_bf16* k_w, float eps,
                                  const int64_t* positions, const float* cos_sin,
                                  const int32_t* slot_2d,
                                  mi_bf16* k_cache, mi_bf16* v_cache,
                                  const int32_t* block_table, int table_stride,
                                  const int32_t* context_lens,
                                  mi_bf16* out, void* workspace, size_t ws_bytes,
                                  int batch, int n_q_heads, int n_kv_heads, int head_dim,
                                  int block_size, float scale, uint64_t* stamps, mi_stream stream);

/* Tuning entry point (tools/attn_exp.py): mi_paged_attn_decode with an explicit number of context splits
 * per (sequence, kv head) (0 = automatic) and explicit element strides of the cache
 * (block, kv head, 16-token tile). */
int mi_paged_attn_decode_ex(const mi_bf16* q, int64_t q_row_stride,
                            const mi_bf16* k_cache, const mi_bf16* v_cache,
                            const int32_t* block_table, int table_stride,
                            const int32_t* context_lens,
                            mi_bf16* out, void* workspace, size_t ws_bytes,
                            int batch, int n_q_heads, int n_kv_heads, int head_dim,
                            int block_size, float scale, int num_splits,
                            int64_t stride_block, int64_t stride_head, int64_t stride_tile,
                            mi_stream stream);

/* Prefill attention, stands in for npu_fused_infer_attention_score_v2 "TND"
 * sparse_mode=3 (attention.py:47-59): per-sequence causal GQA attention.
 * K/V are read back from the paged cache that mi_reshape_and_cache has just
 * filled for this step (bit-identical to the step's k,v, since the scatter is a
 * pure copy), through block_table [n_seqs][table_stride].
 * q [T, n_q_heads, 128]; cu_seqlens_q [n_seqs+1] int32 (model_runner.py:254-258);
 * kv_lens [n_seqs] int32 = tokens of each sequence present in the cache
 * (== query length in the reference, which recomputes everything :248-249;
 *  larger when a cached prefix is skipped).  Query token i of a sequence sits
 * at position kv_len - q_len + i.  out [T, n_q_heads*128]. */
int mi_paged_attn_prefill(const mi_bf16* q, int64_t q_row_stride,
                          const mi_bf16* k_cache, const mi_bf16* v_cache,
                          const int32_t* block_table, int table_stride,
                          const int32_t* cu_seqlens_q, const int32_t* kv_lens,
                          int n_seqs, int max_seqlen_q,
                          mi_bf16* out,
                          int n_q_heads, int n_kv_heads, int head_dim,
                          int block_size, float scale, mi_stream stream);

/* The same with the query side of qwen3.py:79-88 folded into the Q-operand load: q points at the RAW packed qkv
 * rows (linear.py:117-126); per-head RMSNorm with q_w (NULL: no norm) and NeoX RoPE at positions[token] are
 * applied in registers, bit-identical to what mi_qknorm_rope_store writes to q_out.  K and V must already be in
 * the cache: call mi_qknorm_rope_store with q_out = NULL (K / V only) first. */
int mi_paged_attn_prefill_fused(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w, float eps,
                                const int64_t* positions, const float* cos_sin,
                                const mi_bf16* k_cache, const mi_bf16* v_cache,
                                const int32_t* block_table, int table_stride,
                                const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads,
                                int head_dim, int block_size, float scale, mi_stream stream);
/* Tuning / stress-test form of mi_paged_attn_prefill_fused.  variant 0 = the product kernel; 4 = probabilities
 * as bf16 hi + lo (what MI_TUNE_PREFILL_P_SPLIT selects process-wide; every group size).  Schedule variants, for
 * n_q_heads / n_kv_heads == 2 only (the bench model; MI_EUNSUPPORTED otherwise), same results as variant 0 bit for
 * bit (tests): 1 = the first two K/V chunks requested ahead of the Q preparation; 2 = one workgroup barrier per two
 * chunks; 8 = round 3's V operand reads (ds_read2st64_b64: LDS bank conflicts); 16 = round 3's request path (table
 * read + divisions in front of every chunk request); 24 = 8 + 16; 28 = the round-3 kernel as a whole (4 + 8 + 16);
 * 32 = round 5's softmax (the chunk maximum BEFORE the exponentials; the product tests the lane's sum of exponentials
 * instead and computes a maximum only when the reference point has to move: the same results wherever both rescale in
 * the same chunks); 64 = round 5's request addressing (per-lane 64-bit pointers + global_load_lds; the product keeps the
 * tile address on the scalar unit: buffer_load ... lds); 96 = the round-5 kernel as a whole.  The round-3 forms
 * (8, 16, 24, 28) include 96: they are the kernels that were measured then. */
int mi_paged_attn_prefill_fused_ex(const mi_bf16* qkv, int64_t qkv_row_stride, const mi_bf16* q_w, float eps,
                                   const int64_t* positions, const float* cos_sin, const mi_bf16* k_cache,
                                   const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                   const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                   int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads, int head_dim,
                                   int block_size, float scale, int variant, mi_stream stream);

/* ---- normalisation (reference: layers/layernorm.py) ----------------------- */
/* RMSNorm.rms_forward (layernorm.py:16-25):
 *   y = bf16( bf16(x32 * rsqrt(mean(x32^2)+eps)) * w )   — two roundings.
 * x is [outer][inner][cols]; row (o,i) starts at x + o*x_outer_stride + i*cols
 * (so the per-head q/k norm of qwen3.py:83-85 runs on strided views of qkv).
 * y is contiguous [outer*inner][cols].  cols % 8 == 0, cols <= 8192. */
int mi_rmsnorm(const mi_bf16* x, int64_t x_outer_stride, const mi_bf16* w,
               mi_bf16* y, int outer, int inner, int cols, float eps,
               mi_stream stream);

/* RMSNorm.add_rms_forward (layernorm.py:27-38):
 *   s = x32 + r32; residual_out = bf16(s); y = norm(s) as above (variance from
 *   the un-rounded s).  x, residual, y, residual_out are contiguous [rows][cols];
 *   y may alias x and residual_out may alias residual. */
int mi_add_rmsnorm(const mi_bf16* x, const mi_bf16* residual, const mi_bf16* w,
                   mi_bf16* y, mi_bf16* residual_out, int rows, int cols,
                   float eps, mi_stream stream);

/* ---- rotary embedding (reference: layers/rotary_embedding.py) ------------- */
/* RotaryEmbedding.forward (rotary_embedding.py:37-47), NeoX halves, fp32 math:
 *   y1 = x1*cos - x2*sin ; y2 = x2*cos + x1*sin ; cast to bf16 (:6-14).
 * cos_sin [max_pos][128] fp32 = cat(cos[64], sin[64]) (:29-35).
 * positions [n_tokens] int64.  q/k rows: [n_tokens][n_heads][128], token stride
 * given in elements; outputs are contiguous [n_tokens][n_heads][128]. */
int mi_rope(const int64_t* positions, const float* cos_sin,
            const mi_bf16* q, int64_t q_row_stride, int n_q_heads,
            const mi_bf16* k, int64_t k_row_stride, int n_kv_heads,
            mi_bf16* q_out, mi_bf16* k_out, int n_tokens, int head_dim,
            mi_stream stream);

/* Fused q_norm + k_norm + RoPE + KV scatter for one step (qwen3.py:79-88 +
 * attention.py:22-35), reading the packed qkv row [q | k | v] of
 * QKVParallelLinear (linear.py:117-126).  Rounding points are exactly those of
 * the unfused sequence mi_rmsnorm -> mi_rope -> mi_reshape_and_cache /
 * mi_scatter_update_kv, so results are bit-identical to it.
 * q_w/k_w may be NULL (attention_bias=true models skip the norms, qwen3.py:70).
 * slots: flat [n_tokens] if slot_is_2d == 0, else [n_tokens][2].
 * q_out may be NULL for prefill-sized calls (flat slots, n_tokens >= 64): only K and V are processed and
 * mi_paged_attn_prefill_fused prepares the queries itself. */
int mi_qknorm_rope_store(const mi_bf16* qkv, int64_t qkv_row_stride,
                         const mi_bf16* q_w, const mi_bf16* k_w, float eps,
                         const int64_t* positions, const float* cos_sin,
                         mi_bf16* q_out, mi_bf16* k_cache, mi_bf16* v_cache,
                         const int32_t* slots, int slot_is_2d,
                         int n_tokens, int n_q_heads, int n_kv_heads,
                         int head_dim, int block_size, mi_stream stream);

/* ---- activation (reference: layers/activation.py:10-12) ------------------- */
/* SiluAndMul: out[t][i] = bf16( bf16(silu(x[t][i])) * x[t][inter+i] ). */
int mi_silu_mul(const mi_bf16* x, mi_bf16* out, int rows, int inter,
                mi_stream stream);

/* ---- linear (reference: layers/linear.py:51,73,150; embed_head.py:61) ----- */
/* y[M][N] = x[M][K] @ w[N][K]^T (+ bias[N]); bf16 in, fp32 accumulate, one
 * rounding to bf16.  Weight-streaming MFMA kernel for the decode regime:
 * 1 <= M <= 512 (walked in chunks of 64 rows inside the launch), K % 32 == 0 (K % 256 == 0 fastest), N % 16 == 0.
 * x row stride = K, y row stride = N. */
int mi_gemm_bf16_skinny(const mi_bf16* x, const mi_bf16* w, const mi_bf16* bias,
                        mi_bf16* y, int M, int N, int K, mi_stream stream);

/* Fragment-native weight layout for the decode GEMMs: w_packed[N/16][K/32][64][8] with
 * w_packed[(tn*K/32 + tk)*512 + lane*8 + e] = w[(16 tn + lane%16)*K + 32 tk + 8 (lane/16) + e],
 * so one wavefront load is one contiguous 1 KiB MFMA A fragment.  Built once per weight
 * (after TP sharding); N % 16 == 0, K % 32 == 0. */
int mi_pack_weight(const mi_bf16* w, mi_bf16* w_packed, int N, int K, mi_stream stream);

/* The same product for ANY number of rows (reference: F.linear at linear.py:51,73,150 with prefill-sized
 * activations, every decode batch above 64 sequences): 256 x 256 x 64 MFMA tiles, both operands staged by
 * LDS-DMA (csrc/gemm_tile.hip).  x [M][K] with row stride ldx, w [N][K] row-major (the checkpoint layout,
 * no packed copy needed), y [M][N] with row stride ldy (strides in elements).
 * epilogue 0: y = x @ w^T (+ bias[N]);  epilogue 1: rows j and j + N/2 of w are the gate and up rows of
 * MergedColumnParallelLinear (linear.py:76-93) and y[M][N/2] = SiluAndMul (activation.py:10-12) of the product,
 * with the reference's three bf16 roundings.  K % 64 == 0, N % 4 == 0 (epilogue 1: N/2 % 128 == 0, no bias),
 * ldx % 8 == 0, ldy % 4 == 0.
 * Shapes whose tiles cannot fill the chip (few token rows x a narrow projection, e.g. 1024 x 1024 x 3072) are
 * computed as K slices into `workspace` (fp32, mi_gemm_bf16_workspace bytes; 0 = not needed for this shape) and
 * summed in slice order by a second launch - deterministic, one rounding.
 * The operands are addressed with 32-bit byte offsets: (N + 256) * K * 2 and (M + 256) * ldx * 2 must stay below 2^31.
 * mi_gemm_bf16_max_rows answers that contract for a caller (F.linear takes any shape: layers/linear.py walks longer
 * activations through the kernel in row pieces): the largest M one launch takes for this weight shape and row
 * stride, 0 if the weight shape itself is refused (then MI_EUNSUPPORTED from mi_gemm_bf16). */
int64_t mi_gemm_bf16_max_rows(int N, int K, int64_t ldx);
size_t mi_gemm_bf16_workspace(int M, int N, int K, int epilogue);
int mi_gemm_bf16(const mi_bf16* x, int64_t ldx, const mi_bf16* w, const mi_bf16* bias, mi_bf16* y, int64_t ldy,
                 int M, int N, int K, int epilogue, void* workspace, size_t ws_bytes, mi_stream stream);
/* The packed qkv projection of a prefill step with the K / V side of qwen3.py:79-90 + attention.py:55-58 in its
 * epilogue (round 4): qkv[M][N] = x @ w^T (+ bias) as mi_gemm_bf16, but only the q heads (columns
 * [0, n_q_heads * 128)) are written to `qkv`; every k head is k-normed (k_w, or NULL: none), rotated (positions,
 * cos_sin as mi_qknorm_rope_store) and stored into k_cache, every v head into v_cache, at `slots` (flat int32 [M]:
 * block * block_size + offset; negative = not stored) in the fragment-native tile layout - the same bits
 * mi_gemm_bf16 + mi_qknorm_rope_store(q_out = NULL) leave in the caches (tests/test_gemm_qkv_store_gpu.py).
 * The attention that follows reads q from the packed rows (mi_paged_attn_prefill_fused).
 * MI_EUNSUPPORTED unless head_dim == 128, N == (n_q_heads + 2 n_kv_heads) * 128, N % 256 == 0, ldy % 8 == 0,
 * block_size % 16 == 0 and (M, N) is a large-M shape (>= 256 tiles of 256 x 256): the caller then takes the two
 * launches. */
int mi_gemm_bf16_qkv_store(const mi_bf16* x, int64_t ldx, const mi_bf16* w, const mi_bf16* bias, mi_bf16* qkv,
                           int64_t ldy, int M, int N, int K, const mi_bf16* k_w, float eps, const int64_t* positions,
                           const float* cos_sin, mi_bf16* k_cache, mi_bf16* v_cache, const int32_t* slots,
                           int n_q_heads, int n_kv_heads, int head_dim, int block_size, mi_stream stream);

/* Tuning form of mi_gemm_bf16 (tools/gemm_bench.py): variant = 16 * prefetch depth + schedule flags, see
 * csrc/gemm_tile.hip; MI_EUNSUPPORTED for variants that are not compiled. */
int mi_gemm_bf16_ex(const mi_bf16* x, int64_t ldx, const mi_bf16* w, mi_bf16* y, int64_t ldy, int M, int N, int K,
                    int variant, mi_stream stream);

/* As mi_gemm_bf16_skinny on packed weights.  epilogue 0: y[M][N] (+bias).
 * epilogue 1 (SiluAndMul fused, activation.py:10-12 on top of MergedColumnParallelLinear,
 * linear.py:76-93): w is gate|up stacked, y[M][N/2] = bf16(bf16(silu(bf16 g)) * bf16 u)
 * with g = row j, u = row j + N/2 of the product — the unfused rounding points. */
int mi_gemm_bf16_packed(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* bias,
                        mi_bf16* y, int M, int N, int K, int epilogue, mi_stream stream);

/* RowParallelLinear.forward without the all-reduce (linear.py:149-151) for the small-N projections
 * (o_proj, down_proj) under tensor parallelism: y[M][N] = bf16(x[M][K] @ w[N][K]^T), this rank's bf16
 * partial sums as complete rows from N / 4 workgroups of four output features each (every CU streams
 * weights; the all-reduce wants bf16 rows, so split-K partials are no option).  A workgroup reads ALL of
 * x: good for the short K of a TP shard, not for TP = 1 (there split-K + mi_add_rmsnorm_splitk moves a
 * quarter of the activation bytes through each CU and measured faster).  Weight layout
 * w_packed4[N/4][K/32][4][4][8]:
 *   w_packed4[((tn*K/32 + tk)*16 + g*4 + n)*8 + e] = w[(4 tn + n)*K + 32 tk + 8 g + e].
 * 1 <= M <= 512 (chunks of 64 rows), N % 4 == 0, K % 32 == 0. */
int mi_pack_weight_rows4(const mi_bf16* w, mi_bf16* w_packed4, int N, int K, mi_stream stream);
int mi_gemm_bf16_rows4(const mi_bf16* x, const mi_bf16* w_packed4, mi_bf16* y, int M, int N, int K,
                       mi_stream stream);

/* Split-K over workgroups for the small-N row-parallel projections (o_proj, down_proj):
 * partials[ksplit][M][N] fp32, summed in split order and rounded to bf16 by the consumer
 * mi_add_rmsnorm_splitk — together bit-identical to mi_gemm_bf16_packed + mi_add_rmsnorm
 * up to fp32 summation order. 1 <= ksplit <= 16, K % (32*ksplit) == 0. */
int mi_gemm_bf16_packed_splitk(const mi_bf16* x, const mi_bf16* w_packed, float* partials,
                               int M, int N, int K, int ksplit, mi_stream stream);

/* mi_add_rmsnorm whose x is bf16(sum_s partials[s]) (RowParallelLinear output, linear.py:150
 * followed by RMSNorm.add_rms_forward, layernorm.py:27-38). */
int mi_add_rmsnorm_splitk(const float* partials, int nsplit, const mi_bf16* residual,
                          const mi_bf16* w, mi_bf16* y, mi_bf16* residual_out, int rows,
                          int cols, float eps, mi_stream stream);

/* Instrumented forms of the decode chain's launches (tools/chain_timeline.py; the same results, separate kernel
 * instantiations): every wave writes s_memrealtime (the chip-wide 100 MHz clock, comparable between launches) at
 * entry / loads issued / data arrived / sums in LDS / barrier passed / stores issued / stores acknowledged into
 * stamps[workgroup][wave][8] (uint64).  Workgroup = blockIdx.x + gridDim.x * blockIdx.y.
 * mi_gemm_bf16_packed_ex: ksplit 0 = mi_gemm_bf16_packed without bias (epilogue 0 / 1: stamps[N/16 or N/32][16][8]),
 * ksplit > 0 = mi_gemm_bf16_packed_splitk (stamps[N/16 * ksplit][K / ksplit / 64][8]); 17..32 rows and K-slices of 64
 * per wave on 8, 12 or 16 waves only (the Qwen3-0.6B decode chain), MI_EUNSUPPORTED otherwise.
 * mi_add_rmsnorm_splitk_ex: nsplit 4, <= 64 rows of <= 1024 columns (stamps[rows][4][8]). */
int mi_gemm_bf16_packed_ex(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, float* partials, int M, int N,
                           int K, int epilogue, int ksplit, uint64_t* stamps, mi_stream stream);
int mi_add_rmsnorm_splitk_ex(const float* partials, int nsplit, const mi_bf16* residual, const mi_bf16* w,
                             mi_bf16* y, mi_bf16* residual_out, int rows, int cols, float eps,
                             uint64_t* stamps, mi_stream stream);

/* ---- the decode chain in five launches per layer (csrc/gemm_chain5_kernel.hpp) ----------------------------
 * RowParallelLinear.forward (linear.py:149-153) + the residual add and the variance of RMSNorm.add_rms_forward
 * (layernorm.py:27-33) in ONE launch, and the rest of add_rms_forward (layernorm.py:34-38) in the operand load of the
 * NEXT projection (linear.py:72-73): the two mi_add_rmsnorm_splitk launches of a layer disappear.
 *
 * mi_gemm_bf16_rowstat: y = bf16(x[M][K] @ w[N][K]^T) (packed weights, as mi_gemm_bf16_packed);
 *   s_out[M][N] fp32 = float(y) + float(residual)   (the un-rounded sum the reference normalises),
 *   residual_out[M][N] = bf16(s_out),
 *   stat[M][N/16]: per 16-feature tile the sum of s_out^2 in fixed order.
 *   A workgroup owns 16 features of 8 rows over the whole K: nothing is summed across workgroups.
 *   ksplit names the split-K geometry (mi_gemm_bf16_packed_splitk) whose fp32 summation order is reproduced - K / 64
 *   wave slices of 64 in ksplit runs - so that y has that path's bits.  K % 64 == 0, N % 16 == 0, (K / 64) % ksplit == 0,
 *   1 <= ksplit <= 16.
 * mi_gemm_bf16_normed: y = x @ w^T (epilogue 0) or SiluAndMul(x @ w^T) (epilogue 1) as mi_gemm_bf16_packed, with
 *   x[M][K] = bf16(bf16(s * rstd) * norm_w), rstd = 1 / sqrt(sum(stat row) / K + eps) - never written to memory: every
 *   wave builds its own K-slice in LDS.  nstat = K / 16 partials per row, summed in the order of the decode-sized
 *   mi_add_rmsnorm_splitk kernel at K = 1024 (the two chains then agree bit for bit).  Up to 32 rows on sixteen
 *   waves (K a multiple of 1024, or K / 64 in {12, 16}), up to 64 on eight / four; MI_EUNSUPPORTED otherwise.
 * mi_norm_from_stat: the model's final norm y[rows][cols] from (s, stat).
 * _ex: instrumented forms (stamps[workgroup][wave][8] as above; the Qwen3-0.6B decode chain's geometries only). */
int mi_gemm_bf16_rowstat(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* residual, mi_bf16* residual_out,
                         float* s_out, float* stat, int M, int N, int K, int ksplit, mi_stream stream);
int mi_gemm_bf16_normed(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                        const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K, int epilogue, mi_stream stream);
int mi_norm_from_stat(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps, mi_bf16* y,
                      int rows, int cols, mi_stream stream);
int mi_gemm_bf16_rowstat_ex(const mi_bf16* x, const mi_bf16* w_packed, const mi_bf16* residual, mi_bf16* residual_out,
                            float* s_out, float* stat, int M, int N, int K, int ksplit, uint64_t* stamps,
                            mi_stream stream);
int mi_gemm_bf16_normed_ex(const float* s, const float* stat, int nstat, const mi_bf16* norm_w, float eps,
                           const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K, int epilogue, uint64_t* stamps,
                           mi_stream stream);


/* ---- plain-layout attention (csrc/attn_plain.hip) -------------------------
 * The same operators (attention.py:22-93, rotary_embedding.py:6-14) for the head geometries the fragment-native
 * kernels above are not built for: head_dim 64 and GQA groups that are not a power of two (Llama-3.2-1B, Qwen2-0.5B,
 * Qwen2.5-7B - models the reference's README benchmarks).  Cache layout of this family:
 *     k_cache / v_cache  [num_blocks][n_kv_heads][block_size][head_dim]  bf16, any block_size.
 * head_dim 64 or 128, n_q_heads / n_kv_heads <= 8 (any integer). */
/* _npu_reshape_and_cache / scatter_update_ (attention.py:25-35): slots flat int32 (slots_2d = 0; negative: skip)
 * or [n_tokens][2] (block, offset) pairs (slots_2d = 1). */
int mi_kv_store_plain(const mi_bf16* k, const mi_bf16* v, int64_t k_row_stride, int64_t v_row_stride,
                      mi_bf16* k_cache, mi_bf16* v_cache, const int32_t* slots, int slots_2d, int n_tokens,
                      int n_kv_heads, int head_dim, int block_size, mi_stream stream);
/* apply_rotary_emb (rotary_embedding.py:6-14,37-47) for any head_dim % 16 == 0: q [T][Hq][D] and k [T][Hkv][D]
 * rows of the given token strides -> contiguous q_out / k_out; cos_sin [max_pos][D] fp32 = (cos | sin). */
int mi_rope_plain(const int64_t* positions, const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k,
                  int64_t k_row_stride, const float* cos_sin, mi_bf16* q_out, mi_bf16* k_out, int n_tokens,
                  int n_q_heads, int n_kv_heads, int head_dim, mi_stream stream);
/* decode attention (attention.py:63-93): out [batch][n_q_heads * head_dim]; workspace of
 * mi_paged_attn_decode_plain_workspace bytes (context splits of small batches). */
size_t mi_paged_attn_decode_plain_workspace(int batch, int n_q_heads, int head_dim);
int mi_paged_attn_decode_plain(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                               const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                               const int32_t* context_lens, mi_bf16* out, void* workspace, size_t ws_bytes,
                               int batch, int n_q_heads, int n_kv_heads, int head_dim, int block_size,
                               float scale, mi_stream stream);
/* causal prefill attention through the block table (attention.py:46-59; kv_lens >= query lengths: cached prefixes) */
int mi_paged_attn_prefill_plain(const mi_bf16* q, int64_t q_row_stride, const mi_bf16* k_cache,
                                const mi_bf16* v_cache, const int32_t* block_table, int table_stride,
                                const int32_t* cu_seqlens_q, const int32_t* kv_lens, int n_seqs,
                                int max_seqlen_q, mi_bf16* out, int n_q_heads, int n_kv_heads, int head_dim,
                                int block_size, float scale, mi_stream stream);

/* ---- embedding / head (reference: layers/embed_head.py) ------------------- */
/* VocabParallelEmbedding.forward (embed_head.py:34-42): out[t] = w[ids[t]-vocab_start]
 * if vocab_start <= ids[t] < vocab_start+vocab_local else 0 (TP mask). */
int mi_embedding(const int64_t* ids, const mi_bf16* w, mi_bf16* out,
                 int n_tokens, int hidden, int64_t vocab_start, int64_t vocab_local,
                 mi_stream stream);

/* The same lookup for a decode step queued before the previous step's tokens
 * have reached the host: row i takes its id from prev_tokens[src_rows[i]] (the
 * previous step's sampled tokens, device memory) when src_rows[i] >= 0, from
 * ids[i] otherwise.  Lets the engine launch step k+1 behind step k without a
 * host round trip (model_runner.py:344-366 builds input_ids on the host). */
int mi_embedding_from_prev(const int64_t* ids, const int32_t* src_rows,
                           const int64_t* prev_tokens, const mi_bf16* w, mi_bf16* out,
                           int n_tokens, int hidden, int64_t vocab_start,
                           int64_t vocab_local, mi_stream stream);

/* A step's metadata (or its sampled tokens) between PINNED host memory and device memory as a kernel on the step's
 * stream: dst[0, nbytes) = src[0, nbytes), 16-byte pieces, both pointers 16-byte aligned, nbytes % 16 == 0, at most
 * 256 MiB.  Either side may be device-accessible pinned host memory (hipHostMalloc / torch pin_memory).  Replaces the
 * H2D copies of model_runner.py:344-366 (prepare_decode: five per step; here one buffer per step) without the copy
 * engine's cross-queue signals between two decode graphs. */
int mi_stage_copy(void* dst, const void* src, int64_t nbytes, mi_stream stream);

/* ParallelLMHead last-token select in prefill (embed_head.py:58-60):
 * out[s] = x[cu_seqlens_q[s+1]-1]. */
int mi_gather_last_tokens(const mi_bf16* x, const int32_t* cu_seqlens_q,
                          mi_bf16* out, int n_seqs, int hidden, mi_stream stream);

/* ---- sampling (reference: layers/sampler.py:9-17) ------------------------- */
/* Greedy token = lowest index of the row maximum of logits[rows][vocab] (the
 * deterministic path BASELINE config 1 needs; the reference itself only has
 * multinomial sampling).  out [rows] int64. */
int mi_argmax(const mi_bf16* logits, int64_t row_stride, int64_t* out,
              int rows, int vocab, mi_stream stream);

/* Sampler.forward: sample from softmax(logits.float()/temperature) per row via
 * the Gumbel-max identity with a counter-based generator keyed by
 * (seed, step, row, column).  temperatures [rows] fp32; rows whose temperature
 * is <= 0 are greedy.  out [rows] int64. */
int mi_sample(const mi_bf16* logits, int64_t row_stride, const float* temperatures,
              int64_t* out, int rows, int vocab, uint64_t seed, uint64_t step,
              mi_stream stream);

/* The head GEMM and the sampler in one pass (decode, one GPU: embed_head.py:57-61
 * followed by sampler.py:9-17).  y = x @ W^T as mi_gemm_bf16_packed, and every
 * workgroup also reports the best sampling key of its columns per row - the key
 * mi_argmax / mi_sample would form from the ROUNDED logit, bit for bit - into
 * candidates [M][mi_gemm_pick_groups(M, N, K, fp8)] x 8 bytes.  mi_pick_final reduces
 * them to out[rows] (ties -> lowest column): the same tokens as mi_sample over y
 * with the same (seed, step), without reading y again.  rng points to
 * {seed, step} in DEVICE memory so that a captured graph picks up the step the
 * host wrote before replaying it.  temperatures may be NULL (all rows greedy). */
int mi_gemm_pick_groups(int M, int N, int K, int fp8_weights);
int mi_gemm_bf16_packed_pick(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y,
                             int M, int N, int K, const float* temperatures,
                             const uint64_t* rng, void* candidates, mi_stream stream);
/* Tensor-parallel form: w_packed is this rank's vocabulary shard (rows col_offset .. col_offset + N of the whole
 * head, embed_head.py:9-31); the sampler's noise is keyed by the GLOBAL column and the candidates carry global
 * columns.  mi_pick_final_pairs reduces them to this rank's best {key bits u32, column u32} per row;
 * mi_pick_exchange (below, with the exchange kernels) picks over the ranks. */
int mi_gemm_bf16_packed_pick_shard(const mi_bf16* x, const mi_bf16* w_packed, mi_bf16* y, int M, int N, int K,
                                   const float* temperatures, const uint64_t* rng, void* candidates,
                                   int col_offset, mi_stream stream);
int mi_pick_final_pairs(const void* candidates, int n_groups, int rows, void* pairs, mi_stream stream);
int mi_gemm_fp8w_packed_pick(const mi_bf16* x, const uint8_t* w_packed, const float* scale,
                             mi_bf16* y, int M, int N, int K, const float* temperatures,
                             const uint64_t* rng, void* candidates, mi_stream stream);
int mi_pick_final(const void* candidates, int n_groups, int rows, int64_t* out,
                  mi_stream stream);

/* ---- fp8 (e4m3) weights, bf16 activations (BASELINE.json configs[4]) --------
 * No reference semantics exist (the reference is bf16 only): y = x @ (w_q * scale[:, None])^T with
 * w_q OCP e4m3 values and one fp32 scale per weight row (output feature).  Decode GEMMs are bound
 * by the weight stream, not by MFMA rate, so the weights are dequantised to bf16 in registers
 * (exact) and multiplied on the bf16 MFMA: the activations are not quantised and the only error is
 * the weight quantisation.  Same shapes, epilogues and split-K partials as the bf16 entry points;
 * additionally K % 64 == 0.  w_q [N, K] row-major bytes -> mi_pack_weight_fp8 ->
 * [N/16][K/64][64 lanes][16 bytes] (a lane's 16 bytes = its A fragments of two 32-deep k-steps). */
int mi_pack_weight_fp8(const uint8_t* w_q, uint8_t* w_packed, int N, int K, mi_stream stream);
int mi_gemm_fp8w_packed(const mi_bf16* x, const uint8_t* w_packed, const float* scale, mi_bf16* y,
                        int M, int N, int K, int epilogue, mi_stream stream);
int mi_gemm_fp8w_packed_splitk(const mi_bf16* x, const uint8_t* w_packed, const float* scale,
                               float* partials, int M, int N, int K, int ksplit, mi_stream stream);

/* ---- mixture of experts (reference: models/qwen3_moe.py:125-185, BASELINE.json configs[3]) -------------
 * Qwen3MoeSparseMoeBlock.forward as five launches over expert-sorted (token, expert) pairs; P = n_tokens * top_k.
 * The reference loops over the selected experts in Python with three library GEMMs each (:171-184).
 * Expert weights: one fragment-native slab (mi_pack_weight layout) per expert, stacked: gate_up
 * [E][2*inter][hidden] with rows gate | up (MergedColumnParallelLinear, :104-108), down [E][hidden][inter];
 * under tensor parallelism `inter` is this rank's shard and the bf16 outputs of mi_moe_down are partial sums
 * to be all-reduced before mi_moe_combine (RowParallelLinear, :109-113). */
/* Whether mi_moe_gate_up / mi_moe_down have a kernel for experts of [2 * inter][hidden] / [hidden][inter]
 * (inter: this rank's share of moe_intermediate_size, qwen3_moe.py:104-115): MI_OK or MI_EUNSUPPORTED.  Host only. */
int mi_moe_shapes_supported(int hidden, int inter);

/* Routing (:153-161): softmax over the experts in fp32, top-k, renormalise by the top-k sum, cast to bf16.
 * topk_ids / topk_w [n_tokens][top_k], each token's picks in ASCENDING expert id (the order the reference's
 * expert loop adds them in, :171-172); equal probabilities: lower expert id first.  n_experts <= 512. */
int mi_moe_route(const mi_bf16* router_logits, int n_tokens, int n_experts, int top_k,
                 int32_t* topk_ids, mi_bf16* topk_w, mi_stream stream);
/* Group the pairs by expert: expert_offsets [n_experts + 1] (run of expert e = [off[e], off[e+1])),
 * pair_token [P] (token of the pair at each sorted position), pair_index [P] (its index t * top_k + slot in
 * topk_ids).  The order of the pairs INSIDE an expert's run is unspecified (it differs between calls and
 * between tensor-parallel ranks); nothing downstream depends on it. */
int mi_moe_sort(const int32_t* topk_ids, int n_tokens, int top_k, int n_experts,
                int32_t* expert_offsets, int32_t* pair_token, int32_t* pair_index, mi_stream stream);
/* act[pos][inter] = SiluAndMul(x[pair_token[pos]] @ W_gate_up[e]^T) for every pair of every expert, rounding
 * points of mi_gemm_bf16_packed(epilogue 1).  hidden = 64 * {1,2,3,4,6,8,12,16}, 128 * {5,10,12,16} or 256 * {10,16}. */
int mi_moe_gate_up(const mi_bf16* x, const mi_bf16* w_gate_up_packed, const int32_t* expert_offsets,
                   const int32_t* pair_token, mi_bf16* act, int n_experts, int hidden, int inter,
                   mi_stream stream);
/* y[pair_index[pos]][hidden] = bf16(act[pos] @ W_down[e]^T): the expert MLP's output (:121), stored at the
 * pair's own row t * top_k + slot - the same layout on every tensor-parallel rank, so that the ranks' partial
 * sums can be all-reduced row by row.  inter: same set as above. */
int mi_moe_down(const mi_bf16* act, const mi_bf16* w_down_packed, const int32_t* expert_offsets,
                const int32_t* pair_index, mi_bf16* y, int n_experts, int hidden, int inter,
                mi_stream stream);
/* out[t] = sum over the token's pairs j (ascending expert id) of bf16(y[t * top_k + j] * topk_w[t][j]), every
 * partial sum rounded to bf16 - `index_add_` into a bf16 tensor (:178-184). */
int mi_moe_combine(const mi_bf16* y, const mi_bf16* topk_w, mi_bf16* out,
                   int n_tokens, int top_k, int hidden, mi_stream stream);

/* ---- tensor-parallel exchange over xGMI -----------------------------------
 * Stands in for the HCCL all-reduce after every row-parallel projection and the
 * vocab-parallel embedding (linear.py:152-153, embed_head.py:41-42).  One process
 * per GPU; each rank owns one exchange REGION of uncached device memory that its
 * peers map through HIP IPC.  A SUM all-reduce of <= max_bytes is ONE kernel:
 * every workgroup pushes its slice of the local vector into slot [rank] of every
 * peer's region (xGMI is point-to-point: 7 direct writes, no ring), publishes a
 * per-slice epoch flag with a system-scope release, waits for the same slice's
 * flags of all peers and sums the `world` slots in rank order in fp32 - so every
 * rank gets bit-identical results and the launch needs no grid-wide sync, no
 * host interaction, and is hipGraph-capturable (epochs live in device memory and
 * advance on every launch or replay).  Slots are double-buffered by epoch parity.
 * The three init-time calls below are the only entry points of this library that
 * allocate or synchronise. */
typedef struct mi_comm mi_comm;
#define MI_COMM_MAX_WORLD   8
#define MI_IPC_HANDLE_BYTES 64
/* Bytes of one rank's region for vectors of up to max_bytes. */
size_t mi_comm_region_bytes(int world, size_t max_bytes);
/* Allocate + zero a region (hipExtMallocWithFlags, uncached) and export its IPC handle. */
int mi_comm_region_alloc(size_t bytes, void** region, void* ipc_handle /*[MI_IPC_HANDLE_BYTES]*/);
/* Map a peer's region from its IPC handle / unmap / free. */
int mi_comm_region_open(const void* ipc_handle, void** region);
int mi_comm_region_close(void* region);
int mi_comm_region_free(void* region);
/* regions[world]: regions[rank] is this rank's own allocation, the others are opened peers. */
int mi_comm_create(int rank, int world, void* const* regions, size_t max_bytes, mi_comm** out);
int mi_comm_destroy(mi_comm* comm);
/* out[i] = sum over ranks of in[i]; n bf16 elements, n % 8 == 0, 2n <= max_bytes; in == out allowed.
 * Every rank must issue the same sequence of calls.  A peer that does not show up within about
 * a minute raises the communicator's sticky timeout flag instead of hanging the GPU. */
int mi_allreduce_sum_bf16(mi_comm* comm, const mi_bf16* in, mi_bf16* out, int64_t n, mi_stream stream);
/* mi_allreduce_sum_bf16 over the ranks' x [rows, cols] followed by mi_add_rmsnorm(sum, residual, ...)
 * in ONE launch (one wave per row: push, flag, wait, sum, add, normalise) - the row-parallel seam of
 * qwen3.py:118-131 (o_proj / down_proj all-reduce, then the next layer norm).  Bit-identical to the
 * two-call sequence.  rows <= 64, 512 <= cols <= 8192, cols % 8 == 0. */
int mi_allreduce_add_rmsnorm(mi_comm* comm, const mi_bf16* x, const mi_bf16* residual,
                             const mi_bf16* weight, mi_bf16* out, mi_bf16* residual_out,
                             int rows, int cols, float eps, mi_stream stream);
/* The sampler under tensor parallelism, without the logits gather of embed_head.py:62-65: every rank contributes its
 * shard's best {key, token} per row (mi_pick_final_pairs), the pairs cross the exchange region (8 bytes per row and
 * rank), and EVERY rank writes the same tokens[rows] - largest key, ties to the lowest token id, i.e. what
 * sampler.py:9-17 on the gathered logits picks.  Capturable; rows <= 512 * 64. */
int mi_pick_exchange(mi_comm* comm, const void* pairs, int64_t* tokens, int rows, mi_stream stream);
/* mi_comm_status without a device synchronisation: the flag is copied to `timed_out_host` (pinned memory) behind the
 * work already queued on `stream`. */
int mi_comm_status_async(mi_comm* comm, int* timed_out_host, mi_stream stream);
/* Polls (~1 us each) a workgroup waits for a peer before it gives up; default 2^26 (about a minute).
 * Init-time call (synchronises the device). */
int mi_comm_set_spin_limit(mi_comm* comm, uint32_t polls);
/* Copies the sticky timeout flag to *timed_out (synchronises the device). */
int mi_comm_status(mi_comm* comm, int* timed_out);
/* What this rank's FIRST timed-out exchange was waiting for: info = {epoch, slice (row / slice index of the launch), peer
 * rank whose flag never arrived, the flag value seen instead (an older epoch: the peer never got there; a newer one:
 * the peer is ahead)}.  Synchronous; meaningful once mi_comm_status reports a time-out.  Bring-up aid. */
int mi_comm_timeout_info(mi_comm* comm, uint32_t info[4]);

/* ---- host-side hashing (reference: engine/block_manager.py:38-44) --------- */
/* xxh64 of `len` bytes with seed 0, optionally prefixed by the 8 little-endian
 * bytes of `prefix` (has_prefix != 0) — the chained block hash of
 * BlockManager.compute_hash.  The reference calls the third-party `xxhash`
 * package (pyproject.toml:17, unpinned); this is XXH64 as published. */
uint64_t mi_xxh64_chain(const void* data, size_t len, int has_prefix, uint64_t prefix);
/* out[i] = chained hash of block i of `n_blocks` consecutive full blocks of int64 token ids
 * (block 0 chained to `prefix` when has_prefix): BlockManager.allocate's loop (block_manager.py:62-71)
 * in one call. */
int mi_xxh64_chain_blocks(const int64_t* tokens, int n_blocks, int block_size, int has_prefix,
                          uint64_t prefix, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* MI355_NANOVLLM_H */
