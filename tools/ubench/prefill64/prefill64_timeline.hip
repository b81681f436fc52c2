// Per-segment cycle accounting of the 64-column prefill attention kernel (csrc/paged_attn_prefill64.hip built with
// STAMP = true): 16 sequences x T tokens of Qwen3-0.6B heads (16 q / 8 kv heads, head_dim 128), block size 16, random
// cache, fused Q preparation.  Prints the mean over waves of the cycles per stage in each segment and of the per-workgroup
// prologue / epilogue segments.  Build: tools/ubench/build.sh; run on the GPU box: tools/ubench/prefill64_timeline [T]
#include "paged_attn_prefill64.hip"

#include <stdio.h>
#include <stdlib.h>

#include <vector>
namespace mi {
int tuning(int) { return 0; }
int check_launch() { return hipGetLastError() == hipSuccess ? MI_OK : MI_ELAUNCH; }
}  // namespace mi
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 1024, n = 16, hq = 16, hkv = 8, bs = 16, D = 128, G = hq / hkv;
  const int nb = n * T / bs;
  const size_t cache_elems = (size_t)nb * hkv * 16 * D;
  std::vector<uint16_t> hc(cache_elems);
  srand(1);
  for (auto& v : hc) v = (uint16_t)(0x3c00 + (rand() & 0x3ff) - ((rand() & 1) << 15));  // +-[0.0078, 0.0156): finite bf16
  uint16_t *kc, *vc, *qkv, *out, *qw;
  CK(hipMalloc(&kc, cache_elems * 2)); CK(hipMalloc(&vc, cache_elems * 2));
  CK(hipMemcpy(kc, hc.data(), cache_elems * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(vc, hc.data(), cache_elems * 2, hipMemcpyHostToDevice));
  const int64_t qstride = (hq + 2 * hkv) * D;
  std::vector<uint16_t> hq_(n * (size_t)T * qstride);
  for (auto& v : hq_) v = (uint16_t)(0x3f00 + (rand() & 0xff) - ((rand() & 1) << 15));
  CK(hipMalloc(&qkv, hq_.size() * 2)); CK(hipMemcpy(qkv, hq_.data(), hq_.size() * 2, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, (size_t)n * T * hq * D * 2));
  std::vector<uint16_t> hw(D, 0x3f80);
  CK(hipMalloc(&qw, D * 2)); CK(hipMemcpy(qw, hw.data(), D * 2, hipMemcpyHostToDevice));
  std::vector<int32_t> tab(nb), cu(n + 1), kvl(n, T);
  for (int i = 0; i < nb; ++i) tab[i] = (int)(((int64_t)i * 7919) % nb);  // (nb = 2^k * 16...: 7919 is odd -> a permutation)
  for (int i = 0; i <= n; ++i) cu[i] = i * T;
  int32_t *dtab, *dcu, *dkvl;
  CK(hipMalloc(&dtab, nb * 4)); CK(hipMemcpy(dtab, tab.data(), nb * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dcu, (n + 1) * 4)); CK(hipMemcpy(dcu, cu.data(), (n + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dkvl, n * 4)); CK(hipMemcpy(dkvl, kvl.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<int64_t> pos(n * (size_t)T);
  for (size_t i = 0; i < pos.size(); ++i) pos[i] = i % T;
  int64_t* dpos; CK(hipMalloc(&dpos, pos.size() * 8)); CK(hipMemcpy(dpos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice));
  std::vector<float> cs((size_t)T * 128, 0.5f);
  float* dcs; CK(hipMalloc(&dcs, cs.size() * 4)); CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));

  const int tq_wg = 256 / G, n_qblocks = (T + tq_wg - 1) / tq_wg, n_pairs = n * hkv;
  const int wgs = (n_pairs + 7) / 8 * 8 * n_qblocks;
  unsigned long long* dst; CK(hipMalloc(&dst, (size_t)wgs * 4 * 16 * 8)); CK(hipMemset(dst, 0, (size_t)wgs * 4 * 16 * 8));
  const mi::QPrep qp{qw, dpos, dcs, 1e-6f};
  const float sl2 = 0.08838834764831845f * 1.4426950408889634f;
  auto stamped = mi::paged_attn_prefill64_kernel<2, true, true>;
  auto plain = mi::paged_attn_prefill64_kernel<2, true, false>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(stamped), hipFuncAttributeMaxDynamicSharedMemorySize, mi::P64_LDS_BYTES));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(plain), hipFuncAttributeMaxDynamicSharedMemorySize, mi::P64_LDS_BYTES));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int which = 0; which < 2; ++which)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      if (which == 0)
        hipLaunchKernelGGL(plain, dim3(wgs), dim3(256), mi::P64_LDS_BYTES, 0, qkv, qstride, kc, vc, dtab, nb / n, dcu, dkvl, out, hq,
                           hkv, 1, 0, sl2, n_qblocks, n_pairs, qp, nullptr);
      else
        hipLaunchKernelGGL(stamped, dim3(wgs), dim3(256), mi::P64_LDS_BYTES, 0, qkv, qstride, kc, vc, dtab, nb / n, dcu, dkvl, out, hq,
                           hkv, 1, 0, sl2, n_qblocks, n_pairs, qp, dst);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep == 2) printf("%s kernel, 16 x %d tokens: %.1f us per launch (%d workgroups)\n", which ? "stamped" : "plain", T, ms * 1e3, wgs);
    }
  std::vector<unsigned long long> h((size_t)wgs * 4 * 16);
  CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
  double sum[16] = {}; double stages = 0; int waves = 0;
  for (int w = 0; w < wgs * 4; ++w) {
    if (h[w * 16] == 0) continue;
    ++waves; stages += h[w * 16];
    for (int i = 1; i < 16; ++i) sum[i] += h[w * 16 + i];
  }
  printf("%d waves, %.1f stages per wave\n", waves, stages / waves);
  const char* seg[6] = {"", "wait + barrier", "request (DMA issue, next block id)", "mask (diagonal / masked-out chunks)",
                        "instruction stream of the stage", "test + pack"};
  double per_stage = 0;
  for (int i = 1; i <= 5; ++i) { printf("  per stage: %-40s %8.0f cycles\n", seg[i], sum[i] / stages); per_stage += sum[i] / stages; }
  printf("  per stage: %-40s %8.0f cycles\n", "total", per_stage);
  const char* wseg[6] = {"Q preparation (loads, norm, RoPE, -> a[128:191])", "first requests issued, chunk 0 landed", "scores of chunk 0",
                         "the loop", "last PV product + drain + barrier", "epilogue (transpose, stores)"};
  for (int i = 8; i <= 13; ++i) printf("  per workgroup: %-52s %8.0f cycles\n", wseg[i - 8], sum[i] / waves);
  return 0;
}
