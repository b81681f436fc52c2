// Library-level entry points of libmi355_nanovllm.so: error reporting, layout
// helper, and the host-side XXH64 used by the block manager's prefix hashing.
#include <limits.h>
#include <string.h>

#include <atomic>
#include <string>

#include "mi_common.hpp"

namespace mi {
static thread_local std::string g_last_launch_error;

// process-wide tuning knobs (include/mi355_nanovllm.h: mi_tuning_knob): {default, min, max}
static constexpr int kTuneSpec[MI_TUNE_COUNT][3] = {
    {1, 0, 1},       // MI_TUNE_ATTN_PIPE
    {0, 0, 1},       // MI_TUNE_ATTN_RESOLVE
    {4, 1, 4},       // MI_TUNE_NORM_WPR
    {1, 0, 1},       // MI_TUNE_ROPE_BLOCK64
    {512, 1, 4096},  // MI_TUNE_PLAIN_SPLIT_TARGET
    {0, 0, 1},       // MI_TUNE_PREFILL_P_SPLIT
    {1, 0, 1},       // MI_TUNE_GEMM_PIPE
};
static std::atomic<int> g_tuning[MI_TUNE_COUNT] = {
    {kTuneSpec[0][0]}, {kTuneSpec[1][0]}, {kTuneSpec[2][0]}, {kTuneSpec[3][0]}, {kTuneSpec[4][0]}, {kTuneSpec[5][0]},
    {kTuneSpec[6][0]}};

int tuning(int knob) { return g_tuning[knob].load(std::memory_order_relaxed); }

int check_launch() {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return MI_OK;
  g_last_launch_error = hipGetErrorString(e);
  return MI_ELAUNCH;
}
}  // namespace mi

extern "C" const char* mi_strerror(int code) {
  switch (code) {
    case MI_OK: return "ok";
    case MI_EINVAL: return "invalid argument (null/misaligned pointer or negative size)";
    case MI_EUNSUPPORTED: return "shape not supported by the compiled kernels";
    case MI_EWORKSPACE: return "workspace too small";
    case MI_ELAUNCH: return "kernel launch failed (see mi_last_launch_error)";
    case MI_ERUNTIME: return "HIP runtime call failed (allocation / IPC)";
    default: return "unknown error code";
  }
}

extern "C" int mi_set_tuning(int knob, int value) {
  if (knob < 0 || knob >= MI_TUNE_COUNT) return MI_EINVAL;
  if (value < mi::kTuneSpec[knob][1] || value > mi::kTuneSpec[knob][2]) return MI_EINVAL;
  if (knob == MI_TUNE_NORM_WPR && value == 3) return MI_EINVAL;
  mi::g_tuning[knob].store(value, std::memory_order_relaxed);
  return MI_OK;
}

extern "C" int mi_get_tuning(int knob) {
  if (knob < 0 || knob >= MI_TUNE_COUNT) return INT_MIN;
  return mi::tuning(knob);
}

extern "C" const char* mi_version(void) { return "mi355_nanovllm 0.1.0 gfx950"; }

extern "C" const char* mi_last_launch_error(void) { return mi::g_last_launch_error.c_str(); }

extern "C" int64_t mi_kv_elem_offset(int is_v, int slot_in_block, int kv_head, int d, int n_kv_heads,
                                     int block_size) {
  if (slot_in_block < 0 || slot_in_block >= block_size || kv_head < 0 || kv_head >= n_kv_heads || d < 0 ||
      d >= MI_HEAD_DIM || block_size % 16)
    return -1;
  const int tpb = block_size / 16;
  const int64_t base = ((int64_t)kv_head * tpb + (slot_in_block >> 4)) * MI_KV_TILE_ELEMS;
  const int t = slot_in_block & 15;
  return base + (is_v ? mi::v_tile_off(t, d) : mi::k_tile_off(t, d));
}

// ---------------------------------------------------------------------------
// XXH64 (Yann Collet's xxHash, 64-bit variant, as published in the xxHash
// specification) over [8-byte little-endian prefix] ++ data, seed 0.
// ---------------------------------------------------------------------------
namespace {
constexpr uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t round1(uint64_t acc, uint64_t in) { return rotl(acc + in * P2, 31) * P1; }
inline uint64_t merge(uint64_t h, uint64_t v) { return (h ^ round1(0, v)) * P1 + P4; }

// byte source that presents prefix ++ data as one stream
struct Stream {
  uint8_t pre[8];
  size_t npre;
  const uint8_t* data;
  size_t ndata, pos;
  size_t size() const { return npre + ndata; }
  uint8_t at(size_t i) const { return i < npre ? pre[i] : data[i - npre]; }
  uint64_t u64() {
    uint64_t v = 0;
    if (pos >= npre) {
      memcpy(&v, data + (pos - npre), 8);
    } else {
      for (int i = 0; i < 8; ++i) v |= (uint64_t)at(pos + i) << (8 * i);
    }
    pos += 8;
    return v;
  }
  uint32_t u32() {
    uint32_t v = 0;
    for (int i = 0; i < 4; ++i) v |= (uint32_t)at(pos + i) << (8 * i);
    pos += 4;
    return v;
  }
  uint8_t u8() { return at(pos++); }
};
}  // namespace

extern "C" uint64_t mi_xxh64_chain(const void* data, size_t len, int has_prefix, uint64_t prefix) {
  Stream s;
  s.npre = has_prefix ? 8 : 0;
  for (int i = 0; i < 8; ++i) s.pre[i] = (uint8_t)(prefix >> (8 * i));
  s.data = static_cast<const uint8_t*>(data);
  s.ndata = data ? len : 0;
  s.pos = 0;
  const size_t total = s.size();
  uint64_t h;
  if (total >= 32) {
    uint64_t v1 = P1 + P2, v2 = P2, v3 = 0, v4 = 0 - P1;
    while (s.pos + 32 <= total) {
      v1 = round1(v1, s.u64());
      v2 = round1(v2, s.u64());
      v3 = round1(v3, s.u64());
      v4 = round1(v4, s.u64());
    }
    h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
    h = merge(h, v1);
    h = merge(h, v2);
    h = merge(h, v3);
    h = merge(h, v4);
  } else {
    h = P5;
  }
  h += (uint64_t)total;
  while (s.pos + 8 <= total) {
    h ^= round1(0, s.u64());
    h = rotl(h, 27) * P1 + P4;
  }
  if (s.pos + 4 <= total) {
    h ^= (uint64_t)s.u32() * P1;
    h = rotl(h, 23) * P2 + P3;
  }
  while (s.pos < total) {
    h ^= (uint64_t)s.u8() * P5;
    h = rotl(h, 11) * P1;
  }
  h ^= h >> 33;
  h *= P2;
  h ^= h >> 29;
  h *= P3;
  h ^= h >> 32;
  return h;
}

// The chained hashes of n_blocks consecutive full blocks in one call (BlockManager.allocate hashes
// every full block of a prompt: 64 calls per 1024-token sequence otherwise).
extern "C" int mi_xxh64_chain_blocks(const int64_t* tokens, int n_blocks, int block_size, int has_prefix,
                                     uint64_t prefix, uint64_t* out) {
  if (n_blocks < 0 || block_size <= 0 || (n_blocks > 0 && (!tokens || !out))) return MI_EINVAL;
  for (int i = 0; i < n_blocks; ++i) {
    prefix = mi_xxh64_chain(tokens + (size_t)i * block_size, (size_t)block_size * sizeof(int64_t), has_prefix, prefix);
    has_prefix = 1;
    out[i] = prefix;
  }
  return MI_OK;
}
