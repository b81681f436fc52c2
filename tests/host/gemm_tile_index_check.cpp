// Host replay of the index arithmetic of csrc/gemm_tile.hip (gemm_tile_index.hpp): built and run by
// tests/test_gemm_tile_index.py with g++ - no GPU involved.
//   1. every 16-byte cell of a half-tile is written by exactly one (wave, instruction, lane) of the LDS-DMA
//   2. every fragment read finds the (tile row, K chunk) the MFMA operand layout wants
//   3. no ds_read_b128 lane group touches a 16-byte bank slot twice
//   4. a whole 256 x 256 tile computed the way the kernel does it (DMA image -> fragments -> 32x32x16 MFMA
//      semantics -> accumulator layout -> output columns, plain and SwiGLU row pairing) equals x @ w^T
//   5. the same for the 128 x 128 kernel (gemm_mid_kernel)
//   6. the same for the four-wave 256 x 256 kernel (gemm_w4_kernel)
//   7. the 16 x 16 x 32 form of that kernel (gemm_w16_kernel)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_tile_index.hpp"

using namespace mi::gt;

struct Cell { int row = -1, chunk = -1, writers = 0; };

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (fails++ < 10) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

int main() {
  // ---- 1: the DMA image of each half-tile ----
  std::vector<std::vector<Cell>> img(4, std::vector<Cell>(HALF_BYTES / 16));
  for (int h = 0; h < 4; ++h)
    for (int wave = 0; wave < 8; ++wave)
      for (int i = 0; i < 2; ++i)
        for (int lane = 0; lane < 64; ++lane) {
          const int byte = dma_block(wave, i) * 1024 + lane * 16;  // lane-linear destination
          const int lr = dma_local_row(wave, i, lane), c = dma_chunk(wave, i, lane);
          CHECK(c >= 0 && c < 8, "chunk %d", c);
          CHECK(byte == half_off(lr, c), "h %d wave %d i %d lane %d: lands at %d, layout says %d", h, wave, i, lane, byte, half_off(lr, c));
          Cell& cell = img[h][byte / 16];
          cell.row = tile_row(h, lr);
          cell.chunk = c;
          cell.writers++;
        }
  for (int h = 0; h < 4; ++h) {
    std::vector<int> seen(256 * 8, 0);
    for (const Cell& c : img[h]) {
      CHECK(c.writers == 1, "half %d: a cell has %d writers", h, c.writers);
      seen[c.row * 8 + c.chunk]++;
    }
    // a weight half holds rows {fh*128 + half*64 + j}, a token half rows {tq*64 + half*32 + j}: 128 rows x 8 chunks each once
    int rows = 0;
    for (int r = 0; r < 256; ++r) {
      int n = 0;
      for (int c = 0; c < 8; ++c) n += seen[r * 8 + c];
      CHECK(n == 0 || n == 8, "half %d row %d: %d chunks", h, r, n);
      rows += n == 8;
    }
    CHECK(rows == 128, "half %d holds %d rows", h, rows);
  }

  // ---- 2 + 3: fragment reads ----
  static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
  auto check_read = [&](int h, const int (&addr)[64], const int (&want_row)[64], int kk, const char* what) {
    for (int lane = 0; lane < 64; ++lane) {
      const Cell& c = img[h][addr[lane] / 16];
      CHECK(addr[lane] % 16 == 0 && c.row == want_row[lane] && c.chunk == frag_chunk(kk, lane >> 5),
            "%s lane %d: found row %d chunk %d, want row %d chunk %d", what, lane, c.row, c.chunk, want_row[lane], frag_chunk(kk, lane >> 5));
    }
    for (const auto& g : groups) {
      int used[16] = {0};
      for (int l : g) used[(addr[l] / 16) % 16]++;
      for (int s = 0; s < 16; ++s) CHECK(used[s] <= 1, "%s: bank slot %d used %d times in one lane group", what, s, used[s]);
    }
  };
  for (int wave = 0; wave < 8; ++wave) {
    const int fh = wave >> 2, tq = wave & 3;
    for (int kk = 0; kk < 4; ++kk) {
      for (int ah = 0; ah < 2; ++ah)
        for (int a = 0; a < 2; ++a) {
          int addr[64], want[64];
          for (int lane = 0; lane < 64; ++lane) {
            const int hi = lane >> 5, l31 = lane & 31;
            // the kernel's form: fh * 8192 + a * 4096 + rowoff + ((chunk ^ swizzle(l31)) << 4)
            addr[lane] = fh * 8192 + a * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128 + ((frag_chunk(kk, hi) ^ swizzle(l31)) << 4);
            CHECK(addr[lane] == half_off(a_local_row(fh, a, l31), frag_chunk(kk, hi)), "A address form");
            want[lane] = acc_feature(fh, ah, a, 0, 0) + l31;  // MFMA A operand: row l31 of the fragment
          }
          check_read(a_half(ah), addr, want, kk, "A fragment");
        }
      for (int bh = 0; bh < 2; ++bh) {
        int addr[64], want[64];
        for (int lane = 0; lane < 64; ++lane) {
          const int hi = lane >> 5, l31 = lane & 31;
          addr[lane] = tq * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128 + ((frag_chunk(kk, hi) ^ swizzle(l31)) << 4);
          CHECK(addr[lane] == half_off(b_local_row(tq, l31), frag_chunk(kk, hi)), "B address form");
          want[lane] = acc_token(tq, bh, l31);
        }
        check_read(b_half(bh), addr, want, kk, "B fragment");
      }
    }
  }

  // ---- 4: one tile end to end, K = 128 (two K steps), plain and SwiGLU ----
  for (int silu = 0; silu < 2; ++silu) {
    const int K = 128, N = silu ? 512 : 256, M = 256, n0 = 0, m0 = 0;
    std::vector<float> x(M * K), w(N * K);
    unsigned s = 12345u + silu;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 24) - 128) / 64.0f; };
    for (auto& v : x) v = rnd();
    for (auto& v : w) v = rnd();
    const int n_out = silu ? N / 2 : N;
    std::vector<double> got(M * n_out, 0.0), gate(M * n_out, 0.0), up(M * n_out, 0.0);
    for (int wave = 0; wave < 8; ++wave) {
      const int fh = wave >> 2, tq = wave & 3;
      for (int ah = 0; ah < 2; ++ah)
        for (int a = 0; a < 2; ++a)
          for (int bh = 0; bh < 2; ++bh) {
            // accumulator acc[ah * 2 + a][bh] of every lane
            for (int lane = 0; lane < 64; ++lane)
              for (int r = 0; r < 16; ++r) {
                // MFMA 32x32x16 semantics: D[row][col], row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), col = lane & 31;
                // A[row][k] comes from the lane with l31 = row, B[k][col] from the lane with l31 = col
                const int frow = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
                double sum = 0.0;
                for (int kt = 0; kt < K / BK; ++kt)
                  for (int kk = 0; kk < 4; ++kk)
                    for (int hi = 0; hi < 2; ++hi) {
                      const Cell& ca = img[a_half(ah)][half_off(a_local_row(fh, a, frow), frag_chunk(kk, hi)) / 16];
                      const Cell& cb = img[b_half(bh)][half_off(b_local_row(tq, col), frag_chunk(kk, hi)) / 16];
                      // source rows as the kernel computes them for the DMA
                      const int tr = ca.row;
                      const int wrow = silu ? ((tr >> 6) & 1) * (N >> 1) + (n0 >> 1) + (tr >> 7) * 64 + (tr & 63) : n0 + tr;
                      const int xrow = m0 + cb.row;
                      for (int e = 0; e < 8; ++e) {
                        const int k = kt * BK + ca.chunk * 8 + e;
                        CHECK(ca.chunk == cb.chunk, "operand chunks differ");
                        sum += (double)w[wrow * K + k] * (double)x[xrow * K + k];
                      }
                    }
                const int tok = m0 + acc_token(tq, bh, lane & 31);
                if (silu) {
                  const int j = a, rq = r >> 2, e = r & 3, hi = lane >> 5;
                  const int colo = (n0 >> 1) + fh * 64 + j * 32 + 8 * rq + 4 * hi + e;
                  (ah ? up : gate)[tok * n_out + colo] = sum;
                } else {
                  const int j = ah * 2 + a, rq = r >> 2, e = r & 3, hi = lane >> 5;
                  const int colo = n0 + acc_feature(fh, j >> 1, j & 1, 4 * rq, hi) + e;
                  got[tok * n_out + colo] = sum;
                }
              }
          }
    }
    for (int t = 0; t < M; ++t)
      for (int c = 0; c < (silu ? 128 : n_out); ++c) {  // one tile: 128 gate + 128 up rows -> 128 output columns
        if (silu) {
          double g = 0, u = 0;
          for (int k = 0; k < K; ++k) {
            g += (double)x[t * K + k] * w[c * K + k];
            u += (double)x[t * K + k] * w[(N / 2 + c) * K + k];
          }
          CHECK(gate[t * n_out + c] == g && up[t * n_out + c] == u, "silu pairing at (%d, %d)", t, c);
        } else {
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)x[t * K + k] * w[c * K + k];
          CHECK(got[t * n_out + c] == ref, "output (%d, %d): %f vs %f", t, c, got[t * n_out + c], ref);
        }
      }
  }
  // ---- 5: the 128 x 128 kernel: DMA image of its two half-tiles, fragment reads, one tile end to end ----
  {
    std::vector<std::vector<Cell>> mimg(2, std::vector<Cell>(HALF_BYTES / 16));  // [A | B]
    for (int h = 0; h < 2; ++h)
      for (int wave = 0; wave < 4; ++wave)
        for (int i = 0; i < 4; ++i)
          for (int lane = 0; lane < 64; ++lane) {
            const int byte = mid_dma_block(wave, i) * 1024 + lane * 16;
            const int lr = mid_dma_local_row(wave, i, lane), c = mid_dma_chunk(wave, i, lane);
            CHECK(byte == half_off(lr, c), "mid DMA wave %d i %d lane %d lands at %d, layout says %d", wave, i, lane, byte, half_off(lr, c));
            Cell& cell = mimg[h][byte / 16];
            cell.row = lr;
            cell.chunk = c;
            cell.writers++;
          }
    for (int h = 0; h < 2; ++h)
      for (const Cell& c : mimg[h]) CHECK(c.writers == 1 && c.row >= 0 && c.row < 128, "mid half %d: cell writers %d row %d", h, c.writers, c.row);
    for (int wave = 0; wave < 4; ++wave) {
      const int fw = wave >> 1, tw = wave & 1;
      for (int kk = 0; kk < 4; ++kk)
        for (int op = 0; op < 2; ++op)      // 0: A fragments, 1: B fragments
          for (int f = 0; f < 2; ++f) {
            int addr[64];
            for (int lane = 0; lane < 64; ++lane) {
              const int hi = lane >> 5, l31 = lane & 31;
              // the kernel's form: (fw | tw) * 8192 + f * 4096 + kx[kk]
              addr[lane] = (op ? tw : fw) * 8192 + f * 4096 + (l31 >> 3) * 1024 + (l31 & 7) * 128 + ((frag_chunk(kk, hi) ^ swizzle(l31)) << 4);
              const int lr = op ? mid_b_local_row(tw, f, l31) : mid_a_local_row(fw, f, l31);
              CHECK(addr[lane] == half_off(lr, frag_chunk(kk, hi)), "mid fragment address form");
              const Cell& c = mimg[op][addr[lane] / 16];
              CHECK(c.row == lr && c.chunk == frag_chunk(kk, hi), "mid fragment lane %d: row %d chunk %d", lane, c.row, c.chunk);
            }
            for (const auto& g : groups) {
              int used[16] = {0};
              for (int l : g) used[(addr[l] / 16) % 16]++;
              for (int sl = 0; sl < 16; ++sl) CHECK(used[sl] <= 1, "mid: bank slot %d used %d times in one lane group", sl, used[sl]);
            }
          }
    }
    for (int silu = 0; silu < 2; ++silu) {
      const int K = 64, N = silu ? 512 : 256, M = 128, n0 = 128, m0 = 0;  // the second feature tile
      std::vector<float> x(M * K), w(N * K);
      unsigned sd = 777u + silu;
      auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)((int)(sd >> 24) - 128) / 64.0f; };
      for (auto& v : x) v = rnd();
      for (auto& v : w) v = rnd();
      for (int wave = 0; wave < 4; ++wave) {
        const int fw = wave >> 1, tw = wave & 1;
        for (int b = 0; b < 2; ++b)
          for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
              const int hi = lane >> 5, l31 = lane & 31;
              double acc[2];
              for (int f = 0; f < 2; ++f) {
                const int wrow = mid_weight_row(mid_acc_feature(fw, f, r, hi), n0, N, silu != 0);
                const int xrow = m0 + mid_acc_token(tw, b, l31);
                double sum = 0;
                for (int k = 0; k < K; ++k) sum += (double)w[wrow * K + k] * x[xrow * K + k];
                acc[f] = sum;
              }
              // the epilogue's output columns: plain col0 = n0 + fw * 64 + f * 32, SwiGLU col0 = n0 / 2 + fw * 32 (gate: f = 0)
              const int within = (r & 3) + 8 * (r >> 2) + 4 * hi;
              const int tok = m0 + mid_acc_token(tw, b, l31);
              if (silu) {
                const int col = (n0 >> 1) + fw * 32 + within;
                double g = 0, u = 0;
                for (int k = 0; k < K; ++k) {
                  g += (double)x[tok * K + k] * w[col * K + k];
                  u += (double)x[tok * K + k] * w[(N / 2 + col) * K + k];
                }
                CHECK(acc[0] == g && acc[1] == u, "mid silu pairing at (%d, %d)", tok, col);
              } else {
                for (int f = 0; f < 2; ++f) {
                  const int col = n0 + fw * 64 + f * 32 + within;
                  double ref = 0;
                  for (int k = 0; k < K; ++k) ref += (double)x[tok * K + k] * w[col * K + k];
                  CHECK(acc[f] == ref, "mid output (%d, %d)", tok, col);
                }
              }
            }
      }
    }
  }
  // ---- 6: the four-wave 256 x 256 kernel (gemm_w4_kernel): DMA image of a K step, fragment reads, one tile end to end ----
  {
    std::vector<Cell> wimg(W4_STEP_BYTES / 16);  // the whole image, padding included
    std::vector<int> region_of(W4_STEP_BYTES / 16, -1);
    for (int wave = 0; wave < 4; ++wave)
      for (int q = 0; q < W4_PIECES; ++q)
        for (int lane = 0; lane < 64; ++lane) {
          const int region = w4_wave_is_weight(wave) ? 0 : 1;
          const int byte = w4_piece_off(wave, q) + lane * 16;  // lane-linear inside the piece
          const int r = w4_dma_row(wave, q, lane), c = w4_dma_chunk(lane);
          CHECK(c == (lane & 7), "w4: the lanes of a row slice fetch its chunks in order");
          CHECK(byte == w4_row_off(region, r, c), "w4 DMA wave %d piece %d lane %d lands at %d, layout says %d", wave, q, lane, byte, w4_row_off(region, r, c));
          CHECK(w4_piece_rows(w4_piece_of_row(r & 127)) + w4_slice_rows(w4_slice_of_row(r & 127)) == (r & 127), "w4 row split");
          Cell& cell = wimg[byte / 16];
          cell.row = r;
          cell.chunk = c;
          cell.writers++;
          region_of[byte / 16] = region;
        }
    for (int h = 0; h < 2; ++h) {
      std::vector<int> seen(256 * 8, 0);
      for (size_t k = 0; k < wimg.size(); ++k) {
        const Cell& c = wimg[k];
        const bool pad = (k * 16) % W4_PIECE_BYTES >= 1024;
        CHECK(c.writers == (pad ? 0 : 1), "w4 image cell %zu: %d writers", k, c.writers);
        if (!pad && region_of[k] == h) seen[c.row * 8 + c.chunk]++;
      }
      for (int v : seen) CHECK(v == 1, "w4 region %d: a (row, chunk) is held %d times", h, v);
    }
    for (int wave = 0; wave < 4; ++wave) {
      const int fw = wave >> 1, tw = wave & 1;
      for (int kk = 0; kk < 4; ++kk)
        for (int op = 0; op < 2; ++op)
          for (int f = 0; f < 4; ++f) {
            int addr[64];
            for (int lane = 0; lane < 64; ++lane) {
              const int hi = lane >> 5, l31 = lane & 31;
              // the kernel's form: region + half * 16 pieces + per-lane part + immediates
              addr[lane] = op * 32 * W4_PIECE_BYTES + (op ? tw : fw) * 16 * W4_PIECE_BYTES + w4_frag_lane(l31, hi) + w4_frag_imm(f, kk);
              const int r = op ? w4_b_row(tw, f, l31) : w4_a_row(fw, f, l31);
              CHECK(addr[lane] == w4_row_off(op, r, frag_chunk(kk, hi)), "w4 fragment address form: %d vs %d", addr[lane], w4_row_off(op, r, frag_chunk(kk, hi)));
              const Cell& c = wimg[addr[lane] / 16];
              CHECK(region_of[addr[lane] / 16] == op && c.row == r && c.chunk == frag_chunk(kk, hi), "w4 fragment lane %d: row %d chunk %d", lane, c.row, c.chunk);
            }
            for (const auto& g : groups) {
              int used[16] = {0};
              for (int l : g) used[(addr[l] / 16) % 16]++;
              for (int sl = 0; sl < 16; ++sl) CHECK(used[sl] <= 1, "w4: bank slot %d used %d times in one lane group", sl, used[sl]);
            }
          }
    }
    for (int silu = 0; silu < 2; ++silu) {
      const int K = 64, N = silu ? 1024 : 512, M = 256, n0 = 256, m0 = 0;  // the second feature tile
      std::vector<float> x(M * K), w(N * K);
      unsigned sd = 4242u + silu;
      auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return (float)((int)(sd >> 24) - 128) / 64.0f; };
      for (auto& v : x) v = rnd();
      for (auto& v : w) v = rnd();
      for (int wave = 0; wave < 4; ++wave) {
        const int fw = wave >> 1, tw = wave & 1;
        for (int j = 0; j < 4; ++j)
          for (int lane = 0; lane < 64; lane += 7)
            for (int r = 0; r < 16; ++r) {
              const int hi = lane >> 5, l31 = lane & 31;
              double acc[4];
              for (int i = 0; i < 4; ++i) {
                const int wrow = w4_weight_row(w4_acc_feature(fw, i, r, hi), n0, N, silu != 0);
                const int xrow = m0 + w4_acc_token(tw, j, l31);
                double sum = 0;
                for (int k = 0; k < K; ++k) sum += (double)w[wrow * K + k] * x[xrow * K + k];
                acc[i] = sum;
              }
              // the epilogue's output columns: plain col0 = n0 + fw * 128 + i * 32; SwiGLU col0 = n0 / 2 + fw * 64 + i * 32
              // (gate = acc[i], up = acc[i + 2], i = 0, 1)
              const int within = (r & 3) + 8 * (r >> 2) + 4 * hi;
              const int tok = m0 + w4_acc_token(tw, j, l31);
              for (int i = 0; i < (silu ? 2 : 4); ++i) {
                if (silu) {
                  const int col = (n0 >> 1) + fw * 64 + i * 32 + within;
                  double g = 0, u = 0;
                  for (int k = 0; k < K; ++k) {
                    g += (double)x[tok * K + k] * w[col * K + k];
                    u += (double)x[tok * K + k] * w[(N / 2 + col) * K + k];
                  }
                  CHECK(acc[i] == g && acc[i + 2] == u, "w4 silu pairing at (%d, %d)", tok, col);
                } else {
                  const int col = n0 + fw * 128 + i * 32 + within;
                  double ref = 0;
                  for (int k = 0; k < K; ++k) ref += (double)x[tok * K + k] * w[col * K + k];
                  CHECK(acc[i] == ref, "w4 output (%d, %d)", tok, col);
                }
              }
            }
      }
    }
  }
  // ---- 7: the 16 x 16 x 32 form of the four-wave kernel (gemm_w16_kernel): image, fragment reads, bank slots, pairing ----
  {
    std::vector<Cell> wimg(W16_STEP_BYTES / 16);
    std::vector<int> region_of(W16_STEP_BYTES / 16, -1);
    for (int wave = 0; wave < 4; ++wave)
      for (int q = 0; q < W4_PIECES; ++q)
        for (int lane = 0; lane < 64; ++lane) {
          const int region = w4_wave_is_weight(wave) ? 0 : 1;
          const int byte = w16_piece_off(wave, q) + lane * 16;
          const int r = w16_dma_row(wave, q, lane), c = lane & 7;
          CHECK(byte == w16_row_off(region, r, c), "w16 DMA wave %d piece %d lane %d lands at %d, layout says %d", wave, q, lane, byte, w16_row_off(region, r, c));
          Cell& cell = wimg[byte / 16];
          cell.row = r;
          cell.chunk = c;
          cell.writers++;
          region_of[byte / 16] = region;
        }
    for (int h = 0; h < 2; ++h) {
      std::vector<int> seen(256 * 8, 0);
      for (size_t k = 0; k < wimg.size(); ++k) {
        const bool pad = (k * 16) % W16_PIECE_BYTES >= 1024;
        CHECK(wimg[k].writers == (pad ? 0 : 1), "w16 image cell %zu: %d writers", k, wimg[k].writers);
        if (!pad && region_of[k] == h) seen[wimg[k].row * 8 + wimg[k].chunk]++;
      }
      for (int v : seen) CHECK(v == 1, "w16 region %d: a (row, chunk) is held %d times", h, v);
    }
    for (int wave = 0; wave < 4; ++wave) {
      const int fw = wave >> 1, tw = wave & 1;
      for (int kh = 0; kh < 2; ++kh)
        for (int op = 0; op < 2; ++op)
          for (int f = 0; f < 8; ++f) {
            int addr[64];
            for (int lane = 0; lane < 64; ++lane) {
              const int r16 = lane & 15, q = lane >> 4;
              addr[lane] = (op * 32 + (op ? tw : fw) * 16) * W16_PIECE_BYTES + w16_frag_lane(r16, q) + w16_frag_imm(f, kh);
              const int r = w16_frag_row(op ? tw : fw, f, r16), c = w16_frag_chunk(kh, q);
              CHECK(addr[lane] == w16_row_off(op, r, c), "w16 fragment address form: %d vs %d", addr[lane], w16_row_off(op, r, c));
              const Cell& cell = wimg[addr[lane] / 16];
              CHECK(region_of[addr[lane] / 16] == op && cell.row == r && cell.chunk == c, "w16 fragment lane %d: row %d chunk %d", lane, cell.row, cell.chunk);
            }
            for (const auto& g : groups) {
              int used[16] = {0};
              for (int l : g) used[(addr[l] / 16) % 16]++;
              for (int sl = 0; sl < 16; ++sl) CHECK(used[sl] <= 1, "w16: bank slot %d used %d times in one lane group", sl, used[sl]);
            }
          }
    }
    // SwiGLU pairing: accumulator fragments f (gate) and f + 4 (up) of a wave hold the same output column
    for (int fw = 0; fw < 2; ++fw)
      for (int f = 0; f < 4; ++f)
        for (int q = 0; q < 4; ++q)
          for (int e = 0; e < 4; ++e) {
            const int N = 1024, n0 = 512;
            const int g = w16_weight_row(w16_acc_feature(fw, f, e, q), n0, N, true), u = w16_weight_row(w16_acc_feature(fw, f + 4, e, q), n0, N, true);
            CHECK(u == g + N / 2 && g == (n0 >> 1) + fw * 64 + f * 16 + 4 * q + e, "w16 silu pairing: gate row %d up row %d", g, u);
          }
  }
  if (fails) { std::printf("%d check(s) failed\n", fails); return 1; }
  std::printf("gemm_tile index replay: ok\n");
  return 0;
}
