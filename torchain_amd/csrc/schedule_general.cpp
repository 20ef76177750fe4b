// Schedules of the general on-chip kernel: rows of <= 32 arcs sorted by length, dealt 64 at a time to the
// lanes of a wave, in-band ROW cells (chain_internal.h).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {

// Regroups rows of equal length so that consecutive blocks of 32 rows (one half-wave of a slot) have
// distinct "dominant" pdf banks -- in chain graphs most arcs of a row carry one pdf, so this alone
// makes the exp(y) gathers and the gamma atomics of a half-wave conflict-free -- and, among the
// candidates of a bank bucket, a balanced spread of state banks, so that arrange_half can find
// conflict-free steps.  Works on [begin, end) of the row order, all of one length.
template <class Row>
static void group_rows_by_bank(std::vector<Row> &rows, size_t begin, size_t end, const std::vector<int64_t> &order,
                               const int32_t *other, const int32_t *pdf) {
  const size_t n = end - begin;
  if (n < 64) return;
  std::vector<std::vector<size_t>> bucket(32);
  std::vector<std::array<uint8_t, 32>> st_hist(n);
  for (size_t i = 0; i < n; ++i) {
    const Row &r = rows[begin + i];
    int cnt[32] = {0};
    st_hist[i].fill(0);
    for (int k = 0; k < r.len; ++k) {
      const int64_t a = order[r.begin + k];
      cnt[pdf[a] & 31]++;
      st_hist[i][other[a] & 31]++;
    }
    int best = 0;
    for (int b = 1; b < 32; ++b)
      if (cnt[b] > cnt[best]) best = b;
    bucket[best].push_back(i);
  }
  std::vector<Row> out;
  out.reserve(n);
  std::vector<char> taken(n, 0);
  size_t left = n;
  while (left > 0) {
    int hist[32] = {0};
    int got = 0;
    // one row per non-empty bucket, fullest buckets first so that they drain evenly
    std::vector<int> border(32);
    for (int b = 0; b < 32; ++b) border[b] = b;
    std::stable_sort(border.begin(), border.end(), [&](int x, int y) { return bucket[x].size() > bucket[y].size(); });
    for (int round = 0; round < 4 && got < 32; ++round)
      for (int b : border) {
        if (got >= 32) break;
        auto &bk = bucket[b];
        if (bk.empty()) continue;
        if (round == 0 || bk.size() > left / 32) {  // later rounds only take from over-full buckets
          // among the last few candidates pick the one that adds least to the crowded state banks
          size_t best_j = bk.size() - 1;
          int best_cost = 1 << 30;
          for (size_t j = bk.size(); j-- > 0 && bk.size() - j <= 8;) {
            int c = 0;
            for (int q = 0; q < 32; ++q) c += st_hist[bk[j]][q] * hist[q];
            if (c < best_cost) {
              best_cost = c;
              best_j = j;
            }
          }
          const size_t i = bk[best_j];
          bk.erase(bk.begin() + best_j);
          for (int q = 0; q < 32; ++q) hist[q] += st_hist[i][q];
          out.push_back(rows[begin + i]);
          taken[i] = 1;
          ++got;
          --left;
        }
      }
    if (got == 0) break;
  }
  for (size_t i = 0; i < n; ++i)
    if (!taken[i]) out.push_back(rows[begin + i]);
  for (size_t i = 0; i < n; ++i) rows[begin + i] = out[i];
}

// Builds the row/slot schedule for one direction.  key[a] is the state whose sum arc a belongs to
// (destination for the forward pass, source for the backward pass), other[a] the state it gathers.
static void build_one(int H, int Hs, int num_pdfs, int64_t A, const int32_t *key, const int32_t *other,
                      const int32_t *pdf, const float *prob, int unroll, bool even_rows, ScheduleHost *out) {
  out->conflict_cost = out->conflict_free_cost = 0;
  struct Row {
    int32_t state, len, slot;
    int64_t begin;
  };
  // stable counting sort of arcs by key keeps the FST's arc order inside a row
  std::vector<int64_t> first(H + 1, 0);
  for (int64_t a = 0; a < A; ++a) first[key[a] + 1]++;
  for (int h = 0; h < H; ++h) first[h + 1] += first[h];
  std::vector<int64_t> order(A), fill(first.begin(), first.end() - 1);
  for (int64_t a = 0; a < A; ++a) order[fill[key[a]]++] = a;

  std::vector<Row> rows;
  std::vector<std::vector<int2>> fix_of_thread(kThreads);
  int extra = 0;
  for (int h = 0; h < H; ++h) {
    int64_t b = first[h], e = first[h + 1];
    bool first_chunk = true;
    while (b < e) {
      int len = (int)std::min<int64_t>(kMaxRowLen, e - b);
      int slot = h;
      if (!first_chunk) {
        slot = Hs + 4 + extra++;
        fix_of_thread[(h >> 2) % kThreads].push_back(make_int2(h, slot));  // owner of state h (float4 ownership)
      }
      rows.push_back({h, len, slot, b});
      first_chunk = false;
      b += len;
    }
  }
  out->extra_slots = extra;
  out->fix.clear();
  out->fix_begin.assign(kThreads + 1, 0);
  for (int t = 0; t < kThreads; ++t) {
    out->fix_begin[t] = (int)out->fix.size();
    for (auto &f : fix_of_thread[t]) out->fix.push_back(f);
  }
  out->fix_begin[kThreads] = (int)out->fix.size();
  if (out->fix.empty()) out->fix.push_back(make_int2(0, 0));
  std::stable_sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.len > y.len; });
  for (size_t b = 0; b < rows.size();) {
    size_t e = b;
    while (e < rows.size() && rows[e].len == rows[b].len) ++e;
    group_rows_by_bank(rows, b, e, order, other, pdf);
    b = e;
  }
  const int nrows = (int)rows.size();
  const int nslots = (nrows + 63) / 64;

  // longest-processing-time assignment of slots to waves (a slot costs its steps + the ROW cell)
  std::vector<std::vector<int>> per_wave(kWaves);
  std::vector<int64_t> load(kWaves, 0);
  for (int sidx = 0; sidx < nslots; ++sidx) {
    int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    per_wave[w].push_back(sidx);
    load[w] += (rows[(size_t)sidx * 64].len + 1 + (even_rows ? 1 : 0)) & (even_rows ? ~1 : ~0);
  }
  auto bits = [](uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  out->cells.clear();
  out->wave_range.assign(kWaves, make_int2(0, 0));
  int64_t arc_cells = 0;
  for (int w = 0; w < kWaves; ++w) {
    const size_t first = out->cells.size() / 64;
    for (int sidx : per_wave[w]) {
      // even_rows (tied kernel, which consumes cells in pairs): a slot occupies an even number of cells,
      // so every ROW cell sits at an even stream position; the odd slot gets one more padding step
      int steps = rows[(size_t)sidx * 64].len;
      if (even_rows && ((steps + 1) & 1)) ++steps;
      const size_t off = out->cells.size();
      out->cells.resize(off + (size_t)(steps + 1) * 64, ArcRec{0.f, 0u});
      arc_cells += (int64_t)steps * 64;
      for (int half = 0; half < 2; ++half) {
        std::vector<std::vector<int64_t>> lane_arcs(32);
        for (int l = 0; l < 32; ++l) {
          const int r = sidx * 64 + half * 32 + l;
          if (r < nrows)
            for (int k = 0; k < rows[r].len; ++k) lane_arcs[l].push_back(order[rows[r].begin + k]);
        }
        std::vector<std::vector<int>> pos;
        out->conflict_cost += arrange_half(lane_arcs, steps, other, pdf, &pos);
        out->conflict_free_cost += 2 * steps;
        for (int l = 0; l < 32; ++l) {
          const int lane = half * 32 + l;
          const int r = sidx * 64 + lane;
          // ROW cell: {slot | state << 16, flag}; its (unused) gather offsets are lane-aligned, i.e. conflict-free
          const uint32_t free_st = (uint32_t)(H >= 32 ? l : 0), free_pdf = (uint32_t)(num_pdfs >= 32 ? l : 0);
          const uint32_t dummy_idx = (free_pdf << 2) | (free_st << 18);
          out->cells[off + lane] =
              r < nrows ? ArcRec{bits((uint32_t)rows[r].slot | ((uint32_t)rows[r].state << 16)), kRowFlag | dummy_idx}
                        : ArcRec{bits((uint32_t)Hs), kRowFlag | dummy_idx};
          for (int k = 0; k < steps; ++k) {
            ArcRec &cell = out->cells[off + (size_t)(k + 1) * 64 + lane];
            if (pos[l][k] >= 0) {
              const int64_t a = lane_arcs[l][pos[l][k]];
              cell = ArcRec{prob[a], ((uint32_t)pdf[a] << 2) | ((uint32_t)other[a] << 18)};
            } else {
              cell = ArcRec{0.f, dummy_idx};  // padding: w = 0, lane-aligned offsets
            }
          }
        }
      }
    }
    // closing ROW(dummy) cell commits the last row; then pad to the unroll factor
    for (int l = 0; l < 64; ++l) out->cells.push_back(ArcRec{bits((uint32_t)Hs), kRowFlag});  // offsets 0: broadcast
    while ((out->cells.size() / 64 - first) % unroll != 0)
      for (int l = 0; l < 64; ++l) out->cells.push_back(ArcRec{0.f, 0u});
    out->wave_range[w] = make_int2((int)first, (int)(out->cells.size() / 64 - first));
  }
  for (int i = 0; i < 64 * unroll; ++i) out->cells.push_back(ArcRec{0.f, 0u});
  const size_t ncell = out->cells.size() / 64;
  // final memory layout: a lane's cells 2p and 2p+1 adjacent (16 bytes), i.e. [pair][lane][2], so the
  // kernel streams with 16-byte loads (1 KB per wave instruction)
  {
    std::vector<ArcRec> paired(out->cells.size());
    for (size_t c = 0; c < ncell; ++c)
      for (int l = 0; l < 64; ++l) paired[((c >> 1) * 64 + l) * 2 + (c & 1)] = out->cells[c * 64 + l];
    out->cells.swap(paired);
  }
  out->real_arcs = A;
  out->padded_arcs = arc_cells;
  out->rows = nrows;
}

// General (not chain-structured) graphs: one schedule per direction over all arcs.
void build_general(tc_den_graph *g) {
  const int Hs = round4(g->H);
  // forward: alpha_{t+1}(dst) sums over in-arcs, gathers alpha'_t(src)
  build_one(g->H, Hs, g->P, g->A, g->arc_dst.data(), g->arc_src.data(), g->arc_pdf.data(), g->arc_prob.data(), kStreamUnroll, false, &g->fwd);
  // backward: beta'_t(src) sums over out-arcs, gathers beta_{t+1}(dst)
  build_one(g->H, Hs, g->P, g->A, g->arc_src.data(), g->arc_dst.data(), g->arc_pdf.data(), g->arc_prob.data(), kStreamUnroll, false, &g->bwd);
}

}  // namespace tc
