// Host side of the denominator graph: construction from an FST (what the reference does by calling
// kaldi::chain::DenominatorGraph at src/my_lib_example.cpp:129-134), the wavefront schedules the HIP
// kernels stream, the OpenFst binary reader, and the per-device immutable copies.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {

thread_local int g_last_hip_error = 0;

static int round4(int x) { return (x + 3) & ~3; }

// Lays the per-frame working set of one sequence out in LDS.  Returns false if it cannot fit.
bool compute_layout(int H, int P, int T_hint, int extra_slots, bool tied, DenLayout *L) {
  L->Hs = round4(H);
  L->Ps = round4(P);
  const int jv = (L->Hs / 4 + kThreads - 1) / kThreads, pv = (L->Ps / 4 + kThreads - 1) / kThreads;
  if (H > kMaxIndex || P > kMaxIndex) return false;  // 16-bit byte offsets in ArcRec
  // the kernel instantiations (den_kernels.hip): JV in {2, 4} x PV in {1, 2, 3}
  if (jv <= kJvSmall && pv <= kPvSmall) {
    L->JV = kJvSmall;
    L->PV = kPvSmall;
  } else if (jv <= kJvSmall && pv <= kPvMid) {
    L->JV = kJvSmall;
    L->PV = kPvMid;
  } else if (jv <= kJvSmall && pv <= kPvLarge) {
    L->JV = kJvSmall;
    L->PV = kPvLarge;
  } else if (jv <= kJvLarge && pv <= kPvSmall) {
    L->JV = kJvLarge;
    L->PV = kPvSmall;
  } else if (jv <= kJvLarge && pv <= kPvMid) {
    L->JV = kJvLarge;
    L->PV = kPvMid;
  } else if (jv <= kJvLarge && pv <= kPvLarge) {
    L->JV = kJvLarge;
    L->PV = kPvLarge;
  } else {
    return false;
  }
  for (int with_alpha = 1; with_alpha >= 0; --with_alpha) {
    int off = L->PV * 4 * kThreads;  // P region, compile-time size
    L->off_a = off;
    off += L->Hs;  // A / B
    L->off_acc = off;
    L->acc_floats = round4(L->Hs + 4 + extra_slots);
    off += L->acc_floats;  // ACC / BACC (+ dummy row + private slots of split rows)
    L->off_g = off;
    off += L->Ps;
    // tied graphs, roomy layout: owner-private parking of alpha'_{t+1} and a second exp(y) buffer; the
    // tight layout (alpha_in_lds == false) re-reads alpha'_{t+1} from the history and rewrites exp(y) in
    // place behind one more barrier per backward frame -- what lets 4097..12288 pdfs stay on this path
    L->off_al = off;
    if (with_alpha) off += L->Hs + 4;
    L->off_p2 = off;
    if (tied && with_alpha) off += L->PV * 4 * kThreads;
    L->off_red = off;
    off += 4 * kWaves;
    L->off_asum = off;
    off += round4(T_hint + 1);
    L->total_floats = off;
    L->alpha_in_lds = with_alpha != 0;
    if ((int64_t)off * 4 <= kLdsLimitBytes) return true;
  }
  return false;
}

int64_t layout_lds_bytes(const DenLayout &L, int T) {
  return 4 * (int64_t)(L.off_asum + round4(T + 1));
}

// Bank-conflict-aware placement for one 32-lane half of a slot.  ds_read_b32 / ds_add_u32 service a
// wave in two 32-lane groups, one LDS cycle per distinct address per bank (bank = dword index mod 32;
// profiles/microbench: 2.3 cycles conflict-free, 7.0 for uniformly random gathers).  The sum over a
// row is order-independent and padding may sit anywhere, so for every step we pick, per lane, the arc
// of its row whose state-bank and pdf-bank are still free in that step; a few swap passes then remove
// what the greedy pass left.  pos[l][k] = index into lane l's arc list, or -1 for padding.
// Returns the cost sum_k (max state-bank multiplicity + max pdf-bank multiplicity).
static int arrange_half(const std::vector<std::vector<int64_t>> &lane_arcs, int steps, const int32_t *other,
                        const int32_t *pdf, std::vector<std::vector<int>> *pos_out) {
  const int L = (int)lane_arcs.size();
  auto bst = [&](int64_t a) { return other[a] & 31; };
  auto bpd = [&](int64_t a) { return pdf ? (pdf[a] & 31) : 0; };
  const int use_pdf = pdf ? 1 : 0;
  std::vector<std::vector<int>> pos(L, std::vector<int>(steps, -1));
  std::vector<std::vector<char>> used(L);
  std::vector<int> remaining(L);
  for (int l = 0; l < L; ++l) {
    used[l].assign(lane_arcs[l].size(), 0);
    remaining[l] = (int)lane_arcs[l].size();
  }
  std::vector<std::array<int, 32>> cs(steps), cp(steps);
  for (int k = 0; k < steps; ++k) {
    cs[k].fill(0);
    cp[k].fill(0);
    std::vector<int> lanes(L);
    for (int l = 0; l < L; ++l) lanes[l] = l;
    // rows that can no longer defer go first, then the fuller rows
    std::stable_sort(lanes.begin(), lanes.end(), [&](int x, int y) { return remaining[x] > remaining[y]; });
    for (int l : lanes) {
      if (remaining[l] == 0) continue;
      const int slack = (steps - k) - remaining[l];
      int best = -1, best_cost = 1 << 30;
      for (int i = 0; i < (int)lane_arcs[l].size(); ++i) {
        if (used[l][i]) continue;
        const int64_t a = lane_arcs[l][i];
        const int c = cs[k][bst(a)] + use_pdf * cp[k][bpd(a)];
        if (c < best_cost) {
          best_cost = c;
          best = i;
        }
      }
      if (slack > 0 && best_cost > 0) continue;  // pad here, try again at a later step
      used[l][best] = 1;
      remaining[l]--;
      pos[l][k] = best;
      cs[k][bst(lane_arcs[l][best])]++;
      cp[k][bpd(lane_arcs[l][best])]++;
    }
  }
  // improvement: swap two entries of one lane between steps when it removes conflicting pairs
  // (smooth objective: number of same-bank pairs per step, for both gathers)
  for (int pass = 0; pass < 8; ++pass) {
    bool any = false;
    for (int l = 0; l < L; ++l)
      for (int k1 = 0; k1 < steps; ++k1)
        for (int k2 = k1 + 1; k2 < steps; ++k2) {
          const int i1 = pos[l][k1], i2 = pos[l][k2];
          if (i1 == i2) continue;
          // pairs removed/added: moving arc a from step x to step y changes the pair count by
          // (count_y(b) - (count_x(b) - 1)) per attribute
          int delta = 0;
          auto delta_move = [&](int idx, int from, int to, int other_idx) {
            if (idx < 0) return;
            const int64_t a = lane_arcs[l][idx];
            int s_to = cs[to][bst(a)], p_to = cp[to][bpd(a)];
            if (other_idx >= 0) {  // the arc leaving `to` in the same swap
              const int64_t o = lane_arcs[l][other_idx];
              if (bst(o) == bst(a)) s_to--;
              if (bpd(o) == bpd(a)) p_to--;
            }
            delta += s_to - (cs[from][bst(a)] - 1);
            if (use_pdf) delta += p_to - (cp[from][bpd(a)] - 1);
          };
          delta_move(i1, k1, k2, i2);
          delta_move(i2, k2, k1, i1);
          if (delta < 0) {
            auto apply = [&](int idx, int from, int to) {
              if (idx < 0) return;
              const int64_t a = lane_arcs[l][idx];
              cs[from][bst(a)]--;
              cp[from][bpd(a)]--;
              cs[to][bst(a)]++;
              cp[to][bpd(a)]++;
            };
            apply(i1, k1, k2);
            apply(i2, k2, k1);
            std::swap(pos[l][k1], pos[l][k2]);
            any = true;
          }
        }
    if (!any) break;
  }
  int total = 0;
  for (int k = 0; k < steps; ++k) {
    int ms = 0, mp = 0;
    for (int b = 0; b < 32; ++b) {
      ms = std::max(ms, cs[k][b]);
      mp = std::max(mp, cp[k][b]);
    }
    total += std::max(ms, 1) + use_pdf * std::max(mp, 1);
    if (getenv("TC_SCHED_DEBUG")) {
      static long long n = 0, sst = 0, spd = 0;
      n++; sst += std::max(ms, 1); spd += std::max(mp, 1);
      if (n % 2000 == 0) fprintf(stderr, "[sched] steps=%lld avg max-mult state=%.3f pdf=%.3f\n", n, (double)sst / n, (double)spd / n);
    }
  }
  *pos_out = pos;
  return total;
}

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

// ---- tied graphs: "owner-computes" schedules -------------------------------------------------------
// The thread that owns a state (float4 ownership: LDS position p belongs to thread (p / 4) % 1024, its
// row index k = 4 * (p / 4096) + p % 4) also walks that state's arc list, in both directions, so a row
// sum never leaves its thread: no accumulator exchange through LDS, no barrier between the walk and the
// per-state pass, no ROW cells in the stream.  The 64 lanes of a wave run their k-th rows in lockstep
// (a (wave, k) "slot" costs the longest of its 64 rows), so states are PERMUTED: sorted by primary
// in- and out-degree and dealt 64 at a time, which makes the rows of a slot (nearly) equally long in
// both directions.  Everything per-state the kernel touches (pi, tied tables, alpha history) is stored
// in position order; positions never leave the library.  Row ends are wave-uniform and known in
// advance: one mask bit per pair of cells, eight pairs per mask word, read through the scalar cache.
// Arc lists longer than kMaxRowLen keep their first kMaxRowLen arcs at home; the rest become secondary
// rows (k >= K) of whichever waves have room, commit to private slots behind the accumulators and are
// folded in by the owner after a barrier that only such graphs pay.
struct OwnerTask {
  int32_t state;   // original state id (or -1: empty)
  int64_t begin;   // range into the direction's arc order
  int32_t len;
};

static void emit_owner_stream(int Npos, int K, const std::vector<std::vector<std::vector<OwnerTask>>> &slots,
                              const std::vector<int64_t> &order, const int32_t *opos, const float *prob,
                              ScheduleHost *out) {
  // slots[w][k] = 64 tasks (lane order); k >= K are secondary rows
  out->conflict_cost = out->conflict_free_cost = 0;
  out->cells.clear();
  out->wave_range.assign(kWaves, make_int2(0, 0));
  std::vector<std::vector<uint32_t>> wave_masks(kWaves);
  int64_t arc_cells = 0;
  int nrows = 0;
  for (int w = 0; w < kWaves; ++w) {
    const size_t first = out->cells.size() / 64;
    size_t cells_before = 0;
    std::vector<char> row_end;  // per pair of this wave: flags A | B
    for (size_t k = 0; k < slots[w].size(); ++k) {
      const auto &tasks = slots[w][k];
      int steps = 1;  // rows need not be whole pairs: a pair may straddle two rows (flag A below)
      for (const OwnerTask &t : tasks) steps = std::max(steps, t.len);
      const size_t off = out->cells.size();
      out->cells.resize(off + (size_t)steps * 64, ArcRec{0.f, 0u});
      arc_cells += (int64_t)steps * 64;
      for (int half = 0; half < 2; ++half) {
        std::vector<std::vector<int64_t>> lane_arcs(32);
        for (int l = 0; l < 32; ++l) {
          const OwnerTask &t = tasks[half * 32 + l];
          for (int i = 0; i < t.len; ++i) lane_arcs[l].push_back(order[t.begin + i]);
          if (t.len > 0) ++nrows;
        }
        std::vector<std::vector<int>> pos;
        out->conflict_cost += arrange_half(lane_arcs, steps, opos, nullptr, &pos);
        out->conflict_free_cost += steps;
        for (int l = 0; l < 32; ++l) {
          const int lane = half * 32 + l;
          for (int i = 0; i < steps; ++i) {
            ArcRec &cell = out->cells[off + (size_t)i * 64 + lane];
            if (pos[l][i] >= 0) {
              const int64_t a = lane_arcs[l][pos[l][i]];
              cell = ArcRec{prob[a], (uint32_t)opos[a] << 18};
            } else {
              cell = ArcRec{0.f, (uint32_t)(Npos >= 32 ? l : 0) << 18};  // padding: w = 0, conflict-free offset
            }
          }
        }
      }
      cells_before += steps;
      row_end.resize((cells_before + 1) / 2, 0);
      // bit 0 (B): the row ends with the pair's second cell; bit 1 (A): with its first cell
      row_end[(cells_before - 1) / 2] |= ((cells_before - 1) & 1) ? 1 : 2;
    }
    // whole chunks, and at least two of them (the forward walk keeps its first two chunks in registers)
    while ((out->cells.size() / 64 - first) % kStreamUnrollTied != 0 || out->cells.size() / 64 - first < 2 * kStreamUnrollTied)
      for (int l = 0; l < 64; ++l) out->cells.push_back(ArcRec{0.f, 0u});
    out->wave_range[w] = make_int2((int)first, (int)(out->cells.size() / 64 - first));
    if (getenv("TC_SCHED_DEBUG")) fprintf(stderr, "[sched] wave %d: %d cells, %zu rows\n", w, out->wave_range[w].y, slots[w].size());
    auto &mw = wave_masks[w];
    mw.assign((row_end.size() + 7) / 8, 0u);
    for (size_t i = 0; i < row_end.size(); ++i) {
      if (row_end[i] & 1) mw[i / 8] |= 1u << (i % 8);        // B flags: bits 0..7
      if (row_end[i] & 2) mw[i / 8] |= 1u << (8 + i % 8);    // A flags: bits 8..15
    }
  }
  // readable padding: the kernels request up to four chunks past a wave's range
  for (int i = 0; i < 64 * 32; ++i) out->cells.push_back(ArcRec{0.f, 0u});
  size_t stride = 1;
  for (auto &mw : wave_masks) stride = std::max(stride, mw.size());
  stride += 2;  // the walk prefetches one word ahead
  out->mask_stride = (int32_t)stride;
  out->masks.assign(stride * kWaves, 0u);
  for (int w = 0; w < kWaves; ++w) std::copy(wave_masks[w].begin(), wave_masks[w].end(), out->masks.begin() + w * stride);
  out->real_arcs = (int64_t)order.size();
  out->padded_arcs = arc_cells;
  out->rows = nrows;
  // 6-byte cells, [chunk of 8 cells][3 blocks][lane]{16 bytes}: {w0..w3}, {w4..w7}, {off01, off23, off45, off67}
  const size_t ncell = out->cells.size() / 64;
  out->cells6.assign(ncell / 8 * 3 * 64 * 4, 0u);
  for (size_t c = 0; c < ncell; ++c)
    for (int l = 0; l < 64; ++l) {
      const ArcRec &cell = out->cells[c * 64 + l];
      uint32_t x;
      memcpy(&x, &cell.w, 4);
      const size_t chunk = c / 8, i = c % 8;
      uint32_t *base = &out->cells6[chunk * 3 * 64 * 4];
      base[((i / 4) * 64 + l) * 4 + (i % 4)] = x;
      uint32_t &o = base[(2 * 64 + l) * 4 + i / 2];
      const uint32_t off16 = cell.idx >> 16;  // position * 4
      o |= (i & 1) ? off16 << 16 : off16;
    }
  out->cells.clear();
  out->cells.shrink_to_fit();
}

// Returns false when the graph cannot use the owner-computes kernel (too many states for the 16-bit
// offsets or the working set does not fit LDS); the caller then falls back to the general kernel.
static bool build_owner(tc_den_graph *g, const std::vector<char> &special) {
  const int H = g->work_H;  // states of the work graph (tc_den_graph::work_*)
  const int Npos = 4096 * ((H + 4095) / 4096);
  if (Npos > kMaxIndex) return false;
  const int K = Npos / kThreads;
  std::vector<int32_t> src, dst;
  std::vector<float> prob;
  for (int64_t a = 0; a < (int64_t)g->work_src.size(); ++a)
    if (!special[a]) {
      src.push_back(g->work_src[a]);
      dst.push_back(g->work_dst[a]);
      prob.push_back(g->work_prob[a]);
    }
  const int64_t A2 = (int64_t)src.size();
  auto sort_by = [&](const std::vector<int32_t> &key, std::vector<int64_t> *first, std::vector<int64_t> *order) {
    first->assign(H + 1, 0);
    for (int64_t a = 0; a < A2; ++a) (*first)[key[a] + 1]++;
    for (int h = 0; h < H; ++h) (*first)[h + 1] += (*first)[h];
    order->resize(A2);
    std::vector<int64_t> fill(first->begin(), first->end() - 1);
    for (int64_t a = 0; a < A2; ++a) (*order)[fill[key[a]]++] = a;
  };
  std::vector<int64_t> in_first, in_order, out_first, out_order;
  sort_by(dst, &in_first, &in_order);
  sort_by(src, &out_first, &out_order);
  auto deg = [](const std::vector<int64_t> &first, int h) { return (int)(first[h + 1] - first[h]); };

  // ---- the permutation: sort by primary in-length into a few super-buckets, inside by primary out-length
  std::vector<int32_t> st(H);
  std::iota(st.begin(), st.end(), 0);
  auto lin = [&](int h) { return std::min(deg(in_first, h), kMaxRowLen); };
  auto lout = [&](int h) { return std::min(deg(out_first, h), kMaxRowLen); };
  std::stable_sort(st.begin(), st.end(), [&](int x, int y) { return lin(x) > lin(y); });
  const int ngroups = Npos / 64;
  const int nbucket = std::max(1, (int)std::lround(std::sqrt((double)std::max(1, (H + 63) / 64))));
  for (int b = 0; b < nbucket; ++b) {
    const size_t lo = (size_t)H * b / nbucket, hi = (size_t)H * (b + 1) / nbucket;
    std::stable_sort(st.begin() + lo, st.begin() + hi, [&](int x, int y) { return lout(x) > lout(y); });
  }
  // Inside runs of equal (in, out) length the order is free: use it so that every 32 consecutive states --
  // one half-slot, i.e. the 32 lanes that gather exp(y) at f(g) / s(g) and add gamma there in ONE
  // instruction of the per-state passes -- have distinct pdf banks (greedy, first fit).
  if (!getenv("TC_NO_PDF_BANKS")) {
    auto key = [&](int h) { return lin(h) * 64 + lout(h); };
    int used_f[32], used_s[32];
    size_t run_end = 0;
    for (size_t i = 0; i < st.size(); ++i) {
      if (i % 32 == 0) {
        std::fill(used_f, used_f + 32, 0);
        std::fill(used_s, used_s + 32, 0);
      }
      if (i >= run_end) {
        run_end = i + 1;
        while (run_end < st.size() && key(st[run_end]) == key(st[i])) ++run_end;
      }
      size_t best = i;
      int best_cost = 1 << 30;
      for (size_t c = i; c < run_end && best_cost > 0; ++c) {
        const uint32_t fs = g->tied_fs[st[c]];
        const int cost = used_f[((fs & 0xffffu) >> 2) & 31] + used_s[(fs >> 18) & 31];
        if (cost < best_cost) {
          best_cost = cost;
          best = c;
        }
      }
      std::swap(st[i], st[best]);
      const uint32_t fs = g->tied_fs[st[i]];
      used_f[((fs & 0xffffu) >> 2) & 31]++;
      used_s[(fs >> 18) & 31]++;
    }
  }
  st.resize(Npos, -1);  // phantom states: no arcs, pi = 0
  struct Group { int idx, cin, cout; };
  std::vector<Group> groups(ngroups);
  for (int gi = 0; gi < ngroups; ++gi) {
    int mi = 1, mo = 1;
    for (int l = 0; l < 64; ++l) {
      const int h = st[(size_t)gi * 64 + l];
      if (h < 0) continue;
      mi = std::max(mi, lin(h));
      mo = std::max(mo, lout(h));
    }
    groups[gi] = Group{gi, mi, mo};
  }
  // longest-processing-time deal of the groups to the waves, K per wave, balancing both directions
  std::vector<Group> by_cost(groups);
  std::stable_sort(by_cost.begin(), by_cost.end(), [](const Group &x, const Group &y) { return x.cin + x.cout > y.cin + y.cout; });
  std::vector<std::vector<int>> wave_groups(kWaves);
  std::vector<int64_t> load_in(kWaves, 0), load_out(kWaves, 0);
  for (const Group &gr : by_cost) {
    // (shares skewed towards the older waves of each SIMD, which the CU serves first, were measured: no
    // gain -- the walk is bound by the shared stream path, not by any one wave)
    int best = -1;
    int64_t best_t = 0;
    for (int w = 0; w < kWaves; ++w) {
      if ((int)wave_groups[w].size() >= K) continue;
      const int64_t tw = std::max(load_in[w] + gr.cin, load_out[w] + gr.cout);
      if (best < 0 || tw < best_t) {
        best = w;
        best_t = tw;
      }
    }
    wave_groups[best].push_back(gr.idx);
    load_in[best] += gr.cin;
    load_out[best] += gr.cout;
  }
  g->pos.assign(H, 0);
  std::vector<int32_t> state_at(Npos, -1);
  for (int w = 0; w < kWaves; ++w)
    for (int k = 0; k < K; ++k)
      for (int l = 0; l < 64; ++l) {
        const int h = st[(size_t)wave_groups[w][k] * 64 + l];
        const int p = 4 * ((64 * w + l) + kThreads * (k >> 2)) + (k & 3);
        state_at[p] = h;
        if (h >= 0) g->pos[h] = p;
      }

  // ---- lane permutation inside every half-slot: flatten the gathers' bank histograms
  // The 32 lanes of a half-slot gather, step by step, one source state each; ds_read_b32 serves the 32
  // lanes in (max number of distinct addresses on one bank) cycles, bank = position mod 32, and a state's
  // bank is fixed by where it lives: 4 * (lane mod 8) + (k mod 4).  Whatever arrange_half does later,
  // a half-slot of S steps cannot take fewer than max(S, most loaded bank) cycles, and with positions
  // assigned by degree alone the most loaded bank is ~1.8 S.  So before the streams are emitted, states
  // swap lanes WITHIN their half-slot (their own rows stay where they are; only the banks they present
  // to the rows that gather them change) under a greedy local search on sum_b hist[b]^2 over all
  // half-slots of both directions.
  if (!getenv("TC_NO_BANK_SEARCH")) {
    const int nhalf = kWaves * K * 2;
    auto half_of = [&](int p) {
      const int tid = (p >> 2) % kThreads, k = 4 * (p / (4 * kThreads)) + (p & 3);
      return ((tid / 64) * K + k) * 2 + ((tid % 64) >= 32 ? 1 : 0);
    };
    auto bank_of = [](int p) { return p & 31; };
    // hist[dir][half][bank]; dir 0: rows of destinations gather sources, dir 1: rows of sources gather destinations
    std::vector<int32_t> hist((size_t)2 * nhalf * 32, 0);
    auto H_ = [&](int dir, int half, int bank) -> int32_t & { return hist[((size_t)dir * nhalf + half) * 32 + bank]; };
    for (int64_t a = 0; a < A2; ++a) {
      H_(0, half_of(g->pos[dst[a]]), bank_of(g->pos[src[a]]))++;
      H_(1, half_of(g->pos[src[a]]), bank_of(g->pos[dst[a]]))++;
    }
    // moving state u from bank b1 to bank b2 changes sum h^2 by the sum over the rows gathering u
    auto move_delta = [&](int u, int b1, int b2) {
      int64_t d = 0;
      for (int64_t i = out_first[u]; i < out_first[u + 1]; ++i) {  // arcs u -> x: row of x gathers u (forward)
        const int hf = half_of(g->pos[dst[out_order[i]]]);
        int32_t &x1 = H_(0, hf, b1), &x2 = H_(0, hf, b2);
        d += (int64_t)(2 * x2 + 1) - (2 * x1 - 1);
        --x1;
        ++x2;
      }
      for (int64_t i = in_first[u]; i < in_first[u + 1]; ++i) {  // arcs x -> u: row of x gathers u (backward)
        const int hf = half_of(g->pos[src[in_order[i]]]);
        int32_t &x1 = H_(1, hf, b1), &x2 = H_(1, hf, b2);
        d += (int64_t)(2 * x2 + 1) - (2 * x1 - 1);
        --x1;
        ++x2;
      }
      return d;
    };
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
      rng ^= rng << 13;
      rng ^= rng >> 7;
      rng ^= rng << 17;
      return rng;
    };
    const int64_t proposals = (int64_t)Npos * 200;
    int64_t accepted = 0;
    for (int64_t it = 0; it < proposals; ++it) {
      const uint64_t r = next();
      const int w = (int)(r % kWaves), k = (int)((r >> 8) % K), hb = (int)((r >> 16) & 1);
      const int la = hb * 32 + (int)((r >> 20) % 32), lb = hb * 32 + (int)((r >> 28) % 32);
      if ((la & 7) == (lb & 7)) continue;  // same bank: nothing changes
      const int pa = 4 * ((64 * w + la) + kThreads * (k >> 2)) + (k & 3), pb = 4 * ((64 * w + lb) + kThreads * (k >> 2)) + (k & 3);
      const int u = state_at[pa], v = state_at[pb];
      const int ba = bank_of(pa), bb = bank_of(pb);
      // apply both moves, keep them if the total improved
      int64_t d = 0;
      if (u >= 0) d += move_delta(u, ba, bb);
      if (v >= 0) d += move_delta(v, bb, ba);
      if (d < 0) {
        state_at[pa] = v;
        state_at[pb] = u;
        if (u >= 0) g->pos[u] = pb;
        if (v >= 0) g->pos[v] = pa;
        ++accepted;
      } else {
        if (v >= 0) move_delta(v, ba, bb);
        if (u >= 0) move_delta(u, bb, ba);
      }
    }
    if (getenv("TC_SCHED_DEBUG")) {
      fprintf(stderr, "[sched] bank search: %lld of %lld swaps accepted\n", (long long)accepted, (long long)proposals);
      for (int dir = 0; dir < 2; ++dir) {
        int64_t sum_max = 0, sum_avg = 0;
        for (int hf = 0; hf < nhalf; ++hf) {
          int mx = 0, tot = 0;
          for (int b = 0; b < 32; ++b) {
            mx = std::max(mx, (int)H_(dir, hf, b));
            tot += H_(dir, hf, b);
          }
          sum_max += mx;
          sum_avg += (tot + 31) / 32;
        }
        fprintf(stderr, "[sched] dir %d: sum over half-slots of max bank load %lld, of mean bank load %lld\n", dir, (long long)sum_max, (long long)sum_avg);
      }
    }
  }

  // ---- per direction: primary rows at home, secondary rows dealt to the least-loaded waves
  int extra_total[2] = {0, 0};
  for (int dir = 0; dir < 2; ++dir) {
    const std::vector<int64_t> &first = dir == 0 ? in_first : out_first, &order = dir == 0 ? in_order : out_order;
    const std::vector<int32_t> &other = dir == 0 ? src : dst;
    ScheduleHost *out = dir == 0 ? &g->fwd : &g->bwd;
    std::vector<int32_t> opos(A2);
    for (int64_t a = 0; a < A2; ++a) opos[a] = g->pos[other[a]];
    std::vector<std::vector<std::vector<OwnerTask>>> slots(kWaves, std::vector<std::vector<OwnerTask>>(K, std::vector<OwnerTask>(64)));
    std::vector<OwnerTask> secondary;
    std::vector<int64_t> load(kWaves, 0);
    for (int w = 0; w < kWaves; ++w)
      for (int k = 0; k < K; ++k) {
        int steps = 1;
        for (int l = 0; l < 64; ++l) {
          const int p = 4 * ((64 * w + l) + kThreads * (k >> 2)) + (k & 3);
          const int h = state_at[p];
          OwnerTask t{h, 0, 0};
          if (h >= 0) {
            const int d = deg(first, h);
            t.begin = first[h];
            t.len = std::min(d, kMaxRowLen);
            for (int done = t.len; done < d; done += kMaxRowLen)
              secondary.push_back(OwnerTask{h, first[h] + done, std::min(kMaxRowLen, d - done)});
          }
          slots[w][k][l] = t;
          steps = std::max(steps, t.len);
        }
        load[w] += steps;
      }
    std::stable_sort(secondary.begin(), secondary.end(), [](const OwnerTask &x, const OwnerTask &y) { return x.len > y.len; });
    std::vector<std::vector<int2>> fix_of_thread(kThreads);
    std::vector<int> extra_first(kWaves + 1, 0);
    std::vector<std::vector<std::vector<OwnerTask>>> sec_slots(kWaves);
    for (size_t b = 0; b < secondary.size(); b += 64) {
      const int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      std::vector<OwnerTask> tasks(64, OwnerTask{-1, 0, 0});
      for (size_t i = b; i < std::min(secondary.size(), b + 64); ++i) tasks[i - b] = secondary[i];
      load[w] += secondary[b].len;
      sec_slots[w].push_back(tasks);
    }
    // private slots: wave w's j-th secondary row, lane l -> accumulator index Npos + 4 + 64 * (extra_first[w] + j) + l
    for (int w = 0; w < kWaves; ++w) extra_first[w + 1] = extra_first[w] + (int)sec_slots[w].size();
    for (int w = 0; w < kWaves; ++w)
      for (size_t j = 0; j < sec_slots[w].size(); ++j) {
        for (int l = 0; l < 64; ++l) {
          const OwnerTask &t = sec_slots[w][j][l];
          if (t.state < 0) continue;
          const int p = g->pos[t.state];
          fix_of_thread[(p >> 2) % kThreads].push_back(make_int2(p, Npos + 4 + 64 * (extra_first[w] + (int)j) + l));
        }
        slots[w].push_back(sec_slots[w][j]);
      }
    out->extra_first.assign(extra_first.begin(), extra_first.end() - 1);
    out->extra_slots = 64 * extra_first[kWaves];
    extra_total[dir] = out->extra_slots;
    out->fix.clear();
    out->fix_begin.assign(kThreads + 1, 0);
    for (int t = 0; t < kThreads; ++t) {
      out->fix_begin[t] = (int)out->fix.size();
      for (auto &f : fix_of_thread[t]) out->fix.push_back(f);
    }
    out->fix_begin[kThreads] = (int)out->fix.size();
    out->nfix = (int32_t)out->fix.size();
    if (out->fix.empty()) out->fix.push_back(make_int2(0, 0));
    emit_owner_stream(Npos, K, slots, order, opos.data(), prob.data(), out);
  }
  if (!compute_layout(Npos, g->P, 256, std::max(extra_total[0], extra_total[1]), true, &g->layout)) return false;
  // per-state tables in position order
  std::vector<uint32_t> fs(Npos + 4, 0u);
  std::vector<float> ws(Npos + 4, 0.f);
  g->pi_pos.assign(Npos + 4, 0.f);
  for (int h = 0; h < H; ++h) {
    fs[g->pos[h]] = g->tied_fs[h];
    ws[g->pos[h]] = g->tied_w[h];
    g->pi_pos[g->pos[h]] = g->work_pi[h];
  }
  g->tied_fs.swap(fs);
  g->tied_w.swap(ws);
  return true;
}

// Is the work graph tied?  Per state g: every non-self-loop in-arc carries one pdf f(g); self-loops that
// also carry f(g) are ordinary members of that class; at most one further self-loop (pdf s(g)) is
// "special" and is applied by the thread that owns g instead of travelling in the schedules.  Fills
// special[] and the per-state tables tied_fs / tied_w (work-state order).
static bool detect_tied(tc_den_graph *g, std::vector<char> *special) {
  const int H = g->work_H;
  const int64_t A = (int64_t)g->work_src.size();
  special->assign(A, 0);
  std::vector<int32_t> fpdf(H, -1), spdf(H, -1);
  std::vector<float> wself(H, 0.f);
  for (int64_t a = 0; a < A; ++a) {
    const int s = g->work_src[a], d = g->work_dst[a], p = g->work_pdf[a];
    if (s == d) continue;
    if (fpdf[d] >= 0 && fpdf[d] != p) return false;
    fpdf[d] = p;
  }
  for (int64_t a = 0; a < A; ++a) {
    const int s = g->work_src[a], d = g->work_dst[a], p = g->work_pdf[a];
    if (s != d) continue;
    if (fpdf[d] >= 0 && p == fpdf[d]) continue;  // forward class
    if (spdf[d] < 0) {
      spdf[d] = p;
      wself[d] = g->work_prob[a];
      (*special)[a] = 1;
    } else if (fpdf[d] < 0) {
      fpdf[d] = p;
    } else {
      return false;
    }
  }
  const int Hs = round4(H);
  g->tied_fs.assign(Hs + 4, 0u);
  g->tied_w.assign(Hs + 4, 0.f);
  for (int h = 0; h < H; ++h) {
    g->tied_fs[h] = (uint32_t)(std::max(fpdf[h], 0) * 4) | ((uint32_t)(std::max(spdf[h], 0) * 4) << 16);
    g->tied_w[h] = wself[h];
  }
  return true;
}

// Tied-ification.  Real chain graphs are tied except where minimisation merged two phone instances with
// the same self-loop pdf and future but different forward pdfs (LM back-off); one such state would send
// the whole graph to the general kernel.  Splitting state g into one copy per pdf that enters it is
// exact: the copies share g's out-arcs (and its special self-loop), so their futures are identical,
// beta(copy) = beta(g), alpha(g) = sum of the copies' alphas, and pi(g) may sit on any one of them.
// Every arc h -> g is replicated from every copy of h.  Returns false (graph left untouched) when the
// split graph would be more than 1.5x the states or 2x the arcs: arbitrary labelings are not chain graphs.
static bool make_work_graph(tc_den_graph *g) {
  const int H = g->H;
  const int64_t A = g->A;
  // classes of a state: pdfs of its non-self-loop in-arcs, plus self-loop pdfs beyond the first new one
  std::vector<std::vector<int32_t>> cls(H);
  std::vector<int32_t> spdf(H, -1);
  auto has = [](const std::vector<int32_t> &v, int32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); };
  for (int64_t a = 0; a < A; ++a)
    if (g->arc_src[a] != g->arc_dst[a] && !has(cls[g->arc_dst[a]], g->arc_pdf[a])) {
      if (cls[g->arc_dst[a]].size() >= 64) return false;
      cls[g->arc_dst[a]].push_back(g->arc_pdf[a]);
    }
  std::vector<char> is_special(A, 0);
  for (int64_t a = 0; a < A; ++a) {
    const int h = g->arc_src[a], p = g->arc_pdf[a];
    if (h != g->arc_dst[a] || has(cls[h], p)) continue;
    if (spdf[h] < 0 || spdf[h] == p) {
      if (spdf[h] == p) return false;  // two self-loops with one pdf: keep it simple, general path
      spdf[h] = p;
      is_special[a] = 1;
    } else {
      cls[h].push_back(p);
    }
  }
  std::vector<int32_t> first(H + 1, 0);
  for (int h = 0; h < H; ++h) first[h + 1] = first[h] + std::max<int>(1, (int)cls[h].size());
  const int WH = first[H];
  if (WH == H) return false;  // nothing to split: the graph failed the tied test for another reason
  if (WH > H + H / 2 + 64) return false;
  std::vector<int32_t> ws, wd, wp;
  std::vector<float> ww;
  for (int64_t a = 0; a < A; ++a) {
    const int h = g->arc_src[a], d = g->arc_dst[a], p = g->arc_pdf[a];
    const int nh = first[h + 1] - first[h];
    if (is_special[a]) {
      for (int c = 0; c < nh; ++c) {
        ws.push_back(first[h] + c);
        wd.push_back(first[h] + c);
        wp.push_back(p);
        ww.push_back(g->arc_prob[a]);
      }
      continue;
    }
    const int target = first[d] + (int)(std::find(cls[d].begin(), cls[d].end(), p) - cls[d].begin());
    for (int c = 0; c < nh; ++c) {
      ws.push_back(first[h] + c);
      wd.push_back(target);
      wp.push_back(p);
      ww.push_back(g->arc_prob[a]);
    }
    if ((int64_t)ws.size() > 2 * A + 1024) return false;
  }
  g->work_H = WH;
  g->work_src.swap(ws);
  g->work_dst.swap(wd);
  g->work_pdf.swap(wp);
  g->work_prob.swap(ww);
  g->work_pi.assign(WH, 0.f);
  for (int h = 0; h < H; ++h) g->work_pi[first[h]] = g->initial_probs[h];
  g->copy_first = first;
  return true;
}

// CSR lists for the streamed path (chain_internal.h: BigArc)
static void build_big(tc_den_graph *g) {
  const int H = g->H, P = g->P;
  const int64_t A = g->A;
  auto csr = [&](int n, const std::vector<int32_t> &key, std::vector<int32_t> *begin, std::vector<BigArc> *out,
                 const std::vector<int32_t> &fa, const std::vector<int32_t> &fb) {
    begin->assign(n + 1, 0);
    for (int64_t a = 0; a < A; ++a) (*begin)[key[a] + 1]++;
    for (int i = 0; i < n; ++i) (*begin)[i + 1] += (*begin)[i];
    out->assign(std::max<int64_t>(A, 1), BigArc{0, 0, 0.f, 0.f});
    std::vector<int32_t> fill(begin->begin(), begin->end() - 1);
    for (int64_t a = 0; a < A; ++a)  // FST arc order kept
      (*out)[fill[key[a]]++] = BigArc{fa[a], fb[a], g->arc_prob[a], g->initial_probs[g->arc_src[a]]};
  };
  csr(H, g->arc_dst, &g->big_in_begin, &g->big_in, g->arc_src, g->arc_pdf);
  csr(H, g->arc_src, &g->big_out_begin, &g->big_out, g->arc_dst, g->arc_pdf);
  csr(P, g->arc_pdf, &g->big_pdf_begin, &g->big_pdf, g->arc_src, g->arc_dst);
  float sum_pi = 0.f;
  for (int h = 0; h < H; ++h) sum_pi += g->initial_probs[h];
  g->big_sum_pi = sum_pi;
}

int build_schedules(tc_den_graph *g) {
  const int Hs = round4(g->H);
  if (getenv("TC_FORCE_BIG") || g->H > kMaxIndex || g->P > kMaxIndex) {
    g->big = true;
    g->tied = false;
    g->layout_ok = false;
    build_big(g);
    return TC_OK;
  }
  // ---- the tied path: on the FST as it is, or on its tied-ified work graph
  std::vector<char> special;
  {
    g->work_H = g->H;
    g->work_src = g->arc_src;
    g->work_dst = g->arc_dst;
    g->work_pdf = g->arc_pdf;
    g->work_prob = g->arc_prob;
    g->work_pi = g->initial_probs;
    g->copy_first.resize(g->H + 1);
    std::iota(g->copy_first.begin(), g->copy_first.end(), 0);
    bool tied = !getenv("TC_FORCE_GENERAL") && detect_tied(g, &special);
    if (!tied && !getenv("TC_FORCE_GENERAL") && !getenv("TC_NO_SPLIT") && make_work_graph(g)) tied = detect_tied(g, &special);
    g->tied = tied;
  }
  if (g->tied) {
    if (build_owner(g, special)) {
      g->layout_ok = true;
      return TC_OK;
    }
    g->tied = false;  // does not fit the owner-computes layout: use the general kernel
    g->fwd = ScheduleHost();
    g->bwd = ScheduleHost();
  }
  // forward: alpha_{t+1}(dst) sums over in-arcs, gathers alpha'_t(src)
  build_one(g->H, Hs, g->P, g->A, g->arc_dst.data(), g->arc_src.data(), g->arc_pdf.data(), g->arc_prob.data(), kStreamUnroll, false, &g->fwd);
  // backward: beta'_t(src) sums over out-arcs, gathers beta_{t+1}(dst)
  build_one(g->H, Hs, g->P, g->A, g->arc_src.data(), g->arc_dst.data(), g->arc_pdf.data(), g->arc_prob.data(), kStreamUnroll, false, &g->bwd);
  g->layout_ok = compute_layout(g->H, g->P, 256, std::max(g->fwd.extra_slots, g->bwd.extra_slots), false, &g->layout);
  if (!g->layout_ok) {  // the per-frame working set does not fit LDS: stream it
    g->big = true;
    g->fwd = ScheduleHost();
    g->bwd = ScheduleHost();
    build_big(g);
  }
  return TC_OK;
}

}  // namespace tc

using namespace tc;

extern "C" {

int tc_den_graph_create(tc_den_graph **out, int32_t num_states, int64_t num_arcs, const int32_t *arc_src,
                        const int32_t *arc_dst, const int32_t *arc_ilabel, const float *arc_weight,
                        const float *final_weight, int32_t start_state, int32_t num_pdfs) {
  if (!out) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (num_states <= 0 || num_arcs < 0 || num_pdfs <= 0 || !final_weight) return TC_ERR_INVALID_ARGUMENT;
  if (num_arcs > 0 && (!arc_src || !arc_dst || !arc_ilabel || !arc_weight)) return TC_ERR_INVALID_ARGUMENT;
  if (start_state < 0 || start_state >= num_states) return TC_ERR_BAD_FST;
  if (num_arcs >= (int64_t)1 << 30) return TC_ERR_UNSUPPORTED;
  for (int64_t a = 0; a < num_arcs; ++a) {
    if (arc_src[a] < 0 || arc_src[a] >= num_states || arc_dst[a] < 0 || arc_dst[a] >= num_states)
      return TC_ERR_BAD_FST;
    if (arc_ilabel[a] < 1 || arc_ilabel[a] > num_pdfs) return TC_ERR_BAD_FST;  // [K] KALDI_ASSERT on pdf_id
    if (a > 0 && arc_src[a] < arc_src[a - 1]) return TC_ERR_BAD_FST;           // must be state-major
    if (!(arc_weight[a] == arc_weight[a])) return TC_ERR_BAD_FST;
  }
  tc_den_graph *g = new tc_den_graph();
  g->H = num_states;
  g->P = num_pdfs;
  g->A = num_arcs;
  g->arc_src.assign(arc_src, arc_src + num_arcs);
  g->arc_dst.assign(arc_dst, arc_dst + num_arcs);
  g->arc_pdf.resize(num_arcs);
  g->arc_prob.resize(num_arcs);
  for (int64_t a = 0; a < num_arcs; ++a) {
    g->arc_pdf[a] = arc_ilabel[a] - 1;
    g->arc_prob[a] = (float)std::exp(-(double)arc_weight[a]);  // [K] SetTransitions
  }
  // [K] DenominatorGraph::SetInitialProbs: start state only, 100 rounds of per-state-normalised HMM
  // propagation, averaged; double precision, stored as float.
  {
    const int H = num_states;
    std::vector<double> norm(H), cur(H, 0.0), nxt(H, 0.0), avg(H, 0.0);
    for (int s = 0; s < H; ++s) norm[s] = std::exp(-(double)final_weight[s]);
    for (int64_t a = 0; a < num_arcs; ++a) norm[arc_src[a]] += std::exp(-(double)arc_weight[a]);
    for (int s = 0; s < H; ++s) norm[s] = 1.0 / norm[s];
    cur[start_state] = 1.0;
    for (int iter = 0; iter < 100; ++iter) {
      for (int s = 0; s < H; ++s) avg[s] += cur[s] * (1.0 / 100);
      for (int64_t a = 0; a < num_arcs; ++a) {
        int s = arc_src[a];
        nxt[arc_dst[a]] += cur[s] * norm[s] * std::exp(-(double)arc_weight[a]);
      }
      double sum = 0.0;
      for (int s = 0; s < H; ++s) sum += nxt[s];
      for (int s = 0; s < H; ++s) {
        cur[s] = nxt[s] * (1.0 / sum);
        nxt[s] = 0.0;
      }
    }
    g->initial_probs.resize(H);
    for (int s = 0; s < H; ++s) g->initial_probs[s] = (float)avg[s];
  }
  build_schedules(g);
  *out = g;
  return TC_OK;
}

}  // extern "C"

// --- OpenFst binary VectorFst<StdArc> reader ------------------------------------------------------
namespace {
struct Reader {
  FILE *f;
  bool ok = true;
  template <class T>
  T get() {
    T v{};
    if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
    return v;
  }
  std::string str() {
    int32_t n = get<int32_t>();
    if (!ok || n < 0 || n > (1 << 20)) {
      ok = false;
      return std::string();
    }
    std::string s((size_t)n, '\0');
    if (n && fread(&s[0], 1, (size_t)n, f) != (size_t)n) ok = false;
    return s;
  }
  void skip_symbol_table() {
    int32_t magic = get<int32_t>();
    if (magic != 2125658996) {
      ok = false;
      return;
    }
    str();                   // name
    get<int64_t>();          // available key
    int64_t size = get<int64_t>();
    for (int64_t i = 0; ok && i < size; ++i) {
      str();
      get<int64_t>();
    }
  }
};
}  // namespace

extern "C" {

int tc_den_graph_read(tc_den_graph **out, const char *rxfilename, int32_t num_pdfs) {
  if (!out || !rxfilename) return TC_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  FILE *f = fopen(rxfilename, "rb");
  if (!f) return TC_ERR_IO;
  Reader r{f};
  int rc = TC_OK;
  std::vector<int32_t> src, dst, il;
  std::vector<float> w, fin;
  int64_t start = 0, nstates = 0;
  do {
    if (r.get<int32_t>() != 2125659606) { rc = TC_ERR_IO; break; }  // kFstMagicNumber
    std::string fsttype = r.str(), arctype = r.str();
    int32_t version = r.get<int32_t>(), flags = r.get<int32_t>();
    r.get<uint64_t>();  // properties
    start = r.get<int64_t>();
    nstates = r.get<int64_t>();
    int64_t narcs = r.get<int64_t>();
    if (!r.ok || fsttype != "vector" || arctype != "standard" || version < 2 || (flags & 4) /* aligned */ ||
        nstates <= 0 || nstates > (1 << 28)) {
      rc = TC_ERR_IO;
      break;
    }
    if (flags & 1) r.skip_symbol_table();
    if (flags & 2) r.skip_symbol_table();
    if (narcs > 0) { src.reserve(narcs); dst.reserve(narcs); il.reserve(narcs); w.reserve(narcs); }
    fin.resize(nstates);
    for (int64_t s = 0; r.ok && s < nstates; ++s) {
      fin[s] = r.get<float>();
      int64_t n = r.get<int64_t>();
      if (!r.ok || n < 0 || n > (1 << 28)) { r.ok = false; break; }
      for (int64_t i = 0; r.ok && i < n; ++i) {
        int32_t ilabel = r.get<int32_t>();
        r.get<int32_t>();  // olabel (== ilabel in a den.fst)
        float weight = r.get<float>();
        int32_t next = r.get<int32_t>();
        src.push_back((int32_t)s);
        dst.push_back(next);
        il.push_back(ilabel);
        w.push_back(weight);
      }
    }
    if (!r.ok) rc = TC_ERR_IO;
  } while (0);
  fclose(f);
  if (rc != TC_OK) return rc;
  return tc_den_graph_create(out, (int32_t)nstates, (int64_t)src.size(), src.data(), dst.data(), il.data(), w.data(),
                             fin.data(), (int32_t)start, num_pdfs);
}

void tc_den_graph_free(tc_den_graph *g) {
  if (!g) return;
  for (auto &kv : g->dev) {
    if (kv.second.blob) {
      int cur = 0;
      if (hipGetDevice(&cur) == hipSuccess) {
        (void)hipSetDevice(kv.first);
        (void)hipFree(kv.second.blob);
        (void)hipSetDevice(cur);
      }
    }
  }
  delete g;
}

int32_t tc_den_graph_num_states(const tc_den_graph *g) { return g ? g->H : 0; }
int64_t tc_den_graph_num_arcs(const tc_den_graph *g) { return g ? g->A : 0; }
int32_t tc_den_graph_num_pdfs(const tc_den_graph *g) { return g ? g->P : 0; }

int tc_den_graph_initial_probs(const tc_den_graph *g, float *out_host) {
  if (!g || !out_host) return TC_ERR_INVALID_ARGUMENT;
  memcpy(out_host, g->initial_probs.data(), sizeof(float) * (size_t)g->H);
  return TC_OK;
}

int tc_den_graph_stats(const tc_den_graph *g, int64_t *o) {
  if (!g || !o) return TC_ERR_INVALID_ARGUMENT;
  o[0] = g->fwd.padded_arcs;
  o[1] = g->bwd.padded_arcs;
  o[2] = g->layout_ok ? layout_lds_bytes(g->layout, 150) : -1;
  o[3] = kThreads;
  o[4] = g->fwd.rows;
  o[5] = g->bwd.rows;
  o[6] = g->fwd.conflict_free_cost ? 1000 * g->fwd.conflict_cost / g->fwd.conflict_free_cost : 0;
  o[7] = g->bwd.conflict_free_cost ? 1000 * g->bwd.conflict_cost / g->bwd.conflict_free_cost : 0;
  o[8] = g->big ? 2 : (g->tied ? 1 : 0);
  return TC_OK;
}

// Host-side replay of the schedules (diagnostic; never on the product path): performs one arc walk the
// way the kernels consume the streams -- cell order, row-end masks, commit cursor, secondary rows and
// fix-up lists, position permutation, per-state tables -- and returns per state
//   direction 0: sum over in-arcs  (h -> g) of w * gather[h] * pdf_factor[pdf]
//   direction 1: sum over out-arcs (h -> g) of w * gather[g] * pdf_factor[pdf]
// so that tests without a GPU can compare the built schedules with the definition.
int tc_den_graph_debug_walk(const tc_den_graph *g, int direction, const float *gather, const float *pdf_factor,
                            float *out) {
  if (!g || !gather || !pdf_factor || !out || direction < 0 || direction > 1) return TC_ERR_INVALID_ARGUMENT;
  const int H = g->H;
  if (g->big) {
    const std::vector<int32_t> &begin = direction == 0 ? g->big_in_begin : g->big_out_begin;
    const std::vector<BigArc> &arc = direction == 0 ? g->big_in : g->big_out;
    for (int h = 0; h < H; ++h) {
      float sum = 0.f;
      for (int a = begin[h]; a < begin[h + 1]; ++a) sum += arc[a].w * gather[arc[a].a] * pdf_factor[arc[a].b];
      out[h] = sum;
    }
    return TC_OK;
  }
  if (!g->layout_ok) return TC_ERR_UNSUPPORTED;
  const ScheduleHost &sc = direction == 0 ? g->fwd : g->bwd;
  const int Hs = g->layout.Hs;
  std::vector<float> acc((size_t)g->layout.acc_floats + 64, 0.f);
  if (g->tied) {
    const int K = Hs / kThreads;
    std::vector<float> src_pos((size_t)Hs + 4, 0.f);
    // a split state's alpha is the sum of its copies' (forward: the first copy carries the value), its
    // beta is shared by all copies (backward: every copy presents it)
    for (int h = 0; h < H; ++h)
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c) {
        const uint32_t fs = g->tied_fs[g->pos[c]];
        src_pos[g->pos[c]] = direction == 0 ? (c == g->copy_first[h] ? gather[h] : 0.f)
                                            : gather[h] * pdf_factor[(fs & 0xffffu) >> 2];
      }
    for (int w = 0; w < kWaves; ++w) {
      const int first = sc.wave_range[w].x, n = sc.wave_range[w].y;
      for (int l = 0; l < 64; ++l) {
        const int tid = 64 * w + l;
        int k = 0;
        float ax = 0.f, ay = 0.f;
        auto slot = [&]() {
          return k < K ? 4 * (tid + kThreads * (k >> 2)) + (k & 3) : Hs + 4 + 64 * (sc.extra_first[w] + (k - K)) + l;
        };
        auto cell = [&](int i, float *wgt, int *position) {
          const size_t c = (size_t)first + i, chunk = c / 8, q = c % 8;
          const uint32_t *base = &sc.cells6[chunk * 3 * 64 * 4];
          uint32_t x = base[((q / 4) * 64 + l) * 4 + (q % 4)];
          memcpy(wgt, &x, 4);
          const uint32_t o = base[(2 * 64 + l) * 4 + q / 2];
          *position = (int)(((q & 1) ? o >> 16 : o & 0xffffu) >> 2);
        };
        for (int i = 0; i < n; i += 2) {
          float w0, w1;
          int p0, p1;
          cell(i, &w0, &p0);
          cell(i + 1, &w1, &p1);
          const uint32_t m = sc.masks[(size_t)w * sc.mask_stride + (i / 2) / 8];
          const int bit = (i / 2) % 8;
          if ((m >> (8 + bit)) & 1u) {  // the row ends with the pair's first cell
            acc[slot()] = (ax + w0 * src_pos[p0]) + ay;
            ++k;
            ax = 0.f;
            ay = w1 * src_pos[p1];
          } else {
            ax += w0 * src_pos[p0];
            ay += w1 * src_pos[p1];
          }
          if ((m >> bit) & 1u) {
            acc[slot()] = ax + ay;
            ++k;
            ax = ay = 0.f;
          }
        }
      }
    }
    for (int t = 0; t < kThreads; ++t)
      for (int e = sc.fix_begin[t]; e < sc.fix_begin[t + 1]; ++e) acc[sc.fix[e].x] += acc[sc.fix[e].y];
    for (int h = 0; h < H; ++h) {
      float sum = 0.f;
      for (int c = g->copy_first[h]; c < g->copy_first[h + 1]; ++c) {
        const int p = g->pos[c];
        const uint32_t fs = g->tied_fs[p];
        const float self = pdf_factor[fs >> 18] * g->tied_w[p] * (direction == 0 && c != g->copy_first[h] ? 0.f : gather[h]);
        const float v = direction == 0 ? pdf_factor[(fs & 0xffffu) >> 2] * acc[p] + self : acc[p] + self;
        if (direction == 1) {  // every copy computes the state's out-sum: take the first
          sum = v;
          break;
        }
        sum += v;
      }
      out[h] = sum;
    }
    return TC_OK;
  }
  // general schedules: 8-byte cells [pair][lane][2] with in-band ROW cells
  for (int w = 0; w < kWaves; ++w) {
    const int first = sc.wave_range[w].x, n = sc.wave_range[w].y;
    for (int l = 0; l < 64; ++l) {
      int cur = Hs;  // dummy row
      float a = 0.f;
      for (int i = 0; i < n; ++i) {
        const size_t c = (size_t)first + i;
        const ArcRec &r = sc.cells[((c >> 1) * 64 + l) * 2 + (c & 1)];
        if (r.idx & kRowFlag) {
          acc[cur] = a;
          uint32_t x;
          memcpy(&x, &r.w, 4);
          cur = (int)(x & 0xffffu);
          a = 0.f;
        } else {
          a += r.w * gather[std::min<uint32_t>(r.idx >> 18, (uint32_t)H - 1)] * pdf_factor[(r.idx >> 2) & 0x3fffu];
        }
      }
    }
  }
  for (int t = 0; t < kThreads; ++t)
    for (int e = sc.fix_begin[t]; e < sc.fix_begin[t + 1]; ++e) acc[sc.fix[e].x] += acc[sc.fix[e].y];
  for (int h = 0; h < H; ++h) out[h] = acc[h];
  return TC_OK;
}

int tc_den_graph_prepare(tc_den_graph *g, int device) {
  if (!g) return TC_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(g->mu);
  if (g->dev.count(device)) return TC_OK;
  int prev = 0;
  TC_HIP_CHECK(hipGetDevice(&prev));
  TC_HIP_CHECK(hipSetDevice(device));
  const int Hs = (g->H + 3) & ~3;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  struct Part { const void *src; size_t bytes; size_t off; };
  if (g->big) {
    std::vector<float> pi_pad(Hs + 4, 0.f);
    std::copy(g->initial_probs.begin(), g->initial_probs.end(), pi_pad.begin());
    Part parts[] = {
        {g->big_in_begin.data(), g->big_in_begin.size() * 4, 0}, {g->big_in.data(), g->big_in.size() * sizeof(BigArc), 0},
        {g->big_out_begin.data(), g->big_out_begin.size() * 4, 0}, {g->big_out.data(), g->big_out.size() * sizeof(BigArc), 0},
        {g->big_pdf_begin.data(), g->big_pdf_begin.size() * 4, 0}, {g->big_pdf.data(), g->big_pdf.size() * sizeof(BigArc), 0},
        {pi_pad.data(), pi_pad.size() * 4, 0},
    };
    size_t total = 0;
    for (auto &p : parts) {
      p.off = total;
      total += align(p.bytes);
    }
    char *blob = nullptr;
    hipError_t e = hipMalloc((void **)&blob, total);
    if (e == hipSuccess)
      for (auto &p : parts) {
        e = hipMemcpy(blob + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
      }
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
      g_last_hip_error = (int)e;
      return TC_ERR_HIP;
    }
    DenGraphDev d;
    d.blob = blob;
    d.big.in_begin = (const int32_t *)(blob + parts[0].off);
    d.big.in_arc = (const BigArc *)(blob + parts[1].off);
    d.big.out_begin = (const int32_t *)(blob + parts[2].off);
    d.big.out_arc = (const BigArc *)(blob + parts[3].off);
    d.big.pdf_begin = (const int32_t *)(blob + parts[4].off);
    d.big.pdf_arc = (const BigArc *)(blob + parts[5].off);
    d.pi = (const float *)(blob + parts[6].off);
    g->dev[device] = d;
    return TC_OK;
  }
  std::vector<float> pi_pad(Hs + 4, 0.f);
  std::copy(g->initial_probs.begin(), g->initial_probs.end(), pi_pad.begin());
  if (g->tied) pi_pad = g->pi_pos;  // position order (build_owner)
  const std::vector<uint32_t> no_mask(1, 0u);
  const std::vector<int32_t> no_extra(kWaves, 0);
  Part parts[] = {
      {g->fwd.cells6.empty() ? (const void *)g->fwd.cells.data() : (const void *)g->fwd.cells6.data(),
       g->fwd.cells6.empty() ? g->fwd.cells.size() * sizeof(ArcRec) : g->fwd.cells6.size() * 4, 0},
      {g->fwd.wave_range.data(), g->fwd.wave_range.size() * sizeof(int2), 0},
      {g->bwd.cells6.empty() ? (const void *)g->bwd.cells.data() : (const void *)g->bwd.cells6.data(),
       g->bwd.cells6.empty() ? g->bwd.cells.size() * sizeof(ArcRec) : g->bwd.cells6.size() * 4, 0},
      {g->bwd.wave_range.data(), g->bwd.wave_range.size() * sizeof(int2), 0},
      {pi_pad.data(), pi_pad.size() * 4, 0},
      {g->fwd.fix_begin.data(), g->fwd.fix_begin.size() * 4, 0},
      {g->fwd.fix.data(), g->fwd.fix.size() * sizeof(int2), 0},
      {g->bwd.fix_begin.data(), g->bwd.fix_begin.size() * 4, 0},
      {g->bwd.fix.data(), g->bwd.fix.size() * sizeof(int2), 0},
      {g->tied_fs.data(), g->tied_fs.size() * 4, 0},
      {g->tied_w.data(), g->tied_w.size() * 4, 0},
      {g->fwd.masks.empty() ? no_mask.data() : g->fwd.masks.data(), std::max<size_t>(1, g->fwd.masks.size()) * 4, 0},
      {g->bwd.masks.empty() ? no_mask.data() : g->bwd.masks.data(), std::max<size_t>(1, g->bwd.masks.size()) * 4, 0},
      {g->fwd.extra_first.empty() ? no_extra.data() : g->fwd.extra_first.data(), kWaves * 4, 0},
      {g->bwd.extra_first.empty() ? no_extra.data() : g->bwd.extra_first.data(), kWaves * 4, 0},
  };
  size_t total = 0;
  for (auto &p : parts) {
    p.off = total;
    total += align(p.bytes);
  }
  char *blob = nullptr;
  hipError_t e = hipMalloc((void **)&blob, total);
  if (e == hipSuccess) {
    for (auto &p : parts) {
      e = hipMemcpy(blob + p.off, p.src, p.bytes, hipMemcpyHostToDevice);
      if (e != hipSuccess) break;
    }
  }
  (void)hipSetDevice(prev);
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return TC_ERR_HIP;
  }
  DenGraphDev d;
  d.blob = blob;
  d.fwd = ScheduleDev{(const ArcRec *)(blob + parts[0].off), (const int2 *)(blob + parts[1].off),
                      (const int32_t *)(blob + parts[5].off), (const int2 *)(blob + parts[6].off),
                      (const uint32_t *)(blob + parts[11].off), (const int32_t *)(blob + parts[13].off),
                      g->fwd.mask_stride, g->fwd.nfix};
  d.bwd = ScheduleDev{(const ArcRec *)(blob + parts[2].off), (const int2 *)(blob + parts[3].off),
                      (const int32_t *)(blob + parts[7].off), (const int2 *)(blob + parts[8].off),
                      (const uint32_t *)(blob + parts[12].off), (const int32_t *)(blob + parts[14].off),
                      g->bwd.mask_stride, g->bwd.nfix};
  d.pi = (const float *)(blob + parts[4].off);
  if (g->tied) {
    d.tied_fs = (const uint32_t *)(blob + parts[9].off);
    d.tied_w = (const float *)(blob + parts[10].off);
  }
  g->dev[device] = d;
  return TC_OK;
}

const char *tc_strerror(int code) {
  switch (code) {
    case TC_OK: return "ok";
    case TC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TC_ERR_BAD_FST: return "malformed FST";
    case TC_ERR_UNSUPPORTED: return "problem size not supported by the on-chip layout";
    case TC_ERR_WORKSPACE: return "workspace missing or too small";
    case TC_ERR_HIP: return "HIP runtime error";
    case TC_ERR_IO: return "cannot read OpenFst vector/standard file";
    case TC_ERR_NOT_SEPARABLE: return "merged supervision FST does not factor per sequence";
    default: return "unknown error";
  }
}

int tc_version(void) { return 100; }
int tc_last_hip_error(void) { return tc::g_last_hip_error; }

}  // extern "C"
