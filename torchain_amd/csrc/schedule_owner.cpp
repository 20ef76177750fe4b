// Schedules of the tied on-chip kernel (owner-computes), the tied test and the state splitting that
// makes nearly tied graphs tied.
#include <algorithm>
#include <map>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {

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
// Arc lists longer than max_row (kMaxRowLen, or more when the private slots would not fit) keep their first max_row
// arcs at home; the rest become secondary rows (k >= K) of the SAME wave, commit to private slots behind the
// accumulators and are folded in by the owner lane after the wave's walk -- no barrier: one wave's LDS operations
// execute in order.
struct OwnerTask {
  int32_t state;   // original state id (or -1: empty)
  int64_t begin;   // range into the direction's arc order
  int32_t len;
};

// pdf != nullptr: GENERAL graphs (den_general_owner.hip) -- a cell also carries the LDS offset of its arc's pdf in exp(y)
// (8-byte cells: {w, position * 4 | pdf * 4 << 16}), and the placement inside a half-slot weighs both gathers' banks.
// slots_b != nullptr: split gather source (chain_internal.h: kJvPlanesSplit) -- `slots` holds the rows' cells whose source lies
// below position `half_pos`, `slots_b` the others; a wave's stream is [sub-streams of the first half | of the second].
static void emit_owner_stream(int Npos, int K, const std::vector<std::vector<std::vector<OwnerTask>>> &slots,
                              const std::vector<int64_t> &order, const int32_t *opos, const float *prob,
                              ScheduleHost *out, bool planewise, bool count_only, const int32_t *pdf = nullptr, int num_pdfs = 1,
                              const std::vector<std::vector<std::vector<OwnerTask>>> *slots_b = nullptr, int half_pos = 0) {
  const int halves = slots_b ? 2 : 1;
  auto slots_of = [&](int hv) -> const std::vector<std::vector<std::vector<OwnerTask>>> & { return hv ? *slots_b : slots; };
  // slots[w][k] = 64 tasks (lane order); k >= K are secondary rows.  count_only: the number of cells alone (padded_arcs)
  if (count_only) {
    int64_t cells = 0;
    for (int hv = 0; hv < halves; ++hv)
      for (int w = 0; w < kWaves; ++w) {
        const auto &sl = slots_of(hv)[w];
        int64_t in_sub = 0;
        auto close_sub = [&]() {
          cells += (in_sub + kStreamUnrollTied - 1) / kStreamUnrollTied * kStreamUnrollTied * 64;
          in_sub = 0;
        };
        for (size_t k = 0; k < sl.size(); ++k) {
          int steps = 1;
          for (const OwnerTask &t : sl[k]) steps = std::max(steps, t.len);
          in_sub += steps;
          if (planewise && (k + 1 == sl.size() || ((int)k < K && k % 4 == 3))) close_sub();
        }
        close_sub();
      }
    out->padded_arcs = cells;
    return;
  }
  out->conflict_cost = out->conflict_free_cost = out->conflict_bound = 0;
  out->cells.clear();
  // plane-wise form: per wave, sub-stream 0 = its secondary rows, sub-stream 1 + j = the four rows of plane j
  const int subs_half = planewise ? K / 4 + 1 : 1;  // sub-streams per half of the gather source
  const int subs = subs_half * halves;
  const int off_shift = planewise ? 16 : 18;  // the cell's 16-bit offset field: position (inside its half), or position * 4
  out->subs = planewise ? subs_half : 0;
  out->halves = halves;
  out->wave_range.assign((size_t)kWaves * subs, make_int2(0, 0));
  std::vector<std::vector<uint32_t>> wave_masks((size_t)kWaves * subs);
  int64_t arc_cells = 0;
  int nrows = 0;
  for (int w = 0; w < kWaves; ++w)
    for (int sub_all = 0; sub_all < subs; ++sub_all) {
      const int hv = sub_all / subs_half, sub = sub_all % subs_half;
      const auto &wslots = slots_of(hv)[w];
      const int32_t pos_base = hv ? half_pos : 0;
      const size_t first = out->cells.size() / 64;
      size_t cells_before = 0;
      std::vector<char> row_end;  // per pair of this stream: flags A | B
      const size_t k0 = !planewise ? 0 : sub == 0 ? (size_t)K : (size_t)4 * (sub - 1);
      const size_t k1 = !planewise ? wslots.size() : sub == 0 ? wslots.size() : (size_t)4 * sub;
      for (size_t k = k0; k < k1; ++k) {
        const auto &tasks = wslots[k];
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
          std::vector<std::vector<int>> pos, padb;
          if (pdf) {
            out->conflict_cost += arrange_half(lane_arcs, steps, opos, pdf, &pos);
            out->conflict_free_cost += steps;  // (two gathers per cell)
          } else if (debug_flag(kDbgOldArrange)) {
            out->conflict_cost += arrange_half(lane_arcs, steps, opos, nullptr, &pos);
          } else {
            int lb = 0;
            out->conflict_cost += arrange_half_matching(lane_arcs, steps, opos, &pos, &padb, &lb);
            out->conflict_bound += lb;
          }
          out->conflict_free_cost += steps;
          for (int l = 0; l < 32; ++l) {
            const int lane = half * 32 + l;
            for (int i = 0; i < steps; ++i) {
              ArcRec &cell = out->cells[off + (size_t)i * 64 + lane];
              if (pos[l][i] >= 0) {
                const int64_t a = lane_arcs[l][pos[l][i]];
                cell = ArcRec{prob[a], (uint32_t)(opos[a] - pos_base) << off_shift};
                if (pdf) cell.idx = ((uint32_t)opos[a] << 2) | ((uint32_t)pdf[a] << 18);
              } else {
                // padding: w = 0, gathered from a bank that is idle in this step
                cell = ArcRec{0.f, (uint32_t)(Npos >= 32 ? (padb.empty() ? l : padb[l][i]) : 0) << off_shift};
                // (general cells: w = 0 adds nothing, but the backward walk still issues the cell's gamma atomic -- 64 lanes
                // on ONE address serialise, and a short stream is half padding: every lane gets a pdf of its own)
                if (pdf) cell.idx = ((uint32_t)(Npos >= 32 ? l : 0) << 2) | ((uint32_t)((half * 32 + l) % std::max(1, num_pdfs)) << 18);
              }
            }
          }
        }
        cells_before += steps;
        row_end.resize((cells_before + 1) / 2, 0);
        // bit 0 (B): the row ends with the pair's second cell; bit 1 (A): with its first cell
        row_end[(cells_before - 1) / 2] |= ((cells_before - 1) & 1) ? 1 : 2;
      }
      // whole chunks, and (one stream per wave) at least kTiedMinChunks of them: the walks keep a prefix in registers
      while ((out->cells.size() / 64 - first) % kStreamUnrollTied != 0 ||
             (!planewise && out->cells.size() / 64 - first < (size_t)kTiedMinChunks * kStreamUnrollTied))
        for (int l = 0; l < 64; ++l) out->cells.push_back(ArcRec{0.f, pdf ? ((uint32_t)l << 2) | ((uint32_t)(l % std::max(1, num_pdfs)) << 18) : 0u});
      row_end.resize((out->cells.size() / 64 - first + 1) / 2, 0);  // the padding cells end no row
      if (debug_flag(kDbgSchedTrace))
        fprintf(stderr, "[sched] wave %d half %d sub %d: %zu cells, %zu rows\n", w, hv, sub, out->cells.size() / 64 - first, k1 - k0);
      if (planewise) {
        // one stream per wave, cut at chunk boundaries: {first cell of the sub-stream, its END counted from the wave's first
        // cell}, and ONE byte of row-end bits per chunk (bit i: a row ends with cell i), four chunks to a word
        const size_t wave_first = (size_t)out->wave_range[(size_t)w * subs].x;
        const size_t rel = sub_all == 0 ? 0 : first - wave_first;
        if (sub_all == 0) out->wave_range[(size_t)w * subs].x = (int)first;
        out->wave_range[(size_t)w * subs + sub_all] = make_int2((int)first, (int)(rel + out->cells.size() / 64 - first));
        auto &mw = wave_masks[(size_t)w * subs];  // (the wave's words: index 0 of its sub-streams)
        mw.resize((rel + out->cells.size() / 64 - first) / 8 / 4 + 1, 0u);
        for (size_t i = 0; i < row_end.size(); ++i)
          for (int half = 0; half < 2; ++half)
            if (row_end[i] & (half ? 1 : 2)) {  // flag A: the pair's first cell, flag B: its second
              const size_t cell = rel + 2 * i + half, chunk = cell / 8;
              mw[chunk / 4] |= 1u << (8 * (chunk % 4) + cell % 8);
            }
      } else {
        out->wave_range[(size_t)w * subs + sub_all] = make_int2((int)first, (int)(out->cells.size() / 64 - first));
        auto &mw = wave_masks[(size_t)w * subs + sub_all];
        mw.assign((row_end.size() + 7) / 8, 0u);
        for (size_t i = 0; i < row_end.size(); ++i) {
          if (row_end[i] & 1) mw[i / 8] |= 1u << (i % 8);        // B flags: bits 0..7
          if (row_end[i] & 2) mw[i / 8] |= 1u << (8 + i % 8);    // A flags: bits 8..15
        }
      }
    }
  // readable padding: the kernels request up to four chunks past a wave's range
  for (int i = 0; i < 64 * 32; ++i) out->cells.push_back(ArcRec{0.f, 0u});
  size_t stride = 1;
  for (auto &mw : wave_masks) stride = std::max(stride, mw.size());
  if (!planewise) stride += 2;  // (the walk once prefetched a word ahead)
  out->mask_stride = (int32_t)stride;
  if (planewise) {  // [wave][stride]: the wave's words (kept at index 0 of its sub-streams above)
    out->masks.assign(stride * kWaves + 64, 0u);  // (+ a register's worth: the kernel reads 64 words per wave)
    for (int w = 0; w < kWaves; ++w) {
      const auto &mw = wave_masks[(size_t)w * subs];
      std::copy(mw.begin(), mw.end(), out->masks.begin() + (size_t)w * stride);
    }
  } else {
    out->masks.assign(stride * wave_masks.size() + 64, 0u);
    for (size_t i = 0; i < wave_masks.size(); ++i) std::copy(wave_masks[i].begin(), wave_masks[i].end(), out->masks.begin() + i * stride);
  }
  if (debug_flag(kDbgSchedTrace))
    fprintf(stderr, "[sched] gathers: %lld LDS cycles as placed, bound %lld, conflict-free %lld\n", (long long)out->conflict_cost,
            (long long)out->conflict_bound, (long long)out->conflict_free_cost);
  out->real_arcs = (int64_t)order.size();
  out->padded_arcs = arc_cells;
  out->rows = nrows;
  const size_t ncell = out->cells.size() / 64;
  if (pdf) {
    // 8-byte cells, [chunk of 8 cells][4 blocks][lane]{16 bytes}: {w0..w3}, {w4..w7}, {idx0..idx3}, {idx4..idx7}
    out->cells6.assign(ncell / 8 * 4 * 64 * 4, 0u);
    for (size_t c = 0; c < ncell; ++c)
      for (int l = 0; l < 64; ++l) {
        const ArcRec &cell = out->cells[c * 64 + l];
        uint32_t x;
        memcpy(&x, &cell.w, 4);
        const size_t chunk = c / 8, i = c % 8;
        uint32_t *base = &out->cells6[chunk * 4 * 64 * 4];
        base[((i / 4) * 64 + l) * 4 + (i % 4)] = x;
        base[((2 + i / 4) * 64 + l) * 4 + (i % 4)] = cell.idx;
      }
    out->cells.clear();
    out->cells.shrink_to_fit();
    return;
  }
  // 6-byte cells, [chunk of 8 cells][3 blocks][lane]{16 bytes}: {w0..w3}, {w4..w7}, {off01, off23, off45, off67}
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
      const uint32_t off16 = cell.idx >> 16;  // position * 4 (plane-wise form: position)
      o |= (i & 1) ? off16 << 16 : off16;
    }
  out->cells.clear();
  out->cells.shrink_to_fit();
}

// Returns false when the graph cannot use the owner-computes kernel (too many states for the 16-bit
// offsets or the working set does not fit LDS); the caller then falls back to the general kernel.
bool build_owner(tc_den_graph *g, const std::vector<char> &special, int max_row, bool count_only, bool general) {
  const int H = g->work_H;  // states of the work graph (tc_den_graph::work_*)
  // (the per-state tables in work-state order: a successful build leaves them in position order, and a graph may be built
  // more than once -- den_graph.cpp tries several row cuts)
  if (!general) {
    g->tied_fs = g->tied_fs_state;
    g->tied_w = g->tied_w_state;
  }
  const int Npos = 4096 * ((H + 4095) / 4096);
  // beyond 16384 positions: the plane-wise form (chain_internal.h: kJvPlanes), whose rows need not be short either
  // (measured on graphs the 12- / 16-states-per-thread kernels hold: the plane-wise form is 40 % slower there -- X1 6.6 vs 4.7 ms,
  // R2 6.3 vs 4.5, R3 4.9 vs 3.4 --, its per-state values go through L2 where theirs sit in registers)
  const bool planewise = Npos > kMaxIndex;
  if (Npos > kMaxSplitPositions || (planewise && debug_flag(kDbgNoPlanes))) return false;
  // beyond kMaxPlanePositions the gather source is in LDS a half at a time: every row is cut into the cells whose source
  // lies in the first ceil(planes / 2) planes and the others (chain_internal.h: kJvPlanesSplit)
  const bool split_src = Npos > kMaxPlanePositions && !debug_flag(kDbgNoSplitSrc);
  if (Npos > kMaxPlanePositions && !split_src) return false;
  const int half_pos = split_src ? 4096 * ((Npos / 4096 + 1) / 2) : 0;
  // general graphs (den_general_owner.hip): 8 states per thread, alpha'_t of the owned states in LDS
  if (general && Npos > 4 * kThreads * kJvSmall) return false;
  const int K = Npos / kThreads;
  std::vector<int32_t> src, dst, apdf;
  std::vector<float> prob;
  for (int64_t a = 0; a < (int64_t)g->work_src.size(); ++a)
    if (general || !special[a]) {
      src.push_back(g->work_src[a]);
      dst.push_back(g->work_dst[a]);
      prob.push_back(g->work_prob[a]);
      if (general) apdf.push_back(g->work_pdf[a]);
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

  // ---- split gather source: which states live in the first half of the positions.  A row is walked in two parts -- the cells
  // whose source lies in the first half, then the others -- and the 64 rows of a slot run in lockstep, so a part costs its
  // slot the LONGEST of 64.  Halves dealt without regard to the arcs make the parts Binomial(length, 1/2): X2's 360 000 cells
  // took 625 000 slots.  So the states are first divided such that every row's two parts are as equal as the graph allows: a
  // local search on the sum over all rows (both directions) of (first part - second part)^2.
  std::vector<char> in_b(H, 0);  // 1: the state lives in the second half
  const int planes_all = Npos / 4096, planes_first = (planes_all + 1) / 2;
  int n_first = H;
  if (split_src) {
    n_first = std::min<int64_t>(half_pos, std::max<int64_t>((int64_t)H - (Npos - half_pos), ((int64_t)H * planes_first + planes_all - 1) / planes_all));
    for (int h = n_first; h < H; ++h) in_b[h] = 1;  // (FST order to start with)
    std::vector<int32_t> d_in(H, 0), d_out(H, 0);    // per row: cells with the source in the first half minus the others
    for (int64_t a = 0; a < A2; ++a) {
      d_in[dst[a]] += in_b[src[a]] ? -1 : 1;
      d_out[src[a]] += in_b[dst[a]] ? -1 : 1;
    }
    // moving state u to the other half changes d by -+2 in every row that gathers u
    auto move_delta = [&](int u, int sign, bool apply) {  // sign = -1: u leaves the first half
      int64_t delta = 0;
      for (int64_t i = out_first[u]; i < out_first[u + 1]; ++i) {  // arcs u -> x: the in-row of x gathers u
        int32_t &d = d_in[dst[out_order[i]]];
        delta += (int64_t)(d + 2 * sign) * (d + 2 * sign) - (int64_t)d * d;
        if (apply) d += 2 * sign;
      }
      for (int64_t i = in_first[u]; i < in_first[u + 1]; ++i) {  // arcs x -> u: the out-row of x gathers u
        int32_t &d = d_out[src[in_order[i]]];
        delta += (int64_t)(d + 2 * sign) * (d + 2 * sign) - (int64_t)d * d;
        if (apply) d += 2 * sign;
      }
      return delta;
    };
    uint64_t rng = 0xA0761D6478BD642Full;
    auto next = [&]() {
      rng ^= rng << 13;
      rng ^= rng >> 7;
      rng ^= rng << 17;
      return rng;
    };
    // Single moves inside the slack the phantom positions leave (a half may hold any number of states up to its positions),
    // swept over the states in random order until a sweep finds nothing, then trades of two states of different halves.
    // How far it gets: a blind deal leaves a mean squared difference of 9 per row on X2 (rows of 9 cells: Binomial), this
    // search 2.9; trades alone, ten times as many proposals or annealing end between 2.6 and 2.9 as well -- 80 000 rows
    // constrain 40 000 binary choices, so most rows cannot be even.  X2's 360 000 cells take 510 000 slots (blind: 625 000).
    int64_t accepted = 0, count[2] = {n_first, H - n_first};
    const int64_t room[2] = {half_pos, Npos - half_pos};
    std::vector<int32_t> visit(H);
    std::iota(visit.begin(), visit.end(), 0);
    const int sweeps = count_only ? 4 : 40;
    for (int sweep = 0; sweep < sweeps; ++sweep) {
      for (int i = H - 1; i > 0; --i) std::swap(visit[i], visit[(size_t)(next() % (uint64_t)(i + 1))]);
      int64_t moved = 0;
      for (int u : visit) {
        const int to = in_b[u] ? 0 : 1;
        if (count[to] >= room[to]) continue;
        const int su = in_b[u] ? 1 : -1;
        const int64_t du = move_delta(u, su, false);
        if (du < 0 || (du == 0 && (next() & 7) == 0)) {
          move_delta(u, su, true);
          in_b[u] = (char)to;
          count[to]++;
          count[1 - to]--;
          ++moved;
        }
      }
      accepted += moved;
      if (moved == 0) break;
    }
    const int64_t proposals = count_only ? 0 : (int64_t)H * 50;
    for (int64_t it = 0; it < proposals; ++it) {
      const uint64_t r = next();
      const int u = (int)(r % (uint64_t)H), v = (int)((r >> 32) % (uint64_t)H);
      if (in_b[u] == in_b[v]) continue;
      const int su = in_b[u] ? 1 : -1;  // u moves to the other half, v the opposite way
      // (apply u's move first: the two may gather each other or share a row)
      const int64_t du = move_delta(u, su, true), dv = move_delta(v, -su, true);
      if (du + dv < 0) {
        std::swap(in_b[u], in_b[v]);
        ++accepted;
      } else {
        move_delta(v, su, true);
        move_delta(u, -su, true);
      }
    }
    if (debug_flag(kDbgSchedTrace)) {
      int64_t sq = 0;
      for (int h = 0; h < H; ++h) sq += (int64_t)d_in[h] * d_in[h] + (int64_t)d_out[h] * d_out[h];
      fprintf(stderr, "[sched] split source: %lld moves and trades taken, sum of squared part differences %lld over %d rows, %lld + %lld states\n",
              (long long)accepted, (long long)sq, 2 * H, (long long)count[0], (long long)count[1]);
    }
  }
  // the length a row costs its slot: the whole row, or (split source) its two parts one after the other
  std::vector<int32_t> len_in(H), len_out(H);
  for (int h = 0; h < H; ++h) {
    len_in[h] = std::min(deg(in_first, h), max_row);
    len_out[h] = std::min(deg(out_first, h), max_row);
  }
  std::vector<int32_t> part_in[2], part_out[2];  // split source: cells of the row by the half of their source
  if (split_src) {
    for (int hv = 0; hv < 2; ++hv) {
      part_in[hv].assign(H, 0);
      part_out[hv].assign(H, 0);
    }
    for (int64_t a = 0; a < A2; ++a) {
      part_in[(int)in_b[src[a]]][dst[a]]++;
      part_out[(int)in_b[dst[a]]][src[a]]++;
    }
    for (int h = 0; h < H; ++h) {
      len_in[h] = std::min(part_in[0][h], max_row) + std::min(part_in[1][h], max_row);
      len_out[h] = std::min(part_out[0][h], max_row) + std::min(part_out[1][h], max_row);
    }
  }
  auto lin = [&](int h) { return (int)len_in[h]; };
  auto lout = [&](int h) { return (int)len_out[h]; };
  auto sort_states = [&](std::vector<int32_t> &st) {
    const int H = (int)st.size();  // (the states of this list)
    if (H == 0) return;
  // ---- the permutation: sort by primary in-length into a few super-buckets, inside by primary out-length
  std::stable_sort(st.begin(), st.end(), [&](int x, int y) { return lin(x) > lin(y); });
    // How many super-buckets: few make the out-lengths of a group uniform, many the in-lengths.  Which matters depends on the
  // graph (a phone-LM graph has near-constant out-degrees and in-degrees from 1 to hundreds: sqrt(groups) buckets left 7 % of
  // its forward cells as padding), so a handful of counts is tried and the one with the fewest padded steps kept.
  {
    const std::vector<int32_t> by_in(st);
    const int root = std::max(1, (int)std::lround(std::sqrt((double)std::max(1, (H + 63) / 64))));
    int64_t best_steps = -1;
    std::vector<int32_t> best;
    for (int nbucket : {1, std::max(1, root / 2), root, 2 * root, 4 * root, 8 * root, 16 * root, std::max(1, (H + 63) / 64)}) {
      std::vector<int32_t> cand(by_in);
      for (int b = 0; b < nbucket; ++b) {
        const size_t lo = (size_t)H * b / nbucket, hi = (size_t)H * (b + 1) / nbucket;
        std::stable_sort(cand.begin() + lo, cand.begin() + hi, [&](int x, int y) { return lout(x) > lout(y); });
      }
      int64_t steps = 0;
      for (size_t g0 = 0; g0 < cand.size(); g0 += 64) {
        int mi = 1, mo = 1;
        for (size_t i = g0; i < std::min(cand.size(), g0 + 64); ++i) {
          mi = std::max(mi, lin(cand[i]));
          mo = std::max(mo, lout(cand[i]));
        }
        steps += mi + mo;
      }
      if (best_steps < 0 || steps < best_steps) {
        best_steps = steps;
        best.swap(cand);
      }
    }
    st.swap(best);
  }
  };
  const int ngroups = Npos / 64;
  std::vector<int32_t> st;
  if (!split_src) {
    st.resize(H);
    std::iota(st.begin(), st.end(), 0);
    sort_states(st);
  } else {
    // each half sorted by itself and padded with phantom states to its share of the positions: [first half | second half]
    std::vector<int32_t> half_list[2];
    for (int h = 0; h < H; ++h) half_list[(int)in_b[h]].push_back(h);
    sort_states(half_list[0]);
    sort_states(half_list[1]);
    half_list[0].resize(half_pos, -1);
    half_list[1].resize(Npos - half_pos, -1);
    st = half_list[0];
    st.insert(st.end(), half_list[1].begin(), half_list[1].end());
  }
  // Inside runs of equal (in, out) length the order is free: use it so that every 32 consecutive states --
  // one half-slot, i.e. the 32 lanes that gather exp(y) at f(g) / s(g) and add gamma there in ONE
  // instruction of the per-state passes -- have distinct pdf banks (greedy, first fit).
  if (!debug_flag(kDbgNoPdfBanks) && !general && !split_src) {  // (split source: the list has phantoms in its middle)
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
  // ... then a local search on what the instructions really cost.  A half-slot's gather or add takes as many LDS turns as its
  // most loaded bank holds states; first fit leaves C3 at 2.04 turns per half-slot and instruction where the banks' totals would
  // allow 1.3, and the gamma adds alone are 0.1 ms of C3's launch (profiles/r05_ablations.txt §1b).  Two states of different
  // half-slots trade places when neither has longer rows than the other's group already walks (so no walk gains a step) and
  // 64 * (sum over half-slots of max f-bank load + max s-bank load) + (sum of squared loads) does not rise; plateau moves are
  // taken, the squares pull the loads flat so that the maxima can fall later.
  if (!debug_flag(kDbgNoPdfBanks) && !debug_flag(kDbgNoPdfSearch) && !general && !count_only && H > 64 && !split_src) {  // (the trades keep every group's longest rows: nothing a count-only pass counts changes)
    const int nhalf = (H + 31) / 32, ngr = (H + 63) / 64;
    std::vector<int> mi(ngr, 1), mo(ngr, 1);
    for (int i = 0; i < H; ++i) {
      mi[i / 64] = std::max(mi[i / 64], lin(st[i]));
      mo[i / 64] = std::max(mo[i / 64], lout(st[i]));
    }
    struct Half {
      uint8_t cnt[2][32];
      uint8_t lvl[2][34];  // number of banks holding c states
      uint8_t mx[2];
    };
    std::vector<Half> hv(nhalf);
    for (auto &h : hv) {
      memset(&h, 0, sizeof(h));
      h.lvl[0][0] = h.lvl[1][0] = 32;
    }
    int64_t sum_max = 0, sum_sq = 0;
    auto adjust = [&](int half, int which, int bank, int delta) {
      Half &h = hv[half];
      const int old = h.cnt[which][bank], now = old + delta;
      h.cnt[which][bank] = (uint8_t)now;
      h.lvl[which][old]--;
      h.lvl[which][now]++;
      sum_sq += (int64_t)now * now - (int64_t)old * old;
      if (now > h.mx[which]) {
        sum_max += now - h.mx[which];
        h.mx[which] = (uint8_t)now;
      } else if (old == h.mx[which] && h.lvl[which][old] == 0) {
        sum_max -= 1;
        h.mx[which] = (uint8_t)now;
      }
    };
    auto fb = [&](int h) { return (int)(((g->tied_fs[h] & 0xffffu) >> 2) & 31u); };
    auto sb = [&](int h) { return (int)((g->tied_fs[h] >> 18) & 31u); };
    for (int i = 0; i < H; ++i) {
      adjust(i / 32, 0, fb(st[i]), 1);
      adjust(i / 32, 1, sb(st[i]), 1);
    }
    const int64_t max_before = sum_max;
    auto hub = [&](int h) { return deg(in_first, h) > max_row || deg(out_first, h) > max_row; };
    uint64_t rng = 0xD1B54A32D192ED03ull;
    auto next = [&]() {
      rng ^= rng << 13;
      rng ^= rng >> 7;
      rng ^= rng << 17;
      return rng;
    };
    const int span = std::min(H, 4096);
    const int64_t proposals = (int64_t)H * 200;
    int64_t accepted = 0;
    for (int64_t it = 0; it < proposals; ++it) {
      const uint64_t r = next();
      const int i = (int)(r % (uint64_t)H);
      const int u = st[i], hi = i / 32;
      // only a state that sits on a most loaded bank of its half-slot can lower a maximum by leaving
      if (!((hv[hi].mx[0] > 1 && hv[hi].cnt[0][fb(u)] == hv[hi].mx[0]) || (hv[hi].mx[1] > 1 && hv[hi].cnt[1][sb(u)] == hv[hi].mx[1]))) continue;
      const int j = i + (int)((r >> 32) % (uint64_t)(2 * span)) - span;
      if (j < 0 || j >= H || j / 32 == hi) continue;
      const int v = st[j], hj = j / 32;
      if (lin(u) > mi[j / 64] || lout(u) > mo[j / 64] || lin(v) > mi[i / 64] || lout(v) > mo[i / 64] || hub(u) || hub(v)) continue;
      const int64_t before = 64 * sum_max + sum_sq;
      adjust(hi, 0, fb(u), -1);
      adjust(hi, 1, sb(u), -1);
      adjust(hj, 0, fb(v), -1);
      adjust(hj, 1, sb(v), -1);
      adjust(hj, 0, fb(u), 1);
      adjust(hj, 1, sb(u), 1);
      adjust(hi, 0, fb(v), 1);
      adjust(hi, 1, sb(v), 1);
      if (64 * sum_max + sum_sq <= before) {
        std::swap(st[i], st[j]);
        ++accepted;
      } else {
        adjust(hi, 0, fb(v), -1);
        adjust(hi, 1, sb(v), -1);
        adjust(hj, 0, fb(u), -1);
        adjust(hj, 1, sb(u), -1);
        adjust(hj, 0, fb(v), 1);
        adjust(hj, 1, sb(v), 1);
        adjust(hi, 0, fb(u), 1);
        adjust(hi, 1, sb(u), 1);
      }
    }
    if (debug_flag(kDbgSchedTrace))
      fprintf(stderr, "[sched] pdf-bank search: %lld of %lld swaps taken, bank turns of the per-state instructions %lld -> %lld (%d half-slots x 2)\n",
              (long long)accepted, (long long)proposals, (long long)max_before, (long long)sum_max, nhalf);
  }
  st.resize(Npos, -1);  // phantom states: no arcs, pi = 0
  struct Group { int idx, cin, cout, pin, pout; };  // (pin / pout: steps of the group's own rows, without its secondary rows)
  std::vector<Group> groups(ngroups);
  for (int gi = 0; gi < ngroups; ++gi) {
    int mi = 1, mo = 1;
    int64_t over_in = 0, over_out = 0;  // its states' arcs beyond max_row become secondary rows of the same wave, 64 to a slot
    if (!split_src) {
      for (int l = 0; l < 64; ++l) {
        const int h = st[(size_t)gi * 64 + l];
        if (h < 0) continue;
        mi = std::max(mi, lin(h));
        mo = std::max(mo, lout(h));
        over_in += std::max(0, deg(in_first, h) - max_row);
        over_out += std::max(0, deg(out_first, h) - max_row);
      }
    } else {  // the slot's steps: the longest first part + the longest second part of its 64 rows
      mi = mo = 0;
      for (int hv = 0; hv < 2; ++hv) {
        int pi_ = 1, po_ = 1;
        for (int l = 0; l < 64; ++l) {
          const int h = st[(size_t)gi * 64 + l];
          if (h < 0) continue;
          pi_ = std::max(pi_, std::min(part_in[hv][h], max_row));
          po_ = std::max(po_, std::min(part_out[hv][h], max_row));
          over_in += std::max(0, part_in[hv][h] - max_row);
          over_out += std::max(0, part_out[hv][h] - max_row);
        }
        mi += pi_;
        mo += po_;
      }
    }
    groups[gi] = Group{gi, mi, mo, mi, mo};
    groups[gi].cin += (int)((over_in + 63) / 64);
    groups[gi].cout += (int)((over_out + 63) / 64);
  }
  // longest-processing-time deal of the groups to the waves, K per wave, balancing both directions
  // Plane-wise form: its frames run without issue priorities (den_tied_planes.hip), the CU then serves its OLDER waves
  // first wherever waves contend, and the walks of the four wave generations proceed at about 1.25 : 1.15 : 0.95 : 0.8 of
  // the mean (profiles/r05/pw_stamps3.txt); with equal shares the old waves wait for the young ones at every frame's
  // barrier while the stream path runs half empty, so the shares follow the speeds.
  double wave_speed[kWaves];
  {
    static const double by_gen[4] = {1.25, 1.15, 0.95, 0.8};
    // (measured with the deviations scaled by 0 / 0.5 / 1 / 1.5: 9.87 / 9.78 / 9.73 / 9.68 ms on R4, profiles/r05/r05_pw_skew.txt; steeper
    // ladders move R4 by -2 .. +0.5 % without order and a random 28000-state graph not at all: profiles/r05/ab_pw_shares.txt)
    // The one-stream kernels do run issue priorities, youngest generation first (den_tied_device.h: age_prio_on), and there the
    // young waves are the fast ones: forward walks of 21.6 / 17.2 / 15.4 / 14.6 k cycles by generation on R2 with equal shares
    // (profiles/r05_ablations.txt §4).  Equal finishing times are not the aim -- the early waves' per-state passes run beside the
    // late waves' walks -- but a share of 0.7 : 1 : 1.15 : 1.2 is worth 2-4 % on graphs with ten or more chunks per wave.
    static const double by_gen_prio[4] = {0.7, 1.0, 1.15, 1.2};
    for (int w = 0; w < kWaves; ++w) wave_speed[w] = planewise ? by_gen[w / 4] : general ? 1.0 : by_gen_prio[w / 4];
  }
  std::vector<Group> by_cost(groups);
  std::stable_sort(by_cost.begin(), by_cost.end(), [](const Group &x, const Group &y) { return x.cin + x.cout > y.cin + y.cout; });
  std::vector<std::vector<int>> wave_groups(kWaves);
  std::vector<int64_t> load_in(kWaves, 0), load_out(kWaves, 0);
  std::vector<int> taken[2] = {std::vector<int>(kWaves, 0), std::vector<int>(kWaves, 0)};
  for (const Group &gr : by_cost) {
    // (round 3, before the issue priorities: shares skewed towards the older waves of each SIMD, which the CU serves first, were measured: no
    // gain -- the walk is bound by the shared stream path, not by any one wave)
    int best = -1;
    double best_t = 0;
    // (split source: a wave's first 4 * planes_first slots are positions of the first half, and a group is of one half)
    const int gh = split_src && gr.idx >= half_pos / 64 ? 1 : 0;
    for (int w = 0; w < kWaves; ++w) {
      if (!split_src ? (int)wave_groups[w].size() >= K : taken[gh][w] >= (gh ? K - 4 * planes_first : 4 * planes_first)) continue;
      const double tw = (double)std::max(load_in[w] + gr.cin, load_out[w] + gr.cout) / wave_speed[w];
      if (best < 0 || tw < best_t) {
        best = w;
        best_t = tw;
      }
    }
    wave_groups[best].push_back(gr.idx);
    taken[gh][best]++;
    load_in[best] += gr.cin;
    load_out[best] += gr.cout;
  }
  if (split_src)  // first-half groups to the front: slot k < 4 * planes_first is a position of the first half
    for (int w = 0; w < kWaves; ++w)
      std::stable_partition(wave_groups[w].begin(), wave_groups[w].end(), [&](int gi) { return gi < half_pos / 64; });
  // A wave's stream is padded to whole chunks of 8 steps, and a chunk of padding costs a walk as much as a chunk of arcs:
  // dealt by load alone, 10 of C3's 16 forward streams were 57 steps -- 8 chunks -- long.  Groups trade places between
  // waves while that lowers the number of chunks (first the longest stream's, then the sum over waves and directions).
  if (!planewise) {
    auto chunks = [](int64_t steps) { return (std::max<int64_t>(steps, (int64_t)kTiedMinChunks * kStreamUnrollTied) + kStreamUnrollTied - 1) / kStreamUnrollTied; };
    auto cost = [&]() {
      double mx_in = 0, mx_out = 0;
      int64_t sum = 0;
      for (int w = 0; w < kWaves; ++w) {
        mx_in = std::max(mx_in, (double)chunks(load_in[w]) / wave_speed[w]);
        mx_out = std::max(mx_out, (double)chunks(load_out[w]) / wave_speed[w]);
        sum += chunks(load_in[w]) + chunks(load_out[w]);
      }
      return (int64_t)std::llround((mx_in + mx_out) * 1000.0) * 1000 + sum;
    };
    bool improved = true;
    for (int round = 0; round < 50 && improved; ++round) {
      improved = false;
      for (int w1 = 0; w1 < kWaves; ++w1)
        for (int w2 = w1 + 1; w2 < kWaves; ++w2)
          for (size_t i1 = 0; i1 < wave_groups[w1].size(); ++i1)
            for (size_t i2 = 0; i2 < wave_groups[w2].size(); ++i2) {
              const Group &a = groups[wave_groups[w1][i1]], &b = groups[wave_groups[w2][i2]];
              if (a.cin == b.cin && a.cout == b.cout) continue;
              const int64_t before = cost();
              load_in[w1] += b.cin - a.cin;
              load_out[w1] += b.cout - a.cout;
              load_in[w2] += a.cin - b.cin;
              load_out[w2] += a.cout - b.cout;
              if (cost() < before) {
                std::swap(wave_groups[w1][i1], wave_groups[w2][i2]);
                improved = true;
              } else {
                load_in[w1] -= b.cin - a.cin;
                load_out[w1] -= b.cout - a.cout;
                load_in[w2] -= a.cin - b.cin;
                load_out[w2] -= a.cout - b.cout;
              }
            }
    }
  }
  // ... and the longest streams go to the youngest waves, which the kernels run at the highest issue priority
  // (den_tied_device.h: age_prio_on): every frame waits for its slowest wave
  if (!planewise) {
    std::vector<int> order(kWaves);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return load_in[x] + load_out[x] < load_in[y] + load_out[y]; });
    std::vector<std::vector<int>> wg(kWaves);
    std::vector<int64_t> li(kWaves), lo(kWaves);
    for (int w = 0; w < kWaves; ++w) {
      wg[w] = wave_groups[order[w]];
      li[w] = load_in[order[w]];
      lo[w] = load_out[order[w]];
    }
    wave_groups.swap(wg);
    load_in.swap(li);
    load_out.swap(lo);
  }
  // Plane-wise form: a plane's sub-stream is padded to whole chunks in both directions (3.5 steps on average, six or seven
  // planes per wave and direction: 7 % of R4's forward cells).  Which of a wave's groups share a plane is free: groups
  // trade planes inside their wave while that lowers the wave's padding.
  if (planewise) {
    auto pad8 = [](int x) { return (8 - x % 8) % 8; };
    for (int w = 0; w < kWaves; ++w) {
      std::vector<int> &wg = wave_groups[w];
      const int planes = K / 4;
      auto plane_cost = [&](int pl) {
        int si = 0, so = 0;
        for (int c = 0; c < 4; ++c) {
          si += groups[wg[4 * pl + c]].pin;
          so += groups[wg[4 * pl + c]].pout;
        }
        return pad8(si) + pad8(so);
      };
      bool improved = true;
      for (int round = 0; round < 20 && improved; ++round) {
        improved = false;
        for (int a = 0; a < K; ++a)
          for (int b = a + 1; b < K; ++b) {
            if (a / 4 == b / 4) continue;
            if (split_src && (a / 4 < planes_first) != (b / 4 < planes_first)) continue;  // (a group stays in its half)
            const int before = plane_cost(a / 4) + plane_cost(b / 4);
            std::swap(wg[a], wg[b]);
            if (plane_cost(a / 4) + plane_cost(b / 4) < before)
              improved = true;
            else
              std::swap(wg[a], wg[b]);
          }
      }
      if (debug_flag(kDbgSchedTrace)) {
        int total = 0;
        for (int pl = 0; pl < planes; ++pl) total += plane_cost(pl);
        fprintf(stderr, "[sched] wave %d: %d steps of plane padding\n", w, total);
      }
    }
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
  if (!debug_flag(kDbgNoBankSearch) && !count_only) {
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
    auto phi = [](int h) -> int64_t { return (int64_t)h * h; };
    // moving state u from bank b1 to bank b2 changes sum h^2 by the sum over the rows gathering u
    auto move_delta = [&](int u, int b1, int b2) {
      int64_t d = 0;
      for (int64_t i = out_first[u]; i < out_first[u + 1]; ++i) {  // arcs u -> x: row of x gathers u (forward)
        const int hf = half_of(g->pos[dst[out_order[i]]]);
        int32_t &x1 = H_(0, hf, b1), &x2 = H_(0, hf, b2);
        d += (phi(x2 + 1) - phi(x2)) - (phi(x1) - phi(x1 - 1));
        --x1;
        ++x2;
      }
      for (int64_t i = in_first[u]; i < in_first[u + 1]; ++i) {  // arcs x -> u: row of x gathers u (backward)
        const int hf = half_of(g->pos[src[in_order[i]]]);
        int32_t &x1 = H_(1, hf, b1), &x2 = H_(1, hf, b2);
        d += (phi(x2 + 1) - phi(x2)) - (phi(x1) - phi(x1 - 1));
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
    if (debug_flag(kDbgSchedTrace)) {
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
    // Split gather source: a state's list is put in the order [cells whose source lies below half_pos | the others] and
    // becomes TWO rows, one per half; everything below is done per half (halves == 1: the one "half" is the whole list).
    const int halves = split_src ? 2 : 1;
    std::vector<int64_t> order_h(order);
    std::vector<int32_t> len_a(H, 0);  // cells of the state's list in the first half
    if (split_src)
      for (int h = 0; h < H; ++h) {
        auto b0 = order_h.begin() + first[h], b1 = order_h.begin() + first[h + 1];
        len_a[h] = (int32_t)(std::stable_partition(b0, b1, [&](int64_t a) { return opos[a] < half_pos; }) - b0);
      }
    auto part_begin = [&](int h, int hv) { return first[h] + (hv ? len_a[h] : 0); };
    auto part_len = [&](int h, int hv) { return !split_src ? deg(first, h) : hv ? deg(first, h) - len_a[h] : (int)len_a[h]; };
    std::vector<std::vector<std::vector<OwnerTask>>> slots_h[2];
    std::vector<std::vector<int2>> fix_of_thread[2];
    std::vector<std::vector<std::vector<OwnerTask>>> sec_slots_h[2];
    for (int hv = 0; hv < halves; ++hv) {
      auto &slots = slots_h[hv];
      slots.assign(kWaves, std::vector<std::vector<OwnerTask>>(K, std::vector<OwnerTask>(64)));
      // Secondary rows stay on the wave that owns their state: its lanes commit them to private slots and the owner
      // lane reads them back after the WAVE's walk -- LDS operations of one wave execute in order, so the fold needs
      // no workgroup barrier and a wave that has finished its walk goes on to its per-state pass while others still
      // walk (dealt to the least-loaded wave instead, every frame of a graph with popular states paid a barrier at
      // which the fast waves idled for 1-5 k cycles: profiles/r02_phase_stamps_r3.txt).
      std::vector<std::vector<OwnerTask>> secondary(kWaves);
      for (int w = 0; w < kWaves; ++w)
        for (int k = 0; k < K; ++k)
          for (int l = 0; l < 64; ++l) {
            const int p = 4 * ((64 * w + l) + kThreads * (k >> 2)) + (k & 3);
            const int h = state_at[p];
            OwnerTask t{h, 0, 0};
            if (h >= 0) {
              const int d = part_len(h, hv);
              t.begin = part_begin(h, hv);
              t.len = std::min(d, max_row);
              for (int done = t.len; done < d; done += max_row)
                secondary[w].push_back(OwnerTask{h, part_begin(h, hv) + done, std::min(max_row, d - done)});
            }
            slots[w][k][l] = t;
          }
      sec_slots_h[hv].assign(kWaves, {});
      for (int w = 0; w < kWaves; ++w) {
        std::vector<OwnerTask> &sec = secondary[w];
        std::stable_sort(sec.begin(), sec.end(), [](const OwnerTask &x, const OwnerTask &y) { return x.len > y.len; });
        for (size_t b = 0; b < sec.size(); b += 64) {
          std::vector<OwnerTask> tasks(64, OwnerTask{-1, 0, 0});
          for (size_t i = b; i < std::min(sec.size(), b + 64); ++i) tasks[i - b] = sec[i];
          sec_slots_h[hv][w].push_back(tasks);
        }
      }
    }
    // private slots: wave w's j-th secondary row, lane l -> accumulator index Npos + 4 + 64 * (extra_first[w] + j) + l
    // (the two halves of a split source use the same slots one after the other: a wave folds the first half's into its
    // planes' row sums before it walks the second half's)
    std::vector<int> extra_first(kWaves + 1, 0);
    for (int w = 0; w < kWaves; ++w) {
      size_t n = 0;
      for (int hv = 0; hv < halves; ++hv) n = std::max(n, sec_slots_h[hv][w].size());
      extra_first[w + 1] = extra_first[w] + (int)n;
    }
    for (int hv = 0; hv < halves; ++hv) {
      fix_of_thread[hv].assign(kThreads, {});
      for (int w = 0; w < kWaves; ++w)
        for (size_t j = 0; j < sec_slots_h[hv][w].size(); ++j) {
          for (int l = 0; l < 64; ++l) {
            const OwnerTask &t = sec_slots_h[hv][w][j][l];
            if (t.state < 0) continue;
            const int p = g->pos[t.state];
            fix_of_thread[hv][(p >> 2) % kThreads].push_back(make_int2(p, Npos + 4 + 64 * (extra_first[w] + (int)j) + l));
          }
          slots_h[hv][w].push_back(sec_slots_h[hv][w][j]);
        }
    }
    out->extra_first.assign(extra_first.begin(), extra_first.end() - 1);
    out->extra_slots = 64 * extra_first[kWaves];
    extra_total[dir] = out->extra_slots;
    out->fix.clear();
    if (planewise) {
      // per (half,) thread AND plane: the pass of a plane folds that plane's secondary rows
      const int planes = K / 4;
      out->fix_begin.assign((size_t)halves * kThreads * planes + 1, 0);
      for (int hv = 0; hv < halves; ++hv)
        for (int t = 0; t < kThreads; ++t)
          for (int j = 0; j < planes; ++j) {
            out->fix_begin[((size_t)hv * kThreads + t) * planes + j] = (int)out->fix.size();
            for (auto &f : fix_of_thread[hv][t])
              if (f.x / (4 * kThreads) == j) out->fix.push_back(f);
          }
      out->fix_begin[(size_t)halves * kThreads * planes] = (int)out->fix.size();
    } else {
      out->fix_begin.assign(kThreads + 1, 0);
      for (int t = 0; t < kThreads; ++t) {
        out->fix_begin[t] = (int)out->fix.size();
        for (auto &f : fix_of_thread[0][t]) out->fix.push_back(f);
      }
      out->fix_begin[kThreads] = (int)out->fix.size();
    }
    out->nfix = (int32_t)out->fix.size();
    if (out->fix.empty()) out->fix.push_back(make_int2(0, 0));
    emit_owner_stream(Npos, K, slots_h[0], order_h, opos.data(), prob.data(), out, planewise, count_only, general ? apdf.data() : nullptr, g->P,
                      split_src ? &slots_h[1] : nullptr, half_pos);
    if (count_only) continue;
    if (planewise && out->mask_stride > 64) return false;  // (a wave's mask words live in one register: at most 256 chunks)
  }
  if (planewise ? !compute_layout_planes(Npos, g->P, 256, std::max(extra_total[0], extra_total[1]), &g->layout)
                : !compute_layout(Npos, g->P, 256, std::max(extra_total[0], extra_total[1]), !general, &g->layout, general))
    return false;
  if (general) {
    // the walk takes a row's alpha'_t from LDS (one value per row and lane), rows of other lanes would need their owner's
    if (!g->layout.alpha_in_lds || g->layout.JV != kJvSmall || extra_total[1] != 0) return false;
    if (count_only) return true;
    g->pi_pos.assign(Npos + 4, 0.f);
    for (int h = 0; h < H; ++h) g->pi_pos[g->pos[h]] = g->work_pi[h];
    return true;
  }
  if (count_only) return true;
  // per-state tables in position order
  std::vector<uint32_t> fs(Npos + 4, 0u);
  std::vector<float> ws(Npos + 4, 0.f);
  g->pi_pos.assign(Npos + 4, 0.f);
  // Phantom positions (the last plane is rarely full) run the per-state passes like any other: their values are zero, and so
  // are their gamma adds -- but 64 adds of zero to ONE address (pdf 0) are still 64 turns of the LDS atomic unit.  They get the
  // pdf of their lane: distinct banks, nothing added anywhere.  (X1: 2384 phantom positions, 4.8 k cycles of every backward
  // frame: profiles/r05_ablations.txt §4.)
  if (!debug_flag(kDbgPhantomPdf0))
    for (int p = 0; p < Npos; ++p) {
      const uint32_t pdf = (uint32_t)(((p >> 2) % 64) % g->P);
      fs[p] = (pdf * 4u) | ((pdf * 4u) << 16);
    }
  for (int h = 0; h < H; ++h) {
    fs[g->pos[h]] = g->tied_fs[h];
    ws[g->pos[h]] = g->tied_w[h];
    g->pi_pos[g->pos[h]] = g->work_pi[h];
  }
  g->tied_fs.swap(fs);
  g->tied_w.swap(ws);
  if (debug_flag(kDbgSchedTrace)) {
    // the per-state passes' exp(y) gathers and gamma adds: one instruction = the 64 states of a (wave, k) group, served a
    // half at a time in (most loaded bank) cycles -- and an add cannot share an address
    int64_t gather_f = 0, gather_s = 0, add_f = 0, add_s = 0, halves = 0, same_f = 0, same_s = 0;
    for (int w = 0; w < kWaves; ++w)
      for (int k = 0; k < K; ++k)
        for (int hb = 0; hb < 2; ++hb) {
          int nf[32] = {0}, ns[32] = {0};
          std::vector<uint32_t> af[32], as[32];
          for (int l = hb * 32; l < hb * 32 + 32; ++l) {
            const int p = 4 * ((64 * w + l) + kThreads * (k >> 2)) + (k & 3);
            const uint32_t v = g->tied_fs[p], f = (v & 0xffffu) >> 2, s2 = v >> 18;
            nf[f & 31]++;
            ns[s2 & 31]++;
            if (std::find(af[f & 31].begin(), af[f & 31].end(), f) == af[f & 31].end()) af[f & 31].push_back(f);
            if (std::find(as[s2 & 31].begin(), as[s2 & 31].end(), s2) == as[s2 & 31].end()) as[s2 & 31].push_back(s2);
          }
          int mf = 0, ms = 0, gf = 0, gs2 = 0;
          {
            std::map<uint32_t, int> cf, cs;
            for (int l = hb * 32; l < hb * 32 + 32; ++l) {
              const int p = 4 * ((64 * w + l) + kThreads * (k >> 2)) + (k & 3);
              const uint32_t v = g->tied_fs[p];
              cf[(v & 0xffffu) >> 2]++;
              cs[v >> 18]++;
            }
            int a = 0, b2 = 0;
            for (auto &kv : cf) a = std::max(a, kv.second);
            for (auto &kv : cs) b2 = std::max(b2, kv.second);
            same_f += a;
            same_s += b2;
          }
          for (int b = 0; b < 32; ++b) {
            mf = std::max(mf, nf[b]);
            ms = std::max(ms, ns[b]);
            gf = std::max(gf, (int)af[b].size());
            gs2 = std::max(gs2, (int)as[b].size());
          }
          add_f += mf;
          add_s += ms;
          gather_f += gf;
          gather_s += gs2;
          ++halves;
        }
    fprintf(stderr, "[sched] per-state passes, %lld half-slots: gamma adds %lld (forward pdf) + %lld (self-loop pdf) cycles, exp(y) gathers %lld + %lld; most frequent pdf of a half-slot, summed: %lld + %lld\n",
            (long long)halves, (long long)add_f, (long long)add_s, (long long)gather_f, (long long)gather_s, (long long)same_f, (long long)same_s);
  }
  return true;
}

// Is the work graph tied?  Per state g: every non-self-loop in-arc carries one pdf f(g); self-loops that
// also carry f(g) are ordinary members of that class; at most one further self-loop (pdf s(g)) is
// "special" and is applied by the thread that owns g instead of travelling in the schedules.  Fills
// special[] and the per-state tables tied_fs / tied_w (work-state order).
bool detect_tied(tc_den_graph *g, std::vector<char> *special) {
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
  g->tied_f = fpdf;
  g->tied_s = spdf;
  for (int h = 0; h < H; ++h) {
    g->tied_fs[h] = (uint32_t)(std::max(fpdf[h], 0) * 4) | ((uint32_t)(std::max(spdf[h], 0) * 4) << 16);
    g->tied_w[h] = wself[h];
  }
  g->tied_fs_state = g->tied_fs;
  g->tied_w_state = g->tied_w;
  return true;
}

// Tied-ification.  Real chain graphs are tied except where minimisation merged two phone instances with
// the same self-loop pdf and future but different forward pdfs (LM back-off); one such state would send
// the whole graph to the general kernel.  Splitting state g into one copy per pdf that enters it is
// exact: the copies share g's out-arcs (and its special self-loop), so their futures are identical,
// beta(copy) = beta(g), alpha(g) = sum of the copies' alphas, and pi(g) may sit on any one of them.
// Every arc h -> g is replicated from every copy of h.  Returns false (graph left untouched) when the
// split graph would not pay (cost model below): arbitrary labelings are not chain graphs.
bool make_work_graph(tc_den_graph *g) {
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
  // Worth it?  Every arc is replicated from every copy of its source, and beyond 8192 states the tied kernel runs
  // with 16 states per thread; the general kernel pays two gathers and an atomic per arc instead.  Measured on
  // nearly tied graphs of 3000 / 6000 states with 60-90 % of the states entered through 2-3 pdfs (one MI355X,
  // 256 x 150, tied vs general): 1.77 vs 2.04 ms at 1.9x the states, 1.54 vs 2.04 at 2.3x; across the 8192-state
  // step 2.63 vs 3.26 at 1.9x but 3.56 vs 3.24 at 2.3x.
  if (WH > kMaxSplitPositions) return false;
  if (WH <= 8192 ? WH > (int64_t)H * 5 / 2 + 64 : WH > 2 * (int64_t)H + 64) return false;
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
    if ((int64_t)ws.size() > 4 * A + 1024) return false;
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

}  // namespace tc
