// LDS layouts of the on-chip denominator kernels and the bank-conflict-aware step placement shared by
// both schedule builders.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {



// Lays the per-frame working set of one sequence out in LDS.  Returns false if it cannot fit.
bool compute_layout(int H, int P, int T_hint, int extra_slots, bool tied, DenLayout *L, bool need_alpha) {
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
  } else if (tied && jv <= kJvMid && pv <= kPvLarge) {
    // tied graphs of 8193..12288 positions (12 states per thread): the registers the fourth float4 of every per-state
    // array would take go to resident stream chunks (den_tied_kernel.hip: res_fwd / res_bwd)
    L->JV = kJvMid;
    L->PV = pv <= kPvSmall ? kPvSmall : pv <= kPvMid ? kPvMid : kPvLarge;
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
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int with_alpha = need_alpha ? 1 : 1 - attempt;
    const bool asum_global = need_alpha && attempt == 1;
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
    off += asum_global ? 4 : round4(T_hint + 1);
    L->total_floats = off;
    L->alpha_in_lds = with_alpha != 0;
    L->planewise = false;
    L->asum_global = asum_global;
    if ((int64_t)off * 4 <= kLdsLimitBytes) return true;
  }
  return false;
}

// The plane-wise form (chain_internal.h: kJvPlanes): exp(y) in its compile-time region, the gather source for every
// position, four accumulator rows per wave that all planes share (+ the dummy row and the private slots of secondary
// rows), gamma, the reduction scratch and the frame sums.  One exp(y) buffer, alpha'_{t+1} from the history (the tight
// layout's rules).
bool compute_layout_planes(int Npos, int P, int T_hint, int extra_slots, DenLayout *L) {
  L->Hs = Npos;
  L->Ps = round4(P);
  if (Npos % (4 * kThreads) != 0 || Npos > kMaxSplitPositions || L->Ps > 4 * kThreads * kPvSmall) return false;
  L->JV = Npos / (4 * kThreads);  // the kernel is instantiated per plane count (den_tied_planes.hip)
  if (L->JV < 5) return false;
  L->PV = kPvSmall;
  L->planewise = true;
  // beyond kMaxPlanePositions the gather source is in LDS a half at a time (chain_internal.h: kJvPlanesSplit)
  L->src_planes = Npos > kMaxPlanePositions ? (L->JV + 1) / 2 : L->JV;
  int off = L->PV * 4 * kThreads;
  L->off_a = off;
  off += 4 * kThreads * L->src_planes;
  L->off_acc = off;
  L->acc_floats = round4(4 * kThreads + 4 + extra_slots);
  off += L->acc_floats;
  L->off_g = off;
  off += L->Ps;
  L->off_al = L->off_p2 = L->off_red = off;
  off += 4 * kWaves;
  L->off_asum = off;
  L->asum_global = (int64_t)(off + round4(T_hint + 1)) * 4 > kLdsLimitBytes;  // (a long utterance: the sums go to the workspace)
  off += L->asum_global ? 4 : round4(T_hint + 1);
  L->total_floats = off;
  L->alpha_in_lds = false;
  return (int64_t)off * 4 <= kLdsLimitBytes;
}

int64_t layout_lds_bytes(const DenLayout &L, int T) {
  return 4 * (int64_t)(L.off_asum + (L.asum_global ? 4 : round4(T + 1)));
}

// Bank-conflict-aware placement for one 32-lane half of a slot.  ds_read_b32 / ds_add_u32 service a
// wave in two 32-lane groups, one LDS cycle per distinct address per bank (bank = dword index mod 32;
// profiles/microbench: 2.3 cycles conflict-free, 7.0 for uniformly random gathers).  The sum over a
// row is order-independent and padding may sit anywhere, so for every step we pick, per lane, the arc
// of its row whose state-bank and pdf-bank are still free in that step; a few swap passes then remove
// what the greedy pass left.  pos[l][k] = index into lane l's arc list, or -1 for padding.
// Returns the cost sum_k (max state-bank multiplicity + max pdf-bank multiplicity).
int arrange_half(const std::vector<std::vector<int64_t>> &lane_arcs, int steps, const int32_t *other,
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
    if (debug_flag(kDbgSchedTrace)) {
      static long long n = 0, sst = 0, spd = 0;
      n++; sst += std::max(ms, 1); spd += std::max(mp, 1);
      if (n % 2000 == 0) fprintf(stderr, "[sched] steps=%lld avg max-mult state=%.3f pdf=%.3f\n", n, (double)sst / n, (double)spd / n);
    }
  }
  *pos_out = pos;
  return total;
}

// The same problem for schedules with ONE gather per cell (tied graphs), solved step by step as a matching problem.
// A half-slot is a bipartite multigraph lanes x banks; it cannot take fewer than D = max(steps, most loaded bank)
// LDS cycles, and it takes exactly D when every step of cost c (c lanes on its busiest bank) lowers that bound by c:
// the step must serve every lane that has no slack left, and every bank b at least load(b) - (D - c) times.  Both are
// degree constraints of a bipartite b-matching (lanes capacity 1, banks capacity c), found with augmenting paths --
// first from the bank slots that must be filled, then from the lanes that must move (an augmenting path keeps every
// matched vertex matched).  When no step of any cost keeps the bound, the step that loses least is taken.  (The greedy
// placement above spreads a heavy bank's surplus over the steps one conflict at a time, wherever it meets another
// lane; the bound needs the surpluses of different banks to fall into the SAME steps.)
int arrange_half_matching(const std::vector<std::vector<int64_t>> &lane_arcs, int steps, const int32_t *other,
                          std::vector<std::vector<int>> *pos_out, std::vector<std::vector<int>> *pad_bank, int *lower_bound) {
  const int L = (int)lane_arcs.size();
  constexpr int NB = 32, CAP = 64;
  std::vector<std::array<std::vector<int>, NB>> by_bank(L);
  std::vector<int> rem(L);
  int d[NB] = {0};
  for (int l = 0; l < L; ++l) {
    rem[l] = (int)lane_arcs[l].size();
    for (int i = 0; i < rem[l]; ++i) {
      const int b = other[lane_arcs[l][i]] & (NB - 1);
      by_bank[l][b].push_back(i);
      d[b]++;
    }
  }
  std::vector<std::vector<int>> pos(L, std::vector<int>(steps, -1)), padb(L, std::vector<int>(steps, 0));
  if (lower_bound) *lower_bound = std::max(steps, *std::max_element(d, d + NB));
  std::vector<int> match_lane(L), match_slot(NB * CAP);
  std::vector<char> vis_lane(L), vis_slot(NB * CAP);
  int c = 1;
  // slot s = bank * CAP + i, i < c
  std::function<bool(int)> try_slot = [&](int s) -> bool {  // a free slot looks for a lane
    const int b = s / CAP;
    for (int l = 0; l < L; ++l) {
      if (vis_lane[l] || by_bank[l][b].empty()) continue;
      if (match_lane[l] >= 0 && match_lane[l] / CAP == b) continue;
      vis_lane[l] = 1;
      if (match_lane[l] < 0 || try_slot(match_lane[l])) {
        match_lane[l] = s;
        match_slot[s] = l;
        return true;
      }
    }
    return false;
  };
  std::function<bool(int)> try_lane = [&](int l) -> bool {  // a free lane looks for a slot
    for (int pass = 0; pass < 2; ++pass)  // free slots first
      for (int b = 0; b < NB; ++b) {
        if (by_bank[l][b].empty()) continue;
        for (int i = 0; i < c; ++i) {
          const int s = b * CAP + i;
          if (pass == 0) {
            if (match_slot[s] < 0) {
              match_slot[s] = l;
              match_lane[l] = s;
              return true;
            }
            continue;
          }
          if (vis_slot[s] || match_slot[s] < 0) continue;
          vis_slot[s] = 1;
          if (try_lane(match_slot[s])) {
            match_slot[s] = l;
            match_lane[l] = s;
            return true;
          }
        }
      }
    return false;
  };
  for (int k = 0; k < steps; ++k) {
    const int r = steps - k;
    const int D = std::max(r, *std::max_element(d, d + NB));
    bool done = false;
    for (int loss = 0; !done; ++loss)
      for (c = 1; c <= std::min(D - r + 1 + loss, CAP) && !done; ++c) {
        const int Dn = D - c + loss;
        int need[NB];
        bool ok = true;
        for (int b = 0; b < NB; ++b) {
          need[b] = std::max(0, d[b] - Dn);
          if (need[b] > c) ok = false;
        }
        if (!ok) continue;
        std::fill(match_lane.begin(), match_lane.end(), -1);
        std::fill(match_slot.begin(), match_slot.end(), -1);
        for (int b = 0; b < NB && ok; ++b)
          for (int i = 0; i < need[b] && ok; ++i) {
            std::fill(vis_lane.begin(), vis_lane.end(), 0);
            ok = try_slot(b * CAP + i);
          }
        for (int l = 0; l < L && ok; ++l)
          if (rem[l] == r && match_lane[l] < 0) {
            std::fill(vis_slot.begin(), vis_slot.end(), 0);
            ok = try_lane(l);
          }
        if (!ok) continue;
        for (int l = 0; l < L; ++l)
          if (rem[l] > 0 && match_lane[l] < 0) {
            std::fill(vis_slot.begin(), vis_slot.end(), 0);
            try_lane(l);
          }
        done = true;
      }
    int usage[NB] = {0};
    for (int l = 0; l < L; ++l)
      if (match_lane[l] >= 0) {
        const int b = match_lane[l] / CAP;
        pos[l][k] = by_bank[l][b].back();
        by_bank[l][b].pop_back();
        d[b]--;
        rem[l]--;
        usage[b]++;
      }
    for (int l = 0; l < L; ++l)
      if (match_lane[l] < 0) {
        const int b = (int)(std::min_element(usage, usage + NB) - usage);
        padb[l][k] = b;
        usage[b]++;
      }
  }
  // LDS cycles as placed: per step, the most distinct addresses on one bank (equal addresses are one broadcast)
  int total = 0;
  for (int k = 0; k < steps; ++k) {
    std::array<std::vector<int32_t>, NB> addr;
    for (int l = 0; l < L; ++l) {
      const int32_t a = pos[l][k] >= 0 ? other[lane_arcs[l][pos[l][k]]] : padb[l][k];
      auto &v = addr[a & (NB - 1)];
      if (std::find(v.begin(), v.end(), a) == v.end()) v.push_back(a);
    }
    size_t mx = 1;
    for (auto &v : addr) mx = std::max(mx, v.size());
    total += (int)mx;
  }
  *pos_out = pos;
  if (pad_bank) *pad_bank = padb;
  return total;
}

}  // namespace tc
