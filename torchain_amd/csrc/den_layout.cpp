// LDS layouts of the on-chip denominator kernels and the bank-conflict-aware step placement shared by
// both schedule builders.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>

#include "chain_internal.h"

namespace tc {



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

}  // namespace tc
