#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "torchain_hip.h"
// tc_supervision_create / tc_supervision_append over corrupted copies of valid arrays (ASan/UBSan build, host only).
// input file: int32 S, T, P, num_states, num_arcs; float weight; then arc_begin[ns+1], ilabel[na], next[na] (int32),
// arc_weight[na], final[ns] (float)
int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  int32_t hdr[5];
  float weight;
  if (fread(hdr, 4, 5, f) != 5 || fread(&weight, 4, 1, f) != 1) return 2;
  const int S = hdr[0], T = hdr[1], P = hdr[2], ns = hdr[3], na = hdr[4];
  std::vector<int32_t> ab(ns + 1), il(na), nx(na);
  std::vector<float> w(na), fin(ns);
  if (fread(ab.data(), 4, ns + 1, f) != (size_t)ns + 1 || fread(il.data(), 4, na, f) != (size_t)na ||
      fread(nx.data(), 4, na, f) != (size_t)na || fread(w.data(), 4, na, f) != (size_t)na || fread(fin.data(), 4, ns, f) != (size_t)ns)
    return 2;
  fclose(f);
  int ok = 0, bad = 0;
  auto run = [&](std::vector<int32_t> &a, std::vector<int32_t> &l, std::vector<int32_t> &n, std::vector<float> &ww,
                 std::vector<float> &ff, int s, int t, int p) {
    tc_supervision *h = nullptr;
    const int rc = tc_supervision_create(&h, weight, s, t, p, ns, a.data(), l.data(), ww.data(), n.data(), ff.data());
    if (rc == 0) { ++ok; tc_supervision_free(h); } else ++bad;
  };
  run(ab, il, nx, w, fin, S, T, P);
  srand(3);
  for (int i = 0; i < 4000; ++i) {
    std::vector<int32_t> a = ab, l = il, n = nx;
    std::vector<float> ww = w, ff = fin;
    int s = S, t = T, p = P;
    const int what = rand() % 8, k = 1 + rand() % 3;
    for (int j = 0; j < k; ++j) switch (what) {
      case 0: { const size_t ix = rand() % a.size(); a[ix] += (rand() % 7) - 3; if (ix + 1 == a.size() && a[ix] > na) a[ix] = na; } break;  // (the arrays hold arc_begin[num_states] arcs: the caller's contract)
      case 1: l[rand() % l.size()] = (rand() % (2 * P + 4)) - 2; break;
      case 2: n[rand() % n.size()] = (rand() % (ns + 6)) - 3; break;
      case 3: ww[rand() % ww.size()] = (rand() % 3 == 0) ? __builtin_inff() : (float)(rand() % 100) - 50.f; break;
      case 4: ff[rand() % ff.size()] = (rand() % 2) ? 0.f : __builtin_inff(); break;
      case 5: s = S + (rand() % 5) - 2; break;
      case 6: t = T + (rand() % 5) - 2; break;
      case 7: { const int x = rand() % n.size(), y = rand() % n.size(); std::swap(n[x], n[y]); } break;
    }
    run(a, l, n, ww, ff, s, t, p);
  }
  printf("create: ok %d refused %d\n", ok, bad);
  // tc_supervision_append: the valid acceptor twice (its own frame count per piece), corrupted the same way -- chains
  // with dead ends, cycles, arcs out of range, non-final last states; output arrays of exactly the documented capacity
  ok = bad = 0;
  const int frames = S * T;
  for (int i = 0; i < 4000; ++i) {
    std::vector<int32_t> a = ab, l = il, n = nx;
    std::vector<float> ww = w, ff = fin;
    const int what = rand() % 5, k = 1 + rand() % 3;
    if (i > 0)
      for (int j = 0; j < k; ++j) switch (what) {
        case 0: n[rand() % n.size()] = (rand() % (ns + 6)) - 3; break;
        case 1: ff[rand() % ff.size()] = (rand() % 2) ? 0.f : __builtin_inff(); break;
        case 2: { const int x = rand() % n.size(), y = rand() % n.size(); std::swap(n[x], n[y]); } break;
        case 3: { const size_t ix = rand() % a.size(); a[ix] += (rand() % 7) - 3; if (ix + 1 == a.size() && a[ix] > na) a[ix] = na; } break;
        case 4: n[rand() % n.size()] = rand() % ns; break;  // stays in range: dead ends and short cuts
      }
    const int32_t pns[2] = {ns, ns}, pnf[2] = {frames + (what == 3 ? 0 : 0), frames};
    std::vector<int32_t> A, L, N;
    std::vector<float> W, F;
    for (int piece = 0; piece < 2; ++piece) {
      const std::vector<int32_t> &pa = piece ? ab : a, &pl = piece ? il : l, &pn = piece ? nx : n;
      const std::vector<float> &pw = piece ? w : ww, &pf = piece ? fin : ff;
      A.insert(A.end(), pa.begin(), pa.end());
      L.insert(L.end(), pl.begin(), pl.end());
      N.insert(N.end(), pn.begin(), pn.end());
      W.insert(W.end(), pw.begin(), pw.end());
      F.insert(F.end(), pf.begin(), pf.end());
    }
    const int64_t cap_states = 2 * (int64_t)ns, cap_arcs = 2 * (int64_t)na + (int64_t)ns * na;
    std::vector<int32_t> ob(cap_states + 1), ol(cap_arcs), on(cap_arcs);
    std::vector<float> ow(cap_arcs), of(cap_states);
    int32_t out_ns = 0;
    int64_t out_na = 0;
    const int rc = tc_supervision_append(2, pns, pnf, A.data(), L.data(), W.data(), N.data(), F.data(), cap_states, cap_arcs,
                                         &out_ns, &out_na, ob.data(), ol.data(), ow.data(), on.data(), of.data());
    if (rc == 0) ++ok; else ++bad;
  }
  printf("append: ok %d refused %d\n", ok, bad);
  return 0;
}
