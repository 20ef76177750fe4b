// tc_self_test: the library's own smoke-and-property check, callable from C without Python.
//
// What it replaces: my_lib_test_chain (src/my_lib.h:45, src/my_lib_chain.cpp:138-213), which builds a random
// supervision with Kaldi's test helpers and runs ChainDenominatorTest / ChainTrainingTest.  The properties those
// tests assert (src/chain-supervision-test.hpp:239-341, 388-463) are restated here on a small synthetic problem:
//   1. weight = supervision.weight * num_sequences * frames_per_sequence;
//   2. objf <= 0 when the numerator is a weighted subset of the denominator's paths (the reference arranges that
//      with AddWeightToSupervisionFst, my_lib_chain.cpp:200-204);
//   3. every derivative row sums to ~0 (both posteriors sum to one per frame) when l2 = 0;
//   4. the derivative predicts the change of objf under a small perturbation of the nnet output (finite differences);
//   5. the denominator's own checks ([K] BetaGeneralFrameDebug) passed: objf is not the failure value -10 * weight.
#include <cmath>
#include <cstdint>
#include <vector>

#include "chain_internal.h"

namespace {

struct Lcg {
  uint64_t s;
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
  float uniform() { return (float)(next() & 0xffffff) / 16777216.0f; }
  int below(int n) { return (int)(next() % (uint32_t)n); }
};

}  // namespace

extern "C" int tc_self_test(int device, void *stream_v, float *report6) {
  using namespace tc;
  hipStream_t stream = (hipStream_t)stream_v;
  int prev = -1;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return TC_ERR_HIP;
  Lcg rng{12345};
  // ---- a small chain-structured denominator graph: H states, each with a self-loop (its own pdf) and two arcs to
  // other states carrying the destination's forward pdf
  const int H = 24, P = 2 * H, S = 3, T = 9;
  std::vector<int32_t> src, dst, lab;
  std::vector<float> wt;
  for (int h = 0; h < H; ++h) {
    float pr[3] = {0.1f + rng.uniform(), 0.1f + rng.uniform(), 0.1f + rng.uniform()};
    const float z = pr[0] + pr[1] + pr[2];
    const int d1 = rng.below(H), d2 = rng.below(H);
    const int ds[3] = {h, d1 == h ? (h + 1) % H : d1, d2 == h ? (h + 2) % H : d2};
    for (int k = 0; k < 3; ++k) {
      src.push_back(h);
      dst.push_back(ds[k]);
      lab.push_back(k == 0 ? H + h + 1 : ds[k] + 1);  // ilabel = pdf + 1: self-loop pdf H + h, forward pdf of the destination
      wt.push_back(-std::log(pr[k] / z));
    }
  }
  std::vector<float> fin(H, 0.f);
  tc_den_graph *g = nullptr;
  int rc = tc_den_graph_create(&g, H, (int64_t)src.size(), src.data(), dst.data(), lab.data(), wt.data(), fin.data(), 0, P);
  if (rc != TC_OK) return rc;
  std::vector<float> pi(H);
  rc = tc_den_graph_initial_probs(g, pi.data());
  // ---- supervision: one path of the graph per sequence, weighted like the graph (first arc also by pi(start)), so that
  // the numerator is a weighted subset of the denominator's paths; the merged acceptor of S such sequences is a chain
  const float sup_w = 0.5f;
  std::vector<int32_t> ab(S * T + 2, 0), il, nx;
  std::vector<float> aw, fw(S * T + 1, INFINITY);
  for (int s = 0; s < S && rc == TC_OK; ++s) {
    int h = rng.below(H);
    while (pi[h] <= 0.f) h = (h + 1) % H;
    for (int t = 0; t < T; ++t) {
      const int a = 3 * h + rng.below(3);
      il.push_back(lab[a]);
      aw.push_back(wt[a] + (t == 0 ? -std::log(pi[h]) : 0.f));
      nx.push_back(s * T + t + 1);
      ab[s * T + t + 1] = (int32_t)il.size();
      h = dst[a];
    }
  }
  ab[S * T + 1] = (int32_t)il.size();
  fw[S * T] = 0.f;
  tc_supervision *sup = nullptr;
  if (rc == TC_OK) rc = tc_supervision_create(&sup, sup_w, S, T, P, S * T + 1, ab.data(), il.data(), aw.data(), nx.data(), fw.data());
  // ---- two evaluations: y and y + eps * delta
  const int64_t rows = (int64_t)S * T, n = rows * P;
  std::vector<float> y(n), y2(n), delta(n);
  for (int64_t i = 0; i < n; ++i) {
    y[i] = 2.f * rng.uniform() - 1.f;
    delta[i] = 2.f * rng.uniform() - 1.f;
    y2[i] = y[i] + 1e-2f * delta[i];
  }
  float *d_y = nullptr, *d_deriv = nullptr, *d_res = nullptr;
  void *d_ws = nullptr;
  const int64_t ws_bytes = rc == TC_OK ? tc_chain_workspace_bytes(g, S, T) : 0;
  hipError_t e = hipSuccess;
  if (rc == TC_OK && ws_bytes < 0) rc = (int)ws_bytes;
  if (rc == TC_OK) {
    e = hipMalloc((void **)&d_y, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_deriv, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&d_res, 16);
    if (e == hipSuccess) e = hipMalloc(&d_ws, (size_t)ws_bytes);
  }
  float res[2][3] = {{0, 0, 0}, {0, 0, 0}};
  std::vector<float> deriv(n);
  for (int pass = 0; pass < 2 && rc == TC_OK && e == hipSuccess; ++pass) {
    e = hipMemcpyAsync(d_y, pass == 0 ? y.data() : y2.data(), n * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) break;
    rc = tc_chain_objf_and_deriv(g, sup, d_y, rows, P, P, d_res, pass == 0 ? d_deriv : nullptr, P, nullptr, 0, 0.0f, 0.1f, 0.0f,
                                 d_ws, ws_bytes, device, stream);
    if (rc != TC_OK) break;
    e = hipMemcpyAsync(res[pass], d_res, 12, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && pass == 0) e = hipMemcpyAsync(deriv.data(), d_deriv, n * 4, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  for (void *ptr : {(void *)d_y, (void *)d_deriv, (void *)d_res, d_ws})
    if (ptr) (void)hipFree(ptr);
  if (sup) tc_supervision_free(sup);
  tc_den_graph_free(g);
  (void)hipSetDevice(prev);
  if (rc != TC_OK) return rc;
  if (e != hipSuccess) {
    g_last_hip_error = (int)e;
    return TC_ERR_HIP;
  }
  // ---- the properties
  const float objf = res[0][0], weight = res[0][2];
  double worst_row = 0.0, predicted = 0.0;
  for (int64_t r = 0; r < rows; ++r) {
    double sum = 0.0;
    for (int c = 0; c < P; ++c) {
      sum += deriv[r * P + c];
      predicted += (double)deriv[r * P + c] * 1e-2 * delta[r * P + c];
    }
    worst_row = std::fmax(worst_row, std::fabs(sum));
  }
  const double observed = (double)res[1][0] - (double)res[0][0];
  if (report6) {
    report6[0] = objf;
    report6[1] = weight;
    report6[2] = (float)worst_row;
    report6[3] = (float)predicted;
    report6[4] = (float)observed;
    report6[5] = res[0][1];
  }
  if (weight != sup_w * S * T) return 1;
  if (!(objf <= 0.f) || !std::isfinite(objf)) return 2;
  if (objf == -10.f * weight) return 5;
  if (worst_row > 1e-4) return 3;
  if (std::fabs(observed - predicted) > 0.1 * std::fabs(predicted) + 1e-4) return 4;
  return TC_OK;
}
