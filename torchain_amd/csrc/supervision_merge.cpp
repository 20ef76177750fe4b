// Host side of the data format in front of the hot path: merging the supervisions of a minibatch's examples into the
// one acceptor tc_supervision_create takes.
//
// What it replaces: kaldi::nnet3::MergeChainExamples -> [K] chain::AppendSupervision, which the reference reaches
// natively at src/my_lib_example_rand.cpp:160 (and src/my_lib_example.cpp through the same Kaldi call).  For examples
// of equal weight, frames-per-sequence and label-dim [K] concatenates the FSTs (fst::Concat), removes the epsilons --
// every final state f of piece k-1 (final weight w_f) receives copies of piece k's start arcs with weight
// w_f + arc weight and stops being final; piece k's start state disappears -- and renumbers the states breadth-first,
// i.e. in time order.  torchain_amd/egs.py holds the same algorithm in numpy (the check of this one in the tests).
#include <cmath>
#include <cstring>
#include <vector>

#include "chain_internal.h"

extern "C" int tc_supervision_append(int32_t num_pieces, const int32_t *piece_num_states, const int32_t *piece_num_frames,
                                     const int32_t *arc_begin, const int32_t *arc_ilabel, const float *arc_weight,
                                     const int32_t *arc_nextstate, const float *final_weight, int64_t cap_states,
                                     int64_t cap_arcs, int32_t *out_num_states, int64_t *out_num_arcs,
                                     int32_t *out_arc_begin, int32_t *out_ilabel, float *out_weight,
                                     int32_t *out_nextstate, float *out_final) {
  if (num_pieces <= 0 || !piece_num_states || !piece_num_frames || !arc_begin || !final_weight || !out_num_states ||
      !out_num_arcs || !out_arc_begin || !out_final)
    return TC_ERR_INVALID_ARGUMENT;
  const int K = num_pieces;
  // piece k: states [soff[k], soff[k+1]) of the concatenated state arrays, its CSR offsets at arc_begin + soff[k] + k
  // (num_states + 1 entries, local), its arcs at aoff[k] + local arc index
  std::vector<int64_t> soff(K + 1, 0), aoff(K + 1, 0);
  for (int k = 0; k < K; ++k) {
    if (piece_num_states[k] <= 0 || piece_num_frames[k] <= 0) return TC_ERR_BAD_FST;
    soff[k + 1] = soff[k] + piece_num_states[k];
    const int32_t *ab = arc_begin + soff[k] + k;
    if (ab[0] != 0) return TC_ERR_BAD_FST;
    for (int i = 0; i < piece_num_states[k]; ++i)
      if (ab[i + 1] < ab[i]) return TC_ERR_BAD_FST;
    aoff[k + 1] = aoff[k] + ab[piece_num_states[k]];
  }
  const int64_t n_raw = soff[K];
  if ((aoff[K] > 0 && (!arc_ilabel || !arc_weight || !arc_nextstate)) || !out_ilabel || !out_weight || !out_nextstate)
    return aoff[K] > 0 ? TC_ERR_INVALID_ARGUMENT : TC_ERR_BAD_FST;
  // times: every arc goes from time tau to tau + 1; states need not be numbered in time order
  std::vector<int32_t> time(n_raw, -1);
  std::vector<int64_t> toff(K + 1, 0);
  std::vector<int32_t> queue;
  for (int k = 0; k < K; ++k) {
    const int n = piece_num_states[k];
    const int32_t *ab = arc_begin + soff[k] + k;
    const int32_t *nx = arc_nextstate + aoff[k];
    int32_t *tm = time.data() + soff[k];
    queue.assign(1, 0);
    tm[0] = 0;
    for (size_t qi = 0; qi < queue.size(); ++qi) {
      const int s = queue[qi];
      for (int a = ab[s]; a < ab[s + 1]; ++a) {
        const int d = nx[a];
        if (d < 0 || d >= n) return TC_ERR_BAD_FST;
        if (d == 0 && k > 0) return TC_ERR_BAD_FST;  // the start state of a later piece has incoming arcs
        if (tm[d] < 0) {
          // (a dead-end chain longer than the piece -- no final state on it -- would index past the per-time counts below)
          if (tm[s] + 1 > piece_num_frames[k]) return TC_ERR_BAD_FST;
          tm[d] = tm[s] + 1;
          queue.push_back(d);
        } else if (tm[d] != tm[s] + 1) {
          return TC_ERR_BAD_FST;  // paths of unequal lengths
        }
      }
    }
    if ((int)queue.size() != n) return TC_ERR_BAD_FST;  // not connected
    for (int s = 0; s < n; ++s) {
      const bool fin = !std::isinf(final_weight[soff[k] + s]);
      if (fin && (tm[s] != piece_num_frames[k] || ab[s + 1] != ab[s])) return TC_ERR_BAD_FST;  // finals: last frame, no arcs
    }
    toff[k + 1] = toff[k] + piece_num_frames[k];
  }
  // new ids: the surviving states (all but the start states of pieces 1 .. K-1) in (global time, piece, local id) order
  // -- a counting sort over the global times
  const int64_t n_total = n_raw - (K - 1);
  if (n_total > cap_states) return TC_ERR_WORKSPACE;
  std::vector<int32_t> per_time(toff[K] + 2, 0);
  for (int k = 0; k < K; ++k)
    for (int s = (k > 0 ? 1 : 0); s < piece_num_states[k]; ++s) per_time[toff[k] + time[soff[k] + s] + 1]++;
  for (size_t i = 1; i < per_time.size(); ++i) per_time[i] += per_time[i - 1];
  std::vector<int32_t> newid(n_raw, -1);
  for (int k = 0; k < K; ++k)
    for (int s = (k > 0 ? 1 : 0); s < piece_num_states[k]; ++s) newid[soff[k] + s] = per_time[toff[k] + time[soff[k] + s]]++;
  // out-degrees of the merged states
  for (int64_t i = 0; i <= n_total; ++i) out_arc_begin[i] = 0;
  for (int k = 0; k < K; ++k) {
    const int32_t *ab = arc_begin + soff[k] + k;
    const int start_arcs_next = k + 1 < K ? (arc_begin + soff[k + 1] + k + 1)[1] : 0;
    for (int s = (k > 0 ? 1 : 0); s < piece_num_states[k]; ++s) {
      int deg = ab[s + 1] - ab[s];
      if (k + 1 < K && !std::isinf(final_weight[soff[k] + s])) deg += start_arcs_next;
      out_arc_begin[newid[soff[k] + s] + 1] = deg;
    }
  }
  int64_t total_arcs = 0;
  for (int64_t i = 0; i < n_total; ++i) {
    total_arcs += out_arc_begin[i + 1];
    if (total_arcs > (int64_t)INT32_MAX) return TC_ERR_UNSUPPORTED;
    out_arc_begin[i + 1] = (int32_t)total_arcs;
  }
  if (total_arcs > cap_arcs) return TC_ERR_WORKSPACE;
  for (int k = 0; k < K; ++k) {
    const int32_t *ab = arc_begin + soff[k] + k;
    for (int s = (k > 0 ? 1 : 0); s < piece_num_states[k]; ++s) {
      const int64_t g = newid[soff[k] + s];
      int64_t o = out_arc_begin[g];
      for (int a = ab[s]; a < ab[s + 1]; ++a, ++o) {
        out_ilabel[o] = arc_ilabel[aoff[k] + a];
        out_weight[o] = arc_weight[aoff[k] + a];
        out_nextstate[o] = newid[soff[k] + arc_nextstate[aoff[k] + a]];
      }
      const float fw = final_weight[soff[k] + s];
      out_final[g] = INFINITY;
      if (!std::isinf(fw)) {
        if (k + 1 < K) {
          const int32_t *nb = arc_begin + soff[k + 1] + k + 1;
          for (int a = nb[0]; a < nb[1]; ++a, ++o) {
            out_ilabel[o] = arc_ilabel[aoff[k + 1] + a];
            out_weight[o] = fw + arc_weight[aoff[k + 1] + a];
            out_nextstate[o] = newid[soff[k + 1] + arc_nextstate[aoff[k + 1] + a]];
          }
        } else {
          out_final[g] = fw;
        }
      }
    }
  }
  *out_num_states = (int32_t)n_total;
  *out_num_arcs = total_arcs;
  return TC_OK;
}
