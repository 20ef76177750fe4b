/*
 * chain_oracle.c -- CPU restatement of the LF-MMI "chain" objective, TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP path in torchain_amd/csrc.  It is never linked,
 * imported or executed by the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.
 *
 * What it restates.  torchain (the reference, /root/reference) contains none of this arithmetic:
 * src/my_lib_chain.cpp:129-131 calls kaldi::chain::ComputeChainObjfAndDeriv, and
 * src/my_lib_example.cpp:131-132 constructs kaldi::chain::DenominatorGraph.  Kaldi is an
 * un-vendored third-party dependency of the reference (Makefile:16-18,29,60-61, build.py:8-16;
 * pinned only in prose to "kaldi 22fbdd", README.md:3, example/chime5/path.sh:1) and its source is
 * absent from /root/reference.  The functions below therefore restate Kaldi's published CPU
 * algorithm (the non-CUDA branches of src/chain/chain-den-graph.cc, chain-denominator.cc,
 * chain-numerator.cc, chain-training.cc at that vintage): same loop nests, float storage with
 * double accumulators in the denominator, double log-domain in the numerator.  Each function names
 * the Kaldi routine it follows and the reference call site that reaches it.
 *
 * PARITY UNPINNED: the reference holds no golden vector or known-answer test for this path
 * (test/test.py asserts no values and needs private data; README.md:12-32 is not reproducible),
 * and Kaldi cannot be built here, so this restatement cannot be diffed against Kaldi output.
 * It is pinned instead by (i) an independent float64 formulation (oracle/independent_f64.py,
 * un-scaled log-domain recursion + autograd) that must agree with it, (ii) the portable properties
 * the reference's own native test asserts (src/chain-supervision-test.hpp:92-152,239-341,388-463),
 * restated in tests/test_oracle_properties.py, and (iii) weight = w*S*T (README.md:12-32).
 *
 * One documented deviation: exp(y) is clamped to y in [-30, 30] (later Kaldi's ApplyExpLimited);
 * identical to the plain ApplyExp of the 22fbdd vintage for |y| < 30.  Disable with
 * oracle_set_exp_clamp(0).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  float transition_prob;
  int32_t pdf_id;
  int32_t hmm_state;
} oracle_transition; /* [K] DenominatorGraphTransition, chain-den-graph.h */

typedef struct {
  int32_t num_states;
  int32_t num_pdfs;
  int64_t num_arcs;
  int32_t *fwd_first, *fwd_second; /* [K] forward_transitions_: out-arcs of each state  */
  int32_t *bwd_first, *bwd_second; /* [K] backward_transitions_: in-arcs of each state  */
  oracle_transition *transitions;  /* out-lists (state-major) followed by in-lists      */
  float *initial_probs;
} oracle_den_graph;

static int g_exp_clamp = 1;
void oracle_set_exp_clamp(int on) { g_exp_clamp = on; }

static inline float exp_limited(float x) {
  if (g_exp_clamp) {
    if (x < -30.0f) x = -30.0f;
    if (x > 30.0f) x = 30.0f;
  }
  return expf(x);
}

/* ------------------------------------------------------------------------------------------
 * [K] DenominatorGraph::DenominatorGraph / SetTransitions / SetInitialProbs (chain-den-graph.cc),
 * reached from the reference at src/my_lib_example.cpp:129-134 (my_lib_denominator_graph_new).
 * The FST is passed as flat arrays, arcs in state-major order exactly as an
 * fst::StdVectorFst ArcIterator would yield them; ilabel = pdf_id + 1; weights are -log probs;
 * final[s] = +inf for a non-final state (TropicalWeight::Zero()).
 * ------------------------------------------------------------------------------------------ */
oracle_den_graph *oracle_den_graph_new(int32_t num_states, int64_t num_arcs, const int32_t *arc_src,
                                       const int32_t *arc_dst, const int32_t *arc_ilabel,
                                       const float *arc_weight, const float *final_weight,
                                       int32_t start, int32_t num_pdfs) {
  if (num_states <= 0 || num_arcs < 0 || start < 0 || start >= num_states) return NULL;
  for (int64_t a = 0; a < num_arcs; a++) {
    if (arc_src[a] < 0 || arc_src[a] >= num_states || arc_dst[a] < 0 || arc_dst[a] >= num_states)
      return NULL;
    if (arc_ilabel[a] - 1 < 0 || arc_ilabel[a] - 1 >= num_pdfs) return NULL; /* KALDI_ASSERT */
    if (a > 0 && arc_src[a] < arc_src[a - 1]) return NULL;                   /* state-major */
  }
  oracle_den_graph *g = (oracle_den_graph *)calloc(1, sizeof(*g));
  g->num_states = num_states;
  g->num_pdfs = num_pdfs;
  g->num_arcs = num_arcs;
  g->fwd_first = (int32_t *)calloc(num_states, sizeof(int32_t));
  g->fwd_second = (int32_t *)calloc(num_states, sizeof(int32_t));
  g->bwd_first = (int32_t *)calloc(num_states, sizeof(int32_t));
  g->bwd_second = (int32_t *)calloc(num_states, sizeof(int32_t));
  g->transitions = (oracle_transition *)calloc((size_t)(2 * num_arcs + 1), sizeof(oracle_transition));
  g->initial_probs = (float *)calloc(num_states, sizeof(float));

  /* SetTransitions: out-lists in state order. */
  {
    int64_t a = 0;
    for (int32_t s = 0; s < num_states; s++) {
      g->fwd_first[s] = (int32_t)a;
      while (a < num_arcs && arc_src[a] == s) {
        oracle_transition *tr = &g->transitions[a];
        tr->transition_prob = (float)exp(-(double)arc_weight[a]); /* exp(-arc.weight.Value()) */
        tr->pdf_id = arc_ilabel[a] - 1;
        tr->hmm_state = arc_dst[a];
        a++;
      }
      g->fwd_second[s] = (int32_t)a;
    }
  }
  /* in-lists: transitions_in[arc.nextstate].push_back(...) in (state, arc) order -> stable
     counting sort by destination. */
  {
    int64_t *count = (int64_t *)calloc((size_t)num_states + 1, sizeof(int64_t));
    for (int64_t a = 0; a < num_arcs; a++) count[arc_dst[a] + 1]++;
    for (int32_t s = 0; s < num_states; s++) count[s + 1] += count[s];
    for (int32_t s = 0; s < num_states; s++) {
      g->bwd_first[s] = (int32_t)(num_arcs + count[s]);
      g->bwd_second[s] = (int32_t)(num_arcs + count[s + 1]);
    }
    int64_t *fill = (int64_t *)calloc((size_t)num_states, sizeof(int64_t));
    for (int64_t a = 0; a < num_arcs; a++) {
      int32_t d = arc_dst[a];
      oracle_transition *tr = &g->transitions[num_arcs + count[d] + fill[d]++];
      tr->transition_prob = (float)exp(-(double)arc_weight[a]);
      tr->pdf_id = arc_ilabel[a] - 1;
      tr->hmm_state = arc_src[a];
    }
    free(count);
    free(fill);
  }
  /* SetInitialProbs: 100 iterations of normalised HMM propagation from the start state, averaged. */
  {
    const int num_iters = 100;
    double *normalizing_factor = (double *)calloc(num_states, sizeof(double));
    double *cur = (double *)calloc(num_states, sizeof(double));
    double *next = (double *)calloc(num_states, sizeof(double));
    double *avg = (double *)calloc(num_states, sizeof(double));
    for (int32_t s = 0; s < num_states; s++) {
      double tot_prob = exp(-(double)final_weight[s]); /* exp(-inf) = 0 for non-final */
      for (int32_t a = g->fwd_first[s]; a < g->fwd_second[s]; a++)
        tot_prob += exp(-(double)arc_weight[a]);
      normalizing_factor[s] = 1.0 / tot_prob;
    }
    cur[start] = 1.0;
    for (int iter = 0; iter < num_iters; iter++) {
      for (int32_t s = 0; s < num_states; s++) avg[s] += (1.0 / num_iters) * cur[s];
      for (int32_t s = 0; s < num_states; s++) {
        double prob = cur[s] * normalizing_factor[s];
        for (int32_t a = g->fwd_first[s]; a < g->fwd_second[s]; a++)
          next[arc_dst[a]] += prob * exp(-(double)arc_weight[a]);
      }
      double sum = 0.0;
      for (int32_t s = 0; s < num_states; s++) {
        cur[s] = next[s];
        next[s] = 0.0;
        sum += cur[s];
      }
      for (int32_t s = 0; s < num_states; s++) cur[s] *= 1.0 / sum;
    }
    for (int32_t s = 0; s < num_states; s++) g->initial_probs[s] = (float)avg[s];
    free(normalizing_factor);
    free(cur);
    free(next);
    free(avg);
  }
  return g;
}

void oracle_den_graph_free(oracle_den_graph *g) {
  if (!g) return;
  free(g->fwd_first);
  free(g->fwd_second);
  free(g->bwd_first);
  free(g->bwd_second);
  free(g->transitions);
  free(g->initial_probs);
  free(g);
}

int32_t oracle_den_graph_num_states(const oracle_den_graph *g) { return g->num_states; }
int32_t oracle_den_graph_num_pdfs(const oracle_den_graph *g) { return g->num_pdfs; }
int64_t oracle_den_graph_num_arcs(const oracle_den_graph *g) { return g->num_arcs; }
void oracle_den_graph_initial_probs(const oracle_den_graph *g, float *out) {
  memcpy(out, g->initial_probs, sizeof(float) * (size_t)g->num_states);
}
/* copies the 2*A transition records as three parallel arrays + the 4 index arrays */
void oracle_den_graph_transitions(const oracle_den_graph *g, float *prob, int32_t *pdf,
                                  int32_t *state, int32_t *fwd_first, int32_t *fwd_second,
                                  int32_t *bwd_first, int32_t *bwd_second) {
  for (int64_t i = 0; i < 2 * g->num_arcs; i++) {
    prob[i] = g->transitions[i].transition_prob;
    pdf[i] = g->transitions[i].pdf_id;
    state[i] = g->transitions[i].hmm_state;
  }
  memcpy(fwd_first, g->fwd_first, sizeof(int32_t) * (size_t)g->num_states);
  memcpy(fwd_second, g->fwd_second, sizeof(int32_t) * (size_t)g->num_states);
  memcpy(bwd_first, g->bwd_first, sizeof(int32_t) * (size_t)g->num_states);
  memcpy(bwd_second, g->bwd_second, sizeof(int32_t) * (size_t)g->num_states);
}

/* ------------------------------------------------------------------------------------------
 * [K] DenominatorComputation (chain-denominator.cc, CPU branch); reference call sites:
 * src/chain-supervision-test.hpp:403-414,439-443 (direct) and via ComputeChainObjfAndDeriv at
 * src/my_lib_chain.cpp:129-131.
 *
 * Layouts as in Kaldi: nnet_output y is (T*S) x P, row = t*S + s (reference
 * torchain/functions.py:27-30); exp_nnet_output_transposed_ is P x (T*S); alpha_ is
 * (T+1) x (H*S + S) with element [h*S + s] and the alpha-sums in the last S entries; beta_ is
 * 2 x (H*S + S).
 *
 * Returns 0 on success.  *logprob = Forward(); if deriv != NULL runs Backward(deriv_weight, deriv),
 * i.e. deriv += deriv_weight * gamma, and sets *ok as BetaGeneralFrameDebug(0) would.
 * alpha_beta_check / gamma_check (nullable) receive the two debug sums.
 * ------------------------------------------------------------------------------------------ */
int oracle_den_forward_backward(const oracle_den_graph *g, float leaky_hmm_coefficient,
                                int32_t num_sequences, const float *nnet_output, int64_t num_rows,
                                int32_t num_cols, int64_t row_stride, float deriv_weight,
                                float *deriv, int64_t deriv_stride, float *logprob, int32_t *ok_out,
                                float *alpha_beta_check, float *gamma_check) {
  const int32_t S = num_sequences, H = g->num_states, P = g->num_pdfs;
  if (!(leaky_hmm_coefficient > 0.0f && leaky_hmm_coefficient < 1.0f)) return -1; /* KALDI_ASSERT */
  if (S <= 0 || num_rows % S != 0 || num_cols != P) return -2;
  const int32_t T = (int32_t)(num_rows / S);
  const int64_t TS = (int64_t)T * S;
  const int64_t arow = (int64_t)H * S + S;
  const oracle_transition *transitions = g->transitions;

  float *expT = (float *)malloc(sizeof(float) * (size_t)P * (size_t)TS);
  float *alpha = (float *)malloc(sizeof(float) * (size_t)(T + 1) * (size_t)arow);
  float *beta = (float *)malloc(sizeof(float) * 2 * (size_t)arow);
  float *tot_prob = (float *)malloc(sizeof(float) * (size_t)S);
  float *gamma = NULL;
  if (!expT || !alpha || !beta || !tot_prob) return -3;

  /* ctor: exp_nnet_output_transposed_(nnet_output, kTrans); ApplyExp(); zero the alpha/beta sums */
  for (int64_t r = 0; r < TS; r++)
    for (int32_t p = 0; p < P; p++) expT[(int64_t)p * TS + r] = exp_limited(nnet_output[r * row_stride + p]);
  for (int32_t t = 0; t <= T; t++)
    for (int32_t s = 0; s < S; s++) alpha[(int64_t)t * arow + (int64_t)H * S + s] = 0.0f;
  for (int i = 0; i < 2; i++)
    for (int32_t s = 0; s < S; s++) beta[(int64_t)i * arow + (int64_t)H * S + s] = 0.0f;

  /* ---- Forward() ---- */
  for (int32_t t = 0; t <= T; t++) {
    float *this_alpha = alpha + (int64_t)t * arow;
    if (t == 0) {
      /* AlphaFirstFrame: alpha_0(h, s) = initial_probs(h) */
      for (int32_t h = 0; h < H; h++)
        for (int32_t s = 0; s < S; s++) this_alpha[(int64_t)h * S + s] = g->initial_probs[h];
    } else {
      /* AlphaGeneralFrame(t) */
      const float *prev_alpha_dash = alpha + (int64_t)(t - 1) * arow;
      const float *prob_data = expT + (int64_t)(t - 1) * S; /* probs for frame t-1, stride TS */
      for (int32_t h = 0; h < H; h++) {
        for (int32_t s = 0; s < S; s++) {
          double this_tot_alpha = 0.0;
          for (int32_t i = g->bwd_first[h]; i < g->bwd_second[h]; i++) {
            float transition_prob = transitions[i].transition_prob;
            int32_t pdf_id = transitions[i].pdf_id, prev_hmm_state = transitions[i].hmm_state;
            float prob = prob_data[(int64_t)pdf_id * TS + s],
                  this_prev_alpha = prev_alpha_dash[(int64_t)prev_hmm_state * S + s];
            this_tot_alpha += this_prev_alpha * transition_prob * prob;
          }
          float arbitrary_scale = 1.0f / prev_alpha_dash[(int64_t)H * S + s];
          this_alpha[(int64_t)h * S + s] = (float)(this_tot_alpha * arbitrary_scale);
        }
      }
    }
    /* AlphaDash(t): alpha_sum = row-sum over states; alpha += leaky * pi * alpha_sum */
    float *alpha_sum = this_alpha + (int64_t)H * S;
    for (int32_t s = 0; s < S; s++) alpha_sum[s] = 0.0f;
    for (int32_t h = 0; h < H; h++)
      for (int32_t s = 0; s < S; s++) alpha_sum[s] += this_alpha[(int64_t)h * S + s];
    for (int32_t h = 0; h < H; h++) {
      float c = leaky_hmm_coefficient * g->initial_probs[h];
      for (int32_t s = 0; s < S; s++) this_alpha[(int64_t)h * S + s] += c * alpha_sum[s];
    }
  }
  /* ComputeTotLogLike */
  double tot_log_prob = 0.0, log_inv_arbitrary_scales_product = 0.0;
  {
    const float *last_alpha_dash = alpha + (int64_t)T * arow;
    for (int32_t s = 0; s < S; s++) tot_prob[s] = 0.0f;
    for (int32_t h = 0; h < H; h++)
      for (int32_t s = 0; s < S; s++) tot_prob[s] += last_alpha_dash[(int64_t)h * S + s];
    for (int32_t s = 0; s < S; s++) tot_log_prob += (double)logf(tot_prob[s]);
    for (int32_t t = 0; t < T; t++)
      for (int32_t s = 0; s < S; s++)
        log_inv_arbitrary_scales_product += (double)logf(alpha[(int64_t)t * arow + (int64_t)H * S + s]);
  }
  *logprob = (float)((float)tot_log_prob + (float)log_inv_arbitrary_scales_product);

  int ok = 1;
  if (deriv) {
    /* ---- Backward(deriv_weight, deriv) ---- */
    gamma = (float *)calloc((size_t)P * (size_t)S, sizeof(float)); /* one frame of the P x (8*S) staging */
    /* BetaDashLastFrame + Beta(T) */
    for (int32_t t = T; t >= 0; t--) {
      float *this_beta_dash = beta + (int64_t)(t % 2) * arow;
      if (t == T) {
        for (int32_t h = 0; h < H; h++)
          for (int32_t s = 0; s < S; s++) this_beta_dash[(int64_t)h * S + s] = 1.0f / tot_prob[s];
      } else {
        /* BetaDashGeneralFrame(t) */
        const float *this_alpha_dash = alpha + (int64_t)t * arow;
        const float *next_beta = beta + (int64_t)((t + 1) % 2) * arow;
        const float *prob_data = expT + (int64_t)t * S;
        memset(gamma, 0, sizeof(float) * (size_t)P * (size_t)S);
        for (int32_t h = 0; h < H; h++) {
          for (int32_t s = 0; s < S; s++) {
            float this_alpha_dash_prob = this_alpha_dash[(int64_t)h * S + s],
                  inv_arbitrary_scale = this_alpha_dash[(int64_t)H * S + s];
            double tot_variable_factor = 0.0;
            float occupation_factor = this_alpha_dash_prob / inv_arbitrary_scale;
            for (int32_t i = g->fwd_first[h]; i < g->fwd_second[h]; i++) {
              float transition_prob = transitions[i].transition_prob;
              int32_t pdf_id = transitions[i].pdf_id, next_hmm_state = transitions[i].hmm_state;
              float variable_factor = transition_prob * next_beta[(int64_t)next_hmm_state * S + s] *
                                      prob_data[(int64_t)pdf_id * TS + s];
              tot_variable_factor += variable_factor;
              float occupation_prob = variable_factor * occupation_factor;
              gamma[(int64_t)pdf_id * S + s] += occupation_prob;
            }
            this_beta_dash[(int64_t)h * S + s] = (float)(tot_variable_factor / inv_arbitrary_scale);
          }
        }
        if (t == 0) {
          /* BetaGeneralFrameDebug(0) */
          double alpha_beta_product = 0.0, this_log_prob_deriv_sum = 0.0;
          for (int64_t i = 0; i < (int64_t)H * S; i++)
            alpha_beta_product += (double)this_alpha_dash[i] * this_beta_dash[i];
          for (int64_t i = 0; i < (int64_t)P * S; i++) this_log_prob_deriv_sum += gamma[i];
          if (fabs(alpha_beta_product - S) > 2.0) ok = 0;
          if (fabs(this_log_prob_deriv_sum - S) > 2.0) ok = 0;
          if (!(alpha_beta_product - alpha_beta_product == 0.0)) ok = 0;
          if (alpha_beta_check) *alpha_beta_check = (float)alpha_beta_product;
          if (gamma_check) *gamma_check = (float)this_log_prob_deriv_sum;
        }
        /* output_deriv_part.AddMat(deriv_weight, transposed_deriv_part, kTrans) */
        for (int32_t s = 0; s < S; s++) {
          float *drow = deriv + ((int64_t)t * S + s) * deriv_stride;
          for (int32_t p = 0; p < P; p++) drow[p] += deriv_weight * gamma[(int64_t)p * S + s];
        }
      }
      /* Beta(t): beta_dash_sum = leaky * sum_h pi(h) beta_dash(h); beta = beta_dash + sum */
      float *beta_dash_sum = this_beta_dash + (int64_t)H * S;
      for (int32_t s = 0; s < S; s++) beta_dash_sum[s] = 0.0f;
      for (int32_t h = 0; h < H; h++) {
        float c = leaky_hmm_coefficient * g->initial_probs[h];
        for (int32_t s = 0; s < S; s++) beta_dash_sum[s] += c * this_beta_dash[(int64_t)h * S + s];
      }
      for (int32_t h = 0; h < H; h++)
        for (int32_t s = 0; s < S; s++) this_beta_dash[(int64_t)h * S + s] += beta_dash_sum[s];
    }
  }
  if (ok_out) *ok_out = ok;
  free(expT);
  free(alpha);
  free(beta);
  free(tot_prob);
  free(gamma);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * Numerator.  [K] ComputeFstStateTimes (chain-supervision.cc) and NumeratorComputation
 * (chain-numerator.cc); reference call sites src/chain-supervision-test.hpp:99-107,121-122 and via
 * ComputeChainObjfAndDeriv.  The supervision FST ([K] Supervision::fst, fields used by the
 * reference at src/my_lib_example.cpp:82-95) is passed in CSR form: state i has arcs
 * [arc_begin[i], arc_begin[i+1]); ilabel = pdf + 1; weights are tropical (-log); final[i] = +inf
 * when non-final.  For merged egs it is the concatenation of the S per-sequence FSTs.
 * ------------------------------------------------------------------------------------------ */
static const double kMinLogDiffDouble = -36.04365338911715; /* log(DBL_EPSILON) */

static inline double log_add(double x, double y) {
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= kMinLogDiffDouble) return x + log1p(exp(diff));
  return x;
}

/* returns total length (>=0) or -1 if the FST lacks the required properties */
int32_t oracle_fst_state_times(int32_t num_states, const int32_t *arc_begin, const int32_t *arc_ilabel,
                               const int32_t *arc_next, const float *final_weight, int32_t *state_times) {
  int32_t total_length = -1;
  for (int32_t i = 0; i < num_states; i++) state_times[i] = -1;
  state_times[0] = 0;
  for (int32_t state = 0; state < num_states; state++) {
    int32_t next_state_time = state_times[state] + 1;
    if (next_state_time <= 0) return -1;
    for (int32_t a = arc_begin[state]; a < arc_begin[state + 1]; a++) {
      if (arc_ilabel[a] == 0) return -1;
      int32_t *ref = &state_times[arc_next[a]];
      if (*ref == -1)
        *ref = next_state_time;
      else if (*ref != next_state_time)
        return -1;
    }
    if (!isinf(final_weight[state])) {
      if (total_length == -1)
        total_length = next_state_time - 1;
      else if (total_length != next_state_time - 1)
        return -1;
    }
  }
  return total_length;
}

/*
 * NumeratorComputation::Forward + Backward.  *logprob_weighted = weight * log Z.  If deriv != NULL,
 * deriv[row, pdf] += weight * occupation (AddElements).  Returns 0 on success.
 */
int oracle_num_forward_backward(float sup_weight, int32_t num_sequences, int32_t frames_per_sequence,
                                int32_t label_dim, int32_t num_states, const int32_t *arc_begin,
                                const int32_t *arc_ilabel, const float *arc_weight,
                                const int32_t *arc_next, const float *final_weight,
                                const float *nnet_output, int64_t num_rows, int32_t num_cols,
                                int64_t row_stride, float *deriv, int64_t deriv_stride,
                                float *logprob_weighted) {
  const int32_t S = num_sequences, T = frames_per_sequence;
  if ((int64_t)S * T != num_rows || label_dim != num_cols) return -2;
  int32_t *state_times = (int32_t *)malloc(sizeof(int32_t) * (size_t)num_states);
  int32_t total = oracle_fst_state_times(num_states, arc_begin, arc_ilabel, arc_next, final_weight, state_times);
  if (total != S * T) {
    free(state_times);
    return -4;
  }
  const int32_t num_arcs = arc_begin[num_states];

  /* ComputeLookupIndexes: unique (t, pdf) per frame -> index into the gather list */
  int32_t *fst_output_indexes = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_arcs + 1));
  int32_t *idx_row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_arcs + 1));
  int32_t *idx_pdf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_arcs + 1));
  int32_t *map_this_frame = (int32_t *)malloc(sizeof(int32_t) * (size_t)label_dim);
  int32_t *touched = (int32_t *)malloc(sizeof(int32_t) * (size_t)(num_arcs + 1));
  int32_t n_touched = 0, n_index = 0, cur_time = 0, n_out = 0;
  for (int32_t p = 0; p < label_dim; p++) map_this_frame[p] = -1;
  for (int32_t state = 0; state < num_states; state++) {
    int32_t t = state_times[state];
    if (t != cur_time) {
      for (int32_t i = 0; i < n_touched; i++) map_this_frame[touched[i]] = -1;
      n_touched = 0;
      cur_time = t;
    }
    for (int32_t a = arc_begin[state]; a < arc_begin[state + 1]; a++) {
      int32_t pdf_id = arc_ilabel[a] - 1;
      if (pdf_id < 0 || pdf_id >= label_dim) return -5;
      if (map_this_frame[pdf_id] < 0) {
        map_this_frame[pdf_id] = n_index;
        touched[n_touched++] = pdf_id;
        /* ComputeRowIndex(t, T, S) = t / T + S * (t % T) */
        idx_row[n_index] = t / T + S * (t % T);
        idx_pdf[n_index] = pdf_id;
        fst_output_indexes[n_out++] = n_index++;
      } else {
        fst_output_indexes[n_out++] = map_this_frame[pdf_id];
      }
    }
  }
  /* Lookup */
  float *nnet_logprobs = (float *)malloc(sizeof(float) * (size_t)(n_index + 1));
  for (int32_t i = 0; i < n_index; i++) nnet_logprobs[i] = nnet_output[(int64_t)idx_row[i] * row_stride + idx_pdf[i]];

  /* Forward */
  double *log_alpha = (double *)malloc(sizeof(double) * (size_t)num_states);
  for (int32_t i = 0; i < num_states; i++) log_alpha[i] = -INFINITY;
  double tot_log_prob = -INFINITY;
  log_alpha[0] = 0.0;
  {
    const int32_t *it = fst_output_indexes;
    for (int32_t state = 0; state < num_states; state++) {
      double this_log_alpha = log_alpha[state];
      for (int32_t a = arc_begin[state]; a < arc_begin[state + 1]; a++, ++it) {
        float transition_logprob = -arc_weight[a];
        float pseudo_loglike = nnet_logprobs[*it];
        double *next_log_alpha = &log_alpha[arc_next[a]];
        *next_log_alpha = log_add(*next_log_alpha, pseudo_loglike + transition_logprob + this_log_alpha);
      }
      if (!isinf(final_weight[state])) {
        float final_logprob = -final_weight[state];
        tot_log_prob = log_add(tot_log_prob, this_log_alpha + final_logprob);
      }
    }
  }
  *logprob_weighted = (float)(tot_log_prob * sup_weight);

  if (deriv) {
    /* Backward */
    double *log_beta = (double *)malloc(sizeof(double) * (size_t)num_states);
    float *nnet_logprob_derivs = (float *)calloc((size_t)(n_index + 1), sizeof(float));
    const int32_t *end_it = fst_output_indexes + n_out;
    for (int32_t state = num_states - 1; state >= 0; state--) {
      int32_t this_num_arcs = arc_begin[state + 1] - arc_begin[state];
      end_it -= this_num_arcs;
      const int32_t *it = end_it;
      double this_log_beta = -(double)final_weight[state];
      double this_log_alpha = log_alpha[state];
      for (int32_t a = arc_begin[state]; a < arc_begin[state + 1]; a++, it++) {
        double next_log_beta = log_beta[arc_next[a]];
        float transition_logprob = -arc_weight[a];
        float pseudo_loglike = nnet_logprobs[*it];
        this_log_beta = log_add(this_log_beta, pseudo_loglike + transition_logprob + next_log_beta);
        float occupation_logprob =
            (float)(this_log_alpha + pseudo_loglike + transition_logprob + next_log_beta - tot_log_prob);
        float occupation_prob = expf(occupation_logprob);
        nnet_logprob_derivs[*it] += occupation_prob;
      }
      log_beta[state] = this_log_beta;
    }
    /* AddElements(weight, indexes, derivs) */
    for (int32_t i = 0; i < n_index; i++)
      deriv[(int64_t)idx_row[i] * deriv_stride + idx_pdf[i]] += sup_weight * nnet_logprob_derivs[i];
    free(log_beta);
    free(nnet_logprob_derivs);
  }
  free(state_times);
  free(fst_output_indexes);
  free(idx_row);
  free(idx_pdf);
  free(map_this_frame);
  free(touched);
  free(nnet_logprobs);
  free(log_alpha);
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * [K] ComputeChainObjfAndDeriv (chain-training.cc); reference call site
 * src/my_lib_chain.cpp:129-131 and src/chain-supervision-test.hpp:263-265,303-307,642-644.
 * results = {objf, l2_term, weight} as the reference's THFloatTensor (my_lib_chain.cpp:126,130).
 * deriv (nullable) is zeroed here; xent_deriv (nullable) receives the numerator part.
 * ------------------------------------------------------------------------------------------ */
int oracle_compute_chain_objf_and_deriv(
    const oracle_den_graph *g, float l2_regularize, float leaky_hmm_coefficient, float xent_regularize,
    float sup_weight, int32_t num_sequences, int32_t frames_per_sequence, int32_t label_dim,
    int32_t num_states, const int32_t *arc_begin, const int32_t *arc_ilabel, const float *arc_weight,
    const int32_t *arc_next, const float *final_weight, const float *nnet_output, int64_t num_rows,
    int32_t num_cols, int64_t row_stride, float *results3, float *deriv, int64_t deriv_stride,
    float *xent_deriv, int64_t xent_stride) {
  (void)xent_regularize;
  float num_logprob_weighted = 0.0f, den_logprob = 0.0f;
  if (deriv)
    for (int64_t r = 0; r < num_rows; r++) memset(deriv + r * deriv_stride, 0, sizeof(float) * (size_t)num_cols);
  int rc;
  if (deriv) {
    rc = oracle_num_forward_backward(sup_weight, num_sequences, frames_per_sequence, label_dim, num_states,
                                     arc_begin, arc_ilabel, arc_weight, arc_next, final_weight, nnet_output,
                                     num_rows, num_cols, row_stride, deriv, deriv_stride, &num_logprob_weighted);
    if (rc) return rc;
    if (xent_deriv)
      for (int64_t r = 0; r < num_rows; r++)
        memcpy(xent_deriv + r * xent_stride, deriv + r * deriv_stride, sizeof(float) * (size_t)num_cols);
  } else {
    if (xent_deriv)
      for (int64_t r = 0; r < num_rows; r++) memset(xent_deriv + r * xent_stride, 0, sizeof(float) * (size_t)num_cols);
    rc = oracle_num_forward_backward(sup_weight, num_sequences, frames_per_sequence, label_dim, num_states,
                                     arc_begin, arc_ilabel, arc_weight, arc_next, final_weight, nnet_output,
                                     num_rows, num_cols, row_stride, xent_deriv, xent_stride, &num_logprob_weighted);
    if (rc) return rc;
  }
  int32_t ok = 1;
  rc = oracle_den_forward_backward(g, leaky_hmm_coefficient, num_sequences, nnet_output, num_rows, num_cols,
                                   row_stride, -sup_weight, deriv, deriv_stride, &den_logprob, &ok, NULL, NULL);
  if (rc) return rc;
  float objf = num_logprob_weighted - sup_weight * den_logprob;
  float weight = sup_weight * num_sequences * frames_per_sequence;
  if (!(objf - objf == 0) || !ok) {
    if (deriv)
      for (int64_t r = 0; r < num_rows; r++) memset(deriv + r * deriv_stride, 0, sizeof(float) * (size_t)num_cols);
    if (xent_deriv)
      for (int64_t r = 0; r < num_rows; r++) memset(xent_deriv + r * xent_stride, 0, sizeof(float) * (size_t)num_cols);
    float default_objf = -10;
    objf = default_objf * weight;
  }
  float l2_term;
  if (l2_regularize == 0.0f) {
    l2_term = 0.0f;
  } else {
    float scale = sup_weight * l2_regularize;
    double tr = 0.0; /* TraceMatMat(y, y, kTrans) */
    for (int64_t r = 0; r < num_rows; r++)
      for (int32_t p = 0; p < num_cols; p++) {
        float v = nnet_output[r * row_stride + p];
        tr += (double)v * v;
      }
    l2_term = (float)(-0.5 * scale * (float)tr);
    if (deriv)
      for (int64_t r = 0; r < num_rows; r++)
        for (int32_t p = 0; p < num_cols; p++)
          deriv[r * deriv_stride + p] += -1.0f * scale * nnet_output[r * row_stride + p];
  }
  results3[0] = objf;
  results3[1] = l2_term;
  results3[2] = weight;
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * cpu_baseline helper for bench.py: the denominator forward-backward only (the metric's timed
 * region), over independent blocks of sequences on `threads` host threads.  Kaldi's CPU chain
 * path is single-threaded; threads > 1 is the "all host cores" variant of BASELINE.md section 2.
 * Sequences never interact, so a block of sequences is an exact sub-problem.
 * ------------------------------------------------------------------------------------------ */
int oracle_den_forward_backward_blocks(const oracle_den_graph *g, float leaky, int32_t num_sequences,
                                       int32_t frames, const float *nnet_output, int64_t row_stride,
                                       float deriv_weight, float *deriv, int64_t deriv_stride,
                                       int32_t block, int32_t threads, double *logprob_sum) {
  const int32_t P = g->num_pdfs;
  int nblocks = (num_sequences + block - 1) / block;
  double total = 0.0;
  int err = 0;
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) reduction(+ : total) schedule(dynamic, 1)
#endif
  for (int b = 0; b < nblocks; b++) {
    int s0 = b * block, sb = (s0 + block <= num_sequences) ? block : num_sequences - s0;
    float *yb = (float *)malloc(sizeof(float) * (size_t)frames * sb * P);
    float *db = (float *)calloc((size_t)frames * sb * P, sizeof(float));
    for (int t = 0; t < frames; t++)
      for (int s = 0; s < sb; s++)
        memcpy(yb + ((int64_t)t * sb + s) * P, nnet_output + ((int64_t)t * num_sequences + s0 + s) * row_stride,
               sizeof(float) * (size_t)P);
    float lp = 0.f;
    int32_t ok = 1;
    int rc = oracle_den_forward_backward(g, leaky, sb, yb, (int64_t)frames * sb, P, P, deriv_weight, db, P, &lp,
                                         &ok, NULL, NULL);
    if (rc) err = rc;
    if (deriv)
      for (int t = 0; t < frames; t++)
        for (int s = 0; s < sb; s++) {
          float *dst = deriv + ((int64_t)t * num_sequences + s0 + s) * deriv_stride;
          const float *src = db + ((int64_t)t * sb + s) * P;
          for (int p = 0; p < P; p++) dst[p] += src[p];
        }
    total += lp;
    free(yb);
    free(db);
  }
  (void)threads;
  *logprob_sum = total;
  return err;
}
