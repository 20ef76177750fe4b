// Numerator forward-backward and the scalar epilogue of the chain objective for gfx950.
//
// [K] NumeratorComputation (chain-numerator.cc) runs a serial CPU loop over the merged supervision
// FST, bracketed by a device gather and a device scatter-add.  The merged FST factors per sequence
// (supervision.cpp), so here one workgroup of two wavefronts owns one sequence and walks its T time
// levels in LDS (one wave forward, one backward), in the log semiring and in double precision like Kaldi;
// lanes run over the states of a level.
// Occupation probabilities are summed per unique (frame, pdf) in arc order (the order Kaldi adds
// them), then added to the derivative with one plain read-modify-write per unique index -- each
// (row, pdf) is owned by exactly one lane, so no atomics.
#include <math.h>

#include "chain_internal.h"

namespace tc {

__device__ __forceinline__ double log_add(double x, double y) {
  // [K] LogAdd(double, double), kMinLogDiffDouble = log(DBL_EPSILON)
  double diff;
  if (x < y) {
    diff = x - y;
    x = y;
  } else {
    diff = y - x;
  }
  if (diff >= -36.04365338911715) return x + log1p(exp(diff));
  return x;
}

// Two wavefronts per sequence: the alpha and the beta recursions do not depend on each other, so wave 0
// walks the levels forward while wave 1 walks them backward (T steps each, one shared barrier per step);
// the occupations need both and are computed afterwards, one arc per lane.  Per state the operations and
// their order are Kaldi's.
//
// STAGE: the sequence's own tables (level bounds, in / out offsets, in-arc list, arc source / destination / unique id /
// weight: a few KB) are copied into LDS first.  The T steps are one chain of dependent loads per step -- offsets ->
// arc ids -> arc fields -> alpha -- and from global memory every link was an L2 round trip: 0.73 us per step, 0.11 ms
// for 150 frames whatever the batch; from LDS the chain is a fraction of that.
// the 128 threads' shares of a sequence's cross-entropy objective, added in a fixed order
__device__ __forceinline__ void seq_xent_sum(double v, double *out, int tid) {
  __shared__ double xo_sh[128];
  xo_sh[tid] = v;
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
    for (int i = 0; i < 128; ++i) t += xo_sh[i];
    *out = t;
  }
}

template <bool STAGE>
__global__ __launch_bounds__(128) void num_fwd_bwd_kernel(const NumParams p) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  double *log_alpha = reinterpret_cast<double *>(lds_raw);
  double *log_beta = log_alpha + p.lds_states;
  float *ylp = reinterpret_cast<float *>(log_beta + p.lds_states);  // per unique (frame, pdf): y
  float *occ = ylp + p.lds_uniq;                                    // per arc: occupation prob
  int *s_in_begin = reinterpret_cast<int *>(occ + p.lds_arcs);      // STAGE: copies of the tables, in this order
  int *s_out_begin = s_in_begin + p.lds_states + 2;
  int *s_in_arc = s_out_begin + p.lds_states + 2;
  int *s_src = s_in_arc + p.lds_arcs;
  int *s_dst = s_src + p.lds_arcs;
  int *s_uq = s_dst + p.lds_arcs;
  float *s_lw = reinterpret_cast<float *>(s_uq + p.lds_arcs);
  int *s_level = reinterpret_cast<int *>(s_lw + p.lds_arcs);
  __shared__ double tot_sh;

  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.T, S = p.S;
  const int sb = p.t.seq_state_off[q], nst = p.t.seq_state_off[q + 1] - sb;
  const int ab = p.t.seq_arc_off[q], narc = p.t.seq_arc_off[q + 1] - ab;
  const int ub = p.t.seq_uniq_off[q], nu = p.t.seq_uniq_off[q + 1] - ub;
  const int *g_level = p.t.level_begin + (int64_t)q * (T + 2);
  const int *g_out_begin = p.t.out_begin + sb + q;
  const int *g_in_begin = p.t.in_begin + sb + q;
  const int *g_in_arc = p.t.in_arc + ab;
  const int *g_arc_src = p.t.arc_src + ab, *g_arc_dst = p.t.arc_dst + ab, *g_arc_uniq = p.t.arc_uniq + ab;
  const float *g_arc_logw = p.t.arc_logw + ab;
  if (STAGE) {
    for (int i = tid; i <= nst; i += 128) {
      s_in_begin[i] = g_in_begin[i];
      s_out_begin[i] = g_out_begin[i];
    }
    for (int a = tid; a < narc; a += 128) {
      s_in_arc[a] = g_in_arc[a];
      s_src[a] = g_arc_src[a];
      s_dst[a] = g_arc_dst[a];
      s_uq[a] = g_arc_uniq[a];
      s_lw[a] = g_arc_logw[a];
    }
    for (int i = tid; i < T + 2; i += 128) s_level[i] = g_level[i];
  }
  // (separate names per address space: a pointer that may be either would turn every access into a flat load)
  auto level = [&](int i) __attribute__((always_inline)) { return STAGE ? s_level[i] : g_level[i]; };
  auto in_begin = [&](int i) __attribute__((always_inline)) { return STAGE ? s_in_begin[i] : g_in_begin[i]; };
  auto out_begin = [&](int i) __attribute__((always_inline)) { return STAGE ? s_out_begin[i] : g_out_begin[i]; };
  auto in_arc = [&](int i) __attribute__((always_inline)) { return STAGE ? s_in_arc[i] : g_in_arc[i]; };
  auto arc_src = [&](int a) __attribute__((always_inline)) { return STAGE ? s_src[a] : g_arc_src[a]; };
  auto arc_dst = [&](int a) __attribute__((always_inline)) { return STAGE ? s_dst[a] : g_arc_dst[a]; };
  auto arc_uniq = [&](int a) __attribute__((always_inline)) { return STAGE ? s_uq[a] : g_arc_uniq[a]; };
  auto arc_logw = [&](int a) __attribute__((always_inline)) { return STAGE ? s_lw[a] : g_arc_logw[a]; };
  const float *final_logw = p.t.final_logw + sb;
  const bool want_beta = p.deriv != nullptr || p.xent != nullptr || p.seq_xent != nullptr;

  // gather: [K] nnet_output_.Lookup(nnet_output_indexes_, ...); row = t*S + q
  for (int u = tid; u < nu; u += 128)
    ylp[u] = p.y_bct ? p.y[((int64_t)q * p.P + p.t.uniq_pdf[ub + u]) * T + p.t.uniq_t[ub + u]]
                     : p.y[((int64_t)p.t.uniq_t[ub + u] * S + q) * p.y_stride + p.t.uniq_pdf[ub + u]];
  for (int i = tid; i < nst; i += 128) {
    log_alpha[i] = i == 0 ? 0.0 : -INFINITY;
    log_beta[i] = -INFINITY;
  }
  __syncthreads();
  if (wave == 1 && want_beta)
    for (int st = level(T) + lane; st < level(T + 1); st += 64) log_beta[st] = (double)final_logw[st];
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    if (wave == 0) {
      // forward: level step+1 states take the log-sum over their in-arcs, in arc order
      const int l0 = level(step + 1), l1 = level(step + 2);
      for (int st = l0 + lane; st < l1; st += 64) {
        double acc = -INFINITY;
        for (int i = in_begin(st); i < in_begin(st + 1); ++i) {
          const int a = in_arc(i);
          const float sc = ylp[arc_uniq(a)] + arc_logw(a);  // float sum, as Kaldi
          acc = log_add(acc, (double)sc + log_alpha[arc_src(a)]);
        }
        log_alpha[st] = acc;
      }
    } else if (want_beta) {
      // backward: level T-1-step states take the log-sum over their out-arcs
      const int t = T - 1 - step;
      const int l0 = level(t), l1 = level(t + 1);
      for (int st = l0 + lane; st < l1; st += 64) {
        double this_log_beta = -INFINITY;  // interior states are not final
        for (int a = out_begin(st); a < out_begin(st + 1); ++a) {
          const float sc = ylp[arc_uniq(a)] + arc_logw(a);
          this_log_beta = log_add(this_log_beta, (double)sc + log_beta[arc_dst(a)]);
        }
        log_beta[st] = this_log_beta;
      }
    }
    __syncthreads();
  }
  // total: log-add over final states in state order (one lane; a handful of states)
  if (tid == 0) {
    double tot = -INFINITY;
    for (int st = level(T); st < level(T + 1); ++st)
      if (final_logw[st] != -INFINITY) tot = log_add(tot, log_alpha[st] + (double)final_logw[st]);
    tot_sh = tot;
    p.seq_logprob[q] = tot;
  }
  __syncthreads();
  if (!want_beta) return;
  const double tot = tot_sh;
  // occupation of every arc: exp(alpha(src) + score + beta(dst) - tot)
  for (int a = tid; a < narc; a += 128) {
    const float sc = ylp[arc_uniq(a)] + arc_logw(a);
    const float occupation_logprob = (float)(log_alpha[arc_src(a)] + (double)sc + log_beta[arc_dst(a)] - tot);
    occ[a] = __expf(occupation_logprob);
  }
  __syncthreads();
  // scatter: [K] AddElements(weight, indexes, derivs)
  const int *uniq_begin = p.t.uniq_begin + ub + q;
  const int *uniq_arc = p.t.uniq_arc + ab;
  double xo = 0.0;
  for (int u = tid; u < nu; u += 128) {
    float sum = 0.f;
    for (int i = uniq_begin[u]; i < uniq_begin[u + 1]; ++i) sum += occ[uniq_arc[i]];
    const float v = p.weight * sum;
    const int64_t row = (int64_t)p.t.uniq_t[ub + u] * S + q;
    const int pdf = p.t.uniq_pdf[ub + u];
    if (p.staged) {
      p.t.stage[ub + u] = v;  // beside the denominator: launch_num_scatter adds it once deriv is written
      continue;
    }
    // (scales other than 1 only through tc_chain_objf_and_grad: the reference's backward, -deriv and
    // -xent_regularize * xent_deriv, formed here instead of in two more passes over the matrices)
    if (p.deriv) p.deriv[row * p.deriv_stride + pdf] += p.deriv_scale * v;
    const int64_t bct = ((int64_t)q * p.P + pdf) * p.T + p.t.uniq_t[ub + u];  // (sequence, pdf, frame) of a (B, C, T) tensor
    if (p.xent) p.xent[p.xent_bct ? bct : row * p.xent_stride + pdf] = p.xent_scale * v;
    if (p.seq_xent) xo += (double)(p.xent_scale * v) * (double)p.xent_out[p.xent_out_bct ? bct : row * p.xent_out_stride + pdf];
  }
  if (p.seq_xent && !p.staged) seq_xent_sum(xo, p.seq_xent + q, tid);
}

// Second half of a staged numerator: deriv[row, pdf] += weight * posterior, xent_deriv[row, pdf] = it, one owner
// per (row, pdf) as above.
__global__ __launch_bounds__(128) void num_scatter_kernel(const NumParams p) {
  const int q = blockIdx.x, S = p.S;
  const int ub = p.t.seq_uniq_off[q], nu = p.t.seq_uniq_off[q + 1] - ub;
  double xo = 0.0;
  for (int u = threadIdx.x; u < nu; u += 128) {
    const float v = p.t.stage[ub + u];
    const int64_t row = (int64_t)p.t.uniq_t[ub + u] * S + q;
    const int pdf = p.t.uniq_pdf[ub + u];
    if (p.deriv) p.deriv[row * p.deriv_stride + pdf] += p.deriv_scale * v;
    const int64_t bct = ((int64_t)q * p.P + pdf) * p.T + p.t.uniq_t[ub + u];  // (sequence, pdf, frame) of a (B, C, T) tensor
    if (p.xent) p.xent[p.xent_bct ? bct : row * p.xent_stride + pdf] = p.xent_scale * v;
    if (p.seq_xent) xo += (double)(p.xent_scale * v) * (double)p.xent_out[p.xent_out_bct ? bct : row * p.xent_out_stride + pdf];
  }
  if (p.seq_xent) seq_xent_sum(xo, p.seq_xent + q, threadIdx.x);
}

int launch_num_scatter(const NumParams &p, hipStream_t stream) {
  if (!p.deriv && !p.xent && !p.seq_xent) return TC_OK;
  hipLaunchKernelGGL(num_scatter_kernel, dim3(p.S), dim3(128), 0, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

int launch_num(const NumParams &p, hipStream_t stream) {
  count_launch(kCntNum);
  if (p.deriv || p.xent || p.seq_xent) count_launch(kCntNumBackward);
  const size_t lds = (size_t)p.lds_states * 16 + (size_t)p.lds_uniq * 4 + (size_t)p.lds_arcs * 4;
  if (lds > (size_t)kLdsLimitBytes) return TC_ERR_UNSUPPORTED;
  // with the sequence's tables staged in LDS when that keeps a workgroup within a quarter of a CU's LDS
  const size_t staged = lds + ((size_t)2 * (p.lds_states + 2) + (size_t)5 * p.lds_arcs + (size_t)(p.T + 2)) * 4;
  if (staged <= (size_t)kLdsLimitBytes / 4) {
    TC_HIP_CHECK(allow_dynamic_lds((const void *)num_fwd_bwd_kernel<true>, staged));
    hipLaunchKernelGGL(num_fwd_bwd_kernel<true>, dim3(p.S), dim3(128), staged, stream, p);
    TC_HIP_CHECK(hipGetLastError());
    return TC_OK;
  }
  TC_HIP_CHECK(allow_dynamic_lds((const void *)num_fwd_bwd_kernel<false>, lds));
  hipLaunchKernelGGL(num_fwd_bwd_kernel<false>, dim3(p.S), dim3(128), lds, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// ---- scalar epilogue: [K] ComputeChainObjfAndDeriv after the two computations --------------------
__global__ __launch_bounds__(256) void finalize_kernel(const double *den_lp, const double *num_lp, const double *y2,
                                                       const float *ab, const float *gs, int S, int T,
                                                       float sup_weight, float l2, int have_deriv, float *results,
                                                       int32_t *fail_flag, const double *xent_lp, double *xent_total,
                                                       float *loss_out) {
  __shared__ double sh[4][256];
  const int tid = threadIdx.x;
  double d = 0, n = 0, q = 0, a = 0, g = 0, x = 0;
  // fixed-order partial sums: thread i takes sequences i, i+256, ... (deterministic)
  for (int s = tid; s < S; s += 256) {
    d += den_lp[s];
    n += num_lp ? num_lp[s] : 0.0;
    q += y2[s];
    if (have_deriv) {
      a += (double)ab[s];
      g += (double)gs[s];
    }
    if (xent_lp) x += xent_lp[s];
  }
  sh[0][tid] = d;
  sh[1][tid] = n;
  sh[2][tid] = q;
  sh[3][tid] = a;
  __syncthreads();
  __shared__ double sh_g[256], sh_x[256];
  sh_g[tid] = g;
  sh_x[tid] = x;
  __syncthreads();
  if (tid == 0) {
    double D = 0, N = 0, Q = 0, A = 0, G = 0, X = 0;
    for (int i = 0; i < 256; ++i) {
      D += sh[0][i];
      N += sh[1][i];
      Q += sh[2][i];
      A += sh[3][i];
      G += sh_g[i];
      X += sh_x[i];
    }
    const float num_logprob_weighted = (float)(N * (double)sup_weight);
    const float den_logprob = (float)D;
    float objf = num_logprob_weighted - sup_weight * den_logprob;
    const float weight = sup_weight * (float)S * (float)T;
    bool ok = true;
    if (have_deriv) {
      // [K] BetaGeneralFrameDebug(0)
      if (fabs(A - (double)S) > 2.0 || !(A - A == 0.0)) ok = false;
      if (fabs(G - (double)S) > 2.0 || !(G - G == 0.0)) ok = false;
    }
    int fail = 0;
    if (!(objf - objf == 0.0f) || !ok) {
      objf = -10.0f * weight;
      fail = 1;
    }
    float l2_term = 0.f;
    if (l2 != 0.0f) l2_term = (float)(-0.5 * (double)(sup_weight * l2) * (double)(float)Q);
    results[0] = objf;
    results[1] = l2_term;
    results[2] = weight;
    *fail_flag = fail;
    // tc_chain_step's two scalars ride along: the sequences' cross-entropy objective sums (NumParams::seq_xent; 0 after
    // a numerical failure, as the zeroed xent_deriv gives) and the loss -objf / weight (torchain/functions.py:104)
    if (xent_total) *xent_total = fail ? 0.0 : X;
    if (loss_out) loss_out[0] = -objf / weight;
  }
}

int launch_finalize(const double *den_lp, const double *num_lp, const double *y2, const float *ab, const float *gs,
                    int S, int T, float sup_weight, float l2, int have_deriv, float *results, int32_t *fail_flag,
                    hipStream_t stream, const double *xent_lp, double *xent_total, float *loss_out) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream, den_lp, num_lp, y2, ab, gs, S, T, sup_weight, l2,
                     have_deriv, results, fail_flag, xent_lp, xent_total, loss_out);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// Resets the derivative outputs when the finalize step flagged a numerical failure (rare path; exits
// after one uniform load otherwise).  [K] zeroes both derivatives and only then adds the l2 term, so
// deriv becomes -l2_scale*y and xent_deriv 0.
__global__ __launch_bounds__(256) void zero_on_fail_kernel(const int32_t *fail_flag, float *a, int64_t a_stride,
                                                           float *b, int64_t b_stride, const float *y,
                                                           int64_t y_stride, float l2_scale, int64_t rows, int cols) {
  if (*fail_flag == 0) return;
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    if (a) a[r * a_stride + c] = -l2_scale * y[r * y_stride + c];
    if (b) b[r * b_stride + c] = 0.f;
  }
}

int launch_zero_on_fail(const int32_t *fail_flag, float *a, int64_t a_stride, float *b, int64_t b_stride,
                        const float *y, int64_t y_stride, float l2_scale, int64_t rows, int cols,
                        hipStream_t stream) {
  if (!a && !b) return TC_OK;
  hipLaunchKernelGGL(zero_on_fail_kernel, dim3(1024), dim3(256), 0, stream, fail_flag, a, a_stride, b, b_stride, y, y_stride,
                     l2_scale, rows, cols);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// Reduction used by the split entry point tc_den_forward_backward.
__global__ __launch_bounds__(256) void den_reduce_kernel(const double *den_lp, const float *ab, const float *gs, int S,
                                                         double *logprob_out, int32_t *status_out) {
  __shared__ double sh[3][256];
  const int tid = threadIdx.x;
  double d = 0, a = 0, g = 0;
  for (int s = tid; s < S; s += 256) {
    d += den_lp[s];
    if (ab) {
      a += (double)ab[s];
      g += (double)gs[s];
    }
  }
  sh[0][tid] = d;
  sh[1][tid] = a;
  sh[2][tid] = g;
  __syncthreads();
  if (tid == 0) {
    double D = 0, A = 0, G = 0;
    for (int i = 0; i < 256; ++i) {
      D += sh[0][i];
      A += sh[1][i];
      G += sh[2][i];
    }
    if (logprob_out) *logprob_out = D;
    if (status_out) {
      int bad = 0;
      if (!(D - D == 0.0)) bad = 1;
      if (ab && (fabs(A - (double)S) > 2.0 || fabs(G - (double)S) > 2.0 || !(A - A == 0.0) || !(G - G == 0.0))) bad = 1;
      *status_out = bad;
    }
  }
}

int launch_den_reduce(const double *den_lp, const float *ab, const float *gs, int S, double *logprob_out,
                      int32_t *status_out, hipStream_t stream) {
  hipLaunchKernelGGL(den_reduce_kernel, dim3(1), dim3(256), 0, stream, den_lp, ab, gs, S, logprob_out, status_out);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

__global__ __launch_bounds__(256) void sum_double_kernel(const double *in, int n, double scale, double *out) {
  __shared__ double sh[256];
  const int tid = threadIdx.x;
  double d = 0;
  for (int i = tid; i < n; i += 256) d += in[i];
  sh[tid] = d;
  __syncthreads();
  if (tid == 0) {
    double D = 0;
    for (int i = 0; i < 256; ++i) D += sh[i];
    *out = D * scale;
  }
}

// tc_chain_step: loss = -objf / weight (the reference's input.new([results.loss]), torchain/functions.py:104) on the device
__global__ void step_loss_kernel(const float *results3, float *loss1) { loss1[0] = -results3[0] / results3[2]; }

int launch_step_loss(const float *results3, float *loss1, hipStream_t stream) {
  hipLaunchKernelGGL(step_loss_kernel, dim3(1), dim3(1), 0, stream, results3, loss1);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// the sequences' cross-entropy objective sums (NumParams::seq_xent) -> their total; 0 after a numerical failure, as the
// zeroed xent_deriv gives
__global__ __launch_bounds__(256) void xent_total_kernel(const double *in, int n, const int32_t *fail_flag, double *out) {
  __shared__ double sh[256];
  const int tid = threadIdx.x;
  double d = 0;
  for (int i = tid; i < n; i += 256) d += in[i];
  sh[tid] = d;
  __syncthreads();
  if (tid == 0) {
    double D = 0;
    for (int i = 0; i < 256; ++i) D += sh[i];
    *out = *fail_flag ? 0.0 : D;
  }
}

int launch_xent_total(const double *in, int n, const int32_t *fail_flag, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(xent_total_kernel, dim3(1), dim3(256), 0, stream, in, n, fail_flag, out);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

int launch_sum_double(const double *in, int n, double scale, double *out, hipStream_t stream) {
  hipLaunchKernelGGL(sum_double_kernel, dim3(1), dim3(256), 0, stream, in, n, scale, out);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
