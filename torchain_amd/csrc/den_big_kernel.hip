// Streamed denominator forward-backward: the fallback for graphs whose per-frame working set does not
// fit one CU's LDS (more than 16384 states or pdfs, or a layout beyond 160 KB).
//
// Same computation as den_kernels.hip ([K] DenominatorComputation::Forward() + Backward(),
// chain-denominator.cc; reference call site src/my_lib_chain.cpp:129-131), same mapping -- one
// workgroup per sequence for all 2T frames -- but alpha' / beta / exp(y_t) live in global memory (they
// are L2-sized: a few hundred KB per sequence) and the transitions are walked as plain CSR lists
// (chain_internal.h: BigArc).  Per frame:
//   forward : one thread per state sums its in-arcs in FST order (Kaldi's CPU order);
//   backward: one thread per state sums its out-arcs (beta'), and one thread per PDF sums the arcs that
//             carry it (gamma), so the derivative row is written once, without atomics, and the result
//             does not depend on scheduling.
// This path is about correctness on any graph, not about the roofline: arc records are fetched per
// thread (uncoalesced, served by L2) and hub states serialise in one lane.
#include "chain_internal.h"

namespace tc {

namespace {

constexpr int kBigThreads = 1024;
constexpr int kBigWaves = kBigThreads / 64;

__device__ __forceinline__ float big_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup; ends with every thread past a barrier (global writes before it are visible)
__device__ __forceinline__ float big_block_sum(float v, float *red, int slot) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  v = big_wave_sum(v);
  if (lane == 0) red[slot * kBigWaves + wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < kBigWaves; ++i) t += red[slot * kBigWaves + i];
  return t;
}

__device__ __forceinline__ float big_exp(float x) {
  x = x < -30.0f ? -30.0f : x;  // compare-and-clamp keeps NaN
  x = x > 30.0f ? 30.0f : x;
  return __expf(x);
}

template <bool WANT_DERIV, bool ACCUM>
__global__ __launch_bounds__(kBigThreads) void den_big_kernel(const DenParams p) {
  extern __shared__ float lds[];
  float *red = lds;                     // 8 reduction slots x 16 waves
  float *asum_h = lds + 8 * kBigWaves;  // T + 1
  const int tid = threadIdx.x, s = blockIdx.x;
  const int S = p.S, T = p.T, H = p.H, P = p.P, Hs = p.L.Hs, Ps = p.L.Ps;
  const BigDev g = p.big;
  float *E = p.big_expy + (int64_t)s * Ps;
  float *hist = p.alpha_hist + (int64_t)s * Hs;
  const int64_t hist_step = (int64_t)S * Hs;
  const float leaky = p.leaky;
  int slot = 0;  // rotating reduction slot: a slot is rewritten only after a later barrier

  // ---- t = 0  ([K] AlphaFirstFrame + AlphaDash(0)), exp(y_0)
  float part = 0.f;
  for (int h = tid; h < H; h += kBigThreads) part += p.pi[h];
  float asum = big_block_sum(part, red, slot++ & 7);
  for (int h = tid; h < H; h += kBigThreads) {
    const float pi = p.pi[h];
    hist[h] = pi + leaky * pi * asum;
  }
  float y2 = 0.f;
  {
    const float *yrow = p.y + (int64_t)s * p.y_stride;
    for (int i = tid; i < P; i += kBigThreads) {
      const float yv = yrow[i];
      y2 += yv * yv;
      E[i] = big_exp(yv);
    }
  }
  if (tid == 0) asum_h[0] = asum;
  double logsum = 0.0;  // thread 0
  float asum_prev = asum;
  __syncthreads();

  // ---- forward frames  ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  for (int t = 1; t <= T; ++t) {
    const float *prev = hist + (int64_t)(t - 1) * hist_step;
    float *cur = hist + (int64_t)t * hist_step;
    const float inv_prev = 1.0f / asum_prev;
    part = 0.f;
    for (int h = tid; h < H; h += kBigThreads) {
      float sum = 0.f;
      const int e = g.in_begin[h + 1];
      for (int a = g.in_begin[h]; a < e; ++a) {
        const BigArc r = g.in_arc[a];
        sum += prev[r.a] * r.w * E[r.b];
      }
      const float v = sum * inv_prev;
      cur[h] = v;
      part += v;
    }
    asum = big_block_sum(part, red, slot++ & 7);  // every thread is past its reads of E and prev
    float part_tot = 0.f;
    for (int h = tid; h < H; h += kBigThreads) {
      const float a = cur[h] + leaky * p.pi[h] * asum;
      cur[h] = a;
      part_tot += a;
    }
    if (t < T) {
      const float *yrow = p.y + ((int64_t)t * S + s) * p.y_stride;
      for (int i = tid; i < P; i += kBigThreads) {
        const float yv = yrow[i];
        y2 += yv * yv;
        E[i] = big_exp(yv);
      }
    }
    if (tid == 0) {
      asum_h[t] = asum;
      logsum += (double)__logf(asum_prev);
    }
    asum_prev = asum;
    if (t == T) part = part_tot;
    __syncthreads();
  }
  // ---- total probability  ([K] ComputeTotLogLike)
  const float tot = big_block_sum(part, red, slot++ & 7);
  {
    const double y2d = (double)big_block_sum(y2, red, slot++ & 7);
    if (tid == 0) {
      p.seq_logprob[s] = logsum + (double)__logf(tot);
      p.seq_y2[s] = y2d;
    }
  }
  if (!WANT_DERIV) return;

  // ---- backward  ([K] BetaDashLastFrame, Beta(T), BetaDashGeneralFrame(t), Beta(t))
  float *Bcur = p.big_beta + (int64_t)s * Hs;                        // beta_{t+1}
  float *Bnext = p.big_beta + ((int64_t)S + s) * Hs;                 // beta'_t, then beta_t
  const float inv_tot = 1.0f / tot;
  part = 0.f;
  for (int h = tid; h < H; h += kBigThreads) part += leaky * p.pi[h] * inv_tot;
  float bsum = big_block_sum(part, red, slot++ & 7);
  for (int h = tid; h < H; h += kBigThreads) Bcur[h] = inv_tot + bsum;
  for (int t = T - 1; t >= 0; --t) {
    const float *alpha = hist + (int64_t)t * hist_step;
    const float *yrow = p.y + ((int64_t)t * S + s) * p.y_stride;
    for (int i = tid; i < P; i += kBigThreads) E[i] = big_exp(yrow[i]);
    __syncthreads();  // E, Bcur complete
    const float inv_as = 1.0f / asum_h[t];
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
    for (int h = tid; h < H; h += kBigThreads) {
      float sum = 0.f;
      const int e = g.out_begin[h + 1];
      for (int a = g.out_begin[h]; a < e; ++a) {
        const BigArc r = g.out_arc[a];
        sum += r.w * Bcur[r.a] * E[r.b];
      }
      const float bp = sum * inv_as;
      Bnext[h] = bp;
      part += leaky * p.pi[h] * bp;
      if (t == 0) part_ab += alpha[h] * bp;
    }
    {
      float *drow = p.deriv + ((int64_t)t * S + s) * p.deriv_stride;
      for (int i = tid; i < P; i += kBigThreads) {
        float sum = 0.f;
        const int e = g.pdf_begin[i + 1];
        for (int a = g.pdf_begin[i]; a < e; ++a) {
          const BigArc r = g.pdf_arc[a];
          sum += r.w * alpha[r.a] * Bcur[r.b];
        }
        const float gam = sum * E[i] * inv_as;
        if (t == 0) part_g += gam;
        float o = p.deriv_weight * gam - p.l2_scale * yrow[i];
        if (ACCUM) o += drow[i];
        drow[i] = o;
      }
    }
    bsum = big_block_sum(part, red, slot++ & 7);  // all reads of Bcur done, all beta' written
    if (t == 0) {
      const float ab = big_block_sum(part_ab, red, slot++ & 7);
      const float gs = big_block_sum(part_g, red, slot++ & 7);
      if (tid == 0) {
        p.seq_ab[s] = ab;
        p.seq_gsum[s] = gs;
      }
      break;
    }
    for (int h = tid; h < H; h += kBigThreads) Bnext[h] += bsum;
    float *tmp = Bcur;
    Bcur = Bnext;
    Bnext = tmp;
  }
}

}  // namespace

int launch_den_big(const DenParams &p, int accumulate, hipStream_t stream) {
  const size_t lds = (size_t)(8 * kBigWaves + ((p.T + 1 + 3) & ~3)) * sizeof(float);
  if (lds > (size_t)kLdsLimitBytes) return TC_ERR_UNSUPPORTED;
  void (*k)(const DenParams) = nullptr;
  if (!p.deriv)
    k = den_big_kernel<false, false>;
  else
    k = accumulate ? den_big_kernel<true, true> : den_big_kernel<true, false>;
  TC_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kBigThreads), lds, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
