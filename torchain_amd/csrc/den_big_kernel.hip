// Streamed denominator forward-backward: the path for graphs whose per-frame working set does not fit
// one CU's LDS (more than 16384 states or pdfs, or a layout beyond 160 KB) -- which includes many real
// Kaldi den.fst files (tens of thousands of states, hundreds of thousands of arcs).
//
// Same computation as den_kernels.hip ([K] DenominatorComputation::Forward() + Backward(),
// chain-denominator.cc; reference call site src/my_lib_chain.cpp:129-131), different mapping: with the
// state vectors in HBM/L2 anyway, LANES RUN OVER SEQUENCES.  alpha / beta are stored [state][sequence]
// and exp(y_t) is transposed to [pdf][sequence] once per frame, so
//   * an arc record is the same for all 64 lanes of a wave: it arrives through the scalar cache, 16 bytes
//     per 64 sequences (the on-chip kernels pay 6 bytes per arc per SEQUENCE);
//   * every gather is a coalesced 256-byte row segment (64 sequences x 4 bytes) served by L2 / Infinity
//     Cache -- the per-frame alpha matrix (H x S x 4 bytes) is a few tens of MB.
// One launch per frame and pass (the frame recursion is a grid-wide dependency).  Sums over states use
// per-block partials reduced by a second small kernel in a fixed order: no float atomics, results do not
// depend on scheduling.  gamma is computed by one wave per pdf from a by-pdf arc list and transposed back
// to the derivative's row-major layout through LDS.
//
// The history holds UN-dashed alpha_t and the per-frame sums; alpha'_t = alpha_t + leaky*pi*asum_t and
// beta_t = beta'_t + bsum_t are formed on the fly by their consumers.
#include "chain_internal.h"

namespace tc {

namespace {

constexpr int kBT = 256;             // threads per block: 4 waves
constexpr int kStatesPerBlock = 16;  // forward / backward kernels: 4 states per wave
constexpr int kPdfsPerBlock = 64;    // gamma kernel: 16 pdfs per wave

__device__ __forceinline__ float big_exp(float x) {
  x = x < -30.0f ? -30.0f : x;  // compare-and-clamp keeps NaN
  x = x > 30.0f ? 30.0f : x;
  return __expf(x);
}

// small per-sequence arrays inside p.big_small, all with row length Sp
struct BigSmall {
  float *asum;       // [T + 1][Sp]
  float *bsum;       // [2][Sp]      leaky * sum_h pi(h) beta'(h), frames t+1 / t alternate
  float *inv_tot;    // [Sp]
  float *ab, *gs;    // [Sp]  alpha'_0 . beta'_0 and sum_pdf gamma_0
  float *part_a;     // [ceil(H / 16)][Sp]   also the backward's bsum partials
  float *part_ab;    // [ceil(H / 16)][Sp]
  float *part_g;     // [ceil(P / 64)][Sp]
  float *part_y2;    // [ceil(P / 64)][Sp]  running sum of y^2 per pdf tile
};

__host__ __device__ inline BigSmall big_small(const DenParams &p) {
  const int64_t Sp = p.big_Sp;
  const int64_t hb = (p.H + kStatesPerBlock - 1) / kStatesPerBlock, pb = (p.P + kPdfsPerBlock - 1) / kPdfsPerBlock;
  BigSmall s;
  float *q = p.big_small;
  s.asum = q;
  q += (int64_t)(p.T + 1) * Sp;
  s.bsum = q;
  q += 2 * Sp;
  s.inv_tot = q;
  q += Sp;
  s.ab = q;
  q += Sp;
  s.gs = q;
  q += Sp;
  s.part_a = q;
  q += hb * Sp;
  s.part_ab = q;
  q += hb * Sp;
  s.part_g = q;
  q += pb * Sp;
  s.part_y2 = q;
  q += pb * Sp;
  return s;
}

// expT[pdf][s] = exp(y[t*S + s][pdf]); padding lanes (s >= S) get exp(0)
// SUM_SQ (forward pass): also accumulates sum(y^2) per sequence into this block's own slot of part_y2
// (one block per (pdf tile, sequence group), launches are ordered: a plain read-modify-write)
template <bool SUM_SQ>
__global__ __launch_bounds__(kBT) void big_exp_kernel(const DenParams p, int t) {
  __shared__ float tile[64][65];
  __shared__ float sq[64][65];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int p0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
  {
    // all 16 row loads of a wave first (one round trip instead of sixteen), then exp and the LDS stores
    float yv[16];
    const int pdf = p0 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int s = s0 + wave + 4 * i;
      yv[i] = (s < p.S && pdf < p.P) ? p.y[((int64_t)t * p.S + s) * p.y_stride + pdf] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int sl = wave + 4 * i;
      tile[sl][lane] = big_exp(yv[i]);
      if (SUM_SQ) sq[sl][lane] = yv[i] * yv[i];
    }
  }
  __syncthreads();
  if (SUM_SQ && wave == 0) {
    float acc = 0.f;
    for (int i = 0; i < 64; ++i) acc += sq[lane][i];  // sequence s0 + lane, this tile's 64 pdfs
    float *slot = big_small(p).part_y2 + (int64_t)blockIdx.x * p.big_Sp + s0 + lane;
    *slot = (t == 0 ? 0.f : *slot) + acc;
  }
  for (int pl = wave; pl < 64; pl += 4) {
    const int pdf = p0 + pl;
    if (pdf < p.P) p.big_expy[(int64_t)pdf * p.big_Sp + s0 + lane] = tile[lane][pl];
  }
}

// The same for ALL frames in one launch (when the workspace has room for T transposed frames: api.cpp): one block
// per (pdf tile, sequence group) loops over the frames, so sum(y^2) is accumulated in the same order as by the
// per-frame launches, and the forward and the backward pass share the result -- 2 T launches and T redundant
// exp-transposes fewer.
__global__ __launch_bounds__(kBT) void big_exp_all_kernel(const DenParams p) {
  __shared__ float tile[64][65];
  __shared__ float sq[64][65];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int p0 = blockIdx.x * 64, s0 = blockIdx.y * 64;
  float y2 = 0.f;
  for (int t = 0; t < p.T; ++t) {
    float yv[16];
    const int pdf = p0 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int s = s0 + wave + 4 * i;
      yv[i] = (s < p.S && pdf < p.P) ? p.y[((int64_t)t * p.S + s) * p.y_stride + pdf] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int sl = wave + 4 * i;
      tile[sl][lane] = big_exp(yv[i]);
      sq[sl][lane] = yv[i] * yv[i];
    }
    __syncthreads();
    if (wave == 0) {
      float acc = 0.f;
      for (int i = 0; i < 64; ++i) acc += sq[lane][i];
      y2 = y2 + acc;  // (the per-frame kernels add frame by frame in this order)
    }
    float *const out = p.big_expy + p.big_exp_stride * t;
    for (int pl = wave; pl < 64; pl += 4) {
      const int pd = p0 + pl;
      if (pd < p.P) out[(int64_t)pd * p.big_Sp + s0 + lane] = tile[lane][pl];
    }
    __syncthreads();
  }
  if (wave == 0) big_small(p).part_y2[(int64_t)blockIdx.x * p.big_Sp + s0 + lane] = y2;
}

// alpha_0 = pi for every sequence; asum_0 = sum(pi)   ([K] AlphaFirstFrame)
__global__ __launch_bounds__(kBT) void big_alpha0_kernel(const DenParams p) {
  const int64_t n = (int64_t)p.H * p.big_Sp;
  const BigSmall sm = big_small(p);
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBT)
    p.alpha_hist[i] = p.pi[i / p.big_Sp];
  if (blockIdx.x == 0)
    for (int s = threadIdx.x; s < p.big_Sp; s += kBT) sm.asum[s] = p.big_sum_pi;
}

// sums a [rows][Sp] array of per-block partials over the rows, in a fixed order; 16 waves, each with
// eight independent row loads in flight (a serial loop here cost more than the frame kernel it follows)
constexpr int kRT = 1024, kRW = kRT / 64;
__device__ __forceinline__ float big_colsum(const float *part, int rows, int Sp, int s, int wave, float (*red)[64],
                                            int lane) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = wave; r < rows; r += 8 * kRW) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int rr = r + u * kRW;
      acc[u] += rr < rows ? part[(int64_t)rr * Sp + s] : 0.f;
    }
  }
  red[wave][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < kRW; ++w) t += red[w][lane];
  __syncthreads();
  return t;
}

// Sum over one CSR segment, kArcUnroll arcs at a time: the records (wave-uniform, scalar loads) and then
// all the row segments of a group are requested before the first multiply, so a wave keeps 2 * kArcUnroll
// 256-byte loads in flight instead of one dependent pair (the frame kernels are latency-bound otherwise).
// Arcs are still accumulated in list order.  term(r) returns the arc's product for this lane.
constexpr int kArcUnroll = 8;
template <class Term>
__device__ __forceinline__ float big_arc_sum(const BigArc *__restrict__ arcs, int e0, int e1, Term term) {
  float sum = 0.f;
  for (int a = e0; a < e1; a += kArcUnroll) {
    BigArc r[kArcUnroll];
#pragma unroll
    for (int u = 0; u < kArcUnroll; ++u) {
      r[u] = arcs[min(a + u, e1 - 1)];
      if (a + u >= e1) r[u].w = 0.f;  // past the end: a repeat of the last arc with weight 0
    }
    float v[kArcUnroll];
#pragma unroll
    for (int u = 0; u < kArcUnroll; ++u) v[u] = term(r[u]);
#pragma unroll
    for (int u = 0; u < kArcUnroll; ++u) sum += v[u];
  }
  return sum;
}

// forward frame t: alpha_t(h) = sum_in w * alpha'_{t-1}(src) * p_{t-1}(pdf) / asum_{t-1}   ([K] AlphaGeneralFrame)
__global__ __launch_bounds__(kBT) void big_fwd_kernel(const DenParams p, int t) {
  __shared__ float red[4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int Sp = p.big_Sp, s = blockIdx.y * 64 + lane;
  const BigSmall sm = big_small(p);
  const float *prev = p.alpha_hist + (int64_t)(t - 1) * p.H * Sp + s;
  float *cur = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t - 1) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_prev = sm.asum[(int64_t)(t - 1) * Sp + s];
  const float inv = 1.0f / asum_prev, cl_as = p.leaky * asum_prev;
  float part = 0.f;
  const int h0 = blockIdx.x * kStatesPerBlock + wave * (kStatesPerBlock / 4);
  for (int k = 0; k < kStatesPerBlock / 4; ++k) {
    const int h = h0 + k;
    if (h >= p.H) break;
    const int e0 = p.big.in_begin[h], e1 = p.big.in_begin[h + 1];
    const float sum = big_arc_sum(p.big.in_arc, e0, e1, [&](const BigArc &r) {
      return r.w * (prev[(int64_t)r.a * Sp] + cl_as * r.pi) * E[(int64_t)r.b * Sp];
    });
    const float v = sum * inv;
    cur[(int64_t)h * Sp] = v;
    part += v;
  }
  red[wave][lane] = part;
  __syncthreads();
  if (wave == 0) sm.part_a[(int64_t)blockIdx.x * Sp + s] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// asum_t = sum over states of alpha_t
__global__ __launch_bounds__(kRT) void big_asum_kernel(const DenParams p, int t) {
  __shared__ float red[kRW][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int Sp = p.big_Sp, s = blockIdx.x * 64 + lane;
  const BigSmall sm = big_small(p);
  const int rows = (p.H + kStatesPerBlock - 1) / kStatesPerBlock;
  const float tot = big_colsum(sm.part_a, rows, Sp, s, wave, red, lane);
  if (wave == 0) sm.asum[(int64_t)t * Sp + s] = tot;
}

// log-prob, 1/tot and beta_T after the last forward frame; sum of y^2 per sequence from the tile partials
__global__ __launch_bounds__(kBT) void big_total_kernel(const DenParams p) {
  const int s = blockIdx.x * kBT + threadIdx.x;
  if (s >= p.S) return;
  const int Sp = p.big_Sp;
  const BigSmall sm = big_small(p);
  double y2 = 0.0;
  const int pb = (p.P + kPdfsPerBlock - 1) / kPdfsPerBlock;
  for (int b = 0; b < pb; ++b) y2 += (double)sm.part_y2[(int64_t)b * Sp + s];
  p.seq_y2[s] = y2;
  // tot = sum_h alpha'_T(h) = asum_T * (1 + leaky * sum(pi))   ([K] ComputeTotLogLike)
  const float tot = sm.asum[(int64_t)p.T * Sp + s] * (1.0f + p.leaky * p.big_sum_pi);
  double lp = (double)__logf(tot);
  for (int t = 0; t < p.T; ++t) lp += (double)__logf(sm.asum[(int64_t)t * Sp + s]);
  p.seq_logprob[s] = lp;
  sm.inv_tot[s] = 1.0f / tot;
  // beta'_T = 1/tot for every state; beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h)
  sm.bsum[(p.T & 1) * Sp + s] = p.leaky * p.big_sum_pi * (1.0f / tot);
}

__global__ __launch_bounds__(kBT) void big_beta_init_kernel(const DenParams p) {
  const int Sp = p.big_Sp;
  const BigSmall sm = big_small(p);
  float *B = p.big_beta + (int64_t)(p.T & 1) * p.H * Sp;
  const int64_t n = (int64_t)p.H * Sp;
  for (int64_t i = (int64_t)blockIdx.x * kBT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBT) {
    const int s = (int)(i % Sp);
    B[i] = s < p.S ? sm.inv_tot[s] : 0.f;
  }
  if (blockIdx.x == 0)  // padding lanes: defined values
    for (int s = p.S + threadIdx.x; s < Sp; s += kBT) sm.bsum[(p.T & 1) * Sp + s] = 0.f;
}

// backward frame t: beta'_t(h) = sum_out w * beta_{t+1}(dst) * p_t(pdf) / asum_t   ([K] BetaDashGeneralFrame)
__global__ __launch_bounds__(kBT) void big_bwd_kernel(const DenParams p, int t) {
  __shared__ float red[2][4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int Sp = p.big_Sp, s = blockIdx.y * 64 + lane;
  const BigSmall sm = big_small(p);
  const float *Bprev = p.big_beta + (int64_t)((t + 1) & 1) * p.H * Sp + s;
  float *Bcur = p.big_beta + (int64_t)(t & 1) * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s];
  const float *alpha = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  float part = 0.f, part_ab = 0.f;
  const int h0 = blockIdx.x * kStatesPerBlock + wave * (kStatesPerBlock / 4);
  for (int k = 0; k < kStatesPerBlock / 4; ++k) {
    const int h = h0 + k;
    if (h >= p.H) break;
    const int e0 = p.big.out_begin[h], e1 = p.big.out_begin[h + 1];
    const float sum = big_arc_sum(p.big.out_arc, e0, e1, [&](const BigArc &r) {
      return r.w * (Bprev[(int64_t)r.a * Sp] + bs) * E[(int64_t)r.b * Sp];
    });
    const float bp = sum * inv_as;
    Bcur[(int64_t)h * Sp] = bp;
    const float cpi = p.leaky * p.pi[h];
    part += cpi * bp;
    if (t == 0) part_ab += (alpha[(int64_t)h * Sp] + cpi * asum_t) * bp;
  }
  red[0][wave][lane] = part;
  red[1][wave][lane] = part_ab;
  __syncthreads();
  if (wave == 0) {
    sm.part_a[(int64_t)blockIdx.x * Sp + s] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    if (t == 0)
      sm.part_ab[(int64_t)blockIdx.x * Sp + s] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  }
}

// gamma_t(pdf) = p_t(pdf) / asum_t * sum over the arcs carrying pdf of w * alpha'_t(src) * beta_{t+1}(dst);
// derivative row written through an LDS transpose   ([K] BetaDashGeneralFrame's log_nnet_output_deriv part)
template <bool ACCUM>
__global__ __launch_bounds__(kRT) void big_gamma_kernel(const DenParams p, int t) {
  __shared__ float tile[64][65];
  __shared__ float red[kRW][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int Sp = p.big_Sp, s0 = blockIdx.y * 64, s = s0 + lane;
  const BigSmall sm = big_small(p);
  const float *alpha = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  const float *Bprev = p.big_beta + (int64_t)((t + 1) & 1) * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s], cl_as = p.leaky * asum_t;
  const int p0 = blockIdx.x * kPdfsPerBlock;
  float part_g = 0.f;
  for (int k = 0; k < kPdfsPerBlock / kRW; ++k) {
    const int pl = wave * (kPdfsPerBlock / kRW) + k, pdf = p0 + pl;
    float gam = 0.f;
    if (pdf < p.P) {
      const int e0 = p.big.pdf_begin[pdf], e1 = p.big.pdf_begin[pdf + 1];
      const float sum = big_arc_sum(p.big.pdf_arc, e0, e1, [&](const BigArc &r) {
        return r.w * (alpha[(int64_t)r.a * Sp] + cl_as * r.pi) * (Bprev[(int64_t)r.b * Sp] + bs);
      });
      gam = sum * E[(int64_t)pdf * Sp] * inv_as;
    }
    tile[pl][lane] = gam;
    part_g += gam;
  }
  red[wave][lane] = part_g;
  __syncthreads();
  if (t == 0 && wave == 0) {
    float g = 0.f;
#pragma unroll
    for (int w = 0; w < kRW; ++w) g += red[w][lane];
    sm.part_g[(int64_t)blockIdx.x * Sp + s] = g;
  }
  const int pdf = p0 + lane;
  for (int sl = wave; sl < 64; sl += kRW) {
    const int sq = s0 + sl;
    if (sq < p.S && pdf < p.P) {
      const int64_t row = (int64_t)t * p.S + sq;
      float o = p.deriv_weight * tile[lane][sl] - p.l2_scale * p.y[row * p.y_stride + pdf];
      float *d = p.deriv + row * p.deriv_stride + pdf;
      if (ACCUM) o += *d;
      *d = o;
    }
  }
}

// ---- tied graphs: exp(y) applied per state, one row gather per arc and pass ---------------------------
//   forward : alpha_t(g) * asum_{t-1} = p(f(g)) * sum_in w * alpha'_{t-1}(src) + p(s(g)) * w_s(g) * alpha'_{t-1}(g)
//   backward: Y(g) = beta_{t+1}(g) * p_t(f(g));  beta'_t(h) * asum_t = sum_out w * Y(dst) + p_t(s(h)) * w_s(h) * beta_{t+1}(h)
//   gamma   : from per-state quantities (den_kernels.hip, tied path): self-loop occupation
//             w_s * beta_{t+1}(g) * p_t(s) * alpha'_t(g) / asum_t -> pdf s(g); forward-class occupation
//             beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart) -> pdf f(g), alpha_{t+1} being the stored un-dashed value.
__device__ __forceinline__ float big_row(const float *base, int idx, int Sp) {
  return idx >= 0 ? base[(int64_t)idx * Sp] : 0.f;
}

__global__ __launch_bounds__(kBT) void big_fwd_tied_kernel(const DenParams p, int t) {
  __shared__ float red[4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int Sp = p.big_Sp, s = blockIdx.y * 64 + lane;
  const BigSmall sm = big_small(p);
  const float *prev = p.alpha_hist + (int64_t)(t - 1) * p.H * Sp + s;
  float *cur = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t - 1) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_prev = sm.asum[(int64_t)(t - 1) * Sp + s];
  const float inv = 1.0f / asum_prev, cl_as = p.leaky * asum_prev;
  float part = 0.f;
  const int h0 = blockIdx.x * kStatesPerBlock + wave * (kStatesPerBlock / 4);
  // the per-state rows of the wave's four states are requested first: they do not depend on the arc sums,
  // and these kernels are bound by load latency at the occupancy they run at
  constexpr int KS = kStatesPerBlock / 4;
  float own[KS], ef[KS], es[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int h = min(h0 + k, p.H - 1);
    own[k] = prev[(int64_t)h * Sp];
    ef[k] = big_row(E, p.big.tf[h], Sp);
    es[k] = big_row(E, p.big.ts[h], Sp);
  }
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int h = h0 + k;
    if (h >= p.H) break;
    const int e0 = p.big.in_begin[h], e1 = p.big.in_begin[h + 1];
    const float F = big_arc_sum(p.big.in_arc, e0, e1, [&](const BigArc &r) {
      return r.w * (prev[(int64_t)r.a * Sp] + cl_as * r.pi);
    });
    const float a_self = own[k] + cl_as * p.pi[h];
    const float v = (ef[k] * F + es[k] * (p.big.tws[h] * a_self)) * inv;
    cur[(int64_t)h * Sp] = v;
    part += v;
  }
  red[wave][lane] = part;
  __syncthreads();
  if (wave == 0) sm.part_a[(int64_t)blockIdx.x * Sp + s] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Y(g) = beta_{t+1}(g) * p_t(f(g)) for every state
__global__ __launch_bounds__(kBT) void big_y_kernel(const DenParams p, int t) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int Sp = p.big_Sp, s = blockIdx.y * 64 + lane;
  const BigSmall sm = big_small(p);
  const float *Bprev = p.big_beta + (int64_t)((t + 1) & 1) * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t) + s;  // (all frames resident: big_exp_all_kernel)
  const float bs = sm.bsum[((t + 1) & 1) * Sp + s];
  const int h0 = blockIdx.x * kStatesPerBlock + wave * (kStatesPerBlock / 4);
  for (int k = 0; k < kStatesPerBlock / 4; ++k) {
    const int h = h0 + k;
    if (h >= p.H) break;
    p.big_y[(int64_t)h * Sp + s] = (Bprev[(int64_t)h * Sp] + bs) * big_row(E, p.big.tf[h], Sp);
  }
}

__global__ __launch_bounds__(kBT) void big_bwd_tied_kernel(const DenParams p, int t) {
  __shared__ float red[2][4][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int Sp = p.big_Sp, s = blockIdx.y * 64 + lane;
  const BigSmall sm = big_small(p);
  const float *Bprev = p.big_beta + (int64_t)((t + 1) & 1) * p.H * Sp + s;
  float *Bcur = p.big_beta + (int64_t)(t & 1) * p.H * Sp + s;
  const float *Y = p.big_y + s;
  const float *E = p.big_expy + p.big_exp_stride * (t) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s];
  const float *alpha = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  float part = 0.f, part_ab = 0.f;
  const int h0 = blockIdx.x * kStatesPerBlock + wave * (kStatesPerBlock / 4);
  constexpr int KS = kStatesPerBlock / 4;
  float own[KS], es[KS], al[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int h = min(h0 + k, p.H - 1);
    own[k] = Bprev[(int64_t)h * Sp];
    es[k] = big_row(E, p.big.ts[h], Sp);
    al[k] = t == 0 ? alpha[(int64_t)h * Sp] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < KS; ++k) {
    const int h = h0 + k;
    if (h >= p.H) break;
    const int e0 = p.big.out_begin[h], e1 = p.big.out_begin[h + 1];
    const float sum = big_arc_sum(p.big.out_arc, e0, e1, [&](const BigArc &r) { return r.w * Y[(int64_t)r.a * Sp]; });
    const float self = es[k] * p.big.tws[h] * (own[k] + bs);
    const float bp = (sum + self) * inv_as;
    Bcur[(int64_t)h * Sp] = bp;
    const float cpi = p.leaky * p.pi[h];
    part += cpi * bp;
    if (t == 0) part_ab += (al[k] + cpi * asum_t) * bp;
  }
  red[0][wave][lane] = part;
  red[1][wave][lane] = part_ab;
  __syncthreads();
  if (wave == 0) {
    sm.part_a[(int64_t)blockIdx.x * Sp + s] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    if (t == 0)
      sm.part_ab[(int64_t)blockIdx.x * Sp + s] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  }
}

template <bool ACCUM>
__global__ __launch_bounds__(kRT) void big_gamma_tied_kernel(const DenParams p, int t) {
  __shared__ float tile[64][65];
  __shared__ float red[kRW][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int Sp = p.big_Sp, s0 = blockIdx.y * 64, s = s0 + lane;
  const BigSmall sm = big_small(p);
  const float *alpha = p.alpha_hist + (int64_t)t * p.H * Sp + s;
  const float *alpha_up = p.alpha_hist + (int64_t)(t + 1) * p.H * Sp + s;  // un-dashed alpha_{t+1}
  const float *Bprev = p.big_beta + (int64_t)((t + 1) & 1) * p.H * Sp + s;
  const float *E = p.big_expy + p.big_exp_stride * (t) + s;  // (all frames resident: big_exp_all_kernel)
  const float asum_t = sm.asum[(int64_t)t * Sp + s];
  const float inv_as = 1.0f / asum_t, bs = sm.bsum[((t + 1) & 1) * Sp + s], cl_as = p.leaky * asum_t;
  const int p0 = blockIdx.x * kPdfsPerBlock;
  float part_g = 0.f;
  for (int k = 0; k < kPdfsPerBlock / kRW; ++k) {
    const int pl = wave * (kPdfsPerBlock / kRW) + k, pdf = p0 + pl;
    float gam = 0.f;
    if (pdf < p.P) {
      const int e0 = p.big.pdf_begin[pdf], e1 = p.big.pdf_begin[pdf + 1];
      // entries are self-contained (state, role, the state's self-loop pdf / weight / pi), four at a time
      // with all their row loads issued before the arithmetic
      for (int e = e0; e < e1; e += 4) {
        BigArc r[4];
        float b[4], a[4], up[4], esv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = p.big.pdf_arc[min(e + u, e1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          b[u] = Bprev[(int64_t)r[u].a * Sp];
          a[u] = alpha[(int64_t)r[u].a * Sp];
          up[u] = alpha_up[(int64_t)r[u].a * Sp];
          esv[u] = big_row(E, (r[u].b >> 1) - 1, Sp);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float beta = b[u] + bs;
          const float selfpart = esv[u] * r[u].w * (a[u] + cl_as * r[u].pi) * inv_as;
          const float occ = (r[u].b & 1) ? beta * selfpart : beta * fmaxf(up[u] - selfpart, 0.f);
          gam += e + u < e1 ? occ : 0.f;
        }
      }
    }
    tile[pl][lane] = gam;
    part_g += gam;
  }
  red[wave][lane] = part_g;
  __syncthreads();
  if (t == 0 && wave == 0) {
    float g = 0.f;
#pragma unroll
    for (int w = 0; w < kRW; ++w) g += red[w][lane];
    sm.part_g[(int64_t)blockIdx.x * Sp + s] = g;
  }
  const int pdf = p0 + lane;
  for (int sl = wave; sl < 64; sl += kRW) {
    const int sq = s0 + sl;
    if (sq < p.S && pdf < p.P) {
      const int64_t row = (int64_t)t * p.S + sq;
      float o = p.deriv_weight * tile[lane][sl] - p.l2_scale * p.y[row * p.y_stride + pdf];
      float *d = p.deriv + row * p.deriv_stride + pdf;
      if (ACCUM) o += *d;
      *d = o;
    }
  }
}

// bsum_t = leaky * sum_h pi(h) beta'_t(h); at t == 0 also the two checks of [K] BetaGeneralFrameDebug(0)
__global__ __launch_bounds__(kRT) void big_bsum_kernel(const DenParams p, int t) {
  __shared__ float red[kRW][64];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // wave index: scalar
  const int Sp = p.big_Sp, s = blockIdx.x * 64 + lane;
  const BigSmall sm = big_small(p);
  const int rows = (p.H + kStatesPerBlock - 1) / kStatesPerBlock;
  const float b = big_colsum(sm.part_a, rows, Sp, s, wave, red, lane);
  if (wave == 0) sm.bsum[(t & 1) * Sp + s] = b;
  if (t == 0) {
    const float ab = big_colsum(sm.part_ab, rows, Sp, s, wave, red, lane);
    const float gs = big_colsum(sm.part_g, (p.P + kPdfsPerBlock - 1) / kPdfsPerBlock, Sp, s, wave, red, lane);
    if (wave == 0 && s < p.S) {
      p.seq_ab[s] = ab;
      p.seq_gsum[s] = gs;
    }
  }
}

}  // namespace

// floats of p.big_small for this problem size (api.cpp sizes the workspace with it)
int64_t big_small_floats(int H, int P, int T, int Sp) {
  const int64_t hb = (H + kStatesPerBlock - 1) / kStatesPerBlock, pb = (P + kPdfsPerBlock - 1) / kPdfsPerBlock;
  return ((int64_t)(T + 1) + 2 + 3 + 2 * hb + 2 * pb) * Sp;
}

int launch_den_big(const DenParams &p, int accumulate, hipStream_t stream) {
  const int Sp = p.big_Sp, sg = Sp / 64;
  const dim3 blk(kBT);
  const dim3 g_exp((p.P + 63) / 64, sg), g_states((p.H + kStatesPerBlock - 1) / kStatesPerBlock, sg);
  const dim3 g_pdfs((p.P + kPdfsPerBlock - 1) / kPdfsPerBlock, sg);
  const int fill_blocks = (int)std::min<int64_t>(4096, ((int64_t)p.H * Sp + kBT - 1) / kBT);
  hipLaunchKernelGGL(big_alpha0_kernel, dim3(fill_blocks), blk, 0, stream, p);
  const bool tied = p.big.tied != 0;
  const bool exp_all = p.big_exp_stride != 0;
  if (exp_all) hipLaunchKernelGGL(big_exp_all_kernel, g_exp, blk, 0, stream, p);
  for (int t = 1; t <= p.T; ++t) {
    if (!exp_all) hipLaunchKernelGGL(big_exp_kernel<true>, g_exp, blk, 0, stream, p, t - 1);
    if (tied)
      hipLaunchKernelGGL(big_fwd_tied_kernel, g_states, blk, 0, stream, p, t);
    else
      hipLaunchKernelGGL(big_fwd_kernel, g_states, blk, 0, stream, p, t);
    hipLaunchKernelGGL(big_asum_kernel, dim3(sg), dim3(kRT), 0, stream, p, t);
  }
  hipLaunchKernelGGL(big_total_kernel, dim3((p.S + kBT - 1) / kBT), blk, 0, stream, p);
  if (p.deriv) {
    hipLaunchKernelGGL(big_beta_init_kernel, dim3(fill_blocks), blk, 0, stream, p);
    for (int t = p.T - 1; t >= 0; --t) {
      if (!exp_all) hipLaunchKernelGGL(big_exp_kernel<false>, g_exp, blk, 0, stream, p, t);
      if (tied) {
        hipLaunchKernelGGL(big_y_kernel, g_states, blk, 0, stream, p, t);
        hipLaunchKernelGGL(big_bwd_tied_kernel, g_states, blk, 0, stream, p, t);
        if (accumulate)
          hipLaunchKernelGGL(big_gamma_tied_kernel<true>, g_pdfs, dim3(kRT), 0, stream, p, t);
        else
          hipLaunchKernelGGL(big_gamma_tied_kernel<false>, g_pdfs, dim3(kRT), 0, stream, p, t);
      } else {
        hipLaunchKernelGGL(big_bwd_kernel, g_states, blk, 0, stream, p, t);
        if (accumulate)
          hipLaunchKernelGGL(big_gamma_kernel<true>, g_pdfs, dim3(kRT), 0, stream, p, t);
        else
          hipLaunchKernelGGL(big_gamma_kernel<false>, g_pdfs, dim3(kRT), 0, stream, p, t);
      }
      hipLaunchKernelGGL(big_bsum_kernel, dim3(sg), dim3(kRT), 0, stream, p, t);
    }
  }
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace tc
