// Fused denominator forward-backward for GENERAL graphs on gfx950 (any arc labelling); chain-structured
// ("tied") graphs -- the benchmarked case -- take den_tied_kernel.hip, graphs beyond the on-chip layouts
// den_slab_kernel.hip.
//
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), the
// part of the reference's hot call (src/my_lib_chain.cpp:129-131) that the headline metric times.
//
// Mapping.  The frame recursion is serial and sequences never interact, so one workgroup (16 waves,
// one per CU) owns one sequence for all 2T frames and keeps that sequence's per-frame working set in
// LDS: exp(y_t), alpha'_t / beta_{t+1} (gather source), one accumulator per row and, in the backward
// half, gamma_t (u32 fixed point) and alpha'_t.  HBM sees each y row twice (forward, backward), each
// alpha' frame once out and once back, and each derivative row once.  The transition tables are
// streamed from L2 as a per-wave cell stream in a lane-major schedule (schedule_general.cpp): a lane walks
// one state's arc list, a wave instruction loads 64 lanes x 16 contiguous bytes, and the lanes gather
// alpha' and exp(y) from LDS (two gathers per arc, one gamma atomic per arc in the backward walk, in-band
// ROW cells, three barriers per frame).  There are no LDS float atomics (192 cycles per wave-instruction
// on gfx950): every row sum is committed with a plain store and gamma is integer fixed point.
//
// Numerics follow the Kaldi CPU arithmetic: linear domain, fp32, per-frame renormalisation by the
// alpha-sum of the previous frame ("arbitrary_scale"), leaky-HMM mixing, betas carrying 1/tot_prob.
#include "den_device.h"

namespace tc {

// ---- general graphs: walk over 8-byte cells with in-band ROW cells ------------------------------------
// One chunk of kChunk cells of a wave's stream.  All gathers of the chunk are issued before the first
// use (ROW cells carry valid offsets, so their gathers are harmless); only the commit of a finished
// row sits behind a branch, and that branch is scalar (the ROW flag is the same in all 64 lanes).
//   FWD: acc(row) += alpha'(src) * w * p(pdf)
//   BWD: vf = w * beta(dst) * p(pdf); acc(row) += vf; gamma(pdf) += vf * alpha'(row) / alpha_sum
// Folds the private slots of split rows into their state's accumulator; called by the thread that
// owns the state, after the barrier that follows the arc walk and before it reads the sum.
__device__ __forceinline__ void fold_split_rows(const ScheduleDev &sc, int fx0, int fx1, float *ACC) {
  for (int e = fx0; e < fx1; ++e) {
    const int2 f = sc.fix[e];
    ACC[f.x] += ACC[f.y];
  }
}

struct RowState {
  uint32_t row;
  float acc, occf;
};

template <bool BWD, bool ALPHA_LDS>
__device__ __forceinline__ void process_chunk(const uint2 (&q)[kChunk], RowState &rs,
                                              const float *__restrict__ SRC, const float *__restrict__ PB,
                                              float *__restrict__ ACC, float *__restrict__ GM,
                                              const float *__restrict__ AL, const float *__restrict__ hist_t,
                                              float inv_asum) {
  float a[kChunk], pp[kChunk];
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    a[u] = lds_at(SRC, q[u].y >> 16);
    pp[u] = lds_at(PB, q[u].y & 0xfffcu);
  }
#pragma unroll
  for (int u = 0; u < kChunk; ++u) {
    float w = __uint_as_float(q[u].x);
    // the ROW flag is in-band (bit 0 of the offset word, identical in all 64 lanes): testing the cell
    // itself avoids a separate mask stream and its dependent L2 load in every loop iteration
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(q[u].y) & 1u, 0)) {  // rare, scalar branch
      ACC[rs.row] = rs.acc;  // every row owns its slot: plain store, no atomic
      rs.acc = 0.f;
      rs.row = q[u].x & 0xffffu;  // ROW cell x = slot | state << 16
      if (BWD) {
        if (ALPHA_LDS)
          rs.occf = lds_at(AL, (q[u].x >> 14) & 0x3fffcu) * inv_asum * kGammaScale;
        else
          rs.occf = hist_t[q[u].x >> 16] * inv_asum * kGammaScale;
      }
      w = 0.f;  // the arc math below then adds nothing for this cell
    }
    if (!BWD) {
      rs.acc = fmaf(a[u] * w, pp[u], rs.acc);
    } else {
      const float vf = w * a[u] * pp[u];
      rs.acc += vf;
      gamma_add(GM, q[u].y & 0xfffcu, vf * rs.occf);  // a ROW cell adds 0 at its lane-aligned dummy offset
    }
  }
}

// Consumes this wave's cell stream with two register buffers in ping-pong: while one chunk is being
// processed the next one (4 KB per wave, 64 KB per CU) is in flight from L2.  The row mask of the NEXT
// iteration is fetched at the top of this one and only made scalar at the bottom, so its L2 round trip
// is never waited for on its own (vmcnt retires in order: waiting for it early would drain the
// prefetched chunk as well).
template <bool BWD, bool ALPHA_LDS>
__device__ __forceinline__ void walk_rows(const ScheduleDev &sc, int wave, int lane, int dummy_row,
                                          const float *__restrict__ SRC, const float *__restrict__ PB,
                                          float *__restrict__ ACC, float *__restrict__ GM,
                                          const float *__restrict__ AL, const float *__restrict__ hist_t,
                                          float inv_asum) {
  const int2 range = sc.wave_range[wave];
  const int first = __builtin_amdgcn_readfirstlane(range.x);   // multiple of kStreamUnroll
  const int ncells = __builtin_amdgcn_readfirstlane(range.y);  // multiple of kStreamUnroll
  // cells are stored [pair][lane][2]: one 16-byte load brings a lane's cells 2p and 2p+1
  const uint4 *__restrict__ r = reinterpret_cast<const uint4 *>(sc.cells) + (int64_t)(first / 2) * 64 + lane;
  uint2 qa[kChunk], qb[kChunk];
  auto load_chunk = [&](uint2 (&q)[kChunk], int cell0) {
#pragma unroll
    for (int u = 0; u < kChunk / 2; ++u) {
      const uint4 v = r[(cell0 / 2 + u) * 64];
      q[2 * u] = make_uint2(v.x, v.y);
      q[2 * u + 1] = make_uint2(v.z, v.w);
    }
  };
  load_chunk(qa, 0);
  RowState rs;
  rs.row = (uint32_t)dummy_row;  // until the stream's first ROW cell
  rs.acc = 0.f;
  rs.occf = 0.f;
  for (int c = 0; c < ncells; c += kStreamUnroll) {
    // Self-balancing: a wave's issue priority falls as it advances through its stream, so the waves
    // that lag (the arbiter otherwise favours the oldest) outrank the leaders and all 16 reach the
    // barrier together instead of leaving a tail with few active waves.
    if (4 * c < ncells)
      __builtin_amdgcn_s_setprio(3);
    else if (2 * c < ncells)
      __builtin_amdgcn_s_setprio(2);
    else if (4 * c < 3 * ncells)
      __builtin_amdgcn_s_setprio(1);
    else
      __builtin_amdgcn_s_setprio(0);
    // the stream is followed by kStreamUnroll readable padding cells, so these loads need no guard
    load_chunk(qb, c + kChunk);
    process_chunk<BWD, ALPHA_LDS>(qa, rs, SRC, PB, ACC, GM, AL, hist_t, inv_asum);
    load_chunk(qa, c + kStreamUnroll);
    process_chunk<BWD, ALPHA_LDS>(qb, rs, SRC, PB, ACC, GM, AL, hist_t, inv_asum);
  }
  __builtin_amdgcn_s_setprio(0);
}

template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, bool WANT_DERIV>
__global__ __launch_bounds__(kThreads) void den_fwd_bwd_kernel(const DenParams p) {
  extern __shared__ __align__(16) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
  const int H = p.H, P = p.P, S = p.S, T = p.T;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  float *const PB = lds;                          // exp(y_t), at LDS offset 0
  float *const A0 = lds + PV * 4 * kThreads;      // alpha'_t (forward) / beta_{t+1} (backward): gather source
  float *const ACC = lds + p.L.off_acc; // row accumulators: one per state, dummy at Hs, then split-row slots
  float *const GM = lds + p.L.off_g;    // gamma_t as u32 fixed point (backward only)
  float *const AL = lds + p.L.off_al;   // alpha'_t (backward only, when it fits)
  float *const red = lds + p.L.off_red;
  float *const asum_h = lds + p.L.off_asum;  // alpha-sum of every frame

  float4 cpi[JV];  // leaky * pi for the states this thread owns
  float4 pi4[JV];
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    const int h0 = 4 * (tid + kThreads * j);
    pi4[j] = h0 < Hs ? *reinterpret_cast<const float4 *>(p.pi + h0) : make_float4(0.f, 0.f, 0.f, 0.f);
    cpi[j] = make_float4(p.leaky * pi4[j].x, p.leaky * pi4[j].y, p.leaky * pi4[j].z, p.leaky * pi4[j].w);
    part += (pi4[j].x + pi4[j].y) + (pi4[j].z + pi4[j].w);
  }
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  float asum = block_sum(part, red, wave, lane);
  float *hist = p.alpha_hist + (int64_t)s * Hs;  // frame t lives at hist + t*S*Hs
  const int64_t hist_step = (int64_t)S * Hs;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    const int h0 = 4 * (tid + kThreads * j);
    if (h0 < Hs) {
      float4 a = make_float4(pi4[j].x + cpi[j].x * asum, pi4[j].y + cpi[j].y * asum, pi4[j].z + cpi[j].z * asum,
                             pi4[j].w + cpi[j].w * asum);
      *reinterpret_cast<float4 *>(A0 + h0) = a;
      *reinterpret_cast<float4 *>(hist + h0) = a;
      *reinterpret_cast<float4 *>(ACC + h0) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  if (tid < 4) ACC[Hs + tid] = 0.f;
  float y2 = 0.f;
  {
    const float *yrow = p.y + (int64_t)s * p.y_stride;
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      if (i0 < Ps) {
        float4 yv = load_row4(yrow, i0, P, p.y_vec);
        y2 += (yv.x * yv.x + yv.y * yv.y) + (yv.z * yv.z + yv.w * yv.w);
        *reinterpret_cast<float4 *>(PB + i0) =
            make_float4(exp_limited(yv.x), exp_limited(yv.y), exp_limited(yv.z), exp_limited(yv.w));
      }
    }
  }
  double logsum = 0.0;  // thread 0 only
  const int ffx0 = p.fwd.fix_begin[tid], ffx1 = p.fwd.fix_begin[tid + 1];
  if (tid == 0) asum_h[0] = asum;
  float inv_prev = 1.0f / asum;
  float asum_prev = asum;

  // ---- forward frames t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  TC_STAMP_DECL
  for (int t = 1; t <= T; ++t) {
    TC_STAMP(0)
    __syncthreads();  // A0, PB, ACC ready
    TC_STAMP(1)
    float4 yreg[PV];
    {
      // prefetch y_t under the arc walk (the last iteration re-reads row T-1: keeps yreg in registers)
      const float *yrow = p.y + ((int64_t)(t < T ? t : T - 1) * S + s) * p.y_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) yreg[v] = load_row4(yrow, 4 * (tid + kThreads * v), P, p.y_vec);
    }
    walk_rows<false, true>(p.fwd, wave, lane, Hs, A0, PB, ACC, nullptr, nullptr, nullptr, 0.f);
    TC_STAMP(2)
    __syncthreads();  // all row sums committed
    fold_split_rows(p.fwd, ffx0, ffx1, ACC);
    TC_STAMP(3)
    float4 v4[JV];
    part = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      v4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h0 < Hs) {
        const float4 a = *reinterpret_cast<float4 *>(ACC + h0);
        v4[j] = make_float4(a.x * inv_prev, a.y * inv_prev, a.z * inv_prev, a.w * inv_prev);
        part += (v4[j].x + v4[j].y) + (v4[j].z + v4[j].w);
      }
    }
    asum = block_sum(part, red, wave, lane);
    TC_STAMP(4)
    float *hist_t = hist + (int64_t)t * hist_step;
    float part_tot = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      if (h0 < Hs) {
        float4 a = make_float4(v4[j].x + cpi[j].x * asum, v4[j].y + cpi[j].y * asum, v4[j].z + cpi[j].z * asum,
                               v4[j].w + cpi[j].w * asum);
        *reinterpret_cast<float4 *>(A0 + h0) = a;
        *reinterpret_cast<float4 *>(hist_t + h0) = a;
        part_tot += (a.x + a.y) + (a.z + a.w);
      }
    }
    if (t < T) {
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps) {
          float4 yv = yreg[v];
          y2 += (yv.x * yv.x + yv.y * yv.y) + (yv.z * yv.z + yv.w * yv.w);
          *reinterpret_cast<float4 *>(PB + i0) =
              make_float4(exp_limited(yv.x), exp_limited(yv.y), exp_limited(yv.z), exp_limited(yv.w));
        }
      }
    }
    if (tid == 0) {
      asum_h[t] = asum;
      logsum += (double)__logf(asum_prev);  // log of the scale divided out of frame t
    }
    asum_prev = asum;
    inv_prev = 1.0f / asum;
    if (t == T) part = part_tot;
  }
  TC_STAMP(0)
  TC_STAMP_FLUSH(p.stamps)
  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  const float tot = block_sum(part, red + kWaves, wave, lane);
  {
    const double y2d = (double)block_sum(y2, red + 2 * kWaves, wave, lane);
    if (tid == 0) {
      p.seq_logprob[s] = logsum + (double)__logf(tot);
      p.seq_y2[s] = y2d;
    }
  }
  if (!WANT_DERIV) return;

  // ---- backward   ([K] BetaDashLastFrame, Beta(T), then BetaDashGeneralFrame(t) + Beta(t))
  // beta'_T(h) = 1/tot;  beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The same LDS regions now
  // hold B (= beta_{t+1}), BACC, P, GAMMA and alpha'_t.
  const float inv_tot = 1.0f / tot;
  part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) part += ((cpi[j].x + cpi[j].y) + (cpi[j].z + cpi[j].w)) * inv_tot;
  float bsum = block_sum(part, red + 3 * kWaves, wave, lane);  // also orders the A0 reuse below
  float4 areg[JV];
  float4 ycur[PV], ynext[PV];
  const int bfx0 = p.bwd.fix_begin[tid], bfx1 = p.bwd.fix_begin[tid + 1];
  {
    const float *hist_t = hist + (int64_t)(T - 1) * hist_step;
    const float *yrow = p.y + ((int64_t)(T - 1) * S + s) * p.y_stride;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      if (h0 < Hs) {
        *reinterpret_cast<float4 *>(A0 + h0) =
            make_float4(h0 < H ? inv_tot + bsum : 0.f, h0 + 1 < H ? inv_tot + bsum : 0.f,
                        h0 + 2 < H ? inv_tot + bsum : 0.f, h0 + 3 < H ? inv_tot + bsum : 0.f);
        *reinterpret_cast<float4 *>(ACC + h0) = make_float4(0.f, 0.f, 0.f, 0.f);  // states with no out-arcs
        if (ALPHA_LDS) *reinterpret_cast<float4 *>(AL + h0) = *reinterpret_cast<const float4 *>(hist_t + h0);
      }
    }
    if (ALPHA_LDS && tid < 4) AL[Hs + tid] = 0.f;
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      ycur[v] = load_row4(yrow, i0, P, p.y_vec);
      if (i0 < Ps) {
        *reinterpret_cast<float4 *>(PB + i0) = make_float4(exp_limited(ycur[v].x), exp_limited(ycur[v].y),
                                                           exp_limited(ycur[v].z), exp_limited(ycur[v].w));
        *reinterpret_cast<float4 *>(GM + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  TC_STAMP(0)
#ifdef TC_PHASE_STAMPS
  for (int i = 0; i < 8; ++i) st_acc[i] = 0;
#endif
  for (int t = T - 1; t >= 0; --t) {
    TC_STAMP(0)
    __syncthreads();  // B, PB, AL ready; BACC and GAMMA zero
    TC_STAMP(1)
    const float asum_t = asum_h[t];
    const float inv_as = 1.0f / asum_t;
    const float *hist_t = hist + (int64_t)t * hist_step;
    {
      // prefetch frame t-1 (y row and alpha') under the arc walk; at t == 0 it re-reads frame 0
      const int tn = t > 0 ? t - 1 : 0;
      const float *yrow = p.y + ((int64_t)tn * S + s) * p.y_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) ynext[v] = load_row4(yrow, 4 * (tid + kThreads * v), P, p.y_vec);
      if (ALPHA_LDS) {  // alpha'_{t-1} for the next frame's LDS copy
        const float *hist_n = hist + (int64_t)tn * hist_step;
#pragma unroll
        for (int j = 0; j < JV; ++j) {
          const int h0 = 4 * (tid + kThreads * j);
          areg[j] = h0 < Hs ? *reinterpret_cast<const float4 *>(hist_n + h0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    walk_rows<true, ALPHA_LDS>(p.bwd, wave, lane, Hs, A0, PB, ACC, GM, AL, hist_t, inv_as);
    TC_STAMP(2)
    __syncthreads();  // beta' sums and gamma committed
    fold_split_rows(p.bwd, bfx0, bfx1, ACC);
    TC_STAMP(3)
    float4 b4[JV];
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      b4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (h0 < Hs) {
        const float4 a = *reinterpret_cast<float4 *>(ACC + h0);
        float4 al = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t == 0)
          al = ALPHA_LDS ? *reinterpret_cast<float4 *>(AL + h0) : *reinterpret_cast<const float4 *>(hist_t + h0);
        b4[j] = make_float4(a.x * inv_as, a.y * inv_as, a.z * inv_as, a.w * inv_as);  // [K] * inv_arbitrary_scale
        part += (cpi[j].x * b4[j].x + cpi[j].y * b4[j].y) + (cpi[j].z * b4[j].z + cpi[j].w * b4[j].w);
        if (t == 0) part_ab += (al.x * b4[j].x + al.y * b4[j].y) + (al.z * b4[j].z + al.w * b4[j].w);
      }
    }
    bsum = block_sum(part, red, wave, lane);
    TC_STAMP(4)
    {
      float *drow = p.deriv + ((int64_t)t * S + s) * p.deriv_stride;
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * (tid + kThreads * v);
        if (i0 < Ps) {
          const uint4 gu = *reinterpret_cast<uint4 *>(GM + i0);
          *reinterpret_cast<float4 *>(GM + i0) = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 g = make_float4((float)gu.x * kGammaInvScale, (float)gu.y * kGammaInvScale,
                                       (float)gu.z * kGammaInvScale, (float)gu.w * kGammaInvScale);
          if (t == 0) part_g += (g.x + g.y) + (g.z + g.w);
          float4 o = make_float4(p.deriv_weight * g.x - p.l2_scale * ycur[v].x,
                                 p.deriv_weight * g.y - p.l2_scale * ycur[v].y,
                                 p.deriv_weight * g.z - p.l2_scale * ycur[v].z,
                                 p.deriv_weight * g.w - p.l2_scale * ycur[v].w);
          if (ACCUM) {
            float4 old = load_row4(drow, i0, P, p.d_vec);
            o = make_float4(old.x + o.x, old.y + o.y, old.z + o.z, old.w + o.w);
          }
          store_row4(drow, i0, P, p.d_vec, o);
        }
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum(part_ab, red + kWaves, wave, lane);
      const float gs = block_sum(part_g, red + 2 * kWaves, wave, lane);
      if (tid == 0) {
        p.seq_ab[s] = ab;
        p.seq_gsum[s] = gs;
      }
#ifdef TC_PHASE_STAMPS
      if (blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) p.stamps[128 + wave * 8 + i] = st_acc[i];
#endif
      break;
    }
    // beta_t = beta'_t + leaky-sum: the next frame's gather source
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      const int h0 = 4 * (tid + kThreads * j);
      if (h0 < Hs) {
        *reinterpret_cast<float4 *>(A0 + h0) = make_float4(b4[j].x + bsum, b4[j].y + bsum, b4[j].z + bsum, b4[j].w + bsum);
        if (ALPHA_LDS) *reinterpret_cast<float4 *>(AL + h0) = areg[j];
      }
    }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * (tid + kThreads * v);
      ycur[v] = ynext[v];
      if (i0 < Ps)
        *reinterpret_cast<float4 *>(PB + i0) = make_float4(exp_limited(ycur[v].x), exp_limited(ycur[v].y),
                                                           exp_limited(ycur[v].z), exp_limited(ycur[v].w));
    }
  }
}

template <int JV, int PV>
static int launch_jp(const DenParams &p, int accumulate, size_t lds_bytes, hipStream_t stream) {
  const bool want = p.deriv != nullptr;
  const bool al = p.L.alpha_in_lds;
  void (*k)(const DenParams) = nullptr;
  if (!want)
    k = den_fwd_bwd_kernel<JV, PV, true, false, false>;
  else if (accumulate)
    k = al ? den_fwd_bwd_kernel<JV, PV, true, true, true> : den_fwd_bwd_kernel<JV, PV, false, true, true>;
  else
    k = al ? den_fwd_bwd_kernel<JV, PV, true, false, true> : den_fwd_bwd_kernel<JV, PV, false, false, true>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds_bytes));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds_bytes, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// Tied graph, batch of at most half the CUs, workspace with room for the second history: forward and backward
// recursion side by side on two CUs per sequence, then the combining pass.
static int launch_den_tied_split(const DenParams &p, int accumulate, hipStream_t stream, SideStreams *c) {
  TC_HIP_CHECK(hipEventRecord(c->fork, stream));
  TC_HIP_CHECK(hipStreamWaitEvent(c->den_side, c->fork, 0));
  DenParams pf = p;
  pf.deriv = nullptr;  // forward only
  int rc = launch_den_tied(pf, 0, stream);
  if (rc == TC_OK) rc = launch_den_tied_backward_only(p, c->den_side);
  // Join whatever reached the side stream, also on a failure: the caller sees an error and may free or reuse the
  // workspace, which a kernel on the side stream could still be writing.
  hipError_t e = hipEventRecord(c->join, c->den_side);
  if (e == hipSuccess) e = hipStreamWaitEvent(stream, c->join, 0);
  if (rc != TC_OK) return rc;
  TC_HIP_CHECK(e);
  return launch_den_tied_combine(p, accumulate, c->num_cus, stream);
}

static bool split_wanted(const DenParams &p) {
  return p.tied_fs && !p.big.in.rows && p.deriv && p.beta_hist && split_bwd_fits(p.L, p.T) && !debug_flag(kDbgNoPhaseSplit);
}

// Two CUs per sequence that meet in the middle (den_tied_mitm.hip) instead of two pure recursions and a combining pass.
// Its time hardly depends on the batch (0.64 - 0.72 ms from 1 to 128 sequences of the C2 graph) while the combining
// pass costs 2.7 - 9 us per sequence, so the forms cross: between 16 and 40 sequences for graphs of the C2 / C3 / R1 class (8 states
// per thread, up to 4096 pdfs: four resident chunks in both roles), near 40 with more pdfs (C5), near 64 with 12 or 16
// states per thread (R3, X1) -- profiles/r03_configs.txt.
static bool mitm_wanted(const DenParams &p) {
  if (!p.mitm_sync || !p.bwd_norm || debug_flag(kDbgNoMitm) || !mitm_fits(p.L, p.T)) return false;
  const int from = p.L.JV == kJvSmall ? (p.L.PV == kPvSmall ? 32 : 48) : 64;
  return debug_flag(kDbgForceMitm) || p.S >= from;
}

// Two sequences per workgroup, the pair's two recursions on two CUs (den_tied_pair.hip): batches the two-CU form of one
// sequence does not cover.
static bool pair_wanted(const DenParams &p, int num_cus) {
  if (!p.tied_fs || p.big.in.rows || !p.deriv || !p.pair_norm || !p.pair_sync || !p.fwd.cells_pair) return false;
  if (debug_flag(kDbgNoPair) || !pair_fits(p.L, p.pair_extra_slots, p.T)) return false;
  if (debug_flag(kDbgForcePair)) return true;
  // Which of the two is faster depends on the graph (the two-sequence kernel shares the walk between its sequences and
  // wins from about 11 arcs per state on; at C3's 8 the fused kernel is ahead): measured once per graph and device.
  // Batches of up to one sequence per two CUs are the two-CU form's.
  return p.pair_choice > 0 && 2 * p.S > num_cus;
}

// plane-wise kernel: two workgroups per sequence (meeting in the middle) while they fit the chip
static bool planes_two_cus(const DenParams &p, int num_cus) {
  return p.L.planewise && planes_mitm_fits(p) && 2 * p.S <= num_cus && !debug_flag(kDbgNoPhaseSplit) && !debug_flag(kDbgNoMitm);
}

int den_cus_used(const DenParams &p, int num_cus) {
  if (p.big.in.rows) return num_cus;
  if (p.L.planewise) return planes_two_cus(p, num_cus) ? 2 * p.S : (p.S < num_cus ? p.S : num_cus);
  if (pair_wanted(p, num_cus)) return std::min(num_cus, 2 * ((p.S + 1) / 2));
  if (split_wanted(p) && 2 * p.S <= num_cus) return 2 * p.S;
  return p.S < num_cus ? p.S : num_cus;
}

// The kernels built on den_tied_frames.h's frames with gamma (fused, meet-in-the-middle) write xent_zero's rows next to
// the derivative's; mirrors launch_den_mode's choice below.
bool den_zeroes_xent(const DenParams &p, int num_cus) {
  if (p.big.in.rows || !p.tied_fs || !p.deriv) return false;
  if ((size_t)layout_lds_bytes(p.L, p.T) > (size_t)kLdsLimitBytes) return false;
  if (p.L.planewise) return true;  // (both forms of den_tied_planes.hip write the rows)
  if (pair_wanted(p, num_cus)) return false;
  if (split_wanted(p) && 2 * p.S <= num_cus) return mitm_wanted(p);
  return true;
}

// accumulate != 0 selects Kaldi's "deriv += deriv_weight * gamma" form
int launch_den_mode(const DenParams &p, int accumulate, hipStream_t stream) {
  count_launch(kCntDen);
  if (p.deriv) count_launch(kCntDenBackward);
  if (!p.big.in.rows && p.L.asum_global) count_launch(kCntDenAsumGlobal);
  if (p.big.in.rows) return launch_den_big(p, accumulate, stream);  // graph beyond the on-chip layout
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (lds > (size_t)kLdsLimitBytes) return TC_ERR_UNSUPPORTED;
  const int JV = p.L.JV, PV = p.L.PV;
  if (p.tied_fs != nullptr) {
    SideStreams *c = nullptr;
    const int src = side_streams(stream, &c);
    if (src != TC_OK) return src;
    if (p.L.planewise) {  // den_tied_planes.hip
      if (planes_two_cus(p, c->num_cus)) return launch_den_tied_planes_mitm(p, accumulate, stream);
      return launch_den_tied_planes(p, accumulate, stream);
    }
    if (pair_wanted(p, c->num_cus)) return launch_den_tied_pair(p, p.pair_extra_slots, accumulate, stream);
    if (split_wanted(p)) {
      if (2 * p.S <= c->num_cus) {
        if (mitm_wanted(p)) return launch_den_tied_mitm(p, p.mitm_sync, accumulate, stream);
        std::lock_guard<std::recursive_mutex> lock(c->enqueue);
        return launch_den_tied_split(p, accumulate, stream, c);
      }
    }
    DenParams pq = p;
    pq.fwd_norm = nullptr;
    return launch_den_tied(pq, accumulate, stream);  // den_tied_kernel.hip
  }
  if (p.gen_owner) return launch_den_general_owner(p, accumulate, stream);  // den_general_owner.hip
#define TC_DISPATCH(J, V) \
  if (JV == J && PV == V) return launch_jp<J, V>(p, accumulate, lds, stream);
  TC_DISPATCH(kJvSmall, kPvSmall)
  TC_DISPATCH(kJvSmall, kPvMid)
  TC_DISPATCH(kJvSmall, kPvLarge)
  TC_DISPATCH(kJvLarge, kPvSmall)
  TC_DISPATCH(kJvLarge, kPvMid)
  TC_DISPATCH(kJvLarge, kPvLarge)
#undef TC_DISPATCH
  return TC_ERR_UNSUPPORTED;
}

int launch_den(const DenParams &p, hipStream_t stream) { return launch_den_mode(p, 0, stream); }

}  // namespace tc
