// Fused denominator forward-backward for GENERAL graphs (arcs into a state may carry different pdfs, even after the state
// splitting of schedule_owner.cpp: make_work_graph) of at most 8192 states on gfx950 -- round 5's replacement of round 1's
// general kernel (den_kernels.hip, which keeps the graphs this one does not take) on the OWNER-COMPUTES schedules the
// tied kernels use (schedule_owner.cpp, `general`):
//   * states are permuted by degree and the thread that owns a state walks its arc list in both directions, so a row sum
//     never leaves its thread: no in-band ROW cells in the stream (one cell in nine at C3's degrees), no barrier between
//     the walk and the per-state pass (two barriers per frame instead of three), commits by ds_write_addtid;
//   * a cell is {w, position * 4 | pdf * 4 << 16} (8 bytes, [chunk][4 blocks][lane]): two LDS gathers per arc --
//     alpha' / beta of the other state and exp(y) of the arc's pdf -- which is what "general" costs against the tied
//     kernel's one; row ends are wave-uniform bits of mask words held in one register (den_tied_device.h: walk);
//   * buffer-descriptor addressing, the first RES chunks of every wave's stream resident in registers;
//   * gamma still takes one fixed-point LDS atomic per ARC in the backward walk (the occupation of an arc h -> g with pdf c
//     is alpha'_t(h) * w * p_t(c) * beta_{t+1}(g) / asum_t, and the arcs out of h carry different pdfs): alpha'_t of the
//     owned states waits in LDS as one value per row and lane, read when a row starts.
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), reached by the reference
// through src/my_lib_chain.cpp:129-131 for any den.fst (src/my_lib_example.cpp:129-134).
#include "den_tied_device.h"

namespace tc {

namespace {

// a chunk of 8 cells of one lane: {w0..w3}, {w4..w7}, {idx0..idx3}, {idx4..idx7}
struct Chunk8 {
  u4 wa, wb, ia, ib;
};
__device__ __forceinline__ void load_chunk8(Chunk8 &q, rsrc_t stream, uint32_t lane16, int chunk) {
  const uint32_t so = (uint32_t)chunk * (4 * 64 * 16);
  q.wa = bld4u(stream, lane16, so);
  q.wb = bld4u(stream, lane16, so + 1024);
  q.ia = bld4u(stream, lane16, so + 2048);
  q.ib = bld4u(stream, lane16, so + 3072);
}

// Row bookkeeping of the backward walk: the row sums as in the tied kernels, plus the occupation factor of the row in
// progress -- alpha'_t(own state) * scale, one value per row and lane at vocc + 256 * row.
struct GenRows {
  RowCommit rc;
  uint32_t vocc;   // this lane's slot of its wave's row 0 in the alpha' rows
  uint32_t occ_off;
  float occ_scale, occf;
  __device__ __forceinline__ void next_row() {
    occ_off += 256u;
    occf = ldsf(vocc + occ_off) * occ_scale;
  }
};

// acc(row) += w * SRC[position] * exp(y)[pdf] over one chunk; BWD: gamma(pdf) += the same term * the row's factor
template <uint32_t SRC, int HALF, bool BWD>
__device__ __forceinline__ void do_chunk8(const Chunk8 &q, uint32_t m, float &acc, GenRows &g, uint32_t aGM) {
  const uint32_t idx[8] = {q.ia.x, q.ia.y, q.ia.z, q.ia.w, q.ib.x, q.ib.y, q.ib.z, q.ib.w};
  const uint32_t w[8] = {q.wa.x, q.wa.y, q.wa.z, q.wa.w, q.wb.x, q.wb.y, q.wb.z, q.wb.w};
  uint32_t os[8], op[8];
  float a[8], pp[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    os[i] = lo16(idx[i]);
    op[i] = hi16(idx[i]);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = ldsf(SRC + os[i]);
    pp[i] = ldsf(op[i]);  // (exp(y) sits at LDS offset 0)
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float vf = (a[i] * __uint_as_float(w[i])) * pp[i];
    acc += vf;
    if (BWD) gamma_add_a(aGM + op[i], vf * g.occf);
    const int bit = (i & 1) ? 4 * HALF + i / 2 : 8 + 4 * HALF + i / 2;
    if (__builtin_expect((m >> bit) & 1u, 0)) {
      g.rc.commit(acc);
      acc = 0.f;
      if (BWD) g.next_row();
    }
  }
}

// one walk of a wave's stream: RES resident chunks, the rest through two buffers in ping-pong (den_tied_device.h: walk)
template <uint32_t SRC, int RES, bool BWD>
__device__ __forceinline__ void walk8(const Chunk8 (&res)[RES > 0 ? RES : 1], Chunk8 &qa, rsrc_t sbase, uint32_t lane16, int nchunks,
                                      uint32_t vmask, GenRows g, uint32_t aGM) {
  static_assert(RES % 2 == 0, "a mask word covers two chunks");
  auto mk = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)vmask, i); };
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < RES / 2; ++i) {
    const uint32_t m = mk(i);
    do_chunk8<SRC, 0, BWD>(res[2 * i], m, acc, g, aGM);
    do_chunk8<SRC, 1, BWD>(res[2 * i + 1], m, acc, g, aGM);
  }
  Chunk8 qb;
  int c = RES;
  for (; c + 2 < nchunks; c += 2) {
    const uint32_t m = mk(c >> 1);
    load_chunk8(qb, sbase, lane16, c + 1);
    do_chunk8<SRC, 0, BWD>(qa, m, acc, g, aGM);
    load_chunk8(qa, sbase, lane16, c + 2);
    do_chunk8<SRC, 1, BWD>(qb, m, acc, g, aGM);
  }
  if (c + 1 < nchunks) {
    const uint32_t m = mk(c >> 1);
    load_chunk8(qb, sbase, lane16, c + 1);
    do_chunk8<SRC, 0, BWD>(qa, m, acc, g, aGM);
    do_chunk8<SRC, 1, BWD>(qb, m, acc, g, aGM);
  } else if (c < nchunks) {
    do_chunk8<SRC, 0, BWD>(qa, mk(c >> 1), acc, g, aGM);
  }
}

#ifndef TC_GEN_RES
#define TC_GEN_RES 2
#endif

template <int PV, bool ACCUM, bool WANT_DERIV>
__global__ __launch_bounds__(kThreads) void den_general_owner_kernel(const DenParams p) {
  constexpr int JV = kJvSmall, RES = TC_GEN_RES;
  constexpr uint32_t kPB = 0u, kA0 = PV * 16u * kThreads;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = blockIdx.x;
  const int S = p.S, T = p.T, Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads), K = Hs / kThreads;
  const uint32_t own16 = 16u * tid, lane16 = 16u * lane;
  const uint32_t aACC = 4u * (uint32_t)p.L.off_acc, aGM = 4u * (uint32_t)p.L.off_g, aOCC = 4u * (uint32_t)p.L.off_al;
  const uint32_t aRed = 4u * (uint32_t)p.L.off_red;
  const AsumRow asums(p, s);
  const uint32_t vrow = aACC + 256u * (uint32_t)(K * wave) + 4u * lane, vocc = aOCC + 256u * (uint32_t)(K * wave) + 4u * lane;
  const uint32_t row_bytes = 4u * (uint32_t)p.P;
  const rsrc_t r_pi = make_rsrc(p.pi, 4u * (uint32_t)(Hs + 4));
  const float leaky = p.leaky;
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist = p.alpha_hist + (int64_t)s * Hs;
  auto hist_row = [&](int t) { return make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs); };
  auto stream_of = [&](const ScheduleDev &sc, rsrc_t &base, int &nch, uint32_t &vmask) {
    const int2 range = sc.wave_range[wave];
    nch = __builtin_amdgcn_readfirstlane(range.y) / kChunk;
    base = make_rsrc(reinterpret_cast<const char *>(sc.cells) + (int64_t)(__builtin_amdgcn_readfirstlane(range.x) / kChunk) * (4 * 64 * 16),
                     (uint32_t)(nch + 2) * (4 * 64 * 16));
    vmask = wave_masks(sc, wave, lane);
  };

  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  float part = 0.f;
  {
    f4 pi4[JV];
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      pi4[j] = bld4(r_pi, own16, j * kPlane);  // (beyond the graph's planes: zeros)
      part += hsum(pi4[j]);
    }
    const float asum0 = block_sum_a(part, aRed, wave, lane);
    const rsrc_t h0 = hist_row(0);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const f4 a = pi4[j] + (leaky * pi4[j]) * asum0;
        lds4_st(kA0 + own16 + j * kPlane, a);
        bst4(h0, own16 + j * kPlane, a);
      }
    if (tid == 0) asums.st(0, asum0);
    part = asum0;
  }
  float asum = part, inv_prev = __builtin_amdgcn_rcpf(asum), y2 = 0.f, part_tot = 0.f;
  {
    const rsrc_t yrow = make_rsrc(p.y + (int64_t)s * p.y_stride, row_bytes);
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      if (i0 < Ps) {
        const f4 yv = row_ld(yrow, own16 + v * kPlane, p.y_vec);
        y2 += hsum(yv * yv);
        lds4_st(kPB + 4u * i0, exp4(yv));
      }
    }
  }
  rsrc_t fbase;
  int fnch;
  uint32_t fmask;
  stream_of(p.fwd, fbase, fnch, fmask);
  const int ffx0 = p.fwd.nfix ? p.fwd.fix_begin[tid] : 0, ffx1 = p.fwd.nfix ? p.fwd.fix_begin[tid + 1] : 0;
  const uint32_t fsec = aACC + 256u * (uint32_t)(K * kWaves + p.fwd.extra_first[wave]);
  {
    Chunk8 fres[RES > 0 ? RES : 1];
#pragma unroll
    for (int i = 0; i < RES; ++i) load_chunk8(fres[i], fbase, lane16, i);
    // ---- forward frames t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
    for (int t = 1; t <= T; ++t) {
      Chunk8 q0;
      load_chunk8(q0, fbase, lane16, RES);
      __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready
      f4 yreg[PV];
      if (t < T) {
        const rsrc_t yrow = make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes);
#pragma unroll
        for (int v = 0; v < PV; ++v) yreg[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      }
      age_prio_on(wave);
      GenRows g{RowCommit{aACC + 256u * (uint32_t)(K * wave), fsec, K}, 0u, 0u, 0.f, 0.f};
      walk8<kA0, RES, false>(fres, q0, fbase, lane16, fnch, fmask, g, aGM);
      __builtin_amdgcn_s_setprio(0);
      for (int e = ffx0; e < ffx1; ++e) fold_row(p.fwd.fix[e], vrow, aACC, Hs, K);
      f4 v4[JV], cpi[JV];
      part = 0.f;
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        v4[j] = mk4(0.f);
        cpi[j] = bld4(r_pi, own16, j * kPlane);
        if (j < planes) {
          v4[j] = own_rows(vrow, j) * inv_prev;  // alpha_t(g) = sum over in-arcs of w p(pdf) alpha'_{t-1}(src) / asum_{t-1}
          part += hsum(v4[j]);
        }
      }
      asum = block_sum_a(part, aRed, wave, lane);  // every wave has finished its walk: the gather buffer may change
      const rsrc_t hist_t = hist_row(t);
      part_tot = 0.f;
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          const f4 a = v4[j] + (leaky * cpi[j]) * asum;
          lds4_st(kA0 + own16 + j * kPlane, a);
          bst4(hist_t, own16 + j * kPlane, a);
          part_tot += hsum(a);
        }
      if (t < T) {
#pragma unroll
        for (int v = 0; v < PV; ++v) {
          const int i0 = 4 * ((int)tid + kThreads * v);
          if (i0 < Ps) {
            y2 += hsum(yreg[v] * yreg[v]);
            lds4_st(kPB + 4u * i0, exp4(yreg[v]));
          }
        }
      }
      if (tid == 0) asums.st(t, asum);
      inv_prev = __builtin_amdgcn_rcpf(asum);
    }
  }
  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  const float tot = block_sum_a(part_tot, aRed + 4u * kWaves, wave, lane);
  {
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(asums.ld(t));
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d);  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
  }
  if (!WANT_DERIV) return;

  // ---- backward   ([K] BetaDashLastFrame, Beta(T), then BetaDashGeneralFrame(t) + Beta(t))
  // beta'_T(h) = 1/tot;  beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The gather buffer now holds beta_{t+1}; the
  // alpha' rows hold alpha'_t of the owned states, one value per row and lane.
  const float inv_tot = __builtin_amdgcn_rcpf(tot);
  part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) part += hsum(leaky * bld4(r_pi, own16, j * kPlane)) * inv_tot;
  float bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);  // also orders the reuse of the gather buffer
  rsrc_t bbase;
  int bnch;
  uint32_t bmask;
  stream_of(p.bwd, bbase, bnch, bmask);
  auto occ_store = [&](int j, f4 a) {  // alpha' of plane j's four states -> rows 4j..4j+3 of this lane
    ldsf_st(vocc + 256u * (4 * j), a.x);
    ldsf_st(vocc + 256u * (4 * j + 1), a.y);
    ldsf_st(vocc + 256u * (4 * j + 2), a.z);
    ldsf_st(vocc + 256u * (4 * j + 3), a.w);
  };
  // (behind a wave's last row the walk looks one row further for the padding cells' factor, which multiplies zeros: the last
  // wave then reads the four floats behind the rows, which nothing else writes)
  if (tid < 4) ldsf_st(aOCC + 4u * (uint32_t)Hs + 4u * tid, 0.f);
  f4 ycur[PV];
  {
    const rsrc_t hist_up = hist_row(T - 1);
    const rsrc_t yrow = make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const int h0 = 4 * ((int)tid + kThreads * j);
        const float b = inv_tot + bsum;
        lds4_st(kA0 + own16 + j * kPlane, f4{h0 < p.H ? b : 0.f, h0 + 1 < p.H ? b : 0.f, h0 + 2 < p.H ? b : 0.f, h0 + 3 < p.H ? b : 0.f});
        occ_store(j, bld4(hist_up, own16, j * kPlane));
      }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      ycur[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      if (i0 < Ps) {
        lds4_st(kPB + 4u * i0, exp4(ycur[v]));
        lds4_st(aGM + 4u * i0, mk4(0.f));
      }
    }
  }
  Chunk8 bres[RES > 0 ? RES : 1];
#pragma unroll
  for (int i = 0; i < RES; ++i) load_chunk8(bres[i], bbase, lane16, i);
  for (int t = T - 1; t >= 0; --t) {
    Chunk8 q0;
    load_chunk8(q0, bbase, lane16, RES);
    __syncthreads();  // beta_{t+1}, exp(y_t), alpha'_t rows ready; gamma zero
    const float inv_as = __builtin_amdgcn_rcpf(asums.ld(t));
    f4 ynext[PV], areg[JV];
    {
      // frame t-1's y row and alpha' of the owned states under the arc walk; at t == 0 frame 0 again
      const int tn = t > 0 ? t - 1 : 0;
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)tn * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) ynext[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      const rsrc_t hist_n = hist_row(tn);
#pragma unroll
      for (int j = 0; j < JV; ++j) areg[j] = bld4(hist_n, own16, j * kPlane);
    }
    age_prio_on(wave);
    {
      GenRows g{RowCommit{aACC + 256u * (uint32_t)(K * wave), aACC + 256u * (uint32_t)(K * kWaves), K}, vocc, 0u, inv_as * kGammaScale, 0.f};
      g.occf = ldsf(vocc) * g.occ_scale;
      walk8<kA0, RES, true>(bres, q0, bbase, lane16, bnch, bmask, g, aGM);
    }
    __builtin_amdgcn_s_setprio(0);
    f4 b4[JV];
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      b4[j] = mk4(0.f);
      const f4 cp = leaky * bld4(r_pi, own16, j * kPlane);
      if (j < planes) {
        b4[j] = own_rows(vrow, j) * inv_as;  // [K] * inv_arbitrary_scale: beta'_t
        part += hsum(cp * b4[j]);
        if (t == 0) part_ab += hsum(own_rows(vocc, j) * b4[j]);  // alpha'_0 . beta'_0
      }
    }
    bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    {
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) {
          const u4 gu = lds4u(aGM + 4u * i0);
          lds4_st(aGM + 4u * i0, mk4(0.f));
          const f4 gm = f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} * kGammaInvScale;
          if (t == 0) part_g += hsum(gm);
          f4 o = p.deriv_weight * gm - p.l2_scale * ycur[v];
          if (ACCUM) o += row_ld(drow, own16 + v * kPlane, p.d_vec);
          row_st(drow, own16 + v * kPlane, p.d_vec, o);
        }
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum_a(part_ab, aRed + 4u * kWaves, wave, lane);
      const float gsum = block_sum_a(part_g, aRed + 8u * kWaves, wave, lane);
      if (tid == 0) {
        p.seq_ab[s] = ab;
        p.seq_gsum[s] = gsum;
      }
      break;
    }
    // beta_t = beta'_t + leaky-sum: the next frame's gather source; alpha'_{t-1} rows; exp(y_{t-1})
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        lds4_st(kA0 + own16 + j * kPlane, b4[j] + bsum);
        occ_store(j, areg[j]);
      }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      ycur[v] = ynext[v];
      if (i0 < Ps) lds4_st(kPB + 4u * i0, exp4(ycur[v]));
    }
  }
}

template <int PV>
int launch_pv(const DenParams &p, int accumulate, size_t lds, hipStream_t stream) {
  void (*k)(const DenParams) = nullptr;
  if (!p.deriv)
    k = den_general_owner_kernel<PV, false, false>;
  else
    k = accumulate ? den_general_owner_kernel<PV, true, true> : den_general_owner_kernel<PV, false, true>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

int launch_den_general_owner(const DenParams &p, int accumulate, hipStream_t stream) {
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (!p.gen_owner || lds > (size_t)kLdsLimitBytes || p.L.JV != kJvSmall || !p.L.alpha_in_lds) return TC_ERR_UNSUPPORTED;
  if (p.L.PV == kPvSmall) return launch_pv<kPvSmall>(p, accumulate, lds, stream);
  if (p.L.PV == kPvMid) return launch_pv<kPvMid>(p, accumulate, lds, stream);
  if (p.L.PV == kPvLarge) return launch_pv<kPvLarge>(p, accumulate, lds, stream);
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
