// Fused denominator forward-backward for TIED (chain-structured) graphs on gfx950 -- the kernel the
// headline metric times.
//
// What it computes: [K] DenominatorComputation::Forward() + Backward() (chain-denominator.cc), reached by the
// reference through src/my_lib_chain.cpp:129-131.  Same mapping as den_kernels.hip (one workgroup = one
// sequence = one CU, the per-frame working set in LDS) with the factorisation tied graphs allow: every
// non-self-loop arc entering a state g carries one pdf f(g), so exp(y) leaves the arc sums --
//   forward   F(g)  = sum_{h->g} w * alpha'_t(h)            one LDS gather + one FMA per arc
//             alpha_{t+1}(g) * asum_t = p_t(f(g)) * F(g) + p_t(s(g)) * w_s(g) * alpha'_t(g)
//   backward  B(h)  = sum_{h->g} w * Y_t(g),  Y_t(g) = beta_{t+1}(g) * p_t(f(g))
//             gamma_t from per-state quantities only (two integer LDS atomics per STATE, none per arc)
// -- and the schedules are OWNER-COMPUTES (schedule_owner.cpp): the thread that owns a state walks its arc
// list in both directions, so row sums never cross threads.
//
// Where the time went before this file existed: every frame each CU pulled its 6-byte-per-arc cell
// stream (344 KB per direction at C3) from L2, 26 GB per launch chip-wide -- the L2 -> CU path, not HBM and
// not the LDS gathers, set the pace.  A wave's stream is the same every frame, so this kernel keeps the
// first RESF (forward phase) / RESB (backward phase) chunks of every wave's stream in REGISTERS for the
// whole phase and streams only the remainder.  That needs the registers: addresses are uniform base +
// 32-bit lane offset (SGPR-pair + one VGPR instead of 64-bit per-lane pointers), the 16-bit LDS offsets of
// resident cells are unpacked on the fly (the compiler would otherwise hoist the unpacking out of the frame
// loop and spend one more register per cell), and a row end costs one scalar bit test per cell.
#include "den_tied_device.h"

namespace tc {

namespace {

template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, bool WANT_DERIV, int RESF, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_kernel(const DenParams p) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x;
#ifdef TC_STAGGER
  // All workgroups run the same phases in lockstep, so every CU issues its history / derivative stores and its
  // y / history loads at the same instant: the memory system sees bursts.  A start offset of a fraction of a
  // frame per workgroup spreads them over the frame period.
  for (int i = 0; i < (int)(blockIdx.x % TC_STAGGER); ++i) __builtin_amdgcn_s_sleep(TC_STAGGER_SLEEP);
#endif
  const int H = p.H, P = p.P, S = p.S, T = p.T;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  // tied graphs are laid out in whole planes of 4096 positions (schedule_owner.cpp build_owner): which of its
  // JV float4s of states a thread really has is wave-uniform
  const int planes = Hs / (4 * kThreads);
  const int K = Hs / kThreads;  // own rows per lane
  const uint32_t own16 = 16u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u;                  // exp(y_t)
  constexpr uint32_t kA0 = PV * 16u * kThreads;  // alpha'_t (forward) / Y_t (backward): the gather source
  const uint32_t aACC = 4u * (uint32_t)p.L.off_acc;  // row sums [row][lane]: K per wave, then the secondary rows
  const uint32_t vrow = aACC + 256u * (uint32_t)(K * wave) + 4u * lane;  // this thread's slot of its wave's row 0
  const uint32_t aGM = 4u * (uint32_t)p.L.off_g;     // gamma_t, u32 fixed point (backward)
  const uint32_t aAL = 4u * (uint32_t)p.L.off_al;    // alpha'_{t+1} of the owned states (backward, roomy layout)
  const uint32_t aRed = 4u * (uint32_t)p.L.off_red;
  const uint32_t aAsum = 4u * (uint32_t)p.L.off_asum;  // alpha-sum of every frame

  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4), row_bytes = 4u * (uint32_t)P;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  // leaky * pi of the owned states is re-read with the other per-state tables every frame (an L2 hit)
  // rather than held in registers: the registers go to the resident stream
  const float leaky = p.leaky;
  f4 pi4[JV];  // (dead after frame 0)
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    pi4[j] = j < planes ? bld4(r_pi, own16, j * kPlane) : mk4(0.f);
    part += hsum(pi4[j]);
  }
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0))
  float asum = block_sum_a(part, aRed, wave, lane);
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist = p.alpha_hist + (int64_t)s * Hs;  // frame t lives at hist + t * hist_step
#pragma unroll
  for (int j = 0; j < JV; ++j)
    if (j < planes) {
      const f4 a = pi4[j] + (leaky * pi4[j]) * asum;
      lds4_st(kA0 + own16 + j * kPlane, a);
      bst4(make_rsrc(hist, 4u * Hs), own16 + j * kPlane, a);
    }
  float y2 = 0.f;
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
  if (tid == 0) ldsf_st(aAsum, asum);
  float inv_prev = __builtin_amdgcn_rcpf(asum);

  // ---- forward frames t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  {
    const int2 frange = p.fwd.wave_range[wave];
#ifdef TC_ABL_NOSTREAM
    const int fnch = RESF;
#else
    const int fnch = __builtin_amdgcn_readfirstlane(frange.y) / kChunk;
#endif
    // (the descriptor covers the wave's range and the look-ahead past it: the array ends with readable padding)
    const rsrc_t fbase = make_rsrc(reinterpret_cast<const char *>(p.fwd.cells) +
                                       (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                                   (uint32_t)(fnch + 2) * (3 * 64 * 16));
    const uint32_t fmask = wave_masks(p.fwd, wave, lane);
    const int ffx0 = p.fwd.nfix ? p.fwd.fix_begin[tid] : 0, ffx1 = p.fwd.nfix ? p.fwd.fix_begin[tid + 1] : 0;
    const RowCommit frc{aACC + 256u * (uint32_t)(K * wave), aACC + 256u * (uint32_t)(K * kWaves + p.fwd.extra_first[wave]), K};
    Chunk6 fres[RESF > 0 ? RESF : 1];
#pragma unroll
    for (int i = 0; i < RESF; ++i) load_chunk(fres[i], fbase, lane16, i);
    // The forward phase does not use the gamma / alpha'_{t+1} / second exp(y) regions: when they hold the two
    // per-state tables (C3: exactly), each thread parks its own entries there and the per-state pass reads
    // them at LDS latency instead of waiting for L2 every frame.
    const bool tabs_lds = (p.L.off_red - p.L.off_g) >= 2 * Hs;
    const uint32_t aFS = aGM, aWS = aGM + 4u * (uint32_t)Hs;
    if (tabs_lds) {
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          *reinterpret_cast<lds_u4 *>(aFS + own16 + j * kPlane) = bld4u(r_fs, own16, j * kPlane);
          lds4_st(aWS + own16 + j * kPlane, bld4(r_ws, own16, j * kPlane));
        }
    }
    // which resident chunk a wave issues its deferred stores after: one wave generation per chunk
    const int store_slot = RESF >= 4 ? wave >> 2 : RESF >= 2 ? wave >> 3 : 0;
    TC_STAMP_DECL
    for (int t = 1; t <= T; ++t) {
      TC_STAMP(0)
      Chunk6 q0;
      load_chunk(q0, fbase, lane16, RESF);  // (past a short stream: readable padding, never processed)
      __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready
      TC_STAMP(1)
      f4 yreg[PV];
      if (t < T) {  // y_t under the arc walk
        const rsrc_t yrow = make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes);
#pragma unroll
        for (int v = 0; v < PV; ++v) yreg[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      }
      // The history row of frame t-1 is stored from here, not from the end of frame t-1: a CU issues a 1 KB
      // store instruction only every ~60 cycles, so the 32 of a frame, issued back to back by 16 waves,
      // held the frame's tail for ~1.9k cycles (profiles/r02_phase_stamps_before_spread.txt).  Under the walk
      // the store path is idle: the four wave generations issue theirs after resident chunk 0, 1, 2, 3.
      age_prio_on(wave);
      bool stored = false;  // (nothing resident: the four wave generations store after streamed pair 0, 1, 2, 3)
      walk<kA0, RESF>(fres, q0, fbase, lane16, fnch, fmask, frc, [&](int i) {
        if (t > 1 && !stored && (RESF > 0 ? i == store_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
          stored = true;
          const rsrc_t hist_prev = make_rsrc(hist + (int64_t)(t - 1) * hist_step, 4u * Hs);
#ifndef TC_ABL_NOHIST
#pragma unroll
          for (int j = 0; j < JV; ++j)  // alpha'_{t-1} of the owned states: still in the gather buffer
            if (j < planes) bst4(hist_prev, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
#endif
        }
      } TC_WALK_PASS);
      __builtin_amdgcn_s_setprio(0);
      TC_STAMP(2)
      // graphs with hub states: the secondary rows of a state are walked by lanes of the wave that owns it
      // (schedule_owner.cpp), and a wave's LDS operations execute in order: no barrier
      for (int e = ffx0; e < ffx1; ++e) fold_row(p.fwd.fix[e], vrow, aACC, Hs, K);
      TC_STAMP(3)
      f4 v4[JV];
      part = 0.f;
      u4 fs[JV];
      f4 ws[JV], cpi[JV];
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          fs[j] = tabs_lds ? lds4u(aFS + own16 + j * kPlane) : bld4u(r_fs, own16, j * kPlane);
          ws[j] = tabs_lds ? lds4(aWS + own16 + j * kPlane) : bld4(r_ws, own16, j * kPlane);
          cpi[j] = bld4(r_pi, own16, j * kPlane);  // pi: first touched behind the reduction, which hides its L2 trip
        }
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        v4[j] = mk4(0.f);
        if (j < planes) {
          const f4 F = own_rows(vrow, j);
          const f4 al = lds4(kA0 + own16 + j * kPlane);  // alpha'_{t-1} of the owned states
          // alpha_t(g) * asum_{t-1} = p(f(g)) * sum_{h != g} w alpha'_{t-1}(h) + p(s(g)) * w_s * alpha'_{t-1}(g)
          const f4 a = f4{tied_alpha(kPB, fs[j].x, ws[j].x, F.x, al.x), tied_alpha(kPB, fs[j].y, ws[j].y, F.y, al.y),
                          tied_alpha(kPB, fs[j].z, ws[j].z, F.z, al.z), tied_alpha(kPB, fs[j].w, ws[j].w, F.w, al.w)};
          v4[j] = a * inv_prev;
          part += hsum(v4[j]);
        }
      }
      asum = block_sum_a(part, aRed, wave, lane);
      __builtin_amdgcn_sched_barrier(0);  // (keeps the multiply by leaky, and with it the wait for pi, down here)
      TC_STAMP(4)
      float part_tot = 0.f;
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) {
          const f4 a = v4[j] + (leaky * cpi[j]) * asum;
          lds4_st(kA0 + own16 + j * kPlane, a);
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
      if (tid == 0) ldsf_st(aAsum + 4u * t, asum);
      inv_prev = __builtin_amdgcn_rcpf(asum);
      if (t == T) part = part_tot;
    }
    {
      const rsrc_t hist_T = make_rsrc(hist + (int64_t)T * hist_step, 4u * Hs);
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (j < planes) bst4(hist_T, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
    }
    TC_STAMP(0)
    TC_STAMP_FLUSH(p.stamps)
  }
  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h)
  const float tot = block_sum_a(part, aRed + 4u * kWaves, wave, lane);
  {
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      // [K] log-prob = log(tot) + sum over t < T of log(alpha-sum_t): the scales divided out of frames 1..T
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(ldsf(aAsum + 4u * t));
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d);  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
  }
  if (p.fwd_norm) {  // two-CU form: the normalisers the combining pass needs (den_tied_split.hip)
    float *const fn = p.fwd_norm + (int64_t)s * (T + 2);
    for (int i = (int)tid; i <= T; i += kThreads) fn[i] = ldsf(aAsum + 4u * (uint32_t)i);
    if (tid == 0) fn[T + 1] = tot;
  }
  if (!WANT_DERIV) return;

  // ---- backward   ([K] BetaDashLastFrame, Beta(T), then BetaDashGeneralFrame(t) + Beta(t))
  // beta'_T(h) = 1/tot;  beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The LDS regions now hold
  // Y (gather source), the row sums, exp(y_t), exp(y_{t-1}), gamma_t and (roomy layout) alpha'_{t+1}.
  const float inv_tot = __builtin_amdgcn_rcpf(tot);
  part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j)
    if (j < planes) part += hsum(leaky * bld4(r_pi, own16, j * kPlane)) * inv_tot;
  float bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);  // also orders the reuse of the gather buffer
  f4 areg[JV];
  f4 ycur[PV], ynext[PV];
  f4 bown[JV];  // beta_{t+1} of the owned states (the LDS gather source holds Y instead)
  // two exp(y) buffers: frame t (self-loop terms of the per-state pass) and frame t-1 (written under the arc
  // walk, needed to form Y for the next frame); the tight layout has one and pays a barrier instead
  uint32_t pb_cur = kPB, pb_next = ALPHA_LDS ? 4u * (uint32_t)p.L.off_p2 : kPB;
  const int2 brange = p.bwd.wave_range[wave];
#ifdef TC_ABL_NOSTREAM
  const int bnch = RESB;
#else
  const int bnch = __builtin_amdgcn_readfirstlane(brange.y) / kChunk;
#endif
  const rsrc_t bbase = make_rsrc(reinterpret_cast<const char *>(p.bwd.cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(brange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(bnch + 2) * (3 * 64 * 16));
  const uint32_t bmask = wave_masks(p.bwd, wave, lane);
  const int bfx0 = p.bwd.nfix ? p.bwd.fix_begin[tid] : 0, bfx1 = p.bwd.nfix ? p.bwd.fix_begin[tid + 1] : 0;
  const RowCommit brc{aACC + 256u * (uint32_t)(K * wave), aACC + 256u * (uint32_t)(K * kWaves + p.bwd.extra_first[wave]), K};
  Chunk6 bres[RESB > 0 ? RESB : 1];
#pragma unroll
  for (int i = 0; i < RESB; ++i) load_chunk(bres[i], bbase, lane16, i);
  {
    const rsrc_t hist_up = make_rsrc(hist + (int64_t)T * hist_step, 4u * Hs);
    const rsrc_t yrow = make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes);
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      bown[j] = mk4(0.f);
      if (j < planes) {
        const int h0 = 4 * ((int)tid + kThreads * j);
        const float b = inv_tot + bsum;
        bown[j] = f4{h0 < H ? b : 0.f, h0 + 1 < H ? b : 0.f, h0 + 2 < H ? b : 0.f, h0 + 3 < H ? b : 0.f};
        if (ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, bld4(hist_up, own16, j * kPlane));
      }
    }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      ycur[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      if (i0 < Ps) {
        lds4_st(pb_cur + 4u * i0, exp4(ycur[v]));
        lds4_st(aGM + 4u * i0, mk4(0.f));
      }
    }
    __syncthreads();  // exp(y_{T-1}) complete: Y_{T-1}(g) = beta_T(g) * p_{T-1}(f(g))
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const u4 fs = bld4u(r_fs, own16, j * kPlane);
        lds4_st(kA0 + own16 + j * kPlane,
                f4{bown[j].x * ldsf(pb_cur + (fs.x & 0xffffu)), bown[j].y * ldsf(pb_cur + (fs.y & 0xffffu)),
                   bown[j].z * ldsf(pb_cur + (fs.z & 0xffffu)), bown[j].w * ldsf(pb_cur + (fs.w & 0xffffu))});
      }
  }
#ifdef TC_NO_BWD_DEFER
  constexpr bool kDeferDeriv = false;
#else
  constexpr bool kDeferDeriv = ALPHA_LDS;  // (the tight layout has one exp(y) buffer: nowhere to wait)
#endif
  const int bstore_slot = RESB >= 4 ? wave >> 2 : RESB >= 2 ? wave >> 3 : 0;
  TC_STAMP_DECL
  for (int t = T - 1; t >= 0; --t) {
    TC_STAMP(0)
    Chunk6 q0;
    load_chunk(q0, bbase, lane16, RESB);
    __syncthreads();  // Y, exp(y_t), alpha'_{t+1} ready; row sums and gamma zero
    TC_STAMP(1)
    const float asum_t = ldsf(aAsum + 4u * t);
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
    const rsrc_t hist_t = make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs);
    {
      // frame t-1's y row and alpha'_t of the owned states under the arc walk; at t == 0 y re-reads frame 0
      const int tn = t > 0 ? t - 1 : 0;
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)tn * S + s) * p.y_stride, row_bytes);
#pragma unroll
#ifdef TC_ABL_NOY2
      for (int v = 0; v < PV; ++v) ynext[v] = mk4(0.25f) * (float)t;  // (ablation: what the backward pass's second read of y costs)
#else
      for (int v = 0; v < PV; ++v) ynext[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
#endif
#pragma unroll
      for (int j = 0; j < JV; ++j) areg[j] = j < planes ? bld4(hist_t, own16, j * kPlane) : mk4(0.f);
    }
    // beta'_t(h) * asum_t = sum over out-arcs of w * Y(dst): the same walk as forward, no atomics
    // The derivative row of frame t+1 leaves from here, for the reason given at the forward walk (16 stores in
    // a row held the backward tail for ~1k cycles: profiles/r02_phase_stamps.txt, tail of waves 0-3 vs 12-15).
    // It waits, thread-private, in the exp(y) buffer that went dead with frame t+1's per-state pass and that
    // this thread overwrites only after its walk.
    age_prio_on(wave);
    bool dstored = false;
    walk<kA0, RESB>(bres, q0, bbase, lane16, bnch, bmask, brc, [&](int i) {
      if (kDeferDeriv && t < T - 1 && !dstored && (RESB > 0 ? i == bstore_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
        dstored = true;
        const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t + 1) * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
        for (int v = 0; v < PV; ++v)
          if (4 * ((int)tid + kThreads * v) < Ps) row_st(drow, own16 + v * kPlane, p.d_vec, lds4(pb_next + own16 + v * kPlane));
      }
    } TC_WALK_PASS);
    __builtin_amdgcn_s_setprio(0);
    if (ALPHA_LDS) {
      // exp(y_{t-1}) into the other buffer while the slower waves finish their walk
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) lds4_st(pb_next + 4u * i0, exp4(ynext[v]));
      }
    }
    TC_STAMP(2)
    for (int e = bfx0; e < bfx1; ++e) fold_row(p.bwd.fix[e], vrow, aACC, Hs, K);  // (no barrier: as in the forward pass)
    TC_STAMP(3)
    f4 b4[JV];
    uint32_t fpk[JV][2];  // forward-pdf offsets of the owned states, kept for the Y update below
    part = 0.f;
    float part_ab = 0.f, part_g = 0.f;
    const float asum_up = ldsf(aAsum + 4u * (t + 1));
    // With 16 states per thread the tables of plane j + 1 are requested before plane j is worked on: left at the top
    // of their own iteration the loads waited behind the LDS atomics of the plane before, one exposed L2 round trip
    // per plane (17 k cycles per frame for this pass on a 9681-state graph: profiles/r02_phase_stamps_r3.txt).  The
    // instantiations with resident stream chunks have no registers for that.
    constexpr bool kAhead = RESB == 0;
    u4 fs_n = u4{0u, 0u, 0u, 0u};
    f4 ws_n = mk4(0.f), cp_n = mk4(0.f), aup_n = mk4(0.f);
    auto request = [&](int j) {
      fs_n = bld4u(r_fs, own16, j * kPlane);
      ws_n = bld4(r_ws, own16, j * kPlane);
      cp_n = bld4(r_pi, own16, j * kPlane);
      // alpha'_{t+1}: parked by this thread (roomy layout) or re-read from the history (tight layout)
      if (!ALPHA_LDS) aup_n = bld4(make_rsrc(hist + (int64_t)(t + 1) * hist_step, 4u * Hs), own16, j * kPlane);
    };
    if (kAhead) request(0);
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      b4[j] = mk4(0.f);
      if (j < planes) {
        if (!kAhead) request(j);
        const u4 fs = fs_n;
        const f4 ws = ws_n;
        const f4 cp = leaky * cp_n;
        const f4 aup_g = aup_n;
        if (kAhead && j + 1 < planes) request(j + 1);
        f4 a = own_rows(vrow, j);
        const f4 al = areg[j];  // alpha'_t of the owned states
        const f4 aup = ALPHA_LDS ? lds4(aAL + own16 + j * kPlane) : aup_g;
        // Everything the arcs INTO an owned state g contribute to gamma_t, from per-state quantities:
        //   self-loop arc : occ_s = w_s * beta_{t+1}(g) * p_t(s(g)) * alpha'_t(g) / asum_t   -> gamma_t(s(g))
        //   forward class : sum_h w alpha'_t(h) p_t(f(g)) / asum_t = alpha_{t+1}(g) - selfpart, so
        //                   occ_f = beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart)               -> gamma_t(f(g))
        // with alpha_{t+1} = alpha'_{t+1} - leaky*pi*asum_{t+1} from the history (measured against float64 on peaky
        // outputs, profiles/r02_peaky.txt: keeping the un-dashed alpha in the history instead changes nothing).
        // The self-loop arc also adds vf_s = w_s * beta_{t+1}(g) * p_t(s(g)) to beta'_t(g) * asum_t.
        auto one = [&](uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx, float ax) {
          const float ps_ws = ldsf(pb_cur + (fsx >> 16)) * wsx;
          const float selfpart = ps_ws * alx * inv_as;  // self-loop part of alpha_{t+1}(g)
          const float bos = kGammaScale * bo;            // power-of-two scale: exact
          gamma_add_a(aGM + (fsx >> 16), bos * selfpart);
          gamma_add_a(aGM + (fsx & 0xffffu), bos * fmaxf((aupx - cpx * asum_up) - selfpart, 0.f));
          return fmaf(ps_ws, bo, ax);                    // vf_s into beta'_t(g) * asum_t
        };
        a.x = one(fs.x, ws.x, bown[j].x, al.x, aup.x, cp.x, a.x);
        a.y = one(fs.y, ws.y, bown[j].y, al.y, aup.y, cp.y, a.y);
        a.z = one(fs.z, ws.z, bown[j].z, al.z, aup.z, cp.z, a.z);
        a.w = one(fs.w, ws.w, bown[j].w, al.w, aup.w, cp.w, a.w);
        b4[j] = a * inv_as;  // [K] * inv_arbitrary_scale
        fpk[j][0] = (fs.x & 0xffffu) | (fs.y << 16);
        fpk[j][1] = (fs.z & 0xffffu) | (fs.w << 16);
        part += hsum(cp * b4[j]);
        if (t == 0) part_ab += hsum(al * b4[j]);
      }
    }
    bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    TC_STAMP(4)
    {
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) {
          const u4 gu = lds4u(aGM + 4u * i0);
          lds4_st(aGM + 4u * i0, mk4(0.f));
          const f4 g = f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} * kGammaInvScale;
          if (t == 0) part_g += hsum(g);
          f4 o = p.deriv_weight * g - p.l2_scale * ycur[v];
          if (ACCUM) o += row_ld(drow, own16 + v * kPlane, p.d_vec);
#ifndef TC_ABL_NODERIV
          if (kDeferDeriv && t > 0)
            lds4_st(pb_cur + 4u * i0, o);
          else
            row_st(drow, own16 + v * kPlane, p.d_vec, o);
#else
          if (o.x == 123.456f) row_st(drow, own16 + v * kPlane, p.d_vec, o);
#endif
        }
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum_a(part_ab, aRed + 4u * kWaves, wave, lane);
      const float gs = block_sum_a(part_g, aRed + 8u * kWaves, wave, lane);
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
    if (!ALPHA_LDS) {
      // tight layout: exp(y_{t-1}) overwrites exp(y_t) in place -- its readers (the per-state pass) are
      // behind the reduction's barrier -- and one more barrier publishes it to the Y update below
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) lds4_st(kPB + 4u * i0, exp4(ynext[v]));
      }
      __syncthreads();
    }
    // beta_t = beta'_t + leaky-sum; next frame's gather source Y_{t-1} = beta_t * p_{t-1}(f)
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const f4 b = b4[j] + bsum;
        bown[j] = b;
        const f4 yv = f4{b.x * ldsf(pb_next + (fpk[j][0] & 0xffffu)), b.y * ldsf(pb_next + (fpk[j][0] >> 16)),
                         b.z * ldsf(pb_next + (fpk[j][1] & 0xffffu)), b.w * ldsf(pb_next + (fpk[j][1] >> 16))};
        if constexpr (kA0 + (JV - 1) * kPlane < 65536u)
          lds4_st_at(own16, kA0 + j * kPlane, yv);
        else
          lds4_st(kA0 + own16 + j * kPlane, yv);
        if (ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, areg[j]);
      }
#pragma unroll
    for (int v = 0; v < PV; ++v) ycur[v] = ynext[v];
    const uint32_t tmp = pb_cur;
    pb_cur = pb_next;
    pb_next = tmp;
  }
}

// Resident chunks per phase for each instantiation: what the 128 registers of a 1024-thread workgroup hold
// next to the per-thread state (JV float4s of states, PV float4s of pdfs).
#ifndef TC_RESF
#define TC_RESF 4
#endif
#ifndef TC_RESB
#define TC_RESB 4
#endif
#ifndef TC_RES_JV3
#define TC_RES_JV3 4
#endif
#ifndef TC_RES_JV4
#define TC_RES_JV4 2
#endif
constexpr int res_fwd(int jv, int pv) { return jv > 3 ? (pv == 1 ? TC_RES_JV4 : 0) : jv == 3 ? (pv == 1 ? TC_RES_JV3 : 0) : pv == 1 ? TC_RESF : 2; }
constexpr int res_bwd(int jv, int pv) { return jv > 3 ? (pv == 1 ? TC_RES_JV4 : 0) : jv == 3 ? (pv == 1 ? TC_RES_JV3 : 0) : pv == 1 ? TC_RESB : 2; }

template <int JV, int PV>
int launch_jp(const DenParams &p, int accumulate, size_t lds_bytes, hipStream_t stream) {
  constexpr int RF = res_fwd(JV, PV), RB = res_bwd(JV, PV);
  const bool want = p.deriv != nullptr;
  const bool al = p.L.alpha_in_lds;
  void (*k)(const DenParams) = nullptr;
  if (!want)
    k = den_tied_kernel<JV, PV, true, false, false, RF, RB>;
  else if (accumulate)
    k = al ? den_tied_kernel<JV, PV, true, true, true, RF, RB> : den_tied_kernel<JV, PV, false, true, true, RF, RB>;
  else
    k = al ? den_tied_kernel<JV, PV, true, false, true, RF, RB> : den_tied_kernel<JV, PV, false, false, true, RF, RB>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds_bytes));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds_bytes, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

static_assert(TC_RESF <= kTiedMinChunks && TC_RESB <= kTiedMinChunks, "the resident prefix must exist in every wave's stream");

int launch_den_tied(const DenParams &p, int accumulate, hipStream_t stream) {
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (lds > (size_t)kLdsLimitBytes) return TC_ERR_UNSUPPORTED;
  const int JV = p.L.JV, PV = p.L.PV;
#define TC_DISPATCH(J, V) \
  if (JV == J && PV == V) return launch_jp<J, V>(p, accumulate, lds, stream);
#ifdef TC_ONLY_C3
  TC_DISPATCH(kJvSmall, kPvSmall)
#else
  TC_DISPATCH(kJvSmall, kPvSmall)
  TC_DISPATCH(kJvSmall, kPvMid)
  TC_DISPATCH(kJvSmall, kPvLarge)
  TC_DISPATCH(kJvMid, kPvSmall)
  TC_DISPATCH(kJvMid, kPvMid)
  TC_DISPATCH(kJvMid, kPvLarge)
  TC_DISPATCH(kJvLarge, kPvSmall)
  TC_DISPATCH(kJvLarge, kPvMid)
  TC_DISPATCH(kJvLarge, kPvLarge)
#endif
#undef TC_DISPATCH
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
