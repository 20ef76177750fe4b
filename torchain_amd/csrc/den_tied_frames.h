// The frames of a tied (chain-structured) graph's recursions, stated ONCE for the kernels that give a sequence a whole
// workgroup: den_tied_kernel.hip (fused: forward phase, then backward phase with gamma) and den_tied_mitm.hip (two
// workgroups per sequence that meet in the middle: role F = the forward frame, from the meeting frame on with gamma;
// role B = the backward frame, pure with normalisers of its own down to the meeting frame, then the fused kernel's).
//
//   forward frame   F(g) = sum_{h->g} w * alpha'_{t-1}(h)      one LDS gather + one FMA per arc (the walk)
//                   alpha_t(g) * asum_{t-1} = p(f(g)) * F(g) + p(s(g)) * w_s(g) * alpha'_{t-1}(g)
//                   GAMMA (role F behind the meeting frame): the two parts times the stored B_t(g) are the occupations in
//                   frame t-1 of the forward-class arcs into g and of its self-loop
//   backward frame  U(h) = sum_{h->g} w * Y_t(g),  Y_t(g) = beta_{t+1}(g) * p_t(f(g)),  + p_t(s(h)) w_s(h) beta_{t+1}(h)
//                   gamma_t from per-state quantities (two integer LDS atomics per STATE, none per arc);
//                   PURE (role B above the meeting frame, B'_T = 1): no gamma, normaliser n_t = sum_h U(h) / H
//
// A host kernel builds one TiedSeq object per workgroup (all members live in registers or are compile-time constants:
// every method is inlined into the kernel) and calls the phases it needs; what a host does not call costs nothing.
// [K] = kaldi chain-denominator.cc, reached by the reference through src/my_lib_chain.cpp:129-131.
#pragma once

#include <type_traits>

#include "den_tied_device.h"

namespace tc {

// two block sums behind one barrier
__device__ __forceinline__ void block_sum2(float &v1, float &v2, uint32_t red, int wave, uint32_t lane) {
  v1 = wave_sum(v1);
  v2 = wave_sum(v2);
  if (lane == 0) {
    ldsf_st(red + 4u * (uint32_t)wave, v1);
    ldsf_st(red + 4u * (uint32_t)(kWaves + wave), v2);
  }
  __syncthreads();
  float t1 = ldsf(red + 4u * (lane & 15u)), t2 = ldsf(red + 4u * (kWaves + (lane & 15u)));
  t1 = dpp_add<0xB1>(t1);
  t2 = dpp_add<0xB1>(t2);
  t1 = dpp_add<0x4E>(t1);
  t2 = dpp_add<0x4E>(t2);
  t1 = dpp_add<0x124>(t1);
  t2 = dpp_add<0x124>(t2);
  v1 = dpp_add<0x128>(t1);
  v2 = dpp_add<0x128>(t2);
}

// a scalar another CU wrote (vector load: the scalar cache is not covered by an acquire)
__device__ __forceinline__ float vload_f32(const float *ptr) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(ptr, 4u), 0, 0, 0));
}

// ---- the per-state formulas of a tied frame, stated ONCE for every frame driver (TiedSeq below; the plane-wise kernel
// den_tied_planes.hip).  pb: byte address of exp(y_t); aGM: of gamma_t (u32 fixed point).
// forward: alpha_t(g) * asum_{t-1} = p(f(g)) * sum_{h != g} w alpha'_{t-1}(h) + p(s(g)) * w_s * alpha'_{t-1}(g); with
// B_t(g) the two parts are the occupations in frame t-1 of the forward-class arcs into g and of its self-loop
template <bool GAMMA>
__device__ __forceinline__ float tied_fwd_state(uint32_t pb, uint32_t aGM, float inv_prev, uint32_t fsx, float wsx, float Fx, float alx,
                                                float bx, float gs, float &dpart) {
  const float pf = ldsf(pb + (fsx & 0xffffu)), ps = ldsf(pb + (fsx >> 16));
  const float sp = ps * (wsx * alx);
  const float a = fmaf(pf, Fx, sp) * inv_prev;
  if constexpr (GAMMA) {
    const float g = gs * bx, spn = sp * inv_prev;
    gamma_add_a(aGM + (fsx >> 16), g * spn);
    gamma_add_a(aGM + (fsx & 0xffffu), g * fmaxf(a - spn, 0.f));
    dpart = fmaf(a, bx, dpart);
  }
  return a;
}
// backward: everything the arcs INTO an owned state g contribute to gamma_t, from per-state quantities:
//   self-loop arc : occ_s = w_s * beta_{t+1}(g) * p_t(s(g)) * alpha'_t(g) / asum_t   -> gamma_t(s(g))
//   forward class : sum_h w alpha'_t(h) p_t(f(g)) / asum_t = alpha_{t+1}(g) - selfpart, so
//                   occ_f = beta_{t+1}(g) * (alpha_{t+1}(g) - selfpart)               -> gamma_t(f(g))
// with alpha_{t+1} = alpha'_{t+1} - leaky*pi*asum_{t+1} from the history (measured against float64 on peaky
// outputs, profiles/r02_peaky.txt: keeping the un-dashed alpha in the history instead changes nothing; the plane-wise
// kernel does keep it un-dashed and passes asum_up = 0).
// The self-loop arc also adds vf_s = w_s * beta_{t+1}(g) * p_t(s(g)) to beta'_t(g) * asum_t (PURE: to U_t(g)).
template <bool PURE>
__device__ __forceinline__ float tied_bwd_state(uint32_t pb, uint32_t aGM, uint32_t fsx, float wsx, float bo, float alx, float aupx,
                                                float cpx, float ax, float inv_as, float asum_up) {
  const float ps_ws = ldsf(pb + (fsx >> 16)) * wsx;
  if constexpr (!PURE) {
    const float selfpart = ps_ws * alx * inv_as;  // self-loop part of alpha_{t+1}(g)
    const float bos = kGammaScale * bo;            // power-of-two scale: exact
    gamma_add_a(aGM + (fsx >> 16), bos * selfpart);
#ifndef TC_ABL_NOFADD  /* (ablation: what the forward-class half of the gamma adds costs) */
    gamma_add_a(aGM + (fsx & 0xffffu), bos * fmaxf((aupx - cpx * asum_up) - selfpart, 0.f));
#endif
  }
  return fmaf(ps_ws, bo, ax);
}

// MITM = false: the fused kernel (per-state tables parked in LDS during the forward phase, phase stamps of the
// diagnostic builds, normalisers to the workspace after the phase); true: the two roles of den_tied_mitm.hip
// (gamma region live from frame 0, normalisers to the workspace frame by frame, the B history).
template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, int RESF, int RESB, bool MITM>
struct TiedSeq {
  const DenParams &p;
  const uint32_t tid, lane;
  const int wave, s;
  const int H, P, S, T, Hs, Ps;
  // tied graphs are laid out in whole planes of 4096 positions (schedule_owner.cpp build_owner): which of its
  // JV float4s of states a thread really has is wave-uniform
  const int planes, K;  // K: own rows per lane
  const uint32_t own16, lane16;
  static constexpr uint32_t kPB = 0u;                   // exp(y_t)
  static constexpr uint32_t kA0 = PV * 16u * kThreads;  // alpha'_t (forward) / Y_t (backward): the gather source
  const uint32_t aACC;   // row sums [row][lane]: K per wave, then the secondary rows
  const uint32_t vrow;   // this thread's slot of its wave's row 0
  const uint32_t aGM;    // gamma_t, u32 fixed point
  const uint32_t aAL;    // alpha'_{t+1} of the owned states (backward, roomy layout)
  const uint32_t aRed, aAsum;  // reduction scratch; alpha-sum of every frame
  const uint32_t tab_bytes, row_bytes;
  // leaky * pi of the owned states is re-read with the other per-state tables every frame (an L2 hit)
  // rather than held in registers: the registers go to the resident stream
  const rsrc_t r_pi, r_fs, r_ws;
  const float leaky;
  const int64_t hist_step;
  float *const hist;   // alpha' history: frame t lives at hist + t * hist_step
  // two-workgroup form only
  float *const fn;     // [T + 2] asum_0..T, tot (role F writes, role B reads)
  float *const bn;     // [T + 1] role B's normalisers
  float *const bhist;  // B history
  const int M;         // meeting frame

  // Running values.  Wave-uniform members first, per-lane members behind them, and no default initialisers: the
  // compiler turns adjacent members that are set or copied together into one vector value, and a uniform member (a
  // chunk count, a row cursor's start) that shares a vector with a per-lane one is no longer uniform to it -- the
  // walk's scalar loop control then becomes vector code.
  // ---- uniform
  int fnch, store_slot, bnch, bstore_slot;
  uint32_t fsec, bsec;  // byte address of the wave's first secondary row (forward / backward schedule)
  uint32_t aFS, aWS;
  bool tabs_lds;
  // two exp(y) buffers: frame t (self-loop terms of the per-state pass) and frame t-1 (written under the arc
  // walk, needed to form Y for the next frame); the tight layout has one and pays a barrier instead
  uint32_t pb_cur, pb_next;
  rsrc_t fbase, bbase;
  // ---- per lane (to the compiler: block sums are equal in all lanes but live in vector registers)
  float asum, inv_prev, bsum;
  float chat;        // c^_t: the scale the fixed-point adds of the running GAMMA frame use
  float part, y2, part_tot;
  int ffx0, ffx1, bfx0, bfx1;
  uint32_t fmask, bmask;
  Chunk6 fres[RESF > 0 ? RESF : 1];
  Chunk6 bres[RESB > 0 ? RESB : 1];
  f4 v4[JV];         // alpha_t (un-dashed) of the owned states
  f4 bt[JV];         // GAMMA frames: B_t of the owned states
  f4 areg[JV];
  f4 ycur[PV], ynext[PV];
  f4 bown[JV];  // beta_{t+1} / B_{t+1} of the owned states (the LDS gather source holds Y instead)
#ifdef TC_PHASE_STAMPS
  long long st_prev, st_acc[8];
#endif

  __device__ __forceinline__ TiedSeq(const DenParams &pp, int seq, int meet)
      : p(pp), tid(threadIdx.x), lane(threadIdx.x & 63u), wave(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)), s(seq),
        H(pp.H), P(pp.P), S(pp.S), T(pp.T), Hs(pp.L.Hs), Ps(pp.L.Ps), planes(pp.L.Hs / (4 * kThreads)), K(pp.L.Hs / kThreads),
        own16(16u * threadIdx.x), lane16(16u * (threadIdx.x & 63u)), aACC(4u * (uint32_t)pp.L.off_acc),
        vrow(4u * (uint32_t)pp.L.off_acc + 256u * (uint32_t)((pp.L.Hs / kThreads) * __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) +
             4u * (threadIdx.x & 63u)),
        aGM(4u * (uint32_t)pp.L.off_g), aAL(4u * (uint32_t)pp.L.off_al), aRed(4u * (uint32_t)pp.L.off_red),
        aAsum(4u * (uint32_t)pp.L.off_asum), tab_bytes(4u * (uint32_t)(pp.L.Hs + 4)), row_bytes(4u * (uint32_t)pp.P),
        r_pi(make_rsrc(pp.pi, 4u * (uint32_t)(pp.L.Hs + 4))), r_fs(make_rsrc(pp.tied_fs, 4u * (uint32_t)(pp.L.Hs + 4))),
        r_ws(make_rsrc(pp.tied_w, 4u * (uint32_t)(pp.L.Hs + 4))), leaky(pp.leaky), hist_step((int64_t)pp.S * pp.L.Hs),
        hist(pp.alpha_hist + (int64_t)seq * pp.L.Hs), fn(MITM ? pp.fwd_norm + (int64_t)seq * (pp.T + 2) : nullptr),
        bn(MITM ? pp.bwd_norm + (int64_t)seq * (pp.T + 1) : nullptr), bhist(MITM ? pp.beta_hist + (int64_t)seq * pp.L.Hs : nullptr),
        M(meet) {}

  // which of its JV float4s of states a thread really has
  __device__ __forceinline__ bool plane_on(int j) const { return j < planes; }

  // diagnostic builds (-DTC_PHASE_STAMPS): per-phase cycle totals of a phase's frames, by wave, to p.stamps + base
  __device__ __forceinline__ void stamps_reset() {
#ifdef TC_PHASE_STAMPS
    st_prev = clock64();
    for (int i = 0; i < 8; ++i) st_acc[i] = 0;
#endif
  }
  __device__ __forceinline__ void stamps_flush(int base) {
#ifdef TC_PHASE_STAMPS
    TC_STAMP(0)
    if (!MITM && blockIdx.x == 0 && lane == 0)
      for (int i = 0; i < 8; ++i) p.stamps[base + wave * 8 + i] = st_acc[i];
#endif
  }

  // the cross-entropy output's derivative is zero outside the numerator's posteriors: its row of frame t, written where
  // the derivative's row of that frame is (DenParams::xent_zero)
  __device__ __forceinline__ void xent_zero_row(int t) {
    if (p.xent_zero) {
      const rsrc_t xrow = make_rsrc(p.xent_zero + ((int64_t)t * S + s) * p.xent_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v)
        if (4 * ((int)tid + kThreads * v) < Ps) row_st(xrow, own16 + v * kPlane, p.x_vec, mk4(0.f));
    }
  }

  // ---- the per-state formulas (tied_fwd_state / tied_bwd_state above), with this object's scale and regions
  template <bool GAMMA>
  __device__ __forceinline__ float fwd_state(uint32_t fsx, float wsx, float Fx, float alx, float bx, float gs, float &dpart) {
    return tied_fwd_state<GAMMA>(kPB, aGM, inv_prev, fsx, wsx, Fx, alx, bx, gs, dpart);
  }
  template <bool PURE>
  __device__ __forceinline__ float bwd_state(uint32_t pb, uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx, float ax,
                                             float inv_as, float asum_up) {
    return tied_bwd_state<PURE>(pb, aGM, fsx, wsx, bo, alx, aupx, cpx, ax, inv_as, asum_up);
  }

  // ================================================================================================== forward
  // ---- t = 0: alpha_0 = pi, alpha'_0 = pi + leaky*pi*sum(pi)   ([K] AlphaFirstFrame + AlphaDash(0)); then the
  // wave's stream: descriptor, row-end masks, resident chunks
  __device__ __forceinline__ void forward_begin() {
    f4 pi4[JV];  // (dead after frame 0)
    part = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      pi4[j] = plane_on(j) ? bld4(r_pi, own16, j * kPlane) : mk4(0.f);
      part += hsum(pi4[j]);
    }
    asum = block_sum_a(part, aRed, wave, lane);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) {
        const f4 a = pi4[j] + (leaky * pi4[j]) * asum;
        lds4_st(kA0 + own16 + j * kPlane, a);
        bst4(make_rsrc(hist, 4u * Hs), own16 + j * kPlane, a);
      }
    y2 = 0.f;
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
    if (tid == 0) {
      ldsf_st(aAsum, asum);
      if (MITM) fn[0] = asum;
    }
    inv_prev = __builtin_amdgcn_rcpf(asum);

    {
    const int2 frange = p.fwd.wave_range[wave];
#ifdef TC_ABL_NOSTREAM
    fnch = RESF;
#else
    fnch = __builtin_amdgcn_readfirstlane(frange.y) / kChunk;
#endif
    // (the descriptor covers the wave's range and the look-ahead past it: the array ends with readable padding)
    fbase = make_rsrc(reinterpret_cast<const char *>(p.fwd.cells) +
                          (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                      (uint32_t)(fnch + 2) * (3 * 64 * 16));
    fmask = wave_masks(p.fwd, wave, lane);
    ffx0 = p.fwd.nfix ? p.fwd.fix_begin[tid] : 0;
    ffx1 = p.fwd.nfix ? p.fwd.fix_begin[tid + 1] : 0;
    fsec = aACC + 256u * (uint32_t)(K * kWaves + p.fwd.extra_first[wave]);
#pragma unroll
    for (int i = 0; i < RESF; ++i) load_chunk(fres[i], fbase, lane16, i);
    }
    // A forward phase without gamma does not use the gamma / alpha'_{t+1} / second exp(y) regions: when they hold
    // the two per-state tables (C3: exactly), each thread parks its own entries there and the per-state pass reads
    // them at LDS latency instead of waiting for L2 every frame.  (GAMMA frames: forward_unpark first.)
    tabs_lds = (p.L.off_red - p.L.off_g) >= 2 * Hs;
    aFS = aGM;
    aWS = aGM + 4u * (uint32_t)Hs;
    if (tabs_lds) {
#pragma unroll
      for (int j = 0; j < JV; ++j)
        if (plane_on(j)) {
          *reinterpret_cast<lds_u4 *>(aFS + own16 + j * kPlane) = bld4u(r_fs, own16, j * kPlane);
          lds4_st(aWS + own16 + j * kPlane, bld4(r_ws, own16, j * kPlane));
        }
    }
    // which resident chunk a wave issues its deferred stores after: one wave generation per chunk
    store_slot = RESF >= 4 ? wave >> 2 : RESF >= 2 ? wave >> 3 : 0;
#pragma unroll
    for (int j = 0; j < JV; ++j) v4[j] = bt[j] = mk4(0.f);
    chat = 0.f;
    part_tot = 0.f;
    stamps_reset();
  }

  // frame t = 1..T   ([K] AlphaGeneralFrame(t) + AlphaDash(t))
  template <bool GAMMA>
  __device__ __forceinline__ void forward_frame(int t) {
#ifdef TC_PHASE_STAMPS
    long long *const wst = st_acc + 5;  // (the walk's sub-stamps)
#endif
    TC_STAMP(0)
    Chunk6 q0;
    load_chunk(q0, fbase, lane16, RESF);  // (past a short stream: readable padding, never processed)
    __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready; gamma zero
    TC_STAMP(1)
    f4 yreg[PV];
    if (t < T) {  // y_t under the arc walk
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) yreg[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
    }
    float n_t = 1.f;
    if (GAMMA && t < T) n_t = vload_f32(bn + t);  // for c^_{t+1}
    // The history row of frame t-1 is stored from here, not from the end of frame t-1: a CU issues a 1 KB
    // store instruction only every ~60 cycles, so the 32 of a frame, issued back to back by 16 waves,
    // held the frame's tail for ~1.9k cycles (profiles/r02_phase_stamps_before_spread.txt).  Under the walk
    // the store path is idle: the four wave generations issue theirs after resident chunk 0, 1, 2, 3.
    // (GAMMA frames: rows above the meeting frame have no reader.)
    age_prio_on(wave);
    bool stored = false;  // (nothing resident: the four wave generations store after streamed pair 0, 1, 2, 3)
    const RowCommit frc{aACC + 256u * (uint32_t)(K * wave), fsec, K};
    walk<kA0, RESF>(fres, q0, fbase, lane16, fnch, fmask, frc, [&](int i) {
      if (!GAMMA && t > 1 && !stored && (RESF > 0 ? i == store_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
        stored = true;
        const rsrc_t hist_prev = make_rsrc(hist + (int64_t)(t - 1) * hist_step, 4u * Hs);
#ifndef TC_ABL_NOHIST
#pragma unroll
        for (int j = 0; j < JV; ++j)  // alpha'_{t-1} of the owned states: still in the gather buffer
          if (plane_on(j)) bst4(hist_prev, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
#endif
      }
    } TC_WALK_PASS);
    __builtin_amdgcn_s_setprio(0);
    TC_STAMP(2)
    // graphs with hub states: the secondary rows of a state are walked by lanes of the wave that owns it
    // (schedule_owner.cpp), and a wave's LDS operations execute in order: no barrier
    for (int e = ffx0; e < ffx1; ++e) fold_row(p.fwd.fix[e], vrow, aACC, Hs, K);
    TC_STAMP(3)
    part = 0.f;
    float dpart = 0.f;
    u4 fs[JV];
    f4 ws[JV], cpi[JV];
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) {
        fs[j] = tabs_lds ? lds4u(aFS + own16 + j * kPlane) : bld4u(r_fs, own16, j * kPlane);
        ws[j] = tabs_lds ? lds4(aWS + own16 + j * kPlane) : bld4(r_ws, own16, j * kPlane);
        cpi[j] = bld4(r_pi, own16, j * kPlane);  // pi: first touched behind the reduction, which hides its L2 trip
      }
    const float gs = kGammaScale * chat;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      v4[j] = mk4(0.f);
      if (plane_on(j)) {
        const f4 F = own_rows(vrow, j);
        const f4 al = lds4(kA0 + own16 + j * kPlane);  // alpha'_{t-1} of the owned states
        v4[j] = f4{fwd_state<GAMMA>(fs[j].x, ws[j].x, F.x, al.x, bt[j].x, gs, dpart), fwd_state<GAMMA>(fs[j].y, ws[j].y, F.y, al.y, bt[j].y, gs, dpart),
                   fwd_state<GAMMA>(fs[j].z, ws[j].z, F.z, al.z, bt[j].z, gs, dpart), fwd_state<GAMMA>(fs[j].w, ws[j].w, F.w, al.w, bt[j].w, gs, dpart)};
        part += hsum(v4[j]);
      }
    }
    f4 yp[PV];
    if (GAMMA) {
      // y_{t-1} for the derivative row's l2 term (this CU read the row a frame ago: L2) and the next frame's B row
      // (index clamped, assignment unconditional)
      const rsrc_t yprev = make_rsrc(p.y + ((int64_t)(t - 1) * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) yp[v] = row_ld(yprev, own16 + v * kPlane, p.y_vec);
      const rsrc_t brow = make_rsrc(bhist + (int64_t)(t + 1 <= T ? t + 1 : T) * hist_step, 4u * Hs);
#pragma unroll
      for (int j = 0; j < JV; ++j) bt[j] = bld4(brow, own16, j * kPlane);
      block_sum2(part, dpart, aRed, wave, lane);  // its barrier also completes gamma_{t-1}
      asum = part;
    } else {
      asum = block_sum_a(part, aRed, wave, lane);
    }
    __builtin_amdgcn_sched_barrier(0);  // (keeps the multiply by leaky, and with it the wait for pi, down here)
    TC_STAMP(4)
    part_tot = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) {
        const f4 a = v4[j] + (leaky * cpi[j]) * asum;
        lds4_st(kA0 + own16 + j * kPlane, a);
        part_tot += hsum(a);
      }
    if (GAMMA) {
      // the derivative row of frame t-1: gamma_{t-1} * (c_t / c^_t)
      const float c = __builtin_amdgcn_rcpf(dpart);
      const float sa = p.deriv_weight * (kGammaInvScale * (c * __builtin_amdgcn_rcpf(chat)));
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t - 1) * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) {
          const u4 gu = lds4u(aGM + 4u * i0);
          lds4_st(aGM + 4u * i0, mk4(0.f));
          f4 o = sa * f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} - p.l2_scale * yp[v];
          if (ACCUM) o += row_ld(drow, own16 + v * kPlane, p.d_vec);
          row_st(drow, own16 + v * kPlane, p.d_vec, o);
        }
      }
      xent_zero_row(t - 1);
      chat = c * asum * __builtin_amdgcn_rcpf(n_t);  // c^_{t+1} = c_t asum_t / n_t
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
    if (tid == 0) {
      ldsf_st(aAsum + 4u * t, asum);
      if (MITM) fn[t] = asum;
    }
    inv_prev = __builtin_amdgcn_rcpf(asum);
  }

  // before the first GAMMA frame: the parked tables give the gamma region back, gamma starts at zero (the caller's
  // next barrier publishes it)
  __device__ __forceinline__ void forward_unpark() {
    tabs_lds = false;
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      if (i0 < Ps) lds4_st(aGM + 4u * i0, mk4(0.f));
    }
  }

  // alpha'_t, still in the gather buffer, to its history row (the last frame of a forward phase has no walk behind it)
  __device__ __forceinline__ void forward_store_row(int t) {
    const rsrc_t hist_t = make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) bst4(hist_t, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
  }

  // ---- total probability ([K] ComputeTotLogLike): tot = sum_h alpha'_T(h); `bad` = NaN poisons the sequence
  __device__ __forceinline__ float forward_total(double bad) {
    const float tot = block_sum_a(part_tot, aRed + 4u * kWaves, wave, lane);
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      // [K] log-prob = log(tot) + sum over t < T of log(alpha-sum_t): the scales divided out of frames 1..T
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(ldsf(aAsum + 4u * t));
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d) + bad;  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
    return tot;
  }

  // B_{t}, the owned states' values after pure frame t, to its history row (the last pure frame has no walk behind it)
  __device__ __forceinline__ void backward_store_row(int t) {
    const rsrc_t brow = make_rsrc(bhist + (int64_t)t * hist_step, 4u * Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) bst4(brow, own16 + j * kPlane, bown[j]);
  }

  // ================================================================================================== backward
  // ---- [K] BetaDashLastFrame, Beta(T): beta'_T(h) = b_T on the real states (1 / tot: Kaldi's scale; 1: a pure recursion
  // with normalisers of its own), beta_T = beta'_T + leaky * sum_h pi(h) beta'_T(h).  The LDS regions now hold Y
  // (gather source), the row sums, exp(y_t), exp(y_{t-1}), gamma_t and (roomy layout) alpha'_{t+1}.
  template <bool PURE>
  __device__ __forceinline__ void backward_begin(float b_T) {
    part = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (plane_on(j)) part += hsum(leaky * bld4(r_pi, own16, j * kPlane)) * b_T;
    bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);  // also orders the reuse of the gather buffer
    pb_cur = kPB;
    pb_next = ALPHA_LDS ? 4u * (uint32_t)p.L.off_p2 : kPB;
    {
    const int2 brange = p.bwd.wave_range[wave];
#ifdef TC_ABL_NOSTREAM
    bnch = RESB;
#else
    bnch = __builtin_amdgcn_readfirstlane(brange.y) / kChunk;
#endif
    bbase = make_rsrc(reinterpret_cast<const char *>(p.bwd.cells) +
                          (int64_t)(__builtin_amdgcn_readfirstlane(brange.x) / kChunk) * (3 * 64 * 16),
                      (uint32_t)(bnch + 2) * (3 * 64 * 16));
    bmask = wave_masks(p.bwd, wave, lane);
    bfx0 = p.bwd.nfix ? p.bwd.fix_begin[tid] : 0;
    bfx1 = p.bwd.nfix ? p.bwd.fix_begin[tid + 1] : 0;
    bsec = aACC + 256u * (uint32_t)(K * kWaves + p.bwd.extra_first[wave]);
#pragma unroll
    for (int i = 0; i < RESB; ++i) load_chunk(bres[i], bbase, lane16, i);
    }
    {
      const rsrc_t hist_up = make_rsrc(hist + (int64_t)T * hist_step, 4u * Hs);
      const rsrc_t bT = make_rsrc(bhist + (int64_t)T * hist_step, PURE ? 4u * Hs : 0u);
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        bown[j] = areg[j] = mk4(0.f);
        if (plane_on(j)) {
          const int h0 = 4 * ((int)tid + kThreads * j);
          const float b = b_T + bsum;
          bown[j] = f4{h0 < H ? b : 0.f, h0 + 1 < H ? b : 0.f, h0 + 2 < H ? b : 0.f, h0 + 3 < H ? b : 0.f};
          if (PURE) bst4(bT, own16 + j * kPlane, bown[j]);  // B_T for the partner's gamma_{T-1}
          if (!PURE && ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, bld4(hist_up, own16, j * kPlane));
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
        if (plane_on(j)) {
          const u4 fs = bld4u(r_fs, own16, j * kPlane);
          lds4_st(kA0 + own16 + j * kPlane,
                  f4{bown[j].x * ldsf(pb_cur + (fs.x & 0xffffu)), bown[j].y * ldsf(pb_cur + (fs.y & 0xffffu)),
                     bown[j].z * ldsf(pb_cur + (fs.z & 0xffffu)), bown[j].w * ldsf(pb_cur + (fs.w & 0xffffu))});
        }
    }
    bstore_slot = RESB >= 4 ? wave >> 2 : RESB >= 2 ? wave >> 3 : 0;
    stamps_reset();
  }

#ifdef TC_NO_BWD_DEFER
  static constexpr bool kDeferDeriv = false;
#else
  static constexpr bool kDeferDeriv = ALPHA_LDS;  // (the tight layout has one exp(y) buffer: nowhere to wait)
#endif

  // frame t = T-1..0   ([K] BetaDashGeneralFrame(t) + Beta(t)); t_top: the first frame of the run of gamma frames this
  // one belongs to + 1 (the derivative row of frame t+1 waits for this frame's walk only behind a gamma frame).
  // Returns true after frame 0 (the two checks written; nothing follows).
  template <bool PURE>
  __device__ __forceinline__ bool backward_frame(int t, int t_top) {
#ifdef TC_PHASE_STAMPS
    long long *const wst = st_acc + 5;
#endif
    TC_STAMP(0)
    Chunk6 q0;
    load_chunk(q0, bbase, lane16, RESB);
    __syncthreads();  // Y, exp(y_t), alpha'_{t+1} ready; row sums and gamma zero
    TC_STAMP(1)
    const float asum_t = PURE ? 1.f : ldsf(aAsum + 4u * t);
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
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
      if (!PURE) {
        const rsrc_t hist_t = make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs);
#pragma unroll
        for (int j = 0; j < JV; ++j) areg[j] = plane_on(j) ? bld4(hist_t, own16, j * kPlane) : mk4(0.f);
      }
    }
    // beta'_t(h) * asum_t = sum over out-arcs of w * Y(dst): the same walk as forward, no atomics
    // The derivative row of frame t+1 leaves from here, for the reason given at the forward walk (16 stores in
    // a row held the backward tail for ~1k cycles: profiles/r02_phase_stamps.txt, tail of waves 0-3 vs 12-15).
    // It waits, thread-private, in the exp(y) buffer that went dead with frame t+1's per-state pass and that
    // this thread overwrites only after its walk.
    age_prio_on(wave);
    bool dstored = false;
    const RowCommit brc{aACC + 256u * (uint32_t)(K * wave), bsec, K};
    // (PURE: the history row of B_{t+1}, still in this thread's registers, leaves the same way; row T went out in
    // backward_begin, the last pure frame's row goes out in backward_store_row)
    walk<kA0, RESB>(bres, q0, bbase, lane16, bnch, bmask, brc, [&](int i) {
      if ((PURE ? t < T - 1 : kDeferDeriv && t < t_top - 1) && !dstored &&
          (RESB > 0 ? i == bstore_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
        dstored = true;
        if (PURE) {
          const rsrc_t brow_up = make_rsrc(bhist + (int64_t)(t + 1) * hist_step, 4u * Hs);
#pragma unroll
          for (int j = 0; j < JV; ++j)
            if (plane_on(j)) bst4(brow_up, own16 + j * kPlane, bown[j]);
        } else {
          const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t + 1) * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
          for (int v = 0; v < PV; ++v)
            if (4 * ((int)tid + kThreads * v) < Ps) row_st(drow, own16 + v * kPlane, p.d_vec, lds4(pb_next + own16 + v * kPlane));
          xent_zero_row(t + 1);
        }
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
    float part_u = 0.f, part_ab = 0.f, part_g = 0.f;
    const float asum_up = PURE ? 0.f : ldsf(aAsum + 4u * (t + 1));
    // With 16 states per thread the tables of plane j + 1 are requested before plane j is worked on: left at the top
    // of their own iteration the loads waited behind the LDS atomics of the plane before, one exposed L2 round trip
    // per plane (17 k cycles per frame for this pass on a 9681-state graph: profiles/r02_phase_stamps_r3.txt).  The
    // instantiations with resident stream chunks have no registers for that.
    constexpr bool kAhead = RESB == 0;
    u4 fs_n = u4{0u, 0u, 0u, 0u};
    f4 ws_n = mk4(0.f), cp_n = mk4(0.f), aup_n = mk4(0.f);
    auto request = [&](int j) __attribute__((always_inline)) {
      fs_n = bld4u(r_fs, own16, j * kPlane);
      ws_n = bld4(r_ws, own16, j * kPlane);
      cp_n = bld4(r_pi, own16, j * kPlane);
      // alpha'_{t+1}: parked by this thread (roomy layout) or re-read from the history (tight layout)
      if (!PURE && !ALPHA_LDS) aup_n = bld4(make_rsrc(hist + (int64_t)(t + 1) * hist_step, 4u * Hs), own16, j * kPlane);
    };
    if (kAhead) request(0);
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      b4[j] = mk4(0.f);
      if (plane_on(j)) {
        if (!kAhead) request(j);
        const u4 fs = fs_n;
        const f4 ws = ws_n;
        const f4 cp = leaky * cp_n;
        const f4 aup_g = aup_n;
        if (kAhead && j + 1 < planes) request(j + 1);
        f4 a = own_rows(vrow, j);
        const f4 al = areg[j];  // alpha'_t of the owned states
        const f4 aup = (!PURE && ALPHA_LDS) ? lds4(aAL + own16 + j * kPlane) : aup_g;
        a.x = bwd_state<PURE>(pb_cur, fs.x, ws.x, bown[j].x, al.x, aup.x, cp.x, a.x, inv_as, asum_up);
        a.y = bwd_state<PURE>(pb_cur, fs.y, ws.y, bown[j].y, al.y, aup.y, cp.y, a.y, inv_as, asum_up);
        a.z = bwd_state<PURE>(pb_cur, fs.z, ws.z, bown[j].z, al.z, aup.z, cp.z, a.z, inv_as, asum_up);
        a.w = bwd_state<PURE>(pb_cur, fs.w, ws.w, bown[j].w, al.w, aup.w, cp.w, a.w, inv_as, asum_up);
        b4[j] = PURE ? a : a * inv_as;  // [K] * inv_arbitrary_scale
        fpk[j][0] = (fs.x & 0xffffu) | (fs.y << 16);
        fpk[j][1] = (fs.z & 0xffffu) | (fs.w << 16);
        part += hsum(cp * b4[j]);
        if (PURE) part_u += hsum(a);
        if (!PURE && t == 0) part_ab += hsum(al * b4[j]);
      }
    }
    float inv_n = 1.f;
    if (PURE) {
      // n_t = sum_h U_t(h) / H; B'_t = U_t / n_t; leaky sum of B'_t
      block_sum2(part, part_u, aRed, wave, lane);
      const float n = part_u * (1.0f / (float)H);
      inv_n = __builtin_amdgcn_rcpf(n);
      bsum = part * inv_n;
      if (tid == 0) bn[t] = __builtin_amdgcn_rcpf(inv_n);  // (the normaliser actually applied)
    } else {
      bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    }
    TC_STAMP(4)
    if (!PURE) {
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
            lds4_st(pb_cur + 4u * i0, o);  // (and the cross-entropy output's zero row with it, under the next walk)
          else
            row_st(drow, own16 + v * kPlane, p.d_vec, o);
#else
          if (o.x == 123.456f) row_st(drow, own16 + v * kPlane, p.d_vec, o);
#endif
        }
      }
      if (!(kDeferDeriv && t > 0)) xent_zero_row(t);
      if (t == 0) {
        // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
        const float ab = block_sum_a(part_ab, aRed + 4u * kWaves, wave, lane);
        const float gsum = block_sum_a(part_g, aRed + 8u * kWaves, wave, lane);
        if (tid == 0) {
          p.seq_ab[s] = ab;
          p.seq_gsum[s] = gsum;
        }
        return true;
      }
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
      if (plane_on(j)) {
        const f4 b = PURE ? b4[j] * inv_n + bsum : b4[j] + bsum;
        bown[j] = b;
        const f4 yv = f4{b.x * ldsf(pb_next + (fpk[j][0] & 0xffffu)), b.y * ldsf(pb_next + (fpk[j][0] >> 16)),
                         b.z * ldsf(pb_next + (fpk[j][1] & 0xffffu)), b.w * ldsf(pb_next + (fpk[j][1] >> 16))};
        if constexpr (kA0 + (JV - 1) * kPlane < 65536u)
          lds4_st_at(own16, kA0 + j * kPlane, yv);
        else
          lds4_st(kA0 + own16 + j * kPlane, yv);
        if (!PURE && ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, areg[j]);
      }
#pragma unroll
    for (int v = 0; v < PV; ++v) ycur[v] = ynext[v];
    const uint32_t tmp = pb_cur;
    pb_cur = pb_next;
    pb_next = tmp;
    return false;
  }
};

}  // namespace tc
