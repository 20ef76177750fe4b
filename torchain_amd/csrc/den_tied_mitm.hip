// Two CUs per sequence that meet in the middle: the form of den_tied_kernel.hip for batches of at most half the
// chip's CUs in which no combining pass is left.
//
// den_tied_split.hip runs a sequence's forward and backward recursion on two CUs at once and forms gamma in a third,
// HBM-bound pass over both histories (0.15 ms at batch 64, 0.30 at 128, 0.45 at C5).  Here the two recursions stop being
// pure when they have passed each other: role F runs alpha forward over frames 1..M (M = T/2) exactly as the fused
// kernel's forward phase, role B runs B backward over frames T-1..M with normalisers of its own (B'_T = 1,
// n_t = sum_h U_t(h) / H: the recursion is linear, so any positive normaliser does); both publish what they stored
// (alpha'_0..M and asum_0..M; B_M..T and n_M..T-1), wait for each other once, and continue --
//   role F through frames M+1..T, forming gamma_{t-1} in the per-state pass of frame t from the two parts of
//          alpha_t(g) it computes anyway (forward-class part p(f)F, self-loop part p(s) w_s alpha'_{t-1}) and the stored
//          B_t(g), scaled by c_t = 1 / sum_g alpha_t(g) B_t(g) (the fixed-point adds, which need a scale before that sum
//          is known, use c^_t = c_{t-1} asum_{t-1} / n_{t-1} and the row is corrected by c_t / c^_t when it is converted);
//   role B through frames M-1..0 with the fused kernel's backward frame unchanged, after converting its B_M to Kaldi's
//          scale by the same invariant (beta_M = c_M B_M) and taking asum_0..M from role F --
// so every derivative row is written once, by the role that reaches it, and every frame is walked once per role.
// The two roles of a sequence are two workgroups of ONE launch, paired by ticket (den_tied_pair.hip's protocol).
//
// The frame code is den_tied_kernel.hip's (same helpers, same LDS layout, same resident chunks); what differs is
// marked ROLE F / ROLE B.  [K] = kaldi chain-denominator.cc, reached by the reference through
// src/my_lib_chain.cpp:129-131.
#include "den_tied_device.h"

namespace tc {

namespace {

struct MitmParams {
  uint32_t *sync;    // [0] ticket counter, [4 + 2 s + role] hand-over flags
  int M;             // meeting frame
  uint32_t aScr;     // 16 bytes of LDS scratch behind the fused layout: ticket, hand-over result
};

typedef __attribute__((address_space(1))) uint32_t gu32;
typedef __attribute__((address_space(3))) uint32_t lds_u;

constexpr uint32_t kSpinSleep = 16;        // s_sleep units (64 cycles each) between two polls
constexpr uint32_t kSpinLimit = 8u << 20;  // seconds

// "everything this workgroup stored so far may be read by the partner" (MI355X_MICROARCH.md, valid forms: plain stores
// -> vmcnt(0) -> barrier -> release -> vmcnt(0) -> relaxed agent flag store)
__device__ __forceinline__ void publish(uint32_t *flag, uint32_t tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((gu32 *)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// (relaxed poll -> acquire -> vmcnt(0) -> barrier); false if the partner never arrived
__device__ __forceinline__ bool await(uint32_t *flag, uint32_t tid, uint32_t scratch) {
  if (tid == 0) {
    uint32_t spins = 0, ok = 1;
    while (__hip_atomic_load((gu32 *)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
      __builtin_amdgcn_s_sleep(kSpinSleep);
      if (++spins > kSpinLimit) {
        ok = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    *reinterpret_cast<lds_u *>(scratch) = ok;
  }
  __syncthreads();
  return *reinterpret_cast<lds_u *>(scratch) != 0u;
}

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

// a scalar another CU wrote (vector load: the scalar cache is not covered by the acquire)
__device__ __forceinline__ float vload_f32(const float *ptr) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(make_rsrc(ptr, 4u), 0, 0, 0));
}

// =========================================================================================================
// ROLE F: den_tied_kernel's forward phase; frames M+1..T also form gamma_{t-1}.
// =========================================================================================================
template <int JV, int PV, bool ACCUM, int RESF>
__device__ __forceinline__ void mitm_forward(const DenParams &p, const MitmParams &q, int s) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = p.P, S = p.S, T = p.T, M = q.M;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);
  const int K = Hs / kThreads;
  const uint32_t own16 = 16u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u;
  constexpr uint32_t kA0 = PV * 16u * kThreads;
  const uint32_t aACC = 4u * (uint32_t)p.L.off_acc;
  const uint32_t vrow = aACC + 256u * (uint32_t)(K * wave) + 4u * lane;
  const uint32_t aGM = 4u * (uint32_t)p.L.off_g;
  const uint32_t aRed = 4u * (uint32_t)p.L.off_red;
  const uint32_t aAsum = 4u * (uint32_t)p.L.off_asum;
  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4), row_bytes = 4u * (uint32_t)P;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const float leaky = p.leaky;
  float *const fn = p.fwd_norm + (int64_t)s * (T + 2);
  const float *const bn = p.bwd_norm + (int64_t)s * (T + 1);
  const int64_t hist_step = (int64_t)S * Hs;
  float *const hist = p.alpha_hist + (int64_t)s * Hs;
  const float *const bhist = p.beta_hist + (int64_t)s * Hs;

  f4 pi4[JV];
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    pi4[j] = j < planes ? bld4(r_pi, own16, j * kPlane) : mk4(0.f);
    part += hsum(pi4[j]);
  }
  // ---- t = 0   ([K] AlphaFirstFrame + AlphaDash(0))
  float asum = block_sum_a(part, aRed, wave, lane);
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
        lds4_st(aGM + 4u * i0, mk4(0.f));  // ROLE F: gamma starts at zero
      }
    }
  }
  if (tid == 0) {
    ldsf_st(aAsum, asum);
    fn[0] = asum;
  }
  float inv_prev = __builtin_amdgcn_rcpf(asum);

  const int2 frange = p.fwd.wave_range[wave];
  const int fnch = __builtin_amdgcn_readfirstlane(frange.y) / kChunk;
  const rsrc_t fbase = make_rsrc(reinterpret_cast<const char *>(p.fwd.cells) +
                                     (int64_t)(__builtin_amdgcn_readfirstlane(frange.x) / kChunk) * (3 * 64 * 16),
                                 (uint32_t)(fnch + 2) * (3 * 64 * 16));
  const uint32_t fmask = wave_masks(p.fwd, wave, lane);
  const int ffx0 = p.fwd.nfix ? p.fwd.fix_begin[tid] : 0, ffx1 = p.fwd.nfix ? p.fwd.fix_begin[tid + 1] : 0;
  const RowCommit frc{aACC + 256u * (uint32_t)(K * wave), aACC + 256u * (uint32_t)(K * kWaves + p.fwd.extra_first[wave]), K};
  Chunk6 fres[RESF > 0 ? RESF : 1];
#pragma unroll
  for (int i = 0; i < RESF; ++i) load_chunk(fres[i], fbase, lane16, i);
  const int store_slot = RESF >= 4 ? wave >> 2 : RESF >= 2 ? wave >> 3 : 0;

  f4 v4[JV];            // alpha_t (un-dashed) of the owned states
  f4 bt[JV];            // ROLE F, second phase: B_t of the owned states
#pragma unroll
  for (int j = 0; j < JV; ++j) v4[j] = bt[j] = mk4(0.f);
  float chat = 0.f;     // c^_t: the scale the fixed-point adds of the running frame use
  float part_tot = 0.f;

  auto frame = [&](int t, auto gamma_tag) __attribute__((always_inline)) {
    constexpr bool GAMMA = decltype(gamma_tag)::value;
    Chunk6 q0;
    load_chunk(q0, fbase, lane16, RESF);
    __syncthreads();  // alpha'_{t-1}, exp(y_{t-1}) ready; gamma zero
    f4 yreg[PV];
    if (t < T) {
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) yreg[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
    }
    float n_t = 1.f;
    if (GAMMA && t < T) n_t = vload_f32(bn + t);  // for c^_{t+1}
    age_prio_on(wave);
    bool stored = false;
    walk<kA0, RESF>(fres, q0, fbase, lane16, fnch, fmask, frc, [&](int i) {
      // rows 0..M-1 of the alpha' history leave under the walks of the first phase (row M at the hand-over; rows above
      // M have no reader)
      if (!GAMMA && t > 1 && !stored && (RESF > 0 ? i == store_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
        stored = true;
        const rsrc_t hist_prev = make_rsrc(hist + (int64_t)(t - 1) * hist_step, 4u * Hs);
#pragma unroll
        for (int j = 0; j < JV; ++j)
          if (j < planes) bst4(hist_prev, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
      }
    });
    __builtin_amdgcn_s_setprio(0);
    for (int e = ffx0; e < ffx1; ++e) fold_row(p.fwd.fix[e], vrow, aACC, Hs, K);
    part = 0.f;
    float dpart = 0.f;
    u4 fs[JV];
    f4 ws[JV], cpi[JV];
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        fs[j] = bld4u(r_fs, own16, j * kPlane);
        ws[j] = bld4(r_ws, own16, j * kPlane);
        cpi[j] = bld4(r_pi, own16, j * kPlane);
      }
    const float gs = kGammaScale * chat;
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      v4[j] = mk4(0.f);
      if (j < planes) {
        const f4 F = own_rows(vrow, j);
        const f4 al = lds4(kA0 + own16 + j * kPlane);  // alpha'_{t-1} of the owned states
        // alpha_t(g) asum_{t-1} = p(f(g)) F(g) + p(s(g)) w_s alpha'_{t-1}(g); with B_t(g) the two parts are the
        // occupations in frame t-1 of the forward-class arcs into g and of its self-loop
        auto one = [&](uint32_t fsx, float wsx, float Fx, float alx, float bx) __attribute__((always_inline)) {
          const float pf = ldsf(kPB + (fsx & 0xffffu)), ps = ldsf(kPB + (fsx >> 16));
          const float sp = ps * (wsx * alx);
          const float a = fmaf(pf, Fx, sp) * inv_prev;  // (= tied_alpha(...) * inv_prev)
          if constexpr (GAMMA) {
            const float g = gs * bx, spn = sp * inv_prev;
            gamma_add_a(aGM + (fsx >> 16), g * spn);
            gamma_add_a(aGM + (fsx & 0xffffu), g * fmaxf(a - spn, 0.f));
            dpart = fmaf(a, bx, dpart);
          }
          return a;
        };
        v4[j] = f4{one(fs[j].x, ws[j].x, F.x, al.x, bt[j].x), one(fs[j].y, ws[j].y, F.y, al.y, bt[j].y),
                   one(fs[j].z, ws[j].z, F.z, al.z, bt[j].z), one(fs[j].w, ws[j].w, F.w, al.w, bt[j].w)};
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
    __builtin_amdgcn_sched_barrier(0);
    part_tot = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
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
      fn[t] = asum;
    }
    inv_prev = __builtin_amdgcn_rcpf(asum);
  };

  // ---- first phase: frames 1..M
  for (int t = 1; t <= M; ++t) frame(t, std::false_type());
  {
    const rsrc_t hist_M = make_rsrc(hist + (int64_t)M * hist_step, 4u * Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) bst4(hist_M, own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane));
  }
  publish(q.sync + 4 + 2 * s, tid);
  const bool partner_ok = await(q.sync + 4 + 2 * s + 1, tid, q.aScr + 4u);
  {
    // c_M = 1 / sum_g alpha_M(g) B_M(g);  c^_{M+1} = c_M asum_M / n_M;  B_{M+1} for the next frame
    const rsrc_t bM = make_rsrc(bhist + (int64_t)M * hist_step, 4u * Hs);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) d += hsum(v4[j] * bld4(bM, own16, j * kPlane));
    d = block_sum_a(d, aRed + 4u * kWaves, wave, lane);
    chat = __builtin_amdgcn_rcpf(d) * asum * __builtin_amdgcn_rcpf(vload_f32(bn + M));
    const rsrc_t brow = make_rsrc(bhist + (int64_t)(M + 1) * hist_step, 4u * Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j) bt[j] = bld4(brow, own16, j * kPlane);
  }
  // ---- second phase: frames M+1..T with gamma_{t-1}
  for (int t = M + 1; t <= T; ++t) frame(t, std::true_type());

  // ---- total probability ([K] ComputeTotLogLike)
  const float tot = block_sum_a(part_tot, aRed + 4u * kWaves, wave, lane);
  {
    const double y2d = (double)block_sum_a(y2, aRed + 8u * kWaves, wave, lane);
    if (tid == 0) {
      double logsum = 0.0;
      for (int t = 0; t < T; ++t) logsum += (double)__logf(ldsf(aAsum + 4u * t));
      const double bad = partner_ok ? 0.0 : (double)__builtin_nanf("");
      p.seq_logprob[s] = logsum + (double)__logf(tot) + (y2d - y2d) + bad;  // (+ 0, or NaN for a NaN / inf input)
      p.seq_y2[s] = y2d;
    }
  }
}

// =========================================================================================================
// ROLE B: frames T-1..M with normalisers of its own and no gamma; then den_tied_kernel's backward frame.
// =========================================================================================================
template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, int RESB>
__device__ __forceinline__ void mitm_backward(const DenParams &p, const MitmParams &q, int s) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, P = p.P, S = p.S, T = p.T, M = q.M;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);
  const int K = Hs / kThreads;
  const uint32_t own16 = 16u * tid, lane16 = 16u * lane;
  constexpr uint32_t kPB = 0u;
  constexpr uint32_t kA0 = PV * 16u * kThreads;
  const uint32_t aACC = 4u * (uint32_t)p.L.off_acc;
  const uint32_t vrow = aACC + 256u * (uint32_t)(K * wave) + 4u * lane;
  const uint32_t aGM = 4u * (uint32_t)p.L.off_g;
  const uint32_t aAL = 4u * (uint32_t)p.L.off_al;
  const uint32_t aRed = 4u * (uint32_t)p.L.off_red;
  const uint32_t aAsum = 4u * (uint32_t)p.L.off_asum;
  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4), row_bytes = 4u * (uint32_t)P;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const float leaky = p.leaky;
  const float *const fn = p.fwd_norm + (int64_t)s * (T + 2);
  float *const bn = p.bwd_norm + (int64_t)s * (T + 1);
  const int64_t hist_step = (int64_t)S * Hs;
  const float *const hist = p.alpha_hist + (int64_t)s * Hs;
  float *const bhist = p.beta_hist + (int64_t)s * Hs;
  const float inv_h = 1.0f / (float)H;

  // B'_T = 1; B_T = B'_T + leaky sum_h pi(h) B'_T(h)
  float part = 0.f;
#pragma unroll
  for (int j = 0; j < JV; ++j)
    if (j < planes) part += hsum(leaky * bld4(r_pi, own16, j * kPlane));
  float bsum = block_sum_a(part, aRed + 12u * kWaves, wave, lane);
  f4 areg[JV];
  f4 ycur[PV], ynext[PV];
  f4 bown[JV];  // B_{t+1} / beta_{t+1} of the owned states (the LDS gather source holds Y instead)
  uint32_t pb_cur = kPB, pb_next = ALPHA_LDS ? 4u * (uint32_t)p.L.off_p2 : kPB;
  const int2 brange = p.bwd.wave_range[wave];
  const int bnch = __builtin_amdgcn_readfirstlane(brange.y) / kChunk;
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
    const rsrc_t yrow = make_rsrc(p.y + ((int64_t)(T - 1) * S + s) * p.y_stride, row_bytes);
    const rsrc_t bT = make_rsrc(bhist + (int64_t)T * hist_step, 4u * Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      bown[j] = areg[j] = mk4(0.f);
      if (j < planes) {
        const int h0 = 4 * ((int)tid + kThreads * j);
        const float b = 1.0f + bsum;
        bown[j] = f4{h0 < H ? b : 0.f, h0 + 1 < H ? b : 0.f, h0 + 2 < H ? b : 0.f, h0 + 3 < H ? b : 0.f};
        bst4(bT, own16 + j * kPlane, bown[j]);  // ROLE B: B_T for role F's gamma_{T-1}
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
    __syncthreads();  // exp(y_{T-1}) complete: Y_{T-1}(g) = B_T(g) p_{T-1}(f(g))
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const u4 fs = bld4u(r_fs, own16, j * kPlane);
        lds4_st(kA0 + own16 + j * kPlane,
                f4{bown[j].x * ldsf(pb_cur + (fs.x & 0xffffu)), bown[j].y * ldsf(pb_cur + (fs.y & 0xffffu)),
                   bown[j].z * ldsf(pb_cur + (fs.z & 0xffffu)), bown[j].w * ldsf(pb_cur + (fs.w & 0xffffu))});
      }
  }
  constexpr bool kDeferDeriv = ALPHA_LDS;
  const int bstore_slot = RESB >= 4 ? wave >> 2 : RESB >= 2 ? wave >> 3 : 0;

  auto frame = [&](int t, auto pure_tag) __attribute__((always_inline)) {
    constexpr bool PURE = decltype(pure_tag)::value;
    Chunk6 q0;
    load_chunk(q0, bbase, lane16, RESB);
    __syncthreads();  // Y, exp(y_t) (and alpha'_{t+1}) ready; row sums and gamma zero
    const float asum_t = PURE ? 1.f : ldsf(aAsum + 4u * t);
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
    {
      const int tn = t > 0 ? t - 1 : 0;
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)tn * S + s) * p.y_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) ynext[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
      if (!PURE) {
        const rsrc_t hist_t = make_rsrc(hist + (int64_t)t * hist_step, 4u * Hs);
#pragma unroll
        for (int j = 0; j < JV; ++j) areg[j] = j < planes ? bld4(hist_t, own16, j * kPlane) : mk4(0.f);
      }
    }
    age_prio_on(wave);
    bool dstored = false;
    walk<kA0, RESB>(bres, q0, bbase, lane16, bnch, bmask, brc, [&](int i) {
      // (the row of frame t+1 waits in the dead exp(y) buffer: only behind a gamma frame, i.e. below M-1)
      if (!PURE && kDeferDeriv && t < M - 1 && !dstored && (RESB > 0 ? i == bstore_slot : (i == kWalkEnd || i == -1 - (wave >> 2)))) {
        dstored = true;
        const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)(t + 1) * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
        for (int v = 0; v < PV; ++v)
          if (4 * ((int)tid + kThreads * v) < Ps) row_st(drow, own16 + v * kPlane, p.d_vec, lds4(pb_next + own16 + v * kPlane));
      }
    });
    __builtin_amdgcn_s_setprio(0);
    if (ALPHA_LDS) {
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) lds4_st(pb_next + 4u * i0, exp4(ynext[v]));
      }
    }
    for (int e = bfx0; e < bfx1; ++e) fold_row(p.bwd.fix[e], vrow, aACC, Hs, K);
    f4 b4[JV];
    uint32_t fpk[JV][2];
    part = 0.f;
    float part_u = 0.f, part_ab = 0.f, part_g = 0.f;
    const float asum_up = PURE ? 0.f : ldsf(aAsum + 4u * (t + 1));
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      b4[j] = mk4(0.f);
      if (j < planes) {
        const u4 fs = bld4u(r_fs, own16, j * kPlane);
        const f4 ws = bld4(r_ws, own16, j * kPlane);
        const f4 cp = leaky * bld4(r_pi, own16, j * kPlane);
        f4 aup = mk4(0.f);
        if (!PURE) aup = ALPHA_LDS ? lds4(aAL + own16 + j * kPlane) : bld4(make_rsrc(hist + (int64_t)(t + 1) * hist_step, 4u * Hs), own16, j * kPlane);
        f4 a = own_rows(vrow, j);
        const f4 al = areg[j];
        auto one = [&](uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx, float ax) __attribute__((always_inline)) {
          const float ps_ws = ldsf(pb_cur + (fsx >> 16)) * wsx;
          if constexpr (!PURE) {
            const float selfpart = ps_ws * alx * inv_as;
            const float bos = kGammaScale * bo;
            gamma_add_a(aGM + (fsx >> 16), bos * selfpart);
            gamma_add_a(aGM + (fsx & 0xffffu), bos * fmaxf((aupx - cpx * asum_up) - selfpart, 0.f));
          }
          return fmaf(ps_ws, bo, ax);  // the self-loop arc's term of U_t(g) / of beta'_t(g) asum_t
        };
        a.x = one(fs.x, ws.x, bown[j].x, al.x, aup.x, cp.x, a.x);
        a.y = one(fs.y, ws.y, bown[j].y, al.y, aup.y, cp.y, a.y);
        a.z = one(fs.z, ws.z, bown[j].z, al.z, aup.z, cp.z, a.z);
        a.w = one(fs.w, ws.w, bown[j].w, al.w, aup.w, cp.w, a.w);
        b4[j] = PURE ? a : a * inv_as;
        fpk[j][0] = (fs.x & 0xffffu) | (fs.y << 16);
        fpk[j][1] = (fs.z & 0xffffu) | (fs.w << 16);
        part += hsum(cp * b4[j]);
        if (PURE) part_u += hsum(a);
        if (!PURE && t == 0) part_ab += hsum(al * b4[j]);
      }
    }
    float inv_n = 1.f;
    if (PURE) {
      // ROLE B, first phase: n_t = sum_h U_t(h) / H; B'_t = U_t / n_t; leaky sum of B'_t
      block_sum2(part, part_u, aRed, wave, lane);
      const float n = part_u * inv_h;
      inv_n = __builtin_amdgcn_rcpf(n);
      bsum = part * inv_n;
      if (tid == 0) bn[t] = __builtin_amdgcn_rcpf(inv_n);  // (the normaliser actually applied)
    } else {
      bsum = block_sum_a(part, aRed, wave, lane);  // its barrier also completes gamma_t
    }
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
          if (kDeferDeriv && t > 0)
            lds4_st(pb_cur + 4u * i0, o);
          else
            row_st(drow, own16 + v * kPlane, p.d_vec, o);
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
        return;
      }
    }
    if (!ALPHA_LDS) {
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) lds4_st(kPB + 4u * i0, exp4(ynext[v]));
      }
      __syncthreads();
    }
    const rsrc_t brow = make_rsrc(bhist + (int64_t)t * hist_step, PURE ? 4u * Hs : 0u);  // (second phase: stores vanish)
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        const f4 b = b4[j] * inv_n + bsum;
        bown[j] = b;
        const f4 yv = f4{b.x * ldsf(pb_next + (fpk[j][0] & 0xffffu)), b.y * ldsf(pb_next + (fpk[j][0] >> 16)),
                         b.z * ldsf(pb_next + (fpk[j][1] & 0xffffu)), b.w * ldsf(pb_next + (fpk[j][1] >> 16))};
        lds4_st(kA0 + own16 + j * kPlane, yv);
        if (PURE) bst4(brow, own16 + j * kPlane, b);  // ROLE B: B_t for role F (rows M..T-1)
        if (!PURE && ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, areg[j]);
      }
#pragma unroll
    for (int v = 0; v < PV; ++v) ycur[v] = ynext[v];
    const uint32_t tmp = pb_cur;
    pb_cur = pb_next;
    pb_next = tmp;
  };

  // ---- first phase: frames T-1..M, rows B_T..B_M stored
  for (int t = T - 1; t >= M; --t) frame(t, std::true_type());
  publish(q.sync + 4 + 2 * s + 1, tid);
  const bool partner_ok = await(q.sync + 4 + 2 * s, tid, q.aScr + 4u);
  {
    // asum_0..M from role F; c_M = 1 / sum_g alpha_M(g) B_M(g) with alpha_M = alpha'_M - leaky pi asum_M; from here
    // on beta = c_M B: Kaldi's scale, and the frames below are the fused kernel's
    for (int i = (int)tid; i <= M; i += kThreads) ldsf_st(aAsum + 4u * (uint32_t)i, vload_f32(fn + i));
    __syncthreads();
    const float asum_M = ldsf(aAsum + 4u * (uint32_t)M);
    const rsrc_t hist_M = make_rsrc(hist + (int64_t)M * hist_step, 4u * Hs);
    float d = 0.f;
    f4 aM[JV];
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      aM[j] = mk4(0.f);
      if (j < planes) {
        aM[j] = bld4(hist_M, own16, j * kPlane);
        d += hsum((aM[j] - (leaky * bld4(r_pi, own16, j * kPlane)) * asum_M) * bown[j]);
      }
    }
    d = block_sum_a(d, aRed + 4u * kWaves, wave, lane);
    const float c = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        bown[j] = bown[j] * c;
        lds4_st(kA0 + own16 + j * kPlane, lds4(kA0 + own16 + j * kPlane) * c);  // Y_{M-1}
        if (ALPHA_LDS) lds4_st(aAL + own16 + j * kPlane, aM[j]);               // alpha'_M
      }
  }
  // ---- second phase: frames M-1..0, den_tied_kernel's backward frame
  for (int t = M - 1; t >= 0; --t) frame(t, std::false_type());
  if (!partner_ok && tid == 0) p.seq_ab[s] = __builtin_nanf("");
}

template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, int RESF, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_mitm_kernel(const DenParams p, const MitmParams q) {
  // ticket -> (sequence, role): whoever starts next becomes the partner of the last unpaired workgroup
  if (threadIdx.x == 0)
    *reinterpret_cast<lds_u *>(q.aScr) = __hip_atomic_fetch_add((gu32 *)q.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const uint32_t ticket = __builtin_amdgcn_readfirstlane(*reinterpret_cast<lds_u *>(q.aScr));
  const int s = (int)(ticket >> 1);
  if (s >= p.S) return;
  if ((ticket & 1u) == 0u)
    mitm_forward<JV, PV, ACCUM, RESF>(p, q, s);
  else
    mitm_backward<JV, PV, ALPHA_LDS, ACCUM, RESB>(p, q, s);
}

// resident chunks per role (the gamma frames of role F carry B_t and y_{t-1} next to the forward pass)
#ifndef TC_MITM_RF
#define TC_MITM_RF 4
#endif
#ifndef TC_MITM_RB
#define TC_MITM_RB 4
#endif
#ifndef TC_MITM_RF3
#define TC_MITM_RF3 2
#endif
#ifndef TC_MITM_RF4
#define TC_MITM_RF4 2
#endif
#ifndef TC_MITM_RB4
#define TC_MITM_RB4 2
#endif
#ifndef TC_MITM_RFP
#define TC_MITM_RFP 2
#endif
#ifndef TC_MITM_RBP
#define TC_MITM_RBP 2
#endif
constexpr int mitm_res_fwd(int jv, int pv) {
  return jv == kJvSmall ? (pv == kPvSmall ? TC_MITM_RF : TC_MITM_RFP) : pv != kPvSmall ? 0 : jv == kJvMid ? TC_MITM_RF3 : TC_MITM_RF4;
}
constexpr int mitm_res_bwd(int jv, int pv) {
  return jv == kJvSmall ? (pv == kPvSmall ? TC_MITM_RB : TC_MITM_RBP) : pv != kPvSmall ? 0 : jv == kJvMid ? 2 : TC_MITM_RB4;
}

template <int JV, int PV>
int launch_mitm_jp(const DenParams &p, const MitmParams &q, int accumulate, size_t lds_bytes, hipStream_t stream) {
  constexpr int RF = mitm_res_fwd(JV, PV), RB = mitm_res_bwd(JV, PV);
  const bool al = p.L.alpha_in_lds;
  void (*k)(const DenParams, const MitmParams) = nullptr;
  if (accumulate)
    k = al ? den_tied_mitm_kernel<JV, PV, true, true, RF, RB> : den_tied_mitm_kernel<JV, PV, false, true, RF, RB>;
  else
    k = al ? den_tied_mitm_kernel<JV, PV, true, false, RF, RB> : den_tied_mitm_kernel<JV, PV, false, false, RF, RB>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds_bytes));
  hipLaunchKernelGGL(k, dim3(2 * p.S), dim3(kThreads), lds_bytes, stream, p, q);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

size_t mitm_sync_bytes(int S) { return (size_t)(16 + 8 * S + 255) & ~(size_t)255; }

// Fits wherever the fused kernel fits with 16 more bytes of LDS, from two frames on.
bool mitm_fits(const DenLayout &L, int T) { return T >= 2 && (size_t)layout_lds_bytes(L, T) + 16u <= (size_t)kLdsLimitBytes; }

int launch_den_tied_mitm(const DenParams &p, uint32_t *sync, int accumulate, hipStream_t stream) {
  if (!p.deriv || !p.beta_hist || !p.fwd_norm || !p.bwd_norm || !sync || !mitm_fits(p.L, p.T)) return TC_ERR_UNSUPPORTED;
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T) + 16u;
  MitmParams q;
  q.sync = sync;
  q.M = p.T / 2;
  q.aScr = (uint32_t)layout_lds_bytes(p.L, p.T);
  TC_HIP_CHECK(hipMemsetAsync(sync, 0, mitm_sync_bytes(p.S), stream));
  const int JV = p.L.JV, PV = p.L.PV;
#define TC_DISPATCH(J, V) \
  if (JV == J && PV == V) return launch_mitm_jp<J, V>(p, q, accumulate, lds, stream);
  TC_DISPATCH(kJvSmall, kPvSmall)
  TC_DISPATCH(kJvSmall, kPvMid)
  TC_DISPATCH(kJvSmall, kPvLarge)
  TC_DISPATCH(kJvMid, kPvSmall)
  TC_DISPATCH(kJvMid, kPvMid)
  TC_DISPATCH(kJvMid, kPvLarge)
  TC_DISPATCH(kJvLarge, kPvSmall)
  TC_DISPATCH(kJvLarge, kPvMid)
  TC_DISPATCH(kJvLarge, kPvLarge)
#undef TC_DISPATCH
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
