// Tied graphs, small batches: forward and backward recursion of a sequence on TWO CUs at the same time.
//
// The fused kernel (den_tied_kernel.hip) gives a sequence one CU and runs its 2 T frames one after the other; a
// batch of 64 sequences -- the reference recipe's own (example/chime5/train_faster.py) -- then leaves three
// quarters of an MI355X idle for the whole launch.  The frame recursions cannot be split, but they do not depend
// on each other: [K] BetaDashGeneralFrame needs alpha only for gamma, and divides by the forward pass's
// normaliser asum_t only to keep its numbers in range.  So, for batches of at most half the chip's CUs:
//   launch 1 (caller's stream)  den_tied_kernel, forward only: alpha' history, log-prob, asum_0..asum_T, tot
//   launch 2 (side stream)      den_tied_bwd_kernel below: the SAME backward recursion with normalisers n_t of
//                               its own, B'_T = 1:   U_t(h) = sum_out w * B_{t+1}(g) p_t(f(g)) + p_t(s(h)) w_s(h) B_{t+1}(h)
//                                                    B'_t = U_t / n_t,  n_t = sum_h U_t(h) / H,  B_t = B'_t + leaky * sum_h pi(h) B'_t(h)
//                               The recursion is linear, so beta_t = c_t * B_t with c_T = 1 / tot and
//                               c_t = c_{t+1} * n_t / asum_t.  Writes the B history and n_t.
//   launch 3 (caller's stream, after both)  den_tied_combine_kernel: gamma_t and the derivative row of every
//                               (sequence, frame) from alpha'_t, alpha'_{t+1}, c_{t+1} B_{t+1} and y_t with the
//                               fused kernel's per-state formulas -- every frame independent, all CUs busy.
// c_t is formed in double from the logs of the stored normalisers (no invariant is assumed: the two checks of
// [K] BetaGeneralFrameDebug(0) still measure how well the two recursions agree).
#include "den_tied_frames.h"

namespace tc {

namespace {

// two block sums behind one barrier
__device__ __forceinline__ void block_sum2_a(float &v1, float &v2, uint32_t red, int wave, uint32_t lane) {
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

// ---- the backward recursion alone -------------------------------------------------------------------
// den_tied_frames.h's pure backward frame (normalisers of its own, no gamma) over all T frames.  LDS: [exp(y) buffer |
// gather source A0 | row sums ACC | second exp(y) buffer | red] -- the fused kernel's regions without gamma and
// alpha'_{t+1} (the second exp(y) buffer sits where the fused layout has gamma), so graphs that get the fused kernel's
// tight layout fit here with both exp(y) buffers (split_bwd_layout below).
template <int JV, int PV, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_bwd_kernel(const DenParams p) {
  TiedSeq<JV, PV, true, false, 0, RESB, true> q(p, (int)blockIdx.x, 0);
  const int T = q.T;
  q.template backward_begin<true>(1.0f);  // B'_T = 1 on the real states, B_T = B'_T + leaky * sum(pi): row T
  for (int t = T - 1; t >= 0; --t) q.template backward_frame<true>(t, 0);  // rows T-1..1 leave under the walks
  // row 0 holds B'_0 = B_0 - leaky sum (the alpha'.beta' check of [K] BetaGeneralFrameDebug(0) needs it; gamma_0 needs B_1)
  const rsrc_t hist_0 = make_rsrc(q.bhist, 4u * q.Hs);
#pragma unroll
  for (int j = 0; j < JV; ++j)
    if (j < q.planes) bst4(hist_0, q.own16 + j * kPlane, q.bown[j] - q.bsum);
}

// ---- gamma and the derivative from the two histories ----------------------------------------------------
// One workgroup = one sequence x a run of consecutive frames (descending, so alpha'_{t+1} of a frame is the
// alpha'_t its predecessor loaded); the per-state tables stay in registers for the run.  Per state g, with
// beta = c_{t+1} * B_{t+1} -- the fused kernel's formulas (den_tied_kernel.hip, backward per-state pass):
//   selfpart = p_t(s(g)) w_s(g) alpha'_t(g) / asum_t
//   gamma_t(s(g)) += beta(g) * selfpart
//   gamma_t(f(g)) += beta(g) * max(alpha_{t+1}(g) - selfpart, 0),   alpha_{t+1} = alpha'_{t+1} - leaky pi asum_{t+1}
// accumulated in the same unsigned fixed point, so the sums do not depend on the order of the adds.
template <int JV, int PV, bool ACCUM>
__global__ __launch_bounds__(kThreads) void den_tied_combine_kernel(const DenParams p, int groups) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x, grp = blockIdx.y;
  const int P = p.P, S = p.S, T = p.T;
  const int Hs = p.L.Hs, Ps = p.L.Ps;
  const int planes = Hs / (4 * kThreads);
  const uint32_t own16 = 16u * tid;
  constexpr uint32_t kPB = 0u;
  constexpr uint32_t kGM = PV * 16u * kThreads;
  const uint32_t aC = kGM + 4u * (uint32_t)((Ps + 3) & ~3);          // c_0 .. c_T, float
  const uint32_t aD = (aC + 4u * (uint32_t)(T + 2) + 7u) & ~7u;      // log n_u - log asum_u, double
  const uint32_t aRed = aD + 8u * (uint32_t)(T + 1);
  const uint32_t tab_bytes = 4u * (uint32_t)(Hs + 4), row_bytes = 4u * (uint32_t)P;
  const rsrc_t r_pi = make_rsrc(p.pi, tab_bytes), r_fs = make_rsrc(p.tied_fs, tab_bytes), r_ws = make_rsrc(p.tied_w, tab_bytes);
  const int64_t hist_step = (int64_t)S * Hs;
  const float *const ahist = p.alpha_hist + (int64_t)s * Hs;
  const float *const bhist = p.beta_hist + (int64_t)s * Hs;
  const float *const fn = p.fwd_norm + (int64_t)s * (T + 2);
  const float *const bn = p.bwd_norm + (int64_t)s * (T + 1);
  typedef __attribute__((address_space(3))) double lds_d;

  // c_t = exp(-log tot + sum_{u >= t} (log n_u - log asum_u)), in double
  for (int u = (int)tid; u < T; u += kThreads) *reinterpret_cast<lds_d *>(aD + 8u * (uint32_t)u) = log((double)bn[u]) - log((double)fn[u]);
  __syncthreads();
  const double log_tot = log((double)fn[T + 1]);
  for (int u = (int)tid; u <= T; u += kThreads) {
    double acc = -log_tot;
    for (int v = T - 1; v >= u; --v) acc += *reinterpret_cast<lds_d *>(aD + 8u * (uint32_t)v);
    ldsf_st(aC + 4u * (uint32_t)u, (float)exp(acc));
  }
  // the per-state tables stay in registers for the run when a thread owns 8 states; with 16 they are re-read
  // (L2) every frame
  constexpr bool kTablesResident = JV <= 2;
  u4 fs[JV];
  f4 ws[JV], cp[JV];
#pragma unroll
  for (int j = 0; j < JV; ++j) {
    fs[j] = u4{0u, 0u, 0u, 0u};
    ws[j] = cp[j] = mk4(0.f);
    if (kTablesResident && j < planes) {
      fs[j] = bld4u(r_fs, own16, j * kPlane);
      ws[j] = bld4(r_ws, own16, j * kPlane);
      cp[j] = p.leaky * bld4(r_pi, own16, j * kPlane);
    }
  }
  const int t_lo = (int)((int64_t)T * grp / groups), t_hi = (int)((int64_t)T * (grp + 1) / groups);
  f4 aup[JV];
#pragma unroll
  for (int j = 0; j < JV; ++j)
    aup[j] = (j < planes && t_hi > t_lo) ? bld4(make_rsrc(ahist + (int64_t)t_hi * hist_step, 4u * Hs), own16, j * kPlane) : mk4(0.f);
  for (int t = t_hi - 1; t >= t_lo; --t) {
    f4 yv[PV], al[JV], bt[JV];
    {
      const rsrc_t yrow = make_rsrc(p.y + ((int64_t)t * S + s) * p.y_stride, row_bytes);
      const rsrc_t arow = make_rsrc(ahist + (int64_t)t * hist_step, 4u * Hs);
      const rsrc_t brow = make_rsrc(bhist + (int64_t)(t + 1) * hist_step, 4u * Hs);
#pragma unroll
      for (int v = 0; v < PV; ++v) yv[v] = row_ld(yrow, own16 + v * kPlane, p.y_vec);
#pragma unroll
      for (int j = 0; j < JV; ++j) {
        al[j] = j < planes ? bld4(arow, own16, j * kPlane) : mk4(0.f);
        bt[j] = j < planes ? bld4(brow, own16, j * kPlane) : mk4(0.f);
      }
    }
#pragma unroll
    for (int v = 0; v < PV; ++v) {
      const int i0 = 4 * ((int)tid + kThreads * v);
      if (i0 < Ps) {
        lds4_st(kPB + 4u * i0, exp4(yv[v]));
        lds4_st(kGM + 4u * i0, mk4(0.f));
      }
    }
    __syncthreads();  // exp(y_t) complete, gamma zero (and, first time round, the c_t)
    const float asum_t = fn[t], asum_up = fn[t + 1];
    const float inv_as = __builtin_amdgcn_rcpf(asum_t);
    const float c_up = ldsf(aC + 4u * (uint32_t)(t + 1));
    float part_ab = 0.f, part_g = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < planes) {
        auto one = [&](uint32_t fsx, float wsx, float bo, float alx, float aupx, float cpx) {
          const float ps_ws = ldsf(kPB + (fsx >> 16)) * wsx;
          const float selfpart = ps_ws * alx * inv_as;
          const float bos = kGammaScale * bo;
          gamma_add_a(kGM + (fsx >> 16), bos * selfpart);
          gamma_add_a(kGM + (fsx & 0xffffu), bos * fmaxf((aupx - cpx * asum_up) - selfpart, 0.f));
        };
        const f4 b = bt[j] * c_up;
        const u4 fsj = kTablesResident ? fs[j] : bld4u(r_fs, own16, j * kPlane);
        const f4 wsj = kTablesResident ? ws[j] : bld4(r_ws, own16, j * kPlane);
        const f4 cpj = kTablesResident ? cp[j] : p.leaky * bld4(r_pi, own16, j * kPlane);
        one(fsj.x, wsj.x, b.x, al[j].x, aup[j].x, cpj.x);
        one(fsj.y, wsj.y, b.y, al[j].y, aup[j].y, cpj.y);
        one(fsj.z, wsj.z, b.z, al[j].z, aup[j].z, cpj.z);
        one(fsj.w, wsj.w, b.w, al[j].w, aup[j].w, cpj.w);
        if (t == 0) part_ab += hsum(al[j] * (bld4(make_rsrc(bhist, 4u * Hs), own16, j * kPlane) * ldsf(aC)));
        aup[j] = al[j];
      }
    __syncthreads();  // gamma_t complete
    {
      const rsrc_t drow = make_rsrc(p.deriv + ((int64_t)t * S + s) * p.deriv_stride, row_bytes);
#pragma unroll
      for (int v = 0; v < PV; ++v) {
        const int i0 = 4 * ((int)tid + kThreads * v);
        if (i0 < Ps) {
          const u4 gu = lds4u(kGM + 4u * i0);
          const f4 g = f4{(float)gu.x, (float)gu.y, (float)gu.z, (float)gu.w} * kGammaInvScale;
          if (t == 0) part_g += hsum(g);
          f4 o = p.deriv_weight * g - p.l2_scale * yv[v];
          if (ACCUM) o += row_ld(drow, own16 + v * kPlane, p.d_vec);
          row_st(drow, own16 + v * kPlane, p.d_vec, o);
        }
      }
    }
    if (t == 0) {
      // [K] BetaGeneralFrameDebug(0): alpha'.beta' and sum(gamma) must both be ~1 per sequence
      const float ab = block_sum_a(part_ab, aRed, wave, lane);
      const float gs = block_sum_a(part_g, aRed + 4u * kWaves, wave, lane);
      if (tid == 0) {
        p.seq_ab[s] = ab;
        p.seq_gsum[s] = gs;
      }
    }
  }
}

template <int JV, int PV>
int launch_bwd_jp(const DenParams &p, size_t lds_bytes, hipStream_t stream) {
#ifndef TC_SPLIT_RB4
#define TC_SPLIT_RB4 2
#endif
#ifndef TC_SPLIT_RB3
#define TC_SPLIT_RB3 2
#endif
  constexpr int RB = JV > 3 ? (PV == 1 ? TC_SPLIT_RB4 : 0) : JV == 3 ? (PV == 1 ? TC_SPLIT_RB3 : 0) : PV == 1 ? 4 : 2;
  void (*k)(const DenParams) = den_tied_bwd_kernel<JV, PV, RB>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds_bytes));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds_bytes, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

// LDS of the combining pass: exp(y_t), gamma_t, c_0..c_T (float), the log ratios (double), reduction scratch
size_t combine_lds_bytes(int PV, int Ps, int T) {
  return (size_t)PV * 16 * kThreads + 4 * (size_t)((Ps + 3) & ~3) + 4 * (size_t)(T + 2) + 8 + 8 * (size_t)(T + 1) +
         4 * 3 * kWaves + 64;
}

template <int JV, int PV>
int launch_combine_jp(const DenParams &p, int accumulate, int groups, hipStream_t stream) {
  const size_t lds = combine_lds_bytes(PV, p.L.Ps, p.T);
  void (*k)(const DenParams, int) = accumulate ? den_tied_combine_kernel<JV, PV, true> : den_tied_combine_kernel<JV, PV, false>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(p.S, groups), dim3(kThreads), lds, stream, p, groups);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

// The backward-only kernel's LDS: the fused layout's exp(y), A0 and ACC regions as they are (compile-time bases),
// then the second exp(y) buffer and the reduction scratch.  Returns the bytes, or 0 if it does not fit.
size_t split_bwd_layout(DenLayout *L) {
  L->off_p2 = L->off_acc + L->acc_floats;
  L->off_red = L->off_p2 + L->Ps;
  L->off_asum = L->off_red + 4 * kWaves;  // (unused here)
  const size_t bytes = 4 * (size_t)L->off_asum;
  return bytes <= (size_t)kLdsLimitBytes ? bytes : 0;
}

bool split_bwd_fits(const DenLayout &L0, int T) {
  DenLayout L = L0;
  // (the combining pass keeps 12 bytes per frame in LDS: only sequences of many thousand frames fail this)
  return split_bwd_layout(&L) != 0 && combine_lds_bytes(L.PV, L.Ps, T) <= (size_t)kLdsLimitBytes;
}

int launch_den_tied_backward_only(const DenParams &p0, hipStream_t stream) {
  DenParams p = p0;
  const size_t lds = split_bwd_layout(&p.L);
  if (lds == 0) return TC_ERR_UNSUPPORTED;
  const int JV = p.L.JV, PV = p.L.PV;
#define TC_CALL_BWD(J, V) launch_bwd_jp<J, V>(p, lds, stream)
  TC_TIED_DISPATCH(TC_CALL_BWD)
#undef TC_CALL_BWD
  return TC_ERR_UNSUPPORTED;
}

int launch_den_tied_combine(const DenParams &p, int accumulate, int num_cus, hipStream_t stream) {
  const int JV = p.L.JV, PV = p.L.PV;
  // about two workgroups per CU: runs of T / groups frames
  int groups = (2 * num_cus + p.S - 1) / p.S;
  groups = groups < 1 ? 1 : groups > p.T ? p.T : groups;
#define TC_CALL_CMB(J, V) launch_combine_jp<J, V>(p, accumulate, groups, stream)
  TC_TIED_DISPATCH(TC_CALL_CMB)
#undef TC_CALL_CMB
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
