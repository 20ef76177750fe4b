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
// The frames are den_tied_frames.h's, the ones the fused kernel runs (same LDS layout, same resident chunks): role F =
// forward_frame<GAMMA>, role B = backward_frame<PURE>; this file holds the pairing, the hand-over and the two drivers.  [K] = kaldi chain-denominator.cc, reached by the reference through
// src/my_lib_chain.cpp:129-131.
#include "den_handover.h"
#include "den_tied_frames.h"

namespace tc {

namespace {

// =========================================================================================================
// ROLE F: the forward frame (den_tied_frames.h); frames M+1..T also form gamma_{t-1}.
// =========================================================================================================
template <int JV, int PV, bool ACCUM, int RESF>
__device__ __forceinline__ void mitm_forward(const DenParams &p, const MitmParams &mq, int s) {
  TiedSeq<JV, PV, false, ACCUM, RESF, 0, true> q(p, s, mq.M);
  const int T = q.T, M = q.M;
  const uint32_t own16 = q.own16;
  q.forward_begin();
  // ---- first phase: frames 1..M (rows 0..M-1 of the alpha' history leave under the walks, row M here)
  for (int t = 1; t <= M; ++t) q.template forward_frame<false>(t);
  q.forward_store_row(M);
  publish(mq.sync + 4 + 2 * s, q.tid);
  const bool partner_ok = await(mq.sync + 4 + 2 * s + 1, q.tid, mq.aScr + 4u);
  q.forward_unpark();  // (await's barrier is behind every wave's last table read; the block sum below publishes the zeros)
  {
    // c_M = 1 / sum_g alpha_M(g) B_M(g);  c^_{M+1} = c_M asum_M / n_M;  B_{M+1} for the next frame
    const rsrc_t bM = make_rsrc(q.bhist + (int64_t)M * q.hist_step, 4u * q.Hs);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < q.planes) d += hsum(q.v4[j] * bld4(bM, own16, j * kPlane));
    d = block_sum_a(d, q.aRed + 4u * kWaves, q.wave, q.lane);
    q.chat = __builtin_amdgcn_rcpf(d) * q.asum * __builtin_amdgcn_rcpf(vload_f32(q.bn + M));
    const rsrc_t brow = make_rsrc(q.bhist + (int64_t)(M + 1) * q.hist_step, 4u * q.Hs);
#pragma unroll
    for (int j = 0; j < JV; ++j) q.bt[j] = bld4(brow, own16, j * kPlane);
  }
  // ---- second phase: frames M+1..T with gamma_{t-1}
  for (int t = M + 1; t <= T; ++t) q.template forward_frame<true>(t);
  q.forward_total(partner_ok ? 0.0 : (double)__builtin_nanf(""));
}

// =========================================================================================================
// ROLE B: frames T-1..M with normalisers of its own and no gamma; then den_tied_kernel's backward frame.
// =========================================================================================================
template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, int RESB>
__device__ __forceinline__ void mitm_backward(const DenParams &p, const MitmParams &mq, int s) {
  TiedSeq<JV, PV, ALPHA_LDS, ACCUM, 0, RESB, true> q(p, s, mq.M);
  const int T = q.T, M = q.M;
  const uint32_t own16 = q.own16;
  // B'_T = 1; B_T = B'_T + leaky sum_h pi(h) B'_T(h)
  q.template backward_begin<true>(1.0f);
  // ---- first phase: frames T-1..M, rows B_T..B_M stored
  for (int t = T - 1; t >= M; --t) q.template backward_frame<true>(t, 0);
  if (M < T) q.backward_store_row(M);  // (rows T-1..M+1 left under the walks; row T in backward_begin)
  publish(mq.sync + 4 + 2 * s + 1, q.tid);
  const bool partner_ok = await(mq.sync + 4 + 2 * s, q.tid, mq.aScr + 4u);
  {
    // asum_0..M from role F; c_M = 1 / sum_g alpha_M(g) B_M(g) with alpha_M = alpha'_M - leaky pi asum_M; from here
    // on beta = c_M B: Kaldi's scale, and the frames below are the fused kernel's
    for (int i = (int)q.tid; i <= M; i += kThreads) ldsf_st(q.aAsum + 4u * (uint32_t)i, vload_f32(q.fn + i));
    __syncthreads();
    const float asum_M = ldsf(q.aAsum + 4u * (uint32_t)M);
    const rsrc_t hist_M = make_rsrc(q.hist + (int64_t)M * q.hist_step, 4u * q.Hs);
    float d = 0.f;
    f4 aM[JV];
#pragma unroll
    for (int j = 0; j < JV; ++j) {
      aM[j] = mk4(0.f);
      if (j < q.planes) {
        aM[j] = bld4(hist_M, own16, j * kPlane);
        d += hsum((aM[j] - (q.leaky * bld4(q.r_pi, own16, j * kPlane)) * asum_M) * q.bown[j]);
      }
    }
    d = block_sum_a(d, q.aRed + 4u * kWaves, q.wave, q.lane);
    const float c = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int j = 0; j < JV; ++j)
      if (j < q.planes) {
        q.bown[j] = q.bown[j] * c;
        lds4_st(q.kA0 + own16 + j * kPlane, lds4(q.kA0 + own16 + j * kPlane) * c);  // Y_{M-1}
        if (ALPHA_LDS) lds4_st(q.aAL + own16 + j * kPlane, aM[j]);                  // alpha'_M
      }
  }
  // ---- second phase: frames M-1..0, the fused kernel's backward frame
  for (int t = M - 1; t >= 0; --t)
    if (q.template backward_frame<false>(t, M)) break;
  if (!partner_ok && q.tid == 0) p.seq_ab[s] = __builtin_nanf("");
}

template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, int RESF, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_mitm_kernel(const DenParams p, const MitmParams q) {
  const uint32_t ticket = take_ticket(q);
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
#define TC_CALL(J, V) launch_mitm_jp<J, V>(p, q, accumulate, lds, stream)
  TC_TIED_DISPATCH(TC_CALL)
#undef TC_CALL
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
