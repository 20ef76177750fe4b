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
#include "den_tied_frames.h"

namespace tc {

namespace {

// (the frames themselves: den_tied_frames.h, shared with den_tied_mitm.hip)
template <int JV, int PV, bool ALPHA_LDS, bool ACCUM, bool WANT_DERIV, int RESF, int RESB>
__global__ __launch_bounds__(kThreads) void den_tied_kernel(const DenParams p) {
  TiedSeq<JV, PV, ALPHA_LDS, ACCUM, RESF, RESB, false> q(p, (int)blockIdx.x, 0);
  const int T = q.T;
  // ---- forward: alpha'_0, frames 1..T, total probability
  q.forward_begin();
  for (int t = 1; t <= T; ++t) q.template forward_frame<false>(t);
  q.forward_store_row(T);
  q.stamps_flush(0);
  const float tot = q.forward_total(0.0);
  if (p.fwd_norm) {  // two-CU form: the normalisers the combining pass needs (den_tied_split.hip)
    float *const fn = p.fwd_norm + (int64_t)q.s * (T + 2);
    for (int i = (int)q.tid; i <= T; i += kThreads) fn[i] = ldsf(q.aAsum + 4u * (uint32_t)i);
    if (q.tid == 0) fn[T + 1] = tot;
  }
  if (!WANT_DERIV) return;
  // ---- backward: beta'_T = 1 / tot, frames T-1..0 with gamma
  q.template backward_begin<false>(__builtin_amdgcn_rcpf(tot));
  for (int t = T - 1; t > 0; --t) q.template backward_frame<false>(t, T);
  q.template backward_frame<false>(0, T);
  q.stamps_flush(128);
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
#define TC_CALL(J, V) launch_jp<J, V>(p, accumulate, lds, stream)
  TC_TIED_DISPATCH(TC_CALL)
#undef TC_CALL
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
