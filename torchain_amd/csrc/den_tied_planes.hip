// Fused denominator forward-backward for TIED graphs of 16385..28672 positions on gfx950: the "plane-wise" form of
// den_tied_kernel.hip (same mapping: one workgroup = one sequence = one CU, owner-computes schedules, the frames of
// den_tied_frames.h).  This is the size class of the den.fst the reference's recipe really loads (a pruned phone LM with
// Kaldi's default 2000 extra LM states: example/chime5/train_faster.py:91 -> src/my_lib_example.cpp:129-134 takes any
// FST), which until round 5 fell onto the streamed kernels (den_slab_kernel.hip: 18 ms per 256 x 150 batch against 1.4 ms
// for the 8192-state C3 graph).
//
// What changes beyond 16384 positions: the gather source alone is 4 bytes per position of the CU's 160 KB (96 KB at 24576
// positions), so neither the row sums of all states (another 4 bytes per position) nor a second exp(y) buffer nor parked
// alpha' fit, and a thread owns 20-28 states: held in registers the way the smaller instantiations hold them (alpha_t,
// beta_{t+1}, alpha'_t, beta'_t, the forward pdfs) they would be ~100 registers.  So the frame is taken a PLANE (one float4
// of states per thread, 4096 positions) at a time: request the plane's tables and history values, walk the plane's four
// rows, run its per-state pass; the schedule (schedule_owner.cpp, `planewise`) cuts every wave's stream into sub-streams
// -- the wave's secondary rows first, then one per plane -- each padded to whole chunks, with row-end mask words of its
// own, and the row sums of all planes share four accumulator rows per wave.  Cells carry 16-bit POSITIONS (byte offset =
// one SDWA shift).  See TiedSeq::forward_frame_pw / backward_frame_pw.
#include "den_tied_frames.h"

namespace tc {

namespace {

// JV = the graph's planes exactly (5, 6 or 7: den_layout.cpp compute_layout_planes)
template <int JV, bool ACCUM, bool WANT_DERIV>
__global__ __launch_bounds__(kThreads) void den_tied_planes_kernel(const DenParams p) {
  TiedSeq<JV, kPvSmall, false, ACCUM, 0, 0, false, true> q(p, (int)blockIdx.x, 0);
  const int T = q.T;
  // ---- forward: alpha'_0, frames 1..T, total probability
  q.forward_begin();
  for (int t = 1; t <= T; ++t) q.forward_frame_pw(t);
  q.forward_store_row(T);
  const float tot = q.forward_total(0.0);
  if (!WANT_DERIV) return;
  // ---- backward: beta'_T = 1 / tot, frames T-1..0 with gamma
  q.template backward_begin<false>(__builtin_amdgcn_rcpf(tot));
  for (int t = T - 1; t > 0; --t) q.backward_frame_pw(t);
  q.backward_frame_pw(0);
}

template <int JV>
int launch_planes(const DenParams &p, int accumulate, size_t lds, hipStream_t stream) {
  void (*k)(const DenParams) = nullptr;
  if (!p.deriv)
    k = den_tied_planes_kernel<JV, false, false>;
  else
    k = accumulate ? den_tied_planes_kernel<JV, true, true> : den_tied_planes_kernel<JV, false, true>;
  TC_HIP_CHECK(allow_dynamic_lds((const void *)k, lds));
  hipLaunchKernelGGL(k, dim3(p.S), dim3(kThreads), lds, stream, p);
  TC_HIP_CHECK(hipGetLastError());
  return TC_OK;
}

}  // namespace

int launch_den_tied_planes(const DenParams &p, int accumulate, hipStream_t stream) {
  const size_t lds = (size_t)layout_lds_bytes(p.L, p.T);
  if (!p.L.planewise || lds > (size_t)kLdsLimitBytes || p.L.PV != kPvSmall) return TC_ERR_UNSUPPORTED;
  switch (p.L.JV) {
    case 5: return launch_planes<5>(p, accumulate, lds, stream);
    case 6: return launch_planes<6>(p, accumulate, lds, stream);
    case 7: return launch_planes<7>(p, accumulate, lds, stream);
  }
  return TC_ERR_UNSUPPORTED;
}

}  // namespace tc
